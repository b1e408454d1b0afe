// Launch tape: the kernel launches of one captured forward, replayed as PLAIN launches on a caller-chosen stream.
//
// Why: one 32-pair forward is ~270 launches; issued from Python (wrapper + ctypes + torch allocations) they cost 4-5 ms of host
// time per step - more than half of the 9 ms the GPU needs, and the reason the drop-in boundary (H2D + forward + package() on one
// Python thread) ran host-bound.  A hipGraph of the forward (MODEL.AMD.USE_HIP_GRAPH, round 2) cuts the submit time to < 1 ms but
// measured SLOWER end to end with several batches in flight (2199 vs 2790 pairs/s): whole-graph launches of different slots
// overlap less than eagerly launched streams.  The tape keeps both properties: the forward is captured ONCE into a hipGraph (which
// also gives every intermediate tensor a fixed address in the capture's private memory pool), the graph's nodes are read back with
// the graph-introspection API (kernel function, grid, block, dynamic LDS, argument block; memset / memcpy parameters), and a replay
// is a C loop of hipLaunchKernel calls on the slot's own stream - exactly what the eager path enqueues, minus Python.
//
// Order and streams: nodes are issued in a topological order of the captured graph that prefers the capture's creation order.  The
// capture's fork / join structure (the pixel pose net runs on a side stream next to the plane head: + 13 % throughput with four
// batches in flight, - 1 ms of latency at one pair per call) is kept: the DAG is cut into CHAINS (a node continues the chain of a
// predecessor that is still that chain's tail, otherwise it opens a new one); chain 0 is issued on the caller's stream, chain k > 0
// on the tape's own k-th side stream, and every edge between chains becomes an event record behind the producer and a stream wait
// in front of the consumer.  A replay starts by making the side streams wait for the caller's stream and ends by making the caller's
// stream wait for every side chain, so to the caller a replay behaves like work enqueued on its stream.  The argument blocks stay
// owned by the graph nodes: the graph must outlive the tape (the Python side keeps the torch.cuda.CUDAGraph object - created with
// keep_graph=True - next to the tape).
#include <algorithm>
#include <queue>
#include <vector>

#include "common.h"

namespace nps {

enum { TAPE_KERNEL = 0, TAPE_MEMSET = 1, TAPE_MEMCPY = 2, TAPE_NOP = 3 };

struct TapeOp {
    int kind;
    hipKernelNodeParams k;
    hipMemsetParams ms;
    hipMemcpy3DParms cp;
    int chain = 0;                 // 0 = the caller's stream, k > 0 = side stream k - 1
    int record = -1;               // event recorded behind this op (it has a successor on another chain)
    std::vector<int> waits;        // events the op's stream waits for first (predecessors on other chains)
};

struct Tape {
    std::vector<TapeOp> ops;
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> events;
    hipEvent_t start_ev = nullptr;
    std::vector<hipEvent_t> end_ev;      // one per side chain
    int n_kernel = 0, n_memset = 0, n_memcpy = 0, n_dropped = 0, n_chains = 1, max_chains = 1;
    ~Tape() {
        for (hipEvent_t e : events) (void)hipEventDestroy(e);
        for (hipEvent_t e : end_ev) (void)hipEventDestroy(e);
        if (start_ev) (void)hipEventDestroy(start_ev);
        for (hipStream_t q : side) (void)hipStreamDestroy(q);
    }
};

}  // namespace nps

#define TAPE_HIP(call, what)                                                                          \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            nps::set_error("launch tape: %s failed: %s", what, hipGetErrorString(e__));               \
            return (int)e__;                                                                          \
        }                                                                                             \
    } while (0)

extern "C" int nopesac_tape_create(void* hip_graph, void** tape_out, int32_t* counts4) {
    return nopesac_tape_create_ex(hip_graph, 4, tape_out, counts4);
}

extern "C" int nopesac_tape_create_ex(void* hip_graph, int max_streams, void** tape_out, int32_t* counts4) {
    using namespace nps;
    NPS_CHECK_ARG(hip_graph && tape_out && max_streams >= 1 && max_streams <= 17, "tape_create: bad arguments (1 <= max_streams <= 17: the caller's "
                  "stream + at most 16 side chains, the limit nopesac_tape_replay_on enforces)");
    hipGraph_t graph = (hipGraph_t)hip_graph;
    size_t n = 0;
    TAPE_HIP(hipGraphGetNodes(graph, nullptr, &n), "hipGraphGetNodes(count)");
    NPS_CHECK_ARG(n > 0, "tape_create: the graph has no nodes");
    std::vector<hipGraphNode_t> nodes(n);
    TAPE_HIP(hipGraphGetNodes(graph, nodes.data(), &n), "hipGraphGetNodes");
    size_t ne = 0;
    TAPE_HIP(hipGraphGetEdges(graph, nullptr, nullptr, &ne), "hipGraphGetEdges(count)");
    std::vector<hipGraphNode_t> from(ne ? ne : 1), to(ne ? ne : 1);
    if (ne) TAPE_HIP(hipGraphGetEdges(graph, from.data(), to.data(), &ne), "hipGraphGetEdges");
    // node handle -> index (sorted lookup table)
    std::vector<std::pair<hipGraphNode_t, int>> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = {nodes[i], (int)i};
    std::sort(idx.begin(), idx.end());
    auto find = [&](hipGraphNode_t h) -> int {
        auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(h, -1));
        return (it != idx.end() && it->first == h) ? it->second : -1;
    };
    std::vector<std::vector<int>> succ(n), pred(n);
    std::vector<int> indeg(n, 0);
    for (size_t e = 0; e < ne; ++e) {
        const int a = find(from[e]), b = find(to[e]);
        NPS_CHECK_ARG(a >= 0 && b >= 0, "tape_create: an edge names a node the graph does not list");
        succ[a].push_back(b);
        pred[b].push_back(a);
        ++indeg[b];
    }
    // Kahn's algorithm with a min-heap on the creation index: the capture's own order whenever it is a valid one
    std::priority_queue<int, std::vector<int>, std::greater<int>> ready;
    for (size_t i = 0; i < n; ++i)
        if (indeg[i] == 0) ready.push((int)i);
    std::vector<int> order;
    order.reserve(n);
    while (!ready.empty()) {
        const int u = ready.top();
        ready.pop();
        order.push_back(u);
        for (int v : succ[u])
            if (--indeg[v] == 0) ready.push(v);
    }
    NPS_CHECK_ARG(order.size() == n, "tape_create: the captured graph has a cycle (%zu of %zu nodes ordered)", order.size(), n);

    // ---- chains: a node continues the chain of a predecessor that is still the tail of its chain (lowest chain id first)
    std::vector<int> chain(n, -1), tail;                    // tail[c] = last node of chain c so far
    for (int u : order) {
        int best = -1;
        for (int q : pred[u])
            if (tail[chain[q]] == q && (best < 0 || chain[q] < best)) best = chain[q];
        if (best < 0) {
            if ((int)tail.size() < max_streams) { best = (int)tail.size(); tail.push_back(u); }
            else best = 0;                                   // out of streams: serialise onto the caller's stream (waits still apply)
        }
        chain[u] = best;
        tail[best] = u;
    }
    Tape* t = new Tape();
    t->n_chains = (int)tail.size();
    std::vector<int> op_of(n, -1);                           // node index -> position in t->ops
    for (int u : order) {
        hipGraphNodeType ty;
        hipError_t e = hipGraphNodeGetType(nodes[u], &ty);
        if (e != hipSuccess) {
            delete t;
            set_error("launch tape: hipGraphNodeGetType failed: %s", hipGetErrorString(e));
            return (int)e;
        }
        TapeOp op;                                           // (record = -1, no waits)
        op.kind = TAPE_NOP;
        memset(&op.k, 0, sizeof(op.k));
        memset(&op.ms, 0, sizeof(op.ms));
        memset(&op.cp, 0, sizeof(op.cp));
        if (ty == hipGraphNodeTypeKernel) {
            e = hipGraphKernelNodeGetParams(nodes[u], &op.k);
            if (e != hipSuccess || !op.k.func || (!op.k.kernelParams && !op.k.extra)) {
                delete t;
                set_error("launch tape: kernel node %d: parameters not readable (%s)", u, hipGetErrorString(e));
                return e != hipSuccess ? (int)e : NPS_E_ARG;
            }
            if (!op.k.kernelParams) {          // launched with a packed HIP_LAUNCH_PARAM_BUFFER: hipLaunchKernel cannot replay it
                delete t;
                set_error("launch tape: kernel node %d was launched with `extra` arguments (module launch): unsupported", u);
                return NPS_E_ARG;
            }
            {   // a handle hipLaunchKernel cannot resolve (a module / hiprtc function launched with kernelParams) must fail HERE, where
                // the caller still falls back to the whole-graph replay - not at the first replay
                hipFuncAttributes fa;
                const hipError_t fe = hipFuncGetAttributes(&fa, op.k.func);
                if (fe != hipSuccess) {
                    (void)hipGetLastError();
                    delete t;
                    set_error("launch tape: kernel node %d: function handle not launchable through hipLaunchKernel (%s)", u, hipGetErrorString(fe));
                    return NPS_E_ARG;
                }
            }
            op.kind = TAPE_KERNEL;
            ++t->n_kernel;
        } else if (ty == hipGraphNodeTypeMemset) {
            e = hipGraphMemsetNodeGetParams(nodes[u], &op.ms);
            if (e != hipSuccess || !op.ms.dst || op.ms.width == 0 || (op.ms.elementSize != 1 && op.ms.elementSize != 2 && op.ms.elementSize != 4)) {
                delete t;
                set_error("launch tape: memset node %d: parameters not readable / unsupported (%s)", u, hipGetErrorString(e));
                return e != hipSuccess ? (int)e : NPS_E_ARG;
            }
            if (op.ms.height > 1 && op.ms.elementSize != 1) {
                // 2-D memsets are replayed with hipMemset2DAsync, a BYTE fill: only byte-uniform 16 / 32-bit patterns are the same thing
                const unsigned v = op.ms.value, b0 = v & 0xffu;
                const bool uniform = op.ms.elementSize == 2 ? ((v >> 8) & 0xffu) == b0
                                                            : (((v >> 8) & 0xffu) == b0 && ((v >> 16) & 0xffu) == b0 && ((v >> 24) & 0xffu) == b0);
                if (!uniform) {
                    delete t;
                    set_error("launch tape: memset node %d: 2-D memset of %u-byte elements with a non-uniform byte pattern: unsupported", u, op.ms.elementSize);
                    return NPS_E_ARG;
                }
            }
            op.kind = TAPE_MEMSET;
            ++t->n_memset;
        } else if (ty == hipGraphNodeTypeMemcpy) {
            e = hipGraphMemcpyNodeGetParams(nodes[u], &op.cp);
            const bool ok = e == hipSuccess && op.cp.extent.width > 0 && op.cp.extent.height > 0 && op.cp.extent.depth > 0 &&
                            op.cp.srcPtr.ptr && op.cp.dstPtr.ptr && !op.cp.srcArray && !op.cp.dstArray;
            if (!ok) {                          // (1-D copy nodes do not expose 3-D parameters on every runtime)
                delete t;
                set_error("launch tape: memcpy node %d: parameters not readable as a 3-D copy (%s; extent %zu x %zu x %zu)", u,
                          hipGetErrorString(e), op.cp.extent.width, op.cp.extent.height, op.cp.extent.depth);
                return e != hipSuccess ? (int)e : NPS_E_ARG;
            }
            op.kind = TAPE_MEMCPY;
            ++t->n_memcpy;
        } else if (ty == hipGraphNodeTypeEmpty || ty == hipGraphNodeTypeEventRecord || ty == hipGraphNodeTypeWaitEvent) {
            op.kind = TAPE_NOP;                 // ordering-only node: nothing is launched, its edges still order its neighbours
            ++t->n_dropped;
        } else {
            delete t;
            set_error("launch tape: node %d has type %d (host / child graph / memory node): unsupported", u, (int)ty);
            return NPS_E_ARG;
        }
        op.chain = chain[u];
        op_of[u] = (int)t->ops.size();
        t->ops.push_back(op);
    }
    // ---- edges between chains -> events (one per producer), created now, re-recorded by every replay
    auto fail = [&](hipError_t e, const char* what) {
        set_error("launch tape: %s failed: %s", what, hipGetErrorString(e));
        delete t;
        return (int)e;
    };
    for (int u : order)
        for (int v : succ[u])
            if (chain[u] != chain[v]) {
                TapeOp& a = t->ops[op_of[u]];
                if (a.record < 0) {
                    hipEvent_t ev;
                    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                    if (e != hipSuccess) return fail(e, "hipEventCreateWithFlags");
                    a.record = (int)t->events.size();
                    t->events.push_back(ev);
                }
                t->ops[op_of[v]].waits.push_back(a.record);
            }
    if (t->n_chains > 1) {
        hipError_t e = hipEventCreateWithFlags(&t->start_ev, hipEventDisableTiming);
        if (e != hipSuccess) return fail(e, "hipEventCreateWithFlags");
        for (int c = 1; c < t->n_chains; ++c) {
            hipStream_t q;
            e = hipStreamCreateWithFlags(&q, hipStreamNonBlocking);
            if (e != hipSuccess) return fail(e, "hipStreamCreateWithFlags");
            t->side.push_back(q);
            hipEvent_t ev;
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) return fail(e, "hipEventCreateWithFlags");
            t->end_ev.push_back(ev);
        }
    }
    if (counts4) { counts4[0] = t->n_kernel; counts4[1] = t->n_memset; counts4[2] = t->n_memcpy; counts4[3] = t->n_chains; }
    *tape_out = t;
    return 0;
}

// Replay with the side chains on streams of the CALLER's choosing (side_streams[k] runs chain k + 1; missing / null entries fall back
// to the tape's own streams): which hardware queue a side chain shares decides how well several replays overlap
// (nopesac_amd/streams.py), and only the caller knows the queues of its streams.
extern "C" int nopesac_tape_replay_on(void* tape, void* stream, void* const* side_streams, int n_side) {
    using namespace nps;
    NPS_CHECK_ARG(tape && n_side >= 0 && (n_side == 0 || side_streams), "tape_replay: bad arguments");
    const Tape* t = (const Tape*)tape;
    hipStream_t caller = (hipStream_t)stream;
    hipStream_t side[16];
    const size_t ns = t->side.size();
    NPS_CHECK_ARG(ns <= 16, "tape_replay: more than 16 side chains");
    for (size_t c = 0; c < ns; ++c) side[c] = ((int)c < n_side && side_streams[c]) ? (hipStream_t)side_streams[c] : t->side[c];
    if (t->n_chains > 1) {                                   // the side streams start behind everything the caller has enqueued
        TAPE_HIP(hipEventRecord(t->start_ev, caller), "hipEventRecord(start)");
        for (size_t c = 0; c < ns; ++c) TAPE_HIP(hipStreamWaitEvent(side[c], t->start_ev, 0), "hipStreamWaitEvent(start)");
    }
    for (const TapeOp& op : t->ops) {
        hipStream_t st = op.chain == 0 ? caller : side[op.chain - 1];
        for (int w : op.waits) TAPE_HIP(hipStreamWaitEvent(st, t->events[w], 0), "hipStreamWaitEvent");
        hipError_t e = hipSuccess;
        if (op.kind == TAPE_NOP) {
        } else if (op.kind == TAPE_KERNEL) {
            e = hipLaunchKernel(op.k.func, op.k.gridDim, op.k.blockDim, op.k.kernelParams, op.k.sharedMemBytes, st);
        } else if (op.kind == TAPE_MEMSET) {
            if (op.ms.height > 1) e = hipMemset2DAsync(op.ms.dst, op.ms.pitch, (int)op.ms.value, op.ms.width * op.ms.elementSize, op.ms.height, st);
            else if (op.ms.elementSize == 4) e = hipMemsetD32Async((hipDeviceptr_t)op.ms.dst, (int)op.ms.value, op.ms.width, st);
            else if (op.ms.elementSize == 2) e = hipMemsetD16Async((hipDeviceptr_t)op.ms.dst, (unsigned short)op.ms.value, op.ms.width, st);
            else e = hipMemsetAsync(op.ms.dst, (int)op.ms.value, op.ms.width, st);
        } else {
            e = hipMemcpy3DAsync(&op.cp, st);
        }
        if (e != hipSuccess) {
            set_error("launch tape: replay of a %s failed: %s", op.kind == TAPE_KERNEL ? "kernel launch" : op.kind == TAPE_MEMSET ? "memset" : "memcpy",
                      hipGetErrorString(e));
            return (int)e;
        }
        if (op.record >= 0) TAPE_HIP(hipEventRecord(t->events[op.record], st), "hipEventRecord");
    }
    for (size_t c = 0; c < ns; ++c) {                        // the caller's stream continues behind every side chain
        TAPE_HIP(hipEventRecord(t->end_ev[c], side[c]), "hipEventRecord(end)");
        TAPE_HIP(hipStreamWaitEvent(caller, t->end_ev[c], 0), "hipStreamWaitEvent(end)");
    }
    return 0;
}

extern "C" int nopesac_tape_replay(void* tape, void* stream) { return nopesac_tape_replay_on(tape, stream, nullptr, 0); }

extern "C" int nopesac_tape_destroy(void* tape) {
    delete (nps::Tape*)tape;
    return 0;
}
