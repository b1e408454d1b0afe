// One LoFTR-style GNN layer of the plane matcher per launch, one workgroup per plane set (transformer/gnn.py:73-96,
// LocalFeatureTransformer.forward :117-138; called from matching_net/matching_head.py:43-133):
//
//     q = x Wq^T,  k = s Wk^T,  v = s Wv^T                      (s = x for 'self' layers, the other view's set for 'cross')
//     msg = softmax_keys(q k^T / sqrt(32)) v                     8 heads x 32, keys >= klen masked
//     msg = LN1(msg Wm^T)
//     out = x + LN2( relu([x | msg] W0^T) W2^T )
//
// The un-fused bf16 path issues 8-9 launches per layer and set group (234 launches for the 18 layers), each a tiny
// GEMM over <= 3200 rows that cannot fill the chip.  A workgroup owns a block of 64 query rows of one plane set (one block for
// nq <= 64, two for nq <= 128) and walks the source set in chunks of 64 keys with a running (flash-style) softmax, so the
// LDS footprint does not depend on nq; K / V of a chunk are projected by every query block that needs them (2x redundant at
// nq = 128, still one launch per layer).  With 64 x 256 operands the whole layer fits on one CU:
//   * x (and s) are parked in LDS as bf16 operand tiles [64][256+8];
//   * every projection is "LDS tile x fragment-major weights streamed from L2" (see pwchain.hip): weights are read
//     exactly once per workgroup, 1 KB per load instruction, no LDS;
//   * q stays in registers as MFMA operand fragments (accumulator -> operand layout by v_permlane32_swap);
//   * K is kept row-major and V transposed (the V projection is issued with swapped MFMA operands so that its accumulator
//     layout IS the transposed tile), attention runs entirely out of LDS, one head per wave, scores never leave registers;
//   * both LayerNorms reduce across the eight waves through 4 KB of LDS, the 512-wide hidden tile never leaves the CU.
// f32: residual stream x, LayerNorm statistics, all accumulation.  bf16: MFMA operands (as in the per-launch bf16 path).
// The 1/sqrt(32) score scale is folded into Wq by the packer.
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int GN_D = 256, GN_LD = GN_D + 8;          // operand tile row: 264 bf16 = 528 bytes
constexpr int GN_VLD = 64 + 8;                       // V^T row: 72 bf16 = 144 bytes
constexpr int GN_HLD = 512 + 8;                      // hidden tile row
constexpr int GN_TILE = 64 * GN_LD;                  // elements
constexpr int GN_RC = GN_TILE + GN_D * GN_VLD;       // K tile + V^T tile (elements); the hidden tile aliases it
static_assert(64 * GN_HLD <= GN_RC, "hidden tile must fit the K/V region");
constexpr size_t GN_LDS_BYTES = 2 * (size_t)(2 * GN_TILE + GN_RC) + 2 * 8 * 64 * sizeof(float);

struct GnnArgs {
    const float* x; const float* src; float* out;       // [sets][nq][256] f32; block b works on set x_off + b / src_off + b
    int x_off, src_off, out_off, nq;
    const int* qlen; const int* klen;                   // int32 per set (indexed like x / src), may be null
    const bf16_t* wq; const bf16_t* wk; const bf16_t* wv; const bf16_t* wm; const bf16_t* w0; const bf16_t* w2;   // fragment-major
    const float* g1; const float* b1; const float* g2; const float* b2;
    // weight prefetch for the NEXT launch (one pair per call: a layer's 1.28 MB arrive from HBM / the Infinity Cache through ONE
    // workgroup's 128 KB of loads in flight - 20 us of the 45-52 us a cold launch takes): workgroups >= n_work do no layer work; those
    // that sit on an XCD the next launch's workgroups will run on read a slice of the next layer's weights into that XCD's L2 and exit
    const bf16_t* pf[6];                                // next wq, wk, wv, wm, w0, w2 (fragment-major) or all null
    int n_work, pf_xcds;                                // layer workgroups of THIS launch; XCDs to warm (= min(8, next launch's workgroups))
};
constexpr int GN_PF_CHUNK = 512 * 16;                   // bytes one prefetch workgroup touches per step
constexpr int GN_PF_CHUNKS = (4 * 256 * 256 + 512 * 512 + 256 * 512) * 2 / GN_PF_CHUNK;      // 160
constexpr int GN_PF_PER_XCD = 16;                       // prefetch workgroups per warmed XCD (10 chunks = 80 KB each)

__device__ __forceinline__ unsigned int gn_pack2(float lo, float hi) {
    return (unsigned int)f32_to_bf16(lo) | ((unsigned int)f32_to_bf16(hi) << 16);
}

// Weight stream: every GEMM step of a wave is "one 32-channel column tile x 16 k-steps" = 16 fragment loads (16 KB per wave,
// fragment-major, straight from L2).  The steps of a layer form a fixed sequence; step s+1's fragments are issued into the
// other half of a two-deep register ring before step s's MFMAs start, across phase boundaries as well, so the L2 latency is
// covered by MFMA work (and by the second wave of the SIMD).
struct GnRing {
    bf16x8 f[2][16];
};
__device__ __forceinline__ void gn_issue(GnRing& ring, int buf, const bf16_t* __restrict__ w, int kf_total, int kf_off, int nt, int lane) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
        ring.f[buf][kk] = *reinterpret_cast<const bf16x8*>(w + ((long long)(nt * kf_total + kf_off + kk) * 64 + lane) * 8);
}
// acc[r] += A[r*32 + row][k] * W[tile][k] over the 16 k-steps held in ring.f[buf].  A: LDS tile (row-major, `lda` elements),
// already offset to its first column.  SWAP = false: lane holds token l&31 x 4-channel runs (weights are the MFMA row
// operand); SWAP = true: lane holds channel l&31 x 4-token runs (transposed result).
template <bool SWAP>
__device__ __forceinline__ void gn_gemm(const GnRing& ring, int buf, const bf16_t* A, int lda, f32x16 (&acc)[2], int lane) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(A + (r * 32 + l31) * lda + kk * 16 + half * 8);
            if constexpr (SWAP) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, ring.f[buf][kk], acc[r], 0, 0, 0);
            else acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.f[buf][kk], af, acc[r], 0, 0, 0);
        }
}

__device__ __forceinline__ void gn_zero(f32x16 (&acc)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
}

// acc (token l&31 x channels nt*32 + 8q + 4*half + e) -> bf16 tile [64][ld]
template <bool RELU>
__device__ __forceinline__ void gn_store_tile(const f32x16 (&acc)[2], bf16_t* T, int ld, int nt, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            us4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[r][4 * q + e];
                if (RELU) v = v > 0.f ? v : 0.f;
                o[e] = f32_to_bf16(v);
            }
            *reinterpret_cast<us4*>(T + (r * 32 + l31) * ld + nt * 32 + 8 * q + 4 * half) = o;
        }
}

// LayerNorm over the 256 channels of every row; a row's values are spread over the 8 waves (32 channels each).
// Two-pass (mean, then squared deviations) like F.layer_norm; leaves the normalised, affine-transformed values in acc.
__device__ __forceinline__ void gn_layernorm(f32x16 (&acc)[2], const float* __restrict__ gamma, const float* __restrict__ beta,
                                             float* red, int wave, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
    float mean[2], rstd[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float* rp = red + pass * 8 * 64;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = pass == 0 ? acc[r][e] : acc[r][e] - mean[r];
                s += pass == 0 ? d : d * d;
            }
            s += __shfl_xor(s, 32, 64);
            if (half == 0) rp[wave * 64 + r * 32 + l31] = s;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = r * 32 + l31;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += rp[w * 64 + row];
            if (pass == 0) mean[r] = t / GN_D;
            else rstd[r] = rsqrtf(t / GN_D + 1e-5f);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + n), b = *reinterpret_cast<const f32x4*>(beta + n);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][4 * q + e] = (acc[r][4 * q + e] - mean[r]) * rstd[r] * g[e] + b[e];
    }
}

// f32 rows [nq][256] (global) -> bf16 tile [64][GN_LD]; rows >= nq are zero
__device__ __forceinline__ void gn_load_rows(const float* __restrict__ g, int nq, bf16_t* T, int tid) {
    // UNCONDITIONAL loads from a clamped row, zeroed afterwards (round 6, read off the ISA: `if (row < nq) load` compiles to a branch with
    // the s_waitcnt for its own load inside - the four iterations of a tile ran as four back-to-back memory round trips, eight per
    // launch of a kernel that is a 42 us latency chain)
    constexpr int NI = 64 * 32 / 512;
    f32x4 a[NI], b[NI];
    const int last = nq > 0 ? nq - 1 : 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = tid + i * 512, row = c >> 5, col = (c & 31) * 8;
        const int rc = row < nq ? row : last;
        a[i] = *reinterpret_cast<const f32x4*>(g + (long long)rc * GN_D + col);
        b[i] = *reinterpret_cast<const f32x4*>(g + (long long)rc * GN_D + col + 4);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = tid + i * 512, row = c >> 5, col = (c & 31) * 8;
        u32x4 pk = u32x4{gn_pack2(a[i][0], a[i][1]), gn_pack2(a[i][2], a[i][3]), gn_pack2(b[i][0], b[i][1]), gn_pack2(b[i][2], b[i][3])};
        if (row >= nq) pk = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(T + row * GN_LD + col) = pk;
    }
}

// 8 waves: wave w owns channel tile w of every 256-wide result (= head w in the attention), tiles 2w, 2w+1 of the hidden layer.
__global__ __launch_bounds__(512, 2) void gnn_layer_kernel(const GnnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gn_smem[];
    bf16_t* X16 = reinterpret_cast<bf16_t*>(gn_smem);       // x as bf16 (whole layer)
    bf16_t* RB = X16 + GN_TILE;                             // s (cross) -> q -> msg -> LN1(merge(msg))
    bf16_t* Kt = RB + GN_TILE;                              // K [key][256]
    bf16_t* Vt = Kt + GN_TILE;                              // V^T [256][72]
    bf16_t* Ht = Kt;                                        // hidden [64][520] (after attention)
    float* red = reinterpret_cast<float*>(Kt + GN_RC);      // [2][8][64]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= p.n_work) {
        // prefetch workgroup (see GnnArgs): workgroup ids are dealt round-robin over the 8 XCDs
        const int xcd = blockIdx.x & 7, slot = ((int)blockIdx.x - p.n_work) >> 3;     // every (XCD, slot) pair exists once
        if (xcd >= p.pf_xcds || slot >= GN_PF_PER_XCD) return;
        unsigned acc = 0u;
        for (int c = slot; c < GN_PF_CHUNKS; c += GN_PF_PER_XCD) {
            // chunk -> (array, byte offset): wq / wk / wv / wm 16 chunks each, w0 64, w2 32
            const int a = c < 64 ? c >> 4 : (c < 128 ? 4 : 5);
            const int o = c < 64 ? c & 15 : (c < 128 ? c - 64 : c - 128);
            // (an ordinary load: a nontemporal one does not allocate in L2 / the Infinity Cache - measured: no effect at all)
            const u32x4 v = *(reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(p.pf[a]) + (long long)o * GN_PF_CHUNK) + tid);
            acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
        }
        asm volatile("" ::"v"(acc));                                                   // the loads must be issued; their values are not needed
        return;
    }
    const int nq = p.nq, nqb = (nq + 63) / 64;
    const int b = blockIdx.x / nqb, qb = blockIdx.x % nqb, row0 = qb * 64;          // set, block of 64 query rows
    const float* xg = p.x + ((long long)(p.x_off + b) * nq + row0) * GN_D;
    const float* sg = p.src + (long long)(p.src_off + b) * nq * GN_D;
    const bool self_layer = (p.x == p.src) && (p.x_off == p.src_off);
    const int nrow = (p.qlen ? min(p.qlen[p.x_off + b], nq) : nq) - row0;          // valid query rows of this block (may be <= 0)
    const int nkey = p.klen ? min(p.klen[p.src_off + b], nq) : nq;
    const int nchunk = max(1, (nkey + 63) / 64);                                   // key chunks that hold a valid key
    GnRing ring;
    f32x16 acc[2];

    gn_issue(ring, 0, p.wq, 16, 0, wave, lane);                            // step 0: Q
    gn_load_rows(xg, nq - row0, X16, tid);
    __syncthreads();

    // ---- q of this block's 64 rows: head = wave; kept in registers as the B operand of the score MFMAs
    gn_issue(ring, 1, p.wk, 16, 0, wave, lane);                            // step 1: K of chunk 0
    gn_zero(acc);
    gn_gemm<false>(ring, 0, X16, GN_LD, acc, lane);
    bf16x8 qf[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        unsigned int d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = gn_pack2(acc[r][2 * i], acc[r][2 * i + 1]);
#pragma unroll
        for (int g = 0; g < 2; ++g) {                                       // k-step g: channels 16g .. 16g+15 of the head
            auto r0 = __builtin_amdgcn_permlane32_swap(d[4 * g + 0], d[4 * g + 2], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(d[4 * g + 1], d[4 * g + 3], false, false);
            const u32x4 pk = {r0[0], r1[0], r0[1], r1[1]};
            qf[r][g] = __builtin_bit_cast(bf16x8, pk);
        }
    }
    // running softmax state per query group r: max, sum, un-normalised output
    float run_mx[2] = {-INFINITY, -INFINITY}, run_ps[2] = {0.f, 0.f};
    f32x16 oacc[2];
    gn_zero(oacc);
    const int c0 = wave * 32;
    for (int c = 0; c < nchunk; ++c) {
        // ---- K (row-major) and V^T of key chunk c (source rows c*64 .. +63)
        const bool own = self_layer && c == qb;                            // the chunk IS this block's rows: already in X16
        if (c > 0) __syncthreads();                                         // every wave is done reading the previous chunk out of RB
        if (!own) gn_load_rows(sg + (long long)c * 64 * GN_D, nq - c * 64, RB, tid);
        __syncthreads();                                                    // RB staged
        const bf16_t* S16 = own ? X16 : RB;
        gn_issue(ring, 0, p.wv, 16, 0, wave, lane);                        // V of this chunk
        gn_zero(acc);
        gn_gemm<false>(ring, 1, S16, GN_LD, acc, lane);
        gn_store_tile<false>(acc, Kt, GN_LD, wave, lane);
        gn_issue(ring, 1, c + 1 < nchunk ? p.wk : p.wm, 16, 0, wave, lane);   // K of the next chunk, or the merge weights
        gn_zero(acc);
        gn_gemm<true>(ring, 0, S16, GN_LD, acc, lane);
        // lane holds channel wave*32 + l31, tokens r*32 + 8q + 4*half + e
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                us4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(acc[r][4 * q + e]);
                *reinterpret_cast<us4*>(Vt + (wave * 32 + l31) * GN_VLD + r * 32 + 8 * q + 4 * half) = o;
            }
        // ---- attention over this chunk: head = wave = the channel tile this wave produced in K / V^T: no barrier needed
        const int kbase = c * 64;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            f32x16 s[2];
            float mx = run_mx[r];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kt + (kt * 32 + l31) * GN_LD + c0 + ks * 16 + half * 8);
                    s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[r][ks], s[kt], 0, 0, 0);
                }
                // s[kt][e] = score(query r*32 + l31, key kbase + kt*32 + (e&3) + 8*(e>>2) + 4*half)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kbase + kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (key >= nkey) s[kt][e] = -INFINITY;
                    mx = fmaxf(mx, s[kt][e]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // first chunk: run_mx = -inf, nothing accumulated yet (scale irrelevant); a fully masked set keeps mx = -inf: msg = 0 below
            // (hardware v_exp_f32 forms: the probabilities are rounded to bf16 MFMA operands right below; 64 library expf per lane and
            //  chunk were ~2.5 k cycles of VALU on the layer's critical path)
            const float scale = (c == 0 || mx == -INFINITY) ? 1.f : __expf(run_mx[r] - mx);
            const float sub = mx == -INFINITY ? 0.f : mx;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { s[kt][e] = __expf(s[kt][e] - sub); ps += s[kt][e]; }
            ps += __shfl_xor(ps, 32, 64);
            run_ps[r] = run_ps[r] * scale + ps;
            run_mx[r] = mx;
            if (c > 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[r][e] *= scale;
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                unsigned int d[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) d[i] = gn_pack2(s[kt][2 * i], s[kt][2 * i + 1]);
#pragma unroll
                for (int g = 0; g < 2; ++g) {                  // k-step g: keys kt*32 + 16g .. +15
                    auto r0 = __builtin_amdgcn_permlane32_swap(d[4 * g + 0], d[4 * g + 2], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(d[4 * g + 1], d[4 * g + 3], false, false);
                    const u32x4 pk = {r0[0], r1[0], r0[1], r1[1]};
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vt + (c0 + l31) * GN_VLD + kt * 32 + g * 16 + half * 8);
                    oacc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[r], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                                        // every wave is done with RB (the last s chunk)
    // oacc[r][e] = un-normalised msg(query r*32 + l31, channel c0 + (e&3) + 8*(e>>2) + 4*half) -> RB (operand of the merge GEMM)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const bool live = (r * 32 + l31) < nrow && nkey > 0;
        const float inv = live ? 1.f / run_ps[r] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            us4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(live ? oacc[r][4 * q + e] * inv : 0.f);
            *reinterpret_cast<us4*>(RB + (r * 32 + l31) * GN_LD + c0 + 8 * q + 4 * half) = o;
        }
    }
    __syncthreads();

    // ---- msg = LN1(merge(msg)) -> RB (bf16 operand of mlp.0)
    gn_issue(ring, 0, p.w0, 32, 0, 2 * wave, lane);                        // step 4: mlp.0 tile 2w, x half
    gn_zero(acc);
    gn_gemm<false>(ring, 1, RB, GN_LD, acc, lane);
    gn_layernorm(acc, p.g1, p.b1, red, wave, lane);                        // contains barriers: every wave is done reading RB
    gn_store_tile<false>(acc, RB, GN_LD, wave, lane);
    __syncthreads();

    // ---- hidden = relu([x | msg] W0^T): 512 channels, tiles 2w and 2w+1, K = 256 (x) + 256 (msg); K / V^T are dead
    f32x16 acc1[2];
    gn_issue(ring, 1, p.w0, 32, 16, 2 * wave, lane);                       // step 5: tile 2w, msg half
    gn_zero(acc);
    gn_gemm<false>(ring, 0, X16, GN_LD, acc, lane);
    gn_issue(ring, 0, p.w0, 32, 0, 2 * wave + 1, lane);                    // step 6: tile 2w+1, x half
    gn_gemm<false>(ring, 1, RB, GN_LD, acc, lane);
    gn_issue(ring, 1, p.w0, 32, 16, 2 * wave + 1, lane);                   // step 7: tile 2w+1, msg half
    gn_zero(acc1);
    gn_gemm<false>(ring, 0, X16, GN_LD, acc1, lane);
    gn_issue(ring, 0, p.w2, 32, 0, wave, lane);                            // step 8: mlp.2, first 256 hidden channels
    gn_gemm<false>(ring, 1, RB, GN_LD, acc1, lane);
    gn_store_tile<true>(acc, Ht, GN_HLD, 2 * wave, lane);
    gn_store_tile<true>(acc1, Ht, GN_HLD, 2 * wave + 1, lane);
    __syncthreads();

    // ---- out = x + LN2(hidden W2^T)
    gn_issue(ring, 1, p.w2, 32, 16, wave, lane);                           // step 9: mlp.2, last 256 hidden channels
    // the f32 residual rows of the epilogue, requested now: their L2 / HBM round trip runs under the last GEMM + LayerNorm
    f32x4 xres[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = min(r * 32 + l31, max(nq - row0 - 1, 0));           // (rows beyond the set are clamped: loaded, never stored)
#pragma unroll
        for (int q = 0; q < 4; ++q) xres[r][q] = *reinterpret_cast<const f32x4*>(xg + (long long)row * GN_D + wave * 32 + 8 * q + 4 * half);
    }
    gn_zero(acc);
    gn_gemm<false>(ring, 0, Ht, GN_HLD, acc, lane);
    gn_gemm<false>(ring, 1, Ht + 256, GN_HLD, acc, lane);
    gn_layernorm(acc, p.g2, p.b2, red, wave, lane);
    float* og = p.out + ((long long)(p.out_off + b) * nq + row0) * GN_D;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = r * 32 + l31;
        if (row0 + row >= nq) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = wave * 32 + 8 * q + 4 * half;
            const f32x4 xv = xres[r][q];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = xv[e] + acc[r][4 * q + e];
            *reinterpret_cast<f32x4*>(og + (long long)row * GN_D + n) = o;
        }
    }
}

}  // namespace nps

extern "C" int nopesac_gnn_layer_bf16_pf(const float* x, int x_off, const float* src, int src_off, float* out, int out_off, int n_sets,
                                         int nq, const int32_t* qlen, const int32_t* klen, const void* wq, const void* wk, const void* wv,
                                         const void* wmerge, const void* w0, const void* w2, const float* ln1_g, const float* ln1_b,
                                         const float* ln2_g, const float* ln2_b, const void* const* next_weights6, int next_sets,
                                         void* stream);

extern "C" int nopesac_gnn_layer_bf16(const float* x, int x_off, const float* src, int src_off, float* out, int out_off, int n_sets,
                                      int nq, const int32_t* qlen, const int32_t* klen, const void* wq, const void* wk, const void* wv,
                                      const void* wmerge, const void* w0, const void* w2, const float* ln1_g, const float* ln1_b,
                                      const float* ln2_g, const float* ln2_b, void* stream) {
    return nopesac_gnn_layer_bf16_pf(x, x_off, src, src_off, out, out_off, n_sets, nq, qlen, klen, wq, wk, wv, wmerge, w0, w2, ln1_g, ln1_b,
                                     ln2_g, ln2_b, nullptr, 0, stream);
}

extern "C" int nopesac_gnn_layer_bf16_pf(const float* x, int x_off, const float* src, int src_off, float* out, int out_off, int n_sets,
                                         int nq, const int32_t* qlen, const int32_t* klen, const void* wq, const void* wk, const void* wv,
                                         const void* wmerge, const void* w0, const void* w2, const float* ln1_g, const float* ln1_b,
                                         const float* ln2_g, const float* ln2_b, const void* const* next_weights6, int next_sets,
                                         void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && src && out && wq && wk && wv && wmerge && w0 && w2 && ln1_g && ln1_b && ln2_g && ln2_b, "gnn_layer: null pointer");
    NPS_CHECK_ARG(n_sets > 0 && nq > 0 && nq <= 128 && x_off >= 0 && src_off >= 0 && out_off >= 0, "gnn_layer: bad sizes (nq <= 128)");
    // other workgroups still read x / src while this one writes out: same buffer only with disjoint set ranges
    auto disjoint = [&](const float* in, int in_off) { return in != out || in_off + n_sets <= out_off || out_off + n_sets <= in_off; };
    NPS_CHECK_ARG(disjoint(x, x_off) && disjoint(src, src_off), "gnn_layer: out overlaps x / src");
    const void* ptrs[] = {x, src, out, wq, wk, wv, wmerge, w0, w2, ln1_g, ln1_b, ln2_g, ln2_b};
    for (const void* q : ptrs) NPS_CHECK_ARG(((uintptr_t)q & 15) == 0, "gnn_layer: pointers must be 16-byte aligned");
    GnnArgs a;
    a.x = x; a.src = src; a.out = out; a.x_off = x_off; a.src_off = src_off; a.out_off = out_off; a.nq = nq;
    a.qlen = qlen; a.klen = klen;
    a.wq = (const bf16_t*)wq; a.wk = (const bf16_t*)wk; a.wv = (const bf16_t*)wv; a.wm = (const bf16_t*)wmerge;
    a.w0 = (const bf16_t*)w0; a.w2 = (const bf16_t*)w2; a.g1 = ln1_g; a.b1 = ln1_b; a.g2 = ln2_g; a.b2 = ln2_b;
    a.n_work = n_sets * ((nq + 63) / 64);
    a.pf_xcds = 0;
    for (int i = 0; i < 6; ++i) a.pf[i] = nullptr;
    int n_pf = 0;
    if (next_weights6) {
        NPS_CHECK_ARG(next_sets > 0, "gnn_layer: next_sets must be > 0 with next_weights6");
        for (int i = 0; i < 6; ++i) {
            NPS_CHECK_ARG(next_weights6[i] && ((uintptr_t)next_weights6[i] & 15) == 0, "gnn_layer: next_weights6[%d] null / unaligned", i);
            a.pf[i] = (const bf16_t*)next_weights6[i];
        }
        const int next_work = next_sets * ((nq + 63) / 64);
        a.pf_xcds = next_work < 8 ? next_work : 8;
        // GN_PF_PER_XCD octets of workgroups (consecutive ids = one per XCD); with many layer workgroups (batches) the weights are
        // shared by all of them and stay in every L2 anyway: no prefetch workgroups
        n_pf = a.n_work <= 16 ? 8 * GN_PF_PER_XCD : 0;
        if (!n_pf) a.pf_xcds = 0;
    }
    NPS_ENSURE_LDS((int)GN_LDS_BYTES, gnn_layer_kernel);
    hipLaunchKernelGGL(gnn_layer_kernel, dim3(a.n_work + n_pf), dim3(512), GN_LDS_BYTES, (hipStream_t)stream, a);
    NPS_LAUNCH_RET();
}
