// A stack of Linear(+bias)(+activation) layers over the rows of a matrix in ONE launch (bf16 mode of the heads: f32 activations
// between layers, bf16 MFMA operands - the same rounding points as one nopesac_conv2d_nhwc launch per layer).
//
// The camera head's neural one-plane RANSAC (camera_net/camera_head.py:957-990: geo_encoder -> geo_proj_s1 -> decoder_rot ->
// geo_proj_s2 -> decoder_tran -> decoder_rot2 / decoder_tran2 -> rots / trans) is 40 dependent GEMMs over B*nq rows (1600 at
// B = 32, nq = 50) with 256..1280 channels: as one launch per layer each is a 10-15 us latency-bound kernel spread over the whole
// chip.  Here a workgroup (8 waves) owns 32 rows through the WHOLE stack: the activations ping-pong between two LDS regions as
// bf16, every layer is "LDS tile x fragment-major weights streamed from L2 straight into registers" (the idiom of enc_tail.hip /
// pwchain.hip, weights as the MFMA A operand: D[channel][row]), the next block of weights - across layer boundaries too - is
// in flight while the current one is multiplied, and only the layer outputs a consumer needs are written to HBM (f32).
//
//   rows 32 per workgroup: in 82 KB (width <= 1280) + out 66 KB (width <= 1024) of LDS; 1600 rows = 50 workgroups
//   a wave has 4 fragment slots: 4 channel tiles (t = wave + 8 i) of a wide layer, or 2 / 1 tiles x 2 / 4 K ranges of a narrow one;
//   per block it loads 16 weight fragments (1 KB each, coalesced) and issues 16 v_mfma_f32_32x32x16_bf16
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

constexpr int MC_BM = 32;
constexpr int MC_W0 = NOPESAC_MLP_MAX_IN, MC_W1 = NOPESAC_MLP_MAX_WIDTH;       // 1280 / 1024
constexpr int MC_LD0 = MC_W0 + 8, MC_LD1 = MC_W1 + 8;                         // +16 B: conflict-free 16-byte row reads
constexpr size_t MC_LDS_BYTES = 2 * (size_t)MC_BM * (MC_LD0 + MC_LD1);

#define MC_LDS_SYNC()                                          \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    } while (0)

constexpr int MC_TPW = 4, MC_KB = 4;                            // fragment slots per wave and k-step, k-steps (16 channels) per slot and block
// Slot use by layer width: a wave owns `tpw` channel tiles (1 / 2 / 4 for <= 8 / <= 16 / more tiles of 32 channels); with fewer than 4
// the spare slots take OTHER K RANGES of the same tiles (ksplit = 4 / tpw: slot i -> tile wave + 8 (i % tpw), K part i / tpw), their
// accumulators are summed in the epilogue - a 256-wide layer walks its K in a quarter of the block steps (each block step is one L2
// round trip of latency for a workgroup that streams its weights alone).
__host__ __device__ constexpr int mc_tpw(int N) { return N <= 256 ? 1 : (N <= 512 ? 2 : 4); }
// K padded to an EVEN number of blocks of 64 * ksplit channels
__host__ __device__ constexpr int mc_kpad(int K, int N) { return (K + 128 * (4 / mc_tpw(N)) - 1) / (128 * (4 / mc_tpw(N))) * (128 * (4 / mc_tpw(N))); }

// Two slots of 16 weight fragments (128 VGPRs): entry kk * 4 + i of a slot = k-step kk of fragment slot i.
// ONE code shape for every layer width (tile / K-part of a slot are wave-uniform run-time values): the load and MFMA counts stay
// static and the compiler counts vmcnt exactly; width classes as separate instantiations spilled 150 VGPRs and put a branch around
// every MFMA.
struct McRing {
    bf16x8 f[2][16];
};

struct McShape {                // per layer, wave-uniform
    const bf16_t* w;
    int ksteps, ntiles, tpw, ksplit;
};
__device__ __forceinline__ McShape mc_shape(const nopesac_mlp_layer& L) {
    const int tpw = mc_tpw(L.N);
    return McShape{(const bf16_t*)L.w, mc_kpad(L.K, L.N) / 16, (L.N + 31) / 32, tpw, 4 / tpw};
}

template <int SLOT>
__device__ __forceinline__ void mc_issue(McRing& ring, const McShape& S, int blk, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < MC_TPW; ++i) {
        int t = wave + 8 * (i % S.tpw);                                          // (tpw is 1, 2 or 4: the compiler sees and / shift)
        t = t < S.ntiles ? t : S.ntiles - 1;                                     // idle tiles (N not a multiple of 256 * tpw) re-load a valid one
        const int k0 = (blk * S.ksplit + i / S.tpw) * MC_KB;                     // first k-step of this slot in this block
        const bf16_t* base = S.w + (size_t)(t * S.ksteps + k0) * 512;            // wave-uniform
#pragma unroll
        for (int kk = 0; kk < MC_KB; ++kk) ring.f[SLOT][kk * MC_TPW + i] = *reinterpret_cast<const bf16x8*>(base + kk * 512 + lane * 8);
    }
}

template <int SLOT>
__device__ __forceinline__ void mc_gemm(const McRing& ring, const McShape& S, const bf16_t* src, int ld, int blk, f32x16 (&acc)[MC_TPW], int lane) {
    const int l31 = lane & 31, half = lane >> 5;
    const bf16_t* row = src + l31 * ld + half * 8;
    {
#pragma unroll
        for (int kk = 0; kk < MC_KB; ++kk) {
#pragma unroll
            for (int i = 0; i < MC_TPW; ++i) {
                const int ks = (blk * S.ksplit + i / S.tpw) * MC_KB + kk;
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(row + ks * 16);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.f[SLOT][kk * MC_TPW + i], b, acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);               // at most four activation fragments live (all sixteen hoisted = 64 VGPRs: spills)
        }
    }
}

// One layer: dst[row][n] = bf16(act(src[row][:] . W[n][:] + bias[n])), optional f32 copy to HBM.  On entry the weights of K block 0
// are in flight in ring slot 0 (issued by the previous layer / the kernel prologue); on exit block 0 of `next` is in flight.
__device__ __forceinline__ void mc_layer(const nopesac_mlp_layer& L, const McShape& S, const McShape& next, const bf16_t* src, int sld, bf16_t* dst,
                                         int dld, long long row0, int rows, int wave, int lane, McRing& ring) {
    const int nblk = S.ksteps / (MC_KB * S.ksplit);                               // even
    const int ntiles = S.ntiles;
    f32x16 acc[MC_TPW];
#pragma unroll
    for (int i = 0; i < MC_TPW; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int blk = 0; blk < nblk; blk += 2) {
        // every issue is unconditional and pinned in front of the multiply of the OTHER slot: the compiler counts vmcnt(16) for the
        // slot being consumed and the 16 loads in flight have a whole block's MFMAs (16 x 64 cycles x 2 waves per SIMD) of cover
        mc_issue<1>(ring, S, blk + 1, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        mc_gemm<0>(ring, S, src, sld, blk, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        const bool last = blk + 2 >= nblk;                    // next block: this layer's, or block 0 of the next layer (`next` = this
        mc_issue<0>(ring, last ? next : S, last ? 0 : blk + 2, wave, lane);                    // layer again after the last one: a harmless re-load)
        __builtin_amdgcn_sched_barrier(0);
        mc_gemm<1>(ring, S, src, sld, blk + 1, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the K parts of a tile meet: slot i + tpw, i + 2 tpw ... hold the same tile as slot i
    if (S.ksplit == 4) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
    } else if (S.ksplit == 2) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[0][e] += acc[2][e]; acc[1][e] += acc[3][e]; }
    }
    // ---- epilogue: lane holds row (lane & 31), channels t*32 + 8q + 4*(lane >> 5) .. +3 for q = 0..3
    const int l31 = lane & 31, half = lane >> 5;
    const long long row = row0 + l31;
    const bool row_ok = row < rows;
    const bool vec_out = (L.out_ld & 3) == 0 && (L.N & 3) == 0 && ((uintptr_t)L.out & 15) == 0;
#pragma unroll
    for (int i = 0; i < MC_TPW; ++i) {
        const int t = wave + 8 * i;
        if (i >= S.tpw || t >= ntiles) continue;
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (L.bias) b = *reinterpret_cast<const f32x4*>(L.bias + t * 32 + 8 * q + 4 * half);      // padded to a multiple of 32 by the packer
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * q + e] = acc[i][4 * q + e] + b[e];
        }
        if (L.act == NPS_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        } else if (L.act == NPS_ACT_LEAKY) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
        } else if (L.act == NPS_ACT_SIGMOID) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = 1.f / (1.f + expf(-v[e]));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = t * 32 + 8 * q + 4 * half;
            const uint2 o = make_uint2(f32x2_to_bf16x2(v[4 * q], v[4 * q + 1]), f32x2_to_bf16x2(v[4 * q + 2], v[4 * q + 3]));
            *reinterpret_cast<uint2*>(dst + l31 * dld + n) = o;
        }
        if (L.out && row_ok) {
            float* op = L.out + row * L.out_ld + t * 32 + 4 * half;
            if (vec_out) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (t * 32 + 8 * q + 4 * half < L.N) *reinterpret_cast<f32x4*>(op + 8 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            } else {
                for (int q = 0; q < 4; ++q)
                    for (int e = 0; e < 4; ++e)
                        if (t * 32 + 8 * q + 4 * half + e < L.N) op[8 * q + e] = v[4 * q + e];
            }
        }
    }
}

__global__ __launch_bounds__(512) void mlp_chain_kernel(const nopesac_mlp_chain p_byval) {
    // the layer table is indexed at run time: read it in place from the kernel-argument segment (scalar loads) - indexing the
    // by-value copy made the compiler move the whole 544-byte struct to scratch
    typedef const nopesac_mlp_chain __attribute__((address_space(4)))* KArg;
    const KArg pc = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    auto layer_at = [&](int l) {
        nopesac_mlp_layer L;
        L.w = pc->layers[l].w; L.bias = pc->layers[l].bias; L.out = pc->layers[l].out; L.out_ld = pc->layers[l].out_ld;
        L.K = pc->layers[l].K; L.N = pc->layers[l].N; L.act = pc->layers[l].act; L.reserved = pc->layers[l].reserved;
        return L;
    };
    struct {
        const float* x; long long x_ld; const float* xb; long long xb_ld; int x_width, xb_width, xb_rows_per, rows, n_layers;
    } p = {pc->x, pc->x_ld, pc->xb, pc->xb_ld, pc->x_width, pc->xb_width, pc->xb_rows_per, pc->rows, pc->n_layers};
    extern __shared__ __attribute__((aligned(16))) unsigned char mc_smem[];
    bf16_t* R0 = reinterpret_cast<bf16_t*>(mc_smem);             // [32][1288]: chain input, outputs of odd layers
    bf16_t* R1 = R0 + MC_BM * MC_LD0;                            // [32][1032]: outputs of even layers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long row0 = (long long)blockIdx.x * MC_BM;
    McRing ring;
    McShape shape = mc_shape(layer_at(0));
    mc_issue<0>(ring, shape, 0, wave, lane);                                        // weights of layer 0, K block 0
    // ---- LDS: zeros everywhere (padding columns meet zero weights, but must not be NaN patterns), then the input rows
    for (int i = tid; i < (int)(MC_LDS_BYTES / 16); i += 512) reinterpret_cast<uint4*>(mc_smem)[i] = make_uint4(0u, 0u, 0u, 0u);
    MC_LDS_SYNC();
    // the chain input rows (f32 in HBM) as a bf16 tile in region R (row stride ld); columns [K0, kzero) are zeroed
    auto stage_input = [&](bf16_t* R, int ld, int kzero) {
        const int K0 = p.xb_width + p.x_width;
        for (int i = tid; i < MC_BM * (kzero - K0); i += 512) {
            const int r = i / (kzero - K0), c = K0 + i - r * (kzero - K0);
            R[r * ld + c] = 0;
        }
        const bool vec = ((p.xb_width | p.x_width) & 3) == 0 && (p.x_ld & 3) == 0 && (p.xb_ld & 3) == 0 &&
                         (((uintptr_t)p.x | (uintptr_t)p.xb) & 15) == 0;
        if (vec) {
            const int q4 = K0 >> 2;
            for (int i = tid; i < MC_BM * q4; i += 512) {
                const int r = i / q4, c = (i - r * q4) * 4;
                const long long row = row0 + r;
                us4 o = {0, 0, 0, 0};
                if (row < p.rows) {
                    const float* s = c < p.xb_width ? p.xb + (row / p.xb_rows_per) * p.xb_ld + c : p.x + row * p.x_ld + (c - p.xb_width);
                    const f32x4 v = *reinterpret_cast<const f32x4*>(s);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(v[e]);
                }
                *reinterpret_cast<us4*>(R + r * ld + c) = o;
            }
        } else {
            for (int i = tid; i < MC_BM * K0; i += 512) {
                const int r = i / K0, c = i - r * K0;
                const long long row = row0 + r;
                float v = 0.f;
                if (row < p.rows) v = c < p.xb_width ? p.xb[(row / p.xb_rows_per) * p.xb_ld + c] : p.x[row * p.x_ld + (c - p.xb_width)];
                R[r * ld + c] = f32_to_bf16(v);
            }
        }
    };
    stage_input(R0, MC_LD0, p.xb_width + p.x_width);
    MC_LDS_SYNC();
    for (int l = 0; l < p.n_layers; ++l) {
        const nopesac_mlp_layer L = layer_at(l);
        const McShape next = l + 1 < p.n_layers ? mc_shape(layer_at(l + 1)) : shape;
        const bf16_t* src = (l & 1) ? R1 : R0;
        bf16_t* dst = (l & 1) ? R0 : R1;
        const int sld = (l & 1) ? MC_LD1 : MC_LD0, dld = (l & 1) ? MC_LD0 : MC_LD1;
        if (L.reserved & NOPESAC_MLP_RESTART) {               // this layer reads the chain INPUT again (a parallel stack over the same rows):
            stage_input(const_cast<bf16_t*>(src), sld, mc_kpad(L.K, L.N));   // the previous layer's output in this region is dead (its
            MC_LDS_SYNC();                                                   // consumers, if any, got it through L.out)
        }
        mc_layer(L, shape, next, src, sld, dst, dld, row0, p.rows, wave, lane, ring);
        shape = next;
        MC_LDS_SYNC();
    }
}

}  // namespace nps

extern "C" int64_t nopesac_mlp_packed_elems(int N, int K) {
    if (N <= 0 || K <= 0 || N > NOPESAC_MLP_MAX_WIDTH) return 0;
    return (int64_t)((N + 31) / 32 * 32) * nps::mc_kpad(K, N);
}

extern "C" int nopesac_mlp_padded_k(int N, int K) { return (N <= 0 || K <= 0) ? 0 : nps::mc_kpad(K, N); }

extern "C" int nopesac_mlp_chain_bf16(const nopesac_mlp_chain* chain, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(chain && chain->rows > 0 && chain->n_layers >= 1 && chain->n_layers <= NOPESAC_MLP_MAX_LAYERS, "mlp_chain: rows / n_layers");
    NPS_CHECK_ARG(chain->x && chain->x_width > 0 && chain->x_ld >= chain->x_width, "mlp_chain: x");
    NPS_CHECK_ARG(chain->xb_width >= 0 && (chain->xb_width == 0 || (chain->xb && chain->xb_rows_per > 0 && chain->xb_ld >= chain->xb_width)),
                  "mlp_chain: broadcast prefix");
    int width = chain->xb_width + chain->x_width;
    NPS_CHECK_ARG(width <= MC_W0, "mlp_chain: input wider than NOPESAC_MLP_MAX_IN");
    for (int l = 0; l < chain->n_layers; ++l) {
        const nopesac_mlp_layer& L = chain->layers[l];
        if (L.reserved & NOPESAC_MLP_RESTART) width = chain->xb_width + chain->x_width;      // a parallel stack: back to the chain input
        NPS_CHECK_ARG((L.reserved & ~NOPESAC_MLP_RESTART) == 0 && (l > 0 || L.reserved == 0), "mlp_chain: layer flags");
        NPS_CHECK_ARG(L.w && L.K == width && L.N > 0 && L.N <= MC_W1, "mlp_chain: layer K must equal the previous width, N <= NOPESAC_MLP_MAX_WIDTH");
        NPS_CHECK_ARG(mc_kpad(L.K, L.N) <= ((l & 1) ? MC_W1 : MC_W0), "mlp_chain: padded K exceeds the LDS region");
        NPS_CHECK_ARG(((uintptr_t)L.w & 15) == 0 && ((uintptr_t)L.bias & 15) == 0 && ((uintptr_t)L.out & 3) == 0, "mlp_chain: w / bias must be 16-byte aligned");
        NPS_CHECK_ARG(!L.out || L.out_ld >= L.N, "mlp_chain: out_ld");
        NPS_CHECK_ARG(L.act == NPS_ACT_NONE || L.act == NPS_ACT_RELU || L.act == NPS_ACT_LEAKY || L.act == NPS_ACT_SIGMOID, "mlp_chain: act");
        width = L.N;
    }
    NPS_CHECK_ARG(chain->layers[chain->n_layers - 1].out, "mlp_chain: the last layer needs an output");
    NPS_ENSURE_LDS((int)MC_LDS_BYTES, mlp_chain_kernel);
    hipLaunchKernelGGL(mlp_chain_kernel, dim3((chain->rows + MC_BM - 1) / MC_BM), dim3(512), MC_LDS_BYTES, (hipStream_t)stream, *chain);
    NPS_LAUNCH_RET();
}
