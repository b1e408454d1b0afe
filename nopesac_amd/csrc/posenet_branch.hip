// Tail of one branch of the pixel pose net (camera_net/camera_modules.py: `convs_trans` / `convs_rots`, six
// Conv3x3 + BatchNorm + LeakyReLU(0.01) layers with strides 1,2,1,2,1,2 on the 15 x 20 affinity volume; call site
// camera_head.py:642-735): layers 1..5 in ONE launch, one 8-wave workgroup per (image pair, branch).
//
// Un-fused, the five layers behind the first one are five launches of ~22 us each per branch (3x3, 128 -> 128 on 8x10 / 4x5 / 2x3
// maps: 80 down to 6 output pixels per pair - nothing to fill a chip with, every launch a latency chain of its own); with 32 pairs
// they hold 80 workgroups for 0.2 ms per step, with one pair per call they are 0.2 ms of the 4.9 ms latency.
// Here the branch's activations never leave the CU:
//   * layer 0's output (15 x 20 x 128 bf16, 77 KB) is parked in LDS with a zero halo (17 x 22 positions, 272-byte rows);
//   * every layer is an implicit GEMM straight out of that tile: output pixel = GEMM row, K index = (kh*3 + kw)*128 + c (8 k-steps of
//     16 per tap), A fragment = one ds_read_b128 at (oy*s + kh, ox*s + kw) of the halo tile - no im2col buffer;
//   * weights are fragment-major (ops.mfma_fragment_major of [128][1152]) and stream from L2 through an 8-slot rolling register ring:
//     wave w owns output-channel tile w & 3; waves 0-3 take row tiles 0 (and 2), waves 4-7 row tile 1 (layers with <= 32 output
//     pixels run on waves 0-3 only);
//   * BN scale / shift + LeakyReLU in registers, the bf16 result goes into the interior of the OTHER LDS region (zero halo), which
//     is the next layer's input; the last layer's 2 x 3 x 128 outputs leave as f32 (the FC stack's input).
// Arithmetic: bf16 operands, f32 accumulation, activations rounded to bf16 between the layers - the rounding points of the per-layer
// path (conv2d with bf16 output); only the f32 summation order differs.
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

constexpr int PB_C = 128, PB_ROWB = PB_C * 2 + 16;            // bytes per halo-tile position (272: consecutive pixels shift 4 banks)
constexpr int PB_H0 = 15, PB_W0 = 20;
constexpr int PB_R0_POS = (PB_H0 + 2) * (PB_W0 + 2);           // 374 positions: the 15 x 20 input (and, later, the 8 x 10 / 4 x 5 maps)
constexpr int PB_R1_POS = (8 + 2) * (10 + 2);                  // 120 positions: 8 x 10 (and 4 x 5) maps
constexpr int PB_R0_BYTES = PB_R0_POS * PB_ROWB, PB_R1_BYTES = PB_R1_POS * PB_ROWB;
constexpr size_t PB_LDS_BYTES = (size_t)PB_R0_BYTES + PB_R1_BYTES;   // 134.4 KB
constexpr int PB_KS = 9 * PB_C / 16;                           // 72 k-steps per layer
constexpr int PB_RING = 8;

struct PoseBranchArgs {
    const bf16_t* x[2];             // layer-0 output of the two branches, [B][15][20][128] bf16
    const bf16_t* w[2][5];          // fragment-major [128][1152] of layers 1..5
    const float* scale[2][5];       // folded BatchNorm
    const float* bias[2][5];
    float* y[2];                    // [B][2][3][128] f32
    int B;
};

// zero `n16` 16-byte words at p (LDS), cooperatively
__device__ __forceinline__ void pb_zero(unsigned char* p, int n16, int tid) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < n16; i += 512) *reinterpret_cast<uint4*>(p + (size_t)i * 16) = z;
}

// One 3x3 layer: src halo tile (IH x IW interior, pitch IW + 2) -> OH x OW outputs.  NRT row tiles of 32 output pixels.
// LAST = false: bf16 into the interior of dst (halo tile with pitch OW + 2); LAST = true: f32 to global `yout` [OH*OW][128].
template <int IH, int IW, int STRIDE, bool LAST>
__device__ __forceinline__ void pb_layer(const unsigned char* src, unsigned char* dst, const bf16_t* __restrict__ wf, const float* __restrict__ scale,
                                         const float* __restrict__ bias, float* __restrict__ yout, int wave, int lane) {
    constexpr int OH = (IH + 2 - 3) / STRIDE + 1, OW = (IW + 2 - 3) / STRIDE + 1, M = OH * OW;
    constexpr int NRT = (M + 31) / 32;                           // 3, 3, 1, 1, 1
    constexpr int IWP = IW + 2, OWP = OW + 2;
    static_assert(NRT <= 3, "row tiles");
    const int l31 = lane & 31, half = lane >> 5;
    const int nt = wave & 3, grp = wave >> 2;
    // row tiles of this wave: group 0 -> tiles 0 and 2, group 1 -> tile 1
    constexpr int MAXT = NRT > 2 ? 2 : 1;
    int a_base[MAXT];
    bool t_on[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int rt = grp + 2 * t;
        t_on[t] = rt < NRT;
        int p = rt * 32 + l31;
        if (p >= M) p = M - 1;                                   // padded rows compute a copy of the last pixel, never stored
        const int oy = p / OW, ox = p - oy * OW;
        a_base[t] = ((oy * STRIDE) * IWP + ox * STRIDE) * PB_ROWB + half * 16;
    }
    if (!t_on[0]) return;                                        // (wave-uniform: waves 4-7 of the one-tile layers)
    f32x16 acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const bf16_t* wp = wf + ((long long)nt * PB_KS * 64 + lane) * 8;
    bf16x8 ring[PB_RING];
#pragma unroll
    for (int s = 0; s < PB_RING; ++s) ring[s] = *reinterpret_cast<const bf16x8*>(wp + s * 512);
#pragma unroll
    for (int ks = 0; ks < PB_KS; ++ks) {
        const int tap = ks >> 3, kh = tap / 3, kw = tap - kh * 3;
        const int koff = (kh * IWP + kw) * PB_ROWB + (ks & 7) * 32;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t == 0 || t_on[t]) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(src + a_base[t] + koff);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[ks % PB_RING], af, acc[t], 0, 0, 0);
            }
        }
        if (ks + PB_RING < PB_KS) ring[ks % PB_RING] = *reinterpret_cast<const bf16x8*>(wp + (ks + PB_RING) * 512);
    }
    // ---- BN + LeakyReLU; lane holds pixel l31 of its row tile, channels nt*32 + 8q + 4*half + {0..3}
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = nt * 32 + 8 * q + 4 * half;
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(scale + n), b4 = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if (t > 0 && !t_on[t]) continue;
            const int p = (grp + 2 * t) * 32 + l31;
            if (p >= M) continue;
            const int oy = p / OW, ox = p - oy * OW;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = acc[t][4 * q + e] * s4[e];
                x += b4[e];
                v[e] = x > 0.f ? x : 0.01f * x;
            }
            if constexpr (LAST) {
                *reinterpret_cast<f32x4*>(yout + (long long)p * PB_C + n) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                us4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(v[e]);
                *reinterpret_cast<us4*>(dst + ((oy + 1) * OWP + ox + 1) * PB_ROWB + n * 2) = o;
            }
        }
    }
}

__global__ __launch_bounds__(512, 1) void posenet_branch_tail_kernel(const PoseBranchArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pb_smem[];
    unsigned char* R0 = pb_smem;
    unsigned char* R1 = pb_smem + PB_R0_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x >> 1, br = blockIdx.x & 1;
    // ---- layer-0 output of this pair -> interior of R0 (zero halo), R1 zeroed for layer 1's output
    const bf16_t* xg = p.x[br] + (long long)b * PB_H0 * PB_W0 * PB_C;
    constexpr int NCH = PB_H0 * PB_W0 * (PB_C / 8);                // 4800 16-byte chunks
    constexpr int PER = (NCH + 511) / 512;
    us8 rx[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + i * 512;
        rx[i] = us8{0, 0, 0, 0, 0, 0, 0, 0};
        if (c < NCH) rx[i] = *reinterpret_cast<const us8*>(xg + (long long)c * 8);
    }
    pb_zero(R0, PB_R0_BYTES / 16, tid);
    pb_zero(R1, PB_R1_BYTES / 16, tid);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = tid + i * 512;
        if (c < NCH) {
            const int pix = c >> 4, col = c & 15, iy = pix / PB_W0, ix = pix - iy * PB_W0;
            *reinterpret_cast<us8*>(R0 + ((iy + 1) * (PB_W0 + 2) + ix + 1) * PB_ROWB + col * 16) = rx[i];
        }
    }
    __syncthreads();
    float* yg = p.y[br] + (long long)b * 6 * PB_C;
    // layer 1: 15x20 -> 8x10 (stride 2), R0 -> R1
    pb_layer<15, 20, 2, false>(R0, R1, p.w[br][0], p.scale[br][0], p.bias[br][0], nullptr, wave, lane);
    __syncthreads();
    pb_zero(R0, (8 + 2) * (10 + 2) * PB_ROWB / 16, tid);           // layer 2's output tile (10 x 12 positions) lives at the start of R0
    __syncthreads();
    // layer 2: 8x10 -> 8x10, R1 -> R0
    pb_layer<8, 10, 1, false>(R1, R0, p.w[br][1], p.scale[br][1], p.bias[br][1], nullptr, wave, lane);
    __syncthreads();
    pb_zero(R1, (4 + 2) * (5 + 2) * PB_ROWB / 16, tid);
    __syncthreads();
    // layer 3: 8x10 -> 4x5 (stride 2), R0 -> R1
    pb_layer<8, 10, 2, false>(R0, R1, p.w[br][2], p.scale[br][2], p.bias[br][2], nullptr, wave, lane);
    __syncthreads();
    pb_zero(R0, (4 + 2) * (5 + 2) * PB_ROWB / 16, tid);
    __syncthreads();
    // layer 4: 4x5 -> 4x5, R1 -> R0
    pb_layer<4, 5, 1, false>(R1, R0, p.w[br][3], p.scale[br][3], p.bias[br][3], nullptr, wave, lane);
    __syncthreads();
    // layer 5: 4x5 -> 2x3 (stride 2), R0 -> global f32
    pb_layer<4, 5, 2, true>(R0, nullptr, p.w[br][4], p.scale[br][4], p.bias[br][4], yg, wave, lane);
}

}  // namespace nps

extern "C" int nopesac_posenet_branch_tail_bf16(const void* x_trans, const void* x_rots, const void* const* w10, const float* const* scale10,
                                                const float* const* bias10, float* y_trans, float* y_rots, int B, int H, int W, int C,
                                                void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x_trans && x_rots && w10 && scale10 && bias10 && y_trans && y_rots && B > 0, "posenet_branch_tail: null pointer / empty batch");
    NPS_CHECK_ARG(H == PB_H0 && W == PB_W0 && C == PB_C, "posenet_branch_tail: built for the 15 x 20 x 128 map of 480 x 640 inputs (got %d x %d x %d)", H, W, C);
    PoseBranchArgs a;
    a.x[0] = (const bf16_t*)x_trans; a.x[1] = (const bf16_t*)x_rots; a.y[0] = y_trans; a.y[1] = y_rots; a.B = B;
    for (int br = 0; br < 2; ++br)
        for (int i = 0; i < 5; ++i) {
            a.w[br][i] = (const bf16_t*)w10[br * 5 + i]; a.scale[br][i] = scale10[br * 5 + i]; a.bias[br][i] = bias10[br * 5 + i];
            NPS_CHECK_ARG(a.w[br][i] && a.scale[br][i] && a.bias[br][i], "posenet_branch_tail: layer %d of branch %d is missing", i + 1, br);
            NPS_CHECK_ARG(((uintptr_t)a.w[br][i] & 15) == 0 && ((uintptr_t)a.scale[br][i] & 15) == 0 && ((uintptr_t)a.bias[br][i] & 15) == 0,
                          "posenet_branch_tail: weights / scale / bias must be 16-byte aligned");
        }
    NPS_CHECK_ARG(((uintptr_t)x_trans & 15) == 0 && ((uintptr_t)x_rots & 15) == 0 && ((uintptr_t)y_trans & 15) == 0 && ((uintptr_t)y_rots & 15) == 0,
                  "posenet_branch_tail: activations must be 16-byte aligned");
    NPS_ENSURE_LDS((int)PB_LDS_BYTES, posenet_branch_tail_kernel);
    hipLaunchKernelGGL(posenet_branch_tail_kernel, dim3(2 * B), dim3(512), PB_LDS_BYTES, (hipStream_t)stream, a);
    NPS_LAUNCH_RET();
}
