// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels + folded BN + activation, bf16 (the conv2 of the three
// res2 bottlenecks at 120x160: d2 BottleneckBlock, STRIDE_IN_1X1 = False).
//
// The general LDS-DMA kernel serves these layers with 128x64 tiles: every K-tile re-fetches a tap-shifted copy of the same
// pixels (9 x 16 KB per tile) and the 72 KB weight matrix per 128 pixels - 216 KB from L2 for 9.4 MFLOP, 430 TFLOP/s.  Here a
// workgroup owns a 16x16 pixel tile: the 18x18 halo (41 KB) is loaded ONCE into LDS and all nine taps are read from it with
// shifted addresses (no im2col copy, like stem.hip); the weights are streamed fragment-major from L2 through a rolling
// 16-slot register ring, each fragment feeding four MFMAs (wave = one 32-channel column tile x four 32-pixel row tiles).
// L2 traffic per workgroup: 41 KB halo + 2 x 72 KB weights for 18.9 MFLOP.
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

constexpr int C3_C = 64, C3_T = 16, C3_HT = C3_T + 2;            // tile 16x16, halo 18x18
constexpr int C3_PXB = (C3_C + 8) * 2;                            // 144 bytes per pixel in LDS (bank-conflict pad)
// Halo ROW pitch: a multiple of 256 bytes (the 64 banks).  A 16-lane service group of the ds_read_b128 fragment reads
// (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) holds tile columns {0-3, 12-15} of one tile row and {4-11} of
// the next; 9 (the pixel pitch in 16-byte units) is invertible mod 16, so the 16 reads hit 16 different 16-byte bank groups exactly
// when the row-to-row distance is 0 mod 256 B.  With the natural 18 x 144 = 2592 B it was 32 B off: two of the sixteen lanes of
// every group collided (PMC round 2: 47 % of the LDS cycles were bank conflicts).
constexpr int C3_ROWB = ((C3_HT * C3_PXB + 255) / 256) * 256;     // 2816
constexpr int C3_HALO_BYTES = C3_HT * C3_ROWB;                    // 50,688 (three workgroups per CU)
static_assert(256 * C3_PXB <= C3_HALO_BYTES && 3 * (C3_HALO_BYTES + 512) <= 160 * 1024, "output staging tile / occupancy");
constexpr int C3_KS = 9 * C3_C / 16;                              // 36 k-steps of 16
constexpr int C3_RING = 16;

// STAMP: tuning build - cycle stamps of every (workgroup, wave) at the phase boundaries into dbg[workgroup][4 waves][8] (scripts/c64_stamps.py)
template <bool STAMP>
__global__ __launch_bounds__(256, 3) void conv3x3_c64_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wfrag,
                                                             const float* __restrict__ scale, const float* __restrict__ bias,
                                                             bf16_t* __restrict__ y, int H, int W, int act, unsigned long long* dbg) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[C3_HALO_BYTES + 512];
    float* sb_lds = reinterpret_cast<float*>(lds + C3_HALO_BYTES);     // BN scale / shift parked behind the tile (no global loads in the epilogue)
    const float sb_v = (threadIdx.x & 64 ? bias : scale)[threadIdx.x & 63];      // (unguarded load, written to LDS in front of the first barrier)
    unsigned long long ts[8];
    auto stamp = [&](int i) { if constexpr (STAMP) ts[i] = __builtin_readcyclecounter(); };
    stamp(0);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z, y0 = blockIdx.y * C3_T, x0 = blockIdx.x * C3_T;
    const int nt = wave & 1, rt0 = (wave >> 1) * 4;               // column tile, first of this wave's four row tiles

    // weight ring: k-steps 0..15 of column tile nt in flight before anything else
    const bf16_t* wp = wfrag + ((long long)nt * C3_KS * 64 + lane) * 8;
    bf16x8 ring[C3_RING];
#pragma unroll
    for (int s = 0; s < C3_RING; ++s) ring[s] = *reinterpret_cast<const bf16x8*>(wp + s * 512);

    // ---- halo tile -> LDS (zero outside the image)
    const bf16_t* xb = x + (long long)b * H * W * C3_C;
    {   // all 11 loads of a thread go out before the first LDS write (the rolled load -> ds_write loop exposed one memory round trip
        // per iteration: the workgroup spent ~40 k cycles staging for 4.6 k cycles of MFMA work per wave)
        constexpr int NIT = (C3_HT * C3_HT * 8 + 255) / 256;
        us8 hv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int px = i >> 3, ch = (i & 7) * 8;
            const int iy = y0 - 1 + px / C3_HT, ix = x0 - 1 + px % C3_HT;
            // unconditional load from a clamped pixel, zeroed below where the halo lies outside the image: a GUARDED load compiles to a
            // branch that waits for its own load inside the block (round 6, stem.hip: the loads ran one memory round trip after the other)
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            hv[it] = *reinterpret_cast<const us8*>(xb + ((long long)iyc * W + ixc) * C3_C + ch);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int px = i >> 3;
            const int iy = y0 - 1 + px / C3_HT, ix = x0 - 1 + px % C3_HT;
            const bool in = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            if (i < C3_HT * C3_HT * 8) *reinterpret_cast<us8*>(lds + (px / C3_HT) * C3_ROWB + (px % C3_HT) * C3_PXB + (i & 7) * 16) = in ? hv[it] : us8{};
        }
    }
    if (tid < 128) sb_lds[tid] = sb_v;
    stamp(1);
    __syncthreads();
    stamp(2);

    // ---- implicit GEMM out of the halo: lane's pixel in row tile r is (ty, tx) = (2r + (l31 >> 4), l31 & 15)
    f32x16 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
    int a_off[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ty = 2 * (rt0 + r) + (l31 >> 4), tx = l31 & 15;
        a_off[r] = ty * C3_ROWB + tx * C3_PXB + half * 16;
    }
#pragma unroll
    for (int ks = 0; ks < C3_KS; ++ks) {
        const int tap = ks >> 2, kk = ks & 3, kh = tap / 3, kw = tap % 3;
        const int t_off = kh * C3_ROWB + kw * C3_PXB + kk * 32;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(lds + a_off[r] + t_off);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[ks % C3_RING], af, acc[r], 0, 0, 0);
        }
        if (ks + C3_RING < C3_KS) ring[ks % C3_RING] = *reinterpret_cast<const bf16x8*>(wp + (ks + C3_RING) * 512);
        if ((ks & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep the LDS reads of later taps from being hoisted (register pressure)
    }
    stamp(3);
    __syncthreads();                                              // halo is dead: reuse LDS as the [256][72] output staging tile
    stamp(4);

    // ---- BN + activation -> bf16 staging; lane holds pixel (rt0 + r)*32 + l31, channels nt*32 + 8q + 4*half + e
    // (the activation is decided ONCE per 64 values, not per element: apply_act_n, common.h)
    float ev[64];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = nt * 32 + 8 * q + 4 * half;
        const f32x4 s = *reinterpret_cast<const f32x4*>(sb_lds + n), bb = *reinterpret_cast<const f32x4*>(sb_lds + 64 + n);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[r][4 * q + e] * s[e];
                v += bb[e];
                ev[(q * 4 + r) * 4 + e] = v;
            }
    }
    apply_act_n<64>(ev, act);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = nt * 32 + 8 * q + 4 * half;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            us4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(ev[(q * 4 + r) * 4 + e]);
            *reinterpret_cast<us4*>(lds + ((rt0 + r) * 32 + l31) * C3_PXB + n * 2) = o;
        }
    }
    stamp(5);
    __syncthreads();
    stamp(6);
    // ---- store: pixel p = ty*16 + tx; a tile row is 16 px x 128 B = 2 KB contiguous in the NHWC output
    bf16_t* yb = y + (long long)b * H * W * C3_C;
#pragma unroll
    for (int i = 0; i < 256 * 8 / 256; ++i) {
        const int c = tid + i * 256, p = c >> 3, ch = (c & 7) * 8;
        const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
        if (oy < H && ox < W) *reinterpret_cast<us8*>(yb + ((long long)oy * W + ox) * C3_C + ch) = *reinterpret_cast<const us8*>(lds + p * C3_PXB + ch * 2);
    }
    if constexpr (STAMP) {
        stamp(7);
        if (dbg && lane == 0) {
            const long long wg = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            for (int i = 0; i < 8; ++i) dbg[(wg * 4 + wave) * 8 + i] = ts[i];
        }
    }
}

static unsigned long long* g_c64_dbg = nullptr;

}  // namespace nps

extern "C" void nps_c64_debug_buffer(void* buf) { nps::g_c64_dbg = (unsigned long long*)buf; }

extern "C" int nopesac_conv3x3_c64_bf16(const void* x, const void* w_frag, const float* scale, const float* bias, void* y, int B, int H,
                                        int W, int act, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && w_frag && scale && bias && y && B > 0 && H > 0 && W > 0, "conv3x3_c64: bad args");
    NPS_CHECK_ARG(act >= 0 && act <= 3, "conv3x3_c64: bad act %d", act);
    const void* ptrs[] = {x, w_frag, scale, bias, y};
    for (const void* q : ptrs) NPS_CHECK_ARG(((uintptr_t)q & 15) == 0, "conv3x3_c64: pointers must be 16-byte aligned");
    dim3 grid((W + C3_T - 1) / C3_T, (H + C3_T - 1) / C3_T, B);
    if (g_c64_dbg)                                                     // tuning runs only
        hipLaunchKernelGGL(conv3x3_c64_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w_frag, scale, bias,
                           (bf16_t*)y, H, W, act, g_c64_dbg);
    else
        hipLaunchKernelGGL(conv3x3_c64_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w_frag, scale, bias,
                           (bf16_t*)y, H, W, act, (unsigned long long*)nullptr);
    NPS_LAUNCH_RET();
}
