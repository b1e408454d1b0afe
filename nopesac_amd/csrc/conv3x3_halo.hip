// 3x3 / stride 1 / pad 1 convolution, Cin % 64 == 0, Cout % 128 == 0, + folded BN + activation, bf16 (res3 / res4 conv2, the
// 3x3 stacks of the pixel pose net and of the top-down path).
//
// The implicit-GEMM kernels (conv_igemm.hip) fetch, for every 64-channel K-tile, a tap-shifted copy of the same pixels: a
// 128x128 output tile of a 256-channel layer pulls 36 x 16 KB of activations + 590 KB of weights through L2 (the 3x3 256->256
// layer at 60x80 moves 5.7 GB per launch, 14.5 TB/s at 928 TFLOP/s).  Here (generalising conv3x3_c64.hip) a workgroup owns a
// TH x TW pixel tile and 128 output channels and walks the input channels in chunks of 64: per chunk the (TH+2) x (TW+2) halo
// of those 64 channels is loaded ONCE into LDS and all nine taps are shifted reads of it - 9x fewer activation bytes; the
// weights stream fragment-major from L2 through a rolling register ring, wave w owning output-channel tile w, every fragment
// feeding TH*TW/32 MFMAs (8 for the 16x16 tile).  Between two barriers a wave issues 36 x TH*TW/32 MFMAs.
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

constexpr int CH_PXB = (64 + 8) * 2;                               // 144 bytes per halo pixel (64 channels + bank pad)
constexpr int CH_RING = 12;

struct HaloArgs {
    const bf16_t* x; const bf16_t* wfrag; const float* scale; const float* bias; bf16_t* y;
    int H, W, Cin, Cout, act;
};

template <int TH, int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(const HaloArgs p) {
    constexpr int HW_ = TW + 2, NPX = (TH + 2) * HW_, RT = TH * TW / 32;
    constexpr int HALO_BYTES = NPX * CH_PXB, STAGE_BYTES = TH * TW * CH_PXB;
    constexpr int LDS_BYTES = HALO_BYTES > STAGE_BYTES ? HALO_BYTES : STAGE_BYTES;
    static_assert(TH * TW % 32 == 0 && (TW & (TW - 1)) == 0 && 32 % TW == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (p.W + TW - 1) / TW;
    const int b = blockIdx.z, n_blk = blockIdx.y, y0 = (blockIdx.x / tiles_x) * TH, x0 = (blockIdx.x % tiles_x) * TW;
    const int nchunk = p.Cin / 64, kf_total = 9 * p.Cin / 16;
    // this wave's weight fragments: output-channel tile n_blk*4 + wave; k-step (tap, chunk cc, kk) -> (tap*Cin + cc*64)/16 + kk
    const bf16_t* wbase = p.wfrag + ((long long)(n_blk * 4 + wave) * kf_total * 64 + lane) * 8;
    auto wptr = [&](int cc, int s) { return wbase + (long long)((s >> 2) * (p.Cin / 16) + cc * 4 + (s & 3)) * 512; };   // s = tap*4 + kk
    bf16x8 ring[CH_RING];
#pragma unroll
    for (int s = 0; s < CH_RING; ++s) ring[s] = *reinterpret_cast<const bf16x8*>(wptr(0, s));

    f32x16 acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
    int a_off[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int pp = r * 32 + l31, ty = pp / TW, tx = pp % TW;
        a_off[r] = (ty * HW_ + tx) * CH_PXB + half * 16;
    }
    const bf16_t* xb = p.x + (long long)b * p.H * p.W * p.Cin;
    for (int cc = 0; cc < nchunk; ++cc) {
        if (cc) __syncthreads();                                  // every wave finished the previous chunk's taps
        for (int i = tid; i < NPX * 8; i += 256) {                // halo of channels cc*64 .. +64 (zero outside the image)
            const int px = i >> 3, ch = (i & 7) * 8;
            const int iy = y0 - 1 + px / HW_, ix = x0 - 1 + px % HW_;
            us8 v = us8{};
            if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                v = *reinterpret_cast<const us8*>(xb + ((long long)iy * p.W + ix) * p.Cin + cc * 64 + ch);
            *reinterpret_cast<us8*>(lds + px * CH_PXB + ch * 2) = v;
        }
        __syncthreads();
        const bool more = cc + 1 < nchunk;
        // explicit software pipeline: the A fragments of step s+1 are read from LDS before the MFMAs of step s are issued
        bf16x8 af[2][RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) af[0][r] = *reinterpret_cast<const bf16x8*>(lds + a_off[r]);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + 1 < 36) {
                const int tap = (s + 1) >> 2, kk = (s + 1) & 3, kh = tap / 3, kw = tap % 3;
                const int t_off = (kh * HW_ + kw) * CH_PXB + kk * 32;
#pragma unroll
                for (int r = 0; r < RT; ++r) af[(s + 1) & 1][r] = *reinterpret_cast<const bf16x8*>(lds + a_off[r] + t_off);
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[s % CH_RING], af[s & 1][r], acc[r], 0, 0, 0);
            // refill the slot just consumed with the fragment CH_RING steps ahead (36 % 12 == 0: the slot pattern repeats per chunk)
            if (s + CH_RING < 36) ring[s % CH_RING] = *reinterpret_cast<const bf16x8*>(wptr(cc, s + CH_RING));
            else if (more) ring[s % CH_RING] = *reinterpret_cast<const bf16x8*>(wptr(cc + 1, s + CH_RING - 36));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- BN + activation -> bf16 staging, 64 channels (= waves 2*pass, 2*pass+1) per pass; stores are 128-byte runs per pixel
    const int n_wave = (n_blk * 4 + wave) * 32;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
        if ((wave >> 1) == pass) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = (wave & 1) * 32 + 8 * q + 4 * half, n = n_wave + 8 * q + 4 * half;
                const f32x4 s = *reinterpret_cast<const f32x4*>(p.scale + n), bb = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    float ev[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[r][4 * q + e] * s[e];
                        v += bb[e];
                        ev[e] = v;
                    }
                    apply_act_n<4>(ev, p.act);                 // (one wave-uniform decision per four values, not a branch per element)
                    us4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(ev[e]);
                    *reinterpret_cast<us4*>(lds + (r * 32 + l31) * CH_PXB + nl * 2) = o;
                }
            }
        }
        __syncthreads();
        bf16_t* yb = p.y + (long long)b * p.H * p.W * p.Cout + n_blk * 128 + pass * 64;
        for (int c = tid; c < TH * TW * 8; c += 256) {
            const int pp = c >> 3, ch = (c & 7) * 8;
            const int oy = y0 + pp / TW, ox = x0 + pp % TW;
            if (oy < p.H && ox < p.W)
                *reinterpret_cast<us8*>(yb + ((long long)oy * p.W + ox) * p.Cout + ch) = *reinterpret_cast<const us8*>(lds + pp * CH_PXB + ch * 2);
        }
    }
}

}  // namespace nps

extern "C" int nopesac_conv3x3_halo_bf16(const void* x, const void* w_frag, const float* scale, const float* bias, void* y, int B, int H,
                                         int W, int Cin, int Cout, int act, int tile, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && w_frag && scale && bias && y && B > 0 && H > 0 && W > 0, "conv3x3_halo: bad args");
    NPS_CHECK_ARG(Cin > 0 && Cin % 64 == 0 && Cout > 0 && Cout % 128 == 0, "conv3x3_halo: needs Cin %% 64 == 0 and Cout %% 128 == 0");
    NPS_CHECK_ARG(act >= 0 && act <= 3, "conv3x3_halo: bad act %d", act);
    NPS_CHECK_ARG(tile == 0 || tile == 1, "conv3x3_halo: tile must be 0 (16x16 pixels) or 1 (16 rows x 8 columns)");
    const void* ptrs[] = {x, w_frag, scale, bias, y};
    for (const void* q : ptrs) NPS_CHECK_ARG(((uintptr_t)q & 15) == 0, "conv3x3_halo: pointers must be 16-byte aligned");
    HaloArgs a;
    a.x = (const bf16_t*)x; a.wfrag = (const bf16_t*)w_frag; a.scale = scale; a.bias = bias; a.y = (bf16_t*)y;
    a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = act;
    if (tile == 0) {
        dim3 grid(((W + 15) / 16) * ((H + 15) / 16), Cout / 128, B);
        hipLaunchKernelGGL((conv3x3_halo_kernel<16, 16>), grid, dim3(256), 0, (hipStream_t)stream, a);
    } else {
        dim3 grid(((W + 7) / 8) * ((H + 15) / 16), Cout / 128, B);
        hipLaunchKernelGGL((conv3x3_halo_kernel<16, 8>), grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    NPS_LAUNCH_RET();
}
