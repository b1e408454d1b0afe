// Shared pieces of the implicit-GEMM conv kernels (conv_igemm.hip, conv_p8.hip): parameter block, vector typedefs and the fused
// epilogue (f32 tile -> LDS -> 8-channel row chunks -> scale / shift / residual / activation -> store).  Not a public header.
#pragma once
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

struct ConvParams {
    const void* x; const void* w; const float* scale; const float* bias; const void* res; void* y;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW;
    long long x_cs, y_cs, r_cs, w_bs;
    int M, N, K;       // M = rows per grid.y slice
    int rows_per_b;    // OH*OW
    int batched;       // 1: grid.y = batch index, per-batch weights
    int act, out_dt, res_after, epi_vec, use_glds, force, dense1x1, bias_bs;
    int kmajor;        // conv_igemm_glds_kernel: 1 = channel-major K order (the KH*KW taps of one K-tile's channel slice back to back)
    int tiles_m, tiles_n;
    int dbg_tile;
    unsigned long long* dbg;   // tuning builds only: per-workgroup cycle stamps (nullptr in product launches)
    void* sk_ws;               // conv_igemm_p8_kernel<.., SK>: stream-K workspace (arrival counters + partial-tile slabs)
    int sk_ws_bytes;
    int ksplit;                // conv_igemm_p8n_kernel<.., SPLIT>: K slices per output tile (f32 partial tiles in sk_ws, summed by p8n_split_reduce_kernel)
};

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> { static constexpr int BK = 32; static constexpr int VECW = 8; };
template <> struct Cfg<float> { static constexpr int BK = 16; static constexpr int VECW = 4; };

template <typename T, int V> struct VecT;
template <> struct VecT<bf16_t, 8> { typedef us8 type; };
template <> struct VecT<bf16_t, 4> { typedef us4 type; };
template <> struct VecT<bf16_t, 1> { typedef unsigned short type; };
template <> struct VecT<float, 4> { typedef f32x4 type; };
template <> struct VecT<float, 1> { typedef float type; };

// Shared epilogue (see the comment inside): acc -> LDS (f32) -> 8-channel chunks -> scale/shift/residual/act -> store.
// LDS-only workgroup barrier: orders this wave's LDS accesses and synchronises WITHOUT the vmcnt(0) that __syncthreads() carries -
// between the passes of an epilogue that would wait for the previous pass's global STORES to be acknowledged (microseconds under load).
#define NPS_LDS_SYNC()                                         \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    } while (0)

template <int BM, int BN, int TM, int TN, int WAVES_M = 2, int WAVES_N = 2>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[TM][TN], float* epi, int lds_bytes, const ConvParams& p, int m0, int n0,
                                              int bz, int wm, int wn, int lane, int tid) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int NTHREADS = WAVES_M * WAVES_N * 64;
    // ---- epilogue.  The MFMAs were issued with the operands swapped (weights as the row operand), so lane l
    // holds, for pixel (l&31), 4 runs of 4 consecutive channels per 32x32 tile.  The f32 tile is staged through
    // LDS (EN = 64 columns per pass, rows padded by 4 floats -> conflict-free ds_write_b128) and re-read as
    // 8-channel row chunks, so scale/shift, residual and the output are 16/32-byte accesses that cover whole
    // 128/256-byte channel runs per row (the layers with small K are HBM-bound: this is what has to stream).
    constexpr int EN = BN > 64 ? 64 : BN;
    constexpr int ELD = EN + 4;
    constexpr int NPASS = BN / EN;
    const bool vec_ok = p.epi_vec;
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        const int c_wave = wn * WN - pass * EN;          // first column of this wave inside the pass window
        if (c_wave >= 0 && c_wave < EN) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = wm * WM + i * 32 + (lane & 31);
                        const int c = c_wave + j * 32 + 8 * q + 4 * (lane >> 5);
                        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *(f32x4*)(epi + row * ELD + c) = v;
                    }
        }
        NPS_LDS_SYNC();                                  // staged tile visible (the K loop ended with a full barrier: no DMA in flight)
        for (int idx = tid; idx < BM * (EN / 8); idx += NTHREADS) {
            const int row = idx / (EN / 8), ch = (idx % (EN / 8)) * 8;
            const int m = m0 + row, n = n0 + pass * EN + ch;
            if (m >= p.M || n >= p.N) continue;
            const long long pix = (long long)m + (long long)bz * p.rows_per_b;
            float v[8];
            *(f32x4*)(v) = *(const f32x4*)(epi + row * ELD + ch);
            *(f32x4*)(v + 4) = *(const f32x4*)(epi + row * ELD + ch + 4);
            if (vec_ok && n + 8 <= p.N) {
                if (p.scale) {
                    const f32x4 s0 = *(const f32x4*)(p.scale + n), s1 = *(const f32x4*)(p.scale + n + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] *= s0[e]; v[4 + e] *= s1[e]; }
                }
                if (p.bias) {
                    const float* bp = p.bias + (long long)bz * p.bias_bs + n;
                    const f32x4 b0 = *(const f32x4*)(bp), b1 = *(const f32x4*)(bp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                }
                float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (p.res) {
                    if (p.out_dt == NPS_DT_F32) {
                        *(f32x4*)(rv) = *(const f32x4*)((const float*)p.res + pix * p.r_cs + n);
                        *(f32x4*)(rv + 4) = *(const f32x4*)((const float*)p.res + pix * p.r_cs + n + 4);
                    } else {
                        const us8 r8 = *(const us8*)((const bf16_t*)p.res + pix * p.r_cs + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) rv[e] = bf16_to_f32(r8[e]);
                    }
                }
                act_residual8(v, rv, p.act, p.res_after);
                if (p.out_dt == NPS_DT_F32) {
                    float* yp = (float*)p.y + pix * p.y_cs + n;
                    *(f32x4*)(yp) = *(const f32x4*)(v);
                    *(f32x4*)(yp + 4) = *(const f32x4*)(v + 4);
                } else if (p.out_dt == NPS_DT_FP8) {
                    *(uint2*)((unsigned char*)p.y + pix * p.y_cs + n) = f32x8_to_fp8(v);
                } else {
                    us8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(v[e]);
                    *(us8*)((bf16_t*)p.y + pix * p.y_cs + n) = o;
                }
            } else {
                for (int e = 0; e < 8 && n + e < p.N; ++e) {
                    float x = v[e] * (p.scale ? p.scale[n + e] : 1.f) + (p.bias ? p.bias[(long long)bz * p.bias_bs + n + e] : 0.f);
                    float r = 0.f;
                    if (p.res)
                        r = (p.out_dt == NPS_DT_F32) ? ((const float*)p.res)[pix * p.r_cs + n + e]
                                                     : bf16_to_f32(((const bf16_t*)p.res)[pix * p.r_cs + n + e]);
                    x = p.res_after ? apply_act(x, p.act) + r : apply_act(x + r, p.act);
                    if (p.out_dt == NPS_DT_F32) ((float*)p.y)[pix * p.y_cs + n + e] = x;
                    else ((bf16_t*)p.y)[pix * p.y_cs + n + e] = f32_to_bf16(x);
                }
            }
        }
        if (pass + 1 < NPASS) NPS_LDS_SYNC();            // every wave is done reading the staging rows; the stores stay in flight
    }
}

typedef __attribute__((address_space(3))) void* lptr_t;

}  // namespace nps
