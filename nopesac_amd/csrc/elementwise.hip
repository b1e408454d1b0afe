// HBM-bound NHWC / row kernels around the GEMMs: preprocessing, pooling, up-sampling, GroupNorm,
// LayerNorm, row softmax, small re-layouts.  All are streaming kernels: channels are the fastest
// dimension, so a wave touches contiguous bytes; reductions use wave shuffles (+LDS across waves).
#include <algorithm>

#include "common.h"

namespace nps {

// ---------------------------------------------------------------------------------------------
// (x - mean)/std, NCHW f32 -> NHWC(Cpad)
template <typename T>
__global__ void preprocess_kernel(const float* __restrict__ x, T* __restrict__ y, const float* __restrict__ mean,
                                  const float* __restrict__ stdv, int B, int C, int H, int W, int Cpad) {
    const long long npix = (long long)B * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / ((long long)H * W), hw = i % ((long long)H * W);
        for (int c = 0; c < Cpad; ++c) {
            float v = 0.f;
            if (c < C) v = (x[(b * C + c) * H * W + hw] - mean[c]) / stdv[c];
            y[i * Cpad + c] = from_f32<T>(v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Vector helpers: V consecutive channels per thread (16 bytes for V = 8 bf16 / 4 f32).
template <typename T, int V> struct Pack { T v[V]; };
template <typename T, int V> __device__ __forceinline__ void load_vec(const T* p, float out[V]) {
    if constexpr (V == 1) { out[0] = to_f32<T>(*p); }
    else {
        typedef Pack<T, V> __attribute__((aligned(sizeof(T) * V))) PV;
        const PV pk = *reinterpret_cast<const PV*>(p);
#pragma unroll
        for (int i = 0; i < V; ++i) out[i] = to_f32<T>(pk.v[i]);
    }
}
template <typename T, int V> __device__ __forceinline__ void store_vec(T* p, const float in[V]) {
    if constexpr (V == 1) { *p = from_f32<T>(in[0]); }
    else {
        typedef Pack<T, V> __attribute__((aligned(sizeof(T) * V))) PV;
        PV pk;
#pragma unroll
        for (int i = 0; i < V; ++i) pk.v[i] = from_f32<T>(in[i]);
        *reinterpret_cast<PV*>(p) = pk;
    }
}

template <typename T, int V>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C, int K,
                               int stride, int pad, int OH, int OW) {
    const int CV = C / V;
    const long long total = (long long)B * OH * OW * CV;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        long long pix = i / CV;
        const int ow = pix % OW; pix /= OW;
        const int oh = pix % OH;
        const int b = pix / OH;
        float m[V];
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = -INFINITY;
        for (int kh = 0; kh < K; ++kh) {
            const int ih = oh * stride - pad + kh;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int kw = 0; kw < K; ++kw) {
                const int iw = ow * stride - pad + kw;
                if ((unsigned)iw >= (unsigned)W) continue;
                float t[V];
                load_vec<T, V>(x + (((long long)b * H + ih) * W + iw) * C + c, t);
#pragma unroll
                for (int e = 0; e < V; ++e) m[e] = fmaxf(m[e], t[e]);
            }
        }
        store_vec<T, V>(y + (i / CV) * C + c, m);
    }
}

// ---------------------------------------------------------------------------------------------
// bilinear x2, align_corners=False: src = 0.5*(dst+0.5)-0.5, clamped at 0 (PyTorch
// area_pixel_compute_source_index), i1 = min(i0+1, size-1)
template <typename T, int V>
__global__ void upsample_bilinear2x_kernel(const T* __restrict__ x, const T* __restrict__ addend, T* __restrict__ y,
                                           int B, int H, int W, int C, int act) {
    const int OH = 2 * H, OW = 2 * W, CV = C / V;
    const long long total = (long long)B * OH * OW * CV;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        long long pix = i / CV;
        const long long opix = pix;
        const int ow = pix % OW; pix /= OW;
        const int oh = pix % OH;
        const int b = pix / OH;
        float sy = fmaxf(0.5f * (oh + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ow + 0.5f) - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
        const long long base = (long long)b * H * W;
        float v00[V], v01[V], v10[V], v11[V], o[V];
        load_vec<T, V>(x + (base + (long long)y0 * W + x0) * C + c, v00);
        load_vec<T, V>(x + (base + (long long)y0 * W + x1) * C + c, v01);
        load_vec<T, V>(x + (base + (long long)y1 * W + x0) * C + c, v10);
        load_vec<T, V>(x + (base + (long long)y1 * W + x1) * C + c, v11);
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = hy * (hx * v00[e] + lx * v01[e]) + ly * (hx * v10[e] + lx * v11[e]);
        apply_act_n<V>(o, act);
        if (addend) {
            float a[V];
            load_vec<T, V>(addend + opix * C + c, a);
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] += a[e];
        }
        store_vec<T, V>(y + opix * C + c, o);
    }
}

template <typename T, int V>
__global__ void upsample_nearest2x_add_kernel(const T* __restrict__ x, const T* __restrict__ lat, T* __restrict__ y,
                                              int B, int H, int W, int C) {
    const int OH = 2 * H, OW = 2 * W, CV = C / V;
    const long long total = (long long)B * OH * OW * CV;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * V;
        long long pix = i / CV;
        const long long opix = pix;
        const int ow = pix % OW; pix /= OW;
        const int oh = pix % OH;
        const int b = pix / OH;
        float a[V], l[V];
        load_vec<T, V>(x + (((long long)b * H + (oh >> 1)) * W + (ow >> 1)) * C + c, a);
        load_vec<T, V>(lat + opix * C + c, l);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += l[e];
        store_vec<T, V>(y + opix * C + c, a);
    }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm on NHWC: one workgroup per (image, group-chunk); two passes (mean, then centred variance)
// over an L2-resident map, then the normalise pass.  blockDim = 256, each thread owns channel
// (tid % CPB) of the chunk and strides over pixels.
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, int HW,
                                                        int C, int G, float eps, int act, int groups_per_block) {
    const int cpg = C / G;                       // channels per group
    const int CPB = cpg * groups_per_block;      // channels per block (<= 256, divides 256)
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int c_local = threadIdx.x % CPB, prow = threadIdx.x / CPB, nprow = 256 / CPB;
    const int c = chunk * CPB + c_local;
    const int g_local = c_local / cpg;
    const T* xb = x + (long long)b * HW * C;
    T* yb = y + (long long)b * HW * C;
    __shared__ float red[256];
    __shared__ float stat[64];
    const float cnt = (float)HW * cpg;
    // pass 1: mean
    float s = 0.f;
    for (int p = prow; p < HW; p += nprow) s += to_f32<T>(xb[(long long)p * C + c]);
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < groups_per_block) {
        float t = 0.f;
        for (int i = 0; i < 256; ++i)
            if ((i % CPB) / cpg == (int)threadIdx.x) t += red[i];
        stat[threadIdx.x] = t / cnt;
    }
    __syncthreads();
    const float mean = stat[g_local];
    // pass 2: variance
    float s2 = 0.f;
    for (int p = prow; p < HW; p += nprow) {
        const float d = to_f32<T>(xb[(long long)p * C + c]) - mean;
        s2 += d * d;
    }
    __syncthreads();
    red[threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < groups_per_block) {
        float t = 0.f;
        for (int i = 0; i < 256; ++i)
            if ((i % CPB) / cpg == (int)threadIdx.x) t += red[i];
        stat[32 + threadIdx.x] = rsqrtf(t / cnt + eps);
    }
    __syncthreads();
    const float rstd = stat[32 + g_local];
    const float ga = gamma[c] * rstd, be = beta[c] - mean * gamma[c] * rstd;
    for (int p = prow; p < HW; p += nprow) {
        const float v = to_f32<T>(xb[(long long)p * C + c]) * ga + be;
        yb[(long long)p * C + c] = from_f32<T>(apply_act(v, act));
    }
}

// ---------------------------------------------------------------------------------------------
// Fast GroupNorm path (C % 8 == 0, C/8 divides 256): split statistics + apply, both with 16-byte channel
// vectors.  Partial sums go to a caller-provided workspace [B][GN_SPLITS][G][2] and are combined in a fixed
// order (deterministic, no float atomics).
constexpr int GN_SPLITS = 16;

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ part, int HW, int C, int G) {
    const int cv = C / 8, rows_per_iter = 256 / cv;
    const int c8 = threadIdx.x % cv, prow = threadIdx.x / cv;
    const int b = blockIdx.y, sp = blockIdx.x;
    const int p0 = (int)((long long)HW * sp / GN_SPLITS), p1 = (int)((long long)HW * (sp + 1) / GN_SPLITS);
    const T* xb = x + (long long)b * HW * C;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    for (int p = p0 + prow; p < p1; p += rows_per_iter) {
        float v[8];
        load_vec<T, 8>(xb + (long long)p * C + c8 * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += v[e]; ss[e] += v[e] * v[e]; }
    }
    __shared__ float red[2][256][8 + 1];
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][threadIdx.x][e] = s[e]; red[1][threadIdx.x][e] = ss[e]; }
    __syncthreads();
    __shared__ float chs[2][2048];
    for (int c = threadIdx.x; c < C; c += 256) {     // per-channel totals over the pixel rows of this block
        float a = 0.f, q = 0.f;
        for (int r = 0; r < rows_per_iter; ++r) { a += red[0][r * cv + c / 8][c % 8]; q += red[1][r * cv + c / 8][c % 8]; }
        chs[0][c] = a; chs[1][c] = q;
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = threadIdx.x; g < G; g += 256) {
        float a = 0.f, q = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += chs[0][c]; q += chs[1][c]; }
        float* o = part + (((long long)b * GN_SPLITS + sp) * G + g) * 2;
        o[0] = a; o[1] = q;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ part,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, int HW, int C, int G, float eps, int act) {
    // per-channel affine of this image, once per workgroup: y = x * a[c] + d[c]  (a = rstd_g * gamma_c, d = beta_c - mean_g * a)
    __shared__ float s_mean[256], s_rstd[256];
    __shared__ float s_a[2048], s_d[2048];
    const int b = blockIdx.y;
    const int cpg = C / G;
    for (int g = threadIdx.x; g < G; g += 256) {
        float a = 0.f, q = 0.f;
        for (int sp = 0; sp < GN_SPLITS; ++sp) {
            const float* o = part + (((long long)b * GN_SPLITS + sp) * G + g) * 2;
            a += o[0]; q += o[1];
        }
        const float cnt = (float)HW * cpg;
        const float mean = a / cnt;
        const float var = fmaxf(q / cnt - mean * mean, 0.f);
        s_mean[g] = mean; s_rstd[g] = rsqrtf(var + eps);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / cpg;
        const float a = s_rstd[g] * gamma[c];
        s_a[c] = a;
        s_d[c] = beta[c] - s_mean[g] * a;
    }
    __syncthreads();
    const int cv = C / 8;                                  // 256 % cv == 0: a thread keeps its 8-channel chunk across iterations
    const int c = (int)(threadIdx.x % cv) * 8;
    float a8[8], d8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a8[e] = s_a[c + e]; d8[e] = s_d[c + e]; }
    const int rows_per_iter = 256 / cv;
    const T* xb = x + (long long)b * HW * C;
    T* yb = y + (long long)b * HW * C;
    for (long long p = (long long)blockIdx.x * rows_per_iter + threadIdx.x / cv; p < HW; p += (long long)gridDim.x * rows_per_iter) {
        float v[8];
        load_vec<T, 8>(xb + p * C + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * a8[e] + d8[e];
        apply_act_n<8>(v, act);
        store_vec<T, 8>(yb + p * C + c, v);
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over D (one wave per row), optional residual, optional second output y + addend
// (y16 / y2_16: the same two results rounded to bf16 for the GEMMs that consume them - nullable, like y / y2)
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, const float* __restrict__ addend,
                                                        int addend_rows, float* __restrict__ y2, bf16_t* __restrict__ y16,
                                                        bf16_t* __restrict__ y2_16, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int per = D / 64;  // <= 16
    float v[16];
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
        const int d = lane + i * 64;
        float t = x[(long long)row * D + d];
        if (res) t += res[(long long)row * D + d];
        v[i] = t;
        s += t;
    }
    const float mean = wave_sum(s) / D;
    float s2 = 0.f;
    for (int i = 0; i < per; ++i) {
        const float d = v[i] - mean;
        s2 += d * d;
    }
    const float rstd = rsqrtf(wave_sum(s2) / D + eps);
    for (int i = 0; i < per; ++i) {
        const int d = lane + i * 64;
        const float o = (v[i] - mean) * rstd * gamma[d] + beta[d];
        if (y) y[(long long)row * D + d] = o;
        if (y16) y16[(long long)row * D + d] = f32_to_bf16(o);
        if (y2 || y2_16) {
            const float o2 = o + addend[(long long)(row % addend_rows) * D + d];
            if (y2) y2[(long long)row * D + d] = o2;
            if (y2_16) y2_16[(long long)row * D + d] = f32_to_bf16(o2);
        }
    }
}

__global__ void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                long long total, int D, int b_rows) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / D;
        out[i] = a[i] + b[(row % b_rows) * D + (i % D)];
    }
}

// a and a + b (b row index = row % b_rows) as bf16: the two operand tensors of the first encoder layer's q|k and v GEMMs in one pass
// over the f32 rows (was: a cast, an add and another cast); 4 elements per thread, D % 4 == 0.
__global__ void add_rows_bf16_kernel(const float* __restrict__ a, const float* __restrict__ b, bf16_t* __restrict__ a16,
                                     bf16_t* __restrict__ ab16, long long total4, int D, int b_rows) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i * 4, row = e / D;
        const int d = (int)(e - row * D);
        const float4 av = *reinterpret_cast<const float4*>(a + e);
        const float4 bv = *reinterpret_cast<const float4*>(b + (row % b_rows) * D + d);
        ushort4 o, o2;
        o.x = f32_to_bf16(av.x); o.y = f32_to_bf16(av.y); o.z = f32_to_bf16(av.z); o.w = f32_to_bf16(av.w);
        o2.x = f32_to_bf16(av.x + bv.x); o2.y = f32_to_bf16(av.y + bv.y); o2.z = f32_to_bf16(av.z + bv.z); o2.w = f32_to_bf16(av.w + bv.w);
        *reinterpret_cast<ushort4*>(a16 + e) = o;
        *reinterpret_cast<ushort4*>(ab16 + e) = o2;
    }
}

// row softmax, one wave per row, D <= 1024
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[16];
    float m = -INFINITY;
    const int per = (D + 63) / 64;
    for (int i = 0; i < per; ++i) {
        const int d = lane + i * 64;
        v[i] = d < D ? x[(long long)row * D + d] : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
        v[i] = expf(v[i] - m);
        s += v[i];
    }
    s = wave_sum(s);
    for (int i = 0; i < per; ++i) {
        const int d = lane + i * 64;
        if (d < D) y[(long long)row * D + d] = v[i] / s;
    }
}

// the same softmax written into rows of `out_ld` >= D elements, f32 or bf16, columns D .. out_ld-1 zero: the affinity volume in the
// layout and type the branch convs read (was: softmax, a zero fill of the padded tensor and a strided cast copy)
template <typename TO>
__global__ __launch_bounds__(256) void softmax_rows_pad_kernel(const float* __restrict__ x, TO* __restrict__ y, int rows, int D, int out_ld) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[16];
    float m = -INFINITY;
    const int per = (D + 63) / 64;
    for (int i = 0; i < per; ++i) {
        const int d = lane + i * 64;
        v[i] = d < D ? x[(long long)row * D + d] : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int i = 0; i < per; ++i) {
        v[i] = expf(v[i] - m);
        s += v[i];
    }
    s = wave_sum(s);
    const int per_o = (out_ld + 63) / 64;
    for (int i = 0; i < per_o; ++i) {
        const int d = lane + i * 64;
        if (d >= out_ld) continue;
        const float r = (d < D && i < per) ? v[i] / s : 0.f;
        if constexpr (sizeof(TO) == 2) y[(long long)row * out_ld + d] = f32_to_bf16(r);
        else y[(long long)row * out_ld + d] = r;
    }
}

__global__ void transpose_hw_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W,
                                         int C) {
    const long long total = (long long)B * H * W * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = i % C;
        long long r = i / C;
        const int hw = r % (H * W);
        const int b = r / (H * W);
        const int wi = hw / H, hi = hw % H;  // output row index = w*H + h
        y[i] = x[(((long long)b * H + hi) * W + wi) * C + c];
    }
}

__global__ void normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int D, int canon) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s += x[row * D + d] * x[row * D + d];
    const float n = fmaxf(sqrtf(s), 1e-12f);
    float sign = 1.f;
    if (canon && x[row * D] / n < 0.f) sign = -1.f;
    for (int d = 0; d < D; ++d) y[row * D + d] = sign * (x[row * D + d] / n);
}

static inline bool vec_aligned(const void* a, const void* b) {
    return ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
}

static inline int grid_for(long long total, int block = 256) {
    long long g = (total + block - 1) / block;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace nps

using namespace nps;

extern "C" int nopesac_preprocess_nchw_to_nhwc(const float* x, void* y, const float* mean, const float* stdv, int B,
                                               int C, int H, int W, int Cpad, int out_dt, void* stream) {
    NPS_CHECK_ARG(x && y && mean && stdv && B > 0 && C > 0 && Cpad >= C && H > 0 && W > 0, "preprocess: bad args");
    const int g = grid_for((long long)B * H * W);
    if (out_dt == NPS_DT_BF16)
        hipLaunchKernelGGL(preprocess_kernel<bf16_t>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, mean, stdv, B, C, H, W, Cpad);
    else
        hipLaunchKernelGGL(preprocess_kernel<float>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, (float*)y, mean, stdv, B, C, H, W, Cpad);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_maxpool_nhwc(const void* x, void* y, int B, int H, int W, int C, int K, int stride, int pad,
                                    int dt, void* stream) {
    NPS_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && stride > 0, "maxpool: bad args");
    const int OH = (H + 2 * pad - K) / stride + 1, OW = (W + 2 * pad - K) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    if (dt == NPS_DT_BF16) {
        if (C % 8 == 0 && vec_aligned(x, y))
            hipLaunchKernelGGL((maxpool_kernel<bf16_t, 8>), dim3(grid_for((long long)B * OH * OW * C / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, K, stride, pad, OH, OW);
        else
            hipLaunchKernelGGL((maxpool_kernel<bf16_t, 1>), dim3(grid_for((long long)B * OH * OW * C)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, K, stride, pad, OH, OW);
    } else {
        if (C % 4 == 0 && vec_aligned(x, y))
            hipLaunchKernelGGL((maxpool_kernel<float, 4>), dim3(grid_for((long long)B * OH * OW * C / 4)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C, K, stride, pad, OH, OW);
        else
            hipLaunchKernelGGL((maxpool_kernel<float, 1>), dim3(grid_for((long long)B * OH * OW * C)), dim3(256), 0, st, (const float*)x, (float*)y, B, H, W, C, K, stride, pad, OH, OW);
    }
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_upsample2x_bilinear_nhwc(const void* x, const void* addend, void* y, int B, int H, int W,
                                                int C, int act, int dt, void* stream) {
    NPS_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0, "upsample_bilinear: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)B * 4 * H * W * C;
    if (dt == NPS_DT_BF16) {
        if (C % 8 == 0 && vec_aligned(x, y) && vec_aligned(addend, y))
            hipLaunchKernelGGL((upsample_bilinear2x_kernel<bf16_t, 8>), dim3(grid_for(n / 8)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)addend, (bf16_t*)y, B, H, W, C, act);
        else
            hipLaunchKernelGGL((upsample_bilinear2x_kernel<bf16_t, 1>), dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)addend, (bf16_t*)y, B, H, W, C, act);
    } else {
        if (C % 4 == 0 && vec_aligned(x, y) && vec_aligned(addend, y))
            hipLaunchKernelGGL((upsample_bilinear2x_kernel<float, 4>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const float*)x, (const float*)addend, (float*)y, B, H, W, C, act);
        else
            hipLaunchKernelGGL((upsample_bilinear2x_kernel<float, 1>), dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, (const float*)addend, (float*)y, B, H, W, C, act);
    }
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_upsample2x_nearest_add_nhwc(const void* x, const void* lateral, void* y, int B, int H, int W,
                                                   int C, int dt, void* stream) {
    NPS_CHECK_ARG(x && lateral && y && B > 0 && H > 0 && W > 0 && C > 0, "upsample_nearest: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)B * 4 * H * W * C;
    if (dt == NPS_DT_BF16) {
        if (C % 8 == 0 && vec_aligned(x, y) && vec_aligned(lateral, y))
            hipLaunchKernelGGL((upsample_nearest2x_add_kernel<bf16_t, 8>), dim3(grid_for(n / 8)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)lateral, (bf16_t*)y, B, H, W, C);
        else
            hipLaunchKernelGGL((upsample_nearest2x_add_kernel<bf16_t, 1>), dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)lateral, (bf16_t*)y, B, H, W, C);
    } else {
        if (C % 4 == 0 && vec_aligned(x, y) && vec_aligned(lateral, y))
            hipLaunchKernelGGL((upsample_nearest2x_add_kernel<float, 4>), dim3(grid_for(n / 4)), dim3(256), 0, st, (const float*)x, (const float*)lateral, (float*)y, B, H, W, C);
        else
            hipLaunchKernelGGL((upsample_nearest2x_add_kernel<float, 1>), dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, (const float*)lateral, (float*)y, B, H, W, C);
    }
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_groupnorm_nhwc(const void* x, const float* gamma, const float* beta, void* y, int B, int HW,
                                      int C, int G, float eps, int act, int dt, float* workspace, void* stream) {
    NPS_CHECK_ARG(x && gamma && beta && y && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "groupnorm: bad args");
    const int cpg = C / G;
    if (workspace && C % 8 == 0 && 256 % (C / 8) == 0 && C <= 2048 && G <= 256 && vec_aligned(x, y)) {
        hipStream_t st = (hipStream_t)stream;
        // apply: few, fat workgroups (each pays the statistics prologue once)
        dim3 g1(GN_SPLITS, B), g2((unsigned)std::max<long long>(1, std::min<long long>(((long long)HW * (C / 8) + 1023) / 1024, 48)), B);
        if (dt == NPS_DT_BF16) {
            hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, g1, dim3(256), 0, st, (const bf16_t*)x, workspace, HW, C, G);
            hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, g2, dim3(256), 0, st, (const bf16_t*)x, workspace, gamma, beta, (bf16_t*)y, HW, C, G, eps, act);
        } else {
            hipLaunchKernelGGL(gn_stats_kernel<float>, g1, dim3(256), 0, st, (const float*)x, workspace, HW, C, G);
            hipLaunchKernelGGL(gn_apply_kernel<float>, g2, dim3(256), 0, st, (const float*)x, workspace, gamma, beta, (float*)y, HW, C, G, eps, act);
        }
        NPS_LAUNCH_RET();
    }
    NPS_CHECK_ARG(cpg <= 256 && 256 % cpg == 0, "groupnorm: channels/group %d must divide 256", cpg);
    int gpb = 256 / cpg / 16;  // 16 pixel rows per block
    if (gpb < 1) gpb = 1;
    while (G % gpb) --gpb;
    while (256 % (cpg * gpb)) --gpb;
    NPS_CHECK_ARG(gpb >= 1 && gpb <= 32, "groupnorm: cannot tile groups");
    dim3 grid(G / gpb, B);
    if (dt == NPS_DT_BF16)
        hipLaunchKernelGGL(groupnorm_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, gamma, beta, (bf16_t*)y, HW, C, G, eps, act, gpb);
    else
        hipLaunchKernelGGL(groupnorm_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, gamma, beta, (float*)y, HW, C, G, eps, act, gpb);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_layernorm(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                 const float* addend, int addend_rows, float* y2, int rows, int D, float eps,
                                 void* stream) {
    NPS_CHECK_ARG(x && gamma && beta && y && rows > 0 && D > 0 && D % 64 == 0 && D <= 1024, "layernorm: bad args (D=%d)", D);
    NPS_CHECK_ARG(!y2 || (addend && addend_rows > 0), "layernorm: y2 needs addend");
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, y, addend, addend_rows, y2,
                       (nps::bf16_t*)nullptr, (nps::bf16_t*)nullptr, rows, D, eps);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_layernorm_ex(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                    const float* addend, int addend_rows, float* y2, void* y_bf16, void* y2_bf16, int rows, int D,
                                    float eps, void* stream) {
    NPS_CHECK_ARG(x && gamma && beta && (y || y2 || y_bf16 || y2_bf16) && rows > 0 && D > 0 && D % 64 == 0 && D <= 1024, "layernorm_ex: bad args (D=%d)", D);
    NPS_CHECK_ARG(!(y2 || y2_bf16) || (addend && addend_rows > 0), "layernorm_ex: y2 needs addend");
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, y, addend, addend_rows, y2,
                       (nps::bf16_t*)y_bf16, (nps::bf16_t*)y2_bf16, rows, D, eps);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_add_rows(const float* a, const float* b, float* out, int rows, int D, int b_rows, void* stream) {
    NPS_CHECK_ARG(a && b && out && rows > 0 && D > 0 && b_rows > 0, "add_rows: bad args");
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for((long long)rows * D)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)rows * D, D, b_rows);
    NPS_LAUNCH_RET();
}

// The per-pair result row of the runner: [t(3) | q(4) | n1 | n2 | m | t_err | r_err | pair index | non-finite count | 0 | 0] as f32
// (nopesac_amd/runner.py METRIC_WIDTH = 16; the reference gathers the same numbers as python objects, mp3d_evaluation.py:316-319).
namespace nps {
__global__ void metric_rows_kernel(const float* __restrict__ t, const float* __restrict__ q, const int* __restrict__ n1, const int* __restrict__ n2,
                                   const int* __restrict__ m, const float* __restrict__ t_err, const float* __restrict__ r_err,
                                   const int* __restrict__ nonfinite, int pair_idx0, float* __restrict__ rows, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float* r = rows + (long long)b * 16;
    r[0] = t[3 * b]; r[1] = t[3 * b + 1]; r[2] = t[3 * b + 2];
    r[3] = q[4 * b]; r[4] = q[4 * b + 1]; r[5] = q[4 * b + 2]; r[6] = q[4 * b + 3];
    r[7] = (float)n1[b]; r[8] = (float)n2[b]; r[9] = (float)m[b];
    r[10] = t_err ? t_err[b] : 0.f; r[11] = r_err ? r_err[b] : 0.f;
    r[12] = (float)(pair_idx0 + b);
    r[13] = nonfinite ? (float)nonfinite[0] : 0.f;
    r[14] = 0.f; r[15] = 0.f;
}
}  // namespace nps

extern "C" int nopesac_metric_rows(const float* trans, const float* rot, const int32_t* n1, const int32_t* n2, const int32_t* m,
                                   const float* t_err, const float* r_err, const int32_t* nonfinite, int pair_idx0, float* rows, int B,
                                   void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(trans && rot && n1 && n2 && m && rows && B > 0, "metric_rows: bad args");
    hipLaunchKernelGGL(metric_rows_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, trans, rot, n1, n2, m, t_err, r_err, nonfinite,
                       pair_idx0, rows, B);
    NPS_LAUNCH_RET();
}

namespace nps {
__global__ void concat_cols_kernel(const float* __restrict__ a, int Da, const float* __restrict__ b, int Db, float* __restrict__ out, int rows) {
    const int D = Da + Db, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const int r = i / D, c = i % D;
    out[i] = c < Da ? a[r * Da + c] : b[r * Db + (c - Da)];
}
}  // namespace nps

extern "C" int nopesac_concat_cols(const float* a, int Da, const float* b, int Db, float* out, int rows, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(a && b && out && rows > 0 && Da > 0 && Db > 0 && (long long)rows * (Da + Db) < (1ll << 31), "concat_cols: bad args");
    hipLaunchKernelGGL(concat_cols_kernel, dim3((rows * (Da + Db) + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, Da, b, Db, out, rows);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_add_rows_bf16(const float* a, const float* b, void* a_bf16, void* ab_bf16, int rows, int D, int b_rows, void* stream) {
    NPS_CHECK_ARG(a && b && a_bf16 && ab_bf16 && rows > 0 && D > 0 && D % 4 == 0 && b_rows > 0, "add_rows_bf16: bad args (D %% 4 == 0)");
    NPS_CHECK_ARG((((uintptr_t)a | (uintptr_t)b) & 15) == 0 && (((uintptr_t)a_bf16 | (uintptr_t)ab_bf16) & 7) == 0, "add_rows_bf16: alignment");
    const long long total4 = (long long)rows * D / 4;
    hipLaunchKernelGGL(add_rows_bf16_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, a, b, (bf16_t*)a_bf16, (bf16_t*)ab_bf16, total4, D,
                       b_rows);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_softmax_rows_pad(const float* x, void* y, int rows, int D, int out_ld, int out_dt, void* stream) {
    NPS_CHECK_ARG(x && y && rows > 0 && D > 0 && D <= 1024 && out_ld >= D && out_ld <= 1024, "softmax_rows_pad: bad args");
    NPS_CHECK_ARG(out_dt == NPS_DT_F32 || out_dt == NPS_DT_BF16, "softmax_rows_pad: out_dt");
    if (out_dt == NPS_DT_BF16)
        hipLaunchKernelGGL(softmax_rows_pad_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, rows, D, out_ld);
    else
        hipLaunchKernelGGL(softmax_rows_pad_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, (float*)y, rows, D, out_ld);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_softmax_rows(const float* x, float* y, int rows, int D, void* stream) {
    NPS_CHECK_ARG(x && y && rows > 0 && D > 0 && D <= 1024, "softmax_rows: bad args");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, rows, D);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_transpose_hw_rows(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    NPS_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0, "transpose_hw_rows: bad args");
    hipLaunchKernelGGL(transpose_hw_rows_kernel, dim3(grid_for((long long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_normalize_rows(const float* x, float* y, int rows, int D, int canonical_sign, void* stream) {
    NPS_CHECK_ARG(x && y && rows > 0 && D > 0, "normalize_rows: bad args");
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, y, rows, D, canonical_sign);
    NPS_LAUNCH_RET();
}

// ---- finite check (the reference drops into pdb on NaN, camera_head.py:185-187,681-682,1072-1074; here: an error path)
namespace nps {
__global__ void count_nonfinite_kernel(const float* __restrict__ x, long long n, int* __restrict__ count) {
    int bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned u = __float_as_uint(x[i]);
        bad += ((u & 0x7f800000u) == 0x7f800000u) ? 1 : 0;               // exponent all ones: Inf or NaN
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}
struct NonfiniteBatch {
    const float* x[NOPESAC_NONFINITE_MAX_TENSORS];
    long long n[NOPESAC_NONFINITE_MAX_TENSORS];
};
// blockIdx.y = tensor: the camera list + planes of a forward pass (13 small tensors) in one launch instead of one each
__global__ void count_nonfinite_batch_kernel(const NonfiniteBatch b, int* __restrict__ count) {
    const float* __restrict__ x = b.x[blockIdx.y];
    const long long n = b.n[blockIdx.y];
    int bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned u = __float_as_uint(x[i]);
        bad += ((u & 0x7f800000u) == 0x7f800000u) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}
}  // namespace nps

extern "C" int nopesac_count_nonfinite_batch(const float* const* x, const int64_t* n, int n_tensors, int32_t* count, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && n && count && n_tensors > 0 && n_tensors <= NOPESAC_NONFINITE_MAX_TENSORS, "count_nonfinite_batch: bad args");
    NonfiniteBatch b;
    long long nmax = 0;
    for (int i = 0; i < NOPESAC_NONFINITE_MAX_TENSORS; ++i) {
        const int j = i < n_tensors ? i : 0;
        NPS_CHECK_ARG(x[j] && n[j] > 0, "count_nonfinite_batch: null / empty tensor");
        b.x[i] = x[j]; b.n[i] = n[j];
        if (n[j] > nmax) nmax = n[j];
    }
    const int blocks = (int)((nmax + 255) / 256 > 256 ? 256 : (nmax + 255) / 256);
    hipLaunchKernelGGL(count_nonfinite_batch_kernel, dim3(blocks, n_tensors), dim3(256), 0, (hipStream_t)stream, b, count);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_count_nonfinite(const float* x, int64_t n, int32_t* count, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && count && n > 0, "count_nonfinite: bad args");
    const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(count_nonfinite_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long long)n, count);
    NPS_LAUNCH_RET();
}


// ---- engine-clock probe: one wave spins for `spin_cycles` shader cycles and reports (shader cycles, 100 MHz reference ticks) of the
// interval: shader clock = cycles / ticks x 100 MHz.  Launched on a side stream WHILE a workload runs, it reads the clock the
// power-management firmware actually grants under that load (the MFMA-bound conv kernels run at ~1.9 GHz, not the 2.4 GHz the
// peak figures assume - scripts/power_probe.py); bench.py reports the roofline fraction at the measured clock next to the nominal one.
namespace nps {
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long spin_cycles) {
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long c1 = c0;
    while (c1 - c0 < spin_cycles) {
        __builtin_amdgcn_s_sleep(32);
        c1 = __builtin_readcyclecounter();
    }
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}
}  // namespace nps

extern "C" int nopesac_clock_probe(uint64_t* out2, int64_t spin_cycles, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(out2 && spin_cycles > 0 && spin_cycles <= 4000000000ll, "clock_probe: bad args");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out2, (unsigned long long)spin_cycles);
    NPS_LAUNCH_RET();
}


// ---- LDS canary (diagnostic, round 3): a workgroup fills `words` dwords of LDS with an address-derived pattern, spins, and checks it
// again - while other kernels run on other streams.  Every mismatch is counted; the first few are logged as {workgroup, dword index,
// expected, found}.  Found with it: see DESIGN.md section 6 (round 3, "a workgroup's LDS is not private ...").
namespace nps {
__global__ __launch_bounds__(256) void lds_canary_kernel(int words, unsigned long long spin_cycles, int rounds, unsigned* __restrict__ count,
                                                         unsigned* __restrict__ log4, int log_cap) {
    extern __shared__ unsigned canary_lds[];
    const unsigned salt = 0x9E3779B9u * (blockIdx.x + 1);
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) canary_lds[i] = (unsigned)i * 2654435761u ^ salt ^ (unsigned)r;
        __syncthreads();
        const unsigned long long c0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - c0 < spin_cycles) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        for (int i0 = threadIdx.x; i0 < words; i0 += blockDim.x) {
            const int i = (i0 + 67) % words;                 // (dwords another wave wrote)
            const unsigned want = (unsigned)i * 2654435761u ^ salt ^ (unsigned)r, got = canary_lds[i];
            if (got != want) {
                const unsigned k = atomicAdd(count, 1u);
                if ((int)k < log_cap) { log4[4 * k] = blockIdx.x; log4[4 * k + 1] = (unsigned)i; log4[4 * k + 2] = want; log4[4 * k + 3] = got; }
            }
        }
        __syncthreads();
    }
}
}  // namespace nps

extern "C" int nopesac_lds_canary(int workgroups, int lds_bytes, int64_t spin_cycles, int rounds, uint32_t* count, uint32_t* log4, int log_cap,
                                  void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(workgroups > 0 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && lds_bytes % 4 == 0 && spin_cycles >= 0 && rounds > 0 && count &&
                  (log_cap == 0 || log4), "lds_canary: bad args");
    NPS_ENSURE_LDS(160 * 1024, lds_canary_kernel);
    hipLaunchKernelGGL(lds_canary_kernel, dim3(workgroups), dim3(256), lds_bytes, (hipStream_t)stream, lds_bytes / 4, (unsigned long long)spin_cycles, rounds,
                       count, log4, log_cap);
    NPS_LAUNCH_RET();
}

// ---- uint8 image planes -> f32 (exact): lets the boundary take the decoder's 8-bit images over PCIe (a quarter of the bytes of the
// reference mapper's float32 tensors, data/planercnn_transforms.py:225-227) and widen them on the device
namespace nps {
__global__ void u8_to_f32_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, long long n16) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
        const uint4 q = reinterpret_cast<const uint4*>(x)[i];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        float4* o = reinterpret_cast<float4*>(y) + 4 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = make_float4((float)(w[j] & 0xff), (float)((w[j] >> 8) & 0xff), (float)((w[j] >> 16) & 0xff), (float)(w[j] >> 24));
    }
}
__global__ void u8_to_f32_tail_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, long long n0, long long n) {
    const long long i = n0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (float)x[i];
}
}  // namespace nps

extern "C" int nopesac_u8_to_f32(const uint8_t* x, float* y, int64_t n, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && y && n > 0, "u8_to_f32: bad args");
    NPS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "u8_to_f32: pointers must be 16-byte aligned");
    const long long n16 = n / 16;
    if (n16) hipLaunchKernelGGL(u8_to_f32_kernel, dim3(grid_for(n16)), dim3(256), 0, (hipStream_t)stream, x, y, n16);
    if (n % 16) hipLaunchKernelGGL(u8_to_f32_tail_kernel, dim3(1), dim3(16), 0, (hipStream_t)stream, x, y, n16 * 16, (long long)n);
    NPS_LAUNCH_RET();
}

// ---- result fetch: up to NOPESAC_GATHER_MAX_SEGMENTS byte ranges -> ONE destination buffer in ONE launch.  The destination is normally
// PINNED HOST memory (device-mapped: the stores travel over PCIe): the small result tensors of a batch reach the host without the
// torch.cat + pad fills + copy launches the fetch used to issue (12 launches per call), and - being a kernel - the fetch becomes part
// of a captured graph / launch tape.  blockIdx.y = segment; 16-byte vectors when source and destination offsets allow it.
namespace nps {
struct GatherSegs {
    const unsigned char* src[NOPESAC_GATHER_MAX_SEGMENTS];
    long long size[NOPESAC_GATHER_MAX_SEGMENTS];
    long long dst_off[NOPESAC_GATHER_MAX_SEGMENTS];
    const long long* size_dev[NOPESAC_GATHER_MAX_SEGMENTS];      // optional: valid bytes of the segment, known on the device only
};
__global__ void gather_bytes_kernel(const GatherSegs g, unsigned char* __restrict__ dst) {
    const unsigned char* __restrict__ s = g.src[blockIdx.y];
    unsigned char* __restrict__ d = dst + g.dst_off[blockIdx.y];
    long long n = g.size[blockIdx.y];
    if (g.size_dev[blockIdx.y]) n = max(0ll, min(n, *g.size_dev[blockIdx.y]));
    const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0) {
        const long long n16 = n >> 4;
        for (long long i = t0; i < n16; i += step) reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
        for (long long i = (n16 << 4) + t0; i < n; i += step) d[i] = s[i];
    } else {
        for (long long i = t0; i < n; i += step) d[i] = s[i];
    }
}
}  // namespace nps

extern "C" int nopesac_gather_bytes(const void* const* src, const int64_t* size, const int64_t* const* size_dev, const int64_t* dst_off,
                                    int n_segments, void* dst, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(src && size && dst_off && dst && n_segments > 0 && n_segments <= NOPESAC_GATHER_MAX_SEGMENTS, "gather_bytes: bad args");
    GatherSegs g;
    long long nmax = 0;
    for (int i = 0; i < NOPESAC_GATHER_MAX_SEGMENTS; ++i) {
        const int j = i < n_segments ? i : 0;
        NPS_CHECK_ARG(src[j] && size[j] > 0 && dst_off[j] >= 0, "gather_bytes: null / empty segment %d", j);
        g.src[i] = (const unsigned char*)src[j]; g.size[i] = size[j]; g.dst_off[i] = dst_off[j];
        g.size_dev[i] = size_dev ? (const long long*)size_dev[j] : nullptr;
        if (size[j] > nmax) nmax = size[j];
    }
    const long long want = (nmax + 256 * 16 - 1) / (256 * 16);
    const int blocks = (int)(want > 64 ? 64 : want);
    hipLaunchKernelGGL(gather_bytes_kernel, dim3(blocks, n_segments), dim3(256), 0, (hipStream_t)stream, g, (unsigned char*)dst);
    NPS_LAUNCH_RET();
}
