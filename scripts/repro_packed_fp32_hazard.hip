// Stand-alone reproducer attempt for the round-3 finding "a wave executing packed-f32 VALU instructions gets lanes 48-63 of the results
// corrupted while a wave of ANOTHER kernel issues MFMAs on the same SIMD" (profiles/r3_packed_fp32_hazard.txt).  The finding came from this
// repository's own kernels (victim: ransac_score_maps_kernel built with the SLP vectoriser's v_pk_*_f32; aggressors: its MFMA conv kernels);
// the round-3 verdict asked for a reproducer that does not depend on them.  This file has no dependency but the HIP runtime:
//
//   victim<PACKED>   every thread runs a chain of fused multiply-adds on two floats - as ONE v_pk_fma_f32 per step (PACKED = 1, inline asm)
//                    or as two v_fma_f32 (PACKED = 0) - and stores the pair.  Deterministic: every launch must give the same bytes.
//   aggressor<MFMA>  a long loop of v_mfma_f32_32x32x16_bf16 (MFMA = 1) or of plain v_fma_f32 (MFMA = 0) with few registers, so that victim
//                    waves are co-resident with it on the SIMDs.
//
// The victim runs alone (reference), then repeatedly on stream A while the aggressor runs on stream B; mismatching 32-bit words are counted
// per lane.  build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 scripts/repro_packed_fp32_hazard.hip -o /tmp/repro && /tmp/repro
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PACKED>
__global__ __launch_bounds__(256) void victim(f32x2* __restrict__ out, int steps) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 acc = {1.0f + 1e-3f * (gid & 1023), 2.0f - 1e-3f * (gid & 511)};
    const f32x2 a = {1.0000001f, 0.9999999f}, b = {1e-4f, -1e-4f};
    for (int i = 0; i < steps; ++i) {
        if (PACKED) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
        } else {
            float x = acc.x, y = acc.y;
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a.x), "v"(b.x));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(a.y), "v"(b.y));
            acc.x = x; acc.y = y;
        }
    }
    out[gid] = acc;
}

// a denser victim: eight independent pairs, v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 back to back (PACKED = 1) or the same arithmetic
// as scalar instructions (PACKED = 0; mul + add are not contracted: the results of the two builds differ, each is compared with itself)
template <int PACKED>
__global__ __launch_bounds__(256) void victim8(f32x2* __restrict__ out, int steps) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x2{1.0f + 1e-3f * ((gid + 37 * k) & 1023), 2.0f - 1e-3f * ((gid + 11 * k) & 511)};
    const f32x2 a = {1.0000001f, 0.9999999f}, b = {1e-4f, -1e-4f};
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (PACKED) {
                asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(a), "v"(b));
            } else {
                float x = acc[k].x, y = acc[k].y;
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a.x), "v"(b.x));
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(a.y), "v"(b.y));
                acc[k].x = x; acc[k].y = y;
            }
        }
    }
    f32x2 r = acc[0];
    for (int k = 1; k < 8; ++k) { r.x += acc[k].x; r.y += acc[k].y; }
    out[gid] = r;
}

// the operand forms the original victim (ransac_score_maps_kernel built with packed-f32 enabled) contains: SGPR-pair sources, op_sel /
// op_sel_hi selections, neg modifiers
__global__ __launch_bounds__(256) void victim_forms(f32x2* __restrict__ out, int steps, f32x2 ua, f32x2 ub) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = f32x2{1.0f + 1e-3f * ((gid + 37 * k) & 1023), 2.0f - 1e-3f * ((gid + 11 * k) & 511)};
    const f32x2 a = {1.0000001f, 0.9999999f}, b = {1e-4f, -1e-4f};
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "s"(ua), "v"(b));                       // V, V, S, V
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc[k]) : "s"(ub));                                     // V, V, S
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(acc[k]) : "v"(a));
            asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(acc[k]) : "v"(b));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(acc[k]) : "v"(a), "v"(b));
            asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(acc[k]) : "s"(ua));
        }
    }
    f32x2 r = acc[0];
    for (int k = 1; k < 4; ++k) { r.x += acc[k].x; r.y += acc[k].y; }
    out[gid] = r;
}

// compiler-generated packed arithmetic fed by TRANSCENDENTAL results (v_rsq_f32 / v_exp_f32 / v_rcp_f32 - what the original victim's
// quaternion and score math has and the kernels above lack): plain C on float2 vectors, no inline asm, so the compiler's own hazard
// handling is what runs - exactly as in the library
__global__ __launch_bounds__(256) void victim_trans(f32x2* __restrict__ out, int steps) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    f32x2 acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = f32x2{0.3f + 1e-3f * ((gid + 37 * k) & 1023), 0.7f - 1e-3f * ((gid + 11 * k) & 511)};
    const f32x2 b = {1e-3f, -1e-3f}, c = {0.25f, 0.5f};
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = acc[k].x * acc[k].x + acc[k].y * acc[k].y + 1.0f;
            const float r = __builtin_amdgcn_rsqf(t);                 // v_rsq_f32
            acc[k] = acc[k] * f32x2{r, r} + b;                        // v_pk_mul_f32 / v_pk_add_f32 (or v_pk_fma_f32) on the fresh result
            const float e = __builtin_amdgcn_exp2f(-acc[k].x);        // v_exp_f32
            const float q = __builtin_amdgcn_rcpf(1.0f + e);          // v_rcp_f32
            acc[k] = acc[k] * c + f32x2{q, e};
        }
    }
    f32x2 r = acc[0];
    for (int k = 1; k < 4; ++k) r = r + acc[k];
    out[gid] = r;
}

// an aggressor shaped like the library's MFMA kernels: four accumulator tiles in AGPRs (64 registers), LDS traffic between the MFMAs
__global__ __launch_bounds__(256) void aggressor_agpr(float* __restrict__ sink, int iters) {
    __shared__ float lds[256 * 4];
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x & 7)); b[e] = (__bf16)(0.002f * (threadIdx.x & 3)); }
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
        s += lds[(threadIdx.x + i) & 1023 & 255];
    }
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    if (s == 12345.678f) sink[0] = s;
}

template <int MFMA>
__global__ __launch_bounds__(256) void aggressor(float* __restrict__ sink, int iters) {
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x & 7)); b[e] = (__bf16)(0.002f * (threadIdx.x & 3)); }
    float s = 0.5f;
    for (int i = 0; i < iters; ++i) {
        if (MFMA) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        } else {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s) : "v"(1.0000001f), "v"(1e-6f));
        }
    }
    float t = s;
    for (int e = 0; e < 16; ++e) t += acc[e];
    if (t == 12345.678f) sink[0] = t;             // keeps the loops alive
}

template <typename V, typename A>
static void experiment(const char* label, int victim_blocks, int launches, V launch_victim, A launch_aggressor) {
    const size_t n = (size_t)victim_blocks * 256;
    f32x2* d_out;
    float* d_sink;
    CK(hipMalloc(&d_out, n * sizeof(f32x2)));
    CK(hipMalloc(&d_sink, 64));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    std::vector<f32x2> ref(n), got(n);
    launch_victim(sa, d_out);
    CK(hipStreamSynchronize(sa));
    CK(hipMemcpy(ref.data(), d_out, n * sizeof(f32x2), hipMemcpyDeviceToHost));
    long long bad_words = 0, bad_launches = 0, lane_hist[64];
    memset(lane_hist, 0, sizeof(lane_hist));
    const bool with_aggressor = launch_aggressor(sb, d_sink);
    for (int l = 0; l < launches; ++l) {
        CK(hipMemsetAsync(d_out, 0, n * sizeof(f32x2), sa));
        launch_victim(sa, d_out);
        CK(hipMemcpyAsync(got.data(), d_out, n * sizeof(f32x2), hipMemcpyDeviceToHost, sa));
        CK(hipStreamSynchronize(sa));
        long long bad = 0;
        for (size_t i = 0; i < n; ++i) {
            if (memcmp(&ref[i], &got[i], sizeof(f32x2)) != 0) {
                ++bad;
                ++lane_hist[i & 63];
            }
        }
        bad_words += bad;
        bad_launches += bad != 0;
    }
    const bool still_running = with_aggressor && hipStreamQuery(sb) == hipErrorNotReady;
    CK(hipDeviceSynchronize());
    long long hi = 0;
    for (int l = 48; l < 64; ++l) hi += lane_hist[l];
    printf("%-78s launches off %3lld of %d, mismatching results %8lld (lanes 48-63: %lld)%s\n", label, bad_launches, launches, bad_words, hi,
           with_aggressor && !still_running ? "  [aggressor finished early]" : "");
    CK(hipFree(d_out));
    CK(hipFree(d_sink));
    CK(hipStreamDestroy(sa));
    CK(hipStreamDestroy(sb));
}

#ifdef REPRO_SHARED
// the same kernels as a shared library (hipcc -shared -fPIC -DREPRO_SHARED): scripts/repro_packed_fp32_mix.py pairs them with the library's
// own victim / aggressors to see which SIDE of the original observation carries the effect
extern "C" void repro_victim_packed(void* out, int blocks, int steps, void* stream) {
    hipLaunchKernelGGL(victim8<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (f32x2*)out, steps);
}
extern "C" void repro_victim_forms(void* out, int blocks, int steps, void* stream) {
    hipLaunchKernelGGL(victim_forms, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (f32x2*)out, steps, f32x2{1.0000001f, 0.9999999f}, f32x2{1e-5f, -1e-5f});
}
extern "C" void repro_victim_trans(void* out, int blocks, int steps, void* stream) {
    hipLaunchKernelGGL(victim_trans, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (f32x2*)out, steps);
}
extern "C" void repro_aggressor_mfma(void* sink, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)sink, iters);
}
extern "C" void repro_aggressor_agpr(void* sink, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(aggressor_agpr, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)sink, iters);
}
#else
int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    const int cus = prop.multiProcessorCount, steps = 4000, launches = 200;
    auto none = [](hipStream_t, float*) { return false; };
    auto mfma = [&](int blocks, int iters) { return [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 0, s, sink, iters); return true; }; };
    auto valu = [&](int blocks, int iters) { return [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor<0>, dim3(blocks), dim3(256), 0, s, sink, iters); return true; }; };
    auto agpr = [&](int blocks, int iters) { return [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor_agpr, dim3(blocks), dim3(256), 0, s, sink, iters); return true; }; };
    auto v1p = [&](int blocks) { return [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim<1>, dim3(blocks), dim3(256), 0, s, o, steps); }; };
    auto v1s = [&](int blocks) { return [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim<0>, dim3(blocks), dim3(256), 0, s, o, steps); }; };
    auto v8p = [&](int blocks) { return [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim8<1>, dim3(blocks), dim3(256), 0, s, o, steps / 4); }; };
    auto v8s = [&](int blocks) { return [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim8<0>, dim3(blocks), dim3(256), 0, s, o, steps / 4); }; };
    experiment("packed victim (v_pk_fma_f32 chain), no aggressor", 2 * cus, launches, v1p(2 * cus), none);
    experiment("packed victim next to an MFMA aggressor (2 blocks per CU)", 2 * cus, launches, v1p(2 * cus), mfma(2 * cus, 6000000));
    experiment("packed victim next to a VALU-only aggressor", 2 * cus, launches, v1p(2 * cus), valu(2 * cus, 24000000));
    experiment("scalar victim next to an MFMA aggressor", 2 * cus, launches, v1s(2 * cus), mfma(2 * cus, 6000000));
    experiment("packed victim, 8 blocks per CU, next to an MFMA aggressor", 8 * cus, launches, v1p(8 * cus), mfma(2 * cus, 12000000));
    experiment("dense packed victim (8 x pk_mul / pk_add / pk_fma), no aggressor", 4 * cus, launches, v8p(4 * cus), none);
    experiment("dense packed victim next to an MFMA aggressor", 4 * cus, launches, v8p(4 * cus), mfma(2 * cus, 12000000));
    experiment("dense packed victim next to an AGPR + LDS MFMA aggressor (1 block per CU)", 4 * cus, launches, v8p(4 * cus), agpr(cus, 3000000));
    experiment("dense packed victim next to an AGPR + LDS MFMA aggressor (3 blocks per CU)", 4 * cus, launches, v8p(4 * cus), agpr(3 * cus, 3000000));
    auto vf = [&](int blocks) { return [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim_forms, dim3(blocks), dim3(256), 0, s, o, steps / 4, f32x2{1.0000001f, 0.9999999f}, f32x2{1e-5f, -1e-5f}); }; };
    experiment("operand-form victim (SGPR sources, op_sel, neg), no aggressor", 4 * cus, launches, vf(4 * cus), none);
    experiment("operand-form victim next to an MFMA aggressor", 4 * cus, launches, vf(4 * cus), mfma(2 * cus, 12000000));
    experiment("operand-form victim next to an AGPR + LDS MFMA aggressor (3 blocks per CU)", 4 * cus, launches, vf(4 * cus), agpr(3 * cus, 3000000));
    auto vt = [&](int blocks) { return [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim_trans, dim3(blocks), dim3(256), 0, s, o, steps / 4); }; };
    experiment("transcendental -> packed victim, no aggressor", 4 * cus, launches, vt(4 * cus), none);
    experiment("transcendental -> packed victim next to an MFMA aggressor", 4 * cus, launches, vt(4 * cus), mfma(2 * cus, 12000000));
    experiment("transcendental -> packed victim next to an AGPR + LDS MFMA aggressor (3 blocks per CU)", 4 * cus, launches, vt(4 * cus), agpr(3 * cus, 3000000));
    experiment("dense scalar victim next to an AGPR + LDS MFMA aggressor (3 blocks per CU)", 4 * cus, launches, v8s(4 * cus), agpr(3 * cus, 3000000));
    return 0;
}
#endif
