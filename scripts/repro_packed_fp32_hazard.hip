// Stand-alone reproducer for the round-3 finding "results of a kernel that executes packed-f32 VALU instructions (v_pk_*_f32) are corrupted
// while certain MFMA kernels run next to it" (profiles/r3_packed_fp32_hazard.txt, profiles/r4_packed_fp32_recheck.txt).  No dependency but
// the HIP runtime.  What reproduces (MI355X, ROCm 7.2):
//
//     victim_scoremaps   the geometry of this repository's score-map kernel (quaternion -> rotation, plane warps, normalisations, exp),
//                        plain C; with the default target features the SLP vectoriser turns it into ~66 v_pk_*_f32 instructions
//     aggressor_c64like  an MFMA loop shaped like the library's 3x3 conv: accumulators in VGPRs, one 16-byte LDS fragment read per MFMA,
//                        a 16-slot register ring of weight fragments refilled from global memory, 50 KB of LDS, a barrier per round
//
//   victim next to that aggressor: ~100 of 400 launches differ from the idle-GPU result;  the SAME source built with -fno-slp-vectorize
//   (no packed-f32 instruction): never.  Bisected here as well: the aggressor still disturbs without its LDS reads OR without its global
//   refill, but not without both; nothing but MFMAs (VGPR or AGPR destinations), MFMA into AGPRs + LDS reads, VALU-only loops with 128 /
//   168 VGPRs per wave and wave-launch storms never do; hand-written v_pk_* chains (SGPR-pair sources, op_sel / neg modifiers,
//   transcendental-fed) are never disturbed; the victim's tables may live in global memory instead of LDS.  So: compiler-generated
//   packed-f32 code + a neighbour that issues MFMAs into VGPR tiles while loads return into its VGPRs.
//
// build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/repro_packed_fp32_hazard.hip -o /tmp/repro && /tmp/repro
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize scripts/repro_packed_fp32_hazard.hip -o /tmp/repro2 && /tmp/repro2
// Every victim is compared with its own idle-GPU run; mismatching results are counted.
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

// ---- the geometry of the library's score-map kernel (this repository's own csrc/ransac.hip + common.h, reduced to what the kernel needs):
// (K + 1) pose hypotheses x K matched planes, plane warp under each hypothesis, normal / parameter distances, exp(-d).  Built with the
// default target features the SLP vectoriser packs its scalar f32 math into ~140 v_pk_*_f32; with -fno-slp-vectorize into none.
__device__ __forceinline__ void g_quat_to_rot(const float q[4], float R[9]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * y * y - 2 * z * z; R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
    R[3] = 2 * x * y + 2 * w * z; R[4] = 1 - 2 * x * x - 2 * z * z; R[5] = 2 * y * z - 2 * w * x;
    R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = 1 - 2 * x * x - 2 * y * y;
}
__device__ __forceinline__ void g_warp_plane(const float p[3], const float R[9], const float t[3], float out[3]) {
    const float f0 = p[0], f1 = -p[1], f2 = -p[2];
    float e[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float s = __fmul_rn(R[3 * i], f0);
        s = __fmaf_rn(R[3 * i + 1], f1, s);
        s = __fmaf_rn(R[3 * i + 2], f2, s);
        e[i] = __fadd_rn(s, t[i]);
        b[i] = __fsub_rn(e[i], t[i]);
    }
    const float dot = e[0] * b[0] + e[1] * b[1] + e[2] * b[2];
    const float nb = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]) + 1e-5f;
    const float c = dot / (nb * nb);
    out[0] = c * b[0]; out[1] = c * b[1]; out[2] = c * b[2];
}
__device__ __forceinline__ float g_norm3(const float v[3]) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
__device__ __forceinline__ void g_normalize3(const float v[3], float o[3]) {
    const float n = fmaxf(g_norm3(v), 1e-12f);
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}
// USE_LDS = 0: the rotation / translation / geometry tables live in global scratch memory instead of LDS (is the victim's LDS use needed?)
template <int USE_LDS>
__global__ __launch_bounds__(256) void victim_scoremaps(const float* __restrict__ geo, const float* __restrict__ rot_raw,
                                                        const float* __restrict__ trans_raw, int nq, f32x2* __restrict__ out,
                                                        float* __restrict__ scratch) {
    const int b = blockIdx.x, tid = threadIdx.x, NH = nq + 1;
    __shared__ float lR[USE_LDS ? 129 * 9 : 1], lT[USE_LDS ? 129 * 3 : 1], lG[USE_LDS ? 128 * 6 : 1];
    float* sR = USE_LDS ? lR : scratch + (long long)b * 2400;
    float* sT = USE_LDS ? lT : sR + 129 * 9;
    float* sG = USE_LDS ? lG : sT + 129 * 3;
    for (int h = tid; h < NH; h += 256) {
        const float* rr = rot_raw + ((long long)b * NH + h) * 4;
        const float nn = fmaxf(sqrtf(rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] + rr[3] * rr[3]), 1e-12f);
        float q[4];
        for (int d = 0; d < 4; ++d) q[d] = rr[d] / nn;
        g_quat_to_rot(q, sR + 9 * h);
        for (int d = 0; d < 3; ++d) sT[3 * h + d] = trans_raw[((long long)b * NH + h) * 3 + d];
    }
    for (int e = tid; e < nq * 6; e += 256) sG[e] = geo[(long long)b * nq * 6 + e];
    if (!USE_LDS) __threadfence_block();
    __syncthreads();
    for (int e = tid; e < NH * nq; e += 256) {
        const int h = e / nq, j = e % nq;
        const float* gl = sG + 6 * j;
        const float* Rm = sR + 9 * h;
        const float* t = sT + 3 * h;
        const float p0[3] = {gl[0], gl[1], gl[2]}, p1[3] = {gl[3], -gl[4], -gl[5]}, z[3] = {0.f, 0.f, 0.f};
        float w_r[3], w_rt[3], n0[3], n1v[3];
        g_warp_plane(p0, Rm, z, w_r);
        g_warp_plane(p0, Rm, t, w_rt);
        g_normalize3(w_r, n0);
        g_normalize3(p1, n1v);
        const float d0 = n0[0] - n1v[0], d1 = n0[1] - n1v[1], d2 = n0[2] - n1v[2];
        const float dn = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        const float e0 = w_rt[0] - p1[0], e1 = w_rt[1] - p1[1], e2 = w_rt[2] - p1[2];
        const float dl2 = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        out[(long long)b * NH * nq + e] = f32x2{expf(-dn), expf(-dl2)};
    }
}

// pieces of the geometry victim on their own (plain C, compiler-packed): PART 0 = the two plane warps only, 1 = the normalisations +
// distances only (IEEE divisions and square roots), 2 = rotation-matrix build only
template <int PART>
__global__ __launch_bounds__(256) void victim_part(const float* __restrict__ geo, const float* __restrict__ rot_raw,
                                                   const float* __restrict__ trans_raw, int nq, f32x2* __restrict__ out) {
    const int b = blockIdx.x, tid = threadIdx.x, NH = nq + 1;
    for (int e = tid; e < NH * nq; e += 256) {
        const int h = e / nq, j = e % nq;
        const float* gl = geo + ((long long)b * nq + j) * 6;
        const float* rr = rot_raw + ((long long)b * NH + h) * 4;
        const float* t = trans_raw + ((long long)b * NH + h) * 3;
        float x = 0.f, y = 0.f;
        if (PART == 0) {
            float Rm[9] = {rr[0], rr[1], rr[2], rr[3], rr[0] + rr[1], rr[1] - rr[2], rr[2] * 0.5f, rr[3] * 0.5f, rr[0] - rr[3]};
            const float p0[3] = {gl[0], gl[1], gl[2]}, z[3] = {0.f, 0.f, 0.f}, tt[3] = {t[0], t[1], t[2]};
            float w_r[3], w_rt[3];
            g_warp_plane(p0, Rm, z, w_r);
            g_warp_plane(p0, Rm, tt, w_rt);
            x = w_r[0] + w_r[1] + w_r[2]; y = w_rt[0] + w_rt[1] + w_rt[2];
        } else if (PART == 1) {
            const float a[3] = {gl[0] + rr[0], gl[1] + rr[1], gl[2] + rr[2]}, c[3] = {gl[3], -gl[4], -gl[5]}, d[3] = {t[0] + gl[0], t[1] + gl[1], t[2] + gl[2]};
            float n0[3], n1[3], n2[3];
            g_normalize3(a, n0); g_normalize3(c, n1); g_normalize3(d, n2);
            const float d0 = n0[0] - n1[0], d1 = n0[1] - n1[1], d2 = n0[2] - n1[2];
            const float e0 = n2[0] - c[0], e1 = n2[1] - c[1], e2 = n2[2] - c[2];
            x = sqrtf(d0 * d0 + d1 * d1 + d2 * d2); y = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
        } else {
            const float nn = fmaxf(sqrtf(rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] + rr[3] * rr[3]), 1e-12f);
            float q[4], Rm[9];
            for (int d = 0; d < 4; ++d) q[d] = rr[d] / nn;
            g_quat_to_rot(q, Rm);
            x = Rm[0] + Rm[1] + Rm[2] + Rm[3] + Rm[4]; y = Rm[5] + Rm[6] + Rm[7] + Rm[8];
        }
        out[(long long)b * NH * nq + e] = f32x2{x, y};
    }
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

// hand-written packed chains WITH loads returning into the victim's own registers while they run (the geometry victim reads its tables
// in the loop; the chain victims above touch no memory): SRC = 0 global memory, 1 LDS
template <int SRC>
__global__ __launch_bounds__(256) void victim8_loads(const f32x2* __restrict__ tab, f32x2* __restrict__ out, int steps) {
    __shared__ f32x2 ltab[1024];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = threadIdx.x; i < 1024; i += 256) ltab[i] = tab[i];
    __syncthreads();
    f32x2 acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = f32x2{1.0f + 1e-3f * ((gid + 37 * k) & 1023), 2.0f - 1e-3f * ((gid + 11 * k) & 511)};
    const f32x2 b = {1e-4f, -1e-4f};
    for (int i = 0; i < steps; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2 a = SRC ? ltab[(gid * 7 + i * 13 + k * 101) & 1023] : tab[(gid * 7 + i * 13 + k * 101) & 1023];
            asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(a), "v"(b));
        }
    }
    f32x2 r = acc[0];
    for (int k = 1; k < 4; ++k) { r.x += acc[k].x; r.y += acc[k].y; }
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

// the minimal pair: nothing but MFMAs, four accumulator tiles, destination registers forced into VGPRs (ACC_VGPR = 1) or AGPRs (0)
template <int ACC_VGPR>
__global__ __launch_bounds__(256, 3) void aggressor_pure(float* __restrict__ sink, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x & 7)); b[e] = (__bf16)(0.002f * (threadIdx.x & 3)); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (ACC_VGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) s += acc[t][e];
    if (s == 12345.678f) sink[0] = s;
}

// wave-launch storm: very many very short workgroups (WORK = 0: a few VALU operations; WORK = 1: a few MFMAs) - is it the neighbour's
// instructions or the LAUNCHING of its waves (register initialisation by the dispatcher) that disturbs the victim?
template <int WORK>
__global__ __launch_bounds__(256) void aggressor_storm(float* __restrict__ sink, int reps) {
    float s = threadIdx.x * 0.5f;
    if (WORK == 0) {
        for (int i = 0; i < reps; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s) : "v"(1.0000001f), "v"(1e-6f));
    } else {
        f32x16 acc;
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x & 7)); b[e] = (__bf16)(0.002f * (threadIdx.x & 3)); }
        for (int i = 0; i < reps; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        for (int e = 0; e < 16; ++e) s += acc[e];
    }
    if (s == 12345.678f) sink[0] = s;
}

// no MFMA, no LDS: a VALU spin loop whose only special property is its REGISTER ALLOCATION (v127 / v167 marked as used: 128 / 168 VGPRs
// per wave, like the library's conv / tail kernels) - a co-resident victim wave then gets its registers from a different part of the file
template <int NREG>
__global__ __launch_bounds__(256) void aggressor_bigreg(float* __restrict__ sink, int iters) {
    float s = threadIdx.x * 0.5f;
    for (int i = 0; i < iters; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s) : "v"(1.0000001f), "v"(1e-6f));
    if (NREG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (NREG == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    if (s == 12345.678f) sink[0] = s;
}

// closer to the library's conv3x3_c64_kernel: accumulators in VGPRs, one 16-byte LDS fragment read per MFMA, a 16-slot register ring of
// weight fragments re-filled from global memory inside the loop, 50 KB of LDS (three workgroups per CU), a barrier per round
// USE_LDS = 0: the MFMA's second operand is a register constant (no LDS reads);  REFILL = 0: the weight ring is loaded once
template <int USE_LDS, int REFILL>
__global__ __launch_bounds__(256, 3) void aggressor_c64like(const bf16x8* __restrict__ wfrag, float* __restrict__ sink, int rounds) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[50688];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 50688 / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    __syncthreads();
    bf16x8 ring[16];
    const bf16x8* wp = wfrag + lane;
    for (int s2 = 0; s2 < 16; ++s2) ring[s2] = wp[s2 * 64];
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                bf16x8 af = ring[(ks + t) % 16];
                if (USE_LDS) af = *reinterpret_cast<const bf16x8*>(lds + ((t * 2816 + (lane & 31) * 144 + (lane >> 5) * 16 + ks * 32) % 50000 & ~15));
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[ks % 16], af, acc[t], 0, 0, 0);
            }
            if (REFILL && ks + 16 < 36) ring[ks % 16] = wp[(ks + 16) * 64];
        }
        __syncthreads();
        if (REFILL) {
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) ring[s2] = wp[s2 * 64];
        }
    }
    float s2 = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int e = 0; e < 16; ++e) s2 += acc[t][e];
    if (s2 == 12345.678f) sink[0] = s2;
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
    CK(hipMemsetAsync(d_out, 0, n * sizeof(f32x2), sa));     // (a victim may write fewer than n results)
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
extern "C" void repro_aggressor_c64like(void* wfrag, void* sink, int blocks, int rounds, void* stream) {
    hipLaunchKernelGGL((aggressor_c64like<1, 1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)wfrag, (float*)sink, rounds);
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
    const int cus = prop.multiProcessorCount, steps = 4000, launches = 400;
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
    bf16x8* d_w;
    CK(hipMalloc(&d_w, 36 * 64 * sizeof(bf16x8)));
    CK(hipMemset(d_w, 0x3c, 36 * 64 * sizeof(bf16x8)));
    auto c64 = [&](int blocks, int rounds) { return [=](hipStream_t s, float* sink) { hipLaunchKernelGGL((aggressor_c64like<1, 1>), dim3(blocks), dim3(256), 0, s, d_w, sink, rounds); return true; }; };
    auto c64_nolds = [&](int blocks, int rounds) { return [=](hipStream_t s, float* sink) { hipLaunchKernelGGL((aggressor_c64like<0, 1>), dim3(blocks), dim3(256), 0, s, d_w, sink, rounds); return true; }; };
    auto c64_norefill = [&](int blocks, int rounds) { return [=](hipStream_t s, float* sink) { hipLaunchKernelGGL((aggressor_c64like<1, 0>), dim3(blocks), dim3(256), 0, s, d_w, sink, rounds); return true; }; };
    {
        std::vector<f32x2> ht(1024);
        for (int i = 0; i < 1024; ++i) ht[i] = f32x2{1.0f + 1e-7f * (i % 7), 1.0f - 1e-7f * (i % 5)};
        f32x2* dtab;
        CK(hipMalloc(&dtab, 1024 * sizeof(f32x2)));
        CK(hipMemcpy(dtab, ht.data(), 1024 * sizeof(f32x2), hipMemcpyHostToDevice));
        auto vlg = [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim8_loads<0>, dim3(4 * cus), dim3(256), 0, s, dtab, o, steps / 4); };
        auto vll = [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim8_loads<1>, dim3(4 * cus), dim3(256), 0, s, dtab, o, steps / 4); };
        experiment("packed chains fed by GLOBAL loads in the loop, next to the c64-like aggressor", 4 * cus, launches, vlg, c64(20 * cus, 1500));
        experiment("packed chains fed by LDS loads in the loop, next to the c64-like aggressor", 4 * cus, launches, vll, c64(20 * cus, 1500));
    }
    experiment("packed chain victim next to the c64-like aggressor", 2 * cus, launches, v1p(2 * cus), c64(20 * cus, 60));
    experiment("dense packed victim next to the c64-like aggressor", 4 * cus, launches, v8p(4 * cus), c64(20 * cus, 60));
    experiment("operand-form victim next to the c64-like aggressor", 4 * cus, launches, vf(4 * cus), c64(20 * cus, 60));
    experiment("transcendental -> packed victim next to the c64-like aggressor", 4 * cus, launches, vt(4 * cus), c64(20 * cus, 60));
    experiment("dense SCALAR victim next to the c64-like aggressor", 4 * cus, launches, v8s(4 * cus), c64(20 * cus, 60));
    {   // the score-map geometry: 32 pairs x 51 hypotheses x 50 planes, constant pseudo-random inputs
        const int B = 32, nq = 50, NH = 51;
        std::vector<float> hg(B * nq * 6), hr(B * NH * 4), ht(B * NH * 3);
        unsigned x = 12345u;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return ((x >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& v : hg) v = rnd();
        for (auto& v : hr) v = rnd();
        for (auto& v : ht) v = rnd();
        float *dg, *dr, *dt;
        CK(hipMalloc(&dg, hg.size() * 4)); CK(hipMalloc(&dr, hr.size() * 4)); CK(hipMalloc(&dt, ht.size() * 4));
        CK(hipMemcpy(dg, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
        const int out_blocks = (B * NH * nq + 255) / 256;         // experiment() sizes the output as blocks x 256 pairs
        float* dscr;
        CK(hipMalloc(&dscr, (size_t)B * 2400 * 4));
        auto vs = [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim_scoremaps<1>, dim3(B), dim3(256), 0, s, dg, dr, dt, nq, o, dscr); };
        auto vsg = [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim_scoremaps<0>, dim3(B), dim3(256), 0, s, dg, dr, dt, nq, o, dscr); };
        experiment("score-map geometry victim (as compiled: see the v_pk count), no aggressor", out_blocks, launches, vs, none);
        experiment("score-map geometry victim next to an MFMA loop", out_blocks, launches, vs, mfma(2 * cus, 12000000));
        experiment("score-map geometry victim next to the c64-like aggressor", out_blocks, launches, vs, c64(20 * cus, 1500));
        auto vp0 = [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim_part<0>, dim3(B), dim3(256), 0, s, dg, dr, dt, nq, o); };
        auto vp1 = [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim_part<1>, dim3(B), dim3(256), 0, s, dg, dr, dt, nq, o); };
        auto vp2 = [=](hipStream_t s, f32x2* o) { hipLaunchKernelGGL(victim_part<2>, dim3(B), dim3(256), 0, s, dg, dr, dt, nq, o); };
        experiment("   only the two plane warps (compiler-packed), same aggressor", out_blocks, launches, vp0, c64(20 * cus, 1500));
        experiment("   only the normalisations + distances (divisions, square roots), same aggressor", out_blocks, launches, vp1, c64(20 * cus, 1500));
        experiment("   only the quaternion -> rotation build, same aggressor", out_blocks, launches, vp2, c64(20 * cus, 1500));
        experiment("   the victim's tables in global memory instead of LDS, same aggressor", out_blocks, launches, vsg, c64(20 * cus, 1500));
        experiment("   LDS victim, aggressor WITHOUT its LDS fragment reads", out_blocks, launches, vs, c64_nolds(20 * cus, 1500));
        experiment("   LDS victim, aggressor WITHOUT the weight-ring refill from global memory", out_blocks, launches, vs, c64_norefill(20 * cus, 1500));
        auto pure_v = [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor_pure<1>, dim3(3 * cus), dim3(256), 0, s, sink, 3000000); return true; };
        auto pure_a = [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor_pure<0>, dim3(3 * cus), dim3(256), 0, s, sink, 3000000); return true; };
        auto c64_bare = [=](hipStream_t s, float* sink) { hipLaunchKernelGGL((aggressor_c64like<0, 0>), dim3(20 * cus), dim3(256), 0, s, d_w, sink, 1500); return true; };
        experiment("   aggressor with NEITHER LDS reads NOR refill (MFMAs on a loaded-once ring + a barrier per round)", out_blocks, launches, vs, c64_bare);
        auto big128 = [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor_bigreg<128>, dim3(3 * cus), dim3(256), 0, s, sink, 40000000); return true; };
        auto big168 = [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor_bigreg<168>, dim3(3 * cus), dim3(256), 0, s, sink, 40000000); return true; };
        auto big0 = [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor_bigreg<0>, dim3(3 * cus), dim3(256), 0, s, sink, 40000000); return true; };
        experiment("   next to a VALU-only spin loop that allocates 128 VGPRs per wave (no MFMA, no LDS)", out_blocks, launches, vs, big128);
        experiment("   next to a VALU-only spin loop that allocates 168 VGPRs per wave", out_blocks, launches, vs, big168);
        experiment("   next to the same spin loop with a handful of VGPRs", out_blocks, launches, vs, big0);
        auto storm_v = [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor_storm<0>, dim3(4000000), dim3(256), 0, s, sink, 64); return true; };
        auto storm_m = [=](hipStream_t s, float* sink) { hipLaunchKernelGGL(aggressor_storm<1>, dim3(4000000), dim3(256), 0, s, sink, 16); return true; };
        experiment("   next to a wave-launch storm of short VALU-only workgroups (no MFMA anywhere)", out_blocks, launches, vs, storm_v);
        experiment("   next to a wave-launch storm of short MFMA workgroups", out_blocks, launches, vs, storm_m);
        experiment("   next to NOTHING BUT MFMAs whose destination tiles are VGPRs (3 blocks per CU)", out_blocks, launches, vs, pure_v);
        experiment("   next to NOTHING BUT MFMAs whose destination tiles are AGPRs (3 blocks per CU)", out_blocks, launches, vs, pure_a);
    }
    experiment("dense scalar victim next to an AGPR + LDS MFMA aggressor (3 blocks per CU)", 4 * cus, launches, v8s(4 * cus), agpr(3 * cus, 3000000));
    return 0;
}
#endif
