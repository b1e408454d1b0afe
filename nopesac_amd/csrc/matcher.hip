// Matching-head tail, one persistent workgroup per image pair:
//   geometric priors (matching_head.py:75-96)  ->  scores = D1.D2^T/16 - offset/4 - angle/8 (:113-119)
//   -> log-space Sinkhorn with a dustbin row/column, 200 iterations (:228-234, 259-306)
//   -> mutual-nearest-neighbour assignment with exp(score) > thr (camera_modules.py:15-34).
// The reference issues ~400 dependent tiny kernels for the Sinkhorn loop; here the whole (n1+1)x(n2+1)
// coupling matrix lives in LDS for the entire loop (<= 129x131 fp32 = 66 KB) and each iteration is two
// barrier-separated passes of group-of-lanes log-sum-exp reductions (wave shuffles, no atomics).
#include "common.h"

namespace nps {

constexpr float NEG_PAD = -1e30f;

__device__ __forceinline__ float group_max(float v, int tg) {
    for (int o = tg >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float group_sum(float v, int tg) {
    for (int o = tg >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// NT threads, ZREG coupling entries per lane and phase kept in registers.  1024 threads: a group of tg lanes (largest power of two
// with 1024 / tg >= rows) owns a row in the row phase and a column in the column phase; <1024, 16> covers nq <= 127 (tg >= 8),
// <1024, 36> nq = 128 (tg = 4, 33 entries; 148 B of spills at the 128-VGPR cap).  The first version ran 256 threads: nq = 64
// (BASELINE configs[2]) and above fell to an LDS loop with one or two LANES per row - 1.77 ms per launch at nq = 64, 6.9 ms at
// nq = 128 (configs[4]); now 0.72 / 2.77 ms; nq = 50: 0.88 -> 0.63 ms (32 pairs, full plane sets, 200 iterations; scripts/sinkhorn_one.py).
struct SinkLds {
    float *Z, *u, *v, *lmu, *lnu, *g1r, *g1rt, *o1, *g2, *o2, *max0;
    int *idx0, *idx1;
    float* xch;                                    // 4-wave kernel only: [2 phases][4 waves][64 lanes][m, s]
    int R, LD;
};

__device__ __forceinline__ SinkLds sink_lds(float* smem, int nq) {
    SinkLds L;
    L.R = nq + 1;
    L.LD = (L.R & 1) ? L.R : L.R + 1;              // odd leading dimension: column walks hit distinct banks
    const int R = L.R;
    L.Z = smem;                                    // R*LD
    L.u = L.Z + R * L.LD;                          // R
    L.v = L.u + R;                                 // R
    L.lmu = L.v + R;                               // R
    L.lnu = L.lmu + R;                             // R
    L.g1r = L.lnu + R;                             // nq*3 normals of view-1 planes warped by (R,0)
    L.g1rt = L.g1r + 3 * nq;                       // nq*3 normals warped by (R,t)
    L.o1 = L.g1rt + 3 * nq;                        // nq offsets (R,t)
    L.g2 = L.o1 + nq;                              // nq*3 flipped view-2 normals
    L.o2 = L.g2 + 3 * nq;                          // nq
    L.max0 = L.o2 + nq;                            // R
    L.idx0 = (int*)(L.max0 + R);                   // R
    L.idx1 = L.idx0 + R;                           // R
    L.xch = (float*)(L.idx1 + R);
    return L;
}
__host__ __device__ constexpr size_t sink_lds_floats(int nq) {
    return (size_t)(nq + 1) * (((nq + 1) & 1) ? (nq + 1) : (nq + 2)) + 4 * (nq + 1) + 11 * nq + 3 * (nq + 1);
}

// geometric priors + couplings + marginals (u = v = 0) into LDS; ends with a workgroup barrier
template <int NT>
__device__ __forceinline__ float sink_setup(const SinkLds& L, int b, int tid, int nq, int n1, int n2, const float* __restrict__ desc_dot,
                                            const float* __restrict__ planes1, const float* __restrict__ planes2, const float* __restrict__ cam7,
                                            const float* __restrict__ bin_score, float offset_mult, float normal_mult) {
    const int R = L.R, LD = L.LD, R1 = n1 + 1, C1 = n2 + 1;
    const float* cam = cam7 + 7 * b;
    // ---- per-plane geometry
    if (tid < nq) {
        const int i = tid;
        if (i < n1) {
            float Rm[9], q[4] = {cam[3], cam[4], cam[5], cam[6]}, t[3] = {cam[0], cam[1], cam[2]}, z[3] = {0.f, 0.f, 0.f};
            quat_to_rot(q, Rm);
            float p[3] = {planes1[((long long)b * nq + i) * 3], planes1[((long long)b * nq + i) * 3 + 1], planes1[((long long)b * nq + i) * 3 + 2]};
            // reference multiplies the translation by 0 (matching_head.py:82): t*0 keeps the sign of zero only
            float wr[3], wrt[3], nr[3], nrt[3];
            warp_plane(p, Rm, z, wr);
            warp_plane(p, Rm, t, wrt);
            normalize3(wr, nr);
            normalize3(wrt, nrt);
            for (int d = 0; d < 3; ++d) { L.g1r[3 * i + d] = nr[d]; L.g1rt[3 * i + d] = nrt[d]; }
            L.o1[i] = norm3(wrt);
        }
        if (i < n2) {
            float p[3] = {planes2[((long long)b * nq + i) * 3], -planes2[((long long)b * nq + i) * 3 + 1], -planes2[((long long)b * nq + i) * 3 + 2]};
            float nn[3];
            normalize3(p, nn);
            for (int d = 0; d < 3; ++d) L.g2[3 * i + d] = nn[d];
            L.o2[i] = norm3(p);
        }
    }
    __syncthreads();
    // ---- couplings
    const float bin = bin_score[0];
    const float* dd = desc_dot + (long long)b * nq * nq;
    for (int e = tid; e < R1 * C1; e += NT) {
        const int i = e / C1, j = e % C1;
        float val = bin;
        if (i < n1 && j < n2) {
            const float* a = L.g1r + 3 * i;
            const float* c = L.g2 + 3 * j;
            const float* at = L.g1rt + 3 * i;
            const float ntn_r = a[0] * c[0] + a[1] * c[1] + a[2] * c[2];
            const float ang = acosf(fminf(fmaxf(ntn_r, -1.f), 1.f)) / 3.14159265358979323846f * 180.f;
            const float ntn_rt = at[0] * c[0] + at[1] * c[1] + at[2] * c[2];
            float off = ntn_rt < 0.f ? fabsf(L.o1[i] + L.o2[j]) : fabsf(L.o1[i] - L.o2[j]);
            off = fminf(fmaxf(off, 1e-10f), 5.f);
            val = dd[i * nq + j] - off / offset_mult - ang / normal_mult;
        }
        L.Z[i * LD + j] = val;
    }
    const float norm = -logf((float)(n1 + n2));
    for (int i = tid; i < R; i += NT) {
        L.u[i] = 0.f; L.v[i] = 0.f;
        L.lmu[i] = i < n1 ? norm : logf((float)n2) + norm;
        L.lnu[i] = i < n2 ? norm : logf((float)n1) + norm;
    }
    __syncthreads();
    return norm;
}

// final scores Z + u + v - norm (kept in LDS for the assignment) + padded output + mutual nearest neighbours (u, v final in LDS)
template <int NT>
__device__ __forceinline__ void sink_finalize(const SinkLds& L, int b, int tid, int nq, int n1, int n2, float norm, float match_thr,
                                              float* __restrict__ log_scores, float* __restrict__ assignment) {
    const int R = L.R, LD = L.LD, R1 = n1 + 1, C1 = n2 + 1;
    float* Z = L.Z;
    for (int e = tid; e < R1 * C1; e += NT) {
        const int i = e / C1, j = e % C1;
        Z[i * LD + j] = Z[i * LD + j] + L.u[i] + L.v[j] - norm;
    }
    __syncthreads();
    float* out = log_scores + (long long)b * R * R;
    for (int e = tid; e < R * R; e += NT) {
        const int i = e / R, j = e % R;
        const int si = i < n1 ? i : (i == nq ? n1 : -1), sj = j < n2 ? j : (j == nq ? n2 : -1);
        out[e] = (si >= 0 && sj >= 0) ? Z[si * LD + sj] : NEG_PAD;
    }
    // ---- mutual nearest neighbours over the plane block
    for (int i = tid; i < n1; i += NT) {
        float m = -INFINITY; int a = 0;
        for (int j = 0; j < n2; ++j) { const float x = Z[i * LD + j]; if (x > m) { m = x; a = j; } }
        L.max0[i] = m; L.idx0[i] = a;
    }
    for (int j = tid; j < n2; j += NT) {
        float m = -INFINITY; int a = 0;
        for (int i = 0; i < n1; ++i) { const float x = Z[i * LD + j]; if (x > m) { m = x; a = i; } }
        L.idx1[j] = a;
    }
    __syncthreads();
    float* A = assignment + (long long)b * nq * nq;
    for (int e = tid; e < nq * nq; e += NT) {
        const int i = e / nq, j = e % nq;
        float a = 0.f;
        if (i < n1 && j < n2 && n2 > 0 && L.idx0[i] == j && L.idx1[j] == i && expf(L.max0[i]) > match_thr) a = 1.f;
        A[e] = a;
    }
}

template <int NT, int ZREG>
__global__ __launch_bounds__(NT) void matcher_sinkhorn_kernel(
    const float* __restrict__ desc_dot, const float* __restrict__ planes1, const float* __restrict__ planes2,
    const float* __restrict__ cam7, const int* __restrict__ n1p, const int* __restrict__ n2p,
    const float* __restrict__ bin_score, float offset_mult, float normal_mult, int iters, float match_thr, int nq,
    float* __restrict__ log_scores, float* __restrict__ assignment) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const SinkLds L = sink_lds(smem, nq);
    const int LD = L.LD;
    float *Z = L.Z, *u = L.u, *v = L.v, *lmu = L.lmu, *lnu = L.lnu;
    const int n1 = min(max(n1p[b], 0), nq), n2 = min(max(n2p[b], 0), nq);
    const int R1 = n1 + 1, C1 = n2 + 1;
    const float norm = sink_setup<NT>(L, b, tid, nq, n1, n2, desc_dot, planes1, planes2, cam7, bin_score, offset_mult, normal_mult);
    // ---- Sinkhorn: lane groups of tg lanes per row / column
    const int big = max(R1, C1);
    int tg = 64;
    while (tg > 1 && (NT / tg) < big) tg >>= 1;       // largest pow2 group with enough groups; may still need >1 pass
    const int ngroups = NT / tg, grp = tid / tg, gl = tid % tg;
    if (ngroups >= big && (big + tg - 1) / tg <= ZREG) {
        // Every lane group owns exactly one row (row phase) and one column (column phase): the lane's slice of both is
        // loop invariant, so it is read from LDS ONCE and the 200 iterations run out of registers (same partition, same
        // reduction order and therefore the same results as the LDS loop below; only u / v travel through LDS).
        float zr[ZREG], zc[ZREG];
        const bool rok = grp < R1, cok = grp < C1;
#pragma unroll
        for (int k = 0; k < ZREG; ++k) {
            const int j = gl + tg * k;
            zr[k] = (rok && j < C1) ? Z[grp * LD + j] : 0.f;
            zc[k] = (cok && j < R1) ? Z[j * LD + grp] : 0.f;
        }
        for (int it = 0; it < iters; ++it) {
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < ZREG; ++k) {
                const int j = gl + tg * k;
                if (rok && j < C1) m = fmaxf(m, zr[k] + v[j]);
            }
            m = group_max(m, tg);
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < ZREG; ++k) {
                const int j = gl + tg * k;
                if (rok && j < C1) sm += expf(zr[k] + v[j] - m);
            }
            sm = group_sum(sm, tg);
            if (rok && gl == 0) u[grp] = lmu[grp] - (m + logf(sm));
            __syncthreads();
            m = -INFINITY;
#pragma unroll
            for (int k = 0; k < ZREG; ++k) {
                const int i = gl + tg * k;
                if (cok && i < R1) m = fmaxf(m, zc[k] + u[i]);
            }
            m = group_max(m, tg);
            sm = 0.f;
#pragma unroll
            for (int k = 0; k < ZREG; ++k) {
                const int i = gl + tg * k;
                if (cok && i < R1) sm += expf(zc[k] + u[i] - m);
            }
            sm = group_sum(sm, tg);
            if (cok && gl == 0) v[grp] = lnu[grp] - (m + logf(sm));
            __syncthreads();
        }
    } else
    for (int it = 0; it < iters; ++it) {
        for (int i = grp; i < ((R1 + ngroups - 1) / ngroups) * ngroups; i += ngroups) {
            const bool ok = i < R1;
            float m = -INFINITY;
            if (ok) for (int j = gl; j < C1; j += tg) m = fmaxf(m, Z[i * LD + j] + v[j]);
            m = group_max(m, tg);
            float s = 0.f;
            if (ok) for (int j = gl; j < C1; j += tg) s += expf(Z[i * LD + j] + v[j] - m);
            s = group_sum(s, tg);
            if (ok && gl == 0) u[i] = lmu[i] - (m + logf(s));
        }
        __syncthreads();
        for (int j = grp; j < ((C1 + ngroups - 1) / ngroups) * ngroups; j += ngroups) {
            const bool ok = j < C1;
            float m = -INFINITY;
            if (ok) for (int i = gl; i < R1; i += tg) m = fmaxf(m, Z[i * LD + j] + u[i]);
            m = group_max(m, tg);
            float s = 0.f;
            if (ok) for (int i = gl; i < R1; i += tg) s += expf(Z[i * LD + j] + u[i] - m);
            s = group_sum(s, tg);
            if (ok && gl == 0) v[j] = lnu[j] - (m + logf(s));
        }
        __syncthreads();
    }
    sink_finalize<NT>(L, b, tid, nq, n1, n2, norm, match_thr, log_scores, assignment);
}

// Four-wave form for nq <= 63 (round 4; the reference's nq = 50): the 1024-thread kernel above spends its 2.6 us per iteration in
// two 16-wave barriers and two 4-step shuffle reductions per phase - the arithmetic of an iteration (2 x 51 x 51 exp) is 1 us of ONE
// SIMD.  Here a workgroup is four waves, one per SIMD, and LANE = ROW in the row phase, LANE = COLUMN in the column phase:
//   * wave w keeps, for every row i = lane, the KW couplings of columns w*KW .. (and, for every column j = lane, those of rows
//     w*KW ..) in registers; u (lane = row) and v (lane = column) live in one register each in EVERY wave;
//   * a phase: t_k = z_k + v[w*KW + k] (v_readlane with a wave-uniform index: no LDS, no shuffle), the wave's own max m_w and
//     s_w = sum exp(t_k - m_w) over its KW entries - no cross-lane reduction at all - then ONE exchange of (m_w, s_w) through LDS
//     and a 4-wave barrier; every wave merges the four partials the online-softmax way (M = max m_w, S = sum s_w exp(m_w - M):
//     exactly the max-shifted log-sum-exp of the whole row) and ends the phase with u (or v) complete in its own register;
//   * exp / log are the hardware v_exp_f32 / v_log_f32 forms (1 ulp; inputs are <= 0 after the max shift).
// 2 barriers of 4 waves per iteration instead of 2 of 16, no shuffles: ~0.7 us per iteration.
template <int KW>
__global__ __launch_bounds__(256) void matcher_sinkhorn_w4_kernel(
    const float* __restrict__ desc_dot, const float* __restrict__ planes1, const float* __restrict__ planes2,
    const float* __restrict__ cam7, const int* __restrict__ n1p, const int* __restrict__ n2p,
    const float* __restrict__ bin_score, float offset_mult, float normal_mult, int iters, float match_thr, int nq,
    float* __restrict__ log_scores, float* __restrict__ assignment) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SinkLds L = sink_lds(smem, nq);
    const int LD = L.LD;
    const int n1 = min(max(n1p[b], 0), nq), n2 = min(max(n2p[b], 0), nq);
    const int R1 = n1 + 1, C1 = n2 + 1;
    const float norm = sink_setup<256>(L, b, tid, nq, n1, n2, desc_dot, planes1, planes2, cam7, bin_score, offset_mult, normal_mult);
    float zr[KW], zc[KW];
    const bool rok = lane < R1, cok = lane < C1;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int e = w * KW + k;                  // this wave's k-th column (row phase) / row (column phase)
        zr[k] = (rok && e < C1) ? L.Z[lane * LD + e] : -INFINITY;
        zc[k] = (cok && e < R1) ? L.Z[e * LD + lane] : -INFINITY;
    }
    const float lmu_l = rok ? L.lmu[lane] : 0.f, lnu_l = cok ? L.lnu[lane] : 0.f;
    float uu = 0.f, vv = 0.f;                      // u[lane] / v[lane]; 0 outside the valid rows / columns
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t* xr = reinterpret_cast<f32x2_t*>(L.xch);                 // [4 waves][64 lanes] (m, s)
    f32x2_t* xc = xr + 64 * 4;
    auto phase = [&](const float (&z)[KW], float other, float marg, bool ok, f32x2_t* xch) -> float {
        float t[KW];
        float m = -1e30f;                          // finite floor: a wave whose KW entries are all masked contributes (floor, 0)
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            t[k] = z[k] + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, other), w * KW + k));
            m = fmaxf(m, t[k]);
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < KW; ++k) s += __expf(t[k] - m);
        xch[w * 64 + lane] = f32x2_t{m, s};                      // [wave][lane]: 8-byte lane stride, conflict-free writes and reads
        __syncthreads();
        const f32x2_t q0 = xch[lane], q1 = xch[64 + lane], q2 = xch[128 + lane], q3 = xch[192 + lane];
        const float M = fmaxf(fmaxf(q0[0], q1[0]), fmaxf(q2[0], q3[0]));
        const float S = q0[1] * __expf(q0[0] - M) + q1[1] * __expf(q1[0] - M) + q2[1] * __expf(q2[0] - M) + q3[1] * __expf(q3[0] - M);
        return ok ? marg - (M + __logf(S)) : 0.f;
    };
    for (int it = 0; it < iters; ++it) {
        uu = phase(zr, vv, lmu_l, rok, xr);
        vv = phase(zc, uu, lnu_l, cok, xc);
    }
    if (w == 0) {
        if (rok) L.u[lane] = uu;
        if (cok) L.v[lane] = vv;
    }
    __syncthreads();
    sink_finalize<256>(L, b, tid, nq, n1, n2, norm, match_thr, log_scores, assignment);
}

// Row-group form of the four-wave kernel for nq + 1 > 64 (round 6: BASELINE configs[2] runs nq = 64, configs[4] nq = 128; the 1024-thread
// kernel they used costs 0.71 / 2.8 ms per launch - 14 us per iteration at nq = 128 against ~1 us of exp arithmetic on one CU - and, as
// a 16-wave workgroup per pair, holds 32 whole CUs for that long next to the persistent conv kernels).  Same scheme as above with the
// rows / columns in RG groups of 64: 4 x RG waves, wave (g, c) keeps for row (column) 64 g + lane the KW couplings of columns (rows)
// c KW .. in registers; u and v live in RG registers (lane = index within a group of 64) in EVERY wave and are read with v_readlane
// (the group = index >> 6 is wave-uniform); per phase ONE exchange of the (max, sum) partials through LDS and ONE barrier - every wave
// merges the four partials of ALL row groups itself (RG x 4 exp instead of 4), so that the complete u (or v) is back in its own
// registers without a second barrier.  Groups behind max(n1, n2) + 1 rows only take part in the barriers.
template <int RG, int KW>
__global__ __launch_bounds__(256 * RG) void matcher_sinkhorn_wg_kernel(
    const float* __restrict__ desc_dot, const float* __restrict__ planes1, const float* __restrict__ planes2,
    const float* __restrict__ cam7, const int* __restrict__ n1p, const int* __restrict__ n2p,
    const float* __restrict__ bin_score, float offset_mult, float normal_mult, int iters, float match_thr, int nq,
    float* __restrict__ log_scores, float* __restrict__ assignment) {
    constexpr int NT = 256 * RG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), g = w >> 2, c = w & 3;
    const SinkLds L = sink_lds(smem, nq);
    const int LD = L.LD;
    const int n1 = min(max(n1p[b], 0), nq), n2 = min(max(n2p[b], 0), nq);
    const int R1 = n1 + 1, C1 = n2 + 1;
    const float norm = sink_setup<NT>(L, b, tid, nq, n1, n2, desc_dot, planes1, planes2, cam7, bin_score, offset_mult, normal_mult);
    const int ngrp = (max(R1, C1) + 63) >> 6;          // row / column groups in use (workgroup-uniform)
    const bool active = g < ngrp;
    const int idx = g * 64 + lane;                     // this lane's row (row phase) / column (column phase)
    float zr[KW], zc[KW];
    const bool rok = active && idx < R1, cok = active && idx < C1;
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int e = c * KW + k;
        zr[k] = (rok && e < C1) ? L.Z[idx * LD + e] : -INFINITY;
        zc[k] = (cok && e < R1) ? L.Z[e * LD + idx] : -INFINITY;
    }
    float lmu_l[RG], lnu_l[RG], uu[RG], vv[RG];        // marginals and potentials of rows / columns 64 gg + lane, gg < RG (every wave holds all)
#pragma unroll
    for (int gg = 0; gg < RG; ++gg) {
        const int i = gg * 64 + lane;
        lmu_l[gg] = i < R1 ? L.lmu[i] : 0.f;
        lnu_l[gg] = i < C1 ? L.lnu[i] : 0.f;
        uu[gg] = 0.f; vv[gg] = 0.f;
    }
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t* xr = reinterpret_cast<f32x2_t*>(L.xch);                 // [RG groups][4 column splits][64 lanes] (m, s)
    f32x2_t* xc = xr + 64 * 4 * RG;
    // the other potential as wave-uniform values: with several groups a v_readlane per group and a select per entry cost more than the
    // exp they feed (three readlanes + two scalar selects per coupling at nq = 128: 6.1 us per iteration).  Every wave parks its own copy
    // of the vector in a private LDS strip (its own writes, its own reads: no barrier, LDS operations of a wave execute in order) and
    // reads entry c KW + k back as a broadcast load
    float* priv = reinterpret_cast<float*>(xc + 64 * 4 * RG) + w * (64 * RG);
    auto phase = [&](const float (&z)[KW], const float (&other)[RG], const float (&marg)[RG], int n_valid, f32x2_t* xch, float (&res)[RG]) {
        if (active) {
#pragma unroll
            for (int gg = 0; gg < RG; ++gg) priv[gg * 64 + lane] = other[gg];
            float t[KW];
            float m = -1e30f;                          // finite floor: a wave whose KW entries are all masked contributes (floor, 0)
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                t[k] = z[k] + priv[c * KW + k];
                m = fmaxf(m, t[k]);
            }
            float sm = 0.f;
#pragma unroll
            for (int k = 0; k < KW; ++k) sm += __expf(t[k] - m);
            xch[w * 64 + lane] = f32x2_t{m, sm};
        }
        __syncthreads();
#pragma unroll
        for (int gg = 0; gg < RG; ++gg) {
            if (gg < ngrp) {
                const f32x2_t q0 = xch[(gg * 4) * 64 + lane], q1 = xch[(gg * 4 + 1) * 64 + lane], q2 = xch[(gg * 4 + 2) * 64 + lane],
                              q3 = xch[(gg * 4 + 3) * 64 + lane];
                const float M = fmaxf(fmaxf(q0[0], q1[0]), fmaxf(q2[0], q3[0]));
                const float S = q0[1] * __expf(q0[0] - M) + q1[1] * __expf(q1[0] - M) + q2[1] * __expf(q2[0] - M) + q3[1] * __expf(q3[0] - M);
                res[gg] = (gg * 64 + lane < n_valid) ? marg[gg] - (M + __logf(S)) : 0.f;
            }
        }
    };
    for (int it = 0; it < iters; ++it) {
        phase(zr, vv, lmu_l, R1, xr, uu);
        phase(zc, uu, lnu_l, C1, xc, vv);
    }
    if (w == 0) {
#pragma unroll
        for (int gg = 0; gg < RG; ++gg) {
            const int i = gg * 64 + lane;
            if (i < R1) L.u[i] = uu[gg];
            if (i < C1) L.v[i] = vv[gg];
        }
    }
    __syncthreads();
    sink_finalize<NT>(L, b, tid, nq, n1, n2, norm, match_thr, log_scores, assignment);
}

// assignment re-filter under the refined pose (camera_head.py:605-629)
__global__ __launch_bounds__(256) void refilter_kernel(const float* __restrict__ Ain, const float* __restrict__ planes1,
                                                       const float* __restrict__ planes2, const int* __restrict__ n1p,
                                                       const int* __restrict__ n2p, const float* __restrict__ rot,
                                                       const float* __restrict__ trans, int nq, float* __restrict__ Aout) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ float g1r[128 * 3], g1rt[128 * 3], o1[128], g2[128 * 3], o2[128];
    const int n1 = min(max(n1p[b], 0), nq), n2 = min(max(n2p[b], 0), nq);
    if (tid < nq) {
        const int i = tid;
        if (i < n1) {
            float q[4] = {rot[4 * b], rot[4 * b + 1], rot[4 * b + 2], rot[4 * b + 3]};
            if (q[0] < 0.f) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }   // :600-601
            float Rm[9], t[3] = {trans[3 * b], trans[3 * b + 1], trans[3 * b + 2]}, z[3] = {0.f, 0.f, 0.f};
            quat_to_rot(q, Rm);
            float p[3] = {planes1[((long long)b * nq + i) * 3], planes1[((long long)b * nq + i) * 3 + 1], planes1[((long long)b * nq + i) * 3 + 2]};
            float wr[3], wrt[3], nr[3], nrt[3];
            warp_plane(p, Rm, z, wr);
            warp_plane(p, Rm, t, wrt);
            normalize3(wr, nr);
            normalize3(wrt, nrt);
            for (int d = 0; d < 3; ++d) { g1r[3 * i + d] = nr[d]; g1rt[3 * i + d] = nrt[d]; }
            o1[i] = norm3(wrt);
        }
        if (i < n2) {
            float p[3] = {planes2[((long long)b * nq + i) * 3], -planes2[((long long)b * nq + i) * 3 + 1], -planes2[((long long)b * nq + i) * 3 + 2]};
            float nn[3];
            normalize3(p, nn);
            for (int d = 0; d < 3; ++d) g2[3 * i + d] = nn[d];
            o2[i] = norm3(p);
        }
    }
    __syncthreads();
    for (int e = tid; e < nq * nq; e += 256) {
        const int i = e / nq, j = e % nq;
        float a = 0.f;
        if (i < n1 && j < n2) {
            const float ntn_r = g1r[3 * i] * g2[3 * j] + g1r[3 * i + 1] * g2[3 * j + 1] + g1r[3 * i + 2] * g2[3 * j + 2];
            const float ang = acosf(fminf(fmaxf(ntn_r, -1.f), 1.f)) / 3.14159265358979323846f * 180.f;
            const float ntn_rt = g1rt[3 * i] * g2[3 * j] + g1rt[3 * i + 1] * g2[3 * j + 1] + g1rt[3 * i + 2] * g2[3 * j + 2];
            float off = ntn_rt < 0.f ? fabsf(o1[i] + o2[j]) : fabsf(o1[i] - o2[j]);
            off = fminf(fmaxf(off, 1e-4f), 10.f);
            a = Ain[(long long)b * nq * nq + e] * ((ang < 45.f && off < 1.f) ? 1.f : 0.f);
        }
        Aout[(long long)b * nq * nq + e] = a;
    }
}

}  // namespace nps

extern "C" int nopesac_matcher_sinkhorn(const float* desc_dot, const float* planes1, const float* planes2,
                                        const float* cam7, const int32_t* n1, const int32_t* n2,
                                        const float* bin_score, float offset_mult, float normal_mult, int iters,
                                        float match_thr, int B, int nq, float* log_scores, float* assignment,
                                        void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(desc_dot && planes1 && planes2 && cam7 && n1 && n2 && bin_score && log_scores && assignment, "sinkhorn: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128 && iters >= 0, "sinkhorn: bad dims (nq<=128)");
    const size_t lds = sizeof(float) * sink_lds_floats(nq);
    const bool no_w4 = getenv("NOPESAC_SINKHORN_NO_W4") != nullptr;            // A/B runs and the kernel-vs-kernel test (read per call)
    if (nq + 1 <= 64 && !no_w4) {     // four waves, lane = row / column, one exchange per phase (the reference's nq = 50)
        const size_t lds4 = lds + sizeof(float) * 2 * 64 * 4 * 2;
        if (nq + 1 <= 52)
            hipLaunchKernelGGL((matcher_sinkhorn_w4_kernel<13>), dim3(B), dim3(256), lds4, (hipStream_t)stream, desc_dot, planes1, planes2, cam7,
                               n1, n2, bin_score, offset_mult, normal_mult, iters, match_thr, nq, log_scores, assignment);
        else
            hipLaunchKernelGGL((matcher_sinkhorn_w4_kernel<16>), dim3(B), dim3(256), lds4, (hipStream_t)stream, desc_dot, planes1, planes2, cam7,
                               n1, n2, bin_score, offset_mult, normal_mult, iters, match_thr, nq, log_scores, assignment);
    } else if (!getenv("NOPESAC_SINKHORN_NO_WG")) {       // row groups of 64: 8 waves (nq + 1 <= 128) or 12 (nq = 128), one exchange per phase
        const int R = nq + 1;
        const int rg = R <= 128 ? 2 : 3;                // exchange buffers of both phases + one private strip of the potentials per wave
        const size_t ldsg = lds + sizeof(float) * (2 * 64 * 4 * 2 * rg + 4 * rg * 64 * rg);
#define NPS_SINK_WG(RG_, KW_)                                                                                                              \
        do {                                                                                                                               \
            if (ldsg > 64 * 1024) NPS_ENSURE_LDS(160 * 1024 - 256, (matcher_sinkhorn_wg_kernel<RG_, KW_>));                                \
            hipLaunchKernelGGL((matcher_sinkhorn_wg_kernel<RG_, KW_>), dim3(B), dim3(256 * RG_), ldsg, (hipStream_t)stream, desc_dot, planes1, \
                               planes2, cam7, n1, n2, bin_score, offset_mult, normal_mult, iters, match_thr, nq, log_scores, assignment);  \
        } while (0)
        if (R <= 4 * 17) NPS_SINK_WG(2, 17);
        else if (R <= 4 * 26) NPS_SINK_WG(2, 26);
        else if (R <= 128) NPS_SINK_WG(2, 32);
        else NPS_SINK_WG(3, 33);
#undef NPS_SINK_WG
    } else if (nq + 1 <= 128) {              // 1024 threads: lane groups of 8-64 lanes per row / column, <= 16 entries per lane (no spills)
        if (lds > 64 * 1024) NPS_ENSURE_LDS(160 * 1024 - 256, (matcher_sinkhorn_kernel<1024, 16>));
        hipLaunchKernelGGL((matcher_sinkhorn_kernel<1024, 16>), dim3(B), dim3(1024), lds, (hipStream_t)stream, desc_dot, planes1, planes2,
                           cam7, n1, n2, bin_score, offset_mult, normal_mult, iters, match_thr, nq, log_scores, assignment);
    } else {                          // nq = 128: groups of 4 lanes x 33 entries
        if (lds > 64 * 1024) NPS_ENSURE_LDS(160 * 1024 - 256, (matcher_sinkhorn_kernel<1024, 36>));
        hipLaunchKernelGGL((matcher_sinkhorn_kernel<1024, 36>), dim3(B), dim3(1024), lds, (hipStream_t)stream, desc_dot, planes1, planes2,
                           cam7, n1, n2, bin_score, offset_mult, normal_mult, iters, match_thr, nq, log_scores, assignment);
    }
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_refilter_assignment(const float* assignment_in, const float* planes1, const float* planes2,
                                           const int32_t* n1, const int32_t* n2, const float* rot, const float* trans,
                                           int B, int nq, float* assignment_out, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(assignment_in && planes1 && planes2 && n1 && n2 && rot && trans && assignment_out, "refilter: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128, "refilter: bad dims");
    hipLaunchKernelGGL(refilter_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, assignment_in, planes1, planes2, n1, n2,
                       rot, trans, nq, assignment_out);
    NPS_LAUNCH_RET();
}
