// Neural one-plane RANSAC, the wavefront-level (non-GEMM) parts (camera_net/camera_head.py):
//   geo_sequence      : matched pairs -> padded geometry sequences + the 8-d MLP input     (:512-569, 937-957, 1352-1425)
//   ransac_score_maps : (K+1) hypotheses x K matched planes: plane warps under each hypothesis,
//                       normal / plane-parameter distances -> exp(-d) score rows           (:990-1006, 1018-1035)
//   ransac_soft_vote  : score regression, masked softmax over the m+1 live hypotheses,
//                       soft / uniform aggregation of 256-d pose features, pose regression  (:1009-1014, 1038-1099)
// One workgroup per image pair; hypotheses x planes are spread over lanes, reductions over hypotheses /
// feature dims use wave shuffles + a 4-entry LDS exchange.  The MLP stacks between these kernels are
// MFMA GEMMs (conv_igemm.hip).
#include "common.h"

namespace nps {

__global__ __launch_bounds__(128) void geo_sequence_kernel(
    const float* __restrict__ A, const float* __restrict__ planes1, const float* __restrict__ planes2,
    const int* __restrict__ n1p, const int* __restrict__ n2p, const float* __restrict__ init_trans,
    const float* __restrict__ init_rot, int nq, int warp_in_ref, float* __restrict__ geo_local,
    float* __restrict__ geo_global, float* __restrict__ sig, float* __restrict__ geo_enc, int* __restrict__ m_out) {
    const int b = blockIdx.x, i = threadIdx.x;
    __shared__ int cnt[129];
    const int n1 = min(max(n1p[b], 0), nq), n2 = min(max(n2p[b], 0), nq);
    const float* Ab = A + (long long)b * nq * nq;
    int c = 0;
    if (i < n1)
        for (int j = 0; j < n2; ++j) c += Ab[i * nq + j] != 0.f;
    cnt[i] = c;
    __syncthreads();
    if (i == 0) {   // exclusive prefix over <=128 rows
        int run = 0;
        for (int r = 0; r < 128; ++r) { const int t = cnt[r]; cnt[r] = run; run += t; }
        cnt[128] = run;
    }
    __syncthreads();
    const int m = min(cnt[128], nq);
    if (i == 0) m_out[b] = m;
    float Rm[9], q[4] = {init_rot[4 * b], init_rot[4 * b + 1], init_rot[4 * b + 2], init_rot[4 * b + 3]};
    float t[3] = {init_trans[3 * b], init_trans[3 * b + 1], init_trans[3 * b + 2]}, z[3] = {0.f, 0.f, 0.f};
    quat_to_rot(q, Rm);
    if (i < n1) {
        int k = cnt[i];
        for (int j = 0; j < n2 && k < nq; ++j) {
            if (Ab[i * nq + j] == 0.f) continue;
            const long long o = (long long)b * nq + k;
            float p1[3], p2[3], g1[3], ga[3];
            for (int d = 0; d < 3; ++d) { p1[d] = planes1[((long long)b * nq + i) * 3 + d]; p2[d] = planes2[((long long)b * nq + j) * 3 + d]; }
            warp_plane(p1, Rm, t, g1);
            warp_plane(p1, Rm, z, ga);
            const float f2[3] = {p2[0], -p2[1], -p2[2]};
            const float sg = (g1[0] * ga[0] >= 0.f) ? 1.f : -1.f;
            for (int d = 0; d < 3; ++d) {
                geo_local[o * 6 + d] = p1[d]; geo_local[o * 6 + 3 + d] = p2[d];
                geo_global[o * 6 + d] = g1[d]; geo_global[o * 6 + 3 + d] = f2[d];
            }
            sig[o] = sg;
            const float* s0 = warp_in_ref ? g1 : p1;
            const float* s1 = warp_in_ref ? f2 : p2;
            float o0 = norm3(s0), o1 = norm3(s1);
            float e[8];
            for (int d = 0; d < 3; ++d) { e[d] = s0[d] / (o0 + 1e-10f); e[4 + d] = s1[d] / (o1 + 1e-10f); }
            e[3] = o0; e[7] = o1;
            if (warp_in_ref) { for (int d = 0; d < 4; ++d) e[d] *= sg; }
            for (int d = 0; d < 8; ++d) geo_enc[o * 8 + d] = e[d];
            ++k;
        }
    }
    // zero the padding rows [m, nq)
    for (int k = m + i; k < nq; k += 128) {
        const long long o = (long long)b * nq + k;
        for (int d = 0; d < 6; ++d) { geo_local[o * 6 + d] = 0.f; geo_global[o * 6 + d] = 0.f; }
        for (int d = 0; d < 8; ++d) geo_enc[o * 8 + d] = 0.f;
        sig[o] = 1.f;
    }
}

struct PairDist { float ang, dn, doff, dl2; };

__device__ __forceinline__ PairDist hyp_plane_dist(const float* gl, const float* Rm, const float* t) {
    const float p0[3] = {gl[0], gl[1], gl[2]};
    const float p1[3] = {gl[3], -gl[4], -gl[5]};
    const float z[3] = {0.f, 0.f, 0.f};
    float w_r[3], w_rt[3], n0[3], n1v[3], n0t[3];
    warp_plane(p0, Rm, z, w_r);
    warp_plane(p0, Rm, t, w_rt);
    normalize3(w_r, n0);
    normalize3(p1, n1v);
    normalize3(w_rt, n0t);
    PairDist r;
    const float c = n0[0] * n1v[0] + n0[1] * n1v[1] + n0[2] * n1v[2];
    r.ang = acosf(fminf(fmaxf(c, -1.f), 1.f)) / 3.14159265358979323846f * 180.f;
    const float d0 = n0[0] - n1v[0], d1 = n0[1] - n1v[1], d2 = n0[2] - n1v[2];
    r.dn = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float off0 = norm3(w_rt), off1 = norm3(p1);
    const float ntn = n0t[0] * n1v[0] + n0t[1] * n1v[1] + n0t[2] * n1v[2];
    r.doff = ntn < 0.f ? fabsf(off0 + off1) : fabsf(off0 - off1);
    const float e0 = w_rt[0] - p1[0], e1 = w_rt[1] - p1[1], e2 = w_rt[2] - p1[2];
    r.dl2 = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
    return r;
}

__global__ __launch_bounds__(256) void ransac_score_maps_kernel(
    const float* __restrict__ geo_local, const float* __restrict__ rot_raw, const float* __restrict__ trans_raw,
    const float* __restrict__ init_rot, const float* __restrict__ init_trans, const int* __restrict__ mp, int nq,
    float* __restrict__ rots_all, float* __restrict__ trans_all, float* __restrict__ normal_score,
    float* __restrict__ param_score, float* __restrict__ l2_dist, float* __restrict__ normal_angle,
    float* __restrict__ offset_dist, float* __restrict__ dn_sum, float* __restrict__ dl2_sum) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NH = nq + 1;
    __shared__ float sR[129 * 9], sT[129 * 3], sG[128 * 6];
    const int m = min(max(mp[b], 0), nq);
    for (int h = tid; h < NH; h += 256) {
        float q[4], t[3];
        if (h == 0) {
            for (int d = 0; d < 4; ++d) q[d] = init_rot[4 * b + d];
            for (int d = 0; d < 3; ++d) t[d] = init_trans[3 * b + d];
        } else {
            const float* rr = rot_raw + ((long long)b * nq + h - 1) * 4;
            const float nn = fmaxf(sqrtf(rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] + rr[3] * rr[3]), 1e-12f);
            for (int d = 0; d < 4; ++d) q[d] = rr[d] / nn;
            for (int d = 0; d < 3; ++d) t[d] = trans_raw[((long long)b * nq + h - 1) * 3 + d];
        }
        quat_to_rot(q, sR + 9 * h);
        for (int d = 0; d < 3; ++d) sT[3 * h + d] = t[d];
        for (int d = 0; d < 4; ++d) rots_all[((long long)b * NH + h) * 4 + d] = q[d];
        for (int d = 0; d < 3; ++d) trans_all[((long long)b * NH + h) * 3 + d] = t[d];
    }
    for (int e = tid; e < nq * 6; e += 256) sG[e] = geo_local[(long long)b * nq * 6 + e];
    __syncthreads();
    for (int e = tid; e < NH * nq; e += 256) {
        const int h = e / nq, j = e % nq;
        const PairDist d = hyp_plane_dist(sG + 6 * j, sR + 9 * h, sT + 3 * h);
        const float mask = (h <= m && j < m) ? 1.f : 0.f;
        const long long o = (long long)b * NH * nq + e;
        normal_score[o] = expf(-(d.dn * mask)) * mask;
        param_score[o] = expf(-(d.dl2 * mask)) * mask;
        if (l2_dist) l2_dist[o] = d.dl2;
        if (normal_angle) normal_angle[o] = d.ang;
        if (offset_dist) offset_dist[o] = d.doff;
    }
    // masked row sums in j order (deterministic); used by the 'min-cost' selection (:1005,1032,1090-1093)
    for (int h = tid; h < NH; h += 256) {
        float a = 0.f, c = 0.f;
        if (h <= m)
            for (int j = 0; j < m; ++j) {
                const PairDist d = hyp_plane_dist(sG + 6 * j, sR + 9 * h, sT + 3 * h);
                a += d.dn; c += d.dl2;
            }
        if (dn_sum) dn_sum[(long long)b * NH + h] = a;
        if (dl2_sum) dl2_sum[(long long)b * NH + h] = c;
    }
}

__device__ __forceinline__ float block_sum_256(float v, float* sh4) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

__global__ __launch_bounds__(256) void ransac_soft_vote_kernel(
    const float* __restrict__ sf_rot, const float* __restrict__ sf_trans, const float* __restrict__ reg_rot_w,
    const float* __restrict__ reg_rot_b, const float* __restrict__ reg_trans_w, const float* __restrict__ reg_trans_b,
    const float* __restrict__ init_rot_feat, const float* __restrict__ init_trans_feat,
    const float* __restrict__ fused_rot, const float* __restrict__ fused_trans, const float* __restrict__ rots_w,
    const float* __restrict__ rots_b, const float* __restrict__ trans_w, const float* __restrict__ trans_b,
    const float* __restrict__ rots_all, const float* __restrict__ trans_all, const float* __restrict__ dn_sum,
    const float* __restrict__ dl2_sum, const float* __restrict__ init_rot, const float* __restrict__ init_trans,
    const int* __restrict__ mp, int nq, int mode, float* __restrict__ pred_rot, float* __restrict__ pred_trans,
    float* __restrict__ avg_rot, float* __restrict__ avg_trans, float* __restrict__ score_rot,
    float* __restrict__ score_trans) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NH = nq + 1;
    __shared__ float s_r[129], s_t[129], sh4[4], outv[16];
    const int m = min(max(mp[b], 0), nq);
    // mode bit 4: the TRAINING-side twin (__forward_PlaneCamRefHead, camera_head.py:737-923): no m == 0 / m == 1 shortcuts, scores
    // clamped to [0.01, 0.9] and renormalised over the live hypotheses, average pose from the per-plane features only, soft pose out
    const bool train = (mode & 16) != 0;
    mode &= 15;
    for (int h = tid; h < NH; h += 256) { score_rot[(long long)b * NH + h] = 0.f; score_trans[(long long)b * NH + h] = 0.f; }
    if (m == 0 && !train) {   // :964-969
        if (tid < 4) { pred_rot[4 * b + tid] = init_rot[4 * b + tid]; avg_rot[4 * b + tid] = init_rot[4 * b + tid]; }
        if (tid < 3) { pred_trans[3 * b + tid] = init_trans[3 * b + tid]; avg_trans[3 * b + tid] = init_trans[3 * b + tid]; }
        return;
    }
    // ---- score regression (Linear 64->1) for the live hypotheses
    for (int h = tid; h <= m; h += 256) {
        const float* fr = sf_rot + ((long long)b * NH + h) * 64;
        const float* ft = sf_trans + ((long long)b * NH + h) * 64;
        float a = 0.f, c = 0.f;
        for (int d = 0; d < 64; ++d) { a = fmaf(fr[d], reg_rot_w[d], a); c = fmaf(ft[d], reg_trans_w[d], c); }
        s_r[h] = a + reg_rot_b[0];
        s_t[h] = c + reg_trans_b[0];
    }
    __syncthreads();
    if (tid < 2) {   // masked softmax over h in [0, m] (serial, <=129 terms, deterministic)
        float* s = tid == 0 ? s_r : s_t;
        float mx = -INFINITY;
        for (int h = 0; h <= m; ++h) mx = fmaxf(mx, s[h]);
        float sum = 0.f;
        for (int h = 0; h <= m; ++h) { s[h] = expf(s[h] - mx); sum += s[h]; }
        for (int h = 0; h <= m; ++h) s[h] = s[h] / sum;
        if (train) {   // :814-818, :852-854: clamp, mask (live = h <= m and m >= 1), renormalise
            const float live = m >= 1 ? 1.f : 0.f;
            float cs = 0.f;
            for (int h = 0; h <= m; ++h) { s[h] = fminf(fmaxf(s[h], 0.01f), 0.9f) * live; cs += s[h]; }
            for (int h = 0; h <= m; ++h) s[h] = s[h] / (cs + 1e-10f);
        }
    }
    __syncthreads();
    for (int h = tid; h <= m; h += 256) { score_rot[(long long)b * NH + h] = s_r[h]; score_trans[(long long)b * NH + h] = s_t[h]; }
    // ---- aggregate features: thread = feature dim
    const int d = tid;
    const float avg_w = 1.f / ((float)(m + 1) + 1e-10f);
    float fr_avg, ft_avg, fr_soft = 0.f, ft_soft = 0.f;
    const float* FR = fused_rot + (long long)b * nq * 256;
    const float* FT = fused_trans + (long long)b * nq * 256;
    if (train) {   // :856-873: average over the per-plane features (weights avg[1:] / avg[1:].sum()), soft sum over all live hypotheses
        const float a = m >= 1 ? avg_w : 0.f;
        float sa = 0.f;
        for (int h = 1; h <= m; ++h) sa += a;
        float ar = 0.f, at = 0.f;
        fr_soft = init_rot_feat[256 * b + d] * s_r[0];
        ft_soft = init_trans_feat[256 * b + d] * s_t[0];
        for (int h = 1; h <= m; ++h) {
            const float xr = FR[(h - 1) * 256 + d], xt = FT[(h - 1) * 256 + d];
            ar += xr * a / sa; at += xt * a / sa;
            fr_soft += xr * s_r[h]; ft_soft += xt * s_t[h];
        }
        if (m == 0) ar = at = a / sa;   // 0 / 0: the reference's own result for an empty sequence
        fr_avg = ar; ft_avg = at;
    } else if (m > 1) {
        float ar = init_rot_feat[256 * b + d] * avg_w, at = init_trans_feat[256 * b + d] * avg_w;
        fr_soft = init_rot_feat[256 * b + d] * s_r[0];
        ft_soft = init_trans_feat[256 * b + d] * s_t[0];
        for (int h = 1; h <= m; ++h) {
            const float xr = FR[(h - 1) * 256 + d], xt = FT[(h - 1) * 256 + d];
            ar += xr * avg_w; at += xt * avg_w;
            fr_soft += xr * s_r[h]; ft_soft += xt * s_t[h];
        }
        fr_avg = ar; ft_avg = at;
    } else {   // m == 1 (:1059-1063)
        fr_avg = FR[d] * avg_w / avg_w;
        ft_avg = FT[d] * avg_w / avg_w;
    }
    // ---- pose regression: 4 + 3 outputs for avg, 4 + 3 for soft
    float r[14];
    for (int o = 0; o < 4; ++o) r[o] = block_sum_256(fr_avg * rots_w[o * 256 + d], sh4);
    for (int o = 0; o < 3; ++o) r[4 + o] = block_sum_256(ft_avg * trans_w[o * 256 + d], sh4);
    for (int o = 0; o < 4; ++o) r[7 + o] = block_sum_256(fr_soft * rots_w[o * 256 + d], sh4);
    for (int o = 0; o < 3; ++o) r[11 + o] = block_sum_256(ft_soft * trans_w[o * 256 + d], sh4);
    if (tid == 0) {
        float ra[4], rs[4], na = 0.f, ns = 0.f;
        for (int o = 0; o < 4; ++o) { ra[o] = r[o] + rots_b[o]; rs[o] = r[7 + o] + rots_b[o]; na += ra[o] * ra[o]; ns += rs[o] * rs[o]; }
        na = fmaxf(sqrtf(na), 1e-12f); ns = fmaxf(sqrtf(ns), 1e-12f);
        float ta[3], ts[3];
        for (int o = 0; o < 3; ++o) { ta[o] = r[4 + o] + trans_b[o]; ts[o] = r[11 + o] + trans_b[o]; }
        for (int o = 0; o < 4; ++o) { ra[o] /= na; rs[o] /= ns; avg_rot[4 * b + o] = ra[o]; }
        for (int o = 0; o < 3; ++o) avg_trans[3 * b + o] = ta[o];
        float pr[4], pt[3];
        for (int o = 0; o < 4; ++o) pr[o] = ra[o];
        for (int o = 0; o < 3; ++o) pt[o] = ta[o];
        if (train) {
            for (int o = 0; o < 4; ++o) pr[o] = rs[o];
            for (int o = 0; o < 3; ++o) pt[o] = ts[o];
        } else if (m > 1) {
            if (mode == 0) {
                for (int o = 0; o < 4; ++o) pr[o] = rs[o];
                for (int o = 0; o < 3; ++o) pt[o] = ts[o];
            } else if (mode == 2 || mode == 3) {
                int hr = 0, ht = 0;
                if (mode == 2) {   // min geometric cost (:1088-1093)
                    float br = INFINITY, bt = INFINITY;
                    for (int h = 0; h <= m; ++h) {
                        if (dn_sum[(long long)b * NH + h] < br) { br = dn_sum[(long long)b * NH + h]; hr = h; }
                        if (dl2_sum[(long long)b * NH + h] < bt) { bt = dl2_sum[(long long)b * NH + h]; ht = h; }
                    }
                } else {           // max score (:1094-1099)
                    float br = -INFINITY, bt = -INFINITY;
                    for (int h = 0; h <= m; ++h) {
                        if (s_r[h] > br) { br = s_r[h]; hr = h; }
                        if (s_t[h] > bt) { bt = s_t[h]; ht = h; }
                    }
                }
                for (int o = 0; o < 4; ++o) pr[o] = rots_all[((long long)b * NH + hr) * 4 + o];
                for (int o = 0; o < 3; ++o) pt[o] = trans_all[((long long)b * NH + ht) * 3 + o];
            }
        }
        for (int o = 0; o < 4; ++o) pred_rot[4 * b + o] = pr[o];
        for (int o = 0; o < 3; ++o) pred_trans[3 * b + o] = pt[o];
    }
}

// The seven refinement losses of the training-side twin (camera_head.py:883-921; CameraPoseLoss camera_modules.py:355-365) from the
// outputs of the two kernels above: ONE wave, lane = pair (strided), fixed-order wave sums (deterministic).
// losses[0..6] = tran_planeAvgReg, rot_planeAvgReg, tran_planeSoftReg, rot_planeSoftReg, rotIdx, transIdx, paramL2_dist.
__device__ __forceinline__ float quat_dist_normalised(const float* a, const float* b) {
    const float na = fmaxf(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]), 1e-12f);
    const float nb = fmaxf(sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3]), 1e-12f);
    float s = 0.f;
    for (int d = 0; d < 4; ++d) { const float e = a[d] / na - b[d] / nb; s += e * e; }
    return sqrtf(s);
}
__device__ __forceinline__ float vec3_dist(const float* a, const float* b) {
    const float e0 = a[0] - b[0], e1 = a[1] - b[1], e2 = a[2] - b[2];
    return sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
}

__global__ __launch_bounds__(64) void plane_cam_ref_losses_kernel(
    const float* __restrict__ pred_rot, const float* __restrict__ pred_trans, const float* __restrict__ avg_rot,
    const float* __restrict__ avg_trans, const float* __restrict__ rots_all, const float* __restrict__ trans_all,
    const float* __restrict__ score_rot, const float* __restrict__ score_trans, const float* __restrict__ l2_dist,
    const int* __restrict__ mp, const float* __restrict__ gt_pose, int B, int nq, float weight, float* __restrict__ losses) {
    const int NH = nq + 1;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < B; b += 64) {
        const int m = min(max(mp[b], 0), nq);
        const float* gt = gt_pose + 7 * b;
        acc[0] += vec3_dist(gt, avg_trans + 3 * b);
        acc[1] += quat_dist_normalised(gt + 3, avg_rot + 4 * b);
        acc[2] += vec3_dist(gt, pred_trans + 3 * b);
        acc[3] += quat_dist_normalised(gt + 3, pred_rot + 4 * b);
        // best one-plane hypotheses against the ground truth (dead hypotheses count 1e10; first minimum as torch.min)
        float br = INFINITY, bt = INFINITY;
        int hr = 0, ht = 0;
        for (int h = 0; h < NH; ++h) {
            const bool live = h <= m && m >= 1;
            const float er = live ? quat_dist_normalised(gt + 3, rots_all + ((long long)b * NH + h) * 4) : 1e10f;
            const float et = live ? vec3_dist(gt, trans_all + ((long long)b * NH + h) * 3) : 1e10f;
            if (er < br) { br = er; hr = h; }
            if (et < bt) { bt = et; ht = h; }
        }
        acc[4] += fabsf(1.f - score_rot[(long long)b * NH + hr]);
        acc[5] += fabsf(1.f - score_trans[(long long)b * NH + ht]);
        float dsum = 0.f;   // diag(dist_l2_mid_ori[b, 1:]) over ALL nq planes (padded pairs contribute 0), :906-908
        for (int j = 0; j < nq; ++j) dsum += l2_dist[((long long)b * NH + 1 + j) * nq + j];
        acc[6] += dsum / (float)mp[b];
    }
    for (int k = 0; k < 7; ++k) acc[k] = wave_sum(acc[k]);
    if (threadIdx.x == 0) {
        const float inv = 1.f / (float)B;
        losses[0] = acc[0] * inv * weight;
        losses[1] = acc[1] * inv * weight;
        losses[2] = acc[2] * inv * weight;
        losses[3] = acc[3] * inv * weight;
        losses[4] = acc[4] * inv * 0.01f * weight;
        losses[5] = acc[5] * inv * 0.02f * weight;
        losses[6] = acc[6] * inv * 0.1f * weight;
    }
}

// CameraPoseLoss.forward, reduce=True, no mask (camera_modules.py:355-365), also the form of the AIM's reconstruction losses
// (camera_head.py:700-705, :725-731): out[0] = mean_b |gt_t + trans_eps - est_t| * weight, out[1] = mean_b |n(gt_q) - n(est_q)| * weight.
__global__ __launch_bounds__(64) void camera_pose_loss_kernel(const float* __restrict__ est_trans, const float* __restrict__ est_rot,
                                                              const float* __restrict__ gt_trans, int gt_trans_stride,
                                                              const float* __restrict__ gt_rot, int gt_rot_stride, int B,
                                                              float trans_eps, float weight, float* __restrict__ out) {
    float lx = 0.f, lq = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) {
        const float* gt = gt_trans + (long long)b * gt_trans_stride;
        const float g[3] = {gt[0] + trans_eps, gt[1] + trans_eps, gt[2] + trans_eps};
        lx += vec3_dist(g, est_trans + 3 * b);
        lq += quat_dist_normalised(gt_rot + (long long)b * gt_rot_stride, est_rot + 4 * b);
    }
    lx = wave_sum(lx); lq = wave_sum(lq);
    if (threadIdx.x == 0) { out[0] = lx / (float)B * weight; out[1] = lq / (float)B * weight; }
}

// BENCHMARK-ONLY K control (PlaneTR_NopeSAC._force_k, oracle force_k): with the name-seeded random weights the
// threshold-based selection keeps ~1 plane per view, so the stages behind it would see K = 1.  One workgroup per pair: the K
// highest-scoring queries of view 1 (score = logit 0 - logit 1; ascending query order, as topk(..).indices.sort() gives them) become
// the kept planes of view 1; view 2 gets the SAME embeddings permuted (+ 1 % noise).  Everything the 18 torch launches of the first
// version did (topk, sort, two gathers, add, zero fill, scatter), in one launch: feats [2B,nq,D] is written completely.
__global__ __launch_bounds__(256) void force_k_select_kernel(const float* __restrict__ logits, int n_cls, const float* __restrict__ query_feat,
                                                             const long long* __restrict__ perm, const float* __restrict__ noise, int B, int nq,
                                                             int K, int D, float* __restrict__ feats, int32_t* __restrict__ n_kept) {
    __shared__ float sc[128];
    __shared__ int kept[128];                                // kept[q] = 1 iff query q is among the K best
    __shared__ int sel[128];                                 // sel[k] = query index of the k-th kept plane (ascending)
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < nq) sc[tid] = logits[((long long)b * nq + tid) * n_cls] - logits[((long long)b * nq + tid) * n_cls + 1];
    __syncthreads();
    if (tid < nq) {
        // rank by descending score (ties: lower index first); kept iff rank < K; its slot = number of kept queries below it
        const float s = sc[tid];
        int rank = 0;
        for (int j = 0; j < nq; ++j) rank += (sc[j] > s) || (sc[j] == s && j < tid);
        kept[tid] = rank < K ? 1 : 0;                        // (a separate array: for nq > 64 another wave may still be ranking against sc[])
    }
    __syncthreads();
    if (tid < nq && kept[tid]) {
        int pos = 0;
        for (int j = 0; j < tid; ++j) pos += kept[j];
        sel[pos] = tid;
    }
    if (tid == 0) { n_kept[b] = K; n_kept[B + b] = K; }
    __syncthreads();
    for (int k = 0; k < nq; ++k) {
        float* o1 = feats + ((long long)b * nq + k) * D;
        float* o2 = feats + ((long long)(B + b) * nq + k) * D;
        if (k < K) {
            const float* f1 = query_feat + ((long long)b * nq + sel[k]) * D;
            const float* f2 = query_feat + ((long long)b * nq + sel[(int)perm[(long long)b * K + k]]) * D;
            const float* nz = noise + ((long long)b * K + k) * D;
            for (int c = tid; c < D; c += 256) { o1[c] = f1[c]; o2[c] = f2[c] + nz[c]; }
        } else {
            for (int c = tid; c < D; c += 256) { o1[c] = 0.f; o2[c] = 0.f; }
        }
    }
}

}  // namespace nps

extern "C" int nopesac_force_k_select(const float* logits, int n_cls, const float* query_feat, const int64_t* perm, const float* noise, int B,
                                      int nq, int K, int D, float* feats, int32_t* n_kept, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(logits && query_feat && perm && noise && feats && n_kept, "force_k_select: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128 && K > 0 && K <= nq && n_cls >= 2 && D > 0, "force_k_select: bad dims (K <= nq <= 128)");
    hipLaunchKernelGGL(force_k_select_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, n_cls, query_feat, (const long long*)perm, noise,
                       B, nq, K, D, feats, n_kept);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_geo_sequence(const float* assignment, const float* planes1, const float* planes2,
                                    const int32_t* n1, const int32_t* n2, const float* init_trans,
                                    const float* init_rot, int B, int nq, int warp_in_ref, float* geo_local,
                                    float* geo_global, float* sig, float* geo_enc, int32_t* m, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(assignment && planes1 && planes2 && n1 && n2 && init_trans && init_rot && geo_local && geo_global && sig && geo_enc && m,
                  "geo_sequence: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128, "geo_sequence: bad dims (nq<=128)");
    hipLaunchKernelGGL(geo_sequence_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, assignment, planes1, planes2, n1, n2,
                       init_trans, init_rot, nq, warp_in_ref, geo_local, geo_global, sig, geo_enc, m);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_ransac_score_maps(const float* geo_local, const float* rot_raw, const float* trans_raw,
                                         const float* init_rot, const float* init_trans, const int32_t* m, int B,
                                         int nq, float* rots_all, float* trans_all, float* normal_score,
                                         float* param_score, float* l2_dist, float* normal_angle, float* offset_dist,
                                         float* dn_sum, float* dl2_sum, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(geo_local && rot_raw && trans_raw && init_rot && init_trans && m && rots_all && trans_all && normal_score && param_score,
                  "ransac_score_maps: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128, "ransac_score_maps: bad dims (nq<=128)");
    hipLaunchKernelGGL(ransac_score_maps_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, geo_local, rot_raw, trans_raw,
                       init_rot, init_trans, m, nq, rots_all, trans_all, normal_score, param_score, l2_dist, normal_angle,
                       offset_dist, dn_sum, dl2_sum);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_ransac_soft_vote(const float* score_feat_rot, const float* score_feat_trans,
                                        const float* reg_rot_w, const float* reg_rot_b, const float* reg_trans_w,
                                        const float* reg_trans_b, const float* init_rot_feat,
                                        const float* init_trans_feat, const float* fused_rot_feat,
                                        const float* fused_trans_feat, const float* rots_w, const float* rots_b,
                                        const float* trans_w, const float* trans_b, const float* rots_all,
                                        const float* trans_all, const float* dn_sum, const float* dl2_sum,
                                        const float* init_rot, const float* init_trans, const int32_t* m, int B, int nq,
                                        int mode, float* pred_rot, float* pred_trans, float* avg_rot, float* avg_trans,
                                        float* score_rot, float* score_trans, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(score_feat_rot && score_feat_trans && reg_rot_w && reg_rot_b && reg_trans_w && reg_trans_b && init_rot_feat &&
                      init_trans_feat && fused_rot_feat && fused_trans_feat && rots_w && rots_b && trans_w && trans_b && rots_all &&
                      trans_all && init_rot && init_trans && m && pred_rot && pred_trans && avg_rot && avg_trans && score_rot && score_trans,
                  "ransac_soft_vote: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128 && mode >= 0 && ((mode & 15) <= 3) && mode < 32, "ransac_soft_vote: bad dims/mode");
    NPS_CHECK_ARG((mode & 15) != 2 || (dn_sum && dl2_sum), "ransac_soft_vote: min-cost needs dn_sum/dl2_sum");
    hipLaunchKernelGGL(ransac_soft_vote_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, score_feat_rot, score_feat_trans,
                       reg_rot_w, reg_rot_b, reg_trans_w, reg_trans_b, init_rot_feat, init_trans_feat, fused_rot_feat,
                       fused_trans_feat, rots_w, rots_b, trans_w, trans_b, rots_all, trans_all, dn_sum, dl2_sum, init_rot,
                       init_trans, m, nq, mode, pred_rot, pred_trans, avg_rot, avg_trans, score_rot, score_trans);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_plane_cam_ref_losses(const float* pred_rot, const float* pred_trans, const float* avg_rot,
                                            const float* avg_trans, const float* rots_all, const float* trans_all,
                                            const float* score_rot, const float* score_trans, const float* l2_dist,
                                            const int32_t* m, const float* gt_pose, int B, int nq, float weight,
                                            float* losses, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(pred_rot && pred_trans && avg_rot && avg_trans && rots_all && trans_all && score_rot && score_trans && l2_dist && m &&
                      gt_pose && losses, "plane_cam_ref_losses: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128, "plane_cam_ref_losses: bad dims (nq<=128)");
    hipLaunchKernelGGL(plane_cam_ref_losses_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pred_rot, pred_trans, avg_rot, avg_trans,
                       rots_all, trans_all, score_rot, score_trans, l2_dist, m, gt_pose, B, nq, weight, losses);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_camera_pose_loss(const float* est_trans, const float* est_rot, const float* gt_trans, int gt_trans_stride,
                                        const float* gt_rot, int gt_rot_stride, int B, float trans_eps, float weight, float* out,
                                        void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(est_trans && est_rot && gt_trans && gt_rot && out, "camera_pose_loss: null pointer");
    NPS_CHECK_ARG(B > 0 && gt_trans_stride >= 3 && gt_rot_stride >= 4, "camera_pose_loss: bad dims / strides");
    hipLaunchKernelGGL(camera_pose_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, est_trans, est_rot, gt_trans, gt_trans_stride,
                       gt_rot, gt_rot_stride, B, trans_eps, weight, out);
    NPS_LAUNCH_RET();
}
