// Backward (gradient) kernels of the camera head's training-side twin (reference: __forward_PlaneCamRefHead,
// camera_net/camera_head.py:737-923, CameraPoseLoss camera_modules.py:355-365) - round 5.  The forward twins are ransac_score_maps_kernel,
// ransac_soft_vote_kernel (mode | 16) and plane_cam_ref_losses_kernel (ransac.hip); each kernel here is the vector-Jacobian product of one
// of them, recomputing the forward's small intermediates instead of storing them.  The Linear / MLP stacks between them are differentiated
// with the library's f32 GEMM kernel (dgrad = dY W, wgrad = dY^T X: nopesac_amd/training.py) plus the small kernels at the end of this file
// (transpose, column sums, ReLU mask, row-normalisation backward, AdamW / SGD step).  Everything is f32 and deterministic: per-pair
// partial sums of the parameter gradients are written per pair and reduced by nopesac_col_sum_f32 in a fixed order - no atomics.
// Gated against torch.autograd on the oracle (tests/test_training_gpu.py).
#include "common.h"

namespace nps {

__device__ __forceinline__ float rb_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float rb_block_sum_256(float v, float* sh4) {
    v = rb_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

// J^T g of y = x / max(|x|, 1e-12) at x (D <= 4)
__device__ __forceinline__ void normalize_bwd(const float* x, const float* g, int D, float* gx) {
    float n2 = 0.f;
    for (int d = 0; d < D; ++d) n2 += x[d] * x[d];
    const float n = sqrtf(n2);
    if (n < 1e-12f) {                                          // y = x / 1e-12 there
        for (int d = 0; d < D; ++d) gx[d] = g[d] / 1e-12f;
        return;
    }
    float dot = 0.f;
    for (int d = 0; d < D; ++d) dot += g[d] * x[d] / n;
    for (int d = 0; d < D; ++d) gx[d] = (g[d] - dot * x[d] / n) / n;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// (1) the seven losses.  g_loss[7] -> gradients of every differentiable input of plane_cam_ref_losses_kernel.  The hypothesis picked by
// the two index losses (first minimum of the error against the ground truth over the live hypotheses) is a constant of the backward
// pass, as in autograd.  One thread per pair.
__global__ __launch_bounds__(64) void refine_losses_bwd_kernel(
    const float* __restrict__ pred_rot, const float* __restrict__ pred_trans, const float* __restrict__ avg_rot,
    const float* __restrict__ avg_trans, const float* __restrict__ rots_all, const float* __restrict__ trans_all,
    const float* __restrict__ score_rot, const float* __restrict__ score_trans, const int* __restrict__ mp,
    const float* __restrict__ gt_pose, const float* __restrict__ g_loss, int B, int nq, float weight,
    float* __restrict__ g_pred_rot, float* __restrict__ g_pred_trans, float* __restrict__ g_avg_rot, float* __restrict__ g_avg_trans,
    float* __restrict__ g_score_rot, float* __restrict__ g_score_trans, float* __restrict__ g_l2_dist) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int NH = nq + 1;
    const int m = min(max(mp[b], 0), nq);
    const float* gt = gt_pose + 7 * b;
    const float invB = 1.f / (float)B;
    // translation terms: d |gt - t| / dt = (t - gt) / |gt - t|
    auto tran_term = [&](const float* t, float gl, float* out) {
        float e[3], n2 = 0.f;
        for (int d = 0; d < 3; ++d) { e[d] = t[d] - gt[d]; n2 += e[d] * e[d]; }
        const float n = sqrtf(n2);
        for (int d = 0; d < 3; ++d) out[d] = n > 0.f ? gl * weight * invB * e[d] / n : 0.f;
    };
    // rotation terms: l = |n(gt_q) - n(q)|: dl / d n(q) = (n(q) - n(gt_q)) / l, then through the normalisation of q
    auto rot_term = [&](const float* q, float gl, float* out) {
        float gq[4], nq_[4], gn = 0.f, qn = 0.f;
        for (int d = 0; d < 4; ++d) { gn += gt[3 + d] * gt[3 + d]; qn += q[d] * q[d]; }
        gn = fmaxf(sqrtf(gn), 1e-12f); qn = fmaxf(sqrtf(qn), 1e-12f);
        float l2 = 0.f;
        for (int d = 0; d < 4; ++d) { gq[d] = gt[3 + d] / gn; nq_[d] = q[d] / qn; const float e = gq[d] - nq_[d]; l2 += e * e; }
        const float l = sqrtf(l2);
        float gy[4];
        for (int d = 0; d < 4; ++d) gy[d] = l > 0.f ? gl * weight * invB * (nq_[d] - gq[d]) / l : 0.f;
        normalize_bwd(q, gy, 4, out);
    };
    tran_term(avg_trans + 3 * b, g_loss[0], g_avg_trans + 3 * b);
    rot_term(avg_rot + 4 * b, g_loss[1], g_avg_rot + 4 * b);
    tran_term(pred_trans + 3 * b, g_loss[2], g_pred_trans + 3 * b);
    rot_term(pred_rot + 4 * b, g_loss[3], g_pred_rot + 4 * b);
    float br = INFINITY, bt = INFINITY;
    int hr = 0, ht = 0;
    for (int h = 0; h < NH; ++h) {
        const bool live = h <= m && m >= 1;
        float er = 1e10f, et = 1e10f;
        if (live) {
            const float* q = rots_all + ((long long)b * NH + h) * 4;
            float gn = 0.f, qn = 0.f, s = 0.f;
            for (int d = 0; d < 4; ++d) { gn += gt[3 + d] * gt[3 + d]; qn += q[d] * q[d]; }
            gn = fmaxf(sqrtf(gn), 1e-12f); qn = fmaxf(sqrtf(qn), 1e-12f);
            for (int d = 0; d < 4; ++d) { const float e = gt[3 + d] / gn - q[d] / qn; s += e * e; }
            er = sqrtf(s);
            const float* t = trans_all + ((long long)b * NH + h) * 3;
            s = 0.f;
            for (int d = 0; d < 3; ++d) { const float e = gt[d] - t[d]; s += e * e; }
            et = sqrtf(s);
        }
        if (er < br) { br = er; hr = h; }
        if (et < bt) { bt = et; ht = h; }
    }
    for (int h = 0; h < NH; ++h) { g_score_rot[(long long)b * NH + h] = 0.f; g_score_trans[(long long)b * NH + h] = 0.f; }
    {   // d |1 - s| / ds = -sign(1 - s)
        const float sr = score_rot[(long long)b * NH + hr], st = score_trans[(long long)b * NH + ht];
        const float dr = 1.f - sr, dt = 1.f - st;
        g_score_rot[(long long)b * NH + hr] = g_loss[4] * 0.01f * weight * invB * (dr > 0.f ? -1.f : (dr < 0.f ? 1.f : 0.f));
        g_score_trans[(long long)b * NH + ht] = g_loss[5] * 0.02f * weight * invB * (dt > 0.f ? -1.f : (dt < 0.f ? 1.f : 0.f));
    }
    for (int h = 0; h < NH; ++h)
        for (int j = 0; j < nq; ++j) g_l2_dist[((long long)b * NH + h) * nq + j] = 0.f;
    const float gd = g_loss[6] * 0.1f * weight * invB / (float)mp[b];
    for (int j = 0; j < nq; ++j) g_l2_dist[((long long)b * NH + 1 + j) * nq + j] = gd;
}

// (1b) CameraPoseLoss / the AIM's reconstruction losses (camera_pose_loss_kernel): out[0] = w mean_b |gt_t + eps - est_t|,
// out[1] = w mean_b |n(gt_q) - n(est_q)|.  Both sides get a gradient: in the reconstruction losses the "ground truth" is the pixel
// pose, an output of trainable layers.
__global__ __launch_bounds__(64) void camera_pose_loss_bwd_kernel(const float* __restrict__ est_trans, const float* __restrict__ est_rot,
                                                                  const float* __restrict__ gt_trans, int gt_trans_stride, const float* __restrict__ gt_rot,
                                                                  int gt_rot_stride, int B, float trans_eps, float weight, const float* __restrict__ g_out,
                                                                  float* __restrict__ g_est_trans, float* __restrict__ g_est_rot,
                                                                  float* __restrict__ g_gt_trans, float* __restrict__ g_gt_rot) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float k = weight / (float)B;
    {
        const float* gt = gt_trans + (long long)b * gt_trans_stride;
        float e[3], n2 = 0.f;
        for (int d = 0; d < 3; ++d) { e[d] = est_trans[3 * b + d] - (gt[d] + trans_eps); n2 += e[d] * e[d]; }
        const float n = sqrtf(n2);
        for (int d = 0; d < 3; ++d) {
            const float v = n > 0.f ? g_out[0] * k * e[d] / n : 0.f;
            g_est_trans[3 * b + d] = v;
            g_gt_trans[3 * b + d] = -v;
        }
    }
    {
        const float* gq = gt_rot + (long long)b * gt_rot_stride;
        const float* eq = est_rot + 4 * b;
        float gn = 0.f, en = 0.f;
        for (int d = 0; d < 4; ++d) { gn += gq[d] * gq[d]; en += eq[d] * eq[d]; }
        gn = fmaxf(sqrtf(gn), 1e-12f); en = fmaxf(sqrtf(en), 1e-12f);
        float diff[4], l2 = 0.f;
        for (int d = 0; d < 4; ++d) { diff[d] = eq[d] / en - gq[d] / gn; l2 += diff[d] * diff[d]; }
        const float l = sqrtf(l2);
        float ge[4], gg[4], xe[4], xg[4];
        for (int d = 0; d < 4; ++d) {
            ge[d] = l > 0.f ? g_out[1] * k * diff[d] / l : 0.f;
            gg[d] = -ge[d];
            xe[d] = eq[d]; xg[d] = gq[d];
        }
        float oe[4], og[4];
        normalize_bwd(xe, ge, 4, oe);
        normalize_bwd(xg, gg, 4, og);
        for (int d = 0; d < 4; ++d) { g_est_rot[4 * b + d] = oe[d]; g_gt_rot[4 * b + d] = og[d]; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// (2) scoring + aggregation + pose heads (ransac_soft_vote_kernel, mode | 16).  One workgroup per pair, thread = feature dim.
// Parameter gradients are written PER PAIR ([B, ...]) and reduced over the pairs afterwards (nopesac_col_sum_f32).
__global__ __launch_bounds__(256) void refine_vote_bwd_kernel(
    const float* __restrict__ sf_rot, const float* __restrict__ sf_trans, const float* __restrict__ reg_rot_w,
    const float* __restrict__ reg_rot_b, const float* __restrict__ reg_trans_w, const float* __restrict__ reg_trans_b,
    const float* __restrict__ init_rot_feat, const float* __restrict__ init_trans_feat, const float* __restrict__ fused_rot,
    const float* __restrict__ fused_trans, const float* __restrict__ rots_w, const float* __restrict__ rots_b,
    const float* __restrict__ trans_w, const float* __restrict__ trans_b, const int* __restrict__ mp, int nq,
    const float* __restrict__ g_pred_rot, const float* __restrict__ g_pred_trans, const float* __restrict__ g_avg_rot,
    const float* __restrict__ g_avg_trans, const float* __restrict__ g_score_rot, const float* __restrict__ g_score_trans,
    float* __restrict__ g_sf_rot, float* __restrict__ g_sf_trans, float* __restrict__ g_init_rot_feat,
    float* __restrict__ g_init_trans_feat, float* __restrict__ g_fused_rot, float* __restrict__ g_fused_trans,
    float* __restrict__ pb_rots_w, float* __restrict__ pb_rots_b, float* __restrict__ pb_trans_w, float* __restrict__ pb_trans_b,
    float* __restrict__ pb_reg_rot_w, float* __restrict__ pb_reg_rot_b, float* __restrict__ pb_reg_trans_w,
    float* __restrict__ pb_reg_trans_b) {
    const int b = blockIdx.x, tid = threadIdx.x, d = tid;
    const int NH = nq + 1;
    __shared__ float raw_r[129], raw_t[129], p_r[129], p_t[129], s_r[129], s_t[129], gs_r[129], gs_t[129], gr_r[129], gr_t[129];
    __shared__ float sh4[4], S2[2], head[14], ghead[14];
    const int m = min(max(mp[b], 0), nq);
    const float live = m >= 1 ? 1.f : 0.f;
    // ---- forward recompute: raw scores, softmax, clamp, renormalisation
    for (int h = tid; h < NH; h += 256) {
        float a = 0.f, c = 0.f;
        if (h <= m) {
            const float* fr = sf_rot + ((long long)b * NH + h) * 64;
            const float* ft = sf_trans + ((long long)b * NH + h) * 64;
            for (int k = 0; k < 64; ++k) { a = fmaf(fr[k], reg_rot_w[k], a); c = fmaf(ft[k], reg_trans_w[k], c); }
            a += reg_rot_b[0]; c += reg_trans_b[0];
        }
        raw_r[h] = a; raw_t[h] = c;
    }
    __syncthreads();
    if (tid < 2) {
        float* raw = tid == 0 ? raw_r : raw_t;
        float* p = tid == 0 ? p_r : p_t;
        float* s = tid == 0 ? s_r : s_t;
        float mx = -INFINITY;
        for (int h = 0; h <= m; ++h) mx = fmaxf(mx, raw[h]);
        float sum = 0.f;
        for (int h = 0; h <= m; ++h) { p[h] = expf(raw[h] - mx); sum += p[h]; }
        float cs = 0.f;
        for (int h = 0; h <= m; ++h) { p[h] = p[h] / sum; s[h] = fminf(fmaxf(p[h], 0.01f), 0.9f) * live; cs += s[h]; }
        for (int h = 0; h <= m; ++h) s[h] = s[h] / (cs + 1e-10f);
        for (int h = m + 1; h < NH; ++h) { p[h] = 0.f; s[h] = 0.f; }
        S2[tid] = cs;
    }
    __syncthreads();
    // ---- forward recompute: aggregated features (this thread's dim) and the four head outputs
    const float* FR = fused_rot + (long long)b * nq * 256;
    const float* FT = fused_trans + (long long)b * nq * 256;
    const float wavg = m >= 1 ? 1.f / (float)m : 0.f;
    float fr_soft = init_rot_feat[256 * b + d] * s_r[0], ft_soft = init_trans_feat[256 * b + d] * s_t[0], fr_avg = 0.f, ft_avg = 0.f;
    for (int h = 1; h <= m; ++h) {
        const float xr = FR[(h - 1) * 256 + d], xt = FT[(h - 1) * 256 + d];
        fr_avg += xr * wavg; ft_avg += xt * wavg;
        fr_soft += xr * s_r[h]; ft_soft += xt * s_t[h];
    }
    float r[14];
    for (int o = 0; o < 4; ++o) r[o] = rb_block_sum_256(fr_avg * rots_w[o * 256 + d], sh4);
    for (int o = 0; o < 3; ++o) r[4 + o] = rb_block_sum_256(ft_avg * trans_w[o * 256 + d], sh4);
    for (int o = 0; o < 4; ++o) r[7 + o] = rb_block_sum_256(fr_soft * rots_w[o * 256 + d], sh4);
    for (int o = 0; o < 3; ++o) r[11 + o] = rb_block_sum_256(ft_soft * trans_w[o * 256 + d], sh4);
    if (tid == 0) {
        for (int o = 0; o < 4; ++o) { head[o] = r[o] + rots_b[o]; head[7 + o] = r[7 + o] + rots_b[o]; }
        for (int o = 0; o < 3; ++o) { head[4 + o] = r[4 + o] + trans_b[o]; head[11 + o] = r[11 + o] + trans_b[o]; }
        // gradients of the RAW head outputs: rotations through their normalisation, translations directly
        normalize_bwd(head, g_avg_rot + 4 * b, 4, ghead);
        normalize_bwd(head + 7, g_pred_rot + 4 * b, 4, ghead + 7);
        for (int o = 0; o < 3; ++o) { ghead[4 + o] = g_avg_trans[3 * b + o]; ghead[11 + o] = g_pred_trans[3 * b + o]; }
        for (int o = 0; o < 4; ++o) pb_rots_b[4 * b + o] = ghead[o] + ghead[7 + o];
        for (int o = 0; o < 3; ++o) pb_trans_b[3 * b + o] = ghead[4 + o] + ghead[11 + o];
    }
    __syncthreads();
    // ---- heads backward: dW (per pair) and the gradients of the aggregated features
    float g_fr_avg = 0.f, g_ft_avg = 0.f, g_fr_soft = 0.f, g_ft_soft = 0.f;
    for (int o = 0; o < 4; ++o) {
        pb_rots_w[((long long)b * 4 + o) * 256 + d] = ghead[o] * fr_avg + ghead[7 + o] * fr_soft;
        g_fr_avg += ghead[o] * rots_w[o * 256 + d];
        g_fr_soft += ghead[7 + o] * rots_w[o * 256 + d];
    }
    for (int o = 0; o < 3; ++o) {
        pb_trans_w[((long long)b * 3 + o) * 256 + d] = ghead[4 + o] * ft_avg + ghead[11 + o] * ft_soft;
        g_ft_avg += ghead[4 + o] * trans_w[o * 256 + d];
        g_ft_soft += ghead[11 + o] * trans_w[o * 256 + d];
    }
    // ---- aggregation backward: features, and the score gradients g_s[h] = g_score_in[h] + <g_soft, feat[h]>
    g_init_rot_feat[256 * b + d] = s_r[0] * g_fr_soft;
    g_init_trans_feat[256 * b + d] = s_t[0] * g_ft_soft;
    for (int k = 0; k < nq; ++k) {
        const int h = k + 1;
        const bool on = h <= m;
        g_fused_rot[((long long)b * nq + k) * 256 + d] = on ? s_r[h] * g_fr_soft + wavg * g_fr_avg : 0.f;
        g_fused_trans[((long long)b * nq + k) * 256 + d] = on ? s_t[h] * g_ft_soft + wavg * g_ft_avg : 0.f;
    }
    for (int h = 0; h <= m; ++h) {
        const float xr = h == 0 ? init_rot_feat[256 * b + d] : FR[(h - 1) * 256 + d];
        const float xt = h == 0 ? init_trans_feat[256 * b + d] : FT[(h - 1) * 256 + d];
        const float a = rb_block_sum_256(xr * g_fr_soft, sh4), c = rb_block_sum_256(xt * g_ft_soft, sh4);
        if (tid == 0) { gs_r[h] = a + g_score_rot[(long long)b * NH + h]; gs_t[h] = c + g_score_trans[(long long)b * NH + h]; }
    }
    __syncthreads();
    // ---- renormalisation, clamp, softmax backward (serial over <= 129 hypotheses, fixed order)
    if (tid < 2) {
        const float* p = tid == 0 ? p_r : p_t;
        const float* s = tid == 0 ? s_r : s_t;
        const float* gs = tid == 0 ? gs_r : gs_t;
        float* gr = tid == 0 ? gr_r : gr_t;
        const float cs = S2[tid];
        float dot = 0.f;
        for (int h = 0; h <= m; ++h) dot += gs[h] * s[h];
        float dot2 = 0.f;
        for (int h = 0; h <= m; ++h) {
            const float gc = (gs[h] - dot) / (cs + 1e-10f);
            const float gp = (p[h] >= 0.01f && p[h] <= 0.9f) ? gc * live : 0.f;         // torch.clamp passes the gradient on [min, max]
            gr[h] = gp;
            dot2 += gp * p[h];
        }
        for (int h = 0; h <= m; ++h) gr[h] = p[h] * (gr[h] - dot2);
        for (int h = m + 1; h < NH; ++h) gr[h] = 0.f;
        float sb = 0.f;
        for (int h = 0; h <= m; ++h) sb += gr[h];
        (tid == 0 ? pb_reg_rot_b : pb_reg_trans_b)[b] = sb;
    }
    __syncthreads();
    // ---- score regression (Linear 64 -> 1) backward
    for (int e = tid; e < NH * 64; e += 256) {
        const int h = e >> 6, k = e & 63;
        g_sf_rot[((long long)b * NH + h) * 64 + k] = gr_r[h] * reg_rot_w[k];
        g_sf_trans[((long long)b * NH + h) * 64 + k] = gr_t[h] * reg_trans_w[k];
    }
    if (tid < 128) {
        const int k = tid & 63;
        const bool rot = tid < 64;
        const float* sf = rot ? sf_rot : sf_trans;
        const float* gr = rot ? gr_r : gr_t;
        float a = 0.f;
        for (int h = 0; h <= m; ++h) a += gr[h] * sf[((long long)b * NH + h) * 64 + k];
        (rot ? pb_reg_rot_w : pb_reg_trans_w)[(long long)b * 64 + k] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// (3) hypothesis x plane geometry (ransac_score_maps_kernel).  Inputs: the gradients of normal_score = exp(-dn) mask, param_score =
// exp(-dl2) mask and of l2_dist (unmasked: the parameter loss reads its diagonal).  One thread per (pair, hypothesis): its planes are
// walked in order, the gradient of the rotation MATRIX is accumulated and turned into the quaternion's once.
__device__ __forceinline__ void warp_plane_bwd(const float p[3], const float R[9], const float t[3], const float g[3], float GR[9], float gt_[3]) {
    const float f[3] = {p[0], -p[1], -p[2]};
    float u[3], e[3];
    for (int i = 0; i < 3; ++i) {
        u[i] = R[3 * i] * f[0] + R[3 * i + 1] * f[1] + R[3 * i + 2] * f[2];
        e[i] = u[i] + t[i];
    }
    const float nu = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (nu == 0.f) return;                                     // a padded (zero) plane: out = 0 whatever R, t
    const float nb = nu + 1e-5f;
    const float dot = e[0] * u[0] + e[1] * u[1] + e[2] * u[2];
    const float c = dot / (nb * nb);
    const float gu_dot = g[0] * u[0] + g[1] * u[1] + g[2] * u[2];
    float gu[3];
    for (int i = 0; i < 3; ++i) {
        // dc/du = (2u + t) / nb^2 - 2 dot u / (nb^3 |u|);   dc/dt = u / nb^2
        const float dc = (2.f * u[i] + t[i]) / (nb * nb) - 2.f * dot * u[i] / (nb * nb * nb * nu);
        gu[i] = c * g[i] + gu_dot * dc;
        gt_[i] += gu_dot * u[i] / (nb * nb);
    }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) GR[3 * i + k] += gu[i] * f[k];
}

__global__ __launch_bounds__(64) void refine_score_maps_bwd_kernel(
    const float* __restrict__ geo_local, const float* __restrict__ rot_raw, const float* __restrict__ trans_raw,
    const float* __restrict__ init_rot, const float* __restrict__ init_trans, const int* __restrict__ mp, int B, int nq,
    const float* __restrict__ g_normal_score, const float* __restrict__ g_param_score, const float* __restrict__ g_l2_dist,
    float* __restrict__ g_rot_raw, float* __restrict__ g_trans_raw, float* __restrict__ g_init_rot, float* __restrict__ g_init_trans) {
    const int NH = nq + 1;
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= B * NH) return;
    const int b = idx / NH, h = idx % NH;
    const int m = min(max(mp[b], 0), nq);
    float q[4], t[3], raw[4] = {0.f, 0.f, 0.f, 0.f};
    if (h == 0) {
        for (int d = 0; d < 4; ++d) q[d] = init_rot[4 * b + d];
        for (int d = 0; d < 3; ++d) t[d] = init_trans[3 * b + d];
    } else {
        const float* rr = rot_raw + ((long long)b * nq + h - 1) * 4;
        const float nn = fmaxf(sqrtf(rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] + rr[3] * rr[3]), 1e-12f);
        for (int d = 0; d < 4; ++d) { raw[d] = rr[d]; q[d] = rr[d] / nn; }
        for (int d = 0; d < 3; ++d) t[d] = trans_raw[((long long)b * nq + h - 1) * 3 + d];
    }
    float R[9], GR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gt_[3] = {0.f, 0.f, 0.f}, gz[3];
    quat_to_rot(q, R);
    const float z[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < nq; ++j) {
        const float* gl = geo_local + ((long long)b * nq + j) * 6;
        const long long o = ((long long)b * NH + h) * nq + j;
        const float mask = (h <= m && j < m) ? 1.f : 0.f;
        const float g_ns = g_normal_score[o] * mask, g_ps = g_param_score[o] * mask, g_l2 = g_l2_dist[o];
        if (g_ns == 0.f && g_ps == 0.f && g_l2 == 0.f) continue;
        const float p0[3] = {gl[0], gl[1], gl[2]};
        const float p1[3] = {gl[3], -gl[4], -gl[5]};
        float w_r[3], w_rt[3], n0[3], n1v[3];
        warp_plane(p0, R, z, w_r);
        warp_plane(p0, R, t, w_rt);
        normalize3(w_r, n0);
        normalize3(p1, n1v);
        // normal distance dn = |n0 - n1|, score exp(-dn * mask) * mask
        if (g_ns != 0.f) {
            const float d0 = n0[0] - n1v[0], d1 = n0[1] - n1v[1], d2 = n0[2] - n1v[2];
            const float dn = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            const float nw = norm3(w_r);
            if (dn > 0.f && nw >= 1e-12f) {
                const float gdn = -expf(-dn) * g_ns;
                const float gn0[3] = {gdn * d0 / dn, gdn * d1 / dn, gdn * d2 / dn};
                const float dt = gn0[0] * n0[0] + gn0[1] * n0[1] + gn0[2] * n0[2];
                const float gw[3] = {(gn0[0] - dt * n0[0]) / nw, (gn0[1] - dt * n0[1]) / nw, (gn0[2] - dt * n0[2]) / nw};
                gz[0] = gz[1] = gz[2] = 0.f;
                warp_plane_bwd(p0, R, z, gw, GR, gz);              // (t = 0 is a constant of this warp: gz is dropped)
            }
        }
        // parameter distance dl2 = |w_rt - p1|: score exp(-dl2 * mask) * mask and the (unmasked) l2_dist output
        {
            const float e0 = w_rt[0] - p1[0], e1 = w_rt[1] - p1[1], e2 = w_rt[2] - p1[2];
            const float dl2 = sqrtf(e0 * e0 + e1 * e1 + e2 * e2);
            if (dl2 > 0.f) {
                const float gd = -expf(-dl2) * g_ps + g_l2;
                const float gw[3] = {gd * e0 / dl2, gd * e1 / dl2, gd * e2 / dl2};
                warp_plane_bwd(p0, R, t, gw, GR, gt_);
            }
        }
    }
    // rotation matrix -> quaternion (the polynomial of quat_to_rot, differentiated as written)
    const float w = q[0], x = q[1], y = q[2], zz = q[3];
    float gq[4];
    gq[0] = 2.f * (-zz * GR[1] + y * GR[2] + zz * GR[3] - x * GR[5] - y * GR[6] + x * GR[7]);
    gq[1] = 2.f * (y * GR[1] + zz * GR[2] + y * GR[3] - 2.f * x * GR[4] - w * GR[5] + zz * GR[6] + w * GR[7] - 2.f * x * GR[8]);
    gq[2] = 2.f * (-2.f * y * GR[0] + x * GR[1] + w * GR[2] + x * GR[3] + zz * GR[5] - w * GR[6] + zz * GR[7] - 2.f * y * GR[8]);
    gq[3] = 2.f * (-2.f * zz * GR[0] - w * GR[1] + x * GR[2] + w * GR[3] - 2.f * zz * GR[4] + y * GR[5] + x * GR[6] + y * GR[7]);
    if (h == 0) {
        for (int d = 0; d < 4; ++d) g_init_rot[4 * b + d] = gq[d];
        for (int d = 0; d < 3; ++d) g_init_trans[3 * b + d] = gt_[d];
    } else {
        float gr[4];
        normalize_bwd(raw, gq, 4, gr);
        for (int d = 0; d < 4; ++d) g_rot_raw[((long long)b * nq + h - 1) * 4 + d] = gr[d];
        for (int d = 0; d < 3; ++d) g_trans_raw[((long long)b * nq + h - 1) * 3 + d] = gt_[d];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// small f32 utilities of the Linear / MLP backward and of the optimiser step
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ x, int rows, int cols, long long x_ld, float* __restrict__ y) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = x[(long long)(r0 + i) * x_ld + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < cols && r0 + tx < rows) y[(long long)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

// out[c] = sum_r x[r, c]: 64 columns per workgroup, four row phases summed in a fixed order
__global__ __launch_bounds__(256) void col_sum_f32_kernel(const float* __restrict__ x, int rows, int cols, long long x_ld, float* __restrict__ out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    float a = 0.f;
    if (c < cols)
        for (int r = ph; r < rows; r += 4) a += x[(long long)r * x_ld + c];
    part[ph][threadIdx.x & 63] = a;
    __syncthreads();
    if (ph == 0 && c < cols) out[c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void relu_bwd_f32_kernel(const float* __restrict__ g, const float* __restrict__ y, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = y[i] > 0.f ? g[i] : 0.f;
}

// canonical: the forward was y = sign * x / |x| with sign = -1 where x[0] < 0 (ops.normalize_rows(canonical_sign=True), camera_head.py:695-696)
__global__ __launch_bounds__(256) void normalize_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, int rows, int D, int canonical,
                                                                 float* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    float xv[4], gv[4], o[4];
    const float sg = (canonical && x[(long long)r * D] < 0.f) ? -1.f : 1.f;
    for (int d = 0; d < D; ++d) { xv[d] = x[(long long)r * D + d]; gv[d] = sg * g[(long long)r * D + d]; }
    normalize_bwd(xv, gv, D, o);
    for (int d = 0; d < D; ++d) out[(long long)r * D + d] = o[d];
}

// sum of squares of one tensor -> out[0] += (accumulated across tensors by launching on the same `out`; ONE workgroup, fixed order: the
// global gradient norm of torch.nn.utils.clip_grad_norm_ is deterministic) ; scale: x *= s[0]
__global__ __launch_bounds__(256) void sumsq_accum_kernel(const float* __restrict__ x, long long n, float* __restrict__ out) {
    __shared__ float sh4[4];
    float a = 0.f;
    for (long long i = threadIdx.x; i < n; i += 256) a += x[i] * x[i];
    a = rb_block_sum_256(a, sh4);
    if (threadIdx.x == 0) out[0] += a;
}

// large tensors (the 1024 x 1024 weights): two stages so that the whole chip reads the tensor - up to 256 workgroups each sum one contiguous
// slice (stride-256 order inside the slice), then ONE workgroup adds the partial sums in index order.  Fixed slices, fixed order: still
// bit-identical run to run (no atomics)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n, long long chunk, float* __restrict__ partials) {
    __shared__ float sh4[4];
    const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    float a = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) a += x[i] * x[i];
    a = rb_block_sum_256(a, sh4);
    if (threadIdx.x == 0) partials[blockIdx.x] = a;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partials, int nb, float* __restrict__ out) {
    __shared__ float sh4[4];
    float a = (int)threadIdx.x < nb ? partials[threadIdx.x] : 0.f;
    a = rb_block_sum_256(a, sh4);
    if (threadIdx.x == 0) out[0] += a;
}

// clip coefficient of clip_grad_norm_: c = min(1, max_norm / (sqrt(sumsq) + 1e-6)) -> coef[0]
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef) {
    const float total = sqrtf(sumsq[0]);
    const float c = max_norm / (total + 1e-6f);
    coef[0] = c < 1.f ? c : 1.f;
}

__global__ __launch_bounds__(256) void scale_by_kernel(float* __restrict__ x, long long n, const float* __restrict__ coef) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] *= coef[0];
}

// AdamW (torch.optim.AdamW semantics: decoupled weight decay, bias-corrected moments) / SGD with momentum, one launch per tensor
__global__ __launch_bounds__(256) void adamw_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2,
                                                         long long n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float w = p[i];
    const float gi = g[i];
    w *= 1.f - lr * wd;
    const float a = beta1 * m1[i] + (1.f - beta1) * gi;
    const float v = beta2 * m2[i] + (1.f - beta2) * gi * gi;
    m1[i] = a; m2[i] = v;
    const float denom = sqrtf(v) / sqrtf(bc2) + eps;
    p[i] = w - (lr / bc1) * a / denom;
}

__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mom, long long n, float lr,
                                                       float momentum, float wd, int first) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gi = g[i] + wd * p[i];
    if (momentum != 0.f) {
        const float bu = first ? gi : momentum * mom[i] + gi;
        mom[i] = bu;
        gi = bu;
    }
    p[i] -= lr * gi;
}

}  // namespace nps

extern "C" int nopesac_refine_losses_backward(const float* pred_rot, const float* pred_trans, const float* avg_rot, const float* avg_trans,
                                              const float* rots_all, const float* trans_all, const float* score_rot, const float* score_trans,
                                              const int32_t* m, const float* gt_pose, const float* g_loss, int B, int nq, float weight,
                                              float* g_pred_rot, float* g_pred_trans, float* g_avg_rot, float* g_avg_trans, float* g_score_rot,
                                              float* g_score_trans, float* g_l2_dist, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(pred_rot && pred_trans && avg_rot && avg_trans && rots_all && trans_all && score_rot && score_trans && m && gt_pose && g_loss &&
                      g_pred_rot && g_pred_trans && g_avg_rot && g_avg_trans && g_score_rot && g_score_trans && g_l2_dist,
                  "refine_losses_backward: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128, "refine_losses_backward: bad dims (nq<=128)");
    hipLaunchKernelGGL(refine_losses_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, pred_rot, pred_trans, avg_rot, avg_trans,
                       rots_all, trans_all, score_rot, score_trans, m, gt_pose, g_loss, B, nq, weight, g_pred_rot, g_pred_trans, g_avg_rot,
                       g_avg_trans, g_score_rot, g_score_trans, g_l2_dist);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_refine_vote_backward(const float* sf_rot, const float* sf_trans, const float* reg_rot_w, const float* reg_rot_b,
                                            const float* reg_trans_w, const float* reg_trans_b, const float* init_rot_feat,
                                            const float* init_trans_feat, const float* fused_rot, const float* fused_trans, const float* rots_w,
                                            const float* rots_b, const float* trans_w, const float* trans_b, const int32_t* m, int B, int nq,
                                            const float* g_pred_rot, const float* g_pred_trans, const float* g_avg_rot, const float* g_avg_trans,
                                            const float* g_score_rot, const float* g_score_trans, float* g_sf_rot, float* g_sf_trans,
                                            float* g_init_rot_feat, float* g_init_trans_feat, float* g_fused_rot, float* g_fused_trans,
                                            float* pb_rots_w, float* pb_rots_b, float* pb_trans_w, float* pb_trans_b, float* pb_reg_rot_w,
                                            float* pb_reg_rot_b, float* pb_reg_trans_w, float* pb_reg_trans_b, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(sf_rot && sf_trans && reg_rot_w && reg_rot_b && reg_trans_w && reg_trans_b && init_rot_feat && init_trans_feat && fused_rot &&
                      fused_trans && rots_w && rots_b && trans_w && trans_b && m && g_pred_rot && g_pred_trans && g_avg_rot && g_avg_trans &&
                      g_score_rot && g_score_trans && g_sf_rot && g_sf_trans && g_init_rot_feat && g_init_trans_feat && g_fused_rot &&
                      g_fused_trans && pb_rots_w && pb_rots_b && pb_trans_w && pb_trans_b && pb_reg_rot_w && pb_reg_rot_b && pb_reg_trans_w &&
                      pb_reg_trans_b, "refine_vote_backward: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128, "refine_vote_backward: bad dims (nq<=128)");
    hipLaunchKernelGGL(refine_vote_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, sf_rot, sf_trans, reg_rot_w, reg_rot_b, reg_trans_w,
                       reg_trans_b, init_rot_feat, init_trans_feat, fused_rot, fused_trans, rots_w, rots_b, trans_w, trans_b, m, nq, g_pred_rot,
                       g_pred_trans, g_avg_rot, g_avg_trans, g_score_rot, g_score_trans, g_sf_rot, g_sf_trans, g_init_rot_feat, g_init_trans_feat,
                       g_fused_rot, g_fused_trans, pb_rots_w, pb_rots_b, pb_trans_w, pb_trans_b, pb_reg_rot_w, pb_reg_rot_b, pb_reg_trans_w,
                       pb_reg_trans_b);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_refine_score_maps_backward(const float* geo_local, const float* rot_raw, const float* trans_raw, const float* init_rot,
                                                  const float* init_trans, const int32_t* m, int B, int nq, const float* g_normal_score,
                                                  const float* g_param_score, const float* g_l2_dist, float* g_rot_raw, float* g_trans_raw,
                                                  float* g_init_rot, float* g_init_trans, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(geo_local && rot_raw && trans_raw && init_rot && init_trans && m && g_normal_score && g_param_score && g_l2_dist && g_rot_raw &&
                      g_trans_raw && g_init_rot && g_init_trans, "refine_score_maps_backward: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128, "refine_score_maps_backward: bad dims (nq<=128)");
    const int n = B * (nq + 1);
    hipLaunchKernelGGL(refine_score_maps_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, geo_local, rot_raw, trans_raw, init_rot,
                       init_trans, m, B, nq, g_normal_score, g_param_score, g_l2_dist, g_rot_raw, g_trans_raw, g_init_rot, g_init_trans);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_camera_pose_loss_backward(const float* est_trans, const float* est_rot, const float* gt_trans, int gt_trans_stride,
                                                 const float* gt_rot, int gt_rot_stride, int B, float trans_eps, float weight, const float* g_out,
                                                 float* g_est_trans, float* g_est_rot, float* g_gt_trans, float* g_gt_rot, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(est_trans && est_rot && gt_trans && gt_rot && g_out && g_est_trans && g_est_rot && g_gt_trans && g_gt_rot,
                  "camera_pose_loss_backward: null pointer");
    NPS_CHECK_ARG(B > 0 && gt_trans_stride >= 3 && gt_rot_stride >= 4, "camera_pose_loss_backward: bad dims / strides");
    hipLaunchKernelGGL(camera_pose_loss_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, est_trans, est_rot, gt_trans, gt_trans_stride,
                       gt_rot, gt_rot_stride, B, trans_eps, weight, g_out, g_est_trans, g_est_rot, g_gt_trans, g_gt_rot);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_transpose_f32(const float* x, int rows, int cols, int64_t x_ld, float* y, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && y && rows > 0 && cols > 0 && x_ld >= cols, "transpose_f32: bad arguments");
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, rows, cols, (long long)x_ld, y);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_col_sum_f32(const float* x, int rows, int cols, int64_t x_ld, float* out, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && out && rows > 0 && cols > 0 && x_ld >= cols, "col_sum_f32: bad arguments");
    hipLaunchKernelGGL(col_sum_f32_kernel, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, rows, cols, (long long)x_ld, out);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_relu_backward_f32(const float* g, const float* y, int64_t n, float* out, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(g && y && out && n > 0, "relu_backward_f32: bad arguments");
    hipLaunchKernelGGL(relu_bwd_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, y, (long long)n, out);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_normalize_rows_backward(const float* x, const float* g, int rows, int D, int canonical_sign, float* out, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && g && out && rows > 0 && D >= 1 && D <= 4, "normalize_rows_backward: bad arguments (D <= 4)");
    hipLaunchKernelGGL(normalize_rows_bwd_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, g, rows, D, canonical_sign, out);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_sumsq_accumulate_f32(const float* x, int64_t n, float* out, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && out && n > 0, "sumsq_accumulate_f32: bad arguments");
    if (n <= 16384) {
        hipLaunchKernelGGL(sumsq_accum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, (long long)n, out);
    } else {
        // slices of a multiple of 4096 elements, at most 256 of them; the partial sums live in stream-ordered scratch
        long long chunk = ((n + 255) / 256 + 4095) / 4096 * 4096;
        const int nb = (int)((n + chunk - 1) / chunk);
        float* partials = nullptr;
        if (hipMallocAsync((void**)&partials, 256 * sizeof(float), (hipStream_t)stream) != hipSuccess || !partials) {
            (void)hipGetLastError();
            hipLaunchKernelGGL(sumsq_accum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, (long long)n, out);   // (same value class, one CU)
        } else {
            hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, (long long)n, chunk, partials);
            hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, out);
            (void)hipFreeAsync(partials, (hipStream_t)stream);
        }
    }
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_clip_coefficient(const float* sumsq, float max_norm, float* coef, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(sumsq && coef && max_norm > 0.f, "clip_coefficient: bad arguments");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sumsq, max_norm, coef);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_scale_by_f32(float* x, int64_t n, const float* coef, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && coef && n > 0, "scale_by_f32: bad arguments");
    hipLaunchKernelGGL(scale_by_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, coef);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int step, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw_step: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (long long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                                int first_step, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(param && grad && n > 0 && (momentum == 0.f || momentum_buf), "sgd_step: bad arguments");
    hipLaunchKernelGGL(sgd_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, momentum_buf, (long long)n,
                       lr, momentum, weight_decay, first_step);
    NPS_LAUNCH_RET();
}
