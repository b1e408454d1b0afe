// Fused plane post-selection (meta_arch/siamese_planeTR.py:625-803).
//
// The reference materialises sigmoid(mask logits) up-sampled to 50 x 480 x 640 fp32 (61 MB / image) and
// then loops over planes in Python with a device->host sync per plane.  Here the 1/4-resolution
// probability map [h,w,nq] (3.8 MB) is read once per output pixel neighbourhood and never up-sampled in
// memory; areas / centroids are integer atomics (order independent => deterministic); the per-plane
// decisions run in one wave per image, in query order, exactly as the Python loop does.
//   k1 classify : softmax over {plane, non-plane}, score/label test, arg-max fallback      (:652-661)
//   k2 pixels   : per output pixel bilinear taps, arg-max_q(score_q * prob_q), area counts  (:648,667-689)
//   k3 finalize : overlap rule, max-overlap fallback, centroids, gather of kept planes     (:690-800)
#include "common.h"

namespace nps {

// work layout per image (int32 words), NQ = nq
//  [0,NQ) valid  [NQ,2NQ) score bits  [2NQ,3NQ) orig_area  [3NQ,4NQ) area_pass [4NQ,5NQ) sx_pass
//  [5NQ,6NQ) sy_pass [6NQ,7NQ) area_all [7NQ,8NQ) sx_all [8NQ,9NQ) sy_all [9NQ] zero_flag
__host__ __device__ inline int work_words(int nq) { return 9 * nq + 8; }

__global__ __launch_bounds__(64) void ps_classify_kernel(const float* __restrict__ logits, int nq, float score_thr,
                                                         int* __restrict__ work) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int* wk = work + (long long)b * work_words(nq);
    const float* lg = logits + (long long)b * nq * 2;
    int any = 0;
    float best_p0 = -1.f;
    int best_q = 0x7fffffff;
    for (int q = lane; q < nq; q += 64) {
        const float l0 = lg[2 * q], l1 = lg[2 * q + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        const int label = p1 > p0 ? 1 : 0;          // torch.max returns the first maximal index
        const float score = label ? p1 : p0;
        const int valid = (label == 0) && (score > score_thr);
        wk[q] = valid;
        wk[nq + q] = __float_as_int(score);
        any |= valid;
        if (p0 > best_p0) { best_p0 = p0; best_q = q; }   // first max within this lane's ascending q
    }
    any = __any(any);
    if (!any) {
        // arg-max over queries of p0, first occurrence
        for (int o = 32; o > 0; o >>= 1) {
            const float op = __shfl_xor(best_p0, o, 64);
            const int oq = __shfl_xor(best_q, o, 64);
            if (op > best_p0 || (op == best_p0 && oq < best_q)) { best_p0 = op; best_q = oq; }
        }
        if (lane == 0) {
            wk[best_q] = 1;
            wk[nq + best_q] = __float_as_int(best_p0);
        }
    }
    if (lane == 0) wk[9 * nq] = any ? 0 : 1;
}

__device__ __forceinline__ int wave_isum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One workgroup = one TW x TH tile of output pixels.  The (few) source rows x columns of the 1/4-resolution
// probability map that the tile's bilinear taps touch are first copied into LDS with coalesced loads ([row][col][q],
// q contiguous as in memory); the per-pixel loop over the valid queries then reads LDS only (bank = (nq*x + q) % 32:
// distinct for the 16 source columns a wave touches).  Area / centroid sums are aggregated per wave, then per
// workgroup in LDS, then flushed with a few integer global atomics.
constexpr int PS_TW = 64, PS_TH = 8;

__global__ __launch_bounds__(256) void ps_pixels_kernel(const float* __restrict__ prob, int nq, int h, int w, int H,
                                                        int W, float mask_thr, int src_rows, int src_cols,
                                                        int* __restrict__ work, uint8_t* __restrict__ winner) {
    extern __shared__ int sh[];   // [0,nq) valid, [nq,2nq) score, 7*nq accumulators, then the f32 source tile
    float* tile = reinterpret_cast<float*>(sh + 9 * nq);
    const int b = blockIdx.z;
    int* wk = work + (long long)b * work_words(nq);
    for (int i = threadIdx.x; i < 9 * nq; i += 256) sh[i] = i < 2 * nq ? wk[i] : 0;
    const float sch = (float)h / (float)H, scw = (float)w / (float)W;
    const int X0 = blockIdx.x * PS_TW, Y0 = blockIdx.y * PS_TH;
    const int ry0 = min((int)fmaxf(sch * (Y0 + 0.5f) - 0.5f, 0.f), h - 1);
    const int cx0 = min((int)fmaxf(scw * (X0 + 0.5f) - 0.5f, 0.f), w - 1);
    const int nrows = min(src_rows, h - ry0), ncols = min(src_cols, w - cx0);
    const float* pb = prob + (long long)b * h * w * nq;
    for (int r = 0; r < nrows; ++r) {
        const float* srow = pb + ((long long)(ry0 + r) * w + cx0) * nq;
        for (int i = threadIdx.x; i < ncols * nq; i += 256) tile[r * src_cols * nq + i] = srow[i];
    }
    __syncthreads();
    for (int it = 0; it < PS_TW * PS_TH / 256; ++it) {
        const int lp = it * 256 + threadIdx.x;
        const int X = X0 + (lp % PS_TW), Y = Y0 + (lp / PS_TW);
        const bool in = X < W && Y < H;
        const float sy = fmaxf(sch * (Y + 0.5f) - 0.5f, 0.f), sx = fmaxf(scw * (X + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)sy, h - 1), x0 = min((int)sx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
        // clamp the local indices so that out-of-image lanes still read inside the tile
        const int r0 = min(max(y0 - ry0, 0), nrows - 1), r1 = min(max(y1 - ry0, 0), nrows - 1);
        const int c0 = min(max(x0 - cx0, 0), ncols - 1), c1 = min(max(x1 - cx0, 0), ncols - 1);
        const float* p00 = tile + (r0 * src_cols + c0) * nq;
        const float* p01 = tile + (r0 * src_cols + c1) * nq;
        const float* p10 = tile + (r1 * src_cols + c0) * nq;
        const float* p11 = tile + (r1 * src_cols + c1) * nq;
        float best = -INFINITY;
        int win = -1;
        for (int q = 0; q < nq; ++q) {
            if (!sh[q]) continue;   // block-uniform
            float p = 0.f;
            if (in) p = hy * (hx * p00[q] + lx * p01[q]) + ly * (hx * p10[q] + lx * p11[q]);
            const float wgt = __int_as_float(sh[nq + q]) * p;
            if (in && wgt > best) { best = wgt; win = q; }
            const unsigned long long bal = __ballot(in && p >= mask_thr);
            if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&sh[2 * nq + q], __popcll(bal));
        }
        const int pass = in && win >= 0 && best > mask_thr;
        if (in && win >= 0) winner[((long long)b * H + Y) * W + X] = (uint8_t)(win | (pass ? 0x80 : 0));
        // wave-level aggregation: in the common case every lane of the wave has the same winner
        const int w0 = __builtin_amdgcn_readfirstlane(win);
        if (__all(win == w0 || !in)) {
            if (w0 >= 0) {
                const int ok = in && win >= 0;
                const int cnt = wave_isum(ok), sxs = wave_isum(ok ? X : 0), sys = wave_isum(ok ? Y : 0);
                const int cp = wave_isum(pass), sxp = wave_isum(pass ? X : 0), syp = wave_isum(pass ? Y : 0);
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(&sh[6 * nq + w0], cnt); atomicAdd(&sh[7 * nq + w0], sxs); atomicAdd(&sh[8 * nq + w0], sys);
                    if (cp) { atomicAdd(&sh[3 * nq + w0], cp); atomicAdd(&sh[4 * nq + w0], sxp); atomicAdd(&sh[5 * nq + w0], syp); }
                }
            }
        } else if (in && win >= 0) {
            atomicAdd(&sh[6 * nq + win], 1);
            atomicAdd(&sh[7 * nq + win], X);
            atomicAdd(&sh[8 * nq + win], Y);
            if (pass) {
                atomicAdd(&sh[3 * nq + win], 1);
                atomicAdd(&sh[4 * nq + win], X);
                atomicAdd(&sh[5 * nq + win], Y);
            }
        }
    }
    __syncthreads();
    for (int i = 2 * nq + threadIdx.x; i < 9 * nq; i += 256)
        if (sh[i]) atomicAdd(&wk[i], sh[i]);
}

__global__ __launch_bounds__(64) void ps_finalize_kernel(const float* __restrict__ params, const float* __restrict__ feat,
                                                         int nq, int D, int H, int W, float overlap_thr,
                                                         const int* __restrict__ work, int* __restrict__ n_kept,
                                                         int* __restrict__ kept_idx, float* __restrict__ planes,
                                                         float* __restrict__ feats, float* __restrict__ scores,
                                                         int* __restrict__ areas, float* __restrict__ centers,
                                                         uint8_t* __restrict__ winner, int* __restrict__ flags) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int* wk = work + (long long)b * work_words(nq);
    __shared__ int s_keep[128];
    __shared__ int s_n, s_fallback;
    const int zero_flag = wk[9 * nq];
    if (lane == 0) {
        int n = 0, fb = 0;
        float max_overlap = 0.f;
        int max_overlap_q = -1, first_valid = -1;
        for (int q = 0; q < nq; ++q) {
            if (!wk[q]) continue;
            if (first_valid < 0) first_valid = q;
            const int area = wk[3 * nq + q], orig = wk[2 * nq + q];
            if (!zero_flag) {
                if (area < 1 || orig < 1) continue;
                // Python float division, compared in double like the reference (:693-698)
                const double overlap = (double)area / (double)orig;
                if (overlap > (double)max_overlap) { max_overlap = (float)overlap; max_overlap_q = q; }
                if (overlap < (double)overlap_thr) continue;
            }
            s_keep[n++] = q;
        }
        if (n == 0) {   // :741-788 (only reachable when zero_flag is false)
            s_keep[0] = max_overlap_q >= 0 ? max_overlap_q : first_valid;
            n = 1;
            fb = 1;
        }
        s_n = n;
        s_fallback = fb;
        n_kept[b] = n;
        flags[b] = (zero_flag ? 1 : 0) | (fb ? 2 : 0);
    }
    __syncthreads();
    const int n = s_n, fb = s_fallback;
    for (int i = lane; i < nq; i += 64) {
        const long long o = (long long)b * nq + i;
        if (i < n) {
            const int q = s_keep[i];
            int area, sxi, syi;
            double cx, cy;
            if (fb) {
                area = wk[6 * nq + q]; sxi = wk[7 * nq + q]; syi = wk[8 * nq + q];
                cx = ((double)sxi / W) / (double)area;
                cy = ((double)syi / H) / (double)area;
            } else {
                area = wk[3 * nq + q]; sxi = wk[4 * nq + q]; syi = wk[5 * nq + q];
                if (zero_flag && area == 0) {       // :700-702 plane_mask[0,0] = 1
                    area = 1; sxi = 0; syi = 0;
                    winner[(long long)b * H * W] = (uint8_t)(q | 0x80);
                }
                cx = ((double)sxi / W) / ((double)area + 1e-10);
                cy = ((double)syi / H) / ((double)area + 1e-10);
            }
            kept_idx[o] = q;
            areas[o] = area;
            scores[o] = __int_as_float(wk[nq + q]);
            centers[2 * o] = (float)cx;
            centers[2 * o + 1] = (float)cy;
            for (int d = 0; d < 3; ++d) planes[3 * o + d] = params[((long long)b * nq + q) * 3 + d];
        } else {
            kept_idx[o] = -1; areas[o] = 0; scores[o] = 0.f; centers[2 * o] = 0.f; centers[2 * o + 1] = 0.f;
            for (int d = 0; d < 3; ++d) planes[3 * o + d] = 0.f;
        }
    }
    for (int i = 0; i < nq; ++i) {
        const int q = i < n ? s_keep[i] : -1;
        for (int d = lane; d < D; d += 64)
            feats[((long long)b * nq + i) * D + d] = q >= 0 ? feat[((long long)b * nq + q) * D + d] : 0.f;
    }
}

}  // namespace nps

extern "C" int nopesac_postselect_planes(const float* cls_logits, const float* mask_prob, const float* params,
                                         const float* query_feat, int B, int nq, int D, int h, int w, int H, int W,
                                         float score_thr, float mask_thr, float overlap_thr, int32_t* n_kept,
                                         int32_t* kept_idx, float* planes, float* feats, float* scores,
                                         int32_t* areas, float* centers, uint8_t* winner, int32_t* flags,
                                         int32_t* work, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(cls_logits && mask_prob && params && query_feat && n_kept && kept_idx && planes && feats && scores &&
                      areas && centers && winner && flags && work, "postselect: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128 && D > 0 && h > 0 && w > 0 && H > 0 && W > 0, "postselect: bad dims (nq<=128)");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(work, 0, (size_t)B * work_words(nq) * sizeof(int), st);
    if (e != hipSuccess) { set_error("postselect: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(ps_classify_kernel, dim3(B), dim3(64), 0, st, cls_logits, nq, score_thr, work);
    // source extent touched by one PS_TW x PS_TH output tile (+2 for the second tap and the start rounding)
    const int src_rows = (int)(((long long)PS_TH * h + H - 1) / H) + 2, src_cols = (int)(((long long)PS_TW * w + W - 1) / W) + 2;
    const size_t lds = (size_t)9 * nq * sizeof(int) + (size_t)src_rows * src_cols * nq * sizeof(float);
    NPS_CHECK_ARG(lds <= 64 * 1024, "postselect: up-sampling ratio too small for the LDS tile (%zu bytes)", lds);
    dim3 grid((W + PS_TW - 1) / PS_TW, (H + PS_TH - 1) / PS_TH, B);
    hipLaunchKernelGGL(ps_pixels_kernel, grid, dim3(256), lds, st, mask_prob, nq, h, w, H, W, mask_thr, src_rows, src_cols,
                       work, winner);
    hipLaunchKernelGGL(ps_finalize_kernel, dim3(B), dim3(64), 0, st, params, query_feat, nq, D, H, W, overlap_thr, work,
                       n_kept, kept_idx, planes, feats, scores, areas, centers, winner, flags);
    NPS_LAUNCH_RET();
}
