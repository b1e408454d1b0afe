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

// One workgroup = one TW x th tile of output pixels (a wave = one 64-pixel row segment).  Only the VALID queries (score test
// passed; a compact ascending list built per workgroup) take part: the (few) source rows x columns of the 1/4-resolution
// probability map that the tile's bilinear taps touch are gathered into LDS as [row][col][k] with k = position in the list,
// padded to a multiple of 4 - so the per-pixel loop handles FOUR queries per step with 16-byte LDS reads (4 taps) and two
// broadcast reads (ids, scores).  The kernel is issue-bound (38k small workgroups per 64 images, ~23 valid queries on the
// benchmark inputs), so the bookkeeping is scalar or per-lane: the per-query pixel counts live in lane k of a register
// (flushed once per wave), area / centroid sums come from wave ballots (sum of the set lanes' X = X0 * popcount + sum of set
// bit positions) and are aggregated per workgroup in LDS, then flushed with a few integer global atomics (order independent
// => deterministic).
constexpr int PS_TW = 64;                     // tile height `th` (multiple of 4) is chosen by the host
// LDS head (int32 words): [0,9nq) mirror of the work accumulators, [9nq,9nq+nqp) valid ids, [..+nqp) their scores, then the count
__host__ __device__ inline int ps_pad4(int nq) { return (nq + 3) & ~3; }
__host__ __device__ inline int ps_head_words(int nq) { return ((9 * nq + 3) & ~3) + 2 * ps_pad4(nq) + 4; }

__device__ __forceinline__ int mask_lane_sum(unsigned long long m) {              // sum of the indices of the set bits
    return __popcll(m & 0xAAAAAAAAAAAAAAAAull) + 2 * __popcll(m & 0xCCCCCCCCCCCCCCCCull) + 4 * __popcll(m & 0xF0F0F0F0F0F0F0F0ull) +
           8 * __popcll(m & 0xFF00FF00FF00FF00ull) + 16 * __popcll(m & 0xFFFF0000FFFF0000ull) + 32 * __popcll(m & 0xFFFFFFFF00000000ull);
}

// v_writelane_b32 with a wave-uniform value and lane index (no compiler builtin in this toolchain).  The s_nops cover the
// "VALU writes SGPR -> lane select" and "SALU writes M0 -> use" wait states, which the compiler does not see through inline asm.
__device__ __forceinline__ int ps_writelane(int value, int lane_index, int old) {
    int saved_m0;                                                  // M0 is reserved: saved and restored instead of clobbered
    asm volatile("s_mov_b32 %1, m0\n\ts_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 1\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                 : "+v"(old), "=&s"(saved_m0) : "s"(value), "s"(lane_index));
    return old;
}

// X4 (requires ROWS == 4 and H == 4 h): a wave owns four CONSECUTIVE output rows 4j .. 4j+3; with exact 4x up-sampling these tap only the
// source rows clamp(j-1), clamp(j), clamp(j+1) (rows 4j, 4j+1 the first two, rows 4j+2, 4j+3 the last two; the clamped duplicates at the
// image border carry weight 0 or the same value, so the result is the generic formula's bit for bit): the horizontal blends
// hx * a + lx * b of the three source rows are computed once per query and shared by the four output rows - 6 instead of 16
// 16-byte LDS reads and 21 instead of 36 interpolation instructions per four (pixel, query) pairs.
template <int ROWS, bool X4>
__global__ __launch_bounds__(256) void ps_pixels_kernel(const float* __restrict__ prob, int nq, int h, int w, int H,
                                                        int W, float mask_thr, int src_rows, int src_cols,
                                                        int* __restrict__ work, uint8_t* __restrict__ winner, int planar, int th) {
    extern __shared__ __attribute__((aligned(16))) int sh[];
    const int nqp = ps_pad4(nq), acc_words = (9 * nq + 3) & ~3;
    int* vq = sh + acc_words;                                   // valid query ids (16-byte aligned), padded with the last id
    float* vs = reinterpret_cast<float*>(vq + nqp);             // their scores, padded with -1 (a pad can never win)
    int* nvp = vq + 2 * nqp;
    float* tile = reinterpret_cast<float*>(sh + ps_head_words(nq));
    const int b = blockIdx.z, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int* wk = work + (long long)b * work_words(nq);
    const float sch = (float)h / (float)H, scw = (float)w / (float)W;
    const int X0 = blockIdx.x * PS_TW, Y0 = blockIdx.y * th;
    const int ry0 = min((int)fmaxf(sch * (Y0 + 0.5f) - 0.5f, 0.f), h - 1);
    const int cx0 = min((int)fmaxf(scw * (X0 + 0.5f) - 0.5f, 0.f), w - 1);
    const int nrows = min(src_rows, h - ry0), ncols = min(src_cols, w - cx0);
    // ---- accumulators + compact list of the valid queries (ascending q, with their scores)
    for (int i = threadIdx.x; i < 9 * nq; i += 256) sh[i] = 0;
    if (wave == 0) {
        int off = 0;
        for (int base = 0; base < nq; base += 64) {
            const int q = base + lane, qc = q < nq ? q : nq - 1;
            const int vl = wk[qc], sl = wk[nq + qc];              // both loads unconditional and together: ONE round trip in front of the barrier, not two
            const int v = q < nq ? vl : 0;
            const unsigned long long m = __ballot(v != 0);
            if (v) {
                const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
                vq[pos] = q;
                vs[pos] = __int_as_float(sl);
            }
            off += __popcll(m);
        }
        if (lane == 0) *nvp = off;
    }
    __syncthreads();
    const int nv = *nvp, kp = ps_pad4(nv), nk4 = kp >> 2;
    if (nv == 0) return;                                         // nothing can win: winner map and accumulators stay zero
    if (threadIdx.x < kp - nv) { vq[nv + threadIdx.x] = vq[nv - 1]; vs[nv + threadIdx.x] = -1.f; }
    __syncthreads();
    // ---- gather the source tile: item = (source pixel rc, group of 4 list entries); it / nk4 by multiplication (it < 2^15, nk4 <= 32)
    {
        const unsigned inv = ((1u << 20) + nk4 - 1) / nk4;
        const int items = nrows * ncols * nk4;
        const long long plane = (long long)h * w;
        // FOUR items per pass with all sixteen loads in flight (round 6: the rolled loop - load, wait, ds_write, next - was one memory
        // round trip per 256 items: 3 in a row at ~23 valid queries, 14 at K = 128).  The loads are unconditional (a pass's items behind
        // the last one re-read the last item), only the LDS writes are guarded: a branch around a load makes the compiler wait inside it
        for (int it0 = threadIdx.x; it0 < items; it0 += 4 * 256) {
            float4 v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int it = it0 + u * 256 < items ? it0 + u * 256 : items - 1;
                const int rc = (int)(((unsigned)it * inv) >> 20), k4 = it - rc * nk4;
                const int r = rc / ncols, c = rc - r * ncols;
                const int4 q4 = *reinterpret_cast<const int4*>(vq + 4 * k4);
                const long long pix = (long long)(ry0 + r) * w + cx0 + c;
                if (planar) {
                    const float* pb = prob + (long long)b * nq * plane + pix;
                    v[u] = make_float4(pb[q4.x * plane], pb[q4.y * plane], pb[q4.z * plane], pb[q4.w * plane]);
                } else {
                    const float* pb = prob + ((long long)b * plane + pix) * nq;
                    v[u] = make_float4(pb[q4.x], pb[q4.y], pb[q4.z], pb[q4.w]);
                }
                dst[u] = (r * src_cols + c) * kp + 4 * k4;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (it0 + u * 256 < items) *reinterpret_cast<float4*>(tile + dst[u]) = v[u];
        }
    }
    __syncthreads();
    // Round 4 (end): the query loop is the OUTER loop and the ROWS = th / 4 pixel rows of this wave the inner one, so that the per-query
    // pixel count of the wave is a SCALAR sum of ballot popcounts (s_bcnt1 + s_add) that is dropped into lane k once per query
    // (v_writelane) - it was a v_mov + v_cmp + v_cndmask + v_add and two scalar branches per (pixel, query) pair - and the padded
    // list entries need no validity compare (their lanes are never flushed).  Arithmetic per pixel is unchanged (same operations in
    // the same order): 21 -> 14 VALU instructions per pair.
    const int X = X0 + lane;
    const float sx = fmaxf(scw * (X + 0.5f) - 0.5f, 0.f);
    const int x0 = min((int)sx, w - 1), x1 = min(x0 + 1, w - 1);
    const float lx = sx - x0, hx = 1.f - lx;
    const int c0 = min(max(x0 - cx0, 0), ncols - 1), c1 = min(max(x1 - cx0, 0), ncols - 1);   // clamped: out-of-image lanes read inside the tile
    float hy[ROWS], ly[ROWS], best[ROWS];
    int win[ROWS];
    const float *p00[ROWS], *p01[ROWS], *p10[ROWS], *p11[ROWS];
    unsigned long long inm[ROWS];
    bool in[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int Y = X4 ? Y0 + wave * 4 + r : Y0 + r * 4 + wave;
        in[r] = X < W && Y < H;
        inm[r] = __ballot(in[r]);
        const float sy = fmaxf(sch * (Y + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)sy, h - 1), y1 = min(y0 + 1, h - 1);
        ly[r] = sy - y0; hy[r] = 1.f - ly[r];
        const int r0 = min(max(y0 - ry0, 0), nrows - 1), r1 = min(max(y1 - ry0, 0), nrows - 1);
        p00[r] = tile + (r0 * src_cols + c0) * kp; p01[r] = tile + (r0 * src_cols + c1) * kp;
        p10[r] = tile + (r1 * src_cols + c0) * kp; p11[r] = tile + (r1 * src_cols + c1) * kp;
        best[r] = -INFINITY; win[r] = -1;
    }
    int cnt_lo = 0, cnt_hi = 0;                                  // lane k: pixels of this wave with p >= thr for list entry k (k + 64)
    for (int k4 = 0; k4 < nk4; ++k4) {
        const int4 q4 = *reinterpret_cast<const int4*>(vq + 4 * k4);          // wave-uniform (LDS broadcast)
        const float4 s4 = *reinterpret_cast<const float4*>(vs + 4 * k4);
        const int qv[4] = {q4.x, q4.y, q4.z, q4.w};
        const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
        int np[4] = {0, 0, 0, 0};
        if constexpr (X4) {
            static_assert(!X4 || ROWS == 4, "X4 needs four rows per wave");
            float T[3][4];                                       // horizontal blends of source rows clamp(j-1), clamp(j), clamp(j+1)
#pragma unroll
            for (int sr = 0; sr < 3; ++sr) {
                // rows 0/1 tap (max(j-1, 0), j), rows 2/3 tap (j, min(j+1, h-1)); row j is taken from row 2's FIRST tap: at the top border
                // row 0's second tap is row 1, with weight 0 - replacing it by row j = 0 (weight 0 as well) leaves the value unchanged
                const float* pa = sr == 0 ? p00[0] : (sr == 1 ? p00[2] : p10[2]);
                const float* pb = sr == 0 ? p01[0] : (sr == 1 ? p01[2] : p11[2]);
                const float4 a = *reinterpret_cast<const float4*>(pa + 4 * k4), bq = *reinterpret_cast<const float4*>(pb + 4 * k4);
                const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) T[sr][e] = hx * av[e] + lx * bv[e];
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = hy[r] * T[r >> 1][e] + ly[r] * T[(r >> 1) + 1][e];
                    const float wgt = sv[e] * p;
                    if (wgt > best[r]) { best[r] = wgt; win[r] = qv[e]; }
                    np[e] += __popcll(__builtin_amdgcn_ballot_w64(p >= mask_thr) & inm[r]);
                }
            }
        } else
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const float4 a = *reinterpret_cast<const float4*>(p00[r] + 4 * k4), bq = *reinterpret_cast<const float4*>(p01[r] + 4 * k4);
            const float4 c = *reinterpret_cast<const float4*>(p10[r] + 4 * k4), d = *reinterpret_cast<const float4*>(p11[r] + 4 * k4);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w}, cv[4] = {c.x, c.y, c.z, c.w}, dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float p = hy[r] * (hx * av[e] + lx * bv[e]) + ly[r] * (hx * cv[e] + lx * dv[e]);
                const float wgt = sv[e] * p;
                if (wgt > best[r]) { best[r] = wgt; win[r] = qv[e]; }
                np[e] += __popcll(__builtin_amdgcn_ballot_w64(p >= mask_thr) & inm[r]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {                            // every list entry is visited once: its lane is written, not accumulated
            const int k = 4 * k4 + e;
            if (k4 < 16) cnt_lo = ps_writelane(np[e], k, cnt_lo);
            else cnt_hi = ps_writelane(np[e], k - 64, cnt_hi);
        }
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int Y = X4 ? Y0 + wave * 4 + r : Y0 + r * 4 + wave;
        const bool ok = in[r] && win[r] >= 0;
        const bool pass = ok && best[r] > mask_thr;
        if (ok) winner[((long long)b * H + Y) * W + X] = (uint8_t)(win[r] | (pass ? 0x80 : 0));
        // per distinct winner of the wave (usually one): counts and coordinate sums from the ballots
        unsigned long long rem = __ballot(ok);
        while (rem) {
            const int w0 = __builtin_amdgcn_readlane(win[r], __ffsll((long long)rem) - 1);
            const unsigned long long m = __ballot(ok && win[r] == w0), mp = __ballot(pass && win[r] == w0);
            if (lane == 0) {
                const int cnt = __popcll(m), cp = __popcll(mp);
                atomicAdd(&sh[6 * nq + w0], cnt); atomicAdd(&sh[7 * nq + w0], X0 * cnt + mask_lane_sum(m)); atomicAdd(&sh[8 * nq + w0], Y * cnt);
                if (cp) { atomicAdd(&sh[3 * nq + w0], cp); atomicAdd(&sh[4 * nq + w0], X0 * cp + mask_lane_sum(mp)); atomicAdd(&sh[5 * nq + w0], Y * cp); }
            }
            rem &= ~m;
        }
    }
    if (lane < nv && cnt_lo) atomicAdd(&sh[2 * nq + vq[lane]], cnt_lo);
    if (lane + 64 < nv && cnt_hi) atomicAdd(&sh[2 * nq + vq[lane + 64]], cnt_hi);
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
        double max_overlap = 0.0;   // kept as a double like the Python float (:693-698)
        int max_overlap_q = -1, first_valid = -1;
        for (int q = 0; q < nq; ++q) {
            if (!wk[q]) continue;
            if (first_valid < 0) first_valid = q;
            const int area = wk[3 * nq + q], orig = wk[2 * nq + q];
            if (!zero_flag) {
                if (area < 1 || orig < 1) continue;
                // Python float division, compared in double like the reference (:693-698)
                const double overlap = (double)area / (double)orig;
                if (overlap > max_overlap) { max_overlap = overlap; max_overlap_q = q; }
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
    return nopesac_postselect_planes_ex(cls_logits, mask_prob, params, query_feat, B, nq, D, h, w, H, W, score_thr, mask_thr, overlap_thr,
                                        n_kept, kept_idx, planes, feats, scores, areas, centers, winner, flags, work, 0, stream);
}

extern "C" int nopesac_postselect_planes_ex(const float* cls_logits, const float* mask_prob, const float* params,
                                            const float* query_feat, int B, int nq, int D, int h, int w, int H, int W,
                                            float score_thr, float mask_thr, float overlap_thr, int32_t* n_kept,
                                            int32_t* kept_idx, float* planes, float* feats, float* scores,
                                            int32_t* areas, float* centers, uint8_t* winner, int32_t* flags,
                                            int32_t* work, int prob_planar, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(prob_planar == 0 || prob_planar == 1, "postselect: prob_planar must be 0 ([B,h,w,nq]) or 1 ([B,nq,h,w])");
    NPS_CHECK_ARG(cls_logits && mask_prob && params && query_feat && n_kept && kept_idx && planes && feats && scores &&
                      areas && centers && winner && flags && work, "postselect: null pointer");
    NPS_CHECK_ARG(B > 0 && nq > 0 && nq <= 128 && D > 0 && h > 0 && w > 0 && H > 0 && W > 0, "postselect: bad dims (nq<=128)");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(work, 0, (size_t)B * work_words(nq) * sizeof(int), st);
    if (e != hipSuccess) { set_error("postselect: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(ps_classify_kernel, dim3(B), dim3(64), 0, st, cls_logits, nq, score_thr, work);
    // source extent touched by one PS_TW x th output tile (+2 for the second tap and the start rounding).  With the query loop
    // outside (round 4) th = 16 rows measured best at 480x640 x 64 images, 32 valid queries: 256 us (8 rows 298 us, 32 rows 272 us;
    // the row-inside form it replaced: 347 us at its best height of 8 rows); halved until the source tile fits
    int th = 16, src_rows = 0;
    if (const char* e = getenv("NOPESAC_PS_TH")) { const int v = atoi(e); th = v >= 32 ? 32 : (v >= 16 ? 16 : (v >= 8 ? 8 : 4)); }   // tuning aid: 4 / 8 / 16 / 32
    const int src_cols = (int)(((long long)PS_TW * w + W - 1) / W) + 2;
    size_t lds = 0;
    for (;; th >>= 1) {
        src_rows = (int)(((long long)th * h + H - 1) / H) + 2;
        lds = (size_t)ps_head_words(nq) * sizeof(int) + (size_t)src_rows * src_cols * ps_pad4(nq) * sizeof(float);
        if (lds <= 64 * 1024 || th == 4) break;
    }
    NPS_CHECK_ARG(lds <= 64 * 1024, "postselect: up-sampling ratio too small for the LDS tile (%zu bytes)", lds);
    dim3 grid((W + PS_TW - 1) / PS_TW, (H + th - 1) / th, B);
    auto launch_pixels = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, mask_prob, nq, h, w, H, W, mask_thr, src_rows, src_cols, work, winner,
                           prob_planar, th);
    };
    static const bool x4_off = getenv("NOPESAC_PS_X4") && atoi(getenv("NOPESAC_PS_X4")) == 0;          // A/B aid
    if (th == 4) launch_pixels(ps_pixels_kernel<1, false>);      // th = 4 * ROWS (16 by default; halved above while the tile exceeds 64 KB)
    else if (th == 8) launch_pixels(ps_pixels_kernel<2, false>);
    else if (th == 16 && H == 4 * h && !x4_off) launch_pixels(ps_pixels_kernel<4, true>);   // the architecture's 480 x 640 / 120 x 160
    else if (th == 16) launch_pixels(ps_pixels_kernel<4, false>);
    else launch_pixels(ps_pixels_kernel<8, false>);
    hipLaunchKernelGGL(ps_finalize_kernel, dim3(B), dim3(64), 0, st, params, query_feat, nq, D, H, W, overlap_thr, work,
                       n_kept, kept_idx, planes, feats, scores, areas, centers, winner, flags);
    NPS_LAUNCH_RET();
}
