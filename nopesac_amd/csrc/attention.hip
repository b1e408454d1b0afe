// Multi-head softmax attention for SHORT sequences (<= 512 keys, head dim 32): the DETR encoder/decoder
// of the plane head (300 tokens / 50 queries) and the matcher GNN (<= nq planes).
//
// Not a flash-attention port: at these sizes K/V of one (batch, head) are a few KB and stay in L1/L2, so
// there is no tiling over keys through LDS.  A workgroup = 64 query rows x 4 key splits:
//   * lane  = query row (its q[32], o[32] live in VGPRs),
//   * wave  = key split; inside a wave the K/V row address is wave-uniform, so hipcc emits scalar
//     (s_load) loads and the dot products are v_fma with an SGPR operand — no LDS traffic, no bank
//     conflicts;
//   * two passes (row max, then exp/accumulate) give the max-subtracted softmax torch computes; the 4
//     splits exchange (max, sum, o) through 34 KB of LDS.
#include "common.h"

namespace nps {

constexpr int HD = 32;

__global__ __launch_bounds__(256) void attention_small_kernel(
    const float* __restrict__ q, long long q_stride, const float* __restrict__ k, long long k_stride,
    const float* __restrict__ v, long long v_stride, float* __restrict__ o, long long o_stride, int Lq, int Lk,
    float scale, const int* __restrict__ qlen, const int* __restrict__ klen) {
    __shared__ float sh_m[4][64];
    __shared__ float sh_l[4][64];
    __shared__ float sh_o[4][HD][64 + 1];
    const int b = blockIdx.z, h = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int split = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x * 64 + lane;
    const int nq = qlen ? min(qlen[b], Lq) : Lq;
    const int nk = klen ? min(klen[b], Lk) : Lk;
    const bool row_ok = row < nq;

    float qr[HD];
    {
        const float* qp = q + ((long long)b * Lq + (row_ok ? row : 0)) * q_stride + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) qr[d] = qp[d] * scale;
    }
    const float* kb = k + (long long)b * Lk * k_stride + h * HD;
    const float* vb = v + (long long)b * Lk * v_stride + h * HD;
    // keys of this split: j = split, split+4, ...
    float m = -INFINITY;
    for (int j = split; j < nk; j += 4) {
        const float* kr = kb + (long long)j * k_stride;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(qr[d], kr[d], s);
        m = fmaxf(m, s);
    }
    sh_m[split][lane] = m;
    __syncthreads();
    const float M = fmaxf(fmaxf(sh_m[0][lane], sh_m[1][lane]), fmaxf(sh_m[2][lane], sh_m[3][lane]));
    float l = 0.f;
    float acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int j = split; j < nk; j += 4) {
        const float* kr = kb + (long long)j * k_stride;
        const float* vr = vb + (long long)j * v_stride;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(qr[d], kr[d], s);
        const float p = expf(s - M);
        l += p;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, vr[d], acc[d]);
    }
    sh_l[split][lane] = l;
#pragma unroll
    for (int d = 0; d < HD; ++d) sh_o[split][d][lane] = acc[d];
    __syncthreads();
    // combine: thread (split, lane) finalises dims d = split*8 .. +8 of row `lane`
    const float L = sh_l[0][lane] + sh_l[1][lane] + sh_l[2][lane] + sh_l[3][lane];
    if (row < Lq) {
        float* op = o + ((long long)b * Lq + row) * o_stride + h * HD;
#pragma unroll
        for (int dd = 0; dd < HD / 4; ++dd) {
            const int d = split * (HD / 4) + dd;
            const float s = sh_o[0][d][lane] + sh_o[1][d][lane] + sh_o[2][d][lane] + sh_o[3][d][lane];
            op[d] = (row_ok && nk > 0) ? s / L : 0.f;
        }
    }
}

}  // namespace nps

extern "C" int nopesac_attention_small(const float* q, int64_t q_stride, const float* k, int64_t k_stride,
                                       const float* v, int64_t v_stride, float* o, int64_t o_stride, int B, int Lq,
                                       int Lk, int heads, float scale, const int32_t* qlen, const int32_t* klen,
                                       void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(q && k && v && o, "attention: null pointer");
    NPS_CHECK_ARG(B > 0 && Lq > 0 && Lk > 0 && Lk <= 512 && heads > 0, "attention: bad dims B=%d Lq=%d Lk=%d heads=%d", B, Lq, Lk, heads);
    NPS_CHECK_ARG(q_stride >= heads * HD && k_stride >= heads * HD && v_stride >= heads * HD && o_stride >= heads * HD,
                  "attention: row stride smaller than heads*32");
    dim3 grid((Lq + 63) / 64, heads, B);
    hipLaunchKernelGGL(attention_small_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, (long long)q_stride, k,
                       (long long)k_stride, v, (long long)v_stride, o, (long long)o_stride, Lq, Lk, scale, qlen, klen);
    NPS_LAUNCH_RET();
}
