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

// ---------------------------------------------------------------------------------------------
// bf16 MFMA variant (mixed-precision mode): same interface, q/k/v are f32 in memory and are rounded to bf16
// while being staged; scores, softmax statistics and the output accumulate in f32.
//
// One workgroup = 128 query rows of one (batch, head); each wave owns 32 rows.  K ([key][d]) and V^T ([d][key])
// of the (batch, head) are staged once in LDS as bf16.  Per 32-key tile a wave issues
//     S^T = K_tile . Q^T      (2 x v_mfma_f32_32x32x16_bf16; operands swapped so that lane l holds, for query
//                              row l&31, 16 of the 32 scores -> the row max / sum are in-lane reductions plus ONE
//                              exchange with lane l^32)
//     online softmax (running max / sum, rescale of the O accumulator)
//     P -> bf16 pairs, 4 x v_permlane32_swap to turn the accumulator layout into the B-operand layout
//     O^T += V^T_tile . P^T   (2 x MFMA)
// so no score matrix ever touches LDS or HBM.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    return f32x2_to_bf16x2(lo, hi);
}

constexpr int ATT_KSTRIDE = 40;   // bf16 elements per K row in LDS (32 + 8 pad = 80 bytes: conflict-free b128 reads)

// TIO = float: f32 q/k/v/o (operands rounded to bf16 while staged); TIO = bf16_t: q/k/v/o are bf16 in memory (the GEMMs
// that produce / consume them round to bf16 anyway, so this only halves the traffic).
template <typename TIO>
__device__ __forceinline__ void att_load8(const TIO* p, float (&t)[8]);
template <>
__device__ __forceinline__ void att_load8<float>(const float* p, float (&t)[8]) {
    *(f32x4*)(t) = *(const f32x4*)(p);
    *(f32x4*)(t + 4) = *(const f32x4*)(p + 4);
}
template <>
__device__ __forceinline__ void att_load8<bf16_t>(const bf16_t* p, float (&t)[8]) {
    const u32x4 r = *(const u32x4*)(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        t[2 * e] = __uint_as_float(r[e] << 16);
        t[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u);
    }
}

// NW = waves per workgroup (32 query rows each).  4 for short query sets; 10 for 257..320 rows (the encoder's 15 x 20 = 300 tokens): ONE
// workgroup per (batch, head) stages K / V once instead of three times (the staging was half of a workgroup's time) and no
// wave slot idles on a 44-row remainder block.
template <typename TIO, int NW>
__global__ __launch_bounds__(NW * 64) void attention_mfma_kernel(
    const TIO* __restrict__ q, long long q_stride, const TIO* __restrict__ k, long long k_stride,
    const TIO* __restrict__ v, long long v_stride, TIO* __restrict__ o, long long o_stride, int Lq, int Lk,
    int Lk_pad, float scale, const int* __restrict__ qlen, const int* __restrict__ klen) {
    extern __shared__ __attribute__((aligned(16))) unsigned char att_smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(att_smem);                    // [Lk_pad][ATT_KSTRIDE]
    const int vt_stride = Lk_pad + 8;                                    // bf16 elements per V^T row
    bf16_t* Vt = Ks + Lk_pad * ATT_KSTRIDE;                              // [32][vt_stride]
    const int b = blockIdx.z, h = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = qlen ? min(qlen[b], Lq) : Lq;
    const int nk = klen ? min(klen[b], Lk) : Lk;
    // ---- this wave's 32 query rows; lane l: row = l&31, k-half = l>>5 (requested first: in flight while K / V are staged)
    constexpr int NT = NW * 64;
    const int row = blockIdx.x * (NW * 32) + wave * 32 + (lane & 31);
    const bool row_ok = row < nq;
    const int half = lane >> 5;
    float tq[2][8];
    {
        const TIO* qp = q + ((long long)b * Lq + (row_ok ? row : 0)) * q_stride + h * HD;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) att_load8<TIO>(qp + ks * 16 + half * 8, tq[ks]);
    }
    // ---- stage K (row-major) and V^T as bf16; rows >= nk are zero
    const TIO* kb = k + (long long)b * Lk * k_stride + h * HD;
    const TIO* vb = v + (long long)b * Lk * v_stride + h * HD;
    if constexpr (sizeof(TIO) == 2) {
        // bf16 in memory: the 16-byte pieces of up to five passes are requested back to back (unconditional, row clamped) and only then
        // consumed - one memory round trip for the whole K / V of a (batch, head) instead of one per pass (300 keys = 5 passes: the
        // staging was ~10 us of the kernel's 15-20 us per workgroup)
        constexpr int UB = NW == 4 ? 5 : 2;
        const int nkc = max(nk, 1) - 1;
        for (int base = 0; base < Lk_pad * 4; base += NT * UB) {
            u32x4 kr[UB], vr[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * NT + tid, j = min(idx >> 2, nkc), c = (idx & 3) * 8;
                kr[u] = *(const u32x4*)(kb + (long long)j * k_stride + c);
                vr[u] = *(const u32x4*)(vb + (long long)j * v_stride + c);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * NT + tid, j = idx >> 2, c = (idx & 3) * 8;
                if (idx >= Lk_pad * 4) continue;
                const bool live = j < nk;
                u32x4 pk = kr[u];
                if (!live) pk = u32x4{0u, 0u, 0u, 0u};
                *(u32x4*)(Ks + j * ATT_KSTRIDE + c) = pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned int w2 = live ? vr[u][e] : 0u;
                    Vt[(c + 2 * e) * vt_stride + j] = (bf16_t)(w2 & 0xffffu);
                    Vt[(c + 2 * e + 1) * vt_stride + j] = (bf16_t)(w2 >> 16);
                }
            }
        }
    } else
    for (int idx = tid; idx < Lk_pad * 4; idx += NT) {                   // 8 floats per chunk
        const int j = idx >> 2, c = (idx & 3) * 8;
        float kv[8], vv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { kv[e] = 0.f; vv[e] = 0.f; }
        if (j < nk) {
            att_load8<TIO>(kb + (long long)j * k_stride + c, kv);
            att_load8<TIO>(vb + (long long)j * v_stride + c, vv);
        }
        u32x4 pk = {pack_bf16x2(kv[0], kv[1]), pack_bf16x2(kv[2], kv[3]), pack_bf16x2(kv[4], kv[5]), pack_bf16x2(kv[6], kv[7])};
        *(u32x4*)(Ks + j * ATT_KSTRIDE + c) = pk;
#pragma unroll
        for (int e = 0; e < 8; ++e) Vt[(c + e) * vt_stride + j] = f32_to_bf16(vv[e]);
    }
    __syncthreads();
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        // scores are kept in the log2 domain (scale * log2(e) folded into q): the soft-max then needs bare v_exp_f32's
        const float sc2 = scale * 1.4426950408889634f;
        const float* t = tq[ks];
        u32x4 pk = {pack_bf16x2(t[0] * sc2, t[1] * sc2), pack_bf16x2(t[2] * sc2, t[3] * sc2),
                    pack_bf16x2(t[4] * sc2, t[5] * sc2), pack_bf16x2(t[6] * sc2, t[7] * sc2)};
        qf[ks] = __builtin_bit_cast(bf16x8, pk);
    }
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int ntiles = (nk + 31) / 32;
    for (int kt = 0; kt < ntiles; ++kt) {
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 kf = *(const bf16x8*)(Ks + (kt * 32 + (lane & 31)) * ATT_KSTRIDE + ks * 16 + half * 8);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        // s[r] = score(query row, key kt*32 + (r&3) + 8*(r>>2) + 4*half)
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= nk) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // 0 on the first tile (m_run = -inf)
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - m_new); ps += s[r]; }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
        // accumulator layout -> B-operand layout (see header comment): pairs d[i] = keys (2i&3 .. ) of group i>>1
        unsigned int d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = pack_bf16x2(s[2 * i], s[2 * i + 1]);
#pragma unroll
        for (int g = 0; g < 2; ++g) {                      // k-step g: keys 16g .. 16g+15 of the tile
#if __has_builtin(__builtin_amdgcn_permlane32_swap)
            auto r0 = __builtin_amdgcn_permlane32_swap(d[4 * g + 0], d[4 * g + 2], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(d[4 * g + 1], d[4 * g + 3], false, false);
            u32x4 pk = {r0[0], r1[0], r0[1], r1[1]};
#else
            // portable fallback: exchange through ds_bpermute-style shuffles
            const unsigned int a0 = d[4 * g + 0], a1 = d[4 * g + 1], a2 = d[4 * g + 2], a3 = d[4 * g + 3];
            const unsigned int x0 = __shfl_xor(half ? a0 : a2, 32, 64), x1 = __shfl_xor(half ? a1 : a3, 32, 64);
            u32x4 pk = half ? u32x4{x0, x1, a2, a3} : u32x4{a0, a1, x0, x1};
#endif
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
            const bf16x8 vf = *(const bf16x8*)(Vt + (lane & 31) * vt_stride + kt * 32 + g * 16 + half * 8);
            oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc, 0, 0, 0);
        }
    }
    // oacc[r] = O(query row, d = (r&3) + 8*(r>>2) + 4*half)
    if (row < Lq) {
        TIO* op = o + ((long long)b * Lq + row) * o_stride + h * HD;
        const float inv = (row_ok && nk > 0) ? 1.f / l_run : 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 ov = {oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv};
            if (!(row_ok && nk > 0)) ov = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (sizeof(TIO) == 4) {
                *(f32x4*)(op + 8 * g + 4 * half) = ov;
            } else {
                uint2 pk = make_uint2(pack_bf16x2(ov[0], ov[1]), pack_bf16x2(ov[2], ov[3]));
                *(uint2*)(op + 8 * g + 4 * half) = pk;
            }
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

template <typename TIO>
static int launch_attention_mfma(const TIO* q, int64_t q_stride, const TIO* k, int64_t k_stride, const TIO* v, int64_t v_stride, TIO* o,
                                 int64_t o_stride, int B, int Lq, int Lk, int heads, float scale, const int32_t* qlen,
                                 const int32_t* klen, void* stream) {
    using namespace nps;
    constexpr int AL = 16 / sizeof(TIO);      // elements per 16 bytes
    NPS_CHECK_ARG(q && k && v && o, "attention_bf16: null pointer");
    NPS_CHECK_ARG(B > 0 && Lq > 0 && Lk > 0 && Lk <= 512 && heads > 0, "attention_bf16: bad dims B=%d Lq=%d Lk=%d heads=%d", B, Lq, Lk, heads);
    NPS_CHECK_ARG(q_stride >= heads * HD && k_stride >= heads * HD && v_stride >= heads * HD && o_stride >= heads * HD,
                  "attention_bf16: row stride smaller than heads*32");
    NPS_CHECK_ARG(q_stride % AL == 0 && k_stride % AL == 0 && v_stride % AL == 0 && o_stride % AL == 0 &&
                      (uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)v % 16 == 0 && (uintptr_t)o % 16 == 0,
                  "attention_bf16: rows must be 16-byte aligned");
    const int Lk_pad = ((Lk + 31) / 32) * 32;
    const size_t lds = (size_t)Lk_pad * ATT_KSTRIDE * 2 + (size_t)32 * (Lk_pad + 8) * 2;
    static const bool wide_off = getenv("NOPESAC_ATT_WIDE") && atoi(getenv("NOPESAC_ATT_WIDE")) == 0;   // A/B aid
    if (Lq > 256 && Lq <= 320 && !wide_off) {
        NPS_ENSURE_LDS(96 * 1024, (attention_mfma_kernel<TIO, 10>));
        hipLaunchKernelGGL((attention_mfma_kernel<TIO, 10>), dim3(1, heads, B), dim3(640), lds, (hipStream_t)stream, q, (long long)q_stride, k,
                           (long long)k_stride, v, (long long)v_stride, o, (long long)o_stride, Lq, Lk, Lk_pad, scale, qlen, klen);
    } else {
        NPS_ENSURE_LDS(96 * 1024, (attention_mfma_kernel<TIO, 4>));
        hipLaunchKernelGGL((attention_mfma_kernel<TIO, 4>), dim3((Lq + 127) / 128, heads, B), dim3(256), lds, (hipStream_t)stream, q,
                           (long long)q_stride, k, (long long)k_stride, v, (long long)v_stride, o, (long long)o_stride, Lq, Lk, Lk_pad,
                           scale, qlen, klen);
    }
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_attention_small_bf16(const float* q, int64_t q_stride, const float* k, int64_t k_stride,
                                            const float* v, int64_t v_stride, float* o, int64_t o_stride, int B, int Lq,
                                            int Lk, int heads, float scale, const int32_t* qlen, const int32_t* klen,
                                            void* stream) {
    return launch_attention_mfma<float>(q, q_stride, k, k_stride, v, v_stride, o, o_stride, B, Lq, Lk, heads, scale, qlen, klen, stream);
}

extern "C" int nopesac_attention_small_bf16io(const void* q, int64_t q_stride, const void* k, int64_t k_stride, const void* v,
                                              int64_t v_stride, void* o, int64_t o_stride, int B, int Lq, int Lk, int heads,
                                              float scale, const int32_t* qlen, const int32_t* klen, void* stream) {
    using nps::bf16_t;
    return launch_attention_mfma<bf16_t>((const bf16_t*)q, q_stride, (const bf16_t*)k, k_stride, (const bf16_t*)v, v_stride, (bf16_t*)o,
                                         o_stride, B, Lq, Lk, heads, scale, qlen, klen, stream);
}
