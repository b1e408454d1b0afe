// Finest level of the PlaneTR top-down path and the per-plane mask head in ONE launch (bf16 mode;
// planeTR_net/planeTR_head.py:148-162 mask einsum, :241-252 top_down):
//
//     p1[px]   = relu(bn(W_c1 . c1[px])) + relu(bilinear_2x(t1)[px])       c1 = res2 map (120x160x256), t1 = up_conv1 output (60x80)
//     mask[px] = sigmoid(M_b . p1[px] + m_b)                               M_b = per-image 50x256 mask weights (pixel-embedding conv
//                                                                          already folded in, see modeling/plane_head.py)
// Un-fused: lateral conv (629 MB in, 629 MB out), bilinear add (629 + 629 MB), mask GEMM (629 MB in): 3.4 GB per step for a
// tensor nobody else reads.  Here a workgroup (8 waves) owns 128 consecutive pixels of one image: the c1 rows are parked in
// LDS, the lateral GEMM streams W_c1 fragment-major from L2 (pwchain.hip), the up-sampling term is added cooperatively
// (16-byte channel chunks, 4 taps from the L2-resident low-resolution map), p1 stays in LDS as the A operand of the mask GEMM,
// and the 128 x 50 probabilities leave as one contiguous 25 KB run.  HBM: 629 MB in + 246 MB out.
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

#ifndef MH_ABLATE
#define MH_ABLATE 0          // tuning builds (NOPESAC_HIPCC_EXTRA=-DMH_ABLATE=n, WRONG results): 1 no bilinear tap loads, 2 no lateral MFMAs,
#endif                       // 3 no probability store, 4 no c1 load, 5 no mask-GEMM MFMAs + sigmoid
constexpr int MH_C = 256, MH_LD = MH_C + 8, MH_BM = 128, MH_RT = MH_BM / 32, MH_NQP_MAX = 128;
constexpr size_t MH_LDS_BYTES = (size_t)(2 * MH_BM * MH_LD);      // ONE 128 x 264 bf16 tile (67.6 KB): c1 rows -> p1 -> f32 output staging
constexpr int MH_RING = 8;                                          // weight fragments in flight per wave (rolling ring)
static_assert((size_t)MH_BM * MH_NQP_MAX * 4 <= 2 * (size_t)MH_BM * MH_LD, "output staging must fit the A tile");

struct MaskHeadArgs {
    const bf16_t* c1; const bf16_t* t1;          // [B][H][W][256], [B][H/2][W/2][256]
    const bf16_t* wc; const float* sc; const float* bc;   // lateral conv (fragment-major) + folded BN
    const bf16_t* mw; const float* mb;           // [B][NQP][256] fragment-major per image (rows >= nq zero), [B][NQP]; NQP = 64 or 128
    float* prob; bf16_t* p1;                     // [B][H][W][nq] f32; optional [B][H][W][256] bf16
    int B, H, W, nq, apply_sigmoid, planar;      // planar: prob is [B][nq][H][W] (one 512-byte run per plane and workgroup)
    int taps1;                                   // tuning aid / test: one pixel per bilinear item (the form of rounds 1-4)
};

// Two workgroups per CU (67.6 KB LDS, <= 128 registers): while one is in its load / bilinear / store phase the other one's MFMAs
// run.  The weights stream through an 8-slot rolling register ring (16 k-steps per GEMM), the single LDS tile is reused in
// place: c1 rows (A of the lateral GEMM) -> p1 (A of the mask GEMM) -> f32 probabilities.
// NQP: planes padded to 64 (2 column tiles x 4 row tiles = one (tile, rows) pair per wave) or 128 (two pairs per wave: column tiles
// wave & 1 and (wave & 1) + 2 of the same 32 rows).
template <int NQP>
__global__ __launch_bounds__(512, 4) void mask_head_kernel(const MaskHeadArgs p) {
    constexpr int NPASS = NQP / 64, NTILES = NQP / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char mh_smem[];
    bf16_t* At = reinterpret_cast<bf16_t*>(mh_smem);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = p.H * p.W;
    // XCD-aware block order: workgroup ids go round-robin over the 8 XCDs, so give every XCD a CONTIGUOUS range of pixel blocks -
    // its L2 then holds only the rows of the low-resolution map t1 that its own blocks tap (with the plain order every XCD pulled
    // the whole map: 1.30 GB read per launch against 0.79 GB of compulsory c1 + t1 bytes)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const long long m0 = (long long)bid * MH_BM;             // per % 128 == 0: a workgroup never straddles two images
    const int b = (int)(m0 / per), pix0 = (int)(m0 % per);

    bf16x8 ring[MH_RING];
    const bf16_t* wlp = p.wc + ((long long)wave * 16 * 64 + lane) * 8;                              // lateral column tile `wave`
    const bf16_t* wmp = p.mw + (((long long)b * NTILES + (wave & 1)) * 16) * 512 + lane * 8;       // mask column tile wave & 1 of image b
#pragma unroll
    for (int s = 0; s < MH_RING; ++s) ring[s] = *reinterpret_cast<const bf16x8*>(wlp + s * 512);
    // c1 rows -> LDS
#pragma unroll
    for (int i = 0; i < MH_BM * 32 / 512; ++i) {
        const int c = tid + i * 512, r = c >> 5, col = (c & 31) * 8;
#if MH_ABLATE != 4
        *reinterpret_cast<us8*>(At + r * MH_LD + col) = *reinterpret_cast<const us8*>(p.c1 + (m0 + r) * MH_C + col);
#endif
    }
    __syncthreads();

    // ---- lateral = relu(bn(W_c1 c1)): wave owns channels wave*32 .. +32 of all four 32-pixel row tiles
    {
        f32x16 acc[MH_RT];
#pragma unroll
        for (int r = 0; r < MH_RT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
            for (int r = 0; r < MH_RT; ++r) {
#if MH_ABLATE != 2
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(At + (r * 32 + l31) * MH_LD + kk * 16 + half * 8);
                acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[kk % MH_RING], af, acc[r], 0, 0, 0);
#endif
            }
            if (kk + MH_RING < 16) ring[kk % MH_RING] = *reinterpret_cast<const bf16x8*>(wlp + (kk + MH_RING) * 512);
            else ring[kk % MH_RING] = *reinterpret_cast<const bf16x8*>(wmp + (kk + MH_RING - 16) * 512);     // mask GEMM k-steps 0..7
            if ((kk & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                      // every wave is done reading the c1 rows
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = wave * 32 + 8 * q + 4 * half;
            const f32x4 s = *reinterpret_cast<const f32x4*>(p.sc + n), bb = *reinterpret_cast<const f32x4*>(p.bc + n);
#pragma unroll
            for (int r = 0; r < MH_RT; ++r) {
                us4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[r][4 * q + e] * s[e];
                    v += bb[e];
                    o[e] = f32_to_bf16(v > 0.f ? v : 0.f);
                }
                *reinterpret_cast<us4*>(At + (r * 32 + l31) * MH_LD + n) = o;
            }
        }
    }
    __syncthreads();

    // ---- p1 = lateral + relu(bilinear_2x(t1)); F.interpolate align_corners=False.
    // Round 5: thread = (FOUR consecutive output pixels, 8-channel chunk).  Under exact 2x up-sampling the pixels 4j .. 4j+3 of an output
    // row tap the source columns 2j-1, 2j, 2j+1, 2j+2 of the same two source rows: 8 loads of 16 bytes per four outputs instead of 16
    // (the tap loads were a quarter of the kernel's time: ablation build MH_ABLATE=1).  Same arithmetic per pixel - the pairs
    // (x0, x1) are the generic formula's columns (or the same DATA at the clamped borders) and lx / ly are the generic weights - so
    // the result is bit-identical.  W % 4 == 0 (a group never straddles two output rows); otherwise one pixel per item as before.
    {
        const int H2 = p.H >> 1, W2 = p.W >> 1;
        const int oh0 = pix0 / p.W, ow0 = pix0 - oh0 * p.W;       // workgroup-uniform
        const bf16_t* tb = p.t1 + (long long)b * H2 * W2 * MH_C;
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        auto unpack = [](unsigned int wd) { return f32x2_t{__uint_as_float(wd << 16), __uint_as_float(wd & 0xffff0000u)}; };
        auto blend_store = [&](int r, int col, const u32x4_t& v00, const u32x4_t& v01, const u32x4_t& v10, const u32x4_t& v11, float lx, float ly) {
            const float hy = 1.f - ly, hx = 1.f - lx;
            const u32x4_t lw = *reinterpret_cast<const u32x4_t*>(At + r * MH_LD + col);
            u32x4_t ow4;
            // two channels per step on float2; a bf16 pair unpacks with one shift and one mask
            const f32x2_t hx2 = {hx, hx}, lx2 = {lx, lx}, hy2 = {hy, hy}, ly2 = {ly, ly};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2_t top = hx2 * unpack(v00[e]) + lx2 * unpack(v01[e]), bot = hx2 * unpack(v10[e]) + lx2 * unpack(v11[e]);
                f32x2_t u = hy2 * top + ly2 * bot;
                u.x = u.x > 0.f ? u.x : 0.f;
                u.y = u.y > 0.f ? u.y : 0.f;
                u = u + unpack(lw[e]);
                ow4[e] = f32x2_to_bf16x2(u.x, u.y);
            }
            const us8 l8 = __builtin_bit_cast(us8, ow4);
            *reinterpret_cast<us8*>(At + r * MH_LD + col) = l8;
            if (p.p1) *reinterpret_cast<us8*>(p.p1 + (m0 + r) * MH_C + col) = l8;
        };
        if ((p.W & 3) == 0 && !p.taps1) {
#pragma unroll 1
            for (int i = 0; i < MH_BM * 8 / 512; ++i) {          // 32 pixel groups x 32 chunks = 1024 items
                const int c = tid + i * 512, gq = c >> 5, col = (c & 31) * 8;
                const int r0 = gq * 4;
                int oh = oh0, ow = ow0 + r0;                     // (pix0 % 4 == 0 and W % 4 == 0: the four pixels share a row)
                while (ow >= p.W) { ow -= p.W; ++oh; }
                const float sy = fmaxf(0.5f * (oh + 0.5f) - 0.5f, 0.f);
                const int y0 = (int)sy, y1 = min(y0 + 1, H2 - 1);
                const float ly = sy - y0;
                const int j2 = ow >> 1;                           // = 2j
                const int ca = max(j2 - 1, 0), cb = j2, cc = min(j2 + 1, W2 - 1), cd = min(j2 + 2, W2 - 1);
                const bf16_t* r0p = tb + (long long)y0 * W2 * MH_C + col;
                const bf16_t* r1p = tb + (long long)y1 * W2 * MH_C + col;
#if MH_ABLATE == 1
                const u32x4_t a0 = {(unsigned)y0, 0u, 0u, 0u}, b0 = {(unsigned)ca, 0u, 0u, 0u}, c0 = {(unsigned)y1, 0u, 0u, 0u}, d0 = a0, a1 = b0, b1 = c0, c1v = a0, d1 = b0;
#else
                const u32x4_t a0 = *reinterpret_cast<const u32x4_t*>(r0p + ca * MH_C), b0 = *reinterpret_cast<const u32x4_t*>(r0p + cb * MH_C);
                const u32x4_t c0 = *reinterpret_cast<const u32x4_t*>(r0p + cc * MH_C), d0 = *reinterpret_cast<const u32x4_t*>(r0p + cd * MH_C);
                const u32x4_t a1 = *reinterpret_cast<const u32x4_t*>(r1p + ca * MH_C), b1 = *reinterpret_cast<const u32x4_t*>(r1p + cb * MH_C);
                const u32x4_t c1v = *reinterpret_cast<const u32x4_t*>(r1p + cc * MH_C), d1 = *reinterpret_cast<const u32x4_t*>(r1p + cd * MH_C);
#endif
                // generic horizontal weights of the four pixels: sx = max(0.5 (ow + 0.5) - 0.5, 0), lx = sx - floor(sx)
                auto lx_of = [](int o) { const float sx = fmaxf(0.5f * (o + 0.5f) - 0.5f, 0.f); return sx - (float)(int)sx; };
                blend_store(r0, col, a0, b0, a1, b1, lx_of(ow), ly);             // x0 = 2j-1 (ow = 0: lx = 0 and a == b's data)
                blend_store(r0 + 1, col, b0, c0, b1, c1v, lx_of(ow + 1), ly);    // x0 = 2j
                blend_store(r0 + 2, col, b0, c0, b1, c1v, lx_of(ow + 2), ly);    // x0 = 2j
                blend_store(r0 + 3, col, c0, d0, c1v, d1, lx_of(ow + 3), ly);    // x0 = 2j+1 (right border: d clamps to c's column)
            }
        } else {
#pragma unroll 2
            for (int i = 0; i < MH_BM * 32 / 512; ++i) {
                const int c = tid + i * 512, r = c >> 5, col = (c & 31) * 8;
                int oh = oh0, ow = ow0 + r;                      // (pix0 + r) / W, % W without a per-item division
                while (ow >= p.W) { ow -= p.W; ++oh; }
                const float sy = fmaxf(0.5f * (oh + 0.5f) - 0.5f, 0.f), sx = fmaxf(0.5f * (ow + 0.5f) - 0.5f, 0.f);
                const int y0 = (int)sy, x0 = (int)sx, y1 = min(y0 + 1, H2 - 1), x1 = min(x0 + 1, W2 - 1);
                const u32x4_t v00 = *reinterpret_cast<const u32x4_t*>(tb + ((long long)y0 * W2 + x0) * MH_C + col);
                const u32x4_t v01 = *reinterpret_cast<const u32x4_t*>(tb + ((long long)y0 * W2 + x1) * MH_C + col);
                const u32x4_t v10 = *reinterpret_cast<const u32x4_t*>(tb + ((long long)y1 * W2 + x0) * MH_C + col);
                const u32x4_t v11 = *reinterpret_cast<const u32x4_t*>(tb + ((long long)y1 * W2 + x1) * MH_C + col);
                blend_store(r, col, v00, v01, v10, v11, sx - x0, sy - y0);
            }
        }
    }
    __syncthreads();

    // ---- mask logits: NQP (padded) planes = NQP/32 column tiles x 4 row tiles = 8 * NPASS (tile, rows) pairs, NPASS per wave
    {
        const int r = wave >> 1;
        f32x16 acc[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ps][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
#if MH_ABLATE != 5
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(At + (r * 32 + l31) * MH_LD + kk * 16 + half * 8);
                acc[ps] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[kk % MH_RING], af, acc[ps], 0, 0, 0);
#endif
                if (kk + MH_RING < 16) ring[kk % MH_RING] = *reinterpret_cast<const bf16x8*>(wmp + ((long long)ps * 32 + kk + MH_RING) * 512);
                else if (ps + 1 < NPASS) ring[kk % MH_RING] = *reinterpret_cast<const bf16x8*>(wmp + ((long long)(ps + 1) * 32 + kk + MH_RING - 16) * 512);
            }
        }
        __syncthreads();                                      // p1 is dead: the tile becomes the [128][nq] f32 staging buffer
        float* St = reinterpret_cast<float*>(At);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int nt = (wave & 1) + 2 * ps;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nt * 32 + 8 * q + 4 * half;
                const f32x4 mbv = *reinterpret_cast<const f32x4*>(p.mb + (long long)b * NQP + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < p.nq) {
                        float v = acc[ps][4 * q + e] + mbv[e];
#if MH_ABLATE != 5
                        if (p.apply_sigmoid) v = 1.f / (1.f + expf(-v));
#endif
                        if (p.planar) St[(n + e) * MH_BM + r * 32 + l31] = v;
                        else St[(r * 32 + l31) * p.nq + n + e] = v;
                    }
                }
            }
        }
    }
    __syncthreads();
    {
        const float* St = reinterpret_cast<const float*>(At);
        float* og = p.prob + m0 * p.nq;                       // 128 * nq floats, contiguous; 16-byte aligned when nq % 2 == 0 (m0 % 128 == 0)
        const int total4 = MH_BM * p.nq / 4;
        if (p.planar) {
            float* ob = p.prob + (long long)b * p.nq * per + pix0;
            for (int i = tid; i < total4; i += 512)
                *reinterpret_cast<f32x4*>(ob + (long long)(i >> 5) * per + (i & 31) * 4) = *reinterpret_cast<const f32x4*>(St + 4 * i);
        } else {
#if MH_ABLATE != 3
            for (int i = tid; i < total4; i += 512) *reinterpret_cast<f32x4*>(og + 4 * i) = *reinterpret_cast<const f32x4*>(St + 4 * i);
#else
            if (tid == 0 && St[0] == 123456.f) og[0] = St[1];
#endif
        }
    }
}

}  // namespace nps

// The mask GEMM's per-image operands from the folded plane embeddings (plane_head.py: `fold` f32 [B * nq][ld], columns 0..255 = mask
// weights, column 256 = mask bias): mw bf16 [B][nqp / 32][16 k-steps][2 halves][32 planes][8] (the MFMA fragment order this kernel
// reads: lane = half * 32 + plane), mb f32 [B][nqp]; planes >= nq are zero.  One launch (was: two zero fills, two strided copies and
// a permuting copy in torch).
namespace nps {
__global__ void mask_operands_kernel(const float* __restrict__ fold, int ld, bf16_t* __restrict__ mw, float* __restrict__ mb, int B, int nq,
                                     int nqp) {
    const long long total = (long long)B * nqp * 32;                        // 8-element groups
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(g & 31), h = (int)((g >> 5) & 1), kk = (int)((g >> 6) & 15);
        const long long bt = g >> 10;                                        // b * (nqp / 32) + t
        const int nt = nqp / 32, b = (int)(bt / nt), t = (int)(bt % nt), q = t * 32 + r;
        us8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        const float* row = fold + ((long long)b * nq + q) * ld;
        if (q < nq) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(row[kk * 16 + h * 8 + e]);
        }
        *reinterpret_cast<us8*>(mw + g * 8) = o;
        if (kk == 0 && h == 0) mb[(long long)b * nqp + q] = q < nq ? row[256] : 0.f;
    }
}
}  // namespace nps

extern "C" int nopesac_mask_operands(const float* fold, int ld, void* mw, float* mb, int B, int nq, int nqp, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(fold && mw && mb && B > 0 && nq > 0 && nq <= nqp && (nqp == 64 || nqp == 128) && ld >= 257, "mask_operands: bad args");
    NPS_CHECK_ARG(((uintptr_t)mw & 15) == 0, "mask_operands: mw must be 16-byte aligned");
    const long long total = (long long)B * nqp * 32;
    hipLaunchKernelGGL(mask_operands_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fold, ld, (bf16_t*)mw, mb, B, nq, nqp);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_mask_head_bf16(const void* c1, const void* t1, const void* w_lateral, const float* scale, const float* bias,
                                      const void* mask_w, const float* mask_b, float* prob, void* p1_out, int B, int H, int W, int nq,
                                      int apply_sigmoid, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(c1 && t1 && w_lateral && scale && bias && mask_w && mask_b && prob, "mask_head: null pointer");
    NPS_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && (H * W) % MH_BM == 0, "mask_head: H*W must be a multiple of 128, H and W even");
    NPS_CHECK_ARG(nq > 0 && nq <= MH_NQP_MAX && nq % 2 == 0, "mask_head: nq must be even and <= 128");
    const void* ptrs[] = {c1, t1, w_lateral, scale, bias, mask_w, mask_b, prob, p1_out};
    for (const void* q : ptrs) NPS_CHECK_ARG(((uintptr_t)q & 15) == 0, "mask_head: pointers must be 16-byte aligned");
    MaskHeadArgs a;
    a.c1 = (const bf16_t*)c1; a.t1 = (const bf16_t*)t1; a.wc = (const bf16_t*)w_lateral; a.sc = scale; a.bc = bias;
    a.mw = (const bf16_t*)mask_w; a.mb = mask_b; a.prob = prob; a.p1 = (bf16_t*)p1_out;
    a.B = B; a.H = H; a.W = W; a.nq = nq; a.apply_sigmoid = apply_sigmoid & 1; a.planar = (apply_sigmoid >> 1) & 1;
    a.taps1 = (apply_sigmoid >> 2) & 1;           // + 4: one pixel per bilinear item (tests: both forms must agree bit for bit)
    const long long blocks = (long long)B * H * W / MH_BM;
    if (nq <= 64) {
        NPS_ENSURE_LDS((int)MH_LDS_BYTES, mask_head_kernel<64>);
        hipLaunchKernelGGL(mask_head_kernel<64>, dim3((unsigned)blocks), dim3(512), MH_LDS_BYTES, (hipStream_t)stream, a);
    } else {
        NPS_ENSURE_LDS((int)MH_LDS_BYTES, mask_head_kernel<128>);
        hipLaunchKernelGGL(mask_head_kernel<128>, dim3((unsigned)blocks), dim3(512), MH_LDS_BYTES, (hipStream_t)stream, a);
    }
    NPS_LAUNCH_RET();
}
