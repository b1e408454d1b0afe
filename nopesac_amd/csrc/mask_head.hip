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
#include <type_traits>

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

// sigmoid on the hardware transcendentals (v_exp_f32 + v_rcp_f32, ~1 ulp each): 5 instructions instead of the ~25 of expf + an IEEE
// division - the bias + sigmoid + staging phase was 5.3 k of the workgroup's 40 k cycles (in-kernel stamps, profiles/r6_h_*)
__device__ __forceinline__ float mh_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }

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
// STAMP: tuning build - cycle stamps of every (workgroup, wave) at the phase boundaries into dbg[workgroup][8 waves][16] (scripts/mask_head_stamps.py)
template <int NQP, bool STAMP = false>
__global__ __launch_bounds__(512, 4) void mask_head_kernel(const MaskHeadArgs p, unsigned long long* dbg = nullptr) {
    constexpr int NPASS = NQP / 64, NTILES = NQP / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char mh_smem[];
    bf16_t* At = reinterpret_cast<bf16_t*>(mh_smem);
    // round 6: the lateral conv's BN scale / shift (256 + 256 floats) and the image's mask biases (NQP floats) parked behind the tile:
    // as global loads inside the two epilogues they were exposed L2 round trips behind every older load of the wave
    float* SB = reinterpret_cast<float*>(mh_smem + MH_LDS_BYTES);
    unsigned long long ts[16];
    auto stamp = [&](int i) { if constexpr (STAMP) ts[i] = __builtin_readcyclecounter(); };
    stamp(0);
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
    // (unconditional, first in issue order: thread t < 256 the lateral vectors' entry t, every thread a mask bias of its image)
    const float sb_s = p.sc[tid & 255], sb_b = p.bc[tid & 255], sb_m = p.mb[(long long)b * NQP + (tid & (NQP - 1))];
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
    if (tid < 256) { SB[tid] = sb_s; SB[256 + tid] = sb_b; }
    if (tid < NQP) SB[512 + tid] = sb_m;
    stamp(1);
    __syncthreads();
    stamp(2);

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
        stamp(3);
        __syncthreads();                                      // every wave is done reading the c1 rows
        stamp(4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = wave * 32 + 8 * q + 4 * half;
            const f32x4 s = *reinterpret_cast<const f32x4*>(SB + n), bb = *reinterpret_cast<const f32x4*>(SB + 256 + n);
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
    stamp(5);
    __syncthreads();
    stamp(6);

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
    stamp(7);
    __syncthreads();
    stamp(8);

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
        stamp(9);
        __syncthreads();                                      // p1 is dead: the tile becomes the [128][nq] f32 staging buffer
        stamp(10);
        float* St = reinterpret_cast<float*>(At);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int nt = (wave & 1) + 2 * ps;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nt * 32 + 8 * q + 4 * half;
                const f32x4 mbv = *reinterpret_cast<const f32x4*>(SB + 512 + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (n + e < p.nq) {
                        float v = acc[ps][4 * q + e] + mbv[e];
#if MH_ABLATE != 5
                        if (p.apply_sigmoid) v = mh_sigmoid(v);
#endif
                        if (p.planar) St[(n + e) * MH_BM + r * 32 + l31] = v;
                        else St[(r * 32 + l31) * p.nq + n + e] = v;
                    }
                }
            }
        }
    }
    stamp(11);
    __syncthreads();
    stamp(12);
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
    if constexpr (STAMP) {
        stamp(13);
        if (dbg && lane == 0)
            for (int i = 0; i < 14; ++i) dbg[((long long)blockIdx.x * 8 + wave) * 16 + i] = ts[i];
    }
}

static unsigned long long* g_mh_dbg = nullptr;
constexpr size_t MH_LDS_TOTAL = MH_LDS_BYTES + (512 + MH_NQP_MAX) * sizeof(float);
static_assert(2 * MH_LDS_TOTAL <= 160 * 1024, "two workgroups per CU");


// ------------------------------------------------------------------------------------------------------------------------------------
// Round 5: PERSISTENT, software-pipelined form (an experiment that LOST - kept, tested bit-identical, off by default; see the launcher).
// The phase ablation of mask_head_kernel shows its phases
// ADD - c1 load 56 + lateral GEMM 139 + bilinear taps 136 + mask GEMM / sigmoid 84 + store 31 of 555 us: the two workgroups of a CU do
// not hide each other's memory phases.  Here ONE 8-wave workgroup per CU walks its XCD's run of 128-pixel tiles and overlaps the memory
// phases of a tile with its own MFMA phases:
//   * the c1 rows of tile i+1 arrive by LDS-DMA (`buffer_load ... lds`: no registers) into a second, UNPADDED operand tile (512-byte
//     rows, 16-byte chunks XOR-swizzled with row & 15 on the source address and on the fragment reads) while tile i runs its mask GEMM,
//     sigmoid and store;
//   * the 16 bilinear tap vectors of a thread (two items of four pixels x eight taps) are requested BEFORE the lateral GEMM of the
//     same tile into 64 registers (one workgroup per CU: 256 registers per lane) and are consumed after it;
//   * the probability store goes through a bounds-checked buffer descriptor (always four stores per thread), so the wait for the next
//     tile's DMAs at the top of the loop is a counted one that never waits for the stores.
// Same arithmetic and rounding points as mask_head_kernel: bit-identical output (tests).
constexpr int MHP_C1_BYTES = MH_BM * MH_C * 2;                                     // 64 KB, unpadded, swizzled
constexpr int MHP_VEC_BYTES = (256 + 256 + MH_NQP_MAX) * 4;                        // folded BN scale / bias of the lateral conv, mask bias
constexpr size_t MHP_LDS_BYTES = (size_t)MHP_C1_BYTES + MH_LDS_BYTES + MHP_VEC_BYTES;   // + the working tile: 134.1 KB

// LDS-only workgroup barrier: with an LDS-DMA in flight __syncthreads() would drain vmcnt(0) - the next tile's rows AND the stores
#define MHP_SYNC()                                             \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    } while (0)

template <int NQP>
__global__ __launch_bounds__(512, 1) void mask_head_pipe_kernel(const MaskHeadArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NPASS = NQP / 64, NTILES = NQP / 32;
    extern __shared__ __attribute__((aligned(1024))) unsigned char mh_smem[];
    unsigned char* C1t = mh_smem;                                                  // [128][256] bf16, chunk c of row r at (c ^ (r & 15)) * 16
    bf16_t* At = reinterpret_cast<bf16_t*>(mh_smem + MHP_C1_BYTES);               // [128][264]: lateral -> p1 -> f32 staging
    float* SCl = reinterpret_cast<float*>(mh_smem + MHP_C1_BYTES + MH_LDS_BYTES);  // [256] scale, [256] bias, [NQP] mask bias of the tile's image:
    float* BCl = SCl + 256;                                                        // read from LDS so that no epilogue waits on vmcnt
    float* MBl = BCl + 256;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = p.H * p.W;
    const int ntile = (int)((long long)p.B * per / MH_BM);
    // XCD x owns a contiguous run of tiles (its L2 then holds only the t1 rows its own tiles tap); its workgroups walk the run with stride
    const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int tq = ntile / nx, tr = ntile % nx;
    const int run_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int run_end = run_begin + tq + (xcd < tr ? 1 : 0);
    const int stride = ((int)gridDim.x - xcd + 7) / 8;
    int tile = run_begin + slot;
    if (tile >= run_end) return;

    const __amdgpu_buffer_rsrc_t c1src = __builtin_amdgcn_make_buffer_rsrc((void*)p.c1, 0, (int)((long long)p.B * per * MH_C * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t probr = __builtin_amdgcn_make_buffer_rsrc((void*)p.prob, 0, (int)((long long)p.B * per * p.nq * 4), 0x00020000);
    // DMA d of wave w fills rows 2 (8 w + d), + 1: lane -> row 2 (8 w + d) + (lane >> 5), PHYSICAL chunk lane & 31 <- logical chunk (lane & 31) ^ (row & 15)
    // (row & 15 = 2 d + half: the swizzle of DMA d is one XOR away from that of DMA 0)
    // ALWAYS eight DMAs (`on` = false: out of the descriptor's range, nothing is fetched): a conditional DMA would make the compiler's own
    // counted waits for the mask-weight registers assume the path without DMAs - and drain them
    auto stage_c1 = [&](int t, int l31, int half, bool on) {
        const unsigned x0 = (unsigned)((l31 ^ half) * 16), r0b = (unsigned)((wave * 16 + half) * MH_C * 2);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const unsigned vo = on ? r0b + (x0 ^ (unsigned)(d * 32)) : 0x7FFFFFF0u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(c1src, (__attribute__((address_space(3))) void*)(C1t + (wave * 8 + d) * 1024), 16, vo,
                                                     (unsigned)t * (unsigned)(MH_BM * MH_C * 2) + (unsigned)(d * 1024), 0, 0);
        }
    };
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    auto unpack = [](unsigned int wd) { return f32x2_t{__uint_as_float(wd << 16), __uint_as_float(wd & 0xffff0000u)}; };
    const int H2 = p.H >> 1, W2 = p.W >> 1;

    if (tid < 256) { SCl[tid] = p.sc[tid]; BCl[tid] = p.bc[tid]; }              // (visible behind the first barrier; waited for before the DMAs start)
    float mb_in = p.mb[(long long)((long long)tile * MH_BM / per) * NQP + (tid & (NQP - 1))];     // mask bias of the first tile's image
    stage_c1(tile, tid & 31, (tid >> 5) & 1, true);
    bf16x8 ring[MH_RING], mwf[MH_RING];
    bool first = true;
    while (true) {
        // Every per-thread quantity of a tile is derived from an OPAQUE copy of the thread index: left to itself the compiler hoists the
        // thread-dependent halves of all addresses out of the tile loop - dozens of registers that live across it, spill, and whose
        // reloads (scratch loads count in vmcnt like any load) would turn the counted waits below into full drains.
        int tl = tid;
        asm volatile("" : "+v"(tl));
        const int lane = tl & 63, l31 = lane & 31, half = lane >> 5;
        const long long m0 = (long long)tile * MH_BM;
        const int b = (int)(m0 / per), pix0 = (int)(m0 % per);
        const bf16_t* wlp = p.wc + ((long long)wave * 16 * 64 + lane) * 8;
        const bf16_t* wmp = p.mw + (((long long)b * NTILES + (wave & 1)) * 16) * 512 + lane * 8;
        // ---- requests of this tile: lateral weights (ring), bilinear taps (two items of four pixels: 16 vectors)
#pragma unroll
        for (int s = 0; s < MH_RING; ++s) ring[s] = *reinterpret_cast<const bf16x8*>(wlp + s * 512);
        u32x4_t tap[2][8];
        float lyv[2];
        int owv[2];
        const int oh0 = pix0 / p.W, ow0 = pix0 - oh0 * p.W;
        const bf16_t* tb = p.t1 + (long long)b * H2 * W2 * MH_C;
        auto request_taps = [&](auto IC_) {
            constexpr int i = decltype(IC_)::value;
            const int c = tl + i * 512, gq = c >> 5, col = (c & 31) * 8;
            int oh = oh0, ow = ow0 + gq * 4;
            while (ow >= p.W) { ow -= p.W; ++oh; }
            const float sy = fmaxf(0.5f * (oh + 0.5f) - 0.5f, 0.f);
            const int y0 = (int)sy, y1 = min(y0 + 1, H2 - 1);
            lyv[i] = sy - y0; owv[i] = ow;
            const int j2 = ow >> 1;
            const int ca = max(j2 - 1, 0), cb = j2, cc = min(j2 + 1, W2 - 1), cd = min(j2 + 2, W2 - 1);
            const bf16_t* r0p = tb + (long long)y0 * W2 * MH_C + col;
            const bf16_t* r1p = tb + (long long)y1 * W2 * MH_C + col;
            tap[i][0] = *reinterpret_cast<const u32x4_t*>(r0p + ca * MH_C); tap[i][1] = *reinterpret_cast<const u32x4_t*>(r0p + cb * MH_C);
            tap[i][2] = *reinterpret_cast<const u32x4_t*>(r0p + cc * MH_C); tap[i][3] = *reinterpret_cast<const u32x4_t*>(r0p + cd * MH_C);
            tap[i][4] = *reinterpret_cast<const u32x4_t*>(r1p + ca * MH_C); tap[i][5] = *reinterpret_cast<const u32x4_t*>(r1p + cb * MH_C);
            tap[i][6] = *reinterpret_cast<const u32x4_t*>(r1p + cc * MH_C); tap[i][7] = *reinterpret_cast<const u32x4_t*>(r1p + cd * MH_C);
        };
        request_taps(std::integral_constant<int, 0>{});          // item 0 flies under the lateral GEMM; item 1 is requested behind its MFMAs
        // ---- the c1 rows of this tile: 8 DMAs per wave issued one tile ago (or above).  vmcnt retires in issue order; younger than the
        //      DMAs are the 8 + 8 loads just requested and, except for the first tile, the previous tile's NQP / 16 probability stores -
        //      a counted wait that leaves exactly those in flight (the stores are never waited for)
        if (first) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (NQP == 64) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        if (tl < NQP) MBl[tl] = mb_in;                          // (requested BEFORE the DMAs: arrived; read behind three more barriers)
        MHP_SYNC();
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
                    const int row = r * 32 + l31;
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(C1t + row * 512 + (((kk * 2 + half) ^ (l31 & 15)) * 16));
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[kk % MH_RING], af, acc[r], 0, 0, 0);
                }
                if (kk + MH_RING < 16) ring[kk % MH_RING] = *reinterpret_cast<const bf16x8*>(wlp + (kk + MH_RING) * 512);
                else ring[kk % MH_RING] = *reinterpret_cast<const bf16x8*>(wmp + (kk + MH_RING - 16) * 512);     // mask GEMM k-steps 0..7
                if ((kk & 1) == 1) __builtin_amdgcn_sched_barrier(0);
            }
            request_taps(std::integral_constant<int, 1>{});
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = wave * 32 + 8 * q + 4 * half;
                const f32x4 sc4 = *reinterpret_cast<const f32x4*>(SCl + n), bb = *reinterpret_cast<const f32x4*>(BCl + n);
#pragma unroll
                for (int r = 0; r < MH_RT; ++r) {
                    us4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[r][4 * q + e] * sc4[e];
                        v += bb[e];
                        o[e] = f32_to_bf16(v > 0.f ? v : 0.f);
                    }
                    *reinterpret_cast<us4*>(At + (r * 32 + l31) * MH_LD + n) = o;
                }
            }
        }
        MHP_SYNC();                                               // lateral complete in At; every wave is done reading the c1 tile
        // ---- p1 = lateral + relu(bilinear_2x(t1)) from the tap registers
        {
            auto blend_store = [&](int r, int col, const u32x4_t& v00, const u32x4_t& v01, const u32x4_t& v10, const u32x4_t& v11, float lx, float ly) {
                const float hy = 1.f - ly, hx = 1.f - lx;
                const u32x4_t lw = *reinterpret_cast<const u32x4_t*>(At + r * MH_LD + col);
                u32x4_t ow4;
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
            auto lx_of = [](int o) { const float sx = fmaxf(0.5f * (o + 0.5f) - 0.5f, 0.f); return sx - (float)(int)sx; };
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = tl + i * 512, r0 = (c >> 5) * 4, col = (c & 31) * 8;
                const int ow = owv[i];
                blend_store(r0, col, tap[i][0], tap[i][1], tap[i][4], tap[i][5], lx_of(ow), lyv[i]);
                blend_store(r0 + 1, col, tap[i][1], tap[i][2], tap[i][5], tap[i][6], lx_of(ow + 1), lyv[i]);
                blend_store(r0 + 2, col, tap[i][1], tap[i][2], tap[i][5], tap[i][6], lx_of(ow + 2), lyv[i]);
                blend_store(r0 + 3, col, tap[i][2], tap[i][3], tap[i][6], tap[i][7], lx_of(ow + 3), lyv[i]);
            }
        }
        MHP_SYNC();
        // ---- everything the rest of the tile consumes from global memory is requested BEFORE the next tile's DMAs (a consumed load that
        //      is younger than a DMA makes its wait a wait for the DMA: vmcnt retires in order): mask weights k-steps 8..15
        const int mr = wave >> 1;
#pragma unroll
        for (int s2 = 0; s2 < MH_RING; ++s2) mwf[s2] = *reinterpret_cast<const bf16x8*>(wmp + (s2 + MH_RING) * 512);
        // ---- the next tile's c1 rows: the c1 tile is dead since the barrier behind the lateral GEMM
        const int next = tile + stride;
        const bool more = next < run_end;
        if (more) mb_in = p.mb[(long long)((long long)next * MH_BM / per) * NQP + (tl & (NQP - 1))];
        stage_c1(next, l31, half, more);
        // ---- mask logits: NQP (padded) planes = NQP/32 column tiles x 4 row tiles = 8 * NPASS (tile, rows) pairs, NPASS per wave
        {
            const int r = mr;
            f32x16 acc[NPASS];
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ps][e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(At + (r * 32 + l31) * MH_LD + kk * 16 + half * 8);
                    acc[ps] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kk < MH_RING ? ring[kk] : mwf[kk - MH_RING], af, acc[ps], 0, 0, 0);
                    // (NQP = 128 only: the second column tile's fragments are requested behind the DMAs - their wait includes the DMAs')
                    if (ps + 1 < NPASS) {
                        if (kk < MH_RING) ring[kk] = *reinterpret_cast<const bf16x8*>(wmp + ((long long)(ps + 1) * 32 + kk) * 512);
                        else mwf[kk - MH_RING] = *reinterpret_cast<const bf16x8*>(wmp + ((long long)(ps + 1) * 32 + kk) * 512);
                    }
                }
            }
            MHP_SYNC();                                           // p1 is dead: the tile becomes the [128][nq] f32 staging buffer
            float* St = reinterpret_cast<float*>(At);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int nt = (wave & 1) + 2 * ps;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = nt * 32 + 8 * q + 4 * half;
                    const f32x4 mb4 = *reinterpret_cast<const f32x4*>(MBl + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.nq) {
                            float v = acc[ps][4 * q + e] + mb4[e];
                            if (p.apply_sigmoid) v = mh_sigmoid(v);
                            St[(r * 32 + l31) * p.nq + n + e] = v;
                        }
                    }
                }
            }
        }
        MHP_SYNC();
        {
            // 128 * nq floats, contiguous; NQP / 16 16-byte stores per thread whatever nq (<= NQP: NQP * 32 vectors; the vectors >= total4
            // are pointed out of the descriptor's range, where the bounds check drops them - the count of stores in flight is fixed)
            const float* St = reinterpret_cast<const float*>(At);
            const int total4 = MH_BM * p.nq / 4;
            const unsigned base = (unsigned)((m0 * p.nq) * 4);
#pragma unroll
            for (int k = 0; k < NQP / 16; ++k) {
                const int i = tl + k * 512;
                const bool on = i < total4;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(St + 4 * (on ? i : 0));
                __builtin_amdgcn_raw_buffer_store_b128(v, probr, on ? (int)(base + (unsigned)i * 16u) : (int)0x7FFFFFF0, 0, 0);
            }
        }
        MHP_SYNC();                                               // staging reads done before the next tile's lateral epilogue rewrites At
        if (!more) break;
        tile = next;
        first = false;
    }
#endif
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

extern "C" void nps_mask_head_debug_buffer(void* buf) { nps::g_mh_dbg = (unsigned long long*)buf; }

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
    // round 5: the persistent software-pipelined form (row-major probabilities, W % 4 == 0, everything within 32-bit byte offsets) runs
    // only on request - flag bit 3 (+ 8) or NOPESAC_MASK_HEAD_PIPE=1: measured 590 us against 460 us of the two-workgroups-per-CU form
    // at the headline shape (a tile costs the SUM of its HBM, LDS, vector-memory, MFMA and VALU times in one
    // workgroup whatever is prefetched; two workgroups per CU overlap those resources, one does not)
    static const bool pipe_on = getenv("NOPESAC_MASK_HEAD_PIPE") && atoi(getenv("NOPESAC_MASK_HEAD_PIPE")) == 1;
    if ((pipe_on || ((apply_sigmoid >> 3) & 1)) && !a.planar && !a.taps1 && W % 4 == 0 && (long long)B * H * W * 256 * 2 < (1ll << 31) &&
        (long long)B * H * W * nq * 4 < (1ll << 31) - (1 << 20)) {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        (void)hipGetDevice(&dev);
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        const int grid = (int)(blocks < cus ? blocks : cus);
        if (nq <= 64) {
            NPS_ENSURE_LDS((int)MHP_LDS_BYTES, mask_head_pipe_kernel<64>);
            hipLaunchKernelGGL(mask_head_pipe_kernel<64>, dim3(grid), dim3(512), MHP_LDS_BYTES, (hipStream_t)stream, a);
        } else {
            NPS_ENSURE_LDS((int)MHP_LDS_BYTES, mask_head_pipe_kernel<128>);
            hipLaunchKernelGGL(mask_head_pipe_kernel<128>, dim3(grid), dim3(512), MHP_LDS_BYTES, (hipStream_t)stream, a);
        }
        NPS_LAUNCH_RET();
    }
    if (nq <= 64) {
        if (g_mh_dbg) {                                       // tuning runs only (scripts/mask_head_stamps.py)
            NPS_ENSURE_LDS((int)MH_LDS_TOTAL, mask_head_kernel<64, true>);
            hipLaunchKernelGGL((mask_head_kernel<64, true>), dim3((unsigned)blocks), dim3(512), MH_LDS_TOTAL, (hipStream_t)stream, a, g_mh_dbg);
            NPS_LAUNCH_RET();
        }
        NPS_ENSURE_LDS((int)MH_LDS_TOTAL, mask_head_kernel<64>);
        hipLaunchKernelGGL((mask_head_kernel<64>), dim3((unsigned)blocks), dim3(512), MH_LDS_TOTAL, (hipStream_t)stream, a, (unsigned long long*)nullptr);
    } else {
        NPS_ENSURE_LDS((int)MH_LDS_TOTAL, mask_head_kernel<128>);
        hipLaunchKernelGGL((mask_head_kernel<128>), dim3((unsigned)blocks), dim3(512), MH_LDS_TOTAL, (hipStream_t)stream, a, (unsigned long long*)nullptr);
    }
    NPS_LAUNCH_RET();
}
