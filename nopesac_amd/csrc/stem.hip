// Fused ResNet stem for the bf16 path: conv 7x7/s2/p3 (3->64, input padded to 4 channels) + FrozenBN + ReLU +
// max-pool 3x3/s2/p1 in ONE kernel (detectron2 BasicStem; call site meta_arch/siamese_planeTR.py:456).
//
// The un-fused path writes the 64 x 240 x 320 x 64 conv output (629 MB at 64 images) and reads it back for the pool;
// here a workgroup owns a 4 x 20 tile of POOLED pixels:
//   * the 23 x 88 input-pixel patch it needs (8 bytes / pixel) and the 64 x 224 weight matrix go to LDS once;
//   * the 9 x 41 conv outputs under the pooled tile are an implicit GEMM straight out of the LDS patch:
//     K index = (kh*8 + kw)*4 + c with kw padded 7 -> 8 (zero weights), so 8 consecutive k = 2 horizontally adjacent
//     input pixels = one aligned 16-byte ds_read_b128 A fragment - no im2col buffer;
//     M = 369 pixels -> 12 row tiles of 32 (3 per wave), N = 64, K = 224 -> 14 x v_mfma_f32_32x32x16_bf16 per tile pair;
//   * BN + ReLU in registers, conv tile -> LDS (bf16, pool-padding positions = -inf), 3x3/s2 max from LDS,
//     16-byte stores of the pooled NHWC rows.
// HBM traffic per image: 2.4 MB in + 2.4 MB out (instead of 2.4 + 9.8 + 9.8 + 2.4).
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

constexpr int ST_PH = 4, ST_PW = 20;                 // pooled tile
constexpr int ST_CH = 2 * ST_PH + 1, ST_CW = 2 * ST_PW + 1;   // conv tile 9 x 41
constexpr int ST_M = ST_CH * ST_CW;                  // 369 conv pixels
constexpr int ST_IH = 2 * (ST_CH - 1) + 7;           // 23 input rows
constexpr int ST_IW = 88;                            // 2*(41-1)+7 = 87 -> 88 (row = 704 bytes, 16-byte multiple)
constexpr int ST_K = 224, ST_WLD = 232;              // weight row: 224 + 8 pad elements (464 bytes)
constexpr int ST_CLD = 72;                           // conv tile row: 64 + 8 pad elements (144 bytes)
constexpr int ST_PATCH_BYTES = ST_IH * ST_IW * 8 + 64;          // + slack for the padded kw = 7 tap
constexpr int ST_W_BYTES = 64 * ST_WLD * 2;
constexpr int ST_CONV_BYTES = ST_M * ST_CLD * 2;     // 53,136 B: three workgroups per CU (the 384-row version allowed two)
constexpr int ST_LDS = (ST_PATCH_BYTES + ST_W_BYTES) > ST_CONV_BYTES ? (ST_PATCH_BYTES + ST_W_BYTES) : ST_CONV_BYTES;

// RAW = 1: x is not read; the patch comes straight from the f32 NCHW images `xraw` [B,3,H,W], normalised per channel
// ((v - mean[c]) / std[c], then rounded to bf16 - exactly what nopesac_preprocess_nchw_to_nhwc writes) while it is staged:
// the 157 MB bf16 NHWC copy of the batch is never written or read.
// RAW = 2 (round 4): the normalisation is FOLDED into the weights and the BN shift by the caller (w' = w / std[c], shift' = shift +
// scale * sum w' (128 - mean[c])): the patch holds v - 128, which is EXACT in bf16 for 8-bit pixel values - the operand rounding of
// the normalised image (the larger half of the stem's bf16 error, profiles/r4_bf16_attribution_stem.json) is gone, and so are the 24
// f32 divisions per thread.  Positions outside the image hold `mean` = (mean[c] - 128), the value whose folded contribution is zero
// like the reference's zero padding of the normalised image; `stdv` is not read.
// STAMP: tuning build - cycle stamps of every (workgroup, wave) at the phase boundaries into dbg[workgroup][4 waves][16] (scripts/stem_stamps.py)
template <int RAW, bool STAMP = false>
__global__ __launch_bounds__(256, 3) void stem_fused_kernel(const bf16_t* __restrict__ x, const float* __restrict__ xraw,
                                                         const float* __restrict__ mean, const float* __restrict__ stdv,
                                                         const bf16_t* __restrict__ w,
                                                         const float* __restrict__ scale, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ y, int H, int W, int CH, int CW, int PH, int PW,
                                                         unsigned long long* dbg = nullptr) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[ST_LDS + 512];
    // BN scale / shift parked in LDS behind the tile (round 6: as global loads inside the epilogue they were exposed L2 round trips -
    // the epilogue took 6.5 k of the workgroup's 26.9 k cycles for ~800 instructions, in-kernel stamps profiles/r6_e_*)
    float* sb_lds = reinterpret_cast<float*>(lds + ST_LDS);
    const float sb_v = (threadIdx.x & 64 ? bias : scale)[threadIdx.x & 63];      // (unguarded load, written to LDS in front of the first barrier)
    unsigned long long ts[16];
    auto stamp = [&](int i) { if constexpr (STAMP) ts[i] = __builtin_readcyclecounter(); };
    stamp(0);
    unsigned char* patch = lds;
    bf16_t* wl = reinterpret_cast<bf16_t*>(lds + ST_PATCH_BYTES);
    bf16_t* ctile = reinterpret_cast<bf16_t*>(lds);
    const int b = blockIdx.z;
    const int py0 = blockIdx.y * ST_PH, px0 = blockIdx.x * ST_PW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;          // first conv row / col under this pooled tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;          // first input row / col
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- the weights' loads first (all 7 of a thread in flight; L2 hits), then the patch's: ONE memory round trip for both
    constexpr int WIT = 64 * (ST_K / 8) / 256;
    static_assert(64 * (ST_K / 8) % 256 == 0, "weight chunking");
    us8 wr[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int i = tid + it * 256;
        wr[it] = *reinterpret_cast<const us8*>(w + (i / (ST_K / 8)) * ST_K + (i % (ST_K / 8)) * 8);
    }
    // ---- stage the input patch (zero outside the image)
    if constexpr (RAW != 0) {
        const float* xr = xraw + (long long)b * 3 * H * W;
        const float m0 = mean[0], m1 = mean[1], m2 = mean[2];
        const float s0 = RAW == 1 ? stdv[0] : 1.f, s1 = RAW == 1 ? stdv[1] : 1.f, s2 = RAW == 1 ? stdv[2] : 1.f;
        // every load of the patch goes out BEFORE the first one is consumed: the rolled loop (load, divide, ds_write, next) exposed
        // one HBM round trip per iteration - 8 in a row per workgroup; the stem ran at 1.1 TB/s (0.33 ms for 393 MB).
        // (Round 3: 16-byte loads of the image rows - 529 float4 groups per channel instead of 2024 scalar loads - were built and
        //  measured: 322 vs 315 us, no gain; the staging is not bound by the number of load instructions.  Reverted.)
        constexpr int NIT = (ST_IH * ST_IW + 8 + 255) / 256;
        float r0[NIT], r1[NIT], r2[NIT];
        const long long hw = (long long)H * W;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int r = i / ST_IW, c = i % ST_IW;
            const int iy = iy0 + r, ix = ix0 + c;
            // UNCONDITIONAL loads from a clamped (always valid) pixel; positions outside the image / the patch are replaced when the patch
            // is written below.  Round 6, read off the ISA: a guarded load (`if (inside) r = x[o]`) compiles to a branch with its
            // s_waitcnt vmcnt(0) INSIDE the block - the eight iterations ran as eight back-to-back HBM round trips (10.8 k of the
            // workgroup's 24.8 k cycles, in-kernel stamps) although the source issues all loads first
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            const long long o = (long long)iyc * W + ixc;
            r0[it] = xr[o]; r1[it] = xr[hw + o]; r2[it] = xr[2 * hw + o];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const int r = i / ST_IW, c = i % ST_IW;
            const int iy = iy0 + r, ix = ix0 + c;
            uint2 v = make_uint2(0u, 0u);
            if constexpr (RAW == 2) {
                const bool in = i < ST_IH * ST_IW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                v.x = f32x2_to_bf16x2(in ? r0[it] - 128.f : m0, in ? r1[it] - 128.f : m1);
                v.y = f32x2_to_bf16x2(in ? r2[it] - 128.f : m2, 0.f);
            } else {
                const bool in = i < ST_IH * ST_IW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                v.x = f32x2_to_bf16x2(in ? (r0[it] - m0) / s0 : 0.f, in ? (r1[it] - m1) / s1 : 0.f);
                v.y = f32x2_to_bf16x2(in ? (r2[it] - m2) / s2 : 0.f, 0.f);
            }
            if (i < ST_IH * ST_IW + 8) *reinterpret_cast<uint2*>(patch + (size_t)i * 8) = v;
        }
    } else {
        const bf16_t* xb = x + (long long)b * H * W * 4;
        constexpr int NIT0 = (ST_IH * ST_IW + 8 + 255) / 256;
        uint2 pv[NIT0];
#pragma unroll
        for (int it = 0; it < NIT0; ++it) {                       // unconditional loads from a clamped pixel (see RAW above)
            const int i = tid + it * 256;
            const int r = i / ST_IW, c = i % ST_IW;
            const int iy = iy0 + r, ix = ix0 + c;
            const int iyc = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), ixc = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            pv[it] = *reinterpret_cast<const uint2*>(xb + ((long long)iyc * W + ixc) * 4);
        }
#pragma unroll
        for (int it = 0; it < NIT0; ++it) {
            const int i = tid + it * 256;
            const int r = i / ST_IW, c = i % ST_IW;
            const int iy = iy0 + r, ix = ix0 + c;
            const bool in = i < ST_IH * ST_IW && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            if (i < ST_IH * ST_IW + 8) *reinterpret_cast<uint2*>(patch + (size_t)i * 8) = in ? pv[it] : make_uint2(0u, 0u);
        }
    }
    stamp(1);
    {   // weights -> LDS (their loads went out in front of the patch's)
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + it * 256;
            *reinterpret_cast<us8*>(wl + (i / (ST_K / 8)) * ST_WLD + (i % (ST_K / 8)) * 8) = wr[it];
        }
    }
    if (tid < 128) sb_lds[tid] = sb_v;
    stamp(2);
    __syncthreads();
    stamp(3);
    // ---- implicit GEMM: wave owns row tiles t = wave*3 .. +3 (32 conv pixels each), both 32-channel halves
    // Row tile T -> conv pixels: T = 0..8: columns 0..31 of conv row T; T = 9..11: the nine leftover columns - columns 32..39 as
    // 8-lane runs of rows 0..8, then column 40.  A lane's 16-byte A fragment sits at dword 352 cy + 4 cx, i.e. on bank slot
    // (8 cy + cx) mod 16: 32 consecutive columns of ONE row give every ds_read_b128 lane group sixteen different slots, and the 8-lane
    // runs of consecutive rows alternate between slots 0-7 and 8-15.  (32 consecutive pixels of the 41-wide tile wrap to the next row
    // inside most tiles, which shifts the lanes behind the wrap by one slot: a two-way conflict in every group the wrap splits.)
    const int half = lane >> 5;
    auto tile_pixel = [&](int T, int& cy, int& cx) -> bool {
        const int l = lane & 31;
        if (T < ST_CH) { cy = T; cx = l; return true; }
        const int q = (T - ST_CH) * 32 + l;
        if (q < 8 * ST_CH) { cy = q >> 3; cx = 32 + (q & 7); return true; }
        if (q < 9 * ST_CH) { cy = q - 8 * ST_CH; cx = 40; return true; }
        cy = 0; cx = 0;                                         // padded lanes compute garbage that is never stored
        return false;
    };
    static_assert(ST_CW == 41 && ST_CH == 9 && 12 * 32 >= ST_M, "row-tile enumeration");
    static_assert(3 * (ST_LDS + 512) <= 160 * 1024, "three workgroups per CU");
    int a_base[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int cy, cx;
        tile_pixel(wave * 3 + t, cy, cx);
        a_base[t] = ((2 * cy) * ST_IW + 2 * cx + 2 * half) * 8;
    }
    // Channel order of a 32-channel half: MFMA row 8q + 4h + e (lane half h holds rows 8q + 4h + {0..3}) is fed with the weights of
    // channel 16h + 4q + e, so a lane's 16 accumulators are 16 CONSECUTIVE channels and the conv tile is written in 16-byte pieces
    // (ds_write_b128 at the tile's 4-dword row skew: conflict-free; the 8-byte pieces of the natural order were two-way conflicts).  The
    // permuted rows of a ds_read_b128 lane group are the same SET of rows as before: the weight reads stay conflict-free.
    const int l31 = lane & 31;
    const int wrow = 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
    const int b_off = (wrow * ST_WLD + 8 * half) * 2;
    f32x16 acc[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 14; ++ks) {
        const int kh = ks >> 1, kw0 = (ks & 1) * 4;
        bf16x8 af[3], bf[2];
#pragma unroll
        for (int t = 0; t < 3; ++t) af[t] = *reinterpret_cast<const bf16x8*>(patch + a_base[t] + (kh * ST_IW + kw0) * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            bf[j] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(wl) + b_off + (j * 32 * ST_WLD + ks * 16) * 2);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[t], acc[t][j], 0, 0, 0);
    }
    stamp(4);
    __syncthreads();                                            // patch / weights are dead: reuse LDS for the conv tile
    stamp(5);
    // ---- BN + ReLU -> bf16 conv tile in LDS; lane holds pixel (lane&31) of tile t, channels j*32 + 16*half + 4q + {0..3}
    // Round 6 (PMC: 16 VALU instructions per MFMA, the kernel issue-bound once its loads overlapped): the pool-padding select only in
    // tiles that have a conv pixel outside the image (wave-uniform test; the interior tiles skip it).  BN stays multiply-then-add: ONE
    // fused multiply-add was built and measured (-96 VALU per wave) and taken out again - it changes the rounding of every stem output,
    // and the bf16 pose gates are fixed numbers tuned to nothing: camera_initRec R max moved 3.61 -> 4.60 deg on the benchmark pairs (gate 4.5).
    // Padding value: -0.0 (0x8000) - the pool below compares the bf16 bit patterns as SIGNED 16-bit integers, which orders the
    // non-negative ReLU outputs like their values and puts -0.0 below all of them (every pool window holds at least one real pixel).
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        int cy, cx;
        const bool valid = tile_pixel(wave * 3 + t, cy, cx);
        const int p = cy * ST_CW + cx;
        const int gy = cy0 + cy, gx = cx0 + cx;
        const bool inside = valid && (unsigned)gy < (unsigned)CH && (unsigned)gx < (unsigned)CW;
        const bool any_outside = __builtin_amdgcn_ballot_w64(valid && !inside) != 0ull;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int n = j * 32 + 16 * half + 8 * qq;
                uint4 o;
                unsigned ow[4];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const float4 s4 = *reinterpret_cast<const float4*>(sb_lds + n + 4 * jj), b4 = *reinterpret_cast<const float4*>(sb_lds + 64 + n + 4 * jj);
                    const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[t][j][4 * (2 * qq + jj) + e] * sv[e] + bv[e], 0.f);   // (two roundings, as in every other epilogue: see below)
                    ow[2 * jj] = f32x2_to_bf16x2(v[0], v[1]);
                    ow[2 * jj + 1] = f32x2_to_bf16x2(v[2], v[3]);
                }
                if (any_outside) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ow[e] = inside ? ow[e] : 0x80008000u;
                }
                o = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                if (valid) *reinterpret_cast<uint4*>(ctile + p * ST_CLD + n) = o;
            }
    }
    stamp(6);
    __syncthreads();
    stamp(7);
    // ---- 3x3 / s2 max-pool out of LDS; 8 channels (16 bytes) per work item
    // Item -> (pooled pixel, 16-byte channel piece): ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}
    // (+32); a group reads the 2 x 128 bytes of two pooled pixels FOUR apart (8 conv pixels = 288 dwords = 32 banks): conflict-free
    // (consecutive pooled pixels per 8 lanes put three of a group's four pixels on overlapping banks).  A block of 8 pooled pixels per
    // wave and pass; any bijection is correct, the pairing only matters for the banks.
    static_assert(ST_PH * ST_PW % 8 == 0, "pool blocks");
    for (int blk = wave; blk < ST_PH * ST_PW / 8; blk += 4) {
        const int c8 = (lane & 7) * 8;
        const int pp = blk * 8 + 4 * ((lane >> 4) & 1) + 2 * (lane >> 5) + (((lane >> 2) ^ (lane >> 3) ^ (lane >> 4)) & 1);
        const int ly = pp / ST_PW, lx = pp % ST_PW;
        const int py = py0 + ly, px = px0 + lx;
        if (py >= PH || px >= PW) continue;
        // max of the nine taps on the bf16 BIT PATTERNS as packed signed 16-bit integers (v_pk_max_i16: 4 instructions per tap for the
        // 8 channels instead of 8 unpacks + 8 f32 max): exact - ReLU outputs are >= +0 (integer order = value order), padding is -0.0
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        s16x2 m[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = s16x2{(short)0x8000, (short)0x8000};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const uint4 v = *reinterpret_cast<const uint4*>(ctile + ((2 * ly + dy) * ST_CW + 2 * lx + dx) * ST_CLD + c8);
                const unsigned vw[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = __builtin_elementwise_max(m[e], __builtin_bit_cast(s16x2, vw[e]));
            }
        us8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned wv = __builtin_bit_cast(unsigned, m[e]);
            o[2 * e] = (unsigned short)(wv & 0xFFFFu);
            o[2 * e + 1] = (unsigned short)(wv >> 16);
        }
        *reinterpret_cast<us8*>(y + (((long long)b * PH + py) * PW + px) * 64 + c8) = o;
    }
    if constexpr (STAMP) {
        stamp(8);
        if (dbg && lane == 0) {
            const long long wg = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            for (int i = 0; i < 9; ++i) dbg[(wg * 4 + wave) * 16 + i] = ts[i];
        }
    }
}

static unsigned long long* g_stem_dbg = nullptr;
extern "C" void nps_stem_debug_buffer(void* buf) { nps::g_stem_dbg = (unsigned long long*)buf; }

}  // namespace nps

extern "C" int nopesac_stem_fused_bf16(const void* x, const void* w, const float* scale, const float* bias, void* y, int B,
                                       int H, int W, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && w && scale && bias && y && B > 0 && H >= 7 && W >= 7, "stem_fused: bad args");
    NPS_CHECK_ARG(((uintptr_t)x % 8 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)scale % 16 == 0) &&
                  ((uintptr_t)bias % 16 == 0), "stem_fused: alignment");
    const int CH = (H + 6 - 7) / 2 + 1, CW = (W + 6 - 7) / 2 + 1;      // conv 7x7 / s2 / p3
    const int PH = (CH + 2 - 3) / 2 + 1, PW = (CW + 2 - 3) / 2 + 1;    // pool 3x3 / s2 / p1
    dim3 grid((PW + ST_PW - 1) / ST_PW, (PH + ST_PH - 1) / ST_PH, B);
    hipLaunchKernelGGL(stem_fused_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const bf16_t*)w, scale, bias, (bf16_t*)y, H, W, CH, CW, PH, PW);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_stem_fused_raw_bf16(const float* x_nchw, const float* mean, const float* stdv, const void* w, const float* scale,
                                           const float* bias, void* y, int B, int H, int W, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x_nchw && mean && stdv && w && scale && bias && y && B > 0 && H >= 7 && W >= 7, "stem_fused_raw: bad args");
    NPS_CHECK_ARG(((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)scale % 16 == 0) && ((uintptr_t)bias % 16 == 0),
                  "stem_fused_raw: alignment");
    const int CH = (H + 6 - 7) / 2 + 1, CW = (W + 6 - 7) / 2 + 1;      // conv 7x7 / s2 / p3
    const int PH = (CH + 2 - 3) / 2 + 1, PW = (CW + 2 - 3) / 2 + 1;    // pool 3x3 / s2 / p1
    dim3 grid((PW + ST_PW - 1) / ST_PW, (PH + ST_PH - 1) / ST_PH, B);
    hipLaunchKernelGGL(stem_fused_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)nullptr, x_nchw, mean, stdv,
                       (const bf16_t*)w, scale, bias, (bf16_t*)y, H, W, CH, CW, PH, PW);
    NPS_LAUNCH_RET();
}

// Raw-image stem with the normalisation folded into the operands (stem_fused_kernel<2>): x_nchw f32 [B,3,H,W] with 8-bit pixel values
// 0..255 (other values are rounded to bf16 after subtracting 128); pad3 = mean[c] - 128 (the raw value of the reference's zero
// padding, minus 128); w = bf16 [64][224] of w[o][kh][kw][c] / std[c]; bias = BN shift + scale * sum_k w'(128 - mean[c]).
extern "C" int nopesac_stem_fused_raw_shifted_bf16(const float* x_nchw, const float* pad3, const void* w_folded, const float* scale,
                                                   const float* bias_folded, void* y, int B, int H, int W, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x_nchw && pad3 && w_folded && scale && bias_folded && y && B > 0 && H >= 7 && W >= 7, "stem_fused_raw_shifted: bad args");
    NPS_CHECK_ARG(((uintptr_t)w_folded % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)scale % 16 == 0) && ((uintptr_t)bias_folded % 16 == 0),
                  "stem_fused_raw_shifted: alignment");
    const int CH = (H + 6 - 7) / 2 + 1, CW = (W + 6 - 7) / 2 + 1;      // conv 7x7 / s2 / p3
    const int PH = (CH + 2 - 3) / 2 + 1, PW = (CW + 2 - 3) / 2 + 1;    // pool 3x3 / s2 / p1
    dim3 grid((PW + ST_PW - 1) / ST_PW, (PH + ST_PH - 1) / ST_PH, B);
    if (g_stem_dbg)                                                    // tuning runs only
        hipLaunchKernelGGL((stem_fused_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)nullptr, x_nchw, pad3,
                           (const float*)nullptr, (const bf16_t*)w_folded, scale, bias_folded, (bf16_t*)y, H, W, CH, CW, PH, PW, g_stem_dbg);
    else
        hipLaunchKernelGGL((stem_fused_kernel<2>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)nullptr, x_nchw, pad3, (const float*)nullptr,
                           (const bf16_t*)w_folded, scale, bias_folded, (bf16_t*)y, H, W, CH, CW, PH, PW, (unsigned long long*)nullptr);
    NPS_LAUNCH_RET();
}
