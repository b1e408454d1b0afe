// Fused tail of a ResNet bottleneck for the bf16 path (d2 BottleneckBlock, STRIDE_IN_1X1 = False):
//
//     y  = relu( bn3(W3 . b) + shortcut )          shortcut = x            (identity blocks)
//                                                           = bn_sc(Wsc . x[:, ::s, ::s])   (first block of a stage)
//     a' = relu( bn1'(W1' . y) )                   the NEXT block's 1x1 reduce conv (optional)
//
// Un-fused, these are 2-3 HBM-bound launches: the expand conv reads b and the residual and writes y in 128-byte
// segments, the projection shortcut writes and re-reads a full-width tensor, and the next block's conv1 reads y again.
// Here a workgroup owns 64 consecutive pixels and the FULL channel width:
//   * b (64 x C), the optional second source x (64 x C2) and the residual rows (64 x C4) are fetched with every load in
//     flight at once and parked in LDS (one exposed HBM latency per workgroup instead of one per K-step);
//   * phase 1: per wave, 32-channel column tiles; weight fragments come straight from L2 in MFMA operand layout (each
//     wave reads different rows of W3: no redundancy, no LDS), activations from LDS; BN / shortcut / ReLU in registers;
//     the bf16 result overwrites the residual in LDS, so the y tile is complete on chip;
//   * y leaves as whole 2*C4-byte rows (16 B per lane, fully coalesced) and is, at the same time, the A operand of
//   * phase 2: a' = relu(bn(y . W1'^T)) with K = C4, again with register-streamed weight fragments.
// HBM bytes per pixel (res2, C = 64): 128 + 512 in, 512 + 128 out, instead of (128+512+512) + (512+128) un-fused.
// The arithmetic (MFMA K order, f32 epilogue order, bf16 rounding points) is the same as the generic kernels', so the
// identity-block results are bit-identical to the un-fused path; the projection block keeps the shortcut in f32
// instead of rounding it to bf16 in between.
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

struct PwArgs {
    const bf16_t* a1; const bf16_t* w3; const float* s3; const float* b3;      // expand conv: a1 [M][C], w3 [C4][C] (all weights fragment-major)
    const bf16_t* res;                                                         // identity shortcut [M][C4] or null
    const bf16_t* a2; const bf16_t* wsc; const float* ssc; const float* bsc;   // projection shortcut source [B][H2][W2][C2], wsc [C4][C2]
    int a2_H, a2_W, a2_stride, OH, OW;                                         // pixel m = (b, oy, ox) reads a2[b][oy*s][ox*s]
    bf16_t* y;                                                                 // [M][C4]
    const bf16_t* w1; const float* s1; const float* b1; bf16_t* o;             // next conv1: w1 [CN][C4], o [M][CN]
    long long M;
    int o_fp8;                                                                 // 1: o is OCP e4m3fn bytes (fp8 conv2 follows)
};

// 8 channels of the next block's conv1 output (bf16 in LDS) -> o, as bf16 or (o_fp8) saturated / rounded to e4m3fn
template <int CN>
__device__ __forceinline__ void pw_store_o(const PwArgs& p, long long m, int col, us8 v) {
    if (p.o_fp8) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = bf16_to_f32(v[e]);
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.o) + m * CN + col * 8) = f32x8_to_fp8(f);
    } else {
        *reinterpret_cast<us8*>(p.o + m * CN + col * 8) = v;
    }
}

template <int BM, int K>
__device__ __forceinline__ void pw_load_tile(const bf16_t* __restrict__ src, long long row0, long long M, int tid, us8 (&reg)[BM * K / 8 / 256]) {
    constexpr int CPR = K / 8;                       // 16-byte chunks per row
    static_assert(BM * CPR % 256 == 0, "tile chunking");
#pragma unroll
    for (int i = 0; i < BM * CPR / 256; ++i) {
        const int c = tid + i * 256, row = c / CPR, col = c % CPR;
        us8 v = us8{};
        if (row0 + row < M) v = *reinterpret_cast<const us8*>(src + (row0 + row) * K + col * 8);
        reg[i] = v;
    }
}
template <int BM, int K, int LD>
__device__ __forceinline__ void pw_store_tile(bf16_t* lds, int tid, const us8 (&reg)[BM * K / 8 / 256]) {
    constexpr int CPR = K / 8;
#pragma unroll
    for (int i = 0; i < BM * CPR / 256; ++i) {
        const int c = tid + i * 256, row = c / CPR, col = c % CPR;
        *reinterpret_cast<us8*>(lds + row * LD + col * 8) = reg[i];
    }
}

// LDS: region 0 = the phase-1 operands (b tile, then the projection source tile), re-used for the a' staging tile once phase 1
// is over; region 1 = the residual / y tile.
template <int C, int C4, int CN, int C2, int BM>
struct PwLds {
    static constexpr int A_LD = C + 8, A2_LD = C2 + 8, Y_LD = C4 + 8, O_LD = CN + 8;
    static constexpr int OPER = BM * A_LD + (C2 ? BM * A2_LD : 0), OUT = CN ? BM * O_LD : 0;
    static constexpr int R0 = OPER > OUT ? OPER : OUT;
    static constexpr size_t BYTES = 2 * (size_t)(R0 + BM * Y_LD);
    static constexpr int WAVES_PER_SIMD = 3 * BYTES <= 160 * 1024 ? 3 : 2;     // workgroups (4 waves) that fit one CU's LDS
};

template <int C, int C4, int CN, int C2, int BM>
__global__ __launch_bounds__(256, (PwLds<C, C4, CN, C2, BM>::WAVES_PER_SIMD)) void pw_chain_kernel(const PwArgs p) {
    typedef PwLds<C, C4, CN, C2, BM> L;
    constexpr int RT = BM / 32;                              // 32-pixel row tiles per workgroup
    constexpr int A_LD = L::A_LD, A2_LD = L::A2_LD, Y_LD = L::Y_LD, O_LD = L::O_LD;
    constexpr int KF1 = C / 16, KF1S = C2 / 16;              // weight fragments per column tile (expand / shortcut)
    constexpr int NT1 = C4 / 128;                            // column tiles per wave in phase 1
    static_assert(C % 32 == 0 && C4 % 128 == 0 && (CN == 0 || CN % 64 == 0) && C2 % 32 == 0, "channel counts");
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    bf16_t* A1 = reinterpret_cast<bf16_t*>(pw_smem);
    bf16_t* A2 = A1 + BM * A_LD;
    bf16_t* O = A1;                                          // aliases the operands (dead after phase 1)
    bf16_t* Y = A1 + L::R0;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * BM;

    // ---- every global read of the activations goes out first
    us8 ra1[BM * C / 8 / 256];
    pw_load_tile<BM, C>(p.a1, m0, p.M, tid, ra1);
    us8 rres[BM * C4 / 8 / 256];
    const bool has_res = p.res != nullptr;
    if (has_res) pw_load_tile<BM, C4>(p.res, m0, p.M, tid, rres);
    us8 ra2[C2 ? BM * C2 / 8 / 256 : 1];
    if constexpr (C2 > 0) {
        constexpr int CPR = C2 / 8;
#pragma unroll
        for (int i = 0; i < BM * CPR / 256; ++i) {
            const int c = tid + i * 256, row = c / CPR, col = c % CPR;
            us8 v = us8{};
            const long long m = m0 + row;
            if (m < p.M) {
                long long pix = m;
                if (p.a2_stride != 1 || p.a2_H != p.OH || p.a2_W != p.OW) {
                    const int per = p.OH * p.OW;
                    const int b = (int)(m / per), rem = (int)(m % per), oy = rem / p.OW, ox = rem % p.OW;
                    pix = ((long long)b * p.a2_H + oy * p.a2_stride) * p.a2_W + ox * p.a2_stride;
                }
                v = *reinterpret_cast<const us8*>(p.a2 + pix * C2 + col * 8);
            }
            ra2[i] = v;
        }
    }
    // weight fragments of this wave's first column tile (L2 resident)
    bf16x8 wf[2][KF1];
    bf16x8 wsf[2][C2 ? KF1S : 1];
    // every workgroup walks its waves' column tiles in a ROTATED order (start = blockIdx % NT1): the resident workgroups then pull
    // different weight lines from L2 at any one time instead of all queueing on the same channel (results are unaffected: the
    // column tiles are independent)
    const int rot = NT1 > 1 ? (int)(blockIdx.x % NT1) : 0;
    auto load_w1frags = [&](int j, int buf) {
        // fragment-major weights (see the header): one load instruction = 1 KB contiguous = 8 whole cache lines
        const int nt = wave * NT1 + (j + rot) % NT1;
#pragma unroll
        for (int kk = 0; kk < KF1; ++kk) wf[buf][kk] = *reinterpret_cast<const bf16x8*>(p.w3 + ((long long)(nt * KF1 + kk) * 64 + lane) * 8);
        if constexpr (C2 > 0) {
#pragma unroll
            for (int kk = 0; kk < KF1S; ++kk) wsf[buf][kk] = *reinterpret_cast<const bf16x8*>(p.wsc + ((long long)(nt * KF1S + kk) * 64 + lane) * 8);
        }
    };
    load_w1frags(0, 0);
    pw_store_tile<BM, C, A_LD>(A1, tid, ra1);
    if (has_res) pw_store_tile<BM, C4, Y_LD>(Y, tid, rres);
    if constexpr (C2 > 0) pw_store_tile<BM, C2, A2_LD>(A2, tid, ra2);
    __syncthreads();

    // ---- phase 1: y tile = relu(bn3(W3 b) + shortcut), column tiles (wave*NT1 + j), both 32-pixel row tiles
#pragma unroll
    for (int j = 0; j < NT1; ++j) {
        const int buf = j & 1;
        if (j + 1 < NT1) load_w1frags(j + 1, buf ^ 1);
        const int n0 = (wave * NT1 + (j + rot) % NT1) * 32;
        // one 32-pixel row tile at a time (the weight fragments stay in registers, the accumulators are re-used)
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            f32x16 acc, accs;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accs[e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < KF1; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(A1 + (r * 32 + l31) * A_LD + kk * 16 + half * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[buf][kk], af, acc, 0, 0, 0);
            }
            if constexpr (C2 > 0) {
#pragma unroll
                for (int kk = 0; kk < KF1S; ++kk) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(A2 + (r * 32 + l31) * A2_LD + kk * 16 + half * 8);
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wsf[buf][kk], af, accs, 0, 0, 0);
                }
            }
            // lane holds, for pixel r*32 + l31, channels n0 + 8q + 4*half + {0..3}
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 8 * q + 4 * half;
                const f32x4 s3 = *reinterpret_cast<const f32x4*>(p.s3 + n), b3 = *reinterpret_cast<const f32x4*>(p.b3 + n);
                f32x4 ss = {0.f, 0.f, 0.f, 0.f}, bs = {0.f, 0.f, 0.f, 0.f};
                if constexpr (C2 > 0) { ss = *reinterpret_cast<const f32x4*>(p.ssc + n); bs = *reinterpret_cast<const f32x4*>(p.bsc + n); }
                bf16_t* yp = Y + (r * 32 + l31) * Y_LD + n;
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (has_res) {
                    const us4 r4 = *reinterpret_cast<const us4*>(yp);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rv[e] = bf16_to_f32(r4[e]);
                }
                us4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[4 * q + e] * s3[e];
                    v += b3[e];
                    if constexpr (C2 > 0) {
                        float sc = accs[4 * q + e] * ss[e];
                        sc += bs[e];
                        rv[e] = sc;
                    }
                    v += rv[e];
                    o4[e] = f32_to_bf16(v > 0.f ? v : 0.f);
                }
                *reinterpret_cast<us4*>(yp) = o4;
            }
        }
    }
    __syncthreads();

    // ---- y leaves as whole rows
    {
        constexpr int CPR = C4 / 8;
#pragma unroll
        for (int i = 0; i < BM * CPR / 256; ++i) {
            const int c = tid + i * 256, row = c / CPR, col = c % CPR;
            if (m0 + row < p.M) *reinterpret_cast<us8*>(p.y + (m0 + row) * C4 + col * 8) = *reinterpret_cast<const us8*>(Y + row * Y_LD + col * 8);
        }
    }
    if constexpr (CN > 0) {
        // ---- phase 2: a' = relu(bn1(y W1'^T)), K = C4.  CN = 64: wave -> (column tile w&1, row tile w>>1); CN >= 128: wave -> column
        // tiles {w, w+4, ..}, both row tiles.  Weight fragments stream through a two-deep register ring, 8 k-steps per chunk.
        static_assert(CN >= 128 || BM == 64, "CN = 64 needs two row tiles to occupy four waves");
        constexpr int NR2 = CN >= 128 ? RT : 1;
        constexpr int NT2 = CN >= 128 ? CN / 128 : 1;
        constexpr int KCH = 8, NCH = C4 / 16 / KCH;
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
            const int nt = CN >= 128 ? wave + 4 * j : (wave & 1);
            const int r0 = CN >= 128 ? 0 : (wave >> 1);
            const bf16_t* wrow = p.w1 + ((long long)nt * (C4 / 16) * 64 + lane) * 8;      // fragment-major: [nt][kk][lane][8]
            bf16x8 wq[2][KCH];
#pragma unroll
            for (int kk = 0; kk < KCH; ++kk) wq[0][kk] = *reinterpret_cast<const bf16x8*>(wrow + kk * 512);
            f32x16 acc[NR2];
#pragma unroll
            for (int r = 0; r < NR2; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int buf = ch & 1;
                if (ch + 1 < NCH) {
#pragma unroll
                    for (int kk = 0; kk < KCH; ++kk) wq[buf ^ 1][kk] = *reinterpret_cast<const bf16x8*>(wrow + ((ch + 1) * KCH + kk) * 512);
                }
#pragma unroll
                for (int kk = 0; kk < KCH; ++kk)
#pragma unroll
                    for (int r = 0; r < NR2; ++r) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(Y + ((r0 + r) * 32 + l31) * Y_LD + (ch * KCH + kk) * 16 + half * 8);
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[buf][kk], af, acc[r], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = nt * 32 + 8 * q + 4 * half;
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.s1 + n), b1 = *reinterpret_cast<const f32x4*>(p.b1 + n);
#pragma unroll
                for (int r = 0; r < NR2; ++r) {
                    us4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[r][4 * q + e] * s1[e];
                        v += b1[e];
                        o4[e] = f32_to_bf16(v > 0.f ? v : 0.f);
                    }
                    *reinterpret_cast<us4*>(O + ((r0 + r) * 32 + l31) * O_LD + n) = o4;
                }
            }
        }
        __syncthreads();
        constexpr int CPR = CN / 8;
        static_assert(BM * CPR % 256 == 0, "a' tile chunking");
#pragma unroll
        for (int i = 0; i < BM * CPR / 256; ++i) {
            const int c = tid + i * 256, row = c / CPR, col = c % CPR;
            if (m0 + row < p.M) pw_store_o<CN>(p, m0 + row, col, *reinterpret_cast<const us8*>(O + row * O_LD + col * 8));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Wide variant (res4: C = 256, C4 = 1024): 32 pixels per workgroup, EIGHT waves, the whole 32 x C4 y tile in LDS (66 KB) and
// the weight fragments streamed through a two-deep register ring that is refilled one step (= one 32-channel column tile x
// 256 input channels = 16 fragment loads) ahead, across column tiles and across the two phases (enc_tail.hip / gnn_layer.hip).
// Wave w owns column tiles w*NT1 .. of y and w*NT2 .. of a'.
template <int C, int C4, int CN, int C2>
struct PwWide {
    static constexpr int BM = 32;
    static constexpr int A_LD = C + 8, A2_LD = C2 + 8, Y_LD = C4 + 8, O_LD = CN + 8;
    static constexpr int OPER = BM * A_LD + (C2 ? BM * A2_LD : 0), OUT = CN ? BM * O_LD : 0;
    static constexpr int R0 = OPER > OUT ? OPER : OUT;
    static constexpr size_t BYTES = 2 * (size_t)(R0 + BM * Y_LD);
    static constexpr int KS1 = C / 256, KS1S = C2 / 256, PER1 = KS1 + KS1S;     // ring steps per column tile (expand, shortcut)
    static constexpr int NT1 = C4 / 256, S1 = NT1 * PER1;                       // column tiles per wave, steps of phase 1
    static constexpr int KS2 = C4 / 256, NT2 = CN / 256, S2 = NT2 * KS2;
    static_assert(C % 256 == 0 && C4 % 256 == 0 && CN % 256 == 0 && C2 % 256 == 0, "wide tail: multiples of 256 channels");
};

struct PwRing {
    bf16x8 f[2][16];
};

template <int C, int C4, int CN, int C2>
__device__ __forceinline__ void pww_issue(PwRing& ring, const PwArgs& p, int s, int wave, int lane) {
    typedef PwWide<C, C4, CN, C2> W;
    const bf16_t* w;
    int kf_total, kf_off, nt;
    if (s < W::S1) {
        const int j = s / W::PER1, t = s % W::PER1;
        nt = wave * W::NT1 + j;
        if (t < W::KS1) { w = p.w3; kf_total = C / 16; kf_off = 16 * t; }
        else { w = p.wsc; kf_total = C2 / 16; kf_off = 16 * (t - W::KS1); }
    } else if (s < W::S1 + W::S2) {
        const int s2 = s - W::S1, j = s2 / W::KS2, t = s2 % W::KS2;
        nt = wave * W::NT2 + j;
        w = p.w1; kf_total = C4 / 16; kf_off = 16 * t;
    } else {
        return;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
        ring.f[s & 1][kk] = *reinterpret_cast<const bf16x8*>(w + ((long long)(nt * kf_total + kf_off + kk) * 64 + lane) * 8);
}
__device__ __forceinline__ void pww_gemm(const PwRing& ring, int buf, const bf16_t* A, int lda, f32x16& acc, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(A + l31 * lda + kk * 16 + half * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.f[buf][kk], af, acc, 0, 0, 0);
    }
}
template <int ROWS, int K, int LD>
__device__ __forceinline__ void pww_copy_rows(const bf16_t* __restrict__ src, long long row0, long long M, bf16_t* T, int tid) {
    constexpr int CPR = K / 8;
    static_assert(ROWS * CPR % 512 == 0, "tile chunking");
    us8 reg[ROWS * CPR / 512];
#pragma unroll
    for (int i = 0; i < ROWS * CPR / 512; ++i) {
        const int c = tid + i * 512, row = c / CPR, col = c % CPR;
        reg[i] = us8{};
        if (row0 + row < M) reg[i] = *reinterpret_cast<const us8*>(src + (row0 + row) * K + col * 8);
    }
#pragma unroll
    for (int i = 0; i < ROWS * CPR / 512; ++i) {
        const int c = tid + i * 512, row = c / CPR, col = c % CPR;
        *reinterpret_cast<us8*>(T + row * LD + col * 8) = reg[i];
    }
}

template <int C, int C4, int CN, int C2>
__global__ __launch_bounds__(512, 2) void pw_chain_wide_kernel(const PwArgs p) {
    typedef PwWide<C, C4, CN, C2> W;
    constexpr int BM = W::BM, A_LD = W::A_LD, A2_LD = W::A2_LD, Y_LD = W::Y_LD, O_LD = W::O_LD;
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    bf16_t* A1 = reinterpret_cast<bf16_t*>(pw_smem);
    bf16_t* A2 = A1 + BM * A_LD;
    bf16_t* O = A1;
    bf16_t* Y = A1 + W::R0;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * BM;
    const bool has_res = p.res != nullptr;
    PwRing ring;
    pww_issue<C, C4, CN, C2>(ring, p, 0, wave, lane);
    pww_copy_rows<BM, C, A_LD>(p.a1, m0, p.M, A1, tid);
    if (has_res) pww_copy_rows<BM, C4, Y_LD>(p.res, m0, p.M, Y, tid);
    if constexpr (C2 > 0) {
        constexpr int CPR = C2 / 8;
#pragma unroll
        for (int i = 0; i < BM * CPR / 512; ++i) {
            const int c = tid + i * 512, row = c / CPR, col = c % CPR;
            us8 v = us8{};
            const long long m = m0 + row;
            if (m < p.M) {
                long long pix = m;
                if (p.a2_stride != 1 || p.a2_H != p.OH || p.a2_W != p.OW) {
                    const int per = p.OH * p.OW;
                    const int b = (int)(m / per), rem = (int)(m % per), oy = rem / p.OW, ox = rem % p.OW;
                    pix = ((long long)b * p.a2_H + oy * p.a2_stride) * p.a2_W + ox * p.a2_stride;
                }
                v = *reinterpret_cast<const us8*>(p.a2 + pix * C2 + col * 8);
            }
            *reinterpret_cast<us8*>(A2 + row * A2_LD + col * 8) = v;
        }
    }
    __syncthreads();

    // ---- phase 1
    int s = 0;
#pragma unroll
    for (int j = 0; j < W::NT1; ++j) {
        f32x16 acc, accs;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accs[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < W::PER1; ++t) {
            pww_issue<C, C4, CN, C2>(ring, p, s + 1, wave, lane);
            if (t < W::KS1) pww_gemm(ring, s & 1, A1 + 256 * t, A_LD, acc, lane);
            else pww_gemm(ring, s & 1, A2 + 256 * (t - W::KS1), A2_LD, accs, lane);
            ++s;
        }
        const int n0 = (wave * W::NT1 + j) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + 8 * q + 4 * half;
            const f32x4 s3 = *reinterpret_cast<const f32x4*>(p.s3 + n), b3 = *reinterpret_cast<const f32x4*>(p.b3 + n);
            f32x4 ss = {0.f, 0.f, 0.f, 0.f}, bs = {0.f, 0.f, 0.f, 0.f};
            if constexpr (C2 > 0) { ss = *reinterpret_cast<const f32x4*>(p.ssc + n); bs = *reinterpret_cast<const f32x4*>(p.bsc + n); }
            bf16_t* yp = Y + l31 * Y_LD + n;
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (has_res) {
                const us4 r4 = *reinterpret_cast<const us4*>(yp);
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[e] = bf16_to_f32(r4[e]);
            }
            us4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[4 * q + e] * s3[e];
                v += b3[e];
                if constexpr (C2 > 0) {
                    float sc = accs[4 * q + e] * ss[e];
                    sc += bs[e];
                    rv[e] = sc;
                }
                v += rv[e];
                o4[e] = f32_to_bf16(v > 0.f ? v : 0.f);
            }
            *reinterpret_cast<us4*>(yp) = o4;
        }
    }
    __syncthreads();
    {
        constexpr int CPR = C4 / 8;
#pragma unroll
        for (int i = 0; i < BM * CPR / 512; ++i) {
            const int c = tid + i * 512, row = c / CPR, col = c % CPR;
            if (m0 + row < p.M) *reinterpret_cast<us8*>(p.y + (m0 + row) * C4 + col * 8) = *reinterpret_cast<const us8*>(Y + row * Y_LD + col * 8);
        }
    }
    if constexpr (CN > 0) {
        // ---- phase 2: a' = relu(bn1(y W1'^T)), K = C4 in KS2 ring steps per column tile
#pragma unroll
        for (int j = 0; j < W::NT2; ++j) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int t = 0; t < W::KS2; ++t) {
                pww_issue<C, C4, CN, C2>(ring, p, s + 1, wave, lane);
                pww_gemm(ring, s & 1, Y + 256 * t, Y_LD, acc, lane);
                ++s;
            }
            const int n0 = (wave * W::NT2 + j) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 8 * q + 4 * half;
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.s1 + n), b1 = *reinterpret_cast<const f32x4*>(p.b1 + n);
                us4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[4 * q + e] * s1[e];
                    v += b1[e];
                    o4[e] = f32_to_bf16(v > 0.f ? v : 0.f);
                }
                *reinterpret_cast<us4*>(O + l31 * O_LD + n) = o4;
            }
        }
        __syncthreads();
        constexpr int CPR = CN / 8;
#pragma unroll
        for (int i = 0; i < BM * CPR / 512; ++i) {
            const int c = tid + i * 512, row = c / CPR, col = c % CPR;
            if (m0 + row < p.M) pw_store_o<CN>(p, m0 + row, col, *reinterpret_cast<const us8*>(O + row * O_LD + col * 8));
        }
    }
}

template <int C, int C4, int CN, int C2>
static int pw_launch_wide(const PwArgs& a, hipStream_t stream) {
    constexpr size_t lds = PwWide<C, C4, CN, C2>::BYTES;
    NPS_ENSURE_LDS((int)lds, pw_chain_wide_kernel<C, C4, CN, C2>);
    hipLaunchKernelGGL((pw_chain_wide_kernel<C, C4, CN, C2>), dim3((unsigned)((a.M + 31) / 32)), dim3(512), lds, stream, a);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Streaming variant for identity blocks with a very wide y (res4: C4 = 1024).  Streaming weights per 32
// pixels (above) makes the L2 -> CU weight traffic the limiter (1 MB per 32 pixels in res4).  Here a workgroup owns BM = 64
// pixels - every weight fragment feeds two MFMAs - and y is produced in 256-channel CHUNKS: chunk c is computed into a
// [BM][256] LDS tile (over its residual, which was prefetched into registers during chunk c-1), leaves for HBM as 512-byte row
// segments, and is immediately consumed as K-slice c of the next block's conv1, whose accumulators stay in registers across
// the chunks.  LDS = b tile + one y chunk; three barriers per chunk.
template <int C, int C4, int CN, int BM>
struct PwStream {
    static constexpr int RT = BM / 32, A_LD = C + 8, Y_LD = 256 + 8, O_LD = CN + 8;
    static constexpr int NCH = C4 / 256, KS1 = C / 256, NT2 = CN / 256;          // chunks; ring steps per chunk: KS1 + NT2
    static constexpr int OPER = BM * A_LD + BM * Y_LD, OUT = CN ? BM * O_LD : 0;
    static constexpr int ELEMS = OPER > OUT ? OPER : OUT;
    static constexpr size_t BYTES = 2 * (size_t)ELEMS;
    static constexpr int PER = KS1 + NT2, STEPS = NCH * PER;
    static_assert(C % 256 == 0 && C4 % 256 == 0 && CN % 256 == 0 && BM % 32 == 0, "streaming tail: shapes");
};

// Rolling weight ring: 16 fragment registers; fragment kk of step s+1 is loaded into slot kk right after the MFMAs of step s
// have consumed it, so a full step (16 KB per wave) is always in flight with half the registers of a double buffer.
struct PwRoll {
    bf16x8 f[16];
};
template <int C, int C4, int CN, int BM>
__device__ __forceinline__ const bf16_t* pws_step_ptr(const PwArgs& p, int s, int wave, int lane) {
    typedef PwStream<C, C4, CN, BM> S;
    if (s >= S::STEPS) return nullptr;
    const int c = s / S::PER, t = s % S::PER;
    const bf16_t* w;
    int kf_total, kf_off, nt;
    if (t < S::KS1) { w = p.w3; kf_total = C / 16; kf_off = 16 * t; nt = c * 8 + wave; }             // y channels c*256 + wave*32
    else { w = p.w1; kf_total = C4 / 16; kf_off = 16 * c; nt = wave * S::NT2 + (t - S::KS1); }      // conv1 tile, K-slice c
    return w + ((long long)(nt * kf_total + kf_off) * 64 + lane) * 8;
}
template <int RT>
__device__ __forceinline__ void pws_gemm(PwRoll& ring, const bf16_t* __restrict__ next, const bf16_t* A, int lda, f32x16 (&acc)[RT], int lane) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(A + (r * 32 + l31) * lda + kk * 16 + half * 8);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.f[kk], af, acc[r], 0, 0, 0);
        }
        if (next) ring.f[kk] = *reinterpret_cast<const bf16x8*>(next + kk * 512);
        if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting all 16 k-steps' LDS reads (spills)
    }
}

template <int C, int C4, int CN, int BM>
__global__ __launch_bounds__(512, 2) void pw_chain_stream_kernel(const PwArgs p) {
    typedef PwStream<C, C4, CN, BM> S;
    constexpr int RT = S::RT, A_LD = S::A_LD, Y_LD = S::Y_LD, O_LD = S::O_LD;
    constexpr int RCH = BM * 32 / 512;                       // 16-byte chunks per thread of one [BM][256] tile
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    bf16_t* A1 = reinterpret_cast<bf16_t*>(pw_smem);
    bf16_t* Y = A1 + BM * A_LD;
    bf16_t* O = A1;                                          // aliases both tiles once the last chunk has been consumed
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * BM;
    PwRoll ring;
    {
        const bf16_t* w0 = pws_step_ptr<C, C4, CN, BM>(p, 0, wave, lane);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) ring.f[kk] = *reinterpret_cast<const bf16x8*>(w0 + kk * 512);
    }
    us8 rres[RCH];
    auto fetch_res = [&](int c) {
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            const int ch = tid + i * 512, row = ch >> 5, col = (ch & 31) * 8;
            rres[i] = us8{};
            if (m0 + row < p.M) rres[i] = *reinterpret_cast<const us8*>(p.res + (m0 + row) * C4 + c * 256 + col);
        }
    };
    auto park_res = [&]() {
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            const int ch = tid + i * 512, row = ch >> 5, col = (ch & 31) * 8;
            *reinterpret_cast<us8*>(Y + row * Y_LD + col) = rres[i];
        }
    };
    fetch_res(0);
    pww_copy_rows<BM, C, A_LD>(p.a1, m0, p.M, A1, tid);
    park_res();
    __syncthreads();

    f32x16 acc2[CN ? S::NT2 : 1][RT];
#pragma unroll
    for (int j = 0; j < (CN ? S::NT2 : 1); ++j)
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[j][r][e] = 0.f;
#pragma unroll 1
    for (int c = 0; c < S::NCH; ++c) {
        if (c + 1 < S::NCH) fetch_res(c + 1);
        // ---- y chunk c: wave owns channels c*256 + wave*32 .. +32, all RT row tiles
        f32x16 acc[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;
#pragma unroll
        for (int t = 0; t < S::KS1; ++t)
            pws_gemm<RT>(ring, pws_step_ptr<C, C4, CN, BM>(p, c * S::PER + t + 1, wave, lane), A1 + 256 * t, A_LD, acc, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nl = wave * 32 + 8 * q + 4 * half, n = c * 256 + nl;
            const f32x4 s3 = *reinterpret_cast<const f32x4*>(p.s3 + n), b3 = *reinterpret_cast<const f32x4*>(p.b3 + n);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                bf16_t* yp = Y + (r * 32 + l31) * Y_LD + nl;
                const us4 r4 = *reinterpret_cast<const us4*>(yp);
                us4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[r][4 * q + e] * s3[e];
                    v += b3[e];
                    v += bf16_to_f32(r4[e]);
                    o4[e] = f32_to_bf16(v > 0.f ? v : 0.f);
                }
                *reinterpret_cast<us4*>(yp) = o4;
            }
        }
        __syncthreads();                                     // y chunk complete
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            const int ch = tid + i * 512, row = ch >> 5, col = (ch & 31) * 8;
            if (m0 + row < p.M) *reinterpret_cast<us8*>(p.y + (m0 + row) * C4 + c * 256 + col) = *reinterpret_cast<const us8*>(Y + row * Y_LD + col);
        }
        if constexpr (CN > 0) {
#pragma unroll
            for (int j = 0; j < S::NT2; ++j)
                pws_gemm<RT>(ring, pws_step_ptr<C, C4, CN, BM>(p, c * S::PER + S::KS1 + j + 1, wave, lane), Y, Y_LD, acc2[j], lane);
        }
        __syncthreads();                                     // every wave is done with chunk c
        if (c + 1 < S::NCH) {
            park_res();
            __syncthreads();
        }
    }
    if constexpr (CN > 0) {
#pragma unroll
        for (int j = 0; j < S::NT2; ++j) {
            const int n0 = (wave * S::NT2 + j) * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 8 * q + 4 * half;
                const f32x4 s1 = *reinterpret_cast<const f32x4*>(p.s1 + n), b1 = *reinterpret_cast<const f32x4*>(p.b1 + n);
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    us4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc2[j][r][4 * q + e] * s1[e];
                        v += b1[e];
                        o4[e] = f32_to_bf16(v > 0.f ? v : 0.f);
                    }
                    *reinterpret_cast<us4*>(O + (r * 32 + l31) * O_LD + n) = o4;
                }
            }
        }
        __syncthreads();
        constexpr int CPR = CN / 8;
#pragma unroll
        for (int i = 0; i < BM * CPR / 512; ++i) {
            const int ch = tid + i * 512, row = ch / CPR, col = ch % CPR;
            if (m0 + row < p.M) pw_store_o<CN>(p, m0 + row, col, *reinterpret_cast<const us8*>(O + row * O_LD + col * 8));
        }
    }
}

template <int C, int C4, int CN, int BM>
static int pw_launch_stream(const PwArgs& a, hipStream_t stream) {
    constexpr size_t lds = PwStream<C, C4, CN, BM>::BYTES;
    NPS_ENSURE_LDS((int)lds, pw_chain_stream_kernel<C, C4, CN, BM>);
    hipLaunchKernelGGL((pw_chain_stream_kernel<C, C4, CN, BM>), dim3((unsigned)((a.M + BM - 1) / BM)), dim3(512), lds, stream, a);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Four-row-tile variant for the identity blocks of res2 / res3 (round 3).  PMC on the res3 tail above (32 pixels per workgroup,
// profiles/r2_pmc_res3_tail.json): 3.74 GB of L2 requests for 0.79 GB of HBM traffic - every 32 pixels re-stream the block's
// 262 KB of 1x1 weights from L2 - MFMA-busy 15 %, 2.85 TB/s.  Here a 4-wave workgroup owns BM = 128 pixels and y is produced in
// 128-channel chunks, i.e. exactly ONE 32-channel column tile per wave: no two waves ever load the same weight fragment and every
// fragment feeds FOUR MFMAs (the four 32-pixel row tiles) - a quarter of the L2 -> CU weight traffic, without staging weights in
// LDS.  Chunk c: y_c = relu(bn3(b W3_c^T) + x_c) is written in place over its residual in a [128][128] LDS tile, leaves for HBM as
// 256-byte row segments and is at once K-slice c of the next block's conv1, whose accumulators stay in registers across the
// chunks (pw_chain_stream_kernel's scheme with 4 instead of 8 waves).  LDS = b tile + one y chunk = 53-70 KB: two workgroups per
// CU, so one's load / store phases run under the other's MFMAs.  The residual of chunk c+1 is in flight (registers) during GEMM 2
// of chunk c; w1 fragments of chunk c are requested before GEMM 1 of chunk c, w3 fragments of chunk c+1 before GEMM 2 of chunk c.
// Same arithmetic (MFMA K order, f32 epilogue order, bf16 rounding points) as pw_chain_kernel: bit-identical results.
template <int C, int C4, int CN, int C2 = 0>
struct PwRt4 {
    static constexpr int BM = 128, CH = 128, NCH = C4 / CH;
    static constexpr int A_LD = C + 8, A2_LD = C2 + 8, Y_LD = CH + 8, O_LD = CN + 8;
    static constexpr int KF1S = C2 / 16;                                // shortcut-conv fragments per column tile (projection blocks)
    static constexpr int KF1 = C / 16, KF2 = CH / 16;                   // weight fragments per column tile: expand conv / one K-slice of conv1
    static constexpr int NR2 = CN >= 128 ? 4 : 2;                       // row tiles per wave in GEMM 2 (CN = 64: two column tiles x two row pairs)
    static constexpr size_t BYTES = 2 * (size_t)(BM * A_LD + (C2 ? BM * A2_LD : 0) + BM * Y_LD) + (C2 ? 4 : 2) * C4 * sizeof(float);   // + bn scale / shift (f32)
    static_assert(C % 16 == 0 && C4 % CH == 0 && (CN == 0 || CN == 64 || CN == 128), "rt4 tail: shapes");
    static_assert(CN == 0 || BM * O_LD <= BM * Y_LD, "the a' staging tile aliases the y chunk");
};

// EARLY = how many of the 8 residual loads of chunk c+1 go out BEFORE GEMM 1 of chunk c.
// C2 > 0: PROJECTION block with a same-resolution source (res2.0: x2 = the stem output): the shortcut bn_sc(Wsc x2) is a second GEMM
// of every chunk out of a second LDS operand tile, kept in f32 until it is added (like pw_chain_kernel); no residual is read.
template <int C, int C4, int CN, int EARLY, int C2 = 0>
__global__ __launch_bounds__(256, 2) void pw_chain_rt4_kernel(const PwArgs p) {
    typedef PwRt4<C, C4, CN, C2> S;
    constexpr int BM = S::BM, CH = S::CH, NCH = S::NCH, A_LD = S::A_LD, Y_LD = S::Y_LD, O_LD = S::O_LD, KF1 = S::KF1, KF2 = S::KF2, NR2 = S::NR2;
    constexpr int RCH = BM * (CH / 8) / 256;                 // 16-byte chunks per thread of one [BM][CH] tile (8)
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    bf16_t* A1 = reinterpret_cast<bf16_t*>(pw_smem);
    bf16_t* A2 = A1 + BM * A_LD;
    bf16_t* Y = A2 + (C2 ? BM * S::A2_LD : 0);
    bf16_t* O = Y;                                           // aliases the y chunk once the last chunk has been consumed
    // bn3 scale / shift of all C4 channels, parked once: the chunk epilogues read them with ds_read_b128 (as global loads they were
    // eight exposed L2 round trips per chunk, each behind every older load of the wave: vmcnt retires in order)
    float* S3 = reinterpret_cast<float*>(Y + BM * Y_LD);
    float* B3 = S3 + C4;
    float* SSC = B3 + C4;                                    // (projection blocks only)
    float* BSC = SSC + C4;
    constexpr int KF1S = S::KF1S, A2_LD = S::A2_LD;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * BM;
    // GEMM 2 ownership: CN >= 128: column tile = wave, all four row tiles; CN = 64: column tile wave & 1, row tiles 2 * (wave >> 1) + {0, 1}
    // Channel order inside a 32-channel column tile: MFMA row i = 8q + 4h + e of the accumulator layout (lane half h holds rows
    // 8q + 4h + {0..3}, q = 0..3) is fed with the weights of channel 16h + 4q + e, so a lane's 16 accumulators are 16 CONSECUTIVE
    // channels: the epilogues move 16-byte pieces (ds_read_b128 / ds_write_b128 at a 4-dword row skew: conflict-free) instead of 8-byte
    // ones (rows l and l + 16 of a 32-lane group on one bank: two-way conflicts, profiles/r4_pmc_res2_tail.json 28.7 %).  The weight
    // matrices stay in the plain fragment-major order: a lane just loads another lane's 16 bytes of the same 1 KB fragment.
    const int wl = half * 32 + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
    const int nt2 = CN >= 128 ? wave : (wave & 1);
    const int r2 = CN >= 128 ? 0 : 2 * (wave >> 1);

    bf16x8 wa[KF1], wb[CN ? KF2 : 1], ws[C2 ? KF1S : 1];
    auto load_wa = [&](int c) {                              // expand-conv (+ shortcut-conv) fragments of y channels c*128 + wave*32 .. +32
        const bf16_t* w = p.w3 + ((long long)((c * (CH / 32) + wave) * KF1) * 64 + wl) * 8;
#pragma unroll
        for (int kk = 0; kk < KF1; ++kk) wa[kk] = *reinterpret_cast<const bf16x8*>(w + kk * 512);
        if constexpr (C2 > 0) {
            const bf16_t* v = p.wsc + ((long long)((c * (CH / 32) + wave) * KF1S) * 64 + wl) * 8;
#pragma unroll
            for (int kk = 0; kk < KF1S; ++kk) ws[kk] = *reinterpret_cast<const bf16x8*>(v + kk * 512);
        }
    };
    auto load_wb = [&](int c) {                              // conv1 fragments of this wave's column tile, K-slice c
        if constexpr (CN > 0) {
            const bf16_t* w = p.w1 + ((long long)(nt2 * (C4 / 16) + c * KF2) * 64 + wl) * 8;
#pragma unroll
            for (int kk = 0; kk < KF2; ++kk) wb[kk] = *reinterpret_cast<const bf16x8*>(w + kk * 512);
        }
    };
    // tile-relative addressing: wave-uniform 64-bit bases + ONE 32-bit per-thread offset (row tid >> 4, 16-byte column tid & 15); the
    // eight rows a thread touches are 16 rows apart = compile-time byte distances (64-bit per-row addresses, hoisted out of the chunk
    // loop, were 50 registers: spills)
    const int trow = tid >> 4;
    const unsigned toff = (unsigned)(trow * C4 + (tid & 15) * 8) * 2u;                  // bytes
    // y store: ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32); with the 4-dword row skew
    // a group is conflict-free when its lanes read ONE row (16 x 16 bytes = all 64 banks), so the group index - not lane >> 4 - picks
    // the row of a pair
    const int srow = (tid >> 5) * 2 + (((lane >> 2) ^ (lane >> 3) ^ (lane >> 4)) & 1);
    const unsigned soff = (unsigned)(srow * C4 + (tid & 15) * 8) * 2u;
    const char* res_b = reinterpret_cast<const char*>(C2 ? p.y : p.res) + m0 * C4 * 2;   // (never read in a projection block)
    char* y_b = reinterpret_cast<char*>(p.y + m0 * C4);
    us8 rres[RCH];
    auto fetch_res = [&](int c, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            if (i >= i0 && i < i1) rres[i] = *reinterpret_cast<const us8*>(res_b + c * (CH * 2) + (toff + (unsigned)(i * 16 * C4 * 2)));
        }
    };
    auto park_res = [&]() {
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            const int ch = tid + i * 256, row = ch >> 4, col = (ch & 15) * 8;
            *reinterpret_cast<us8*>(Y + row * Y_LD + col) = rres[i];
        }
    };
    // ---- prologue: every global read goes out first (b tile, residual chunk 0, the first weight fragments)
    {
        // (the bn scale / shift loads FIRST: loads return in issue order, so nothing parked behind them waits for the big operand loads)
        constexpr int NSB = (C4 / 4 + 256 - 1) / 256;
        f32x4 sbv[NSB][C2 > 0 ? 4 : 2];
#pragma unroll
        for (int it = 0; it < NSB; ++it) {
            const int i = tid + it * 256 < C4 / 4 ? tid + it * 256 : C4 / 4 - 1;
            sbv[it][0] = *reinterpret_cast<const f32x4*>(p.s3 + 4 * i);
            sbv[it][1] = *reinterpret_cast<const f32x4*>(p.b3 + 4 * i);
            if constexpr (C2 > 0) {
                sbv[it][2] = *reinterpret_cast<const f32x4*>(p.ssc + 4 * i);
                sbv[it][3] = *reinterpret_cast<const f32x4*>(p.bsc + 4 * i);
            }
        }
        constexpr int CPR = C / 8, NB = BM * CPR / 256;
        us8 rb[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int ch = tid + i * 256, row = ch / CPR, col = (ch % CPR) * 8;
            rb[i] = *reinterpret_cast<const us8*>(p.a1 + (m0 + row) * C + col);
        }
        constexpr int CPR2 = C2 ? C2 / 8 : 1, NB2 = C2 ? BM * CPR2 / 256 : 1;
        us8 rb2[NB2];
        if constexpr (C2 > 0) {
#pragma unroll
            for (int i = 0; i < NB2; ++i) {
                const int ch = tid + i * 256, row = ch / CPR2, col = (ch % CPR2) * 8;
                rb2[i] = *reinterpret_cast<const us8*>(p.a2 + (m0 + row) * C2 + col);
            }
        } else {
            fetch_res(0, 0, RCH);
        }
        load_wa(0);
        // bn scale / shift -> LDS.  Round 6 (in-kernel stamps + ISA): as a guarded loop (`for i = tid; i < C4 / 4`) this compiled to a branch
        // with an s_waitcnt vmcnt(0) behind EACH of its loads - every operand and weight load issued above had to land first, then two to
        // four more round trips followed one after the other: a third of the projection tail's workgroup time.  Now: unconditional loads
        // (index clamped), only the LDS writes are guarded.
        {
#pragma unroll
            for (int it = 0; it < NSB; ++it) {
                const int i = tid + it * 256;
                if (i < C4 / 4) {
                    *reinterpret_cast<f32x4*>(S3 + 4 * i) = sbv[it][0];
                    *reinterpret_cast<f32x4*>(B3 + 4 * i) = sbv[it][1];
                    if constexpr (C2 > 0) {
                        *reinterpret_cast<f32x4*>(SSC + 4 * i) = sbv[it][2];
                        *reinterpret_cast<f32x4*>(BSC + 4 * i) = sbv[it][3];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int ch = tid + i * 256, row = ch / CPR, col = (ch % CPR) * 8;
            *reinterpret_cast<us8*>(A1 + row * A_LD + col) = rb[i];
        }
        if constexpr (C2 > 0) {
#pragma unroll
            for (int i = 0; i < NB2; ++i) {
                const int ch = tid + i * 256, row = ch / CPR2, col = (ch % CPR2) * 8;
                *reinterpret_cast<us8*>(A2 + row * A2_LD + col) = rb2[i];
            }
        } else {
            park_res();
        }
    }
    __syncthreads();

    f32x16 acc2[CN ? NR2 : 1];
#pragma unroll
    for (int r = 0; r < (CN ? NR2 : 1); ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[r][e] = 0.f;
    // (fully unrolled: a run-time `c + 1 < NCH` around the prefetches is a branch the s_waitcnt pass cannot count loads across)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        load_wb(c);
        if (C2 == 0 && EARLY > 0 && c + 1 < NCH) fetch_res(c + 1, 0, EARLY);   // in flight during BOTH GEMMs of chunk c (4 live registers each in GEMM 1)
        // ---- GEMM 1: y chunk c, this wave's 32 channels x 128 pixels, as two passes over 64 pixels (the weight fragments stay
        //      in registers for both: half the accumulator registers, no extra weight traffic)
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
            f32x16 acc[2], accs[C2 ? 2 : 1];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc[r][e] = 0.f; if (C2 > 0) accs[r][e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < KF1; ++kk)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(A1 + ((2 * rp + r) * 32 + l31) * A_LD + kk * 16 + half * 8);
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kk], af, acc[r], 0, 0, 0);
                    if (r == 1 && (kk & 1)) __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting every k-step's LDS reads (spills)
                }
            if constexpr (C2 > 0) {
#pragma unroll
                for (int kk = 0; kk < KF1S; ++kk)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(A2 + ((2 * rp + r) * 32 + l31) * A2_LD + kk * 16 + half * 8);
                        accs[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws[kk], af, accs[r], 0, 0, 0);
                        if (r == 1 && (kk & 1)) __builtin_amdgcn_sched_barrier(0);
                    }
            }
            // lane holds, for pixel (2 rp + r)*32 + l31, channels wave*32 + 16*half + 4q + {0..3} of the chunk (q = 0..3)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int nl = wave * 32 + 16 * half + 8 * qq, n = c * CH + nl;
                f32x4 s3[2], b3[2], ss[2], bs[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    s3[j] = *reinterpret_cast<const f32x4*>(S3 + n + 4 * j); b3[j] = *reinterpret_cast<const f32x4*>(B3 + n + 4 * j);
                    if constexpr (C2 > 0) { ss[j] = *reinterpret_cast<const f32x4*>(SSC + n + 4 * j); bs[j] = *reinterpret_cast<const f32x4*>(BSC + n + 4 * j); }
                }
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    bf16_t* yp = Y + ((2 * rp + r) * 32 + l31) * Y_LD + nl;
                    us8 r8 = {0, 0, 0, 0, 0, 0, 0, 0};
                    if constexpr (C2 == 0) r8 = *reinterpret_cast<const us8*>(yp);
                    us8 o8;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[r][4 * (2 * qq + j) + e] * s3[j][e];
                            v += b3[j][e];
                            if constexpr (C2 > 0) {
                                float sc = accs[r][4 * (2 * qq + j) + e] * ss[j][e];
                                sc += bs[j][e];
                                v += sc;
                            } else {
                                v += bf16_to_f32(r8[4 * j + e]);
                            }
                            o8[4 * j + e] = f32_to_bf16(v > 0.f ? v : 0.f);
                        }
                    *reinterpret_cast<us8*>(yp) = o8;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                     // y chunk complete
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            const us8 v = *reinterpret_cast<const us8*>(Y + (srow + 16 * i) * Y_LD + (tid & 15) * 8);
            *reinterpret_cast<us8*>(y_b + c * (CH * 2) + (soff + (unsigned)(i * 16 * C4 * 2))) = v;
        }
        __builtin_amdgcn_sched_barrier(0);                   // (the store's staging registers are dead before the prefetches below go live)
        if (c + 1 < NCH) {
            load_wa(c + 1);                                  // (wa is dead until the next chunk's GEMM 1)
            if (C2 == 0 && EARLY < RCH) fetch_res(c + 1, EARLY, RCH);   // the rest: into the registers the GEMM 1 accumulators just left; lands during GEMM 2
        }
        if constexpr (CN > 0) {
            // ---- GEMM 2: a' += y_c W1'[:, chunk c]^T
#pragma unroll
            for (int kk = 0; kk < KF2; ++kk)
#pragma unroll
                for (int r = 0; r < NR2; ++r) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(Y + ((r2 + r) * 32 + l31) * Y_LD + kk * 16 + half * 8);
                    acc2[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[kk], af, acc2[r], 0, 0, 0);
                    if (r == NR2 - 1 && (kk & 1)) __builtin_amdgcn_sched_barrier(0);
                }
        }
        __syncthreads();                                     // every wave is done with chunk c
        if (C2 == 0 && c + 1 < NCH) {
            park_res();
            __syncthreads();
        }
    }
    if constexpr (CN > 0) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int n = nt2 * 32 + 16 * half + 8 * qq;
            f32x4 s1[2], b1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { s1[j] = *reinterpret_cast<const f32x4*>(p.s1 + n + 4 * j); b1[j] = *reinterpret_cast<const f32x4*>(p.b1 + n + 4 * j); }
#pragma unroll
            for (int r = 0; r < NR2; ++r) {
                us8 o8;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc2[r][4 * (2 * qq + j) + e] * s1[j][e];
                        v += b1[j][e];
                        o8[4 * j + e] = f32_to_bf16(v > 0.f ? v : 0.f);
                    }
                *reinterpret_cast<us8*>(O + ((r2 + r) * 32 + l31) * O_LD + n) = o8;
            }
        }
        __syncthreads();
        constexpr int CPR = CN / 8, RPI = 256 / CPR;          // 16-byte chunks per row, rows per pass of the 256 threads
        static_assert(CN == 0 || BM * CPR % 256 == 0, "a' tile chunking");
        const int orow = tid / CPR, ocol = tid % CPR;
        unsigned char* o_b = reinterpret_cast<unsigned char*>(p.o) + m0 * CN * (p.o_fp8 ? 1 : 2);
#pragma unroll
        for (int i = 0; i < BM * CPR / 256; ++i) {
            const int row = orow + i * RPI;
            const us8 v = *reinterpret_cast<const us8*>(O + row * O_LD + ocol * 8);
            if (p.o_fp8) {                                   // e4m3fn bytes for an fp8 conv2 (saturating RNE)
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = bf16_to_f32(v[e]);
                *reinterpret_cast<uint2*>(o_b + (unsigned)(row * CN + ocol * 8)) = f32x8_to_fp8(f);
            } else {
                *reinterpret_cast<us8*>(o_b + (unsigned)(row * CN + ocol * 8) * 2u) = v;
            }
        }
    }
}

template <int C, int C4, int CN, int C2>
static int pw_launch_rt4_proj(const PwArgs& a, hipStream_t stream) {
    constexpr size_t lds = PwRt4<C, C4, CN, C2>::BYTES;
    NPS_ENSURE_LDS((int)lds, pw_chain_rt4_kernel<C, C4, CN, 0, C2>);
    hipLaunchKernelGGL((pw_chain_rt4_kernel<C, C4, CN, 0, C2>), dim3((unsigned)((a.M + 127) / 128)), dim3(256), lds, stream, a);
    return 0;
}

template <int C, int C4, int CN>
static int pw_launch_rt4(const PwArgs& a, hipStream_t stream) {
    constexpr size_t lds = PwRt4<C, C4, CN>::BYTES;
    static const bool late = getenv("NOPESAC_TAIL_RT4_LATE") != nullptr;       // A/B switch: residual prefetch behind GEMM 1
    // measured (profiles/r3_c_tail_ab.txt): res2 (C = 64) gains 3-6 % from the residual prefetch going out before GEMM 1 (308 vs 326 us,
    // 361 vs 371 us); res3 (C = 128, 248 registers without it) loses 2 % even with only three of the eight loads early (197 vs 192 us)
    constexpr int EARLY = C >= 128 ? 0 : 8;
    if (late) {
        NPS_ENSURE_LDS((int)lds, pw_chain_rt4_kernel<C, C4, CN, 0>);
        hipLaunchKernelGGL((pw_chain_rt4_kernel<C, C4, CN, 0>), dim3((unsigned)((a.M + 127) / 128)), dim3(256), lds, stream, a);
    } else {
        NPS_ENSURE_LDS((int)lds, pw_chain_rt4_kernel<C, C4, CN, EARLY>);
        hipLaunchKernelGGL((pw_chain_rt4_kernel<C, C4, CN, EARLY>), dim3((unsigned)((a.M + 127) / 128)), dim3(256), lds, stream, a);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Eight-wave form of the 128-pixel chunked tail for the EDGE blocks of res3, which the four-wave kernel cannot hold:
//   * res3.3 -> res4.0 (CN = 256): the next conv1 has eight 32-channel column tiles - with four waves that is 2 x 4 accumulators of
//     16 registers per wave next to everything else;
//   * res3.0 (projection shortcut, K = C2 = 256, from the stride-2 sampled res2 output): a second operand tile of 68 KB and sixteen
//     more weight fragments per column tile.
// One 512-thread workgroup per CU (137 KB of LDS in the projection form), two waves per SIMD.  GEMM 1 of a chunk: wave w owns column
// tile w & 3 and the 64-pixel half w >> 2, one 32-pixel row tile at a time (the weight fragments stay in registers for both; the two
// waves that share a column tile read the same fragments: L1 hits).  GEMM 2: CN = 256: column tile w, all four row tiles; CN = 128:
// column tile w & 3, pixel half w >> 2.  Otherwise pw_chain_rt4_kernel's scheme: y chunk in place over its residual (identity form),
// bn scale / shift in LDS, full tiles only, fully unrolled chunk loop, same arithmetic and rounding points as pw_chain_kernel.
template <int C, int C4, int CN, int C2>
struct PwRt8 {
    static constexpr int BM = 128, CH = 128, NCH = C4 / CH;
    static constexpr int A_LD = C + 8, A2_LD = C2 + 8, Y_LD = CH + 8, O_LD = CN + 8;
    static constexpr int KF1 = C / 16, KF1S = C2 / 16, KF2 = CH / 16;
    static constexpr int NR2 = CN >= 256 ? 4 : 2;
    static constexpr int OPER = BM * A_LD + (C2 ? BM * A2_LD : 0) + BM * Y_LD;      // bf16 elements of the operand tiles
    static constexpr size_t BYTES = 2 * (size_t)OPER + (C2 ? 4 : 2) * C4 * sizeof(float);
    static_assert(C % 16 == 0 && C4 % CH == 0 && (CN == 128 || CN == 256) && C2 % 16 == 0, "rt8 tail: shapes");
    static_assert(BM * O_LD <= (C2 ? BM * Y_LD : BM * A_LD + BM * Y_LD), "the a' staging tile aliases dead operand tiles");
    static_assert(BYTES <= 160 * 1024, "LDS");
};

// (C = 256 - the res4 form, round 5 experiment - holds 16 + 8 weight fragments per lane: one workgroup per CU, 256 registers)
// STAMP: tuning build - cycle stamps of every (workgroup, wave) at the phase boundaries into dbg[workgroup][8 waves][24] (scripts/rt8_stamps.py)
template <int C, int C4, int CN, int C2, bool STAMP = false>
__global__ __launch_bounds__(512, (C >= 256 ? 1 : 2)) void pw_chain_rt8_kernel(const PwArgs p, unsigned long long* dbg = nullptr) {
    unsigned long long ts[24];
    auto stamp = [&](int i) { if constexpr (STAMP) ts[i] = __builtin_readcyclecounter(); };
    stamp(0);
    typedef PwRt8<C, C4, CN, C2> S;
    constexpr int BM = S::BM, CH = S::CH, NCH = S::NCH, A_LD = S::A_LD, A2_LD = S::A2_LD, Y_LD = S::Y_LD, O_LD = S::O_LD;
    constexpr int KF1 = S::KF1, KF1S = S::KF1S, KF2 = S::KF2, NR2 = S::NR2;
    constexpr int RCH = BM * (CH / 8) / 512;                 // 16-byte chunks per thread of one [BM][CH] tile (4)
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    bf16_t* A1 = reinterpret_cast<bf16_t*>(pw_smem);
    bf16_t* A2 = A1 + BM * A_LD;
    bf16_t* Y = A2 + (C2 ? BM * A2_LD : 0);
    bf16_t* O = C2 ? Y : A1;                                 // a' staging: over the y chunk (CN = 128) or over b tile + y chunk (CN = 256)
    float* S3 = reinterpret_cast<float*>(A1 + S::OPER);
    float* B3 = S3 + C4;
    float* SSC = B3 + C4;
    float* BSC = SSC + C4;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * BM;
    // (pw_chain_rt4_kernel's channel order: MFMA row 8q + 4h + e carries channel 16h + 4q + e of the column tile - 16-byte epilogues)
    const int wl = half * 32 + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);
    const int ct = wave & 3, rh = wave >> 2;                 // GEMM 1: column tile of the chunk, 64-pixel half
    const int nt2 = CN >= 256 ? wave : (wave & 3);           // GEMM 2: column tile of a'
    const int r2 = CN >= 256 ? 0 : 2 * (wave >> 2);          //         first row tile

    bf16x8 wa[KF1], wb[KF2], ws[C2 ? KF1S : 1];
    auto load_wa = [&](int c) {
        const bf16_t* w = p.w3 + ((long long)((c * (CH / 32) + ct) * KF1) * 64 + wl) * 8;
#pragma unroll
        for (int kk = 0; kk < KF1; ++kk) wa[kk] = *reinterpret_cast<const bf16x8*>(w + kk * 512);
        if constexpr (C2 > 0) {
            const bf16_t* v = p.wsc + ((long long)((c * (CH / 32) + ct) * KF1S) * 64 + wl) * 8;
#pragma unroll
            for (int kk = 0; kk < KF1S; ++kk) ws[kk] = *reinterpret_cast<const bf16x8*>(v + kk * 512);
        }
    };
    auto load_wb = [&](int c) {
        const bf16_t* w = p.w1 + ((long long)(nt2 * (C4 / 16) + c * KF2) * 64 + wl) * 8;
#pragma unroll
        for (int kk = 0; kk < KF2; ++kk) wb[kk] = *reinterpret_cast<const bf16x8*>(w + kk * 512);
    };
    const int trow = tid >> 4;                               // 32 rows per pass of the 512 threads, 16-byte column tid & 15
    const unsigned toff = (unsigned)(trow * C4 + (tid & 15) * 8) * 2u;
    const int srow = (tid >> 5) * 2 + (((lane >> 2) ^ (lane >> 3) ^ (lane >> 4)) & 1);   // y store: one row per ds_read_b128 lane group
    const unsigned soff = (unsigned)(srow * C4 + (tid & 15) * 8) * 2u;
    const char* res_b = reinterpret_cast<const char*>(C2 ? p.y : p.res) + m0 * C4 * 2;       // (never read in the projection form)
    char* y_b = reinterpret_cast<char*>(p.y + m0 * C4);
    us8 rres[RCH];
    auto fetch_res = [&](int c) {
#pragma unroll
        for (int i = 0; i < RCH; ++i) rres[i] = *reinterpret_cast<const us8*>(res_b + c * (CH * 2) + (toff + (unsigned)(i * 32 * C4 * 2)));
    };
    auto park_res = [&]() {
#pragma unroll
        for (int i = 0; i < RCH; ++i) *reinterpret_cast<us8*>(Y + (trow + 32 * i) * Y_LD + (tid & 15) * 8) = rres[i];
    };
    // ---- prologue
    {
        // (the bn scale / shift loads FIRST: loads return in issue order, so nothing parked behind them waits for the big operand loads)
        constexpr int NSB = (C4 / 4 + 512 - 1) / 512;
        f32x4 sbv[NSB][C2 > 0 ? 4 : 2];
#pragma unroll
        for (int it = 0; it < NSB; ++it) {
            const int i = tid + it * 512 < C4 / 4 ? tid + it * 512 : C4 / 4 - 1;
            sbv[it][0] = *reinterpret_cast<const f32x4*>(p.s3 + 4 * i);
            sbv[it][1] = *reinterpret_cast<const f32x4*>(p.b3 + 4 * i);
            if constexpr (C2 > 0) {
                sbv[it][2] = *reinterpret_cast<const f32x4*>(p.ssc + 4 * i);
                sbv[it][3] = *reinterpret_cast<const f32x4*>(p.bsc + 4 * i);
            }
        }
        constexpr int CPR = C / 8, NB = BM * CPR / 512;
        us8 rb[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int ch = tid + i * 512, row = ch / CPR, col = (ch % CPR) * 8;
            rb[i] = *reinterpret_cast<const us8*>(p.a1 + (m0 + row) * C + col);
        }
        constexpr int CPR2 = C2 ? C2 / 8 : 1, NB2 = C2 ? BM * CPR2 / 512 : 1;
        us8 rb2[NB2];
        if constexpr (C2 > 0) {
            // projection source: pixel m = (b, oy, ox) reads x2[b][oy * s][ox * s] (a 2 C2-byte row each).  Round 6: the thread's rows are
            // (tid / CPR2) + RSTEP i - ONE (image, row, column) decomposition per thread, then a step of RSTEP pixels with carries (the
            // per-load 64-bit m / per, m % per, rem / OW were three software divisions per load: ~1000 VALU instructions per thread in
            // front of the first load of a one-workgroup-per-CU kernel)
            static_assert(512 % CPR2 == 0, "a thread's rows are RSTEP apart");
            constexpr int RSTEP = 512 / CPR2;
            const int per = p.OH * p.OW;
            const int col = (tid % CPR2) * 8;
            const long long mq = m0 + tid / CPR2;
            int bb = (int)(mq / per), rem = (int)(mq - (long long)bb * per), oy = rem / p.OW, ox = rem - oy * p.OW;
#pragma unroll
            for (int i = 0; i < NB2; ++i) {
                const long long pix = ((long long)bb * p.a2_H + oy * p.a2_stride) * p.a2_W + ox * p.a2_stride;
                rb2[i] = *reinterpret_cast<const us8*>(p.a2 + pix * C2 + col);
                ox += RSTEP;
                while (ox >= p.OW) { ox -= p.OW; ++oy; }
                if (oy >= p.OH) { oy -= p.OH; ++bb; }
            }
        } else {
            fetch_res(0);
        }
        load_wa(0);
        stamp(1);
        // bn scale / shift -> LDS.  Round 6 (in-kernel stamps + ISA): as a guarded loop (`for i = tid; i < C4 / 4`) this compiled to a branch
        // with an s_waitcnt vmcnt(0) behind EACH of its loads - every operand and weight load issued above had to land first, then two to
        // four more round trips followed one after the other: a third of the projection tail's workgroup time.  Now: unconditional loads
        // (index clamped), only the LDS writes are guarded.
        {
#pragma unroll
            for (int it = 0; it < NSB; ++it) {
                const int i = tid + it * 512;
                if (i < C4 / 4) {
                    *reinterpret_cast<f32x4*>(S3 + 4 * i) = sbv[it][0];
                    *reinterpret_cast<f32x4*>(B3 + 4 * i) = sbv[it][1];
                    if constexpr (C2 > 0) {
                        *reinterpret_cast<f32x4*>(SSC + 4 * i) = sbv[it][2];
                        *reinterpret_cast<f32x4*>(BSC + 4 * i) = sbv[it][3];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int ch = tid + i * 512, row = ch / CPR, col = (ch % CPR) * 8;
            *reinterpret_cast<us8*>(A1 + row * A_LD + col) = rb[i];
        }
        if constexpr (C2 > 0) {
#pragma unroll
            for (int i = 0; i < NB2; ++i) {
                const int ch = tid + i * 512, row = ch / CPR2, col = (ch % CPR2) * 8;
                *reinterpret_cast<us8*>(A2 + row * A2_LD + col) = rb2[i];
            }
        } else {
            park_res();
        }
    }
    stamp(2);
    __syncthreads();
    stamp(3);

    f32x16 acc2[NR2];
#pragma unroll
    for (int r = 0; r < NR2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[r][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        load_wb(c);
        // ---- GEMM 1: this wave's 32 channels x 64 pixels of y chunk c, one 32-pixel row tile at a time
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int rt = 2 * rh + r;
            f32x16 acc, accs;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accs[e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < KF1; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(A1 + (rt * 32 + l31) * A_LD + kk * 16 + half * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kk], af, acc, 0, 0, 0);
                if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (C2 > 0) {
#pragma unroll
                for (int kk = 0; kk < KF1S; ++kk) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(A2 + (rt * 32 + l31) * A2_LD + kk * 16 + half * 8);
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws[kk], af, accs, 0, 0, 0);
                    if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            // round 6: the NEXT chunk's expand / shortcut fragments go out here, behind this chunk's last GEMM-1 MFMA (their registers are
            // dead from here on) - they used to be requested behind the y store, with only GEMM 2's 16 MFMAs of cover for an L2 round trip
            if (r == 1 && c + 1 < NCH) {
                __builtin_amdgcn_sched_barrier(0);
                load_wa(c + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int nl = ct * 32 + 16 * half + 8 * qq, n = c * CH + nl;
                bf16_t* yp = Y + (rt * 32 + l31) * Y_LD + nl;
                us8 r8 = {0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (C2 == 0) r8 = *reinterpret_cast<const us8*>(yp);
                us8 o8;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 s3 = *reinterpret_cast<const f32x4*>(S3 + n + 4 * j), b3 = *reinterpret_cast<const f32x4*>(B3 + n + 4 * j);
                    f32x4 ss = {0.f, 0.f, 0.f, 0.f}, bs = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (C2 > 0) { ss = *reinterpret_cast<const f32x4*>(SSC + n + 4 * j); bs = *reinterpret_cast<const f32x4*>(BSC + n + 4 * j); }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[4 * (2 * qq + j) + e] * s3[e];
                        v += b3[e];
                        if constexpr (C2 > 0) {
                            float sc = accs[4 * (2 * qq + j) + e] * ss[e];
                            sc += bs[e];
                            v += sc;
                        } else {
                            v += bf16_to_f32(r8[4 * j + e]);
                        }
                        o8[4 * j + e] = f32_to_bf16(v > 0.f ? v : 0.f);
                    }
                    if (C2 > 0) __builtin_amdgcn_sched_barrier(0);
                }
                *reinterpret_cast<us8*>(yp) = o8;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp(4 + 4 * c);
        __syncthreads();                                     // y chunk complete
        stamp(5 + 4 * c);
#pragma unroll
        for (int i = 0; i < RCH; ++i) {
            const us8 v = *reinterpret_cast<const us8*>(Y + (srow + 32 * i) * Y_LD + (tid & 15) * 8);
            *reinterpret_cast<us8*>(y_b + c * (CH * 2) + (soff + (unsigned)(i * 32 * C4 * 2))) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (C2 == 0 && c + 1 < NCH) fetch_res(c + 1);
        // ---- GEMM 2: a' += y_c W1'[:, chunk c]^T
#pragma unroll
        for (int kk = 0; kk < KF2; ++kk)
#pragma unroll
            for (int r = 0; r < NR2; ++r) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(Y + ((r2 + r) * 32 + l31) * Y_LD + kk * 16 + half * 8);
                acc2[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[kk], af, acc2[r], 0, 0, 0);
                if (r == NR2 - 1 && (kk & 1)) __builtin_amdgcn_sched_barrier(0);
            }
        stamp(6 + 4 * c);
        __syncthreads();                                     // every wave is done with chunk c
        if (C2 == 0 && c + 1 < NCH) {
            park_res();
            __syncthreads();
        }
        stamp(7 + 4 * c);
    }
    // ---- a' = relu(bn1(acc2)) -> LDS -> whole rows
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int n = nt2 * 32 + 16 * half + 8 * qq;
        f32x4 s1[2], b1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { s1[j] = *reinterpret_cast<const f32x4*>(p.s1 + n + 4 * j); b1[j] = *reinterpret_cast<const f32x4*>(p.b1 + n + 4 * j); }
#pragma unroll
        for (int r = 0; r < NR2; ++r) {
            us8 o8;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc2[r][4 * (2 * qq + j) + e] * s1[j][e];
                    v += b1[j][e];
                    o8[4 * j + e] = f32_to_bf16(v > 0.f ? v : 0.f);
                }
            *reinterpret_cast<us8*>(O + ((r2 + r) * 32 + l31) * O_LD + n) = o8;
        }
    }
    __syncthreads();
    constexpr int CPRO = CN / 8, RPI = 512 / CPRO;
    static_assert(BM * CPRO % 512 == 0, "a' tile chunking");
    const int orow = tid / CPRO, ocol = tid % CPRO;
    unsigned char* o_b = reinterpret_cast<unsigned char*>(p.o) + m0 * CN * (p.o_fp8 ? 1 : 2);
#pragma unroll
    for (int i = 0; i < BM * CPRO / 512; ++i) {
        const int row = orow + i * RPI;
        const us8 v = *reinterpret_cast<const us8*>(O + row * O_LD + ocol * 8);
        if (p.o_fp8) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = bf16_to_f32(v[e]);
            *reinterpret_cast<uint2*>(o_b + (unsigned)(row * CN + ocol * 8)) = f32x8_to_fp8(f);
        } else {
            *reinterpret_cast<us8*>(o_b + (unsigned)(row * CN + ocol * 8) * 2u) = v;
        }
    }
    if constexpr (STAMP) {
        stamp(4 + 4 * NCH);
        if (dbg && lane == 0)
            for (int i = 0; i < 5 + 4 * NCH; ++i) dbg[((long long)blockIdx.x * 8 + wave) * 24 + i] = ts[i];
    }
}

static unsigned long long* g_rt8_dbg = nullptr;

// ------------------------------------------------------------------------------------------------------------------------
// Round 6: the edge tails of res3 as FOUR-wave workgroups of 64 pixels, two per CU (pw_chain_rt4h_kernel).  In-kernel stamps of the
// eight-wave kernel above (scripts/rt8_stamps.py, profiles/r6_g_*): 56 k cycles per 128-pixel workgroup of the projection block for
// 16 k cycles of MFMA work; a third of it is the prologue - 96 KB of operands per workgroup, requested by all 256 single-resident
// workgroups AT ONCE (2 TB/s over the launch; the two-per-CU identity tails stream at 4.2-5.1) - and nothing runs on the CU meanwhile.
// (Not the L2 -> CU weight stream: a 96-pixel / 256-channel-chunk form that loads every fragment once instead of twice measured the
// same 310 us and was dropped.)  The remedy is the identity tails': TWO workgroups per CU, so that one's loads and stores run under the
// other's MFMAs - which takes <= 80 KB of LDS and <= 256 registers at 8 waves per CU: 64 pixels, four waves, 128-channel chunks = one
// column tile per wave (b tile 17 KB + projection source 34 KB + y chunk 17 KB + bn vectors 8 KB = 77 KB in the projection form).
// GEMM 2: CN = 128: column tile `wave`; CN = 256: column tiles wave and wave + 4.  Same operands, same K order in every accumulator,
// same multiply-then-add epilogues as the kernels above: bit-identical results.
template <int C, int C4, int CN, int C2>
struct PwRt4h {
    static constexpr int BM = 64, CH = 128, NCH = C4 / CH, NRT = BM / 32, NW = 4, CPT = CN / 32 / NW;
    static constexpr int A_LD = C + 8, A2_LD = C2 + 8, Y_LD = CH + 8, O_LD = CN + 8;
    static constexpr int KF1 = C / 16, KF1S = C2 / 16, KF2 = CH / 16;
    static constexpr int OPER = BM * A_LD + (C2 ? BM * A2_LD : 0) + BM * Y_LD;
    static constexpr size_t BYTES = 2 * (size_t)OPER + (C2 ? 4 : 2) * C4 * sizeof(float);
    static_assert(C % 16 == 0 && C4 % CH == 0 && (CN == 128 || CN == 256) && C2 % 16 == 0, "rt4h tail: shapes");
    static_assert(O_LD <= Y_LD || (C2 == 0 && BM * O_LD <= BM * A_LD + BM * Y_LD), "the a' staging tile aliases dead operand tiles");
    static_assert(2 * BYTES <= 160 * 1024, "two workgroups per CU");
};

template <int C, int C4, int CN, int C2>
__global__ __launch_bounds__(256, 2) void pw_chain_rt4h_kernel(const PwArgs p) {
    typedef PwRt4h<C, C4, CN, C2> S;
    constexpr int BM = S::BM, CH = S::CH, NCH = S::NCH, NRT = S::NRT, NW = S::NW, CPT = S::CPT;
    constexpr int A_LD = S::A_LD, A2_LD = S::A2_LD, Y_LD = S::Y_LD, O_LD = S::O_LD, KF1 = S::KF1, KF1S = S::KF1S, KF2 = S::KF2;
    constexpr int RCH = BM * (CH / 8) / 256;                 // 16-byte chunks per thread of one [BM][CH] tile (4)
    extern __shared__ __attribute__((aligned(16))) unsigned char pw_smem[];
    bf16_t* A1 = reinterpret_cast<bf16_t*>(pw_smem);
    bf16_t* A2 = A1 + BM * A_LD;
    bf16_t* Y = A2 + (C2 ? BM * A2_LD : 0);
    bf16_t* O = O_LD <= Y_LD ? Y : A1;                       // a' staging: over the y chunk, or (CN = 256, identity form) over b tile + y chunk
    float* S3 = reinterpret_cast<float*>(A1 + S::OPER);
    float* B3 = S3 + C4;
    float* SSC = B3 + C4;
    float* BSC = SSC + C4;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * BM;
    // (the rt4 / rt8 channel order: MFMA row 8q + 4h + e carries channel 16h + 4q + e of the column tile - 16-byte epilogues)
    const int wl = half * 32 + 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3);

    bf16x8 wa[KF1], wb[CPT][KF2], ws[C2 ? KF1S : 1];
    auto load_wa = [&](int c) {                              // expand (+ shortcut) fragments of y channels c*128 + wave*32 .. +32
        const bf16_t* w = p.w3 + ((long long)((c * (CH / 32) + wave) * KF1) * 64 + wl) * 8;
#pragma unroll
        for (int kk = 0; kk < KF1; ++kk) wa[kk] = *reinterpret_cast<const bf16x8*>(w + kk * 512);
        if constexpr (C2 > 0) {
            const bf16_t* v = p.wsc + ((long long)((c * (CH / 32) + wave) * KF1S) * 64 + wl) * 8;
#pragma unroll
            for (int kk = 0; kk < KF1S; ++kk) ws[kk] = *reinterpret_cast<const bf16x8*>(v + kk * 512);
        }
    };
    auto load_wb = [&](int c) {                              // conv1 fragments of column tiles wave + 4 t, K-slice c
#pragma unroll
        for (int t = 0; t < CPT; ++t) {
            const bf16_t* w = p.w1 + ((long long)((wave + NW * t) * (C4 / 16) + c * KF2) * 64 + wl) * 8;
#pragma unroll
            for (int kk = 0; kk < KF2; ++kk) wb[t][kk] = *reinterpret_cast<const bf16x8*>(w + kk * 512);
        }
    };
    // [BM][CH] tile <-> threads: 16 rows per pass (16 16-byte chunks per row), RCH passes
    const int trow = tid >> 4, tcol = (tid & 15) * 8;
    const unsigned toff = (unsigned)(trow * C4 + tcol) * 2u;
    const char* res_b = reinterpret_cast<const char*>(C2 ? p.y : p.res) + m0 * C4 * 2;       // (never read in the projection form)
    char* y_b = reinterpret_cast<char*>(p.y + m0 * C4);
    us8 rres[RCH];
    auto fetch_res = [&](int c) {
#pragma unroll
        for (int i = 0; i < RCH; ++i) rres[i] = *reinterpret_cast<const us8*>(res_b + c * (CH * 2) + (toff + (unsigned)(i * 16 * C4 * 2)));
    };
    auto park_res = [&]() {
#pragma unroll
        for (int i = 0; i < RCH; ++i) *reinterpret_cast<us8*>(Y + (trow + 16 * i) * Y_LD + tcol) = rres[i];
    };
    // ---- prologue: bn vectors first (loads return in issue order), then the operand tiles, then the first weight fragments
    {
        constexpr int NSB = (C4 / 4 + 255) / 256;
        f32x4 sbv[NSB][C2 > 0 ? 4 : 2];
#pragma unroll
        for (int it = 0; it < NSB; ++it) {
            const int i = tid + it * 256 < C4 / 4 ? tid + it * 256 : C4 / 4 - 1;
            sbv[it][0] = *reinterpret_cast<const f32x4*>(p.s3 + 4 * i);
            sbv[it][1] = *reinterpret_cast<const f32x4*>(p.b3 + 4 * i);
            if constexpr (C2 > 0) {
                sbv[it][2] = *reinterpret_cast<const f32x4*>(p.ssc + 4 * i);
                sbv[it][3] = *reinterpret_cast<const f32x4*>(p.bsc + 4 * i);
            }
        }
        constexpr int CPR = C / 8, NB = BM * CPR / 256;
        static_assert(BM * CPR % 256 == 0, "b tile chunking");
        us8 rb[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int ch = tid + i * 256, row = ch / CPR, col = (ch % CPR) * 8;
            rb[i] = *reinterpret_cast<const us8*>(p.a1 + (m0 + row) * C + col);
        }
        constexpr int CPR2 = C2 ? C2 / 8 : 1, NB2 = C2 ? BM * CPR2 / 256 : 1;
        us8 rb2[NB2];
        if constexpr (C2 > 0) {
            static_assert(256 % CPR2 == 0 && BM * CPR2 % 256 == 0, "projection tile chunking");
            constexpr int RSTEP = 256 / CPR2;
            const int per = p.OH * p.OW;
            const int col = (tid % CPR2) * 8;
            const long long mq = m0 + tid / CPR2;
            int bb = (int)(mq / per), rem = (int)(mq - (long long)bb * per), oy = rem / p.OW, ox = rem - oy * p.OW;
#pragma unroll
            for (int i = 0; i < NB2; ++i) {
                const long long pix = ((long long)bb * p.a2_H + oy * p.a2_stride) * p.a2_W + ox * p.a2_stride;
                rb2[i] = *reinterpret_cast<const us8*>(p.a2 + pix * C2 + col);
                ox += RSTEP;
                while (ox >= p.OW) { ox -= p.OW; ++oy; }
                if (oy >= p.OH) { oy -= p.OH; ++bb; }
            }
        } else {
            fetch_res(0);
        }
        load_wa(0);
#pragma unroll
        for (int it = 0; it < NSB; ++it) {
            const int i = tid + it * 256;
            if (i < C4 / 4) {
                *reinterpret_cast<f32x4*>(S3 + 4 * i) = sbv[it][0];
                *reinterpret_cast<f32x4*>(B3 + 4 * i) = sbv[it][1];
                if constexpr (C2 > 0) {
                    *reinterpret_cast<f32x4*>(SSC + 4 * i) = sbv[it][2];
                    *reinterpret_cast<f32x4*>(BSC + 4 * i) = sbv[it][3];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int ch = tid + i * 256, row = ch / CPR, col = (ch % CPR) * 8;
            *reinterpret_cast<us8*>(A1 + row * A_LD + col) = rb[i];
        }
        if constexpr (C2 > 0) {
#pragma unroll
            for (int i = 0; i < NB2; ++i) {
                const int ch = tid + i * 256, row = ch / CPR2, col = (ch % CPR2) * 8;
                *reinterpret_cast<us8*>(A2 + row * A2_LD + col) = rb2[i];
            }
        } else {
            park_res();
        }
    }
    __syncthreads();

    f32x16 acc2[CPT][NRT];
#pragma unroll
    for (int t = 0; t < CPT; ++t)
#pragma unroll
        for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[t][r][e] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // ---- GEMM 1: this wave's 32 channels of y chunk c, one 32-pixel row tile at a time (the fragments stay in registers for both)
#pragma unroll
        for (int r = 0; r < NRT; ++r) {
            f32x16 acc, accs;
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accs[e] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < KF1; ++kk) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(A1 + (r * 32 + l31) * A_LD + kk * 16 + half * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[kk], af, acc, 0, 0, 0);
                if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (C2 > 0) {
#pragma unroll
                for (int kk = 0; kk < KF1S; ++kk) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(A2 + (r * 32 + l31) * A2_LD + kk * 16 + half * 8);
                    accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws[kk], af, accs, 0, 0, 0);
                    if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (r == NRT - 1) {                              // the fragments are dead: GEMM 2's and the next chunk's go out behind the last MFMA
                __builtin_amdgcn_sched_barrier(0);
                load_wb(c);
                if (c + 1 < NCH) load_wa(c + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int nl = wave * 32 + 16 * half + 8 * qq, n = c * CH + nl;
                bf16_t* yp = Y + (r * 32 + l31) * Y_LD + nl;
                us8 r8 = {0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (C2 == 0) r8 = *reinterpret_cast<const us8*>(yp);
                us8 o8;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 s3 = *reinterpret_cast<const f32x4*>(S3 + n + 4 * j), b3 = *reinterpret_cast<const f32x4*>(B3 + n + 4 * j);
                    f32x4 ss = {0.f, 0.f, 0.f, 0.f}, bs = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (C2 > 0) { ss = *reinterpret_cast<const f32x4*>(SSC + n + 4 * j); bs = *reinterpret_cast<const f32x4*>(BSC + n + 4 * j); }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[4 * (2 * qq + j) + e] * s3[e];
                        v += b3[e];
                        if constexpr (C2 > 0) {
                            float sc = accs[4 * (2 * qq + j) + e] * ss[e];
                            sc += bs[e];
                            v += sc;
                        } else {
                            v += bf16_to_f32(r8[4 * j + e]);
                        }
                        o8[4 * j + e] = f32_to_bf16(v > 0.f ? v : 0.f);
                    }
                    if (C2 > 0) __builtin_amdgcn_sched_barrier(0);
                }
                *reinterpret_cast<us8*>(yp) = o8;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                     // y chunk complete
#pragma unroll
        for (int i = 0; i < RCH; ++i) {                      // y chunk -> HBM as 256-byte row segments
            const us8 v = *reinterpret_cast<const us8*>(Y + (trow + 16 * i) * Y_LD + tcol);
            *reinterpret_cast<us8*>(y_b + c * (CH * 2) + (toff + (unsigned)(i * 16 * C4 * 2))) = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (C2 == 0 && c + 1 < NCH) fetch_res(c + 1);
        // ---- GEMM 2: a' += y_c W1'[:, chunk c]^T
#pragma unroll
        for (int kk = 0; kk < KF2; ++kk)
#pragma unroll
            for (int r = 0; r < NRT; ++r) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(Y + (r * 32 + l31) * Y_LD + kk * 16 + half * 8);
#pragma unroll
                for (int t = 0; t < CPT; ++t) acc2[t][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[t][kk], af, acc2[t][r], 0, 0, 0);
                if (r == NRT - 1 && (kk & 1)) __builtin_amdgcn_sched_barrier(0);
            }
        __syncthreads();                                     // every wave is done with chunk c
        if (C2 == 0 && c + 1 < NCH) {
            park_res();
            __syncthreads();
        }
    }
    // ---- a' = relu(bn1(acc2)) -> LDS -> whole rows
#pragma unroll
    for (int t = 0; t < CPT; ++t)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int n = (wave + NW * t) * 32 + 16 * half + 8 * qq;
            f32x4 s1[2], b1[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { s1[j] = *reinterpret_cast<const f32x4*>(p.s1 + n + 4 * j); b1[j] = *reinterpret_cast<const f32x4*>(p.b1 + n + 4 * j); }
#pragma unroll
            for (int r = 0; r < NRT; ++r) {
                us8 o8;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc2[t][r][4 * (2 * qq + j) + e] * s1[j][e];
                        v += b1[j][e];
                        o8[4 * j + e] = f32_to_bf16(v > 0.f ? v : 0.f);
                    }
                *reinterpret_cast<us8*>(O + (r * 32 + l31) * O_LD + n) = o8;
            }
        }
    __syncthreads();
    constexpr int CPRO = CN / 8, RPI = 256 / CPRO;
    static_assert(BM * CPRO % 256 == 0, "a' tile chunking");
    const int orow = tid / CPRO, ocol = tid % CPRO;
    unsigned char* o_b = reinterpret_cast<unsigned char*>(p.o) + m0 * CN * (p.o_fp8 ? 1 : 2);
#pragma unroll
    for (int i = 0; i < BM * CPRO / 256; ++i) {
        const int row = orow + i * RPI;
        const us8 v = *reinterpret_cast<const us8*>(O + row * O_LD + ocol * 8);
        if (p.o_fp8) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = bf16_to_f32(v[e]);
            *reinterpret_cast<uint2*>(o_b + (unsigned)(row * CN + ocol * 8)) = f32x8_to_fp8(f);
        } else {
            *reinterpret_cast<us8*>(o_b + (unsigned)(row * CN + ocol * 8) * 2u) = v;
        }
    }
}

template <int C, int C4, int CN, int C2>
static int pw_launch_rt4h(const PwArgs& a, hipStream_t stream) {
    constexpr size_t lds = PwRt4h<C, C4, CN, C2>::BYTES;
    NPS_ENSURE_LDS((int)lds, pw_chain_rt4h_kernel<C, C4, CN, C2>);
    hipLaunchKernelGGL((pw_chain_rt4h_kernel<C, C4, CN, C2>), dim3((unsigned)(a.M / 64)), dim3(256), lds, stream, a);
    return 0;
}

template <int C, int C4, int CN, int C2>
static int pw_launch_rt8(const PwArgs& a, hipStream_t stream) {
    constexpr size_t lds = PwRt8<C, C4, CN, C2>::BYTES;
    if (g_rt8_dbg && C == 128) {                              // tuning runs only
        NPS_ENSURE_LDS((int)lds, pw_chain_rt8_kernel<C, C4, CN, C2, true>);
        hipLaunchKernelGGL((pw_chain_rt8_kernel<C, C4, CN, C2, true>), dim3((unsigned)((a.M + 127) / 128)), dim3(512), lds, stream, a, g_rt8_dbg);
        return 0;
    }
    NPS_ENSURE_LDS((int)lds, pw_chain_rt8_kernel<C, C4, CN, C2>);
    hipLaunchKernelGGL((pw_chain_rt8_kernel<C, C4, CN, C2>), dim3((unsigned)((a.M + 127) / 128)), dim3(512), lds, stream, a, (unsigned long long*)nullptr);
    return 0;
}

template <int C, int C4, int CN, int C2, int BM>
static int pw_launch(const PwArgs& a, hipStream_t stream) {
    constexpr size_t lds = PwLds<C, C4, CN, C2, BM>::BYTES;
    NPS_ENSURE_LDS((int)lds, pw_chain_kernel<C, C4, CN, C2, BM>);
    const unsigned blocks = (unsigned)((a.M + BM - 1) / BM);
    hipLaunchKernelGGL((pw_chain_kernel<C, C4, CN, C2, BM>), dim3(blocks), dim3(256), lds, stream, a);
    return 0;
}

}  // namespace nps

extern "C" void nps_rt8_debug_buffer(void* buf) { nps::g_rt8_dbg = (unsigned long long*)buf; }

extern "C" int nopesac_bottleneck_tail_bf16(const void* b, const void* w3, const float* scale3, const float* bias3, const void* residual,
                                            const void* x2, const void* wsc, const float* scale_sc, const float* bias_sc, int B, int OH,
                                            int OW, int x2_H, int x2_W, int x2_stride, int C, int C4, int C2, void* y, const void* w1,
                                            const float* scale1, const float* bias1, int CN, void* o, void* stream) {
    return nopesac_bottleneck_tail_bf16_ex(b, w3, scale3, bias3, residual, x2, wsc, scale_sc, bias_sc, B, OH, OW, x2_H, x2_W, x2_stride, C, C4,
                                           C2, y, w1, scale1, bias1, CN, o, NPS_DT_BF16, stream);
}

extern "C" int nopesac_bottleneck_tail_bf16_ex(const void* b, const void* w3, const float* scale3, const float* bias3, const void* residual,
                                               const void* x2, const void* wsc, const float* scale_sc, const float* bias_sc, int B, int OH,
                                               int OW, int x2_H, int x2_W, int x2_stride, int C, int C4, int C2, void* y, const void* w1,
                                               const float* scale1, const float* bias1, int CN, void* o, int o_dt, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(o_dt == NPS_DT_BF16 || o_dt == NPS_DT_FP8, "bottleneck_tail: o_dt must be NPS_DT_BF16 or NPS_DT_FP8");
    NPS_CHECK_ARG(b && w3 && scale3 && bias3 && y && B > 0 && OH > 0 && OW > 0, "bottleneck_tail: bad args");
    NPS_CHECK_ARG((residual != nullptr) != (x2 != nullptr), "bottleneck_tail: exactly one of residual / x2 (projection shortcut)");
    NPS_CHECK_ARG(!x2 || (wsc && scale_sc && bias_sc && C2 > 0 && x2_stride >= 1 && (OH - 1) * x2_stride < x2_H && (OW - 1) * x2_stride < x2_W),
                  "bottleneck_tail: projection shortcut arguments");
    NPS_CHECK_ARG(CN == 0 || (w1 && scale1 && bias1 && o), "bottleneck_tail: next-conv1 arguments");
    const void* ptrs[] = {b, w3, scale3, bias3, residual, x2, wsc, scale_sc, bias_sc, y, w1, scale1, bias1, o};
    for (const void* q : ptrs) NPS_CHECK_ARG(((uintptr_t)q & 15) == 0, "bottleneck_tail: pointers must be 16-byte aligned");
    PwArgs a;
    memset(&a, 0, sizeof(a));
    a.a1 = (const bf16_t*)b; a.w3 = (const bf16_t*)w3; a.s3 = scale3; a.b3 = bias3; a.res = (const bf16_t*)residual;
    a.a2 = (const bf16_t*)x2; a.wsc = (const bf16_t*)wsc; a.ssc = scale_sc; a.bsc = bias_sc;
    a.a2_H = x2_H; a.a2_W = x2_W; a.a2_stride = x2_stride; a.OH = OH; a.OW = OW;
    a.y = (bf16_t*)y; a.w1 = (const bf16_t*)w1; a.s1 = scale1; a.b1 = bias1; a.o = (bf16_t*)o;
    a.M = (long long)B * OH * OW;
    a.o_fp8 = o_dt == NPS_DT_FP8 ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    const int c2 = x2 ? C2 : 0;
    // identity blocks of res2 / res3: the four-row-tile kernel (a quarter of the L2 weight traffic); NOPESAC_TAIL_NO_RT4=1 keeps the
    // round-1 form (A/B comparisons: scripts/tail_one.py)
    static const bool no_rt4 = getenv("NOPESAC_TAIL_NO_RT4") != nullptr;
    if (!x2 && !no_rt4 && a.M % 128 == 0) {
#define PW_RT4(c, c4, cn) if (C == c && C4 == c4 && CN == cn) { pw_launch_rt4<c, c4, cn>(a, st); NPS_LAUNCH_RET(); }
        PW_RT4(128, 512, 128) PW_RT4(128, 512, 0) PW_RT4(64, 256, 64) PW_RT4(64, 256, 128) PW_RT4(64, 256, 0)
#undef PW_RT4
    }
    // res3's edge blocks (CN = 256 into res4; the stride-2 projection of res3.0): the eight-wave form
    static const bool no_rt8 = getenv("NOPESAC_TAIL_NO_RT8") != nullptr;
    // round 6: four-wave workgroups of 64 pixels, two per CU (NOPESAC_TAIL_NO_RT4H=1: the eight-wave 128-pixel form below)
    if (!no_rt4 && !no_rt8 && a.M % 64 == 0 && C == 128 && C4 == 512 && !getenv("NOPESAC_TAIL_NO_RT4H")) {
        if (!x2 && CN == 256) { pw_launch_rt4h<128, 512, 256, 0>(a, st); NPS_LAUNCH_RET(); }
        if (x2 && CN == 128 && c2 == 256) { pw_launch_rt4h<128, 512, 128, 256>(a, st); NPS_LAUNCH_RET(); }
    }
    if (!no_rt4 && !no_rt8 && a.M % 128 == 0 && C == 128 && C4 == 512) {
        if (!x2 && CN == 256) { pw_launch_rt8<128, 512, 256, 0>(a, st); NPS_LAUNCH_RET(); }
        if (x2 && CN == 128 && c2 == 256) { pw_launch_rt8<128, 512, 128, 256>(a, st); NPS_LAUNCH_RET(); }
    }
    // res2.0: projection shortcut from a same-resolution source (the stem output), in the same 128-pixel form
    if (x2 && !no_rt4 && a.M % 128 == 0 && x2_stride == 1 && x2_H == OH && x2_W == OW) {
        if (C == 64 && C4 == 256 && CN == 64 && c2 == 64) { pw_launch_rt4_proj<64, 256, 64, 64>(a, st); NPS_LAUNCH_RET(); }
        if (C == 64 && C4 == 256 && CN == 0 && c2 == 64) { pw_launch_rt4_proj<64, 256, 0, 64>(a, st); NPS_LAUNCH_RET(); }
    }
#define PW_CASE(c, c4, cn, cc2, bm) if (C == c && C4 == c4 && CN == cn && c2 == cc2) { pw_launch<c, c4, cn, cc2, bm>(a, st); NPS_LAUNCH_RET(); }
    // (measured in round 2: 64 pixels per workgroup for res3 - half the weight traffic from L2 but ONE 4-wave workgroup per CU - is
    //  slower, 322 vs 272 us on the identity tail: the 32-pixel form stays)
    PW_CASE(64, 256, 64, 0, 64) PW_CASE(64, 256, 128, 0, 64) PW_CASE(64, 256, 64, 64, 64) PW_CASE(64, 256, 0, 0, 64) PW_CASE(64, 256, 0, 64, 64)
    PW_CASE(128, 512, 128, 0, 32) PW_CASE(128, 512, 256, 0, 32) PW_CASE(128, 512, 128, 256, 32) PW_CASE(128, 512, 0, 0, 32) PW_CASE(128, 512, 0, 256, 32)
#undef PW_CASE
    // res4 identity blocks on the eight-wave 128-pixel chunked form (round-5 experiment, NOPESAC_TAIL_RT8_WIDE=1)
    if (!x2 && a.M % 128 == 0 && C == 256 && C4 == 1024 && CN == 256 && getenv("NOPESAC_TAIL_RT8_WIDE")) {
        pw_launch_rt8<256, 1024, 256, 0>(a, st);
        NPS_LAUNCH_RET();
    }
    // identity blocks of res4 / res5: chunk-streaming kernel (weights shared by 128 / 64 pixels)
    if (!x2 && !getenv("NOPESAC_TAIL_NO_STREAM")) {
        if (C == 256 && C4 == 1024 && CN == 256) { pw_launch_stream<256, 1024, 256, 64>(a, st); NPS_LAUNCH_RET(); }
        if (C == 256 && C4 == 1024 && CN == 0) { pw_launch_stream<256, 1024, 0, 64>(a, st); NPS_LAUNCH_RET(); }
        // (res5, C4 = 2048, was measured too: 0.185 ms fused vs 0.158 ms per-layer at 15x20 - only 300 workgroups - so it stays per-layer)
    }
#define PW_WIDE(c, c4, cn, cc2) if (C == c && C4 == c4 && CN == cn && c2 == cc2) { pw_launch_wide<c, c4, cn, cc2>(a, st); NPS_LAUNCH_RET(); }
    PW_WIDE(256, 1024, 256, 0) PW_WIDE(256, 1024, 512, 0) PW_WIDE(256, 1024, 256, 512) PW_WIDE(256, 1024, 0, 0) PW_WIDE(256, 1024, 0, 512)
#undef PW_WIDE
    set_error("bottleneck_tail: unsupported channel configuration C=%d C4=%d CN=%d C2=%d", C, C4, CN, c2);
    return NPS_E_ARG;
}
