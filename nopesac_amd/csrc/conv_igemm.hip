// Implicit-GEMM NHWC convolution / linear layer on the CDNA4 matrix cores, fused epilogue.
//
//   GEMM view: M = B*OH*OW output pixels, N = Cout, K = KH*KW*Cin, A = im2col(x) gathered on the fly,
//   B^T = w [N][K] (K contiguous), C -> y NHWC.
//
// Structure (one 256-thread workgroup = 4 waves in a 2x2 grid, each wave owns a (BM/2)x(BN/2) tile):
//   global --(16-byte vectors, im2col address math per vector)--> registers --> LDS (double buffered,
//   rows padded to 80 bytes so ds_read_b128 fragment reads are bank-conflict free) --> MFMA
//   32x32x16 bf16 / 32x32x2 f32 with f32 accumulators --> scale/bias/residual/activation --> store.
// The next K-tile's global loads are issued before the current tile's MFMAs (one barrier per K-tile).
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous run of tiles that share
// A rows.
#include <stdlib.h>

#include "conv_common.h"

namespace nps {

// TA = element type of x in memory (float with T = bf16_t is the mixed mode: f32 activations are rounded to
// bf16 while being staged, weights are bf16, MFMA runs at the bf16 rate).
template <typename TA, typename T, int BM, int BN, int VEC, int KMUL = 1, bool TWO_LEVEL_ACC = false>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    // KMUL > 1: deeper K-tile (fewer, fatter pipeline steps) for the small latency-bound head GEMMs
    constexpr int BK = Cfg<T>::BK * KMUL;
    constexpr int LDS_STRIDE = BK + Cfg<T>::VECW;          // elements; 80 bytes per row for both dtypes
    constexpr int VPR = BK / VEC;                           // vectors per tile row
    constexpr int A_VECS = BM * VPR / 256;                  // vectors per thread (A)
    constexpr int B_VECS = BN * VPR / 256;
    static_assert(BM * VPR % 256 == 0 && BN * VPR % 256 == 0, "tile/vec mismatch");
    constexpr int WM = BM / 2, WN = BN / 2;                 // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;               // 32x32 MFMA tiles per wave
    typedef typename VecT<T, VEC>::type vec_t;

    __shared__ __attribute__((aligned(16))) T lds[2 * (BM + BN) * LDS_STRIDE];
    constexpr int BUF_ELEMS = (BM + BN) * LDS_STRIDE;    // A tile then B tile, per buffer

    // ---- XCD-aware tile mapping: block id -> contiguous chunk per XCD (bijective form)
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
    const int bz = p.batched ? blockIdx.y : 0;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const TA* __restrict__ X = (const TA*)p.x;
    const T* __restrict__ Wt = (const T*)p.w + (long long)bz * p.w_bs;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- per-thread A row bookkeeping (rows are fixed across K-tiles)
    long long a_base[A_VECS];   // element offset of pixel (b, ih0, iw0) channel 0; may point outside (checked per tap)
    int a_ih0[A_VECS], a_iw0[A_VECS];
    bool a_ok[A_VECS];
#pragma unroll
    for (int i = 0; i < A_VECS; ++i) {
        const int v = tid + i * 256;
        const int row = v / VPR;
        int m = m0 + row;
        a_ok[i] = m < p.M;
        if (!a_ok[i]) m = 0;
        const int mg = m + bz * p.rows_per_b;          // global pixel index
        if (p.dense1x1) {                              // 1x1 / stride 1 / no padding: pixel index == row index
            a_ih0[i] = 0; a_iw0[i] = 0; a_base[i] = mg;
        } else {
            const int b = mg / p.rows_per_b, rem = mg % p.rows_per_b;
            const int oh = rem / p.OW, ow = rem % p.OW;
            a_ih0[i] = oh * p.stride - p.pad;
            a_iw0[i] = ow * p.stride - p.pad;
            a_base[i] = (long long)b * p.H * p.W;
        }
    }
    const bool is1x1 = (p.KH == 1 && p.KW == 1);

    // PF2 (round 4; bf16 activations and weights, BK = 128: the small latency-bound GEMMs / one-pair convs, a handful of workgroups
    // each): TWO tiles of global loads in flight.  With one (the other builds) a K step costs a full memory round trip minus the 256
    // cycles of its MFMAs: ~1.2 us per step when nothing else runs on the CU.  The loads must be BRANCH-FREE for that to work - behind
    // the im2col bounds branches the compiler can only wait with vmcnt(0), which also drains the tile requested last - so this build
    // fetches through bounds-checked buffer descriptors: an out-of-image tap / row >= M / channel >= N gets an out-of-range voffset
    // and reads zeros.  Offsets are relative to the first image (dense 1x1: first pixel) of the tile, so they fit 32 bits whatever the
    // batch (the launcher checks one tile's span).
#ifdef NPS_NO_PF2                                           // A/B builds (NOPESAC_HIPCC_EXTRA=-DNPS_NO_PF2)
    constexpr bool PF2 = false;
#else
    constexpr bool PF2 = sizeof(T) == 2 && sizeof(TA) == 2 && KMUL == 4 && VEC == 8;
#endif
    vec_t a_regs[PF2 ? 2 : 1][A_VECS], b_regs[PF2 ? 2 : 1][B_VECS];
    constexpr unsigned OOB = 0xFFFFFF00u;
    unsigned a_rel[PF2 ? A_VECS : 1], b_off[PF2 ? B_VECS : 1];
    __amdgpu_buffer_rsrc_t xsrc, wsrc;
    if constexpr (PF2) {
        const long long mg0 = (long long)m0 + (long long)bz * p.rows_per_b;
        const long long px0 = p.dense1x1 ? mg0 : mg0 / p.rows_per_b * p.H * p.W;            // first pixel the tile can touch
        const long long left = ((long long)p.B * p.H * p.W - px0) * p.x_cs * 2;
        xsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(X + px0 * p.x_cs), 0, (int)(left < 0x7FFFFFFFll ? left : 0x7FFFFFFFll), 0x00020000);
        wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, (int)((long long)p.N * p.K * 2), 0x00020000);
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            // a_base: pixel index of (b, 0, 0) / of the row; + the (ih0, iw0) corner (mod 2^32: a padding corner is "negative")
            a_rel[i] = (unsigned)((a_base[i] - px0 + (long long)a_ih0[i] * p.W + a_iw0[i]) * p.x_cs * 2);
            if (!a_ok[i]) a_ih0[i] = -(1 << 20);                // rows >= M: every tap out of the image
        }
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) {
            const int v = tid + i * 256;
            const int n = n0 + v / VPR;
            b_off[i] = n < p.N ? (unsigned)(((long long)n * p.K + (v % VPR) * VEC) * 2) : OOB;
        }
    }

    auto load_tile = [&](int kt, auto SETC) {
        constexpr int SET = decltype(SETC)::value;
        vec_t (&a_reg)[A_VECS] = a_regs[SET];
        vec_t (&b_reg)[B_VECS] = b_regs[SET];
        if constexpr (PF2) {
            // 256 % VPR == 0: all of a thread's vectors sit at the same k -> one tap decode per tile, no branch anywhere (a 1x1 layer
            // has Cin == K: tap 0; the dense form has ih0 = iw0 = 0).  k >= K (the last tile of a K that is no multiple of 128): both
            // operands out of range -> zeros
            const int k = kt * BK + (tid % VPR) * VEC;
            const bool kin = k < p.K;
            const int tap = k / p.Cin, c = k - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const unsigned tapoff = (unsigned)((kh * p.W + kw) * (int)p.x_cs + c) * 2u;
#pragma unroll
            for (int i = 0; i < A_VECS; ++i) {
                const bool in = (unsigned)(a_ih0[i] + kh) < (unsigned)p.H && (unsigned)(a_iw0[i] + kw) < (unsigned)p.W;
                const unsigned vo = in && kin ? a_rel[i] + tapoff : OOB;       // a_rel: pixel (ih0, iw0) - may be "negative", the sum is not
                a_reg[i] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(xsrc, (int)vo, 0, 0));
            }
#pragma unroll
            for (int i = 0; i < B_VECS; ++i)
                b_reg[i] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(wsrc, (int)(kin ? b_off[i] : OOB), kt * BK * 2, 0));
            __builtin_amdgcn_sched_barrier(0);                 // the requests go out HERE, not after the MFMAs of the tile in LDS
            return;
        }
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            const int v = tid + i * 256;
            const int kv = v % VPR;
            const int k = kt * BK + kv * VEC;
            vec_t val = vec_t{};
            if (a_ok[i] && k < p.K) {
                int c = k, ih = a_ih0[i], iw = a_iw0[i];
                if (!is1x1) {
                    const int tap = k / p.Cin;
                    c = k - tap * p.Cin;
                    const int kh = tap / p.KW, kw = tap - kh * p.KW;
                    ih += kh; iw += kw;
                }
                if (p.dense1x1 || ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)) {
                    const TA* src = X + (a_base[i] + (p.dense1x1 ? 0ll : (long long)ih * p.W + iw)) * p.x_cs + c;
                    if constexpr (sizeof(TA) == sizeof(T)) {
                        val = *(const vec_t*)src;
                    } else if constexpr (VEC == 1) {
                        val = f32_to_bf16(*src);
                    } else {
#pragma unroll
                        for (int e4 = 0; e4 < VEC / 4; ++e4) {
                            const f32x4 f = *(const f32x4*)(src + 4 * e4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[4 * e4 + e] = f32_to_bf16(f[e]);
                        }
                    }
                }
            }
            a_reg[i] = val;
        }
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) {
            const int v = tid + i * 256;
            const int row = v / VPR, kv = v % VPR;
            const int n = n0 + row, k = kt * BK + kv * VEC;
            vec_t val = vec_t{};
            if (n < p.N && k < p.K) val = *(const vec_t*)(Wt + (long long)n * p.K + k);
            b_reg[i] = val;
        }
    };
    auto store_tile = [&](int buf, auto SETC) {
        constexpr int SET = decltype(SETC)::value;
        vec_t (&a_reg)[A_VECS] = a_regs[SET];
        vec_t (&b_reg)[B_VECS] = b_regs[SET];
#pragma unroll
        for (int i = 0; i < A_VECS; ++i) {
            const int v = tid + i * 256;
            *(vec_t*)(lds + buf * BUF_ELEMS + (v / VPR) * LDS_STRIDE + (v % VPR) * VEC) = a_reg[i];
        }
#pragma unroll
        for (int i = 0; i < B_VECS; ++i) {
            const int v = tid + i * 256;
            *(vec_t*)(lds + buf * BUF_ELEMS + (BM + v / VPR) * LDS_STRIDE + (v % VPR) * VEC) = b_reg[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // f32 (parity) path: TWO-LEVEL accumulation.  One MFMA accumulator chain over K = 2304 .. 4608 terms carries a rounding error
    // ~sqrt(K) ulp; the reference's CPU GEMMs sum in 8-16 parallel lanes and blocks, i.e. with a much shorter chain.  Every
    // FLUSH K-tiles (128 terms) the running accumulator is added into a second-level one and cleared: error ~sqrt(128) + sqrt(K/128)
    // ulp - about 5x smaller for the long-K layers - which is what keeps pred_plane / camera.tran inside the ABSOLUTE 1e-4 gate
    // against the reference (they sit at 0.5-1e-4 with a single chain).  Not used in the bf16 / mixed modes.
    constexpr bool TWO_LEVEL = sizeof(T) == 4 && TWO_LEVEL_ACC;        // the launcher picks this build for f32 layers with K >= 512
    constexpr int FLUSH = 8;
    f32x16 acc_hi[TWO_LEVEL ? TM : 1][TWO_LEVEL ? TN : 1];
    if constexpr (TWO_LEVEL) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_hi[i][j][r] = 0.f;
    }
    const int nk = (p.K + BK - 1) / BK;
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, PF2 ? 1 : 0> S1;
    load_tile(0, S0{});
    store_tile(0, S0{});
    if constexpr (PF2) load_tile(nk > 1 ? 1 : 0, S1{});
    __syncthreads();

    auto compute_tile = [&](int kt) {
        const int buf = kt & 1;
        const T* Ab = lds + buf * BUF_ELEMS + (wm * WM + (lane & 31)) * LDS_STRIDE;
        const T* Bb = lds + buf * BUF_ELEMS + (BM + wn * WN + (lane & 31)) * LDS_STRIDE;
        if constexpr (sizeof(T) == 2) {
            // v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8*(l>>5) .. +8], same for B^T
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8 af[TM], bfr[TN];
                const int ko = kk * 16 + (lane >> 5) * 8;
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(Ab + i * 32 * LDS_STRIDE + ko);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = *(const bf16x8*)(Bb + j * 32 * LDS_STRIDE + ko);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            }
        } else {
            // v_mfma_f32_32x32x2_f32 (exact f32): lane l supplies A[row l&31][k = l>>5]; we feed it the
            // k-slots {4h+s, h=l>>5} for s=0..3 from one 16-byte fragment read (any consistent A/B k
            // pairing is a valid contraction order).
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 af[TM], bfr[TN];
                const int ko = kk * 8 + (lane >> 5) * 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(Ab + i * 32 * LDS_STRIDE + ko);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = *(const f32x4*)(Bb + j * 32 * LDS_STRIDE + ko);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[j][s], af[i][s], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (TWO_LEVEL) {
            if ((kt % FLUSH) == FLUSH - 1 && kt + 1 < nk) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { acc_hi[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
            }
        }
    };
    if constexpr (PF2) {
        // tile kt in LDS, tile kt+1 in register set (kt+1)&1 (in flight), tile kt+2 requested into set kt&1 before tile kt's MFMAs
        // The steady-state body is straight-line (a request past the last tile re-reads the last one; the one or two tiles left at the
        // end are a separate tail): any branch around a load or a store makes the compiler merge wait states conservatively, and the
        // loop top then waits for loads that were issued one half-iteration earlier.
        int kt = 0;
        for (; kt + 2 < nk; kt += 2) {
            load_tile(kt + 2, S0{});
            compute_tile(kt);
            store_tile(1, S1{});
            __syncthreads();
            load_tile(kt + 3 < nk ? kt + 3 : nk - 1, S1{});
            compute_tile(kt + 1);
            store_tile(0, S0{});
            __syncthreads();
        }
        compute_tile(kt);
        if (kt + 1 < nk) {
            store_tile(1, S1{});
            __syncthreads();
            compute_tile(kt + 1);
        }
        __syncthreads();
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tile(kt + 1, S0{});
            compute_tile(kt);
            if (kt + 1 < nk) store_tile((kt & 1) ^ 1, S0{});
            __syncthreads();
        }
    }
    if constexpr (TWO_LEVEL) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += acc_hi[i][j][r];
    }

    conv_epilogue<BM, BN, TM, TN>(acc, reinterpret_cast<float*>(lds), (int)sizeof(lds), p, m0, n0, bz, wm, wn, lane, tid);
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant (bf16, Cin % 64 == 0): both operand tiles go HBM -> LDS with global_load_lds_dwordx4
// (no VGPR round trip, no ds_write pass), BK = 64, two stages, ONE barrier per K-tile: the next tile's
// DMAs are issued before the current tile's MFMAs and drained (vmcnt(0)) at the barrier that follows them.
// The LDS image is lane-linear per DMA (1 KB = 8 rows x 128 B); bank conflicts of the ds_read_b128
// fragment reads are removed by an XOR swizzle applied on the SOURCE side (the lane that fills physical
// 16-byte slot pc of row r fetches logical slot pc ^ ((r>>1)&7)) and again on the read.
// Padding taps / rows beyond M / channels beyond N get an out-of-range buffer offset (the load returns zeros).

template <int BM, int BN, int NSTAGE = 2, int WAVES_M = 2, int BKT = 64>
__global__ __launch_bounds__(WAVES_M * 128, (NSTAGE == 2 || WAVES_M == 4) ? (BKT == 32 ? 4 : 2) : 1) void conv_igemm_glds_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource builtins below exist only in the device pass
    typedef bf16_t T;
    constexpr int BK = BKT, ROWB = BK * 2;                      // 128 (64) bytes per tile row
    constexpr int CPR = BK / 8;                                  // 16-byte chunks per row
    constexpr int RPD = 64 / CPR;                                // tile rows per 1-KB DMA
    constexpr int SWSH = BK == 64 ? 1 : 2;                       // rows per 256-byte bank row = 1 << SWSH
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int NWAVES = WAVES_M * 2;
    constexpr int WM = BM / WAVES_M, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_DMA = BM / RPD / NWAVES, B_DMA = BN / RPD / NWAVES;   // 1-KB DMAs per wave per stage
    constexpr int EPI_BYTES = BM * ((BN > 64 ? 64 : BN) + 4) * 4;
    constexpr int LDS_BYTES = NSTAGE * STAGE > EPI_BYTES ? NSTAGE * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
    const int bz = p.batched ? blockIdx.y : 0;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const T* __restrict__ X = (const T*)p.x;
    const T* __restrict__ Wt = (const T*)p.w + (long long)bz * p.w_bs;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- per-lane DMA sources.  DMA j of this wave fills physical rows (wave*A_DMA + j)*8 + (lane>>3), slot lane&7.
    // Addressing is split so that the K-loop does almost no VALU work (PMC: the first version spent 8.6 VALU
    // instructions per MFMA on 64-bit im2col address math):
    //   buffer resource  = x shifted back by the padding offset (so every tap offset is >= 0), bounds-checked
    //   voffset (VGPR)   = byte offset of the lane's output pixel + its swizzled 8-channel chunk   (loop invariant)
    //   soffset (SGPR)   = byte offset of the K-tile's tap (kh,kw) and channel base                  (wave uniform)
    //   out-of-image taps / rows >= M / channels >= N: voffset = OOB  ->  the buffer load returns zeros
    const int slot = lane % CPR, rsub = lane / CPR;
    constexpr unsigned OOB = 0xFFFFFF00u;
    const long long padb = ((long long)p.pad * p.W + p.pad) * p.x_cs * 2;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x - padb), 0, (int)(((long long)p.B * p.H * p.W * p.x_cs) * 2 + padb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Wt, 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    unsigned a_voff[A_DMA], a_mask[A_DMA];
#pragma unroll
    for (int j = 0; j < A_DMA; ++j) {
        const int pr = (wave * A_DMA + j) * RPD + rsub;
        const int m = m0 + pr;
        const int coff = (slot ^ ((pr >> SWSH) & (CPR - 1))) * 8;           // logical 8-channel chunk this lane fetches
        a_voff[j] = OOB; a_mask[j] = 0u;
        if (m < p.M) {
            const int mg = m + bz * p.rows_per_b;
            const int b = mg / p.rows_per_b, rem = mg % p.rows_per_b;
            const int oh = rem / p.OW, ow = rem % p.OW;
            const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
            a_voff[j] = (unsigned)((((long long)b * p.H + oh * p.stride) * p.W + ow * p.stride) * p.x_cs + coff) * 2u;
            unsigned mk = 0u;
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw)
                    if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) mk |= 1u << (kh * p.KW + kw);
            a_mask[j] = mk;
        }
    }
    unsigned b_voff[B_DMA];
#pragma unroll
    for (int j = 0; j < B_DMA; ++j) {
        const int pr = (wave * B_DMA + j) * RPD + rsub;
        const int n = n0 + pr;
        b_voff[j] = n < p.N ? (unsigned)((long long)n * p.K + (slot ^ ((pr >> SWSH) & (CPR - 1))) * 8) * 2u : OOB;
    }
    // wave-uniform K-tile cursor (no integer division in the loop): tap index, its byte offset, channel base
    int cur_tap = 0, cur_kw = 0, cur_c0 = 0;
    unsigned cur_tapoff = 0u;                                    // ((kh*W + kw) * x_cs) * 2
    unsigned cur_k0b = 0u;                                       // k0 * 2
    auto issue = [&](int /*kt*/, int stage) {
        unsigned char* sbase = lds + stage * STAGE;
        const unsigned a_soff = cur_tapoff + (unsigned)cur_c0 * 2u;
#pragma unroll
        for (int j = 0; j < A_DMA; ++j) {
            const unsigned vo = ((a_mask[j] >> cur_tap) & 1u) ? a_voff[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lptr_t)(sbase + (wave * A_DMA + j) * 1024), 16, vo, a_soff, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_DMA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lptr_t)(sbase + A_BYTES + (wave * B_DMA + j) * 1024), 16, b_voff[j], cur_k0b, 0, 0);
        // advance the cursor to the next K-tile (tiles are always issued in order)
        if (!p.kmajor) {                                     // tap-major: the order of the weight rows
            cur_k0b += BK * 2;
            cur_c0 += BK;
            if (cur_c0 >= p.Cin) {
                cur_c0 = 0; ++cur_tap; ++cur_kw;
                cur_tapoff += (unsigned)p.x_cs * 2u;
                if (cur_kw == p.KW) { cur_kw = 0; cur_tapoff += (unsigned)(p.W - p.KW) * (unsigned)p.x_cs * 2u; }
            }
        } else {
            // channel-major (round 4, as in conv_igemm_p8_kernel): the KH*KW taps of one BK-channel slice, then the next slice - the
            // pixels re-read for the taps of a slice are one 128-byte line each and stay in L2 between the taps.  Tap-major streams
            // all Cin channels of the tile once per tap: the 3x3 2048 -> 128 layer of the pose net read 700 MB from HBM for a 79 MB input
            ++cur_tap; ++cur_kw;
            cur_k0b += (unsigned)p.Cin * 2u;
            cur_tapoff += (unsigned)p.x_cs * 2u;
            if (cur_kw == p.KW) { cur_kw = 0; cur_tapoff += (unsigned)(p.W - p.KW) * (unsigned)p.x_cs * 2u; }
            if (cur_tap == p.KH * p.KW) {
                cur_tap = 0; cur_kw = 0; cur_tapoff = 0u;
                cur_c0 += BK;
                cur_k0b = (unsigned)cur_c0 * 2u;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    const int sw = (lane >> SWSH) & (CPR - 1);                   // swizzle key of this lane's fragment rows
    const int a_row_off = (wm * WM + (lane & 31)) * ROWB;
    const int b_row_off = A_BYTES + (wn * WN + (lane & 31)) * ROWB;
    auto compute = [&](int stage) {
        const unsigned char* sb = lds + stage * STAGE;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 af[TM], bfr[TN];
            const int so = ((kk * 2 + (lane >> 5)) ^ sw) * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(sb + a_row_off + i * 32 * ROWB + so);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *(const bf16x8*)(sb + b_row_off + j * 32 * ROWB + so);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (NSTAGE == 2) {
        issue(0, 0);
        __syncthreads();                                         // carries the vmcnt(0) for the pending LDS-DMAs
        for (int kt = 0; kt < nk; ++kt) {
            const int stage = kt & 1;
            if (kt + 1 < nk) issue(kt + 1, stage ^ 1);
            compute(stage);
            __syncthreads();  // next tile landed (vmcnt(0)) and every wave is done reading this stage
        }
    } else {
        // NSTAGE-deep ring, NSTAGE-1 tiles of DMA in flight across the barrier: counted vmcnt (each tile = DPT DMAs
        // per wave, retired in issue order) + raw s_barrier, never vmcnt(0) in the steady state.  At iteration kt:
        //   wait until tile kt has landed (<= (NSTAGE-2)*DPT newer DMAs may stay outstanding)  -> barrier
        //   (all waves' pieces of tile kt are visible AND everyone finished reading tile kt-1's stage)
        //   -> refill that stage with tile kt+NSTAGE-1 -> MFMAs on tile kt.
        constexpr int DPT = A_DMA + B_DMA;
        for (int t = 0; t < NSTAGE - 1; ++t)
            if (t < nk) issue(t, t);
        int stage = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = min(nk - 1 - kt, NSTAGE - 2);      // tiles issued after tile kt that may stay in flight
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPT) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + NSTAGE - 1 < nk) {
                int st = stage + NSTAGE - 1;
                if (st >= NSTAGE) st -= NSTAGE;
                issue(kt + NSTAGE - 1, st);
            }
            compute(stage);
            if (++stage == NSTAGE) stage = 0;
        }
        __syncthreads();
    }
    conv_epilogue<BM, BN, TM, TN, WAVES_M>(acc, reinterpret_cast<float*>(lds), LDS_BYTES, p, m0, n0, bz, wm, wn, lane, tid);
#endif
}

// ---------------------------------------------------------------------------------------------
// "A through LDS, B from L2" variant (bf16, Cin % 64 == 0, N % 128 == 0).  PMC on the kernel above shows the waves parked 40 %
// of the time at the barrier that waits for the next K-tile's DMAs: with 128x128 tiles and both operands in LDS, only ONE tile
// of prefetch fits at 2 workgroups/CU, and one tile of MFMA work (512 cycles per wave) does not cover the ~2 us DMA latency.
// Here only the activation tile goes through LDS (16 KB per stage -> a 3-deep ring at 3 workgroups per CU; a 4-deep ring at 2
// workgroups per CU measured slower; K-tile 32 at 4 workgroups per CU wins on the bandwidth-leaning 1x1 layers); the weights are
// pre-permuted to MFMA fragment-major order and each wave streams the fragments of ITS OWN 32 output channels straight from
// L2 into a two-deep register ring (wave tile = 128 pixels x 32 channels: no weight fragment is fetched twice in a workgroup,
// every load is 1 KB contiguous).  vmcnt is counted (DMAs and fragment loads retire in issue order): at the top of K-tile j
// only "everything up to B(j)" has to be back, A(j+1..) and B(j+1) stay in flight across the barrier.
// FP8 = true: x and the weights are OCP e4m3fn bytes, a K-tile row is still BKT * 2 bytes (= BKT * 2 channels), the MFMA is the
// K = 64 v_mfma_f32_32x32x64_f8f6f4 with unit block scales (32 bytes per lane and operand = two 16-byte pieces; the weight pieces
// are stored [K/64][2][64 lanes][16 B] so that every load is still 1 KB contiguous per wave and counts as one vmcnt op).
template <int NSTAGE, int BKT = 64, bool FP8 = false>
__global__ __launch_bounds__(256, BKT == 32 ? 4 : (NSTAGE == 3 ? 3 : 2)) void conv_igemm_bfrag_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef bf16_t T;
    constexpr int EB = FP8 ? 1 : 2;                             // bytes per element
    constexpr int BM = 128, BN = 128, BK = BKT * 2 / EB, ROWB = BKT * 2, CPR = BKT / 8, RPD = 64 / CPR, SWSH = BKT == 64 ? 1 : 2;
    constexpr int KF = BKT / 16;                                // 16-byte-per-lane weight pieces per K-tile (bf16: one per k16 step)
    constexpr int A_BYTES = BM * ROWB;                          // 16 (8) KB per stage
    constexpr int TM = 4, A_DMA = BM / RPD / 4;                 // 4 (2) DMAs per wave per K-tile
    constexpr int EPI_BYTES = BM * (64 + 4) * 4;
    constexpr int LDS_BYTES = NSTAGE * A_BYTES > EPI_BYTES ? NSTAGE * A_BYTES : EPI_BYTES;
    constexpr int DA = NSTAGE - 1;                              // A tiles issued ahead of the one being consumed
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int slot = lane % CPR, rsub = lane / CPR;
    constexpr unsigned OOB = 0xFFFFFF00u;
    const long long padb = ((long long)p.pad * p.W + p.pad) * p.x_cs * EB;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x - padb), 0, (int)(((long long)p.B * p.H * p.W * p.x_cs) * EB + padb), 0x00020000);
    unsigned a_voff[A_DMA], a_mask[A_DMA];
#pragma unroll
    for (int j = 0; j < A_DMA; ++j) {
        const int pr = (wave * A_DMA + j) * RPD + rsub;
        const int m = m0 + pr;
        const int coff = (slot ^ ((pr >> SWSH) & (CPR - 1))) * (16 / EB);
        a_voff[j] = OOB; a_mask[j] = 0u;
        if (m < p.M) {
            const int b = m / p.rows_per_b, rem = m % p.rows_per_b;
            const int oh = rem / p.OW, ow = rem % p.OW;
            const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
            a_voff[j] = (unsigned)((((long long)b * p.H + oh * p.stride) * p.W + ow * p.stride) * p.x_cs + coff) * (unsigned)EB;
            unsigned mk = 0u;
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw)
                    if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) mk |= 1u << (kh * p.KW + kw);
            a_mask[j] = mk;
        }
    }
    int cur_tap = 0, cur_kw = 0, cur_c0 = 0;
    unsigned cur_tapoff = 0u;
    // p.force = 1: channel-major K order (as in conv_igemm_p8_kernel): the pixels a workgroup re-reads for the KH*KW taps of one channel
    // slice are ONE 128-byte line each and stay in the XCD's L2 between the taps.  Tap-major streams all Cin channels of the tile's
    // pixels once per tap: the 3x3 128 -> 128 layers of res3 read 291 MB from HBM for a 79 MB input (profiles/r3_s_pmc_traffic.json).
    const bool kmajor = p.force != 0;
    const int ntaps = p.KH * p.KW;
    auto issue_a = [&](int stage) {
        unsigned char* sbase = lds + stage * A_BYTES;
        const unsigned a_soff = cur_tapoff + (unsigned)cur_c0 * (unsigned)EB;
#pragma unroll
        for (int j = 0; j < A_DMA; ++j) {
            const unsigned vo = ((a_mask[j] >> cur_tap) & 1u) ? a_voff[j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lptr_t)(sbase + (wave * A_DMA + j) * 1024), 16, vo, a_soff, 0, 0);
        }
        if (!kmajor) {                                       // tap-major: all channels of tap 0, then tap 1 ... (the order of the weight rows)
            cur_c0 += BK;
            if (cur_c0 >= p.Cin) {
                cur_c0 = 0; ++cur_tap; ++cur_kw;
                cur_tapoff += (unsigned)p.x_cs * (unsigned)EB;
                if (cur_kw == p.KW) { cur_kw = 0; cur_tapoff += (unsigned)(p.W - p.KW) * (unsigned)p.x_cs * (unsigned)EB; }
            }
        } else {                                             // CHANNEL-major: the KH*KW taps of one BK-channel slice, then the next slice
            ++cur_tap; ++cur_kw;
            cur_tapoff += (unsigned)p.x_cs * (unsigned)EB;
            if (cur_kw == p.KW) { cur_kw = 0; cur_tapoff += (unsigned)(p.W - p.KW) * (unsigned)p.x_cs * (unsigned)EB; }
            if (cur_tap == ntaps) { cur_tap = 0; cur_kw = 0; cur_tapoff = 0u; cur_c0 += BK; }
        }
    };
    // this wave's weight fragments: column tile n0/32 + wave, fragment-major [N/32][K*EB/32 pieces][64][16 B]
    const int nk = p.K / BK, kf_total = p.K * EB / 32;
    const T* wfr = (const T*)p.w + ((long long)(n0 / 32 + wave) * kf_total * 64 + lane) * 8;
    bf16x8 bring[2][KF];
    // K-tile kt of the WALK -> K-tile of the (tap-major) weight layout.  Channel-major walk: kt = slice * ntaps + tap -> tap * slices + slice
    const int cslices = p.Cin / BK;
    auto wkt_of = [&](int kt) { return kmajor ? (kt % ntaps) * cslices + kt / ntaps : kt; };
    auto load_b = [&](int wkt, int buf, int kk) { bring[buf][kk] = *(const bf16x8*)(wfr + (long long)(wkt * KF + kk) * 512); };

    f32x16 acc[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    const int sw = (lane >> SWSH) & (CPR - 1);
    const int a_row_off = (lane & 31) * ROWB;

    // prologue: B(0), A(0..DA-1), B(1)
#pragma unroll
    for (int kk = 0; kk < KF; ++kk) load_b(0, 0, kk);
    for (int t = 0; t < DA; ++t)
        if (t < nk) issue_a(t);
    if (nk > 1) {
        const int w1 = wkt_of(1);
#pragma unroll
        for (int kk = 0; kk < KF; ++kk) load_b(w1, 1, kk);
    }
    // weight cursor of walk tile j + 2 (refilled while tile j is consumed): (tap, slice) advanced once per step
    int b_tap = 2 % ntaps, b_sl = 2 / ntaps;
    // steady state per K-tile j (buf = j & 1): newer than B(j) are A(j+DA-1)?.. see the counts below
    auto step = [&](int j, int buf, int stage_j) {
        // outstanding-op budget when tile j is needed: B(j+1) (4 ops) + the A tiles issued after B(j)
        //   issue order: ... B(j) | A(j+DA-1)* | B(j+1) | A(j+DA) ...   (* issued at the top of iteration j-1, after its barrier)
        // => ops newer than B(j): A_DMA (A(j+DA-1)) + KF (B(j+1)), fewer near the tail.
        const int rem = nk - 1 - j;                          // tiles after j
        if (rem >= 1 && j + DA - 1 < nk && j >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_DMA + KF) : "memory");
        else if (rem >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KF) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // every wave's piece of A(j) landed; everyone left tile j-1's stage
        asm volatile("" ::: "memory");
        if (j + DA < nk) {
            int st = stage_j + DA;
            if (st >= NSTAGE) st -= NSTAGE;
            issue_a(st);                                     // A(j+DA) into the stage tile j-1 just vacated
        }
        const unsigned char* sb = lds + stage_j * A_BYTES;
        const int wk2 = kmajor ? b_tap * cslices + b_sl : j + 2;      // weight K-tile of walk tile j + 2
        if (++b_tap == ntaps) { b_tap = 0; ++b_sl; }
        if constexpr (FP8) {
#pragma unroll
            for (int f = 0; f < KF / 2; ++f) {                   // one K = 64 MFMA step: lane needs bytes 32*(lane>>5) .. +31 of the 64
                const int so0 = ((f * 4 + (lane >> 5) * 2) ^ sw) * 16, so1 = ((f * 4 + (lane >> 5) * 2 + 1) ^ sw) * 16;
                i32x4 a0[TM], a1[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    a0[i] = *(const i32x4*)(sb + a_row_off + i * 32 * ROWB + so0);
                    a1[i] = *(const i32x4*)(sb + a_row_off + i * 32 * ROWB + so1);
                }
                const i32x4 b0 = __builtin_bit_cast(i32x4, bring[buf][2 * f]), b1 = __builtin_bit_cast(i32x4, bring[buf][2 * f + 1]);
                const i32x8 bw = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const i32x8 aw = {a0[i][0], a0[i][1], a0[i][2], a0[i][3], a1[i][0], a1[i][1], a1[i][2], a1[i][3]};
                    acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bw, aw, acc[i][0], 0, 0, 0, 0, 0, 0);
                }
                if (j + 2 < nk) { load_b(wk2, buf, 2 * f); load_b(wk2, buf, 2 * f + 1); }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < KF; ++kk) {
                const int so = ((kk * 2 + (lane >> 5)) ^ sw) * 16;
                bf16x8 af[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const bf16x8*)(sb + a_row_off + i * 32 * ROWB + so);
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bring[buf][kk], af[i], acc[i][0], 0, 0, 0);
                if (j + 2 < nk) load_b(wk2, buf, kk);            // refill the slot that was just consumed
            }
        }
    };
    {
        int stage = 0, j = 0;
        for (; j + 1 < nk; j += 2) {
            step(j, 0, stage);
            if (++stage == NSTAGE) stage = 0;
            step(j + 1, 1, stage);
            if (++stage == NSTAGE) stage = 0;
        }
        if (j < nk) step(j, 0, stage);
        __syncthreads();
    }
    conv_epilogue<BM, BN, TM, 1, 1, 4>(acc, reinterpret_cast<float*>(lds), LDS_BYTES, p, m0, n0, 0, 0, wave, lane, tid);
#endif
}

template <typename TA, typename T, int BM, int BN>
static int launch_cfg(const ConvParams& p0, hipStream_t stream, int vec) {
    ConvParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    dim3 grid(p.tiles_m * p.tiles_n, p.batched ? p.B : 1, 1);
    constexpr int VW = Cfg<T>::VECW;
    if constexpr (sizeof(T) == 2 && BM == 64) {
        // head GEMMs (a few hundred blocks, K >= 128): BK = 128 halves/quarters the number of exposed-latency steps
        // (bf16 activations: 32-bit offsets relative to the tile's first image - see PF2 in the kernel - must cover the images one
        //  tile of BM rows can span)
        const long long span = ((long long)BM / p.rows_per_b + 2) * p.H * p.W * p.x_cs * 2;
#ifdef NPS_NO_PF2                                           // A/B build: without PF2's out-of-bounds zero fill a K tail is not handled
        constexpr bool K_TAIL_OK = false;
#else
        constexpr bool K_TAIL_OK = sizeof(TA) == 2;
#endif
        if (vec == VW && (p.K % 128 == 0 || (K_TAIL_OK && p.K > 128)) && (long long)p.tiles_m * p.tiles_n * (p.batched ? p.B : 1) <= 2048 &&
            (sizeof(TA) != 2 || span < 0x7FFFFFFFll)) {
            hipLaunchKernelGGL((conv_igemm_kernel<TA, T, BM, BN, VW, 4>), grid, dim3(256), 0, stream, p);
            return 0;
        }
    }
    if constexpr (sizeof(T) == 4) {
        // f32 parity path, long K: two-level accumulation (a second accumulator set costs registers: only where the chain is long)
        if (vec == VW && p.K >= 512) {
            hipLaunchKernelGGL((conv_igemm_kernel<TA, T, BM, BN, VW, 1, true>), grid, dim3(256), 0, stream, p);
            return 0;
        }
    }
    if (vec == VW) hipLaunchKernelGGL((conv_igemm_kernel<TA, T, BM, BN, VW>), grid, dim3(256), 0, stream, p);
    else if (sizeof(T) == 2 && vec == 4) {
        if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((conv_igemm_kernel<TA, T, BM, BN, 4>), grid, dim3(256), 0, stream, p);
    } else hipLaunchKernelGGL((conv_igemm_kernel<TA, T, BM, BN, 1>), grid, dim3(256), 0, stream, p);
    return 0;
}

template <typename TA, typename T>
static int launch_dtype(const ConvParams& p, hipStream_t stream) {
    // widest vector such that every vector stays inside one (kh,kw) tap and is 16/8-byte aligned
    int vec = 1;
    constexpr int VW = Cfg<T>::VECW;
    auto ok = [&](int v) {
        const size_t xa = sizeof(TA) == sizeof(T) ? v * sizeof(T) : (v >= 4 ? 16 : sizeof(TA));   // f32x4 pieces in mixed mode
        const int xm = sizeof(TA) == sizeof(T) ? v : (v >= 4 ? 4 : 1);
        return p.Cin % v == 0 && p.K % v == 0 && p.x_cs % xm == 0 && p.w_bs % v == 0 &&
               ((uintptr_t)p.x % xa == 0) && ((uintptr_t)p.w % (v * sizeof(T)) == 0);
    };
    if (ok(VW)) vec = VW;
    else if (sizeof(T) == 2 && ok(4)) vec = 4;
    const long long tiles128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * (p.batched ? p.B : 1);
    if constexpr (sizeof(TA) == 2 && sizeof(T) == 2) {
        // LDS-DMA kernel: bf16, every K-tile of 64 inside one tap, 16-byte aligned 8-channel chunks
        // measured (scripts/conv_microbench.py): the DMA kernel wins on 3x3 layers and on 1x1 layers with K >= 1024,
        // loses on the HBM-bound small-K 1x1 layers (2 blocks/CU keep too few bytes in flight)
        // few tiles but a long K loop (K >= 4096, e.g. one pair's res5 3x3 convs: 40 workgroups of 128x64 walking 72 K-tiles, 60 us): the
        // 64x64-tile kernel with two tiles of prefetch (PF2 in conv_igemm_kernel) has twice the workgroups and half the steps and
        // took the one-pair call from 4.04 to 3.97 ms; the DMA kernel keeps these shapes only when asked for (tuner / NOPESAC_CONV_FORCE)
        const bool pf2_small = p.K >= 128 && (long long)((p.M + 63) / 64) * ((p.N + 63) / 64) * (p.batched ? p.B : 1) <= 2048;
        const bool dma_ok = p.use_glds && vec == 8 && p.Cin % 64 == 0 && (tiles128 >= 192 || (p.K >= 4096 && (p.force >= 3 || !pf2_small))) &&
                            (p.KH * p.KW > 1 || p.K >= 1024 || p.force >= 3) && p.KH * p.KW <= 32 &&
                            (long long)p.B * p.H * p.W * p.x_cs * 2 + ((long long)p.pad * p.W + p.pad) * p.x_cs * 2 < (1ll << 31) && (long long)p.N * p.K * 2 < (1ll << 31);
        if (dma_ok) {
            ConvParams q = p;
            q.tiles_m = (q.M + 127) / 128;
            const char* km = getenv("NOPESAC_GLDS_KMAJOR");                               // "0": tap-major (A/B runs, the order test; read per call)
            const bool no_kmajor = km && !strcmp(km, "0");
            q.kmajor = (p.KH * p.KW > 1 && p.stride == 1 && !no_kmajor) ? 1 : 0;
            // few 128x128 tiles but a long K loop (e.g. 2048 -> 128, 3x3 at 15x20: 150 tiles, K = 18432): 128x64 tiles double the
            // number of workgroups
            const bool narrow = p.N <= 64 || (tiles128 < 256 && p.N % 64 == 0);
            if (!narrow) {
                q.tiles_n = (q.N + 127) / 128;
                const dim3 g(q.tiles_m * q.tiles_n, q.batched ? q.B : 1);
                if (p.force == 4) hipLaunchKernelGGL((conv_igemm_glds_kernel<128, 128, 2, 2, 32>), g, dim3(256), 0, stream, q);
                else hipLaunchKernelGGL((conv_igemm_glds_kernel<128, 128>), g, dim3(256), 0, stream, q);
            } else {
                q.tiles_n = (q.N + 63) / 64;
                hipLaunchKernelGGL((conv_igemm_glds_kernel<128, 64>), dim3(q.tiles_m * q.tiles_n, q.batched ? q.B : 1), dim3(256), 0, stream, q);
            }
            return 0;
        }
    }
    if (p.force == 1) return launch_cfg<TA, T, 128, 128>(p, stream, vec);
    if (p.force == 2) return launch_cfg<TA, T, 64, 64>(p, stream, vec);
    // measured (scripts/conv_microbench.py): 1x1 layers with K <= 256 are HBM-bound streaming ops; the 64x64 tile
    // (8 waves/SIMD, 8 blocks/CU) keeps more bytes in flight than the 128x128 tile (3.8 vs 2.6 TB/s on 64->256+res)
    const bool small_k_stream = p.KH * p.KW == 1 && p.K <= 256;
    if (p.N > 64 && tiles128 >= 192 && !small_k_stream) return launch_cfg<TA, T, 128, 128>(p, stream, vec);
    return launch_cfg<TA, T, 64, 64>(p, stream, vec);
}

}  // namespace nps

extern "C" int nopesac_conv2d_nhwc_ex(const void* x, const void* w, const float* scale, const float* bias,
                                      const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int KH,
                                      int KW, int stride, int pad, int64_t x_cstride, int64_t y_cstride,
                                      int64_t r_cstride, int64_t w_bstride, int act, int in_dt, int out_dt,
                                      int kernel_cfg, void* stream);

extern "C" int nopesac_conv2d_nhwc(const void* x, const void* w, const float* scale, const float* bias,
                                   const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int KH,
                                   int KW, int stride, int pad, int64_t x_cstride, int64_t y_cstride,
                                   int64_t r_cstride, int64_t w_bstride, int act, int in_dt, int out_dt,
                                   void* stream) {
    return nopesac_conv2d_nhwc_ex(x, w, scale, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, x_cstride,
                                  y_cstride, r_cstride, w_bstride, act, in_dt, out_dt, NPS_CONV_AUTO, stream);
}

extern "C" int nopesac_conv2d_nhwc_ex(const void* x, const void* w, const float* scale, const float* bias,
                                      const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int KH,
                                      int KW, int stride, int pad, int64_t x_cstride, int64_t y_cstride,
                                      int64_t r_cstride, int64_t w_bstride, int act, int in_dt, int out_dt,
                                      int kernel_cfg, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && w && y, "conv2d: null pointer");
    NPS_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                  "conv2d: bad dims B=%d H=%d W=%d Cin=%d Cout=%d k=%dx%d s=%d p=%d", B, H, W, Cin, Cout, KH, KW, stride, pad);
    NPS_CHECK_ARG(in_dt == NPS_DT_F32 || in_dt == NPS_DT_BF16 || in_dt == NPS_DT_F32_BF16W, "conv2d: bad in_dt %d", in_dt);
    NPS_CHECK_ARG(out_dt == NPS_DT_F32 || out_dt == NPS_DT_BF16 || (out_dt == NPS_DT_FP8 && in_dt == NPS_DT_BF16 && !residual),
                  "conv2d: bad out_dt %d (fp8 output: bf16 conv without residual only)", out_dt);
    NPS_CHECK_ARG(x_cstride >= Cin && y_cstride >= Cout, "conv2d: channel stride smaller than channel count");
    NPS_CHECK_ARG(!residual || r_cstride >= Cout, "conv2d: residual stride");
    const int res_after = (act & NPS_ACT_RES_AFTER) ? 1 : 0;
    const int bias_batched = (act & NPS_ACT_BIAS_BATCHED) ? 1 : 0;
    NPS_CHECK_ARG(!bias_batched || (w_bstride != 0 && bias), "conv2d: NPS_ACT_BIAS_BATCHED needs batched weights and a bias");
    act &= 0xff;
    NPS_CHECK_ARG(act >= 0 && act <= 3, "conv2d: bad act %d", act);
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.res = residual; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (W + 2 * pad - KW) / stride + 1;
    NPS_CHECK_ARG(p.OH > 0 && p.OW > 0, "conv2d: empty output");
    p.x_cs = x_cstride; p.y_cs = y_cstride; p.r_cs = r_cstride; p.w_bs = w_bstride;
    p.rows_per_b = p.OH * p.OW;
    p.batched = w_bstride != 0;
    p.M = p.batched ? p.rows_per_b : B * p.rows_per_b;
    p.N = Cout; p.K = KH * KW * Cin;
    p.act = act; p.out_dt = out_dt; p.res_after = res_after; p.bias_bs = bias_batched ? Cout : 0;
    p.dense1x1 = (KH == 1 && KW == 1 && stride == 1 && pad == 0) ? 1 : 0;
    {
        // kernel_cfg (per call, e.g. from the load-time autotuner) or NOPESAC_CONV_FORCE (tuning aid):
        // t128 | t64 | glds | glds32 | unset = heuristic
        const char* e = getenv("NOPESAC_CONV_FORCE");
        p.use_glds = 1;
        p.force = 0;
        NPS_CHECK_ARG(kernel_cfg >= 0 && kernel_cfg <= 4, "conv2d: bad kernel_cfg %d", kernel_cfg);
        if (kernel_cfg == NPS_CONV_T128) { p.force = 1; p.use_glds = 0; }
        else if (kernel_cfg == NPS_CONV_T64) { p.force = 2; p.use_glds = 0; }
        else if (kernel_cfg == NPS_CONV_DMA64) p.force = 3;
        else if (kernel_cfg == NPS_CONV_DMA32) p.force = 4;
        if (e) {
            if (!strcmp(e, "t128")) { p.force = 1; p.use_glds = 0; }
            else if (!strcmp(e, "t64")) { p.force = 2; p.use_glds = 0; }
            else if (!strcmp(e, "glds")) { p.force = 3; }
            else if (!strcmp(e, "glds32")) { p.force = 4; }
        }
    }
    {   // vectorised epilogue needs 8-channel runs that are 16-byte aligned in every buffer it touches
        const size_t osz = out_dt == NPS_DT_F32 ? 4 : 2;
        const int al = out_dt == NPS_DT_F32 ? 4 : 8;     // elements per 16 bytes
        bool ok = (y_cstride % al == 0) && ((uintptr_t)y % 16 == 0);
        if (residual) ok = ok && (r_cstride % al == 0) && ((uintptr_t)residual % 16 == 0);
        if (scale) ok = ok && ((uintptr_t)scale % 16 == 0);
        if (bias) ok = ok && ((uintptr_t)bias % 16 == 0) && (!bias_batched || Cout % 4 == 0);
        (void)osz;
        p.epi_vec = ok ? 1 : 0;
        NPS_CHECK_ARG(out_dt != NPS_DT_FP8 || (ok && Cout % 8 == 0), "conv2d: fp8 output needs Cout %% 8 == 0 and 8-channel-aligned y / scale / bias");
    }
    if (in_dt == NPS_DT_BF16) launch_dtype<bf16_t, bf16_t>(p, (hipStream_t)stream);
    else if (in_dt == NPS_DT_F32_BF16W) launch_dtype<float, bf16_t>(p, (hipStream_t)stream);
    else launch_dtype<float, float>(p, (hipStream_t)stream);
    NPS_LAUNCH_RET();
}

static int bfrag_launch(bool fp8, const void* x, const void* w_frag, const float* scale, const float* bias, const void* residual, void* y,
                        int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride,
                        int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int nstage, void* stream) {
    using namespace nps;
    const int eb = fp8 ? 1 : 2;
    NPS_CHECK_ARG(x && w_frag && y, "conv2d_bfrag: null pointer");
    NPS_CHECK_ARG(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && KH * KW <= 32, "conv2d_bfrag: bad dims");
    NPS_CHECK_ARG(Cin > 0 && Cin % 64 == 0 && Cout > 0 && Cout % 128 == 0, "conv2d_bfrag: needs Cin %% 64 == 0 and Cout %% 128 == 0");
    NPS_CHECK_ARG(out_dt == NPS_DT_F32 || out_dt == NPS_DT_BF16 || out_dt == NPS_DT_FP8, "conv2d_bfrag: bad out_dt %d", out_dt);
    const int kmajor = (nstage >> 8) & 1;       // + 256: channel-major K order (better L2 reuse of the taps of a KxK conv)
    nstage &= 0xff;
    NPS_CHECK_ARG(nstage == 3 || nstage == 32, "conv2d_bfrag: variant must be 3 (3-stage ring, K-tile 64) or 32 (4-stage ring, K-tile 32), + 256: channel-major K");
    NPS_CHECK_ARG(!(fp8 && nstage == 3) || Cin % 128 == 0, "conv2d_fp8: variant 3 (K-tile 128) needs Cin %% 128 == 0");
    NPS_CHECK_ARG(x_cstride >= Cin && x_cstride % (16 / eb) == 0 && y_cstride >= Cout && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w_frag % 16 == 0),
                  "conv2d_bfrag: strides / alignment");
    NPS_CHECK_ARG(!residual || (r_cstride >= Cout && out_dt != NPS_DT_FP8), "conv2d_bfrag: residual stride / residual with fp8 output");
    const int res_after = (act & NPS_ACT_RES_AFTER) ? 1 : 0;
    NPS_CHECK_ARG((act & ~(0xff | NPS_ACT_RES_AFTER)) == 0, "conv2d_bfrag: unsupported act flags");
    act &= 0xff;
    NPS_CHECK_ARG(act >= 0 && act <= 3, "conv2d_bfrag: bad act %d", act);
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w_frag; p.scale = scale; p.bias = bias; p.res = residual; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (W + 2 * pad - KW) / stride + 1;
    NPS_CHECK_ARG(p.OH > 0 && p.OW > 0, "conv2d_bfrag: empty output");
    p.x_cs = x_cstride; p.y_cs = y_cstride; p.r_cs = r_cstride; p.w_bs = 0;
    p.rows_per_b = p.OH * p.OW;
    p.batched = 0;
    p.M = B * p.rows_per_b; p.N = Cout; p.K = KH * KW * Cin;
    p.act = act; p.out_dt = out_dt; p.res_after = res_after; p.force = kmajor;
    NPS_CHECK_ARG((long long)B * H * W * x_cstride * eb + ((long long)pad * W + pad) * x_cstride * eb < (1ll << 31), "conv2d_bfrag: input larger than 2 GB");
    {
        const int al = out_dt == NPS_DT_F32 ? 4 : 8;
        bool ok = (y_cstride % al == 0) && ((uintptr_t)y % 16 == 0);
        if (residual) ok = ok && (r_cstride % al == 0) && ((uintptr_t)residual % 16 == 0);
        if (scale) ok = ok && ((uintptr_t)scale % 16 == 0);
        if (bias) ok = ok && ((uintptr_t)bias % 16 == 0);
        p.epi_vec = ok ? 1 : 0;
        NPS_CHECK_ARG(ok || out_dt != NPS_DT_FP8, "conv2d_fp8: fp8 output needs 8-channel-aligned y / scale / bias");
    }
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = p.N / 128;
    const dim3 grid(p.tiles_m * p.tiles_n);
    if (fp8) {
        if (nstage == 3) hipLaunchKernelGGL((conv_igemm_bfrag_kernel<3, 64, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv_igemm_bfrag_kernel<4, 32, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (nstage == 3) hipLaunchKernelGGL((conv_igemm_bfrag_kernel<3>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv_igemm_bfrag_kernel<4, 32>), grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_conv2d_nhwc_bfrag(const void* x, const void* w_frag, const float* scale, const float* bias, const void* residual,
                                         void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                         int64_t x_cstride, int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int nstage,
                                         void* stream) {
    return bfrag_launch(false, x, w_frag, scale, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, x_cstride, y_cstride, r_cstride,
                        act, out_dt, nstage, stream);
}

extern "C" int nopesac_conv2d_nhwc_fp8(const void* x, const void* w_frag8, const float* scale, const float* bias, const void* residual,
                                       void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                       int64_t x_cstride, int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int variant,
                                       void* stream) {
    return bfrag_launch(true, x, w_frag8, scale, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, x_cstride, y_cstride, r_cstride,
                        act, out_dt, variant, stream);
}
