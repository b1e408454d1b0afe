// 256x256-tile implicit-GEMM NHWC convolution for the MFMA-bound bf16 layers (Cin % 64 == 0, Cout % 256 == 0): PERSISTENT
// 8-wave workgroups (one per CU), both operands through LDS-DMA, phase-interleaved schedule with counted vmcnt, the epilogue of
// one tile overlapped with the first DMAs of the next.
//
// Why another kernel: conv_igemm_bfrag_kernel / conv_igemm_glds_kernel own 128x128 tiles - every K = 64 step moves 32 KB from
// L2 for 2.1 MFLOP (64 FLOP/B; the 3x3 256->256 layer at 60x80 pulls 5.7 GB = 14.5 TB/s through L2 at 928 TFLOP/s) and the
// waves of a 4-wave workgroup read fragments and issue MFMAs in lockstep, so the matrix pipe idles during every LDS-read /
// barrier interval (PMC: MFMA-busy 44 %, waves parked 45 % of their cycles).  Here:
//   * tile 256 pixels x 256 channels x K 64: 64 KB per K-tile for 8.4 MFLOP (128 FLOP/B, half the L2 traffic per FLOP);
//     wave grid 2 (pixels) x 4 (channels), wave tile 128 x 64 = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16;
//   * the two waves that share a SIMD (wave w and w + 4 = the two pixel halves) run STAGGERED by one barrier interval:
//     while one issues its LDS fragment reads and its share of the next tiles' DMAs, the other issues 8 MFMAs (256 cycles of
//     the SIMD's matrix pipe) - the pipe is fed from one wave or the other all the time (MI355X_MICROARCH.md "Two waves per
//     SIMD"); the younger half gets one static s_setprio 1.  Measured with in-kernel cycle stamps: 282 cycles per interval
//     (256 = the 8 MFMAs);
//   * a K-tile is 4 phases (pixel half mi x K half kh: 2x2 accumulators x 2 k-steps = 8 MFMAs each, 4 independent
//     accumulator chains); fragment reads per phase 8 / 8 / 4 / 4 ds_read_b128 (the channel fragments of both K halves stay in
//     registers for the second pixel half);
//   * LDS = 2 K-tile buffers x 64 KB; each buffer is staged as FOUR 16 KB pieces (pixel-half A0 / A1, channel halves BL / BH),
//     one piece per phase, into whichever piece is already dead: A0, BL, BH are last read in phase 2, A1 in phase 4, so
//     while tile t computes, phases 1-3 stage BL, BH, A1 of tile t+1 (other buffer) and phase 4 stages A0 of tile t+2 (this
//     buffer).  DMA completion is a counted s_waitcnt vmcnt: "A1(t) landed" before phase 3 (6 newer DMAs may stay in flight),
//     "A0, BL, BH of tile t+1 landed" before the next tile (4 newer ones in flight) - never vmcnt(0) in the steady state;
//     every piece has >= 2 phases (~1000 cycles) between issue and wait;
//   * im2col addressing as in conv_igemm_glds_kernel: loop-invariant per-lane voffset, tap-validity bit mask, wave-uniform
//     SGPR tap offset, out-of-image taps / rows >= M read zeros through the buffer bounds check; bank-conflict XOR swizzle on
//     the source chunk and on the ds_read_b128;
//   * PERSISTENT: the grid is one workgroup per CU; a workgroup walks its XCD's contiguous run of tiles with stride 32.  With
//     one workgroup per CU nothing else can hide a tile's prologue (index math + the first DMA round trip: 12 k cycles) and
//     epilogue (13-50 k cycles; all CUs reach it together) - together they were as long as the 82 k-cycle K loop.  So, after
//     the K loop of tile i: compute tile i+1's addresses, issue the four pieces of ITS K-tile 0 into buffer 0, and only then
//     run tile i's epilogue (f32 staging in the LDS above buffer 0: four passes of 64 pixel rows; thread t owns the 8-channel
//     chunk t % 32, scale / bias loaded once, every wave-uniform decision taken once per pass, each wave store instruction
//     writes two whole 512-byte pixel rows; LDS-only barriers so that stores are never waited for).  The stores drain under
//     the next tile's K loop.
// Hazards (the rules of cdna_hip_programming.md "8-phase template"): a piece is re-staged >= 2 phases after its last
// ds_read (the lagging wave group reads one interval later and its reads retire after the following barrier); a staged piece
// is read >= 1 phase after the wait that retires it (both groups have then passed their own vmcnt and a common barrier).
#include "conv_common.h"

namespace nps {

template <int K>
struct IC { static constexpr int value = K; };

constexpr int P8_BM = 256, P8_BN = 256, P8_BK = 64, P8_ROWB = 128;
constexpr int P8_A_BYTES = P8_BM * P8_ROWB, P8_B_BYTES = P8_BN * P8_ROWB, P8_STAGE = P8_A_BYTES + P8_B_BYTES;   // 32 + 32 KB
constexpr int P8_ELD = P8_BN + 4;                                  // floats per row of the epilogue's staging buffer
constexpr int P8_EPI_ROWS = 64;                                    // pixel rows per epilogue pass
constexpr int P8_EPI_OFF = P8_STAGE;                               // the staging buffer sits ABOVE K-tile buffer 0
constexpr int P8_EPI_BYTES = P8_EPI_ROWS * P8_ELD * 4;             // 65 KB
constexpr int P8_TAB_OFF = (2 * P8_STAGE > P8_EPI_OFF + P8_EPI_BYTES) ? 2 * P8_STAGE : P8_EPI_OFF + P8_EPI_BYTES;   // row table above everything
constexpr int P8_FLAG_OFF = P8_TAB_OFF + P8_BM * 8;                // stream-K: one word "value of the arrival counter" (thread 0 -> workgroup)
constexpr int P8_LDS = P8_FLAG_OFF + 16;                           // 131 KB (ONE __shared__ object: a second one makes hipcc drain vmcnt before every ds_read)
// ---- stream-K (conv_igemm_p8_kernel<.., SK = true>): the workspace = [P8_SK_MAX_TILES arrival counters][slabs]; a slab = one partial
//      256 x 256 f32 accumulator tile in REGISTER order: 16-byte element (b * 4 + q) * 512 + tid, b = accumulator block i * 2 + j
constexpr int P8_SK_MAX_TILES = 4096;
constexpr int P8_SK_HDR = P8_SK_MAX_TILES * 4;
constexpr int P8_SK_SLAB = P8_BM * P8_BN * 4;                      // 256 KB

// LDS-only workgroup barrier: orders this wave's LDS accesses (lgkmcnt) and synchronises, WITHOUT the vmcnt(0) that
// __syncthreads() carries - the epilogue's global stores must not be waited for.
#define P8_LDS_SYNC()                                          \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    } while (0)

// Epilogue of one 256x256 tile: four passes, pass q = accumulator row tile q of BOTH pixel halves (64 pixel rows x 256
// channels f32 in LDS, rows padded by 16 bytes: conflict-free ds_write_b128).  Arithmetic order is conv_epilogue's
// (v * scale + bias, residual before or after the activation): results are identical to the other conv kernels'.
// `drain_dma`: the next tile's first DMAs were issued before this epilogue - wait for them (vmcnt(0)) before the LAST pass's
// stores (the earlier passes' stores are a few thousand cycles old by then and mostly acknowledged).
// Builds 1-3 (no residual, bf16 output): the whole epilogue arithmetic happens in the ACCUMULATOR layout, before staging - a lane
// owns, for its pixel row, the 32 channels wc*64 + j*32 + 8q + 4*(lane >> 5) + e: their scale / bias values sit in 64 registers
// (loaded once per tile, before the next tile's DMAs), the staged tile is bf16 (half the LDS bytes of the f32 staging) and the
// second stage is a pure 16-byte LDS -> global copy.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct P8EpiRegs16 {
    float sc[2][4][4], bs[2][4][4];
};

__device__ __forceinline__ void p8_epilogue16_prefetch(P8EpiRegs16& R, const ConvParams& p, int n0, int wc, int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wc * 64 + j * 32 + 8 * q + 4 * (lane >> 5);
            f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) s4 = *(const f32x4*)(p.scale + n);
            if (p.bias) b4 = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) { R.sc[j][q][e] = s4[e]; R.bs[j][q][e] = b4[e]; }
        }
}

constexpr int P8_ELD16 = P8_BN + 8;                             // bf16 elements per staged row (528 bytes)

template <int EPI>
__device__ __forceinline__ void p8_epilogue16(f32x16 (&acc)[4][2], unsigned char* lds, const ConvParams& p, const __amdgpu_buffer_rsrc_t& ry,
                                              int m0, int n0, int wr, int wc, int lane, int tid, bool drain_dma, const P8EpiRegs16& R) {
    bf16_t* epi = reinterpret_cast<bf16_t*>(lds + P8_EPI_OFF);
    const int c8 = tid & 31, rg = tid >> 5;
    constexpr int act = EPI == 1 ? NPS_ACT_RELU : (EPI == 2 ? NPS_ACT_NONE : NPS_ACT_LEAKY);
    auto do_pass = [&](auto PASSC) {
        constexpr int pass = decltype(PASSC)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = acc[pass][j][4 * q + e] * R.sc[j][q][e];
                    t += R.bs[j][q][e];
                    t += 0.f;                                    // the generic epilogue adds the (absent) residual 0: -0 -> +0
                    if (act == NPS_ACT_RELU) t = t > 0.f ? t : 0.f;
                    else if (act == NPS_ACT_LEAKY) t = t > 0.f ? t : 0.01f * t;
                    v[e] = t;
                }
                const uint2 o = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
                *(uint2*)(epi + (wr * 32 + (lane & 31)) * P8_ELD16 + wc * 64 + j * 32 + 8 * q + 4 * (lane >> 5)) = o;
            }
        if constexpr (pass == 3) {
            // the next tile's first DMAs must have landed before its K loop reads buffer 0.  They are the OLDEST operations in flight
            // (issued before pass 0); younger are exactly the 12 stores of passes 0-2 (unconditional: rows >= M are dropped by the
            // descriptor's bounds check) - a counted wait, the stores are never waited for (round 4; was vmcnt(0))
            if (drain_dma) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        }
        P8_LDS_SYNC();
        u32x4 o[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) o[it] = *(const u32x4*)(epi + (it * 16 + rg) * P8_ELD16 + c8 * 8);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rs = it * 16 + rg;
            const unsigned m = (unsigned)(m0 + (rs >> 5) * 128 + pass * 32 + (rs & 31));
            __builtin_amdgcn_raw_buffer_store_b128(o[it], ry, (int)((m * (unsigned)p.y_cs + (unsigned)(n0 + c8 * 8)) * 2u), 0, 0);
        }
        P8_LDS_SYNC();
    };
    do_pass(IC<0>{});
    do_pass(IC<1>{});
    do_pass(IC<2>{});
    do_pass(IC<3>{});
}

struct P8EpiRegs {                 // what the epilogue loads BEFORE the next tile's DMAs are issued (vmcnt retires in order: a
    float sc[8], bs[8];            // load issued behind the DMAs could only be consumed after they - an HBM burst - have landed)
    us8 r0[4];                     // bf16 residual rows of pass 0
};

template <int EPI>
__device__ __forceinline__ void p8_epilogue_prefetch(P8EpiRegs& R, const ConvParams& p, int m0, int n0, int tid) {
    const int c8 = tid & 31, rg = tid >> 5;
    const int n = n0 + c8 * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { R.sc[e] = 1.f; R.bs[e] = 0.f; }
    if (p.scale) {
        *(f32x4*)(R.sc) = *(const f32x4*)(p.scale + n);
        *(f32x4*)(R.sc + 4) = *(const f32x4*)(p.scale + n + 4);
    }
    if (p.bias) {
        *(f32x4*)(R.bs) = *(const f32x4*)(p.bias + n);
        *(f32x4*)(R.bs + 4) = *(const f32x4*)(p.bias + n + 4);
    }
    if (EPI == 4 || (EPI == 0 && p.res && p.out_dt != NPS_DT_F32)) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rs = it * 16 + rg;
            const int m = m0 + (rs >> 5) * 128 + (rs & 31);
            R.r0[it] = us8{0, 0, 0, 0, 0, 0, 0, 0};
            if (m < p.M) R.r0[it] = *(const us8*)((const bf16_t*)p.res + (long long)m * p.r_cs + n);
        }
    }
}

// EPI: 0 = generic (residual kind, activation and output type decided at run time: ~2100 lines of ISA per pass);
//      1 / 2 / 3 = no residual, bf16 output, activation ReLU / none / LeakyReLU fixed at compile time (~200 lines per pass);
//      4 = bf16 residual added before a ReLU, bf16 output (the expand conv of a bottleneck).
// The whole kernel with the generic epilogue is 60 KB of code - the instruction cache (64 KB per CU pair) then re-fetches the
// tile-boundary code on every tile: the measured 10 k cycles for ~600 instructions of index math were instruction misses.
template <bool STAMP, int EPI>
__device__ __forceinline__ void p8_epilogue(f32x16 (&acc)[4][2], unsigned char* lds, const ConvParams& p, int m0, int n0, int wr, int wc,
                                            int lane, int tid, bool drain_dma, const P8EpiRegs& R, unsigned long long* est) {
    float* epi = reinterpret_cast<float*>(lds + P8_EPI_OFF);
    const int c8 = tid & 31, rg = tid >> 5;
    const int n = n0 + c8 * 8;
    const float (&sc)[8] = R.sc;
    const float (&bs)[8] = R.bs;
    const bool res16 = EPI == 4 || (EPI == 0 && p.res && p.out_dt != NPS_DT_F32);   // bf16 residual rows: prefetched per pass
    const bool res32 = EPI == 0 && p.res && p.out_dt == NPS_DT_F32;
    const int act = EPI == 0 ? p.act : ((EPI == 1 || EPI == 4) ? NPS_ACT_RELU : EPI == 2 ? NPS_ACT_NONE : NPS_ACT_LEAKY);
    const int res_after = EPI == 0 ? p.res_after : 0;
    const int out_dt = EPI == 0 ? p.out_dt : NPS_DT_BF16;
    auto do_pass = [&](auto PASSC) {
        constexpr int pass = decltype(PASSC)::value;
        // staging row rs (0..63): pixel half rs >> 5, row (rs & 31) of accumulator row tile `pass`
        auto row_of = [&](int it) { const int rs = it * 16 + rg; return m0 + (rs >> 5) * 128 + pass * 32 + (rs & 31); };
        us8 r8[4];
        if (res16) {                                            // residual rows of the pass: in flight during the staging writes
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if constexpr (pass == 0) {
                    r8[it] = R.r0[it];
                } else {
                    r8[it] = us8{0, 0, 0, 0, 0, 0, 0, 0};
                    if (row_of(it) < p.M) r8[it] = *(const us8*)((const bf16_t*)p.res + (long long)row_of(it) * p.r_cs + n);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rs = wr * 32 + (lane & 31);
                const int c = wc * 64 + j * 32 + 8 * q + 4 * (lane >> 5);
                const f32x16& a = acc[pass][j];
                const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                *(f32x4*)(epi + rs * P8_ELD + c) = v;
            }
        if constexpr (STAMP && pass == 0) est[7] = __builtin_readcyclecounter();
        P8_LDS_SYNC();
        if constexpr (STAMP && pass == 0) est[8] = __builtin_readcyclecounter();
        // all 4 items of the thread at once: straight-line code, every wave-uniform decision (residual kind, activation, output
        // type) taken ONCE per pass around fully unrolled item loops (a per-element activation switch compiled to 20 k lines of
        // ISA - larger than the instruction cache - and a 43 k-cycle epilogue)
        float v[4][8];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rs = it * 16 + rg;
            *(f32x4*)(v[it]) = *(const f32x4*)(epi + rs * P8_ELD + c8 * 8);
            *(f32x4*)(v[it] + 4) = *(const f32x4*)(epi + rs * P8_ELD + c8 * 8 + 4);
        }
        if constexpr (pass == 3) {                              // as late as possible: the DMAs had the whole epilogue to land
            if (drain_dma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[it][e] *= sc[e]; v[it][e] += bs[e]; }
        if constexpr (STAMP && pass == 0) est[9] = __builtin_readcyclecounter();
        auto add_res = [&]() {
            if (res16) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[it][e] += bf16_to_f32(r8[it][e]);
            } else if (res32) {
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (row_of(it) < p.M) {
                        const float* rp = (const float*)p.res + (long long)row_of(it) * p.r_cs + n;
                        const f32x4 r0 = *(const f32x4*)(rp), r1 = *(const f32x4*)(rp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[it][e] += r0[e]; v[it][4 + e] += r1[e]; }
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[it][e] += 0.f;             // conv_epilogue adds rv = 0 (turns -0 into +0)
            }
        };
        if (!res_after) add_res();
        if (act == NPS_ACT_RELU) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = v[it][e] > 0.f ? v[it][e] : 0.f;
        } else if (act == NPS_ACT_LEAKY) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = v[it][e] > 0.f ? v[it][e] : 0.01f * v[it][e];
        } else if (act == NPS_ACT_SIGMOID) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = 1.f / (1.f + expf(-v[it][e]));
        }
        if (res_after) add_res();
        if constexpr (STAMP && pass == 0) est[10] = __builtin_readcyclecounter();
        if (out_dt == NPS_DT_F32) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (row_of(it) < p.M) {
                    float* yp = (float*)p.y + (long long)row_of(it) * p.y_cs + n;
                    *(f32x4*)(yp) = *(const f32x4*)(v[it]);
                    *(f32x4*)(yp + 4) = *(const f32x4*)(v[it] + 4);
                }
        } else if (out_dt == NPS_DT_FP8) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (row_of(it) < p.M) *(uint2*)((unsigned char*)p.y + (long long)row_of(it) * p.y_cs + n) = f32x8_to_fp8(v[it]);
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                if (row_of(it) < p.M) {
                    uint4 o;
                    o.x = f32x2_to_bf16x2(v[it][0], v[it][1]); o.y = f32x2_to_bf16x2(v[it][2], v[it][3]);
                    o.z = f32x2_to_bf16x2(v[it][4], v[it][5]); o.w = f32x2_to_bf16x2(v[it][6], v[it][7]);
                    *(uint4*)((bf16_t*)p.y + (long long)row_of(it) * p.y_cs + n) = o;
                }
        }
        if constexpr (STAMP) est[pass] = __builtin_readcyclecounter();
        P8_LDS_SYNC();                                          // staging rows free for the next pass / for the DMAs of buffer 1
    };
    do_pass(IC<0>{});
    do_pass(IC<1>{});
    do_pass(IC<2>{});
    do_pass(IC<3>{});
}

// EPI 4 (bf16 residual added before a ReLU, bf16 output: the expand conv of a bottleneck - K = Cin <= 512, four K-tiles, so the
// tile IS its epilogue: 128 KB of residual in, 128 KB out per 9 k cycles of MFMAs).  Round 4: the generic form above issued the
// residual loads of pass q at the START of pass q, i.e. BEHIND the stores of pass q-1 - and vmcnt retires in issue order, so the
// wait for the residual was a wait for the previous pass's store acknowledgements plus one exposed HBM round trip, four times per
// tile (PMC: waves parked 56 % of their cycles, 3.5 TB/s).  Here:
//   * residual and output go through buffer descriptors: rows >= M read zeros / are dropped by the bounds check, so the epilogue
//     is branch-free straight-line code and every s_waitcnt vmcnt is a COUNTED one;
//   * the residual rows of pass q+1 are requested in the middle of pass q (after the staging read-back, before pass q's own
//     arithmetic and stores): they have a whole pass (staging writes, barrier, read-back) to arrive and the wait for them leaves
//     the 4 stores of pass q in flight (vmcnt(4) / vmcnt(8)) - stores are never waited for inside the epilogue.  (Two passes of
//     rows in flight - 16 more registers - measured slower on the same box: 90.0 / 90.9 vs 87.5 / 86.6 us on res4's expand conv,
//     profiles/r4_e_epi4_prefetch_depth_ab.txt: the pass is not waiting for its residual any more);
//   * the next tile's first DMAs (issued before pass 0, older than every residual load) have landed once pass 3's residual has:
//     no vmcnt(0) at the end either.
struct P8Epi4Regs {
    float sc[8], bs[8];
    u32x4 r0[4];                    // residual rows of pass 0 (requested BEFORE the next tile's DMAs)
};

__device__ __forceinline__ unsigned p8_epi4_row(int m0, int it, int rg, int pass) {
    const int rs = it * 16 + rg;
    return (unsigned)(m0 + (rs >> 5) * 128 + pass * 32 + (rs & 31));
}

__device__ __forceinline__ void p8_epilogue4_prefetch(P8Epi4Regs& R, const ConvParams& p, const __amdgpu_buffer_rsrc_t& rres, int m0, int n0,
                                                      int tid) {
    const int c8 = tid & 31, rg = tid >> 5;
    const int n = n0 + c8 * 8;
    *(f32x4*)(R.sc) = *(const f32x4*)(p.scale + n);
    *(f32x4*)(R.sc + 4) = *(const f32x4*)(p.scale + n + 4);
    *(f32x4*)(R.bs) = *(const f32x4*)(p.bias + n);
    *(f32x4*)(R.bs + 4) = *(const f32x4*)(p.bias + n + 4);
#pragma unroll
    for (int it = 0; it < 4; ++it)
        R.r0[it] = __builtin_amdgcn_raw_buffer_load_b128(rres, (int)((p8_epi4_row(m0, it, rg, 0) * (unsigned)p.r_cs + (unsigned)n) * 2u), 0, 0);
}

__device__ __forceinline__ void p8_epilogue4(f32x16 (&acc)[4][2], unsigned char* lds, const ConvParams& p, const __amdgpu_buffer_rsrc_t& rres,
                                             const __amdgpu_buffer_rsrc_t& ry, int m0, int n0, int wr, int wc, int lane, int tid,
                                             const P8Epi4Regs& R) {
    float* epi = reinterpret_cast<float*>(lds + P8_EPI_OFF);
    const int c8 = tid & 31, rg = tid >> 5;
    const int n = n0 + c8 * 8;
    u32x4 rc[4], rn[4];                                         // residual rows of this pass / of the next (in flight)
#pragma unroll
    for (int it = 0; it < 4; ++it) rc[it] = R.r0[it];
    auto do_pass = [&](auto PASSC) {
        constexpr int pass = decltype(PASSC)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rs = wr * 32 + (lane & 31);
                const int c = wc * 64 + j * 32 + 8 * q + 4 * (lane >> 5);
                const f32x16& a = acc[pass][j];
                const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                *(f32x4*)(epi + rs * P8_ELD + c) = v;
            }
        P8_LDS_SYNC();
        float v[4][8];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rs = it * 16 + rg;
            *(f32x4*)(v[it]) = *(const f32x4*)(epi + rs * P8_ELD + c8 * 8);
            *(f32x4*)(v[it] + 4) = *(const f32x4*)(epi + rs * P8_ELD + c8 * 8 + 4);
        }
        if constexpr (pass < 3) {                               // the NEXT pass's residual rows: in front of this pass's stores
#pragma unroll
            for (int it = 0; it < 4; ++it)
                rn[it] = __builtin_amdgcn_raw_buffer_load_b128(rres, (int)((p8_epi4_row(m0, it, rg, pass + 1) * (unsigned)p.r_cs + (unsigned)n) * 2u), 0, 0);
        } else {
            // pass 3: younger than its residual are only pass 2's four stores; everything older - the next tile's first DMAs
            // included - has then landed (vmcnt retires in issue order)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[it][e] *= R.sc[e]; v[it][e] += R.bs[e]; }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const us8 r8 = __builtin_bit_cast(us8, rc[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = v[it][e] + bf16_to_f32(r8[e]);
                v[it][e] = t > 0.f ? t : 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            u32x4 o;
            o.x = f32x2_to_bf16x2(v[it][0], v[it][1]); o.y = f32x2_to_bf16x2(v[it][2], v[it][3]);
            o.z = f32x2_to_bf16x2(v[it][4], v[it][5]); o.w = f32x2_to_bf16x2(v[it][6], v[it][7]);
            __builtin_amdgcn_raw_buffer_store_b128(o, ry, (int)((p8_epi4_row(m0, it, rg, pass) * (unsigned)p.y_cs + (unsigned)n) * 2u), 0, 0);
        }
        if constexpr (pass < 3) {
#pragma unroll
            for (int it = 0; it < 4; ++it) rc[it] = rn[it];
        }
        P8_LDS_SYNC();                                          // staging rows free for the next pass / for the DMAs of buffer 1
    };
    do_pass(IC<0>{});
    do_pass(IC<1>{});
    do_pass(IC<2>{});
    do_pass(IC<3>{});
}

// STAMP: tuning build - per (workgroup, wave) cycle stamps of the first tile into p.dbg (prologue / K loop / epilogue split).
// SK (stream-K, round 5): the K-tiles of an XCD's run of tiles form ONE line of units that is cut evenly over the XCD's workgroups, so
// 300 tiles on 256 workgroups cost 1.17 tile times instead of 2.  A workgroup's range = [tail piece of a tile][whole tiles][head piece of
// a tile]; a partial piece ends in an ARRIVAL on the tile's counter: whoever arrives last adds the other pieces' slabs to its registers
// (in K order, whoever it is: the sum is the same bit pattern in every arrival order) and runs the epilogue, everybody else writes its
// accumulators to its slab first.  Nobody ever waits for another workgroup (no co-residency assumption: with several streams in flight a
// workgroup of this launch may not have started yet).  Publish protocol (cdna_hip_programming.md, in-launch split-K): write-through (sc1)
// slab stores -> s_waitcnt vmcnt(0) -> barrier -> relaxed agent-scope fetch_add; the reducer reads the slabs with sc1 loads.  The last
// arriver re-zeroes the counter: the workspace's counter block is zero between launches (the caller zeroes it once).
template <bool STAMP, int EPI, bool SK = false>
__global__ __launch_bounds__(512) void conv_igemm_p8_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[P8_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // pixel half / channel quarter of the workgroup tile; group = wr
    unsigned long long ts[4] = {0, 0, 0, 0}, est[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (STAMP) ts[0] = __builtin_readcyclecounter();

    // ---- this workgroup's tiles: XCD x (= blockIdx % 8: private L2) owns a contiguous run of tiles, its workgroups walk the run
    //      with stride = workgroups on that XCD, so the 32 CUs of an XCD always work on 32 neighbouring tiles (shared halo rows)
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;     // XCDs that host workgroups of this grid
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int tq = ntiles / nx, tr = ntiles % nx;
    const int run_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int run_end = run_begin + tq + (xcd < tr ? 1 : 0);
    const int stride = ((int)gridDim.x - xcd + 7) / 8;
    const int nk = p.K / P8_BK;
    // SK: this workgroup's units [ub, ue) of the XCD's line of (run_end - run_begin) * nk K-tile units; `cur` = next unit
    const int sk_units = SK ? (run_end - run_begin) * nk : 0;
    auto sk_begin = [&](int s_) { return (int)(((long long)sk_units * s_) / stride); };
    const int ue = SK ? sk_begin(slot + 1) : 0;
    int cur = SK ? sk_begin(slot) : 0;
    int tile = SK ? run_begin + cur / nk : run_begin + slot;
    if (SK ? cur >= ue : tile >= run_end) return;

    // ---- DMA sources.  One 1-KB DMA = 8 tile rows x 128 B, lane -> row (lane >> 3), physical 16-byte slot (lane & 7).
    //   A piece h (pixel half mi = h of BOTH wave rows): DMA j of wave w fills rows j*128 + h*64 + w*8 .. +7
    //   B piece h (channels h*128 .. +127):              DMA j of wave w fills rows h*128 + (2w + j)*8 .. +7
    constexpr unsigned OOB = 0xFFFFFF00u;
    const int slot8 = lane & 7, rsub = lane >> 3;
    const long long padb = ((long long)p.pad * p.W + p.pad) * p.x_cs * 2;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x - padb), 0, (int)(((long long)p.B * p.H * p.W * p.x_cs) * 2 + padb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    // EPI 1-4: output (EPI 4: and residual) through descriptors that end with row M - 1 (rows >= M of the last tile: zeros in, dropped out)
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.res, 0, EPI == 4 ? (int)((((long long)p.M - 1) * p.r_cs + p.N) * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        p.y, 0, EPI >= 1 ? (int)((((long long)p.M - 1) * p.y_cs + p.N) * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(p.sk_ws, 0, SK ? p.sk_ws_bytes : 0, 0x00020000);
    unsigned a_voff[2][2], a_mask[2][2], b_voff[2][2];
    int m0 = 0, n0 = 0;
    // wave-uniform K-tile cursor of the NEXT tile to stage (tiles are staged strictly in order)
    int cur_tap = 0, cur_kw = 0, cur_c0 = 0;
    unsigned cur_tapoff = 0u, cur_k0b = 0u;
    const int ntaps = p.KH * p.KW;
    // Index math of a tile.  Every tile row is addressed by 8 lanes (its eight 16-byte chunks) of 4 different DMAs, so the
    // row -> (image, oh, ow) -> byte offset + tap-validity mask computation is done ONCE per row by threads 0..255 into a 2 KB LDS
    // table, and every lane then picks up its four rows (first version: each lane did it 4 times - ~900 instructions per wave with
    // 32-bit integer divisions, 64-bit products and a 9-tap loop: 5-6 k cycles per tile):
    //   * row -> (image, oh, ow) by float reciprocal + one-step fix-up (exact for M < 2^23)
    //   * every product is a 24-bit multiply (v_mul_u32_u24): pixel indices, channel strides and K are < 2^24 and every byte offset
    //     < 2^31 (checked by the launcher)
    //   * tap mask = row-validity bits x column-validity bits
    const float rcp_rpb = 1.0f / (float)p.rows_per_b, rcp_ow = 1.0f / (float)p.OW;
    auto divmod = [](int a, int d, float rcp, int& q, int& r) {
        q = (int)((float)a * rcp);
        r = a - __mul24(q, d);
        if (r < 0) { --q; r += d; }
        if (r >= d) { ++q; r -= d; }
    };
    const unsigned xcs = (unsigned)p.x_cs;
    uint2* rowtab = reinterpret_cast<uint2*>(lds + P8_TAB_OFF);        // [256] {byte offset of the row's pixel, tap mask}
    unsigned a_coff[2][2], b_base[2][2];                               // per-lane constants: swizzled chunk offsets / weight-row offsets
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pr = j * 128 + h * 64 + wave * 8 + rsub;
            a_coff[h][j] = (unsigned)(slot8 ^ ((pr >> 1) & 7)) * 16u;
            const int br = h * 128 + (wave * 2 + j) * 8 + rsub;
            b_base[h][j] = (__umul24((unsigned)br, (unsigned)p.K) + (unsigned)(slot8 ^ ((br >> 1) & 7)) * 8u) * 2u;
        }
    auto set_tile = [&](int t) {                                       // workgroup-collective (contains a barrier)
        m0 = (t / p.tiles_n) * P8_BM;
        n0 = (t % p.tiles_n) * P8_BN;
        if (tid < P8_BM) {
            const int m = m0 + tid;
            uint2 e = make_uint2(0u, 0u);
            if (m < p.M) {
                int b, rem, oh, ow;
                divmod(m, p.rows_per_b, rcp_rpb, b, rem);
                divmod(rem, p.OW, rcp_ow, oh, ow);
                const int ih0 = __mul24(oh, p.stride) - p.pad, iw0 = __mul24(ow, p.stride) - p.pad;
                const unsigned pix = __umul24(__umul24((unsigned)b, (unsigned)p.H) + (unsigned)(ih0 + p.pad), (unsigned)p.W) + (unsigned)(iw0 + p.pad);
                unsigned colbits = 0u;
                for (int kw = 0; kw < p.KW; ++kw) colbits |= ((unsigned)(iw0 + kw) < (unsigned)p.W ? 1u : 0u) << kw;
                unsigned mk = 0u;
                for (int kh = 0; kh < p.KH; ++kh)
                    if ((unsigned)(ih0 + kh) < (unsigned)p.H) mk |= colbits << __mul24(kh, p.KW);
                e = make_uint2(__umul24(pix, xcs) * 2u, mk);
            }
            rowtab[tid] = e;
        }
        P8_LDS_SYNC();
        const unsigned nb = __umul24((unsigned)n0, (unsigned)p.K) * 2u;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint2 e = rowtab[j * 128 + h * 64 + wave * 8 + rsub];
                a_voff[h][j] = e.x + a_coff[h][j];                  // rows >= M have mask 0: every tap reads out of bounds
                a_mask[h][j] = e.y;
                b_voff[h][j] = b_base[h][j] + nb;                    // Cout % 256 == 0: every weight row of the tile exists
            }
        cur_tap = 0; cur_kw = 0; cur_c0 = 0; cur_tapoff = 0u; cur_k0b = 0u;
    };
    // K order: p.force == 0: tap-major (all channels of tap 0, then tap 1 ...: the natural order of the weight rows);
    //          p.force == 1: CHANNEL-major (the KH*KW taps of channels 0-63, then of channels 64-127 ...): the pixels a workgroup
    //          re-reads for the nine taps of one 64-channel slice are ONE 128-byte line each, so the slice stays in the XCD's L2
    //          between the taps (tap-major streams all Cin channels of the tile's pixels nine times: L2 hit rate 76 %).
    auto advance = [&]() {
        if (p.force == 0) {
            cur_k0b += P8_BK * 2;
            cur_c0 += P8_BK;
            if (cur_c0 >= p.Cin) {
                cur_c0 = 0; ++cur_tap; ++cur_kw;
                cur_tapoff += (unsigned)p.x_cs * 2u;
                if (cur_kw == p.KW) { cur_kw = 0; cur_tapoff += (unsigned)(p.W - p.KW) * (unsigned)p.x_cs * 2u; }
            }
        } else {
            ++cur_tap; ++cur_kw;
            cur_k0b += (unsigned)p.Cin * 2u;
            cur_tapoff += (unsigned)p.x_cs * 2u;
            if (cur_kw == p.KW) { cur_kw = 0; cur_tapoff += (unsigned)(p.W - p.KW) * (unsigned)p.x_cs * 2u; }
            if (cur_tap == ntaps) {
                cur_tap = 0; cur_kw = 0; cur_tapoff = 0u;
                cur_c0 += P8_BK;
                cur_k0b = (unsigned)cur_c0 * 2u;
            }
        }
    };
    auto set_cursor = [&](int kt) {                          // SK: cursor -> K-tile kt of the current tile (wave-uniform integer math)
        int tap, c0;
        if (p.force == 0) { const int per = p.Cin / P8_BK; tap = kt / per; c0 = (kt - tap * per) * P8_BK; }
        else { const int cs = kt / ntaps; tap = kt - cs * ntaps; c0 = cs * P8_BK; }
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        cur_tap = tap; cur_kw = kw; cur_c0 = c0;
        cur_tapoff = (unsigned)(kh * p.W + kw) * (unsigned)p.x_cs * 2u;
        cur_k0b = (unsigned)(tap * p.Cin + c0) * 2u;
    };
    auto stage_a = [&](unsigned char* buf, int h) {          // pixel-half piece h at the cursor's K-tile
        const unsigned soff = cur_tapoff + (unsigned)cur_c0 * 2u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = ((a_mask[h][j] >> cur_tap) & 1u) ? a_voff[h][j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lptr_t)(buf + (j * 128 + h * 64 + wave * 8) * P8_ROWB), 16, vo, soff, 0, 0);
        }
    };
    auto stage_b = [&](unsigned char* buf, int h) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lptr_t)(buf + P8_A_BYTES + (h * 128 + (wave * 2 + j) * 8) * P8_ROWB), 16,
                                                     b_voff[h][j], cur_k0b, 0, 0);
    };
    auto stage_first = [&]() {                               // the four pieces of K-tile 0 -> buffer 0 (cursor at K-tile 0)
        stage_a(lds, 0);
        stage_b(lds, 0);
        stage_b(lds, 1);
        stage_a(lds, 1);
    };

    const int sw = (lane >> 1) & 7;
    const int a_row_off = (wr * 128 + (lane & 31)) * P8_ROWB;
    const int b_row_off = P8_A_BYTES + (wc * 64 + (lane & 31)) * P8_ROWB;
    const int so0 = ((0 + (lane >> 5)) ^ sw) * 16, so1 = ((2 + (lane >> 5)) ^ sw) * 16, so2 = ((4 + (lane >> 5)) ^ sw) * 16,
              so3 = ((6 + (lane >> 5)) ^ sw) * 16;

#define P8_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_s_barrier();              \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

    set_tile(tile);
    int kb = 0;                                              // SK: first K-tile of the current piece (only a range's first piece starts inside a tile)
    if constexpr (SK) {
        kb = cur - (tile - run_begin) * nk;
        if (kb) set_cursor(kb);
    }
    stage_first();
    bool first = true;
    int tile_no = 0;
    while (true) {
        const bool stamp_now = STAMP && tile_no == p.dbg_tile;
        // K-tiles of this piece (SK: the rest of the tile or the rest of the range, whichever ends first)
        const int nkl = SK ? ((nk - kb) < (ue - cur) ? (nk - kb) : (ue - cur)) : nk;
        // ---- on entry: the four pieces of this tile's K-tile 0 are issued (first tile) or landed (later tiles: waited for in the
        //      previous tile's epilogue); the cursor stands at K-tile 0
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        bf16x8 af[2][2], bfr[2][4];
        advance();                                               // cursor -> K-tile 1
        if (nkl > 1) stage_a(lds + P8_STAGE, 0);                 // A0(1)
        if (first) {                                             // A0(0), BL(0), BH(0) must have landed; A1(0) [, A0(1)] may fly
            if (nkl > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (STAMP) {
            if (stamp_now) ts[1] = __builtin_readcyclecounter();
        }
        if (wr == 1) {                                           // the lagging group: one barrier interval behind, higher priority
            __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_s_barrier();
        }

        // one K-tile = 4 phases.  BUF = buffer of tile t; the cursor points at tile t+1 on entry and on exit at tile t+2.
        auto ktile = [&](auto BUFC, int t) {
            constexpr int BUF = decltype(BUFC)::value;
            unsigned char* sb = lds + BUF * P8_STAGE;            // tile t (and tile t+2's A0 piece)
            unsigned char* nb = lds + (BUF ^ 1) * P8_STAGE;      // tile t+1
            const bool has1 = t + 1 < nkl, has2 = t + 2 < nkl;
            const bool pst = STAMP && stamp_now && t == 10;      // per-phase stamps of one mid-loop K-tile (tuning build)
            auto mfma_phase = [&](auto MIC, auto KHC) {
                constexpr int MI = decltype(MIC)::value, KH = decltype(KHC)::value;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[MI * 2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j][KH * 2 + kk], af[i][kk], acc[MI * 2 + i][j], 0, 0, 0);
            };
            auto load_a = [&](int mi, int s_lo, int s_hi) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i][0] = *(const bf16x8*)(sb + a_row_off + (mi * 64 + i * 32) * P8_ROWB + s_lo);
                    af[i][1] = *(const bf16x8*)(sb + a_row_off + (mi * 64 + i * 32) * P8_ROWB + s_hi);
                }
            };
            // ---- phase 1: pixel half 0, K half 0.  reads A0 + B(kh0); stages BL(t+1)
            if constexpr (STAMP) { if (pst) est[0] = __builtin_readcyclecounter(); }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bfr[j][0] = *(const bf16x8*)(sb + b_row_off + j * 32 * P8_ROWB + so0);
                bfr[j][1] = *(const bf16x8*)(sb + b_row_off + j * 32 * P8_ROWB + so1);
            }
            load_a(0, so0, so1);
            if (has1) stage_b(nb, 0);
            P8_BAR();
            if constexpr (STAMP) { if (pst) est[1] = __builtin_readcyclecounter(); }
            mfma_phase(IC<0>{}, IC<0>{});
            P8_BAR();
            if constexpr (STAMP) { if (pst) est[2] = __builtin_readcyclecounter(); }
            // ---- phase 2: pixel half 0, K half 1.  reads A0 + B(kh1); stages BH(t+1); A1(t) must have landed before phase 3
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bfr[j][2] = *(const bf16x8*)(sb + b_row_off + j * 32 * P8_ROWB + so2);
                bfr[j][3] = *(const bf16x8*)(sb + b_row_off + j * 32 * P8_ROWB + so3);
            }
            load_a(0, so2, so3);
            if (has1) {
                stage_b(nb, 1);
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // newer than A1(t): A0(t+1), BL(t+1), BH(t+1)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            P8_BAR();
            if constexpr (STAMP) { if (pst) est[3] = __builtin_readcyclecounter(); }
            mfma_phase(IC<0>{}, IC<1>{});
            P8_BAR();
            if constexpr (STAMP) { if (pst) est[4] = __builtin_readcyclecounter(); }
            // ---- phase 3: pixel half 1, K half 0.  reads A1; stages A1(t+1)
            load_a(1, so0, so1);
            if (has1) stage_a(nb, 1);
            P8_BAR();
            if constexpr (STAMP) { if (pst) est[5] = __builtin_readcyclecounter(); }
            mfma_phase(IC<1>{}, IC<0>{});
            P8_BAR();
            if constexpr (STAMP) { if (pst) est[6] = __builtin_readcyclecounter(); }
            // ---- phase 4: pixel half 1, K half 1.  reads A1; stages A0(t+2) into THIS buffer (A0 of tile t is dead since phase 2);
            //      A0 / BL / BH of tile t+1 must have landed before the next tile's phase 1
            load_a(1, so2, so3);
            if (has1) advance();                                  // cursor -> tile t+2
            if (has2) {
                stage_a(sb, 0);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // newer than BH(t+1): A1(t+1), A0(t+2)
            } else if (has1) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // newer than BH(t+1): A1(t+1)
            }
            P8_BAR();
            if constexpr (STAMP) { if (pst) est[7] = __builtin_readcyclecounter(); }
            mfma_phase(IC<1>{}, IC<1>{});
            P8_BAR();
            if constexpr (STAMP) { if (pst) est[8] = __builtin_readcyclecounter(); }
        };
        {
            int t = 0;
            for (; t + 1 < nkl; t += 2) {
                ktile(IC<0>{}, t);
                ktile(IC<1>{}, t + 1);
            }
            if (t < nkl) ktile(IC<0>{}, t);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();               // pairs with the lagging group's last barrier
        else __builtin_amdgcn_s_setprio(0);
        P8_LDS_SYNC();                                           // every fragment read of the tile is done: both buffers are free
        if constexpr (STAMP) {
            if (stamp_now) ts[2] = __builtin_readcyclecounter();
        }
        // ---- next tile's addresses and the DMAs of its K-tile 0 (buffer 0) BEFORE this tile's epilogue (staging above buffer 0)
        const int cur_m0 = m0, cur_n0 = n0;
        if constexpr (SK) cur += nkl;
        const int next = SK ? tile + 1 : tile + stride;
        const bool more = SK ? cur < ue : next < run_end;
        if constexpr (SK) {
            if (nkl < nk) {
                // ---- a PARTIAL piece of tile `tile`: arrive.  Contributors = the XCD's workgroup slots whose ranges meet the tile's units
                //      [a, b): s_lo = the first slot with end > a, s_hi = the last slot with begin < b (ranges: [floor(U s / S), floor(U (s + 1) / S)))
                const long long ua = (long long)(tile - run_begin) * nk, ub_ = ua + nk;
                const int s_lo = (int)(((ua + 1) * stride + sk_units - 1) / sk_units) - 1;
                const int s_hi = (int)((ub_ * stride + sk_units - 1) / sk_units) - 1;
                const int ncon = s_hi - s_lo + 1;
                volatile int* flag = reinterpret_cast<volatile int*>(lds + P8_FLAG_OFF);
                int* cnt = reinterpret_cast<int*>(p.sk_ws) + tile;
                auto slab_of = [&](int s_) {                     // byte offset of slot s_'s slab for THIS tile: slab 0 = its range's first piece
                    const int first_tile = run_begin + sk_begin(s_) / nk;
                    return (unsigned)P8_SK_HDR + (unsigned)(((s_ * 8 + xcd) * 2 + (first_tile == tile ? 0 : 1))) * (unsigned)P8_SK_SLAB;
                };
                bool last = false;
                if (kb == 0) {
                    // the head piece is the LAST piece of this workgroup's range: the others (tail / middle pieces, done first in their
                    // ranges) have normally arrived long ago - look before writing 256 KB nobody would read
                    if (tid == 0) flag[0] = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    P8_LDS_SYNC();
                    last = __builtin_amdgcn_readfirstlane(flag[0]) == ncon - 1;     // (uniform by construction - say so, or every loop bound downstream becomes a VGPR)
                    P8_LDS_SYNC();
                }
                if (!last) {
                    const unsigned so = slab_of(slot);
#pragma unroll
                    for (int b = 0; b < 8; ++b)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x16& a = acc[b >> 1][b & 1];
                            const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rws, tid * 16, (int)(so + (unsigned)(b * 4 + q) * 8192u), 16);
                        }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every slab store of this wave is written through
                    P8_LDS_SYNC();                                           // ... of every wave
                    if (tid == 0) flag[0] = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    P8_LDS_SYNC();
                    last = __builtin_amdgcn_readfirstlane(flag[0]) == ncon - 1;     // (uniform by construction - say so, or every loop bound downstream becomes a VGPR)
                    P8_LDS_SYNC();
                }
                if (!last) {
                    // somebody else finishes this tile.  Next piece (if any): its first DMAs, waited for at the loop top like a first tile's
                    if (!more) break;
                    tile = next; kb = 0; first = true; ++tile_no;
                    set_tile(tile);
                    stage_first();
                    continue;
                }
                if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero again for the next launch
                // ---- reduce: acc = sum over the contributors in K (= slot) order.  With this piece at position 0 or 1 of that order the
                //      running sum can live in acc itself ((me + c1) + c2 ... and (c0 + me) + c2 ... are the canonical (c0 + c1) + c2 ...
                //      with the first pair commuted); from position 2 on the pieces before it are summed separately first.
                const int me = slot - s_lo;
                auto add_slab = [&](int s_, auto COPYC) {
                    constexpr bool COPY = decltype(COPYC)::value != 0;
                    const unsigned so = slab_of(s_);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        u32x4 t[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) t[e] = __builtin_amdgcn_raw_buffer_load_b128(rws, tid * 16, (int)(so + (unsigned)(g * 8 + e) * 8192u), 16);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int b = (g * 8 + e) >> 2, q = (g * 8 + e) & 3;
                            const f32x4 v = __builtin_bit_cast(f32x4, t[e]);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if constexpr (COPY) acc[b >> 1][b & 1][4 * q + r] = v[r];
                                else acc[b >> 1][b & 1][4 * q + r] += v[r];
                            }
                        }
                    }
                };
                if (me >= 2) {
                    // (a piece at position >= 1 came through the fetch_add path: its own slab is in the workspace, bit for bit the registers)
                    add_slab(s_lo, IC<1>{});
                    for (int c = 1; c < ncon; ++c) add_slab(s_lo + c, IC<0>{});
                } else {
                    for (int c = 0; c < ncon; ++c)
                        if (c != me) add_slab(s_lo + c, IC<0>{});
                }
            }
        }
        constexpr bool EPI16 = EPI >= 1 && EPI <= 3;             // arithmetic in the accumulator layout + bf16 staging
        P8EpiRegs epr;
        P8EpiRegs16 epr16;
        P8Epi4Regs epr4;
        if constexpr (EPI16) p8_epilogue16_prefetch(epr16, p, cur_n0, wc, lane);
        else if constexpr (EPI == 4) p8_epilogue4_prefetch(epr4, p, rres, cur_m0, cur_n0, tid);
        else p8_epilogue_prefetch<EPI>(epr, p, cur_m0, cur_n0, tid);  // BEFORE the DMAs (in-order vmcnt)
        if constexpr (STAMP) {
            if (stamp_now) est[9] = __builtin_readcyclecounter();
        }
        if constexpr (EPI == 4) {
            // straight-line form: the four DMA pieces are ALWAYS issued (after the last tile with every row out of bounds: zeros
            // into buffer 0, nobody reads them), so the compiler counts every vmcnt of the epilogue exactly - the residual rows of
            // pass 0 are waited for with the 8 DMAs (and the next pass's rows) still in flight
            if (more) {
                set_tile(next);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 2; ++j) { a_mask[h][j] = 0u; b_voff[h][j] = OOB; }
                cur_tap = 0; cur_kw = 0; cur_c0 = 0; cur_tapoff = 0u; cur_k0b = 0u;
            }
            stage_first();
        } else if (more) {
            set_tile(next);                                      // ALU work under the latency of the prefetch loads
            if constexpr (STAMP) {
                if (stamp_now) est[10] = __builtin_readcyclecounter();
            }
            // retire the prefetch loads HERE: behind the conditional DMAs the compiler cannot count (0 or 8 younger operations) and
            // would wait vmcnt(0) at the first use of scale / bias - i.e. for the DMAs' HBM round trip (measured: 12 k cycles)
            if constexpr (EPI16) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) asm volatile("" :: "v"(epr16.sc[j][q][e]), "v"(epr16.bs[j][q][e]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" :: "v"(epr.sc[e]), "v"(epr.bs[e]));
            }
            if (EPI == 0 && p.res && p.out_dt != NPS_DT_F32) {
#pragma unroll
                for (int it = 0; it < 4; ++it) asm volatile("" :: "v"(epr.r0[it]));
            }
            stage_first();
            if constexpr (STAMP) {
                if (stamp_now) est[11] = __builtin_readcyclecounter();
            }
        }
        if constexpr (EPI16) p8_epilogue16<EPI>(acc, lds, p, ry, cur_m0, cur_n0, wr, wc, lane, tid, more, epr16);
        else if constexpr (EPI == 4) p8_epilogue4(acc, lds, p, rres, ry, cur_m0, cur_n0, wr, wc, lane, tid, epr4);
        else p8_epilogue<STAMP, EPI>(acc, lds, p, cur_m0, cur_n0, wr, wc, lane, tid, more, epr, est);
        if constexpr (STAMP) {
            if (stamp_now) {
                ts[3] = __builtin_readcyclecounter();
                if (p.dbg && lane == 0) {
                    unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 16;
                    d[0] = ts[0]; d[1] = ts[1]; d[2] = ts[2]; d[3] = ts[3];
#pragma unroll
                    for (int e = 0; e < 12; ++e) d[4 + e] = est[e];
                }
            }
        }
        if (!more) break;
        tile = next;
        first = false;
        if constexpr (SK) kb = 0;
        ++tile_no;
        if constexpr (STAMP) {
            if (tile_no == p.dbg_tile) ts[0] = __builtin_readcyclecounter();     // "prologue" of a later tile = from here to its first barrier
        }
    }
#undef P8_BAR
#endif
}

}  // namespace nps

static unsigned long long* g_p8_dbg = nullptr;
// tuning aid (not declared in the public header): cycle-stamp buffer for the variant-24 build, [workgroups][8 waves][16] u64
static int g_p8_dbg_tile = 0;
extern "C" void nps_p8_debug_buffer(void* buf) { g_p8_dbg = (unsigned long long*)buf; }
extern "C" void nps_p8_debug_tile(int tile_no) { g_p8_dbg_tile = tile_no; }     // which tile of every workgroup gets stamped

static int p8_num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (cus[dev] == 0) {
        hipDeviceProp_t prop;
        cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cus[dev];
}

int nps_p8_num_cus() { return p8_num_cus(); }             // (conv_p8n.hip)

// x [B,H,W,Cin] bf16 (pixel stride x_cstride), w [Cout][KH][KW][Cin] bf16 (plain K-contiguous rows - NOT fragment-major),
// Cin % 64 == 0, Cout % 256 == 0; epilogue = nopesac_conv2d_nhwc's (scale / bias / residual / activation / output dtype).
// ws != nullptr: the stream-K build (nopesac_conv2d_nhwc_p8_sk).
static int p8_launch(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                     int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride,
                     int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int variant, void* ws, int64_t ws_bytes, void* stream) {
    using namespace nps;
    const bool sk = ws != nullptr;
    NPS_CHECK_ARG(x && w && y, "conv2d_p8: null pointer");
    NPS_CHECK_ARG(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && KH * KW <= 32, "conv2d_p8: bad dims");
    NPS_CHECK_ARG(Cin > 0 && Cin % 64 == 0 && Cout > 0 && Cout % 256 == 0, "conv2d_p8: needs Cin %% 64 == 0 and Cout %% 256 == 0");
    NPS_CHECK_ARG(out_dt == NPS_DT_F32 || out_dt == NPS_DT_BF16 || out_dt == NPS_DT_FP8, "conv2d_p8: bad out_dt %d", out_dt);
    const int kmajor = (variant >> 5) & 1;      // + 32: channel-major K order
    const int generic_epi = (variant >> 6) & 1; // + 64: force the generic (run-time decided) epilogue build
    const int grid_cap = variant >> 8;          // tuning aid: (cap << 8) limits the number of persistent workgroups
    variant &= 0x9f;
    NPS_CHECK_ARG(variant == 0 || (variant == 24 && !sk), "conv2d_p8: variant must be 0 (+32: channel-major K order); 24 = cycle-stamp build");
    NPS_CHECK_ARG(x_cstride >= Cin && x_cstride % 8 == 0 && y_cstride >= Cout && ((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0),
                  "conv2d_p8: strides / alignment");
    NPS_CHECK_ARG(!residual || (r_cstride >= Cout && out_dt != NPS_DT_FP8), "conv2d_p8: residual stride / residual with fp8 output");
    const int res_after = (act & NPS_ACT_RES_AFTER) ? 1 : 0;
    NPS_CHECK_ARG((act & ~(0xff | NPS_ACT_RES_AFTER)) == 0, "conv2d_p8: unsupported act flags");
    act &= 0xff;
    NPS_CHECK_ARG(act >= 0 && act <= 3, "conv2d_p8: bad act %d", act);
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.res = residual; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (W + 2 * pad - KW) / stride + 1;
    NPS_CHECK_ARG(p.OH > 0 && p.OW > 0, "conv2d_p8: empty output");
    p.x_cs = x_cstride; p.y_cs = y_cstride; p.r_cs = r_cstride; p.w_bs = 0;
    p.rows_per_b = p.OH * p.OW;
    p.M = B * p.rows_per_b; p.N = Cout; p.K = KH * KW * Cin;
    p.act = act; p.out_dt = out_dt; p.res_after = res_after; p.force = kmajor;
    NPS_CHECK_ARG((long long)B * H * W * x_cstride * 2 + ((long long)pad * W + pad) * x_cstride * 2 < (1ll << 31), "conv2d_p8: input larger than 2 GB");
    NPS_CHECK_ARG((long long)p.N * p.K * 2 < (1ll << 31), "conv2d_p8: weights larger than 2 GB");
    NPS_CHECK_ARG(p.M < (1 << 23) && (long long)B * H * W < (1 << 24) && x_cstride < (1 << 24) && p.K < (1 << 24) && Cout < (1 << 24),
                  "conv2d_p8: pixel count / strides beyond the 24-bit index math of this kernel");
    {
        const int al = out_dt == NPS_DT_F32 ? 4 : 8;
        bool ok = (y_cstride % al == 0) && ((uintptr_t)y % 16 == 0);
        if (residual) ok = ok && (r_cstride % al == 0) && ((uintptr_t)residual % 16 == 0);
        if (scale) ok = ok && ((uintptr_t)scale % 16 == 0);
        if (bias) ok = ok && ((uintptr_t)bias % 16 == 0);
        p.epi_vec = ok ? 1 : 0;
        NPS_CHECK_ARG(ok, "conv2d_p8: y / residual / scale / bias must be 16-byte aligned with 8-channel-aligned strides");
    }
    p.tiles_m = (p.M + P8_BM - 1) / P8_BM;
    p.tiles_n = p.N / P8_BN;
    const int ntiles = p.tiles_m * p.tiles_n;
    int nwg = p8_num_cus();                        // persistent: one workgroup per CU (129 KB of LDS each)
    if (grid_cap > 0 && grid_cap < nwg) nwg = grid_cap;
    if (sk) {
        // stream-K: every XCD's run of tiles must hold at least one K-tile unit per workgroup of that XCD (smallest run x K-tiles >= the
        // largest per-XCD workgroup count), else fewer workgroups
        const int nkt = p.K / P8_BK;
        NPS_CHECK_ARG(ntiles <= P8_SK_MAX_TILES, "conv2d_p8_sk: more than %d tiles (use the plain kernel: nothing to balance)", P8_SK_MAX_TILES);
        NPS_CHECK_ARG((uintptr_t)ws % 16 == 0, "conv2d_p8_sk: workspace alignment");
        while (nwg > 1) {
            const int nx = nwg < 8 ? nwg : 8;
            if ((long long)(ntiles / nx) * nkt >= (nwg + 7) / 8 && ntiles >= nx) break;
            --nwg;
        }
        const long long need = (long long)P8_SK_HDR + (long long)nwg * 2 * P8_SK_SLAB;
        NPS_CHECK_ARG(ws_bytes >= need && need < (1ll << 31), "conv2d_p8_sk: workspace of %lld bytes, %lld needed", (long long)ws_bytes, need);
        p.sk_ws = ws;
        p.sk_ws_bytes = (int)need;
    } else if (ntiles < nwg) nwg = ntiles;
    const dim3 grid(nwg);
    // epilogue specialisation (code size, see p8_epilogue): the common no-residual / bf16-output forms get their own build
    int epi = 0;
    // (the specialised builds address the output / residual rows with 32-bit buffer offsets)
    const bool off32 = ((long long)p.M + 256) * y_cstride * 2 < (1ll << 31) && (!residual || ((long long)p.M + 256) * r_cstride * 2 < (1ll << 31));
    if (!residual && out_dt == NPS_DT_BF16 && !res_after && off32) epi = act == NPS_ACT_RELU ? 1 : act == NPS_ACT_NONE ? 2 : act == NPS_ACT_LEAKY ? 3 : 0;
    if (residual && out_dt == NPS_DT_BF16 && !res_after && act == NPS_ACT_RELU && scale && bias && off32) epi = 4;
    if (generic_epi) epi = 0;
    const hipStream_t st = (hipStream_t)stream;
    if (sk) {
        if (epi == 1) hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 1, true>), grid, dim3(512), 0, st, p);
        else if (epi == 2) hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 2, true>), grid, dim3(512), 0, st, p);
        else if (epi == 3) hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 3, true>), grid, dim3(512), 0, st, p);
        else if (epi == 4) hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 4, true>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 0, true>), grid, dim3(512), 0, st, p);
    } else if (variant == 24) {
        p.dbg = g_p8_dbg; p.dbg_tile = g_p8_dbg_tile;
        if (epi == 1) hipLaunchKernelGGL((conv_igemm_p8_kernel<true, 1>), grid, dim3(512), 0, st, p);
        else hipLaunchKernelGGL((conv_igemm_p8_kernel<true, 0>), grid, dim3(512), 0, st, p);
    } else if (epi == 1) hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 1>), grid, dim3(512), 0, st, p);
    else if (epi == 2) hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 2>), grid, dim3(512), 0, st, p);
    else if (epi == 3) hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 3>), grid, dim3(512), 0, st, p);
    else if (epi == 4) hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 4>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_p8_kernel<false, 0>), grid, dim3(512), 0, st, p);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_conv2d_nhwc_p8(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                                      int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride,
                                      int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int variant, void* stream) {
    return p8_launch(x, w, scale, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, x_cstride, y_cstride, r_cstride, act, out_dt, variant,
                     nullptr, 0, stream);
}

// Stream-K form (conv_igemm_p8_kernel<.., SK>): same arguments + a workspace of nopesac_conv2d_p8_sk_workspace_bytes() bytes whose first
// 16 KB (the arrival counters) are ZERO on entry; they are zero again when the launch has completed, so one workspace serves every launch
// of a stream (launches on different streams need different workspaces).
extern "C" int64_t nopesac_conv2d_p8_sk_workspace_bytes(void) {
    return (int64_t)nps::P8_SK_HDR + (int64_t)p8_num_cus() * 2 * nps::P8_SK_SLAB;
}

extern "C" int nopesac_conv2d_nhwc_p8_sk(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                                         int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride,
                                         int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int variant, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
    NPS_CHECK_ARG(workspace, "conv2d_p8_sk: null workspace");
    return p8_launch(x, w, scale, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, x_cstride, y_cstride, r_cstride, act, out_dt, variant,
                     workspace, workspace_bytes, stream);
}
