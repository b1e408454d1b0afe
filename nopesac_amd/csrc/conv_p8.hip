// 256x256-tile implicit-GEMM NHWC convolution for the MFMA-bound bf16 layers (Cin % 64 == 0, Cout % 256 == 0): 8 waves,
// both operands through LDS-DMA, phase-interleaved schedule with counted vmcnt.
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
//     SIMD"); the younger half gets one static s_setprio 1;
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
//     the source chunk and on the ds_read_b128.
// Hazards (the rules of cdna_hip_programming.md "8-phase template"): a piece is re-staged >= 2 phases after its last
// ds_read (the lagging wave group reads one interval later and its reads retire after the following barrier); a staged piece
// is read >= 1 phase after the wait that retires it (both groups have then passed their own vmcnt and a common barrier).
#include "conv_common.h"

namespace nps {

constexpr int P8_BM = 256, P8_BN = 256, P8_BK = 64, P8_ROWB = 128;
constexpr int P8_A_BYTES = P8_BM * P8_ROWB, P8_B_BYTES = P8_BN * P8_ROWB, P8_STAGE = P8_A_BYTES + P8_B_BYTES;   // 32 + 32 KB
template <int K>
struct IC { static constexpr int value = K; };

constexpr int P8_ELD = P8_BN + 4;                                  // floats per row of the epilogue's staging buffer
constexpr int P8_EPI_BYTES = 128 * P8_ELD * 4;                    // 128 pixel rows x 256 channels f32 (+ bank pad) = 130 KB
constexpr int P8_LDS = 2 * P8_STAGE > P8_EPI_BYTES ? 2 * P8_STAGE : P8_EPI_BYTES;

// Epilogue of the 256x256 tile.  With ONE workgroup per CU nothing else runs while a tile is written out, so the generic
// conv_epilogue (4 passes of 64 columns in which only 2 of the 8 waves stage data, scale / bias re-loaded per 8-channel chunk:
// 43 k cycles per tile, measured with in-kernel cycle stamps - half of the K loop's 82 k) is replaced by:
//   two passes over the tile's pixel halves; in a pass EVERY wave writes its 2 x 2 accumulators (64 pixels x 64 channels)
//   into a [128 pixels][256 channels] f32 staging buffer (16 ds_write_b128 per lane, rows padded by 16 bytes: conflict-free);
//   thread t then owns the 8-channel chunk t % 32 of rows (t / 32) + 16 k: its scale / bias vectors are loaded ONCE (before
//   the first pass), the residual rows of a pass are prefetched before the staging writes, and every wave store instruction
//   writes two complete 512-byte pixel rows.
// Arithmetic order is conv_epilogue's (v * scale + bias, residual before or after the activation), results are identical.
// LDS-only workgroup barrier: orders this wave's LDS accesses (lgkmcnt) and synchronises, WITHOUT the vmcnt(0) that
// __syncthreads() carries - the epilogue's global stores of one pass must not be waited for before the next pass starts
// (measured: with __syncthreads() every pass pays the full store-acknowledge latency; all CUs reach their epilogues together).
#define P8_LDS_SYNC()                                          \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    } while (0)

template <bool STAMP = false, bool NOSTORE = false>
__device__ __forceinline__ void p8_epilogue(f32x16 (&acc)[4][2], unsigned char* lds, const ConvParams& p, int m0, int n0, int wr, int wc,
                                            int lane, int tid, unsigned long long* est = nullptr) {
    float* epi = reinterpret_cast<float*>(lds);
    const int c8 = tid & 31, rg = tid >> 5;
    const int n = n0 + c8 * 8;
    float sc[8], bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; bs[e] = 0.f; }
    if (p.scale) {
        *(f32x4*)(sc) = *(const f32x4*)(p.scale + n);
        *(f32x4*)(sc + 4) = *(const f32x4*)(p.scale + n + 4);
    }
    if (p.bias) {
        *(f32x4*)(bs) = *(const f32x4*)(p.bias + n);
        *(f32x4*)(bs + 4) = *(const f32x4*)(p.bias + n + 4);
    }
    const bool res16 = p.res && p.out_dt != NPS_DT_F32;           // bf16 residual rows: prefetched per pass
    auto do_pass = [&](auto PASSC) {
        constexpr int pass = decltype(PASSC)::value;
        auto row_of = [&](int it) { const int rs = it * 16 + rg; return m0 + (rs >> 6) * 128 + pass * 64 + (rs & 63); };
        us8 r8[8];
        if (res16) {                                            // residual rows of the pass: in flight during the staging writes
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                r8[it] = us8{0, 0, 0, 0, 0, 0, 0, 0};
                if (row_of(it) < p.M) r8[it] = *(const us8*)((const bf16_t*)p.res + (long long)row_of(it) * p.r_cs + n);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int rs = wr * 64 + i * 32 + (lane & 31);
                    const int c = wc * 64 + j * 32 + 8 * q + 4 * (lane >> 5);
                    const f32x16& a = acc[pass * 2 + i][j];
                    const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
                    *(f32x4*)(epi + rs * P8_ELD + c) = v;
                }
        if constexpr (STAMP) est[pass * 3 + 0] = __builtin_readcyclecounter();
        P8_LDS_SYNC();
        if constexpr (STAMP) est[pass * 3 + 1] = __builtin_readcyclecounter();
        // all 8 items of the thread at once (straight-line code, every wave-uniform decision - residual kind, activation,
        // output type - taken ONCE per pass around fully unrolled item loops: the LDS reads, the arithmetic and the stores of
        // different items overlap; the first version re-decided per element and was 20 k lines of ISA)
        float v[8][8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rs = it * 16 + rg;
            *(f32x4*)(v[it]) = *(const f32x4*)(epi + rs * P8_ELD + c8 * 8);
            *(f32x4*)(v[it] + 4) = *(const f32x4*)(epi + rs * P8_ELD + c8 * 8 + 4);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it)
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[it][e] *= sc[e]; v[it][e] += bs[e]; }
        const int act = p.act, res_after = p.res_after;
        auto add_res = [&]() {
            if (res16) {
#pragma unroll
                for (int it = 0; it < 8; ++it)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[it][e] += bf16_to_f32(r8[it][e]);
            } else if (p.res) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    if (row_of(it) < p.M) {
                        const float* rp = (const float*)p.res + (long long)row_of(it) * p.r_cs + n;
                        const f32x4 r0 = *(const f32x4*)(rp), r1 = *(const f32x4*)(rp + 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[it][e] += r0[e]; v[it][4 + e] += r1[e]; }
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < 8; ++it)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[it][e] += 0.f;             // conv_epilogue adds rv = 0 (turns -0 into +0)
            }
        };
        if (!res_after) add_res();
        if (act == NPS_ACT_RELU) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = v[it][e] > 0.f ? v[it][e] : 0.f;
        } else if (act == NPS_ACT_LEAKY) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = v[it][e] > 0.f ? v[it][e] : 0.01f * v[it][e];
        } else if (act == NPS_ACT_SIGMOID) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[it][e] = 1.f / (1.f + expf(-v[it][e]));
        }
        if (res_after) add_res();
        if constexpr (NOSTORE) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                asm volatile("" :: "v"(v[it][0]), "v"(v[it][1]), "v"(v[it][2]), "v"(v[it][3]), "v"(v[it][4]), "v"(v[it][5]), "v"(v[it][6]), "v"(v[it][7]));
        } else if (p.out_dt == NPS_DT_F32) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                if (row_of(it) < p.M) {
                    float* yp = (float*)p.y + (long long)row_of(it) * p.y_cs + n;
                    *(f32x4*)(yp) = *(const f32x4*)(v[it]);
                    *(f32x4*)(yp + 4) = *(const f32x4*)(v[it] + 4);
                }
        } else if (p.out_dt == NPS_DT_FP8) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                if (row_of(it) < p.M) *(uint2*)((unsigned char*)p.y + (long long)row_of(it) * p.y_cs + n) = f32x8_to_fp8(v[it]);
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                if (row_of(it) < p.M) {
                    uint4 o;
                    o.x = f32x2_to_bf16x2(v[it][0], v[it][1]); o.y = f32x2_to_bf16x2(v[it][2], v[it][3]);
                    o.z = f32x2_to_bf16x2(v[it][4], v[it][5]); o.w = f32x2_to_bf16x2(v[it][6], v[it][7]);
                    *(uint4*)((bf16_t*)p.y + (long long)row_of(it) * p.y_cs + n) = o;
                }
        }
        if constexpr (STAMP) est[pass * 3 + 2] = __builtin_readcyclecounter();
        if constexpr (pass == 0) P8_LDS_SYNC();
    };
    do_pass(IC<0>{});
    do_pass(IC<1>{});
}

// PRIO: 0 = one static s_setprio 1 for the lagging (younger) half; 1 = s_setprio 1 around every MFMA cluster (both halves);
//       2 = no priority hints.  (A/B-able in one binary through nopesac_conv2d_nhwc_p8's `variant`.)
// ABL (ablation, timing experiments only - results are wrong): bit 0 = no A DMAs, bit 1 = no B DMAs, bit 2 = no fragment reads.
template <int PRIO, int ABL = 0>
__global__ __launch_bounds__(512) void conv_igemm_p8_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[P8_LDS];
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, within = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    }
    const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
    const int m0 = tile_m * P8_BM, n0 = tile_n * P8_BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // pixel half / channel quarter of the workgroup tile; group = wr
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    if constexpr (ABL & 8) ts0 = __builtin_readcyclecounter();

    // ---- DMA sources.  One 1-KB DMA = 8 tile rows x 128 B, lane -> row (lane >> 3), physical 16-byte slot (lane & 7).
    //   A piece h (pixel half mi = h of BOTH wave rows): DMA j of wave w fills rows j*128 + h*64 + w*8 .. +7
    //   B piece h (channels h*128 .. +127):              DMA j of wave w fills rows h*128 + (2w + j)*8 .. +7
    constexpr unsigned OOB = 0xFFFFFF00u;
    const int slot = lane & 7, rsub = lane >> 3;
    const long long padb = ((long long)p.pad * p.W + p.pad) * p.x_cs * 2;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x - padb), 0, (int)(((long long)p.B * p.H * p.W * p.x_cs) * 2 + padb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    unsigned a_voff[2][2], a_mask[2][2], b_voff[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pr = j * 128 + h * 64 + wave * 8 + rsub;
            const int m = m0 + pr;
            const int coff = (slot ^ ((pr >> 1) & 7)) * 8;
            a_voff[h][j] = OOB; a_mask[h][j] = 0u;
            if (m < p.M) {
                const int b = m / p.rows_per_b, rem = m % p.rows_per_b;
                const int oh = rem / p.OW, ow = rem % p.OW;
                const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
                a_voff[h][j] = (unsigned)((((long long)b * p.H + oh * p.stride) * p.W + ow * p.stride) * p.x_cs + coff) * 2u;
                unsigned mk = 0u;
                for (int kh = 0; kh < p.KH; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw)
                        if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) mk |= 1u << (kh * p.KW + kw);
                a_mask[h][j] = mk;
            }
            const int br = h * 128 + (wave * 2 + j) * 8 + rsub;
            const int n = n0 + br;
            b_voff[h][j] = n < p.N ? (unsigned)((long long)n * p.K + (slot ^ ((br >> 1) & 7)) * 8) * 2u : OOB;
        }
    // wave-uniform K-tile cursor of the NEXT tile to stage (tiles are staged strictly in order)
    int cur_tap = 0, cur_kw = 0, cur_c0 = 0;
    unsigned cur_tapoff = 0u, cur_k0b = 0u;
    // K order: p.force == 0: tap-major (all channels of tap 0, then tap 1 ...: the natural order of the weight rows);
    //          p.force == 1: CHANNEL-major (the KH*KW taps of channels 0-63, then of channels 64-127 ...): the pixels a workgroup
    //          re-reads for the nine taps of one 64-channel slice are ONE 128-byte line each, so the slice stays in the XCD's L2
    //          between the taps (tap-major streams all Cin channels of the tile's pixels nine times: 4.3 MB per XCD, L2 hit 76 %).
    const int ntaps = p.KH * p.KW;
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
    auto stage_a = [&](unsigned char* buf, int h) {          // pixel-half piece h at the cursor's K-tile
        if constexpr (ABL & 1) {
            asm volatile("" :: "s"(cur_tapoff), "s"(cur_c0));
            return;
        }
        const unsigned soff = cur_tapoff + (unsigned)cur_c0 * 2u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = ((a_mask[h][j] >> cur_tap) & 1u) ? a_voff[h][j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lptr_t)(buf + (j * 128 + h * 64 + wave * 8) * P8_ROWB), 16, vo, soff, 0, 0);
        }
    };
    auto stage_b = [&](unsigned char* buf, int h) {
        if constexpr (ABL & 2) {
            asm volatile("" :: "s"(cur_k0b));
            return;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lptr_t)(buf + P8_A_BYTES + (h * 128 + (wave * 2 + j) * 8) * P8_ROWB), 16,
                                                     b_voff[h][j], cur_k0b, 0, 0);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / P8_BK;
    const int sw = (lane >> 1) & 7;
    const int a_row_off = (wr * 128 + (lane & 31)) * P8_ROWB;
    const int b_row_off = P8_A_BYTES + (wc * 64 + (lane & 31)) * P8_ROWB;
    const int so0 = ((0 + (lane >> 5)) ^ sw) * 16, so1 = ((2 + (lane >> 5)) ^ sw) * 16, so2 = ((4 + (lane >> 5)) ^ sw) * 16,
              so3 = ((6 + (lane >> 5)) ^ sw) * 16;
    bf16x8 af[2][2], bfr[2][4];

    // ---- prologue: A0(0), BL(0), BH(0), A1(0), A0(1); tile 0's first three pieces must have landed before phase 1
    stage_a(lds, 0);
    stage_b(lds, 0);
    stage_b(lds, 1);
    stage_a(lds, 1);
    advance();                                               // cursor -> tile 1
    if (nk > 1) {
        stage_a(lds + P8_STAGE, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) {                                           // the lagging group: one barrier interval behind, higher priority
        if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(1);
        __builtin_amdgcn_s_barrier();
    }

    if constexpr (ABL & 8) ts1 = __builtin_readcyclecounter();
#define P8_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_s_barrier();              \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

    // one K-tile = 4 phases.  BUF = buffer of tile t; the cursor points at tile t+1 on entry and on exit at tile t+2.
    auto tile = [&](auto BUFC, int t) {
        constexpr int BUF = decltype(BUFC)::value;
        unsigned char* sb = lds + BUF * P8_STAGE;            // tile t (and tile t+2's A0 piece)
        unsigned char* nb = lds + (BUF ^ 1) * P8_STAGE;      // tile t+1
        const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
        auto mfma_phase = [&](auto MIC, auto KHC) {
            constexpr int MI = decltype(MIC)::value, KH = decltype(KHC)::value;
            if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[MI * 2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j][KH * 2 + kk], af[i][kk], acc[MI * 2 + i][j], 0, 0, 0);
            if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        };
        auto load_a = [&](int mi, int s_lo, int s_hi) {
            if constexpr (ABL & 4) {
                if (t != 0) return;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i][0] = *(const bf16x8*)(sb + a_row_off + (mi * 64 + i * 32) * P8_ROWB + s_lo);
                af[i][1] = *(const bf16x8*)(sb + a_row_off + (mi * 64 + i * 32) * P8_ROWB + s_hi);
            }
        };
        // ---- phase 1: pixel half 0, K half 0.  reads A0 + B(kh0); stages BL(t+1)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if ((ABL & 4) && t != 0) break;
            bfr[j][0] = *(const bf16x8*)(sb + b_row_off + j * 32 * P8_ROWB + so0);
            bfr[j][1] = *(const bf16x8*)(sb + b_row_off + j * 32 * P8_ROWB + so1);
        }
        load_a(0, so0, so1);
        if (has1) stage_b(nb, 0);
        P8_BAR();
        mfma_phase(IC<0>{}, IC<0>{});
        P8_BAR();
        // ---- phase 2: pixel half 0, K half 1.  reads A0 + B(kh1); stages BH(t+1); A1(t) must have landed before phase 3
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if ((ABL & 4) && t != 0) break;
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
        mfma_phase(IC<0>{}, IC<1>{});
        P8_BAR();
        // ---- phase 3: pixel half 1, K half 0.  reads A1; stages A1(t+1)
        load_a(1, so0, so1);
        if (has1) stage_a(nb, 1);
        P8_BAR();
        mfma_phase(IC<1>{}, IC<0>{});
        P8_BAR();
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
        mfma_phase(IC<1>{}, IC<1>{});
        P8_BAR();
    };
    {
        int t = 0;
        for (; t + 1 < nk; t += 2) {
            tile(IC<0>{}, t);
            tile(IC<1>{}, t + 1);
        }
        if (t < nk) tile(IC<0>{}, t);
    }
#undef P8_BAR
    if (wr == 0) __builtin_amdgcn_s_barrier();               // pairs with the lagging group's last barrier
    else if constexpr (PRIO == 0) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if constexpr (ABL & 8) ts2 = __builtin_readcyclecounter();
    if constexpr (ABL & 8) {
        unsigned long long est[6];
        p8_epilogue<true, (ABL & 16) != 0>(acc, lds, p, m0, n0, wr, wc, lane, tid, est);
        __syncthreads();
        ts3 = __builtin_readcyclecounter();
        if (p.dbg && lane == 0) {
            unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 10;
            d[0] = ts0; d[1] = ts1; d[2] = ts2; d[3] = ts3;
#pragma unroll
            for (int e = 0; e < 6; ++e) d[4 + e] = est[e];
        }
    } else {
        p8_epilogue(acc, lds, p, m0, n0, wr, wc, lane, tid);
    }
#endif
}

}  // namespace nps

// x [B,H,W,Cin] bf16 (pixel stride x_cstride), w [Cout][KH][KW][Cin] bf16 (plain K-contiguous rows - NOT fragment-major),
// Cin % 64 == 0, Cout % 256 == 0; epilogue = nopesac_conv2d_nhwc's (scale / bias / residual / activation / output dtype).
static unsigned long long* g_p8_dbg = nullptr;
// tuning aid (not declared in the public header): cycle-stamp buffer for the variant-24 build, [workgroups][8 waves][10] u64
extern "C" void nps_p8_debug_buffer(void* buf) { g_p8_dbg = (unsigned long long*)buf; }

extern "C" int nopesac_conv2d_nhwc_p8(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                                      int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride,
                                      int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int variant, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && w && y, "conv2d_p8: null pointer");
    NPS_CHECK_ARG(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && KH * KW <= 32, "conv2d_p8: bad dims");
    NPS_CHECK_ARG(Cin > 0 && Cin % 64 == 0 && Cout > 0 && Cout % 256 == 0, "conv2d_p8: needs Cin %% 64 == 0 and Cout %% 256 == 0");
    NPS_CHECK_ARG(out_dt == NPS_DT_F32 || out_dt == NPS_DT_BF16 || out_dt == NPS_DT_FP8, "conv2d_p8: bad out_dt %d", out_dt);
    const int kmajor = (variant >> 5) & 1;      // + 32: channel-major K order
    variant &= ~32;
    NPS_CHECK_ARG((variant >= 0 && variant <= 2) || (variant >= 16 && variant <= 31),
                  "conv2d_p8: variant must be 0 (static priority), 1 (per-cluster priority) or 2 (none); 16 + bits = ablation builds");
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
    const dim3 grid(p.tiles_m * p.tiles_n);
    if (variant == 0) hipLaunchKernelGGL(conv_igemm_p8_kernel<0>, grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (variant == 1) hipLaunchKernelGGL(conv_igemm_p8_kernel<1>, grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (variant == 2) hipLaunchKernelGGL(conv_igemm_p8_kernel<2>, grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (variant == 17) hipLaunchKernelGGL((conv_igemm_p8_kernel<0, 1>), grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (variant == 18) hipLaunchKernelGGL((conv_igemm_p8_kernel<0, 2>), grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (variant == 19) hipLaunchKernelGGL((conv_igemm_p8_kernel<0, 3>), grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (variant == 23) hipLaunchKernelGGL((conv_igemm_p8_kernel<0, 7>), grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (variant == 24) { p.dbg = g_p8_dbg; hipLaunchKernelGGL((conv_igemm_p8_kernel<0, 8>), grid, dim3(512), 0, (hipStream_t)stream, p); }
    else if (variant == 31) { p.dbg = g_p8_dbg; hipLaunchKernelGGL((conv_igemm_p8_kernel<0, 15>), grid, dim3(512), 0, (hipStream_t)stream, p); }
    else if (variant == 25) { p.dbg = g_p8_dbg; hipLaunchKernelGGL((conv_igemm_p8_kernel<0, 24>), grid, dim3(512), 0, (hipStream_t)stream, p); }
    else NPS_CHECK_ARG(false, "conv2d_p8: ablation variant not built");
    NPS_LAUNCH_RET();
}
