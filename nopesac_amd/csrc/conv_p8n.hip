// 256x128-tile implicit-GEMM NHWC convolution for the MFMA-bound bf16 layers with Cout % 128 == 0 that conv_igemm_p8_kernel cannot take
// (Cout % 256 != 0: the 3x3 128 -> 128 convs of res3 and of the top-down / pose-net stacks - 0.25-0.27 of the MFMA peak on
// conv_igemm_bfrag_kernel<4, 32> in round 4): the SAME machinery as conv_p8.hip - persistent 8-wave workgroups (one per CU) walking
// their XCD's run of tiles, both operands through LDS-DMA with counted vmcnt, the two waves of a SIMD staggered by one barrier interval
// (one issues 8 MFMAs = 256 cycles of the matrix pipe while the other reads fragments and issues DMAs) - on a tile that is half as wide.
//
// What changes with 128 output channels per tile (wave grid 2 pixel halves x 4 channel quarters, wave tile 128 x 32 = 4 x 1 accumulators):
//   * a K-tile (64 channels of one tap) is 16 MFMAs per wave = TWO phases of 8 (pixel half mi x all four k-steps; p8: four phases over
//     pixel half x K half).  The channel fragments of all four k-steps are read in phase 1 and stay in registers for phase 2
//     (12 + 8 ds_read_b128 per K-tile = 1.25 per MFMA; p8: 0.75 - LDS read time 640 of 1024 MFMA cycles per K-tile and CU);
//   * a K-tile is 48 KB in LDS (pixels 32 KB + weights 16 KB), staged as THREE groups of DMAs: G = {pixel half 0, weights} (4 per wave)
//     in phase 1 and H = {pixel half 1} (2 per wave) in phase 2 - 6 DMAs per 4.2 MFLOP (p8: 8 per 8.4);
//   * the ring is THREE K-tiles deep (144 KB): with two phases per K-tile a two-deep ring would leave one phase (~300 cycles) between a
//     DMA's issue and its wait; here tile t + 2 is staged while tile t computes, into the slot tile t - 1 left: G(t + 2) in phase 1 (its
//     pieces were last read in phase 1 of t - 1), H(t + 2) in phase 2 (last read in phase 2 of t - 1) - the rule of conv_p8.hip (a piece
//     is re-staged >= 2 phases after its last ds_read, read >= 1 phase after the wait that retires it) with four phases between issue and
//     use.  Issue order G0 H0 G1 H1 G2 ...: "H(t) landed" before phase 2 of t leaves G(t+1), H(t+1), G(t+2) in flight (vmcnt 10),
//     "G(t+1) landed" before the next tile leaves H(t+1), G(t+2), H(t+2) (vmcnt 8);
//   * tile boundary as in p8: after the K loop the NEXT tile's row table and its first TWO K-tiles (slots 0 and 1) are issued before the
//     epilogue, which stages through slot 2 (bf16, arithmetic in the accumulator layout, 16-byte stores through a bounds-checked buffer
//     descriptor, LDS-only barriers: stores are never waited for; the DMAs are the oldest operations in flight and are retired by one
//     counted wait in the last pass).
// SPLIT (round 6): split-K for the layers with FEW tiles and a LONG K loop (the pose net's first conv: 3x3, 2048 -> 128 at 15 x 20 x 64 images =
// 75 tiles x 288 K-tiles - 75 busy CUs for 230 us, or 221 us on 300 workgroups of the round-1 LDS-DMA kernel at 0.16 of the MFMA peak).  A
// work unit = (tile, slice s of the 64-channel groups): the K walk is channel-major, so a slice is a contiguous run of groups with all their
// taps; the unit's accumulators leave as raw f32 into workspace[s][M][N] (16-byte stores straight from the accumulator layout: a row's 32
// channels of one wave = one 128-byte line), and p8n_split_reduce_kernel sums the slices in fixed order and applies the epilogue.
// Epilogue forms: no residual, bf16 output, ReLU / none / LeakyReLU (every Cout = 128 layer on the path is one of these); anything else is
// rejected by the entry point and stays on the other conv kernels.
#include "conv_common.h"

namespace nps {

template <int K>
struct ICn { static constexpr int value = K; };

constexpr int N8_BM = 256, N8_BN = 128, N8_BK = 64, N8_ROWB = 128;
constexpr int N8_A_BYTES = N8_BM * N8_ROWB, N8_B_BYTES = N8_BN * N8_ROWB, N8_STAGE = N8_A_BYTES + N8_B_BYTES;   // 32 + 16 KB
constexpr int N8_NSTAGE = 3;
constexpr int N8_ELD16 = N8_BN + 8;                                 // bf16 elements per staged epilogue row (272 bytes)
constexpr int N8_EPI_OFF = 2 * N8_STAGE;                            // the epilogue stages through ring slot 2 (slots 0 / 1: the next tile's first K-tiles)
constexpr int N8_TAB_OFF = N8_NSTAGE * N8_STAGE;                    // row table above the ring
constexpr int N8_LDS = N8_TAB_OFF + N8_BM * 8;                      // 146 KB

typedef unsigned int n8_u32x4 __attribute__((ext_vector_type(4)));

#define N8_LDS_SYNC()                                          \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_s_barrier();                          \
        asm volatile("" ::: "memory");                         \
    } while (0)

#define N8_BAR()                                   \
    do {                                           \
        __builtin_amdgcn_sched_barrier(0);         \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_s_barrier();              \
        asm volatile("" ::: "memory");             \
        __builtin_amdgcn_sched_barrier(0);         \
    } while (0)

struct N8EpiRegs {
    float sc[4][4], bs[4][4];      // the lane's 16 channels: wc * 32 + 8 q + 4 (lane >> 5) + e
};

// ACT: NPS_ACT_RELU / NONE / LEAKY fixed at compile time
template <int ACT, bool SPLIT = false>
__global__ __launch_bounds__(512) void conv_igemm_p8n_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[N8_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                 // pixel half / 32-channel quarter of the workgroup tile; stagger group = wr

    // ---- tiles of this workgroup: XCD x (= blockIdx % 8) owns a contiguous run, its workgroups walk it with stride = workgroups on that XCD
    const int ntiles = p.tiles_m * p.tiles_n * (SPLIT ? p.ksplit : 1);       // SPLIT: work units (tile, K slice), slice fastest
    const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int tq = ntiles / nx, tr = ntiles % nx;
    const int run_begin = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int run_end = run_begin + tq + (xcd < tr ? 1 : 0);
    const int stride = ((int)gridDim.x - xcd + 7) / 8;
    int tile = run_begin + slot;
    if (tile >= run_end) return;

    // ---- DMA sources (as conv_p8.hip): one 1-KB DMA = 8 tile rows x 128 B, lane -> row (lane >> 3), physical 16-byte slot (lane & 7)
    //   A piece h (pixel half mi = h of BOTH wave rows): DMA j of wave w fills rows j*128 + h*64 + w*8 .. +7
    //   B (the tile's 128 channels):                       DMA j of wave w fills rows (2w + j)*8 .. +7
    constexpr unsigned OOB = 0xFFFFFF00u;
    const int slot8 = lane & 7, rsub = lane >> 3;
    const long long padb = ((long long)p.pad * p.W + p.pad) * p.x_cs * 2;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.x - padb), 0, (int)(((long long)p.B * p.H * p.W * p.x_cs) * 2 + padb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long long)p.N * p.K * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((((long long)p.M - 1) * p.y_cs + p.N) * 2), 0x00020000);
    unsigned a_voff[2][2], a_mask[2][2], b_voff[2];
    int m0 = 0, n0 = 0;
    int cur_tap = 0, cur_kw = 0, cur_c0 = 0;                 // wave-uniform K-tile cursor of the NEXT K-tile to stage
    unsigned cur_tapoff = 0u, cur_k0b = 0u;
    const float rcp_rpb = 1.0f / (float)p.rows_per_b, rcp_ow = 1.0f / (float)p.OW;
    auto divmod = [](int a, int d, float rcp, int& q, int& r) {
        q = (int)((float)a * rcp);
        r = a - __mul24(q, d);
        if (r < 0) { --q; r += d; }
        if (r >= d) { ++q; r -= d; }
    };
    const unsigned xcs = (unsigned)p.x_cs;
    uint2* rowtab = reinterpret_cast<uint2*>(lds + N8_TAB_OFF);         // [256] {byte offset of the row's pixel, tap mask}
    unsigned a_coff[2][2], b_base[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pr = j * 128 + h * 64 + wave * 8 + rsub;
            a_coff[h][j] = (unsigned)(slot8 ^ ((pr >> 1) & 7)) * 16u;
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int br = (wave * 2 + j) * 8 + rsub;
        b_base[j] = (__umul24((unsigned)br, (unsigned)p.K) + (unsigned)(slot8 ^ ((br >> 1) & 7)) * 8u) * 2u;
    }
    const int ntaps = p.KH * p.KW;
    int nk = p.K / N8_BK;                                     // K-tiles of the current work unit (SPLIT: of its slice)
    int ks = 0;                                               // SPLIT: the unit's slice
    auto set_tile = [&](int t) {                                       // workgroup-collective (contains a barrier)
        int c_begin = 0;
        if constexpr (SPLIT) {
            const int tt = t / p.ksplit, ngroups = p.Cin / N8_BK;
            ks = t - tt * p.ksplit;
            c_begin = (ks * ngroups) / p.ksplit;              // 64-channel groups [c_begin, c_end) with all their taps (channel-major K order)
            nk = (((ks + 1) * ngroups) / p.ksplit - c_begin) * ntaps;
            c_begin *= N8_BK;
            t = tt;
        }
        m0 = (t / p.tiles_n) * N8_BM;
        n0 = (t % p.tiles_n) * N8_BN;
        if (tid < N8_BM) {
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
        N8_LDS_SYNC();
        const unsigned nb = __umul24((unsigned)n0, (unsigned)p.K) * 2u;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint2 e = rowtab[j * 128 + h * 64 + wave * 8 + rsub];
                a_voff[h][j] = e.x + a_coff[h][j];
                a_mask[h][j] = e.y;
            }
#pragma unroll
        for (int j = 0; j < 2; ++j) b_voff[j] = b_base[j] + nb;          // Cout % 128 == 0: every weight row of the tile exists
        cur_tap = 0; cur_kw = 0; cur_c0 = c_begin; cur_tapoff = 0u; cur_k0b = (unsigned)c_begin * 2u;
    };
    // K order: p.force == 0 tap-major, 1 channel-major (the KH*KW taps of a 64-channel slice back to back: the slice stays in L2)
    auto advance = [&]() {
        if (p.force == 0) {
            cur_k0b += N8_BK * 2;
            cur_c0 += N8_BK;
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
                cur_c0 += N8_BK;
                cur_k0b = (unsigned)cur_c0 * 2u;
            }
        }
    };
    auto stage_a = [&](unsigned char* buf, int h) {          // pixel-half piece h at the cursor's K-tile
        const unsigned soff = cur_tapoff + (unsigned)cur_c0 * 2u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = ((a_mask[h][j] >> cur_tap) & 1u) ? a_voff[h][j] : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lptr_t)(buf + (j * 128 + h * 64 + wave * 8) * N8_ROWB), 16, vo, soff, 0, 0);
        }
    };
    auto stage_b = [&](unsigned char* buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wsrc, (lptr_t)(buf + N8_A_BYTES + ((wave * 2 + j) * 8) * N8_ROWB), 16, b_voff[j], cur_k0b, 0, 0);
    };
    auto stage_first = [&]() {                               // G0 H0 [G1 H1] -> slots 0 [, 1]; leaves the cursor at K-tile min(2, nk)
        stage_a(lds, 0); stage_b(lds); stage_a(lds, 1);
        advance();
        if (nk > 1) {
            stage_a(lds + N8_STAGE, 0); stage_b(lds + N8_STAGE); stage_a(lds + N8_STAGE, 1);
            advance();
        }
    };

    const int sw = (lane >> 1) & 7;
    const int a_row_off = (wr * 128 + (lane & 31)) * N8_ROWB;
    const int b_row_off = N8_A_BYTES + (wc * 32 + (lane & 31)) * N8_ROWB;
    int so[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) so[ks] = ((2 * ks + (lane >> 5)) ^ sw) * 16;

    set_tile(tile);
    stage_first();
    bool first = true;
    while (true) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 af[2][4], bfr[4];
        if (first) {                                             // G0 must have landed; H0 [, G1, H1] may fly
            if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wr == 1) {                                           // the lagging group: one barrier interval behind, higher priority
            __builtin_amdgcn_s_setprio(1);
            __builtin_amdgcn_s_barrier();
        }
        // one K-tile = 2 phases.  SLOT = ring slot of tile t; the cursor points at tile t+2 on entry and at t+3 on exit.
        auto ktile = [&](auto SLOTC, int t) {
            constexpr int SLOT = decltype(SLOTC)::value;
            unsigned char* sb = lds + SLOT * N8_STAGE;                          // tile t
            unsigned char* fb = lds + ((SLOT + 2) % N8_NSTAGE) * N8_STAGE;      // tile t+2 (the slot tile t-1 left)
            const bool has1 = t + 1 < nk, has2 = t + 2 < nk;
            auto load_a = [&](int mi) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        af[i][ks] = *(const bf16x8*)(sb + a_row_off + (mi * 64 + i * 32) * N8_ROWB + so[ks]);
            };
            auto mfma_phase = [&](auto MIC) {
                constexpr int MI = decltype(MIC)::value;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[MI * 2 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks], af[i][ks], acc[MI * 2 + i], 0, 0, 0);
            };
            // ---- phase 1: pixel half 0, the whole K-tile.  reads A0 + B; stages G(t+2); H(t) must have landed before phase 2
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bfr[ks] = *(const bf16x8*)(sb + b_row_off + so[ks]);
            load_a(0);
            if (has2) {
                stage_a(fb, 0);
                stage_b(fb);
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");    // newer than H(t): G(t+1), H(t+1), G(t+2)
            } else if (has1) {
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");     // newer than H(t): G(t+1), H(t+1)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            N8_BAR();
            mfma_phase(ICn<0>{});
            N8_BAR();
            // ---- phase 2: pixel half 1.  reads A1; stages H(t+2); G(t+1) must have landed before the next tile
            load_a(1);
            if (has2) {
                stage_a(fb, 1);
                advance();                                            // cursor -> tile t+3
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // newer than G(t+1): H(t+1), G(t+2), H(t+2)
            } else if (has1) {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");     // newer than G(t+1): H(t+1)
            }
            N8_BAR();
            mfma_phase(ICn<1>{});
            N8_BAR();
        };
        {
            int t = 0;
            for (; t + 2 < nk; t += 3) {
                ktile(ICn<0>{}, t);
                ktile(ICn<1>{}, t + 1);
                ktile(ICn<2>{}, t + 2);
            }
            if (t < nk) ktile(ICn<0>{}, t);
            if (t + 1 < nk) ktile(ICn<1>{}, t + 1);
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();               // pairs with the lagging group's last barrier
        else __builtin_amdgcn_s_setprio(0);
        N8_LDS_SYNC();                                           // every fragment read of the tile is done: the ring is free

        // ---- next tile's addresses and its first two K-tiles (slots 0, 1) BEFORE this tile's epilogue (staging in slot 2)
        const int cur_m0 = m0, cur_n0 = n0, cur_ks = ks;
        const int next = tile + stride;
        const bool more = next < run_end;
        if constexpr (SPLIT) {
            // raw f32 partial tile -> workspace[slice][M][N]: lane (pixel l31, half) holds channels wc * 32 + 8 q + 4 half + e of its pixel -
            // 16 stores of 16 bytes per lane, unconditional (rows >= M go to an out-of-range offset the descriptor drops)
            if (more) { set_tile(next); stage_first(); }
            const __amdgpu_buffer_rsrc_t rws = __builtin_amdgcn_make_buffer_rsrc(p.sk_ws, 0, p.sk_ws_bytes, 0x00020000);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const unsigned m = (unsigned)(cur_m0 + wr * 128 + a * 32 + (lane & 31));
                const unsigned row = ((unsigned)cur_ks * (unsigned)p.M + m) * (unsigned)p.N + (unsigned)(cur_n0 + wc * 32 + 4 * (lane >> 5));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[a][4 * q], acc[a][4 * q + 1], acc[a][4 * q + 2], acc[a][4 * q + 3]};
                    const unsigned off = m < (unsigned)p.M ? (row + 8u * q) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(n8_u32x4, v), rws, (int)off, 0, 0);
                }
            }
            if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the next unit's 12 DMAs are older than these 16 stores
            if (!more) break;
            tile = next;
            first = false;
            continue;
        }
        N8EpiRegs R;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = cur_n0 + wc * 32 + 8 * q + 4 * (lane >> 5);
            f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) s4 = *(const f32x4*)(p.scale + n);
            if (p.bias) b4 = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) { R.sc[q][e] = s4[e]; R.bs[q][e] = b4[e]; }
        }
        if (more) {
            set_tile(next);                                      // ALU work under the latency of the scale / bias loads
            // retire the scale / bias loads HERE: behind the conditional DMAs the compiler could only wait vmcnt(0) at their first use
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("" :: "v"(R.sc[q][e]), "v"(R.bs[q][e]));
            stage_first();
        }
        // ---- epilogue: four passes, pass q = accumulator row tile q of BOTH pixel halves (64 rows x 128 channels, bf16, rows padded by 16 B)
        {
            bf16_t* epi = reinterpret_cast<bf16_t*>(lds + N8_EPI_OFF);
            const int c8 = tid & 15, rg = tid >> 4;              // 16 chunks of 8 channels, 32 row groups
            auto do_pass = [&](auto PASSC) {
                constexpr int pass = decltype(PASSC)::value;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = acc[pass][4 * q + e] * R.sc[q][e];
                        t += R.bs[q][e];
                        t += 0.f;                                // the generic epilogue adds the (absent) residual 0: -0 -> +0
                        if (ACT == NPS_ACT_RELU) t = t > 0.f ? t : 0.f;
                        else if (ACT == NPS_ACT_LEAKY) t = t > 0.f ? t : 0.01f * t;
                        v[e] = t;
                    }
                    const uint2 o = make_uint2(f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3]));
                    *(uint2*)(epi + (wr * 32 + (lane & 31)) * N8_ELD16 + wc * 32 + 8 * q + 4 * (lane >> 5)) = o;
                }
                if constexpr (pass == 3) {
                    // the next tile's DMAs (12 per wave, issued before pass 0) are the OLDEST operations in flight; younger are exactly the
                    // 6 stores of passes 0-2 (unconditional: rows >= M are dropped by the descriptor's bounds check)
                    if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                }
                N8_LDS_SYNC();
                n8_u32x4 o[2];
#pragma unroll
                for (int it = 0; it < 2; ++it) o[it] = *(const n8_u32x4*)(epi + (it * 32 + rg) * N8_ELD16 + c8 * 8);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const unsigned m = (unsigned)(cur_m0 + it * 128 + pass * 32 + rg);
                    __builtin_amdgcn_raw_buffer_store_b128(o[it], ry, (int)((m * (unsigned)p.y_cs + (unsigned)(cur_n0 + c8 * 8)) * 2u), 0, 0);
                }
                N8_LDS_SYNC();
            };
            do_pass(ICn<0>{});
            do_pass(ICn<1>{});
            do_pass(ICn<2>{});
            do_pass(ICn<3>{});
        }
        if (!more) break;
        tile = next;
        first = false;
    }
#endif
}

// Sum of the K slices' partial tiles (fixed order s = 0, 1, ..: deterministic) + the bf16 epilogue of conv_igemm_p8n_kernel
// (v * scale + bias, + 0, activation), 8 channels per thread.
__global__ __launch_bounds__(256) void p8n_split_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N, const float* __restrict__ scale,
                                                               const float* __restrict__ bias, bf16_t* __restrict__ y, long long y_cs, int act) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x, per_row = N / 8;
    if (i >= (long long)M * per_row) return;
    const long long m = i / per_row;
    const int n = (int)(i - m * per_row) * 8;
    const float* src = ws + m * N + n;
    f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    for (int s = 1; s < splits; ++s) {
        const float* q = src + (long long)s * M * N;
        const f32x4 c = *reinterpret_cast<const f32x4*>(q), d = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] += c[e]; b[e] += d[e]; }
    }
    f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (scale) { s0 = *reinterpret_cast<const f32x4*>(scale + n); s1 = *reinterpret_cast<const f32x4*>(scale + n + 4); }
    if (bias) { b0 = *reinterpret_cast<const f32x4*>(bias + n); b1 = *reinterpret_cast<const f32x4*>(bias + n + 4); }
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = a[e] * s0[e];
        v[e] += b0[e];
        v[4 + e] = b[e] * s1[e];
        v[4 + e] += b1[e];
    }
    us8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = v[e] + 0.f;
        if (act == NPS_ACT_RELU) t = t > 0.f ? t : 0.f;
        else if (act == NPS_ACT_LEAKY) t = t > 0.f ? t : 0.01f * t;
        o[e] = f32_to_bf16(t);
    }
    *reinterpret_cast<us8*>(y + m * y_cs + n) = o;
}

}  // namespace nps

extern int nps_p8_num_cus();

// x [B,H,W,Cin] bf16 (pixel stride x_cstride), w [Cout][KH][KW][Cin] bf16 (plain K-contiguous rows), Cin % 64 == 0, Cout % 128 == 0;
// y bf16 = act(conv * scale + bias), act in {none, ReLU, LeakyReLU}; variant: 0, + 32 = channel-major K order, + (n << 8) = at most n
// persistent workgroups (tuning aid).
static int p8n_launch(const void* x, const void* w, const float* scale, const float* bias, void* y, int B, int H, int W,
                      int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride, int64_t y_cstride, int act,
                      int variant, int splits, void* workspace, int64_t workspace_bytes, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(x && w && y, "conv2d_p8n: null pointer");
    NPS_CHECK_ARG(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && KH * KW <= 32, "conv2d_p8n: bad dims");
    NPS_CHECK_ARG(Cin > 0 && Cin % 64 == 0 && Cout > 0 && Cout % 128 == 0, "conv2d_p8n: needs Cin %% 64 == 0 and Cout %% 128 == 0");
    NPS_CHECK_ARG(act == NPS_ACT_NONE || act == NPS_ACT_RELU || act == NPS_ACT_LEAKY, "conv2d_p8n: act must be none / ReLU / LeakyReLU (no residual forms)");
    const int kmajor = (variant >> 5) & 1;
    const int grid_cap = variant >> 8;
    NPS_CHECK_ARG((variant & 0xdf) == 0, "conv2d_p8n: variant must be 0 (+ 32: channel-major K order, + (n << 8): grid cap)");
    NPS_CHECK_ARG(x_cstride >= Cin && x_cstride % 8 == 0 && y_cstride >= Cout && y_cstride % 8 == 0 && ((uintptr_t)x % 16 == 0) &&
                      ((uintptr_t)w % 16 == 0) && ((uintptr_t)y % 16 == 0) && (!scale || (uintptr_t)scale % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0),
                  "conv2d_p8n: strides / alignment");
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.res = nullptr; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.OH = (H + 2 * pad - KH) / stride + 1;
    p.OW = (W + 2 * pad - KW) / stride + 1;
    NPS_CHECK_ARG(p.OH > 0 && p.OW > 0, "conv2d_p8n: empty output");
    p.x_cs = x_cstride; p.y_cs = y_cstride;
    p.rows_per_b = p.OH * p.OW;
    p.M = B * p.rows_per_b; p.N = Cout; p.K = KH * KW * Cin;
    p.act = act; p.out_dt = NPS_DT_BF16; p.force = kmajor;
    NPS_CHECK_ARG((long long)B * H * W * x_cstride * 2 + ((long long)pad * W + pad) * x_cstride * 2 < (1ll << 31), "conv2d_p8n: input larger than 2 GB");
    NPS_CHECK_ARG((long long)p.N * p.K * 2 < (1ll << 31), "conv2d_p8n: weights larger than 2 GB");
    NPS_CHECK_ARG(((long long)p.M + 256) * y_cstride * 2 < (1ll << 31), "conv2d_p8n: output larger than 2 GB");
    NPS_CHECK_ARG(p.M < (1 << 23) && (long long)B * H * W < (1 << 24) && x_cstride < (1 << 24) && p.K < (1 << 24) && Cout < (1 << 24),
                  "conv2d_p8n: pixel count / strides beyond the 24-bit index math of this kernel");
    p.tiles_m = (p.M + N8_BM - 1) / N8_BM;
    p.tiles_n = p.N / N8_BN;
    const hipStream_t st = (hipStream_t)stream;
    if (splits > 1) {
        const long long need = (long long)splits * p.M * p.N * 4;
        NPS_CHECK_ARG(kmajor && splits <= Cin / 64 && splits <= 16, "conv2d_p8n_splitk: channel-major K order, 2 <= splits <= min(Cin / 64, 16)");
        NPS_CHECK_ARG(workspace && ((uintptr_t)workspace % 16 == 0) && workspace_bytes >= need && need < (1ll << 31),
                      "conv2d_p8n_splitk: workspace of splits * M * N * 4 bytes (< 2 GB), 16-byte aligned");
        p.ksplit = splits; p.sk_ws = workspace; p.sk_ws_bytes = (int)need;
        const int units = p.tiles_m * p.tiles_n * splits;
        int nwg = nps_p8_num_cus();
        if (grid_cap > 0 && grid_cap < nwg) nwg = grid_cap;
        if (units < nwg) nwg = units;
        hipLaunchKernelGGL((conv_igemm_p8n_kernel<NPS_ACT_NONE, true>), dim3(nwg), dim3(512), 0, st, p);
        const long long items = (long long)p.M * (p.N / 8);
        hipLaunchKernelGGL(p8n_split_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, (const float*)workspace, splits, p.M, p.N,
                           scale, bias, (bf16_t*)y, (long long)y_cstride, act);
        NPS_LAUNCH_RET();
    }
    const int ntiles = p.tiles_m * p.tiles_n;
    int nwg = nps_p8_num_cus();
    if (grid_cap > 0 && grid_cap < nwg) nwg = grid_cap;
    if (ntiles < nwg) nwg = ntiles;
    const dim3 grid(nwg);
    if (act == NPS_ACT_RELU) hipLaunchKernelGGL((conv_igemm_p8n_kernel<NPS_ACT_RELU>), grid, dim3(512), 0, st, p);
    else if (act == NPS_ACT_LEAKY) hipLaunchKernelGGL((conv_igemm_p8n_kernel<NPS_ACT_LEAKY>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((conv_igemm_p8n_kernel<NPS_ACT_NONE>), grid, dim3(512), 0, st, p);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_conv2d_nhwc_p8n(const void* x, const void* w, const float* scale, const float* bias, void* y, int B, int H, int W,
                                       int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride, int64_t y_cstride, int act,
                                       int variant, void* stream) {
    return p8n_launch(x, w, scale, bias, y, B, H, W, Cin, Cout, KH, KW, stride, pad, x_cstride, y_cstride, act, variant, 1, nullptr, 0, stream);
}

// Split-K form (round 6): `splits` K slices per 256 x 128 output tile (channel-major K order: variant must carry + 32), f32 partial tiles in
// `workspace` (>= splits * M * Cout * 4 bytes, M = B * OH * OW), summed in fixed order by a second launch that applies the epilogue.
extern "C" int nopesac_conv2d_nhwc_p8n_splitk(const void* x, const void* w, const float* scale, const float* bias, void* y, int B, int H, int W,
                                              int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride, int64_t y_cstride,
                                              int act, int variant, int splits, void* workspace, int64_t workspace_bytes, void* stream) {
    NPS_CHECK_ARG(splits >= 2, "conv2d_p8n_splitk: splits >= 2");
    return p8n_launch(x, w, scale, bias, y, B, H, W, Cin, Cout, KH, KW, stride, pad, x_cstride, y_cstride, act, variant, splits, workspace,
                      workspace_bytes, stream);
}
