// COCO run-length encoding of the kept plane masks, straight from the post-selection winner map.
//
// The reference packs every kept plane into an `instances` entry whose "segmentation" is
// pycocotools.mask.encode(np.asfortranarray(mask)) and whose "bbox" is mask.toBbox(rle)
// (meta_arch/siamese_planeTR.py:685-720, 741-766).  pycocotools is a third-party dependency that is not in the
// reference tree; its published algorithm (cocoapi maskApi.c rleEncode / rleToString / rleToBbox) is:
//   * scan the mask in COLUMN-major order, emit alternating run lengths starting with a run of zeros;
//   * string form: each count (from the 4th on, minus the count two places earlier) as 5-bit groups, LSB first,
//     bit 5 = "more", + 48 to land in printable ASCII;
//   * bbox = tight box of the ones, [x, y, w, h]; zeros when there are fewer than two runs.
//
// Device side (this file): the n masks of a view are never materialised.  One pass turns the row-major winner map
// into a column-major map of plane ordinals (0xFF = no kept plane); then one workgroup per (view, plane) streams that
// 300 KB map (L2 resident) 16 bytes per lane and compacts the positions where "pixel belongs to plane p" flips.
// String side: nopesac_rle_compress_device (a workgroup per mask, two passes: sizes + boxes, then the characters) turns the flip
// positions into the COCO strings + boxes on the device; nopesac_rle_compress_host is the plain-C single-mask form.
#include "common.h"

namespace nps {

__global__ __launch_bounds__(256) void rle_labels_kernel(const uint8_t* __restrict__ winner, const int* __restrict__ kept_idx,
                                                         const int* __restrict__ n_kept, const int* __restrict__ flags,
                                                         uint8_t* __restrict__ labels, int H, int W, int nq) {
    __shared__ uint8_t lut[128];
    __shared__ uint8_t tile[32][33];
    const int v = blockIdx.z, tx = threadIdx.x, ty = threadIdx.y, tid = ty * 32 + tx;
    if (tid < 128) lut[tid] = 0xFF;
    __syncthreads();
    const int n = n_kept[v];
    if (tid < n) {
        const int q = kept_idx[(long long)v * nq + tid];
        if (q >= 0 && q < 128) lut[q] = (uint8_t)tid;
    }
    __syncthreads();
    const bool fallback = (flags[v] & 2) != 0;      // :743 fallback mask = arg-max only, no probability test
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const uint8_t* wv = winner + (long long)v * H * W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = y0 + ty + 8 * i, x = x0 + tx;
        uint8_t l = 0xFF;
        if (y < H && x < W) {
            const uint8_t w8 = wv[(long long)y * W + x];
            if (fallback || (w8 & 0x80)) l = lut[w8 & 0x7F];
        }
        tile[ty + 8 * i][tx] = l;
    }
    __syncthreads();
    uint8_t* lv = labels + (long long)v * H * W;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x0 + ty + 8 * i, y = y0 + tx;
        if (x < W && y < H) lv[(long long)x * H + y] = tile[tx][ty + 8 * i];
    }
}

// grid (nq, V).  counts[v*nq+p] = number of flips of plane p; when positions != nullptr the flip positions are
// written (ascending) at positions[offsets[v*nq+p] ...].
// Wave w owns the contiguous quarter [w*Q, (w+1)*Q) of the map (Q a multiple of 1024 = 64 lanes x 16 pixels) and sweeps it 1024
// pixels at a time; a lane's 16 labels become a 16-bit "is plane p" mask with four SWAR byte compares.  Counting needs no
// communication inside the sweep (popcounts add up, one reduction at the end); writing needs the flips before a lane's, which is a
// wave-level scan per step plus the wave's base (the counting sweep's per-wave totals) - no workgroup barrier inside either sweep
// (the first form block-scanned every 4096 pixels: 75 double barriers per mask, 178 us per pass for 2048 masks).
__device__ __forceinline__ uint32_t rle_eq16(const uint4& q, uint32_t pb) {
    // bit j = (byte j of q == p); pb = p * 0x01010101.  Zero-byte test of x = w ^ pb, exact per byte (no borrow across bytes):
    // ((x & 0x7f7f7f7f) + 0x7f7f7f7f) | x has bit 7 of a byte clear iff that byte is zero.
    const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t x = wd[i] ^ pb;
        const uint32_t t = ~((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) | 0x7f7f7f7fu);      // 0x80 in every zero byte
        m |= (((t >> 7) * 0x00204081u) >> 21 & 0xFu) << (4 * i);                          // gather bits 0, 8, 16, 24 -> 4 bits
    }
    return m;
}

// 16 waves per (plane, view): a wave walks its share of the image 1024 pixels at a time, and with few planes (one pair per call:
// ~10 workgroups on the chip) the length of that walk is the kernel's time - 75 dependent steps with 4 waves (40 + 68 us for the two
// passes), 19 with 16.
constexpr int RLE_TW = 16;
__global__ __launch_bounds__(RLE_TW * 64) void rle_transitions_kernel(const uint8_t* __restrict__ labels, const int* __restrict__ n_kept,
                                                                      const long long* __restrict__ offsets, int* __restrict__ counts,
                                                                      uint32_t* __restrict__ positions, int N, int nq) {
    __shared__ int wave_tot[RLE_TW];
    const int p = blockIdx.x, v = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p >= n_kept[v]) {
        if (tid == 0) counts[v * nq + p] = 0;
        return;
    }
    const uint8_t* lv = labels + (long long)v * N;
    const bool vec_ok = (N % 16 == 0) && (((uintptr_t)lv & 15) == 0);
    const uint32_t pb = (uint32_t)p * 0x01010101u;
    const int Q = ((N + RLE_TW * 1024 - 1) / (RLE_TW * 1024)) * 1024;    // pixels per wave, multiple of 1024
    const int k_begin = wave * Q, k_end = min(N, k_begin + Q);
    // flips among the 16 pixels k .. k+15 (bit j: pixel k+j differs from pixel k+j-1 in "belongs to p"; pixel -1 counts as outside)
    auto flips = [&](int k) -> uint32_t {
        uint32_t m = 0;
        if (vec_ok) m = rle_eq16(*reinterpret_cast<const uint4*>(lv + k), pb);
        else
            for (int j = 0; j < 16 && k + j < N; ++j) m |= (uint32_t)(lv[k + j] == (uint8_t)p) << j;
        const uint32_t prev = k > 0 ? (uint32_t)(lv[k - 1] == (uint8_t)p) : 0u;
        uint32_t tr = (m ^ ((m << 1) | prev)) & 0xFFFFu;
        if (k + 16 > N) tr &= (1u << (N - k)) - 1u;
        return tr;
    };
    int mine = 0;
    for (int k = k_begin + lane * 16; k < k_end; k += 1024) mine += __popc(flips(k));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d, 64);
    if (lane == 0) wave_tot[wave] = mine;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < RLE_TW; ++w) {
        const int t = wave_tot[w];
        if (w < wave) wbase += t;
        total += t;
    }
    if (tid == 0) counts[v * nq + p] = total;
    if (!positions) return;
    uint32_t* out = positions + offsets[v * nq + p] + wbase;
    int base = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += 1024) {
        const int k = k0 + lane * 16;
        uint32_t tr = k < k_end ? flips(k) : 0u;
        const int c = __popc(tr);
        int incl = c;                                       // wave inclusive scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        int o = base + incl - c;
        while (tr) {
            const int j = __ffs(tr) - 1;
            tr &= tr - 1;
            out[o++] = (uint32_t)(k + j);
        }
        base += __shfl(incl, 63, 64);
    }
}

}  // namespace nps

extern "C" int nopesac_rle_labels(const uint8_t* winner, const int32_t* kept_idx, const int32_t* n_kept, const int32_t* flags,
                                  uint8_t* labels, int V, int H, int W, int nq, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(winner && kept_idx && n_kept && flags && labels, "rle_labels: null pointer");
    NPS_CHECK_ARG(V > 0 && H > 0 && W > 0 && nq > 0 && nq <= 128, "rle_labels: bad sizes (nq <= 128)");
    dim3 grid((W + 31) / 32, (H + 31) / 32, V);
    hipLaunchKernelGGL(rle_labels_kernel, grid, dim3(32, 8), 0, (hipStream_t)stream, winner, kept_idx, n_kept, flags, labels, H, W, nq);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_rle_transitions(const uint8_t* labels, const int32_t* n_kept, const int64_t* offsets, int32_t* counts,
                                       uint32_t* positions, int V, int N, int nq, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(labels && n_kept && counts, "rle_transitions: null pointer");
    NPS_CHECK_ARG((positions == nullptr) == (offsets == nullptr), "rle_transitions: offsets and positions go together");
    NPS_CHECK_ARG(V > 0 && N > 0 && nq > 0 && nq <= 128, "rle_transitions: bad sizes");
    hipLaunchKernelGGL(rle_transitions_kernel, dim3(nq, V), dim3(RLE_TW * 64), 0, (hipStream_t)stream, labels, n_kept,
                       (const long long*)offsets, counts, positions, N, nq);
    NPS_LAUNCH_RET();
}

// ---- dense masks of the kept planes of ALL views in one launch (the reference's `pred_plane_masks`, siamese_planeTR.py:685 / :743):
// masks[offsets[v] + p][pixel] = (arg-max query of the pixel == kept_idx[v][p]) && (above the mask threshold || fallback view).
// Replaces three torch launches per view (192 per 32-pair batch).
namespace nps {
__global__ __launch_bounds__(256) void decode_masks_kernel(const uint8_t* __restrict__ winner, const int* __restrict__ kept_idx,
                                                           const int* __restrict__ n_kept, const int* __restrict__ flags,
                                                           const long long* __restrict__ offsets, uint8_t* __restrict__ masks, int N, int nq) {
    __shared__ int kept[128];
    const int v = blockIdx.y, tid = threadIdx.x;
    const int n = min(n_kept[v], nq);
    if (tid < n) kept[tid] = kept_idx[(long long)v * nq + tid];
    __syncthreads();
    const bool fallback = (flags[v] & 2) != 0;
    const long long base = (long long)blockIdx.x * 256 * 16 + tid * 16;          // 16 pixels per thread: 16-byte loads and stores
    if (base >= N) return;
    const uint8_t* wv = winner + (long long)v * N + base;
    uint8_t w[16];
    const bool full = base + 16 <= N && ((((uintptr_t)wv) & 15) == 0);
    if (full) *reinterpret_cast<uint4*>(w) = *reinterpret_cast<const uint4*>(wv);
    else
        for (int j = 0; j < 16; ++j) w[j] = base + j < N ? wv[j] : 0;
    uint8_t* out = masks + offsets[v] * N + base;
    for (int p = 0; p < n; ++p) {
        const int q = kept[p];
        uint8_t m[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) m[j] = ((w[j] & 0x7F) == q && (fallback || (w[j] & 0x80))) ? 1 : 0;
        uint8_t* o = out + (long long)p * N;
        if (full && ((((uintptr_t)o) & 15) == 0)) *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(m);
        else
            for (int j = 0; j < 16 && base + j < N; ++j) o[j] = m[j];
    }
}
}  // namespace nps

extern "C" int nopesac_decode_masks(const uint8_t* winner, const int32_t* kept_idx, const int32_t* n_kept, const int32_t* flags,
                                    const int64_t* offsets, uint8_t* masks, int V, int H, int W, int nq, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(winner && kept_idx && n_kept && flags && offsets && masks, "decode_masks: null pointer");
    NPS_CHECK_ARG(V > 0 && H > 0 && W > 0 && nq > 0 && nq <= 128, "decode_masks: bad sizes (nq <= 128)");
    const int N = H * W;
    hipLaunchKernelGGL(decode_masks_kernel, dim3((N + 4095) / 4096, V), dim3(256), 0, (hipStream_t)stream, winner, kept_idx, n_kept, flags,
                       (const long long*)offsets, masks, N, nq);
    NPS_LAUNCH_RET();
}

// ---- device-side string compression ------------------------------------------------------------------------------------------
// A 32-pair step with K kept planes per view has 2 B K masks and millions of flip positions: compressing them on one host core
// (and copying 4 bytes per flip over PCIe first) was 2/3 of package()'s time.  The string format is local - character group i
// depends on runs i and i-2 only - so a workgroup per mask encodes 256 runs at a time (1..6 characters each), block-scans
// the lengths and writes the characters; pass 1 (out == nullptr) only sizes the strings and reduces the box.
namespace nps {

// characters of one count (cocoapi rleToString): 5 data bits per character, bit 5 = more, + 48
__device__ __forceinline__ int rle_chars(long long x, char (&c)[8]) {
    int n = 0;
    bool more = true;
    while (more) {
        char ch = (char)(x & 0x1f);
        x >>= 5;
        more = (ch & 0x10) ? x != -1 : x != 0;
        if (more) ch |= 0x20;
        c[n++] = (char)(ch + 48);
    }
    return n;
}

// grid = masks (mask i owns positions[offsets[i] .. + counts[i])).  out == nullptr: lens[i] = string length, bbox4[4 i ..] = box.
// out != nullptr: the string is written at out + out_off[i].
__global__ __launch_bounds__(256) void rle_compress_kernel(const uint32_t* __restrict__ positions, const long long* __restrict__ offsets,
                                                           const int* __restrict__ counts, int H, int W, int* __restrict__ lens,
                                                           double* __restrict__ bbox4, char* __restrict__ out,
                                                           const long long* __restrict__ out_off, long long cap) {
    __shared__ int wave_tot[4];
    __shared__ long long red[4][4];
    __shared__ int red_full[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (out && cap >= 0 && out_off[i] + lens[i] > cap) return;   // capped pass 2: a string that does not fit is left out (the host sees it in out_off / lens)
    const int n_pos = counts[i];
    const uint32_t* pos = positions + offsets[i];
    const long long N = (long long)H * W;
    const int m = n_pos + 1;                                  // runs: [pos0, pos1-pos0, ..., N-pos_last]
    auto edge = [&](int j) -> long long { return j < 0 ? 0 : (j >= n_pos ? N : (long long)pos[j]); };     // end of run j
    char* o = out ? out + out_off[i] : nullptr;
    int base = 0;
    const int me = (m / 2) * 2;
    long long xs = W, ys = H, xe = -1, ye = -1;
    int full = 0;
    for (int j0 = 0; j0 < m; j0 += 256) {
        const int j = j0 + tid;
        char c[8];
        int n = 0;
        if (j < m) {
            long long x = edge(j) - edge(j - 1);
            if (j > 2) x -= edge(j - 2) - edge(j - 3);
            n = rle_chars(x, c);
            if (!out && j < me) {                             // tight box of the ones (rleToBbox): runs 1, 3, 5, ... are ones
                const long long t = edge(j) - (j % 2), y = t % H, xx = (t - y) / H;
                if (j % 2 == 1) {
                    const long long tp = edge(j - 1), xp = (tp - tp % H) / H;
                    if (xp < xx) full = 1;
                }
                xs = xx < xs ? xx : xs; xe = xx > xe ? xx : xe;
                ys = y < ys ? y : ys; ye = y > ye ? y : ye;
            }
        }
        int incl = n;                                         // block exclusive scan of the lengths
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int wbase = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int t = wave_tot[w];
            if (w < wave) wbase += t;
            total += t;
        }
        if (o) {
            char* q = o + base + wbase + incl - n;
            for (int e = 0; e < n; ++e) q[e] = c[e];
        }
        base += total;
        __syncthreads();
    }
    if (out) return;
    // box reduction across the block
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const long long a = __shfl_xor(xs, d, 64), b = __shfl_xor(ys, d, 64), cx = __shfl_xor(xe, d, 64), cy = __shfl_xor(ye, d, 64);
        xs = a < xs ? a : xs; ys = b < ys ? b : ys; xe = cx > xe ? cx : xe; ye = cy > ye ? cy : ye;
        full |= __shfl_xor(full, d, 64);
    }
    if (lane == 0) { red[wave][0] = xs; red[wave][1] = ys; red[wave][2] = xe; red[wave][3] = ye; red_full[wave] = full; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) {
            xs = red[w][0] < xs ? red[w][0] : xs; ys = red[w][1] < ys ? red[w][1] : ys;
            xe = red[w][2] > xe ? red[w][2] : xe; ye = red[w][3] > ye ? red[w][3] : ye;
            full |= red_full[w];
        }
        lens[i] = base;
        double* bb = bbox4 + 4 * (long long)i;
        if (me == 0) { bb[0] = bb[1] = bb[2] = bb[3] = 0.0; }
        else {
            if (full) { ys = 0; ye = H - 1; }
            bb[0] = (double)xs; bb[1] = (double)ys; bb[2] = (double)(xe - xs + 1); bb[3] = (double)(ye - ys + 1);
        }
    }
}

}  // namespace nps

// Device-side counterpart of nopesac_rle_compress_batch_host.  positions / offsets / counts: device memory (mask i owns
// positions[offsets[i] .. offsets[i] + counts[i])).  Pass 1 (out == NULL): lens[i] = string length of mask i, bbox4[4 i ..] = its box.
// Pass 2 (out != NULL): the strings are written at out + out_off[i] (out_off = exclusive prefix sums of lens, device).
extern "C" int nopesac_rle_compress_device(const uint32_t* positions, const int64_t* offsets, const int32_t* counts, int n_masks, int H, int W,
                                           int32_t* lens, double* bbox4, char* out, const int64_t* out_off, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(offsets && counts && n_masks > 0 && H > 0 && W > 0, "rle_compress_device: bad args");
    NPS_CHECK_ARG(out ? (out_off != nullptr) : (lens && bbox4), "rle_compress_device: pass 1 needs lens + bbox4, pass 2 needs out + out_off");
    hipLaunchKernelGGL(rle_compress_kernel, dim3(n_masks), dim3(256), 0, (hipStream_t)stream, positions, (const long long*)offsets, counts, H, W,
                       lens, bbox4, out, (const long long*)out_off, -1LL);
    NPS_LAUNCH_RET();
}

// Pass 2 into a buffer of FIXED capacity (no host round trip for the total length): string i is written at out + out_off[i] only
// if out_off[i] + lens[i] <= cap; lens = pass 1's output (read here).  The caller compares out_off / lens with cap afterwards.
extern "C" int nopesac_rle_compress_device_capped(const uint32_t* positions, const int64_t* offsets, const int32_t* counts, int n_masks, int H,
                                                  int W, const int32_t* lens, char* out, const int64_t* out_off, int64_t cap, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(offsets && counts && lens && out && out_off && n_masks > 0 && H > 0 && W > 0 && cap >= 0, "rle_compress_device_capped: bad args");
    hipLaunchKernelGGL(rle_compress_kernel, dim3(n_masks), dim3(256), 0, (hipStream_t)stream, positions, (const long long*)offsets, counts, H, W,
                       const_cast<int32_t*>(lens), (double*)nullptr, out, (const long long*)out_off, (long long)cap);
    NPS_LAUNCH_RET();
}

// Host-only: flip positions of ONE mask (ascending, column-major pixel indices) -> COCO compressed counts string
// (not NUL-terminated) + bbox [x, y, w, h].  Returns the string length, or a negative error code.
extern "C" int nopesac_rle_compress_host(const uint32_t* positions, int n_pos, int H, int W, char* out, int cap, double* bbox4) {
    using namespace nps;
    NPS_CHECK_ARG((positions || n_pos == 0) && n_pos >= 0 && H > 0 && W > 0 && out && bbox4, "rle_compress: bad args");
    const long long N = (long long)H * W;
    const int m = n_pos + 1;                                  // runs: [pos0, pos1-pos0, ..., N-pos_last]
    auto run = [&](int i) -> long long {
        const long long lo = i == 0 ? 0 : positions[i - 1];
        const long long hi = i == n_pos ? N : positions[i];
        return hi - lo;
    };
    int len = 0;
    for (int i = 0; i < m; ++i) {
        long long x = run(i);
        NPS_CHECK_ARG(x >= 0, "rle_compress: positions not ascending");
        if (i > 2) x -= run(i - 2);
        bool more = true;
        while (more) {
            char c = (char)(x & 0x1f);
            x >>= 5;
            more = (c & 0x10) ? x != -1 : x != 0;
            if (more) c |= 0x20;
            NPS_CHECK_ARG(len < cap, "rle_compress: output buffer too small");
            out[len++] = (char)(c + 48);
        }
    }
    // tight box of the ones (rleToBbox): runs 1, 3, 5, ... are ones
    const int me = (m / 2) * 2;
    if (me == 0) {
        bbox4[0] = bbox4[1] = bbox4[2] = bbox4[3] = 0.0;
        return len;
    }
    long long xs = W, ys = H, xe = 0, ye = 0, xp = 0, cc = 0;
    for (int j = 0; j < me; ++j) {
        cc += run(j);
        const long long t = cc - (j % 2), y = t % H, x = (t - y) / H;
        if (j % 2 == 0) xp = x;
        else if (xp < x) { ys = 0; ye = H - 1; }
        xs = x < xs ? x : xs; xe = x > xe ? x : xe;
        ys = y < ys ? y : ys; ye = y > ye ? y : ye;
    }
    bbox4[0] = (double)xs; bbox4[1] = (double)ys; bbox4[2] = (double)(xe - xs + 1); bbox4[3] = (double)(ye - ys + 1);
    return len;
}

// Host-only, batch form: n_masks masks at once (one ctypes call per batch instead of one per plane - the per-plane calls were
// 2/3 of package()'s 32 ms per 32-pair step).  Mask i owns positions[offsets[i] .. offsets[i] + counts[i]); its string is written
// at out + out_off[i] (out_off[n_masks] = total bytes used), its box at bbox4 + 4 i.  Returns the total length or < 0.
extern "C" long long nopesac_rle_compress_batch_host(const uint32_t* positions, const long long* offsets, const int* counts, int n_masks,
                                                     int H, int W, char* out, long long cap, long long* out_off, double* bbox4) {
    using namespace nps;
    if (!((positions || n_masks == 0) && offsets && counts && n_masks >= 0 && out && out_off && bbox4)) {
        set_error("rle_compress_batch: bad args");
        return NPS_E_ARG;
    }
    long long used = 0;
    for (int i = 0; i < n_masks; ++i) {
        out_off[i] = used;
        const long long room = cap - used;
        const int n = nopesac_rle_compress_host(counts[i] ? positions + offsets[i] : nullptr, counts[i], H, W, out + used,
                                                (int)(room > 0x7fffffff ? 0x7fffffff : room), bbox4 + 4 * (long long)i);
        if (n < 0) return n;
        used += n;
    }
    out_off[n_masks] = used;
    return used;
}
