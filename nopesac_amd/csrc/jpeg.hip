// Baseline-JPEG decode on the device: the reference's data mapper reads its frames with detectron2's utils.read_image (PIL /
// libjpeg-turbo, NopeSAC_Net/data/planercnn_transforms.py:210-227, :306-314) on host threads - 3200 pairs/s with 32 of them, less than
// ONE GPU's model throughput.  Three kernels, results bit for bit those of libjpeg-turbo's default decompression (JDCT_ISLOW, fancy
// upsampling, RGB), host side in nopesac_amd/jpeg.py (marker walk, restart split, stuffing removal, table layout - no entropy decoding):
//
//   jpeg_sync_kernel<0|1|2>, jpeg_sync_scan_kernel, jpeg_dc_kernel
//                        restart-free files: self-synchronising parallel Huffman decode, one lane per 2048-bit subsequence (further down)
//   jpeg_huffman_kernel  one WAVE per restart interval (per image when the file has none and the lanes above did not settle).  Entropy decoding is a serial chain per
//                        interval (every code's position depends on all codes before it), so the wave runs it on the SCALAR unit:
//                        all state is wave-uniform, the bit stream and the tables are read with scalar loads from the constant
//                        address space, lane 0 stores the coefficients (zigzag order, DC prediction resolved).  Throughput comes from
//                        the number of intervals in flight - restart intervals when the encoder wrote them, images otherwise.
//   jpeg_idct_kernel     one thread per 8x8 block: de-zigzag + dequantise + jidctint.c's two-pass 13-bit fixed-point inverse DCT in
//                        registers, samples into the component's plane.
//   jpeg_color_kernel    one thread per output pixel: jdsample.c's h2v1 / h2v2 "fancy" (triangle) chroma upsampling evaluated at the
//                        pixel (incl. its edge rules and jdmainct.c's replicated context rows), jdcolor.c's 16-bit fixed-point
//                        YCbCr -> RGB, interleaved RGB or BGR bytes out (the layout nopesac_resize_bilinear_u8 reads).
#include "common.h"

namespace nps {

constexpr int JP_LOOK = 9;
constexpr int JP_HUFF_BYTES = NOPESAC_JPEG_HUFF_BYTES, JP_TABLES_BYTES = NOPESAC_JPEG_TABLES_BYTES;
constexpr int JP_I32 = NOPESAC_JPEG_IMG_I32, JP_I64 = NOPESAC_JPEG_IMG_I64;

typedef const __attribute__((address_space(4))) uint32_t* c_u32;      // constant address space: uniform loads become s_load
typedef const __attribute__((address_space(4))) int32_t* c_i32;
typedef const __attribute__((address_space(4))) int64_t* c_i64;
typedef const __attribute__((address_space(4))) uint8_t* c_u8;
template <typename P>
__device__ __forceinline__ P as_const(const void* p) { return (P)(uintptr_t)p; }

// Bit reader: the interval's words sit in a 64-word VGPR window (lane l = word l of the current 256-byte chunk, the next chunk already
// requested into a second register); a word reaches the scalar side with v_readlane - no memory access on the decode chain (the first
// version read the stream with scalar loads: every word a scalar-cache miss, ~450 cycles per symbol, 50 ms for a 968 x 1296 frame).
struct JpBits {
    uint64_t acc;      // the next bits of the stream, most significant first
    int nb;            // valid bits in acc
    uint32_t cur, nxt; // VGPR windows
    int wi;            // next word of `cur`
    long long pos;     // word index (in the batch's word array) of cur's lane 0
};
__device__ __forceinline__ uint32_t jp_load_window(const uint32_t* __restrict__ words, long long n_words, long long at) {
    const long long i = at + (int)threadIdx.x;
    return i < n_words ? words[i] : 0u;
}
__device__ __forceinline__ void jp_fill(JpBits& b, const uint32_t* __restrict__ words, long long n_words) {
    if (b.nb <= 32) {
        const uint32_t w = __builtin_amdgcn_readlane(b.cur, b.wi);
        b.acc |= (uint64_t)w << (32 - b.nb);
        b.nb += 32;
        if (++b.wi == 64) {
            b.wi = 0;
            b.cur = b.nxt;
            b.pos += 64;
            b.nxt = jp_load_window(words, n_words, b.pos + 64);
        }
    }
}
__device__ __forceinline__ void jp_skip(JpBits& b, int n) {
    b.acc <<= n;
    b.nb -= n;
}
// the 9-bit look-ahead table of one Huffman table as four VGPRs (lane l of register j = dword 64 j + l of the uint16[512] array)
struct JpLook {
    uint32_t r[4];
};
// one Huffman symbol (jdhuff.c HUFF_DECODE: 9-bit look-ahead table, then one bit at a time against maxcode[]); >= 33 valid bits on
// entry: a code (<= 16) and its value bits (<= 15)
__device__ __forceinline__ int jp_symbol(JpBits& b, const JpLook& lk, c_u8 tab) {
    const unsigned li = (unsigned)(b.acc >> (64 - JP_LOOK));
    const int lane = (li >> 1) & 63, j = li >> 7;
    const uint32_t w0 = __builtin_amdgcn_readlane(lk.r[0], lane), w1 = __builtin_amdgcn_readlane(lk.r[1], lane);
    const uint32_t w2 = __builtin_amdgcn_readlane(lk.r[2], lane), w3 = __builtin_amdgcn_readlane(lk.r[3], lane);
    const uint32_t ew = j == 0 ? w0 : (j == 1 ? w1 : (j == 2 ? w2 : w3));
    const unsigned e = (li & 1) ? ew >> 16 : ew & 0xFFFFu;
    if (e >> 8) {
        jp_skip(b, e >> 8);
        return e & 255;
    }
    // longer codes (rare): maxcode / valoffset / huffval with scalar loads (whole dwords: there are no 8- / 16-bit ones on gfx950)
    c_i32 maxcode = (c_i32)(tab + 1024), valoff = (c_i32)(tab + 1096);
    int l = JP_LOOK + 1;
    int code = (int)(b.acc >> (64 - l));
    while (code > maxcode[l]) {
        ++l;
        code = (int)(b.acc >> (64 - l));
    }
    if (l > 16) {                                                   // not a code of this table (corrupt data): libjpeg warns and returns 0
        jp_skip(b, 16);
        return 0;
    }
    jp_skip(b, l);
    const unsigned vi = (unsigned)(code + valoff[l]) & 255u;
    return (((c_u32)(tab + 1168))[vi >> 2] >> (8 * (vi & 3))) & 255;
}
// the s value bits after a symbol, sign-extended (T.81 F.2.2.1 EXTEND)
__device__ __forceinline__ int jp_value(JpBits& b, int s) {
    if (s == 0) return 0;
    const int v = (int)(b.acc >> (64 - s));
    jp_skip(b, s);
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

__global__ __launch_bounds__(64) void jpeg_huffman_kernel(const int32_t* __restrict__ img32, const int64_t* __restrict__ img64,
                                                          const uint8_t* __restrict__ tables, const int32_t* __restrict__ seg32,
                                                          const int64_t* __restrict__ seg64, const uint32_t* __restrict__ words,
                                                          long long n_words, int16_t* __restrict__ coef, const int32_t* __restrict__ par_done) {
    const int sg = blockIdx.x;
    c_i32 S = as_const<c_i32>(seg32 + (long long)sg * NOPESAC_JPEG_SEG_I32);
    const int im = S[0], first = S[1], count = S[2];
    if (par_done && par_done[im]) return;                           // the image went through the self-synchronising decoder below
    c_i32 I = as_const<c_i32>(img32 + (long long)im * JP_I32);
    c_i64 I8 = as_const<c_i64>(img64 + (long long)im * JP_I64);
    const uint8_t* Tg = tables + (long long)im * JP_TABLES_BYTES;
    c_u8 T = as_const<c_u8>(Tg);
    const int ncomp = I[2], mcux = I[5];
    JpLook look[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) look[t].r[j] = reinterpret_cast<const uint32_t*>(Tg + t * JP_HUFF_BYTES)[j * 64 + threadIdx.x];
    JpBits b;
    b.acc = 0; b.nb = 0; b.wi = 0;
    b.pos = as_const<c_i64>(seg64 + (long long)sg * NOPESAC_JPEG_SEG_I64)[0];
    b.cur = jp_load_window(words, n_words, b.pos);
    b.nxt = jp_load_window(words, n_words, b.pos + 64);
    int pred[3] = {0, 0, 0};
    const bool writer = threadIdx.x == 0;
    int my = first / mcux, mx = first % mcux;
    for (int m = 0; m < count; ++m) {
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            if (ci >= ncomp) break;
            const int ch = ci == 0 ? I[3] : 1, cv = ci == 0 ? I[4] : 1, bw = I[8 + ci];
            const int tdc = I[20 + ci], tac = I[23 + ci];
            c_u8 dc = T + tdc * JP_HUFF_BYTES, ac = T + tac * JP_HUFF_BYTES;
            JpLook ldc, lac;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ldc.r[j] = tdc ? look[1].r[j] : look[0].r[j];
                lac.r[j] = tac == 3 ? look[3].r[j] : look[2].r[j];
            }
            int16_t* cc = coef + I8[ci];
            for (int v = 0; v < cv; ++v)
                for (int h = 0; h < ch; ++h) {
                    int16_t* blk = cc + ((long long)(my * cv + v) * bw + (mx * ch + h)) * 64;
                    jp_fill(b, words, n_words);
                    const int s = jp_symbol(b, ldc, dc);
                    pred[ci] += jp_value(b, s & 15);
                    if (writer) blk[0] = (int16_t)pred[ci];
                    int k = 1;
                    while (k < 64) {
                        jp_fill(b, words, n_words);
                        const int rs = jp_symbol(b, lac, ac);
                        const int r = rs >> 4, sz = rs & 15;
                        if (sz) {
                            k += r;
                            const int val = jp_value(b, sz);
                            if (writer) blk[k & 63] = (int16_t)val;
                            ++k;
                        } else if (r == 15) {
                            k += 16;
                        } else {
                            break;
                        }
                    }
                }
        }
        if (++mx == mcux) { mx = 0; ++my; }
    }
}

// ---- Self-synchronising parallel Huffman decode (files without restart markers: one serial chain of ~250 k codes per 968 x 1296 frame,
// 60 ms on the scalar kernel above).  The stream is cut into subsequences of JP_SUB_BITS bits, one LANE each.  A decoder started at an
// arbitrary bit with a guessed state (block-in-MCU, coefficient index) produces garbage at first, but Huffman streams re-align: after a
// while it sits on real code boundaries with the real state (the block phase does a random walk while it is out of step, and once it
// coincides it stays).  Scheme (after Klein / Wiseman and Weissenberger / Schmidt):
//   init      lane t decodes its subsequence from (bit t S, block 0, DC next) and records the state at the first code boundary >= (t+1) S
//             (its EXIT) and the blocks it completed; lane 0 starts from the true state
//   iterate   lane t re-decodes from lane t-1's exit whenever that differs from the entry it used last time; a fixed point is the true
//             decode (lane 0 is true; true exits propagate at least one subsequence per pass, in practice all lanes agree after 2-4)
//   scan      exclusive scan of the completed blocks -> first block number of every lane
//   write     every lane decodes once more from its entry and stores the coefficients of its blocks (DC as DIFFERENCE)
//   dc        running sum of the DC differences per component in scan order
// An image whose lanes still changed in the last pass keeps par_done = 0 and is decoded by the scalar kernel instead.
constexpr int JP_SUB_WORDS = NOPESAC_JPEG_SUB_WORDS, JP_SUB_BITS = JP_SUB_WORDS * 32;
constexpr int JP_ITERS = NOPESAC_JPEG_SYNC_PASSES;

struct JpLaneBits {
    uint64_t acc;
    int nb;
    long long wi;                      // next word (index into the batch's word array)
};
__device__ __forceinline__ uint32_t jp_word(const uint32_t* __restrict__ words, long long n_words, long long i) { return i < n_words ? words[i] : 0u; }
__device__ __forceinline__ void jp_lane_seek(JpLaneBits& b, const uint32_t* __restrict__ words, long long n_words, long long word0, long long bit) {
    const long long w = word0 + (bit >> 5);
    const int sh = (int)(bit & 31);
    b.acc = (((uint64_t)jp_word(words, n_words, w) << 32) | jp_word(words, n_words, w + 1)) << sh;
    b.nb = 64 - sh;
    b.wi = w + 2;
}
__device__ __forceinline__ void jp_lane_fill(JpLaneBits& b, const uint32_t* __restrict__ words, long long n_words) {
    if (b.nb <= 32) {
        b.acc |= (uint64_t)jp_word(words, n_words, b.wi++) << (32 - b.nb);
        b.nb += 32;
    }
}
// tables of one image in LDS: look u16[4][512], maxcode i32[4][18], valoff i32[4][18], huffval u8[4][256]
struct JpLdsTables {
    unsigned short look[4][512];
    int maxcode[4][18], valoff[4][18];
    unsigned char huffval[4][256];
};
__device__ __forceinline__ int jp_lane_symbol(JpLaneBits& b, const JpLdsTables& T, int t) {
    const unsigned e = T.look[t][(unsigned)(b.acc >> (64 - JP_LOOK))];
    if (e >> 8) {
        b.acc <<= (e >> 8); b.nb -= (e >> 8);
        return e & 255;
    }
    int l = JP_LOOK + 1;
    int code = (int)(b.acc >> (64 - l));
    while (code > T.maxcode[t][l]) {
        ++l;
        code = (int)(b.acc >> (64 - l));
    }
    if (l > 16) {
        b.acc <<= 16; b.nb -= 16;
        return 0;
    }
    b.acc <<= l; b.nb -= l;
    return T.huffval[t][(code + T.valoff[t][l]) & 255];
}
__device__ __forceinline__ int jp_lane_value(JpLaneBits& b, int s) {
    if (s == 0) return 0;
    const int v = (int)(b.acc >> (64 - s));
    b.acc <<= s; b.nb -= s;
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}
__device__ __forceinline__ long long jp_pack(long long bit, int blk, int k) { return bit | ((long long)blk << 40) | ((long long)k << 44); }

// One lane's pass over its subsequence.  WRITE = false: returns the exit state and the number of completed blocks.  WRITE = true:
// stores the coefficients of blocks first_block .. (DC differences), nothing beyond the image's last block.
template <bool WRITE>
__device__ __forceinline__ long long jp_lane_decode(const JpLdsTables& T, const int32_t* __restrict__ I, const int64_t* __restrict__ I8,
                                                    const uint32_t* __restrict__ words, long long n_words, long long entry, long long end_bit,
                                                    int& n_blocks, long long first_block, int16_t* __restrict__ coef) {
    const long long word0 = I8[7];
    const int hs = I[3], vs = I[4], hv = hs * vs, bpm = I[2] == 1 ? 1 : hv + 2, mcux = I[5];
    const long long total_blocks = (long long)I[5] * I[6] * bpm;
    long long bit = entry & ((1ll << 40) - 1);
    int blk = (int)((entry >> 40) & 15), k = (int)((entry >> 44) & 127);
    JpLaneBits b;
    jp_lane_seek(b, words, n_words, word0, bit);
    n_blocks = 0;
    long long g = first_block;                                      // number of the block being decoded (scan order)
    int16_t* dst = nullptr;
    auto block_address = [&]() {
        if (!WRITE) return;
        dst = nullptr;
        if (g >= total_blocks) return;
        const long long m = g / bpm;
        const int bb = (int)(g - m * bpm);
        const int ci = bb < hv ? 0 : bb - hv + 1, v = bb < hv ? bb / hs : 0, h = bb < hv ? bb - v * hs : 0;
        const int my = (int)(m / mcux), mx = (int)(m - (long long)my * mcux);
        const int cv = ci == 0 ? vs : 1, ch = ci == 0 ? hs : 1;
        dst = coef + I8[ci] + ((long long)(my * cv + v) * I[8 + ci] + (mx * ch + h)) * 64;
    };
    block_address();
    while (true) {
        const long long pos = (b.wi << 5) - b.nb - (word0 << 5);
        if (pos >= end_bit) return jp_pack(pos, blk, k);
        jp_lane_fill(b, words, n_words);
        const int ci = blk < hv ? 0 : blk - hv + 1;
        if (k == 0) {
            const int s = jp_lane_symbol(b, T, I[20 + ci]);
            const int val = jp_lane_value(b, s & 15);
            if (WRITE && dst) dst[0] = (int16_t)val;
            k = 1;
        } else {
            const int rs = jp_lane_symbol(b, T, I[23 + ci]);
            const int r = rs >> 4, sz = rs & 15;
            if (sz) {
                k += r;
                const int val = jp_lane_value(b, sz);
                if (WRITE && dst) dst[k & 63] = (int16_t)val;
                ++k;
            } else if (r == 15) {
                k += 16;
            } else {
                k = 64;
            }
        }
        if (k >= 64) {
            k = 0;
            if (++blk == bpm) blk = 0;
            ++n_blocks;
            ++g;
            block_address();
        }
    }
}

__device__ __forceinline__ void jp_load_tables(JpLdsTables& T, const uint8_t* __restrict__ Tg) {
    for (int t = 0; t < 4; ++t) {
        const uint8_t* src = Tg + t * JP_HUFF_BYTES;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) reinterpret_cast<uint32_t*>(T.look[t])[i] = reinterpret_cast<const uint32_t*>(src)[i];
        for (int i = threadIdx.x; i < 18; i += blockDim.x) {
            T.maxcode[t][i] = reinterpret_cast<const int32_t*>(src + 1024)[i];
            T.valoff[t][i] = reinterpret_cast<const int32_t*>(src + 1096)[i];
        }
        for (int i = threadIdx.x; i < 64; i += blockDim.x) reinterpret_cast<uint32_t*>(T.huffval[t])[i] = reinterpret_cast<const uint32_t*>(src + 1168)[i];
    }
    __syncthreads();
}

// MODE 0 init, 1 iterate (pass number in `pass`), 2 write.  One wave = 64 consecutive subsequences of ONE image (lane_img[first lane]).
template <int MODE>
__global__ __launch_bounds__(64) void jpeg_sync_kernel(const int32_t* __restrict__ img32, const int64_t* __restrict__ img64,
                                                       const uint8_t* __restrict__ tables, const int32_t* __restrict__ lane_img,
                                                       const uint32_t* __restrict__ words, long long n_words, long long* __restrict__ exit_state,
                                                       long long* __restrict__ entry_used, int32_t* __restrict__ n_blk,
                                                       const long long* __restrict__ first_block, int32_t* __restrict__ changed, int pass,
                                                       int n_images, const int32_t* __restrict__ par_done, int16_t* __restrict__ coef) {
    __shared__ JpLdsTables T;
    const long long gid = (long long)blockIdx.x * 64 + threadIdx.x;
    const int im = lane_img[(long long)blockIdx.x * 64];
    if (MODE == 2 && !par_done[im]) return;
    if (MODE == 1 && pass > 0 && changed[(pass - 1) * n_images + im] == 0) return;      // nothing moved in the previous pass: fixed point
    const int32_t* I = img32 + (long long)im * JP_I32;
    const int64_t* I8 = img64 + (long long)im * JP_I64;
    jp_load_tables(T, tables + (long long)im * JP_TABLES_BYTES);
    const int t = (int)(gid - I[28]);
    if (t >= I[29]) return;                                          // padding lane
    long long entry;
    if (t == 0) entry = 0;                                           // bit 0, block 0, DC next: the true start
    else if (MODE == 0) entry = jp_pack((long long)t * JP_SUB_BITS, 0, 0);
    else entry = exit_state[gid - 1];
    if (MODE == 1) {
        if (t == 0 || entry == entry_used[gid]) return;
        atomicAdd(&changed[pass * n_images + im], 1);
    }
    int nb = 0;
    if (MODE == 2) {
        jp_lane_decode<true>(T, I, I8, words, n_words, entry, (long long)(t + 1) * JP_SUB_BITS, nb, first_block[gid], coef);
        return;
    }
    const long long ex = jp_lane_decode<false>(T, I, I8, words, n_words, entry, (long long)(t + 1) * JP_SUB_BITS, nb, 0, nullptr);
    exit_state[gid] = ex;
    entry_used[gid] = entry;
    n_blk[gid] = nb;
}

// per image: converged? (no lane changed in the last pass) -> par_done; exclusive scan of the lanes' block counts
__global__ __launch_bounds__(256) void jpeg_sync_scan_kernel(const int32_t* __restrict__ img32, const int32_t* __restrict__ n_blk,
                                                             const int32_t* __restrict__ changed, int n_images, long long* __restrict__ first_block,
                                                             int32_t* __restrict__ par_done) {
    __shared__ long long wsum[4];
    __shared__ long long carry;
    const int im = blockIdx.x;
    const int32_t* I = img32 + (long long)im * JP_I32;
    const int n = I[29];
    if (n == 0) return;                                              // not a parallel image
    const bool ok = changed[(JP_ITERS - 1) * n_images + im] == 0;
    if (threadIdx.x == 0) { par_done[im] = ok ? 1 : 0; carry = 0; }
    __syncthreads();
    if (!ok) return;
    const long long base = I[28];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const long long v = i < n ? n_blk[base + i] : 0;
        long long incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        long long off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (i < n) first_block[base + i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = off + incl;
        __syncthreads();
    }
}

// running sum of the DC differences of component blockIdx.y of image blockIdx.x, in scan order
__global__ __launch_bounds__(256) void jpeg_dc_kernel(const int32_t* __restrict__ img32, const int64_t* __restrict__ img64,
                                                      const int32_t* __restrict__ par_done, int16_t* __restrict__ coef) {
    __shared__ int wsum[4];
    __shared__ int carry;
    const int im = blockIdx.x, ci = blockIdx.y;
    const int32_t* I = img32 + (long long)im * JP_I32;
    if (!par_done[im] || ci >= I[2]) return;
    const int hs = ci == 0 ? I[3] : 1, vs = ci == 0 ? I[4] : 1, hv = hs * vs, mcux = I[5], bw = I[8 + ci];
    const int n = I[5] * I[6] * hv;
    int16_t* c = coef + img64[(long long)im * JP_I64 + ci];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        int16_t* p = nullptr;
        if (j < n) {
            const int m = j / hv, sub = j - m * hv, v = sub / hs, h = sub - v * hs, my = m / mcux, mx = m - my * mcux;
            p = c + ((long long)(my * vs + v) * bw + (mx * hs + h)) * 64;
        }
        const int v0 = p ? (int)*p : 0;
        int incl = v0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (p) *p = (int16_t)(off + incl);
        __syncthreads();
        if (threadIdx.x == 255) carry = off + incl;
        __syncthreads();
    }
}

// ---- jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2
#define JP_IDCT_1D(i0, i1, i2, i3, i4, i5, i6, i7, SH, OUT)                                                  \
    {                                                                                                        \
        int z2 = i2, z3 = i6;                                                                                \
        int z1 = (z2 + z3) * 4433;                                                                           \
        int tmp2 = z1 - z3 * 15137, tmp3 = z1 + z2 * 6270;                                                   \
        int tmp0 = (i0 + i4) << 13, tmp1 = (i0 - i4) << 13;                                                  \
        const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;        \
        tmp0 = i7; tmp1 = i5; tmp2 = i3; tmp3 = i1;                                                          \
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;                                                \
        int z4 = tmp1 + tmp3;                                                                                \
        const int z5 = (z3 + z4) * 9633;                                                                     \
        tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;                                           \
        z1 *= -7373; z2 *= -20995; z3 = z3 * -16069 + z5; z4 = z4 * -3196 + z5;                              \
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;                                  \
        const int rnd = 1 << ((SH) - 1);                                                                     \
        OUT(0, (tmp10 + tmp3 + rnd) >> (SH)); OUT(7, (tmp10 - tmp3 + rnd) >> (SH));                          \
        OUT(1, (tmp11 + tmp2 + rnd) >> (SH)); OUT(6, (tmp11 - tmp2 + rnd) >> (SH));                          \
        OUT(2, (tmp12 + tmp1 + rnd) >> (SH)); OUT(5, (tmp12 - tmp1 + rnd) >> (SH));                          \
        OUT(3, (tmp13 + tmp0 + rnd) >> (SH)); OUT(4, (tmp13 - tmp0 + rnd) >> (SH));                          \
    }

__global__ __launch_bounds__(64) void jpeg_idct_kernel(const int32_t* __restrict__ img32, const int64_t* __restrict__ img64,
                                                       const uint8_t* __restrict__ tables, int n_images, int n_blocks,
                                                       const int16_t* __restrict__ coef, uint8_t* __restrict__ planes) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;            // block index in the batch-wide list
    if (g >= n_blocks) return;
    // image of the block: binary search over the images' first-block numbers (img32[.][26])
    int lo = 0, hi = n_images - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (img32[(long long)mid * JP_I32 + 26] <= g) lo = mid; else hi = mid - 1;
    }
    const int32_t* I = img32 + (long long)lo * JP_I32;
    const int64_t* I8 = img64 + (long long)lo * JP_I64;
    int bi = g - I[26], ci = 0;
    while (ci < I[2] - 1 && bi >= I[8 + ci] * I[11 + ci]) { bi -= I[8 + ci] * I[11 + ci]; ++ci; }
    const int bw = I[8 + ci], by = bi / bw, bx = bi % bw;
    const uint16_t* q = reinterpret_cast<const uint16_t*>(tables + (long long)lo * JP_TABLES_BYTES + 4 * JP_HUFF_BYTES + 128 * ci);
    const int16_t* c = coef + I8[ci] + (long long)bi * 64;
    int x[64];
    {
        short zz[64];
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(zz + 8 * i) = *reinterpret_cast<const uint4*>(c + 8 * i);
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            constexpr unsigned char ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                              41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                              30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
            x[ZZ[k]] = (int)zz[k] * (int)q[ZZ[k]];
        }
    }
    // pass 1: columns -> workspace (in place), descale by CONST_BITS - PASS1_BITS
#pragma unroll
    for (int col = 0; col < 8; ++col) {
#define JP_O1(r, val) x[(r) * 8 + col] = (val)
        JP_IDCT_1D(x[col], x[8 + col], x[16 + col], x[24 + col], x[32 + col], x[40 + col], x[48 + col], x[56 + col], 11, JP_O1)
#undef JP_O1
    }
    // pass 2: rows, descale by CONST_BITS + PASS1_BITS + 3, range limit (the table of jdmaster.c: wraps mod 1024, centred on 128)
    uint8_t* out = planes + I8[3 + ci] + ((long long)by * 8 * bw + bx) * 8;
#pragma unroll
    for (int row = 0; row < 8; ++row) {
        int y[8];
#define JP_O2(cidx, val) y[cidx] = (val)
        JP_IDCT_1D(x[row * 8], x[row * 8 + 1], x[row * 8 + 2], x[row * 8 + 3], x[row * 8 + 4], x[row * 8 + 5], x[row * 8 + 6], x[row * 8 + 7], 18, JP_O2)
#undef JP_O2
        unsigned lo4 = 0, hi4 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int t = y[i] & 1023;
            t = t >= 512 ? t - 1024 : t;
            t = min(max(t + 128, 0), 255);
            if (i < 4) lo4 |= (unsigned)t << (8 * i); else hi4 |= (unsigned)t << (8 * (i - 4));
        }
        *reinterpret_cast<uint2*>(out + (long long)row * bw * 8) = make_uint2(lo4, hi4);
    }
}

// chroma sample at output pixel (y, x): jdsample.c fullsize / h2v1_fancy / h2v2_fancy (replication when downsampled_width <= 2)
__device__ __forceinline__ int jp_chroma(const uint8_t* __restrict__ p, int pitch, int dw, int dh, int hs, int vs, int y, int x) {
    if (hs == 1) return p[(long long)y * pitch + x];
    const int i = x >> 1;
    if (vs == 1) {
        const uint8_t* r = p + (long long)y * pitch;
        if (dw <= 2) return r[i];
        if (x & 1) return i == dw - 1 ? r[i] : (3 * r[i] + r[i + 1] + 2) >> 2;
        return i == 0 ? r[0] : (3 * r[i] + r[i - 1] + 1) >> 2;
    }
    const int rr = y >> 1;
    if (dw <= 2) return p[(long long)rr * pitch + i];
    const int ro = (y & 1) ? min(rr + 1, dh - 1) : max(rr - 1, 0);       // the context row: below for odd output rows, above for even
    const uint8_t* r0 = p + (long long)rr * pitch;
    const uint8_t* r1 = p + (long long)ro * pitch;
    const int cs = 3 * r0[i] + r1[i];
    if (x & 1) return i == dw - 1 ? (4 * cs + 7) >> 4 : (3 * cs + 3 * r0[i + 1] + r1[i + 1] + 7) >> 4;
    return i == 0 ? (4 * cs + 8) >> 4 : (3 * cs + 3 * r0[i - 1] + r1[i - 1] + 8) >> 4;
}

// four consecutive pixels (row-major index 4t .. 4t+3, possibly across a row end) per thread: 12 output bytes as three dword stores
// (the image's offset in `out` is a multiple of 16; one byte store per sample ran at 0.6 TB/s)
__global__ __launch_bounds__(256) void jpeg_color_kernel(const int32_t* __restrict__ img32, const int64_t* __restrict__ img64,
                                                         const uint8_t* __restrict__ planes, uint8_t* __restrict__ out, int bgr) {
    const int32_t* I = img32 + (long long)blockIdx.y * JP_I32;
    const int64_t* I8 = img64 + (long long)blockIdx.y * JP_I64;
    const int W = I[0], H = I[1], n = W * H;
    const int px0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (px0 >= n) return;
    int y = px0 / W, x = px0 - y * W;
    const bool gray = I[2] == 1;
    const int ypitch = I[8] * 8;
    unsigned char o[12];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int r = 0, g = 0, b = 0;
        if (px0 + e < n) {
            const int yy = planes[I8[3] + (long long)y * ypitch + x];
            if (gray) {
                r = g = b = yy;
            } else {
                const int cb = jp_chroma(planes + I8[4], I[9] * 8, I[15], I[18], I[3], I[4], y, x) - 128;
                const int cr = jp_chroma(planes + I8[5], I[10] * 8, I[16], I[19], I[3], I[4], y, x) - 128;
                // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, ONE_HALF folded into the Cr->R, Cb->B and Cb->G tables
                r = min(max(yy + ((91881 * cr + 32768) >> 16), 0), 255);
                g = min(max(yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16), 0), 255);
                b = min(max(yy + ((116130 * cb + 32768) >> 16), 0), 255);
            }
        }
        o[3 * e] = (unsigned char)(bgr ? b : r);
        o[3 * e + 1] = (unsigned char)g;
        o[3 * e + 2] = (unsigned char)(bgr ? r : b);
        if (++x == W) { x = 0; ++y; }
    }
    uint8_t* dst = out + I8[6] + (long long)px0 * 3;
    if (px0 + 4 <= n) {
        uint32_t w[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = (uint32_t)o[4 * i] | ((uint32_t)o[4 * i + 1] << 8) | ((uint32_t)o[4 * i + 2] << 16) | ((uint32_t)o[4 * i + 3] << 24);
        uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
        d32[0] = w[0]; d32[1] = w[1]; d32[2] = w[2];
    } else {
        for (int i = 0; i < 3 * (n - px0); ++i) dst[i] = o[i];
    }
}

}  // namespace nps

extern "C" int nopesac_jpeg_huffman(const int32_t* img32, const int64_t* img64, const uint8_t* tables, const int32_t* seg32,
                                    const int64_t* seg64, int n_segments, const uint32_t* words, int64_t n_words, int16_t* coef,
                                    const int32_t* par_done, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(img32 && img64 && tables && seg32 && seg64 && words && coef && n_segments > 0 && n_words > 0, "jpeg_huffman: bad args");
    NPS_CHECK_ARG(((uintptr_t)tables & 3) == 0 && ((uintptr_t)coef & 15) == 0, "jpeg_huffman: tables / coef alignment");
    hipLaunchKernelGGL(jpeg_huffman_kernel, dim3(n_segments), dim3(64), 0, (hipStream_t)stream, img32, img64, tables, seg32, seg64, words,
                       (long long)n_words, coef, par_done);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_jpeg_huffman_parallel(const int32_t* img32, const int64_t* img64, const uint8_t* tables, int n_images,
                                             const int32_t* lane_img, int64_t n_lanes, const uint32_t* words, int64_t n_words,
                                             int64_t* exit_state, int64_t* entry_used, int32_t* n_blk, int64_t* first_block,
                                             int32_t* changed, int32_t* par_done, int16_t* coef, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(img32 && img64 && tables && lane_img && words && exit_state && entry_used && n_blk && first_block && changed && par_done && coef,
                  "jpeg_huffman_parallel: null pointer");
    NPS_CHECK_ARG(n_images > 0 && n_lanes > 0 && n_lanes % 64 == 0 && n_words > 0, "jpeg_huffman_parallel: bad sizes (n_lanes: a multiple of 64)");
    const dim3 grid((unsigned)(n_lanes / 64)), blk(64);
    hipStream_t st = (hipStream_t)stream;
    long long* ex = (long long*)exit_state;
    long long* eu = (long long*)entry_used;
    long long* fb = (long long*)first_block;
    hipLaunchKernelGGL(jpeg_sync_kernel<0>, grid, blk, 0, st, img32, img64, tables, lane_img, words, (long long)n_words, ex, eu, n_blk, fb, changed, 0,
                       n_images, par_done, coef);
    for (int pass = 0; pass < JP_ITERS; ++pass)
        hipLaunchKernelGGL(jpeg_sync_kernel<1>, grid, blk, 0, st, img32, img64, tables, lane_img, words, (long long)n_words, ex, eu, n_blk, fb, changed,
                           pass, n_images, par_done, coef);
    hipLaunchKernelGGL(jpeg_sync_scan_kernel, dim3(n_images), dim3(256), 0, st, img32, n_blk, changed, n_images, fb, par_done);
    hipLaunchKernelGGL(jpeg_sync_kernel<2>, grid, blk, 0, st, img32, img64, tables, lane_img, words, (long long)n_words, ex, eu, n_blk, fb, changed, 0,
                       n_images, par_done, coef);
    hipLaunchKernelGGL(jpeg_dc_kernel, dim3(n_images, 3), dim3(256), 0, st, img32, img64, par_done, coef);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_jpeg_idct(const int32_t* img32, const int64_t* img64, const uint8_t* tables, int n_images, int n_blocks,
                                 const int16_t* coef, uint8_t* planes, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(img32 && img64 && tables && coef && planes && n_images > 0 && n_blocks > 0, "jpeg_idct: bad args");
    NPS_CHECK_ARG(((uintptr_t)coef & 15) == 0 && ((uintptr_t)planes & 7) == 0, "jpeg_idct: coef / planes alignment");
    hipLaunchKernelGGL(jpeg_idct_kernel, dim3((n_blocks + 63) / 64), dim3(64), 0, (hipStream_t)stream, img32, img64, tables, n_images, n_blocks,
                       coef, planes);
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_jpeg_color(const int32_t* img32, const int64_t* img64, int n_images, int max_pixels, const uint8_t* planes,
                                  uint8_t* out, int bgr, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(img32 && img64 && planes && out && n_images > 0 && max_pixels > 0, "jpeg_color: bad args");
    NPS_CHECK_ARG(((uintptr_t)out & 15) == 0, "jpeg_color: out must be 16-byte aligned (and so must every image offset img64[.][6])");
    hipLaunchKernelGGL(jpeg_color_kernel, dim3((max_pixels + 1023) / 1024, n_images), dim3(256), 0, (hipStream_t)stream, img32, img64, planes, out, bgr);
    NPS_LAUNCH_RET();
}

// ---- host side of the scan (plain C++, no device work): what nopesac_amd/jpeg.py did with bytes.replace / regex / numpy per file (0.25 ms,
// under the interpreter lock) as one pass over the entropy-coded bytes, callable from many reader threads at once (ctypes releases the lock)
extern "C" int64_t nopesac_jpeg_prepare_scan(const uint8_t* data, int64_t n, int has_restart, uint32_t* words, int64_t words_cap,
                                             int64_t* seg_off, int64_t* seg_cnt, int64_t* seg_bytes, int64_t max_segs, int64_t* consumed) {
    if (!data || n < 0 || !words || !seg_off || !seg_cnt || !seg_bytes || max_segs < 1 || !consumed) return -1;
    int64_t nseg = 0, w = 0, seg_start_w = 0, nbytes = 0, i = 0;
    uint32_t cur = 0;
    int fill = 0;
    auto put = [&](uint8_t b) -> bool {
        cur = (cur << 8) | b;
        ++nbytes;
        if (++fill == 4) {
            if (w >= words_cap) return false;
            words[w++] = cur;
            cur = 0; fill = 0;
        }
        return true;
    };
    auto close_segment = [&]() -> bool {
        if (nseg >= max_segs) return false;
        const int64_t real = nbytes;
        while (fill != 0)
            if (!put(0)) return false;
        if (w + 4 > words_cap) return false;
        for (int k = 0; k < 4; ++k) words[w++] = 0u;                 // the decoder reads ahead; libjpeg also feeds zero bits past the end
        seg_off[nseg] = seg_start_w; seg_cnt[nseg] = w - seg_start_w; seg_bytes[nseg] = real;
        ++nseg;
        seg_start_w = w; nbytes = 0;
        return true;
    };
    while (i < n) {
        if (data[i] != 0xFF) {                                        // the run up to the next FF (1 byte in 256 of coded data): bulk copy
            const uint8_t* q = (const uint8_t*)memchr(data + i, 0xFF, (size_t)(n - i));
            int64_t run = q ? (int64_t)(q - (data + i)) : n - i;
            const uint8_t* src = data + i;
            i += run;
            while (fill != 0 && run > 0) {
                if (!put(*src++)) return -1;
                --run;
            }
            const int64_t nw = run >> 2;
            if (w + nw > words_cap) return -1;
            for (int64_t k = 0; k < nw; ++k) {
                uint32_t v;
                memcpy(&v, src + 4 * k, 4);
                words[w + k] = __builtin_bswap32(v);
            }
            w += nw; nbytes += 4 * nw; src += 4 * nw; run -= 4 * nw;
            while (run > 0) {
                if (!put(*src++)) return -1;
                --run;
            }
            continue;
        }
        if (i + 1 >= n) break;                                        // a lone FF at the end of the buffer: truncated file
        const uint8_t m = data[i + 1];
        if (m == 0x00) {                                              // stuffed zero: the FF is data
            if (!put(0xFF)) return -1;
            i += 2;
        } else if (m >= 0xD0 && m <= 0xD7 && has_restart) {          // RSTn: the interval ends
            if (!close_segment()) return -1;
            i += 2;
        } else if (m == 0xFF) {                                       // fill byte in front of a marker
            ++i;
        } else {
            break;                                                    // any other marker ends the scan (normally EOI)
        }
    }
    if (!close_segment()) return -1;
    *consumed = i;
    return nseg;
}
