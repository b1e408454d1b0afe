// Host-side PNG decode for the data mapper (no kernel in this file): the mp3d split - the headline configuration's dataset - stores its
// frames as 480 x 640 PNG files (reference: data/planercnn_transforms.py:210-227 `call_mp3d` -> detectron2 utils.read_image -> PIL).
// PIL decodes a PNG with the interpreter lock HELD (measured round 5: 85 images/s with one thread, 92 with eight), so the reader threads
// of data.LazyPairs could not scale: 45 pairs/s per process against 3800 pairs/s of model.  This decoder is called through ctypes (lock
// released): chunk walk (PNG spec, ISO/IEC 15948 section 5), zlib inflate of the concatenated IDAT stream, the five row filters of
// section 9 (None / Sub / Up / Average / Paeth), and the colour conversion PIL's `convert("RGB")` applies to the stored mode (grey ->
// replicated, palette -> table lookup, alpha dropped).  Lossless format: the result is the file's samples, bit for bit what PIL returns
// (tests/test_host_cpu.py compares on every supported colour type).  Not handled (-> negative return, the caller falls back to PIL):
// 16-bit samples, sub-byte depths, Adam7 interlacing.
// `nopesac_png_decode_files_host` decodes a whole BATCH of files on its own threads, straight into one (pinned) batch buffer and, if asked,
// channel-major: the per-image Python around a ctypes call (open / read / numpy allocation / transpose - interpreter lock held) capped 32
// reader threads at 6.1 k images/s on a host whose cores inflate 11 k; one call per batch has no per-image interpreter work at all.
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>

#include <immintrin.h>

#include "common.h"
#include "inflate_host.h"

// zlib is OPTIONAL and only serves the NOPESAC_PNG_ZLIB_INFLATE=1 A/B path: the decoder's own inflate (inflate_host.h) needs nothing.
// nopesac_amd/build.py decides once (a test compile + link against -lz) and passes -DNPS_HAVE_ZLIB=0/1 together with -lz, so the object
// never references zlib symbols the link line does not provide (a bare __has_include also finds headers under conda / sysroot paths).
#ifndef NPS_HAVE_ZLIB
#define NPS_HAVE_ZLIB 0
#endif
#if NPS_HAVE_ZLIB
#include <zlib.h>
#endif

namespace {

#if NPS_HAVE_ZLIB
// what one decode needs besides its input and output: the inflated scanlines and the inflate state.  The batch entry point keeps one per
// pool thread (a fresh 900 KB block and a fresh 40 KB inflate state per image are malloc / mmap traffic from dozens of threads at once)
struct PngScratch {
    std::vector<unsigned char> raw, file, idat;
    nps_inflate::Tables tables;
    z_stream zs;
    bool zinit = false;
    ~PngScratch() { if (zinit) inflateEnd(&zs); }
};
#else
struct PngScratch { std::vector<unsigned char> raw, file, idat; nps_inflate::Tables tables; };
#endif

// CRC-32 (ISO 3309, the PNG chunk check) eight bytes per step ("slicing by 8": eight 256-entry tables of the byte-at-a-time table's
// k-fold images): the chunk CRCs were 14 % of a frame's decode with zlib 1.2.11's four-byte version.  Same value as zlib's crc32().
struct Crc32Tables {
    uint32_t t[8][256];
    Crc32Tables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 255];
    }
};
inline uint32_t crc32_bytes(const unsigned char* p, int64_t n) {
    static const Crc32Tables T;
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= c;
        c = T.t[7][a & 255] ^ T.t[6][(a >> 8) & 255] ^ T.t[5][(a >> 16) & 255] ^ T.t[4][a >> 24] ^
            T.t[3][b & 255] ^ T.t[2][(b >> 8) & 255] ^ T.t[1][(b >> 16) & 255] ^ T.t[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n-- > 0) c = T.t[0][(c ^ *p++) & 255] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

inline int paeth(int a, int b, int c) {                      // p = a + b - c: |p - a| = |b - c|, |p - b| = |a - c|, |p - c| = |a + b - 2c|
    // branch-free: on photographic rows the three-way choice is a coin toss for a predictor (measured: the compare-and-branch form
    // cost 2.8 ms per all-Paeth 480 x 640 frame, more than the inflate)
    const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
    const int take_a = -(int)((pa <= pb) & (pa <= pc)), take_b = -(int)(pb <= pc);
    return (a & take_a) | (((b & take_b) | (c & ~take_b)) & ~take_a);
}

// The three filters with a LEFT neighbour, bytes-per-pixel known at compile time.  The left (and for Paeth the upper-left) sample of each
// channel stays in a register: read back from the row, every sample waits for the store of the sample BPP bytes earlier to be forwarded -
// the longest link of a chain that is serial per channel anyway (measured on all-Paeth 480 x 640 RGB frames: 2.2 -> 0.9 ms).  One
// function per filter, not inlined: the code of one must not depend on what the compiler does with the others.
template <int BPP>
__attribute__((noinline)) void unfilter_sub(unsigned char* __restrict__ row, int64_t stride) {
    int a[BPP];
    for (int k = 0; k < BPP; ++k) a[k] = k < stride ? row[k] : 0;
    int64_t i = BPP;
    for (; i + BPP <= stride; i += BPP)
#pragma unroll
        for (int k = 0; k < BPP; ++k) { a[k] = (row[i + k] + a[k]) & 255; row[i + k] = (unsigned char)a[k]; }
    for (; i < stride; ++i) row[i] = (unsigned char)(row[i] + row[i - BPP]);
}

template <int BPP>
__attribute__((noinline)) void unfilter_avg(unsigned char* __restrict__ row, const unsigned char* __restrict__ up, int64_t stride) {
    for (int64_t i = 0; i < BPP && i < stride; ++i) row[i] = (unsigned char)(row[i] + (up[i] >> 1));
    int a[BPP];
    for (int k = 0; k < BPP; ++k) a[k] = k < stride ? row[k] : 0;
    int64_t i = BPP;
    for (; i + BPP <= stride; i += BPP)
#pragma unroll
        for (int k = 0; k < BPP; ++k) { a[k] = (row[i + k] + ((a[k] + up[i + k]) >> 1)) & 255; row[i + k] = (unsigned char)a[k]; }
    for (; i < stride; ++i) row[i] = (unsigned char)(row[i] + ((row[i - BPP] + up[i]) >> 1));
}

// the same pixel step with SSSE3 / SSE4.1 instructions (abs in one instruction, the two selections as blends): 9 instead of 12 dependent
// operations per pixel; chosen at run time.  Returns the first byte index it did not process.
template <int BPP>
__attribute__((target("sse4.1"), noinline)) int64_t unfilter_paeth_sse41(unsigned char* __restrict__ row, const unsigned char* __restrict__ up, int64_t stride) {
    const __m128i zero = _mm_setzero_si128(), lo8 = _mm_set1_epi16(0x00ff);
#define NPS_LOAD4(p) ([&] { int v_; memcpy(&v_, (p), 4); return v_; }())        /* (a lambda does not inherit the target attribute) */
    __m128i a = _mm_cvtepu8_epi16(_mm_cvtsi32_si128(NPS_LOAD4(row))), c = _mm_cvtepu8_epi16(_mm_cvtsi32_si128(NPS_LOAD4(up)));
    int64_t i = BPP;
    for (; i + BPP <= stride; i += BPP) {
        const __m128i b = _mm_cvtepu8_epi16(_mm_cvtsi32_si128(NPS_LOAD4(up + i))), x = _mm_cvtepu8_epi16(_mm_cvtsi32_si128(NPS_LOAD4(row + i)));
        const __m128i dbc = _mm_sub_epi16(b, c), dac = _mm_sub_epi16(a, c);
        const __m128i pa = _mm_abs_epi16(dbc), pb = _mm_abs_epi16(dac), pc = _mm_abs_epi16(_mm_add_epi16(dbc, dac));
        const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
        const __m128i bc = _mm_blendv_epi8(c, b, _mm_cmpeq_epi16(smallest, pb));
        const __m128i pred = _mm_blendv_epi8(bc, a, _mm_cmpeq_epi16(smallest, pa));                      // ties: a, then b, then c
        a = _mm_and_si128(_mm_add_epi16(x, pred), lo8);
        c = b;
        const int out = _mm_cvtsi128_si32(_mm_packus_epi16(a, a));
        if (BPP == 3) {
            const unsigned short lo = (unsigned short)out;
            memcpy(row + i, &lo, 2);
            row[i + 2] = (unsigned char)(out >> 16);
        } else {
            memcpy(row + i, &out, 4);
        }
    }
#undef NPS_LOAD4
    (void)zero;
    return i;
}

template <int BPP>
__attribute__((noinline)) void unfilter_paeth(unsigned char* __restrict__ row, const unsigned char* __restrict__ up, int64_t stride) {
    for (int64_t i = 0; i < BPP && i < stride; ++i) row[i] = (unsigned char)(row[i] + up[i]);       // a = c = 0 -> the predictor is b
    int64_t i = BPP;
    if constexpr (BPP == 3 || BPP == 4) {
        // one PIXEL per step in 16-bit SSE2 lanes (x86-64 baseline): the channels' chains run side by side in one register, and the code
        // is the same whatever the compiler makes of the scalar form (clang's took 2.9 ms per all-Paeth 480 x 640 frame).  |b - c| does not
        // depend on the left pixel and leaves the chain.  Four bytes are READ per pixel (BPP = 3: the fourth is the next pixel's first byte or the
        // slack behind the scanline buffer's end) and BPP bytes written.
        static const bool sse41 = __builtin_cpu_supports("sse4.1");
        if (stride >= 2 * BPP && sse41) {
            i = unfilter_paeth_sse41<BPP>(row, up, stride);
        } else if (stride >= 2 * BPP) {
            const __m128i zero = _mm_setzero_si128(), lo8 = _mm_set1_epi16(0x00ff);
            auto load4 = [&](const unsigned char* p) { int v; memcpy(&v, p, 4); return _mm_unpacklo_epi8(_mm_cvtsi32_si128(v), zero); };
            __m128i a = load4(row), c = load4(up);
            for (; i + BPP <= stride; i += BPP) {
                int xin;
                memcpy(&xin, row + i, 4);
                const __m128i b = load4(up + i), x = _mm_unpacklo_epi8(_mm_cvtsi32_si128(xin), zero);
                const __m128i dbc = _mm_sub_epi16(b, c), dac = _mm_sub_epi16(a, c), dsum = _mm_add_epi16(dbc, dac);
                const __m128i pa = _mm_max_epi16(dbc, _mm_sub_epi16(zero, dbc)), pb = _mm_max_epi16(dac, _mm_sub_epi16(zero, dac));
                const __m128i pc = _mm_max_epi16(dsum, _mm_sub_epi16(zero, dsum));
                const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
                const __m128i ma = _mm_cmpeq_epi16(smallest, pa), mb = _mm_cmpeq_epi16(smallest, pb);       // ties: a, then b, then c
                const __m128i bc = _mm_or_si128(_mm_and_si128(mb, b), _mm_andnot_si128(mb, c));
                const __m128i pred = _mm_or_si128(_mm_and_si128(ma, a), _mm_andnot_si128(ma, bc));
                a = _mm_and_si128(_mm_add_epi16(x, pred), lo8);
                c = b;
                const int out = _mm_cvtsi128_si32(_mm_packus_epi16(a, a));
                if (BPP == 3) {                                  // exactly three bytes: a four-byte store would overlap the next pixel's
                    const unsigned short lo = (unsigned short)out;               // four-byte load by one byte - a store-forwarding stall per
                    memcpy(row + i, &lo, 2);                                      // pixel, in the middle of the chain
                    row[i + 2] = (unsigned char)(out >> 16);
                } else {
                    memcpy(row + i, &out, 4);
                }
            }
        }
    } else if (stride >= 2 * BPP) {
        int a[BPP], c[BPP];
        for (int k = 0; k < BPP; ++k) { a[k] = row[k]; c[k] = up[k]; }
        for (; i + BPP <= stride; i += BPP) {
#pragma unroll
            for (int k = 0; k < BPP; ++k) {
                const int b = up[i + k];
                a[k] = (row[i + k] + paeth(a[k], b, c[k])) & 255;
                c[k] = b;
                row[i + k] = (unsigned char)a[k];
            }
        }
    }
    for (; i < stride; ++i) row[i] = (unsigned char)(row[i] + paeth(row[i - BPP], up[i], up[i - BPP]));
}

template <int BPP>
inline int unfilter_row(unsigned char* __restrict__ row, const unsigned char* __restrict__ up, int64_t stride, int ftype) {
    switch (ftype) {
        case 0: return 0;
        case 1: unfilter_sub<BPP>(row, stride); return 0;
        case 2:
            if (up) for (int64_t i = 0; i < stride; ++i) row[i] = (unsigned char)(row[i] + up[i]);
            return 0;
        case 3:
            if (up) unfilter_avg<BPP>(row, up, stride);
            else for (int64_t i = BPP; i < stride; ++i) row[i] = (unsigned char)(row[i] + (row[i - BPP] >> 1));
            return 0;
        case 4:
            if (up) unfilter_paeth<BPP>(row, up, stride);
            else unfilter_sub<BPP>(row, stride);                 // first row: b = c = 0 -> the predictor is a
            return 0;
        default: return -3;
    }
}

// 16 interleaved RGB pixels -> 16 bytes of each plane (SSSE3 byte shuffles; the scalar loop was 10 % of a frame's decode).  Returns the
// number of pixels handled (a multiple of 16; the caller finishes the row).
__attribute__((target("ssse3"))) inline int rgb_to_planes_ssse3(const unsigned char* __restrict__ row, int W, unsigned char* __restrict__ p0,
                                                                 unsigned char* __restrict__ p1, unsigned char* __restrict__ p2) {
    const __m128i a0 = _mm_setr_epi8(0, 3, 6, 9, 12, 15, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m128i b0 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, 2, 5, 8, 11, 14, -1, -1, -1, -1, -1);
    const __m128i c0 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 1, 4, 7, 10, 13);
    const __m128i a1 = _mm_setr_epi8(1, 4, 7, 10, 13, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m128i b1 = _mm_setr_epi8(-1, -1, -1, -1, -1, 0, 3, 6, 9, 12, 15, -1, -1, -1, -1, -1);
    const __m128i c1 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 2, 5, 8, 11, 14);
    const __m128i a2 = _mm_setr_epi8(2, 5, 8, 11, 14, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m128i b2 = _mm_setr_epi8(-1, -1, -1, -1, -1, 1, 4, 7, 10, 13, -1, -1, -1, -1, -1, -1);
    const __m128i c2 = _mm_setr_epi8(-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 3, 6, 9, 12, 15);
    int x = 0;
    for (; x + 16 <= W; x += 16) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + 3 * x));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + 3 * x + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + 3 * x + 32));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(p0 + x), _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(a, a0), _mm_shuffle_epi8(b, b0)), _mm_shuffle_epi8(c, c0)));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(p1 + x), _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(a, a1), _mm_shuffle_epi8(b, b1)), _mm_shuffle_epi8(c, c1)));
        _mm_storeu_si128(reinterpret_cast<__m128i*>(p2 + x), _mm_or_si128(_mm_or_si128(_mm_shuffle_epi8(a, a2), _mm_shuffle_epi8(b, b2)), _mm_shuffle_epi8(c, c2)));
    }
    return x;
}

}  // namespace

// Geometry of a PNG file without decoding it: 0 and *height / *width / *channels (samples per pixel as stored) / *supported (1 if
// nopesac_png_decode_host can decode it); negative if the bytes are not a PNG header.
extern "C" int nopesac_png_info_host(const unsigned char* data, int64_t n, int* height, int* width, int* channels, int* supported) {
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (!data || n < 33 || memcmp(data, sig, 8) != 0 || be32(data + 8) != 13 || memcmp(data + 12, "IHDR", 4) != 0) return -1;
    const uint32_t w = be32(data + 16), h = be32(data + 20);
    const int depth = data[24], ctype = data[25], comp = data[26], filt = data[27], inter = data[28];
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (height) *height = (int)h;
    if (width) *width = (int)w;
    if (channels) *channels = ch;
    if (supported) *supported = (ch && depth == 8 && comp == 0 && filt == 0 && inter == 0 && w > 0 && h > 0 && w < (1u << 15) && h < (1u << 15)) ? 1 : 0;
    return 0;
}

// data[n] = the file; out = H * W * 3 bytes, RGB (bgr = 0) or BGR (bgr = 1) interleaved.  Returns 0, or: -1 not a PNG, -2 unsupported
// variant (16-bit / sub-byte / interlaced), -3 truncated or corrupt (chunk structure, CRC, inflate, filter byte), -4 out too small,
// (-100 is no longer returned: the decoder's own inflate needs no zlib).  Thread-safe, no global state.
static int png_decode_impl(const unsigned char* data, int64_t n, unsigned char* out, int64_t out_bytes, int bgr, int chw, PngScratch& sc) {
    int H = 0, W = 0, ch = 0, ok = 0;
    if (nopesac_png_info_host(data, n, &H, &W, &ch, &ok) != 0) return -1;
    if (!ok) return -2;
    if (!out || out_bytes < (int64_t)H * W * 3) return -4;
    const int ctype = data[25];
    unsigned char pal[256 * 3];
    memset(pal, 0, sizeof(pal));
    const int64_t stride = (int64_t)W * ch, raw_bytes = (stride + 1) * H;
    if ((int64_t)sc.raw.size() < raw_bytes + 64) sc.raw.resize((size_t)raw_bytes + 64);
    unsigned char* raw = sc.raw.data();
    // The IDAT payloads (CRC-checked chunk by chunk) are gathered into one buffer and inflated in one go by csrc/inflate_host.h;
    // NOPESAC_PNG_ZLIB_INFLATE=1 keeps zlib's streaming inflate (the first round-5 form; A/B and a way out).
#if NPS_HAVE_ZLIB
    static const bool use_zlib = getenv("NOPESAC_PNG_ZLIB_INFLATE") && atoi(getenv("NOPESAC_PNG_ZLIB_INFLATE")) == 1;
    z_stream& zs = sc.zs;
#else
    constexpr bool use_zlib = false;                             // (built without zlib: the A/B switch has nothing to switch to)
#endif
    if (use_zlib) {
#if NPS_HAVE_ZLIB
        if (!sc.zinit) {
            memset(&zs, 0, sizeof(zs));
            if (inflateInit(&zs) != Z_OK) return -3;
            sc.zinit = true;
        } else if (inflateReset(&zs) != Z_OK) {
            return -3;
        }
        zs.next_out = raw;
        zs.avail_out = (uInt)raw_bytes;
#endif
    } else if ((int64_t)sc.idat.size() < n + NPS_INFLATE_IN_SLACK) {
        sc.idat.resize((size_t)n + NPS_INFLATE_IN_SLACK);
    }
    int64_t idat_len = 0;
    int rc = 0, zend = 0, have_plte = 0;
    int64_t p = 8;
    while (true) {
        if (p + 12 > n) { rc = -3; break; }
        const uint32_t L = be32(data + p);
        const unsigned char* type = data + p + 4;
        if ((int64_t)L > n - p - 12) { rc = -3; break; }
        const unsigned char* body = data + p + 8;
        if (crc32_bytes(type, (int64_t)L + 4) != be32(body + L)) { rc = -3; break; }
        if (memcmp(type, "IDAT", 4) == 0) {
            if (!use_zlib) {
                memcpy(sc.idat.data() + idat_len, body, L);
                idat_len += L;
            }
#if NPS_HAVE_ZLIB
            else if (!zend && L) {
                zs.next_in = (Bytef*)body;
                zs.avail_in = L;
                const int r = inflate(&zs, Z_NO_FLUSH);
                if (r == Z_STREAM_END) zend = 1;
                else if (r != Z_OK && !(r == Z_BUF_ERROR && zs.avail_out == 0)) { rc = -3; break; }
            }
#endif
        } else if (memcmp(type, "PLTE", 4) == 0) {
            if (L % 3 != 0 || L > 768) { rc = -3; break; }
            memcpy(pal, body, L);
            have_plte = 1;
        } else if (memcmp(type, "IEND", 4) == 0) {
            break;
        }
        p += 12 + (int64_t)L;
    }
    int64_t got;
    if (use_zlib) {
#if NPS_HAVE_ZLIB
        got = raw_bytes - (int64_t)zs.avail_out;
#else
        got = -1;
#endif
    } else {
        memset(sc.idat.data() + idat_len, 0, NPS_INFLATE_IN_SLACK);
        got = rc == 0 ? nps_inflate::inflate_zlib(sc.idat.data(), idat_len, raw, raw_bytes, sc.tables) : -1;
    }
    if (rc == 0 && (got != raw_bytes || (ctype == 3 && !have_plte))) rc = -3;
    if (rc != 0) return rc;
    // ---- row filters (in place: a row's reconstructed samples are the next row's "up" samples)
    for (int y = 0; y < H && rc == 0; ++y) {
        unsigned char* row = raw + (int64_t)y * (stride + 1) + 1;
        const unsigned char* up = y ? row - (stride + 1) : nullptr;
        const int ft = row[-1];
        rc = ch == 3 ? unfilter_row<3>(row, up, stride, ft) : ch == 4 ? unfilter_row<4>(row, up, stride, ft)
           : ch == 1 ? unfilter_row<1>(row, up, stride, ft) : unfilter_row<2>(row, up, stride, ft);
    }
    if (rc != 0) return rc;
    // ---- stored mode -> RGB / BGR (what PIL's convert("RGB") gives: grey replicated, palette looked up, alpha dropped)
    const int r0 = bgr ? 2 : 0, b0 = bgr ? 0 : 2;
    for (int y = 0; y < H && chw; ++y) {                         // channel-major [3][H][W]: plane r0 = red, 1 = green, b0 = blue
        const unsigned char* row = raw + (int64_t)y * (stride + 1) + 1;
        unsigned char* pr = out + ((int64_t)r0 * H + y) * W;
        unsigned char* pg = out + ((int64_t)1 * H + y) * W;
        unsigned char* pb = out + ((int64_t)b0 * H + y) * W;
        if (ctype == 2 || ctype == 6) {
            static const bool ssse3 = __builtin_cpu_supports("ssse3");
            const int x0 = (ch == 3 && ssse3) ? rgb_to_planes_ssse3(row, W, pr, pg, pb) : 0;
            for (int x = x0; x < W; ++x) { pr[x] = row[ch * x]; pg[x] = row[ch * x + 1]; pb[x] = row[ch * x + 2]; }
        } else if (ctype == 0 || ctype == 4) {
            for (int x = 0; x < W; ++x) { const unsigned char v = row[ch * x]; pr[x] = v; pg[x] = v; pb[x] = v; }
        } else {
            for (int x = 0; x < W; ++x) { const unsigned char* c = pal + 3 * row[x]; pr[x] = c[0]; pg[x] = c[1]; pb[x] = c[2]; }
        }
    }
    for (int y = 0; y < H && !chw; ++y) {
        const unsigned char* row = raw + (int64_t)y * (stride + 1) + 1;
        unsigned char* o = out + (int64_t)y * W * 3;
        if (ctype == 2 || ctype == 6) {
            for (int x = 0; x < W; ++x) { o[3 * x + r0] = row[ch * x]; o[3 * x + 1] = row[ch * x + 1]; o[3 * x + b0] = row[ch * x + 2]; }
        } else if (ctype == 0 || ctype == 4) {
            for (int x = 0; x < W; ++x) { const unsigned char v = row[ch * x]; o[3 * x] = v; o[3 * x + 1] = v; o[3 * x + 2] = v; }
        } else {
            for (int x = 0; x < W; ++x) { const unsigned char* c = pal + 3 * row[x]; o[3 * x + r0] = c[0]; o[3 * x + 1] = c[1]; o[3 * x + b0] = c[2]; }
        }
    }
    return 0;
}

extern "C" int nopesac_png_decode_host(const unsigned char* data, int64_t n, unsigned char* out, int64_t out_bytes, int bgr) {
    PngScratch sc;
    return png_decode_impl(data, n, out, out_bytes, bgr, 0, sc);
}

namespace {
// The batch decoder's threads: created once, parked on a condition variable between batches (a std::thread per call and per core is 31
// stack mappings made and torn down per batch).  One batch at a time
// (callers queue on `job`); the calling thread works too.  Never destroyed: the threads are parked, not joined, at process exit.
struct PngPool {
    std::mutex job, m;
    std::condition_variable wake, done;
    std::vector<std::thread> threads;
    std::function<void(PngScratch&)> work;
    uint64_t generation = 0;
    int wanted = 0, running = 0;

    void worker(int index) {
        PngScratch sc;
        uint64_t seen = 0;
        while (true) {
            std::function<void(PngScratch&)> w;
            {
                std::unique_lock<std::mutex> lk(m);
                wake.wait(lk, [&] { return generation != seen && index < wanted; });
                seen = generation;
                w = work;
            }
            w(sc);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--running == 0) done.notify_all();
            }
        }
    }

    void run(int helpers, const std::function<void(PngScratch&)>& w, PngScratch& mine) {
        std::lock_guard<std::mutex> one(job);
        {
            std::lock_guard<std::mutex> lk(m);
            while ((int)threads.size() < helpers) {
                const int index = (int)threads.size();
                threads.emplace_back([this, index] { worker(index); });
                threads.back().detach();
            }
            work = w;
            wanted = helpers;
            running = helpers;
            ++generation;
        }
        wake.notify_all();
        w(mine);
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return running == 0; });
        wanted = 0;
    }
};

// (a forked child inherits the pool object but none of its threads: it gets a fresh pool - the old one is leaked, its mutexes may have
//  been held at the moment of the fork)
std::atomic<PngPool*> g_png_pool{nullptr};
std::once_flag g_png_atfork;

PngPool& png_pool() {
    std::call_once(g_png_atfork, [] { pthread_atfork(nullptr, nullptr, [] { g_png_pool.store(nullptr); }); });
    PngPool* p = g_png_pool.load();
    if (!p) {
        PngPool* fresh = new PngPool();
        if (g_png_pool.compare_exchange_strong(p, fresh)) p = fresh;
        else delete fresh;
    }
    return *p;
}
}  // namespace

// A batch of PNG FILES decoded by `threads` threads of this call (no interpreter involved): file i of paths[n] -> out + i * image_stride,
// H x W x 3 interleaved or (flags bit 1) 3 x H x W channel-major, RGB or (flags bit 0) BGR.  status[i] = 0, or the code of
// nopesac_png_decode_host, or -5 the file's geometry is not H x W, -6 the file cannot be read.  Returns the number of files with a
// non-zero status (the caller decodes those with PIL), negative on bad arguments.
extern "C" int nopesac_png_decode_files_host(const char* const* paths, int n, unsigned char* out, int64_t image_stride, int H, int W, int flags,
                                             int threads, int* status) {
    if (!paths || !out || !status || n < 0 || H <= 0 || W <= 0 || image_stride < (int64_t)H * W * 3) return -1;
    std::atomic<int> next(0), failed(0);
    auto work = [&](PngScratch& sc) {
        std::vector<unsigned char>& file = sc.file;
        while (true) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            int rc = -6;
            FILE* f = paths[i] ? fopen(paths[i], "rb") : nullptr;
            if (f) {
                if (fseek(f, 0, SEEK_END) == 0) {
                    const long sz = ftell(f);
                    if (sz > 0 && fseek(f, 0, SEEK_SET) == 0) {
                        if ((long)file.size() < sz) file.resize((size_t)sz);
                        if (fread(file.data(), 1, (size_t)sz, f) == (size_t)sz) {
                            int h = 0, w = 0, ch = 0, ok = 0;
                            if (nopesac_png_info_host(file.data(), sz, &h, &w, &ch, &ok) != 0) rc = -1;
                            else if (!ok) rc = -2;
                            else if (h != H || w != W) rc = -5;
                            else rc = png_decode_impl(file.data(), sz, out + (int64_t)i * image_stride, (int64_t)H * W * 3, flags & 1, (flags >> 1) & 1, sc);
                        }
                    }
                }
                fclose(f);
            }
            status[i] = rc;
            if (rc != 0) failed.fetch_add(1);
        }
    };
    const int T = threads < 1 ? 1 : (threads > n ? (n > 0 ? n : 1) : threads);
    PngScratch mine;
    if (T <= 1) work(mine);
    else png_pool().run(T - 1, work, mine);
    return failed.load();
}

// The inflate of csrc/inflate_host.h on its own (tests: against zlib on every block type, level and strategy, and on damaged streams):
// a whole zlib stream in[n] -> out[out_cap]; returns the number of bytes written or -1.  No slack is required of the caller's buffers.
extern "C" int64_t nopesac_inflate_zlib_host(const unsigned char* in, int64_t n, unsigned char* out, int64_t out_cap) {
    if (!in || n < 0 || (!out && out_cap > 0) || out_cap < 0) return -1;
    std::vector<unsigned char> src((size_t)n + NPS_INFLATE_IN_SLACK, 0), dst((size_t)out_cap + 16);
    memcpy(src.data(), in, (size_t)n);
    PngScratch* sc = new PngScratch();
    const int64_t got = nps_inflate::inflate_zlib(src.data(), n, dst.data(), out_cap, sc->tables);
    delete sc;
    if (got > 0) memcpy(out, dst.data(), (size_t)got);
    return got;
}
