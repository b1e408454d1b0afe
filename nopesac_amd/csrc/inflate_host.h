// Host-side inflate (RFC 1951 deflate streams inside the RFC 1950 zlib wrapper) of a WHOLE stream held in memory into a buffer of known
// size - the IDAT stream of a PNG frame (csrc/png_host.hip).  zlib's streaming inflate is the floor of the mp3d split's input path (2.45 of
// the 2.95 ms a 480 x 640 frame costs on one core of the MI355X box); a decoder that never has to suspend can keep 56+ bits in a register,
// refill without a branch, resolve a symbol with one table load (11-bit primary table for literals / lengths, 9-bit for distances,
// second-level tables behind long codes), emit up to three literals per refill and copy matches in 8-byte steps into an output buffer with
// slack.  Written from the RFCs; the table layout and the loop are this file's own.
//
// Contract: `in` must be readable for 24 bytes (NPS_INFLATE_IN_SLACK) past `in_len` (the caller concatenates the IDAT payloads into scratch
// with zeroed slack: the loop lets ip reach in_end + 8 before a refill, the refill advances it by up to 7, and the second refill in front of a
// distance code then loads the 8 bytes at in_end + 15 .. + 23);
// `out` must be writable for 16 bytes past `out_cap` (match copies run up to 7 bytes over their end, never past out_cap + 8).  Returns the
// number of bytes written, or -1 on any malformed input (bad header, invalid / over-subscribed / incomplete code, distance too far back,
// output overflow, input exhausted).  The Adler-32 trailer is NOT checked here: a PNG chunk's CRC-32 has covered the same bytes already.
#pragma once
#define NPS_INFLATE_IN_SLACK 24
#include <cstdint>
#include <cstring>

namespace nps_inflate {

constexpr int LIT_BITS = 11, DIST_BITS = 9;
constexpr int LIT_TABLE = (1 << LIT_BITS) + 288 * 16, DIST_TABLE = (1 << DIST_BITS) + 32 * 64;
enum : uint32_t { K_LIT = 0, K_BASE = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4 };
// entry: bits 0-4 code bits to consume, 5-7 kind, 8-12 extra-bit count (K_BASE) / index bits of the second-level table (K_SUB),
//        13-14 number of literals (K_LIT: 1 or 2), 16-31 literal / base value / offset of the second-level table.  A K_LIT entry can
//        hold TWO literals whose codes together fit the primary index (first in bits 16-23, second in 24-31, bits 0-4 = both code
//        lengths): a literal-heavy stream (photographic PNG rows: 4-6 bits per symbol) then resolves up to two symbols per table load -
//        the load is the loop's critical path - and the emit is the same two-byte store either way (no branch on one-or-two: in a stream
//        where half the pairs fit, that branch is a coin toss)
inline constexpr uint32_t entry(uint32_t nbits, uint32_t kind, uint32_t extra, uint32_t value, uint32_t nlit = 0) {
    return nbits | (kind << 5) | (extra << 8) | (nlit << 13) | (value << 16);
}

struct Tables {
    uint32_t lit[LIT_TABLE];
    uint32_t dist[DIST_TABLE];
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t code, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// Canonical Huffman code of lens[n] (RFC 1951 3.2.2) -> decode table indexed by the next `primary` stream bits (LSB first).
// sym_entry(symbol, code bits to consume) makes the entry of a symbol.  false: over-subscribed, or incomplete (other than the one-code
// case deflate allows for distances), or a table overflow.
template <typename F>
inline bool build_table(const uint8_t* lens, int n, int primary, uint32_t* table, int capacity, bool allow_incomplete, F sym_entry) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    int used = 0, left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;                              // over-subscribed
        used += count[l];
    }
    const int size = 1 << primary;
    for (int i = 0; i < size; ++i) table[i] = entry(1, K_BAD, 0, 0);
    if (used == 0) return allow_incomplete;                      // no code at all: legal only where the block never uses one
    if (left > 0 && !(allow_incomplete && used == 1 && count[1] == 1)) return false;   // incomplete set (zlib: only a single 1-bit code)
    uint32_t next[16], code = 0;
    for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
    // pass 1: the longest code behind every primary index that leads to long codes
    uint8_t sub_bits[1 << LIT_BITS];
    memset(sub_bits, 0, (size_t)size);
    {
        uint32_t nx[16];
        memcpy(nx, next, sizeof(nx));
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (!l) continue;
            const uint32_t r = reverse_bits(nx[l]++, l);
            if (l > primary) {
                const uint32_t p = r & (uint32_t)(size - 1);
                if (l - primary > sub_bits[p]) sub_bits[p] = (uint8_t)(l - primary);
            }
        }
    }
    int top = size;
    for (int p = 0; p < size; ++p) {
        if (!sub_bits[p]) continue;
        const int sz = 1 << sub_bits[p];
        if (top + sz > capacity) return false;
        table[p] = entry((uint32_t)primary, K_SUB, sub_bits[p], (uint32_t)top);
        for (int i = 0; i < sz; ++i) table[top + i] = entry(1, K_BAD, 0, 0);
        top += sz;
    }
    // pass 2: fill
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = reverse_bits(next[l]++, l);
        if (l <= primary) {
            const uint32_t e = sym_entry(s, l);
            for (uint32_t i = r; i < (uint32_t)size; i += 1u << l) table[i] = e;
        } else {
            const uint32_t p = r & (uint32_t)(size - 1), head = table[p];
            const int sb = (int)((head >> 8) & 31), base = (int)(head >> 16), rem = l - primary;
            const uint32_t e = sym_entry(s, rem);
            for (uint32_t i = r >> primary; i < (1u << sb); i += 1u << rem) table[base + i] = e;
        }
    }
    return true;
}

inline uint32_t litlen_entry(int s, int nbits) {
    if (s < 256) return entry((uint32_t)nbits, K_LIT, 0, (uint32_t)s, 1);
    if (s == 256) return entry((uint32_t)nbits, K_EOB, 0, 0);
    if (s > 285) return entry((uint32_t)nbits, K_BAD, 0, 0);
    return entry((uint32_t)nbits, K_BASE, LEN_EXTRA[s - 257], LEN_BASE[s - 257]);
}
inline uint32_t dist_entry(int s, int nbits) {
    if (s > 29) return entry((uint32_t)nbits, K_BAD, 0, 0);
    return entry((uint32_t)nbits, K_BASE, DIST_EXTRA[s], DIST_BASE[s]);
}

// after build_table of a literal / length code: merge pairs of short literals into two-literal entries (reads the one-symbol entries it is
// about to overwrite only at indices whose known bits already decide them, so a scratch copy of the primary table is all it needs)
inline void pair_literals(uint32_t* table) {
    constexpr int size = 1 << LIT_BITS;
    uint32_t one[size];
    memcpy(one, table, sizeof(one));
    for (int i = 0; i < size; ++i) {
        const uint32_t e1 = one[i];
        if (((e1 >> 5) & 7) != K_LIT) continue;                 // (entries of `one` hold one literal each)
        const uint32_t l1 = e1 & 31;
        if (l1 >= LIT_BITS) continue;
        const uint32_t e2 = one[(uint32_t)i >> l1];              // the index bits behind the first code, zero-extended
        if (((e2 >> 5) & 7) != K_LIT || (e2 & 31) > LIT_BITS - l1) continue;      // (decided only if its code ends inside the known bits)
        table[i] = entry(l1 + (e2 & 31), K_LIT, 0, (e1 >> 16) | ((e2 >> 16) << 8), 2);
    }
}

inline uint64_t load64(const unsigned char* p) {
    uint64_t v;
    memcpy(&v, p, 8);                                            // (little-endian host: x86-64 / the build's only target)
    return v;
}

// `t`: scratch for the block tables (26 KB + 10 KB; the caller keeps one per thread)
inline int64_t inflate_zlib(const unsigned char* in, int64_t in_len, unsigned char* out, int64_t out_cap, Tables& t) {
    if (in_len < 6) return -1;
    if ((in[0] & 15) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 32)) return -1;   // CM = 8, window <= 32 K, FCHECK, no preset dictionary
    const unsigned char* const in_end = in + in_len;
    const unsigned char* const in_limit = in_end + 8;           // reads stay inside NPS_INFLATE_IN_SLACK bytes past in_end
    const unsigned char* ip = in + 2;
    unsigned char* op = out;
    unsigned char* const out_end = out + out_cap;
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0;
#define NPS_REFILL()                                                     \
    do {                                                                 \
        bitbuf |= load64(ip) << bitcnt;                                  \
        ip += (63 - bitcnt) >> 3;                                        \
        bitcnt |= 56;                                                    \
    } while (0)
#define NPS_TAKE(n) (bitbuf >>= (n), bitcnt -= (uint32_t)(n))
    bool last = false;
    while (!last) {
        if (ip > in_limit) return -1;
        NPS_REFILL();
        last = bitbuf & 1;
        const uint32_t type = (uint32_t)(bitbuf >> 1) & 3;
        NPS_TAKE(3);
        if (type == 0) {                                         // stored: to the byte boundary, LEN, ~LEN, bytes
            NPS_TAKE(bitcnt & 7);
            const unsigned char* p = ip - (bitcnt >> 3);         // first byte not yet in the (now byte-aligned) bit buffer ... is behind it
            bitbuf = 0;
            bitcnt = 0;
            if (p + 4 > in_end) return -1;
            const uint32_t len = p[0] | ((uint32_t)p[1] << 8), nlen = p[2] | ((uint32_t)p[3] << 8);
            if ((len ^ 0xffffu) != nlen) return -1;
            p += 4;
            if (p + len > in_end || op + len > out_end) return -1;
            memcpy(op, p, len);
            op += len;
            ip = p + len;
            continue;
        }
        if (type == 3) return -1;
        if (type == 1) {                                         // fixed codes (RFC 1951 3.2.6)
            uint8_t lens[288 + 32];
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
            if (!build_table(lens, 288, LIT_BITS, t.lit, LIT_TABLE, false, litlen_entry)) return -1;
            if (!build_table(lens + 288, 32, DIST_BITS, t.dist, DIST_TABLE, false, dist_entry)) return -1;
            pair_literals(t.lit);
        } else {                                                 // dynamic codes (3.2.7)
            const int hlit = (int)(bitbuf & 31) + 257, hdist = (int)((bitbuf >> 5) & 31) + 1, hclen = (int)((bitbuf >> 10) & 15) + 4;
            NPS_TAKE(14);
            if (hlit > 286 || hdist > 30) return -1;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; ++i) {
                if (bitcnt < 3) { if (ip > in_limit) return -1; NPS_REFILL(); }
                cl[order[i]] = (uint8_t)(bitbuf & 7);
                NPS_TAKE(3);
            }
            uint32_t ct[128];
            if (!build_table(cl, 19, 7, ct, 128, false, [](int s, int nb) { return entry((uint32_t)nb, K_LIT, 0, (uint32_t)s); })) return -1;
            uint8_t lens[286 + 30 + 138];
            int n = 0;
            const int total = hlit + hdist;
            while (n < total) {
                if (ip > in_limit) return -1;
                NPS_REFILL();
                const uint32_t e = ct[bitbuf & 127];
                if (((e >> 5) & 7) != K_LIT) return -1;
                NPS_TAKE(e & 31);
                const uint32_t s = e >> 16;
                if (s < 16) { lens[n++] = (uint8_t)s; continue; }
                int rep;
                uint8_t v = 0;
                if (s == 16) {
                    if (n == 0) return -1;
                    v = lens[n - 1];
                    rep = 3 + (int)(bitbuf & 3);
                    NPS_TAKE(2);
                } else if (s == 17) {
                    rep = 3 + (int)(bitbuf & 7);
                    NPS_TAKE(3);
                } else {
                    rep = 11 + (int)(bitbuf & 127);
                    NPS_TAKE(7);
                }
                if (n + rep > total) return -1;
                memset(lens + n, v, (size_t)rep);
                n += rep;
            }
            if (lens[256] == 0) return -1;                       // no end-of-block code
            if (!build_table(lens, hlit, LIT_BITS, t.lit, LIT_TABLE, true, litlen_entry)) return -1;
            if (!build_table(lens + hlit, hdist, DIST_BITS, t.dist, DIST_TABLE, true, dist_entry)) return -1;
            pair_literals(t.lit);
        }
        // ---- the block's symbols
        const uint32_t* const lt = t.lit;
        const uint32_t* const dt = t.dist;
        for (;;) {
            if (ip > in_limit) return -1;
            NPS_REFILL();                                        // >= 56 bits: three codes of <= 15 bits, then a length's <= 5 extra bits
            uint32_t e = lt[bitbuf & ((1u << LIT_BITS) - 1)];
#define NPS_LITLEN_STEP()                                                                    \
    if (((e >> 5) & 7) == K_SUB) {                                                           \
        NPS_TAKE(LIT_BITS);                                                                  \
        e = lt[(e >> 16) + (uint32_t)(bitbuf & ((1u << ((e >> 8) & 31)) - 1))];             \
    }                                                                                        \
    NPS_TAKE(e & 31);
            NPS_LITLEN_STEP();
            // up to three table loads per refill, each one or two literals; the first entry that is neither falls through with its code
            // bits taken and >= 11 bits left (a length's extra bits are <= 5)
#define NPS_EMIT_OR(done)                                                                   \
    {                                                                                        \
        if (((e >> 5) & 7) != K_LIT) goto done;                                              \
        const uint32_t nlit = (e >> 13) & 3;                                                 \
        if (op + nlit > out_end) return -1;                                                  \
        const uint16_t two = (uint16_t)(e >> 16);                                            \
        memcpy(op, &two, 2);                  /* (one literal: the second byte lands in the slack or under the next symbol) */ \
        op += nlit;                                                                          \
    }
            NPS_EMIT_OR(not_a_literal)
            e = lt[bitbuf & ((1u << LIT_BITS) - 1)];
            NPS_LITLEN_STEP();
            NPS_EMIT_OR(not_a_literal)
            e = lt[bitbuf & ((1u << LIT_BITS) - 1)];
            NPS_LITLEN_STEP();
            NPS_EMIT_OR(not_a_literal)
            continue;
        not_a_literal:
            const uint32_t kind = (e >> 5) & 7;
            if (kind == K_EOB) break;
            if (kind != K_BASE) return -1;
            const uint32_t lx = (e >> 8) & 31;
            const uint32_t length = (e >> 16) + (uint32_t)(bitbuf & ((1u << lx) - 1));
            NPS_TAKE(lx);
            NPS_REFILL();                                        // a distance: <= 15 code bits + <= 13 extra bits
            uint32_t d = dt[bitbuf & ((1u << DIST_BITS) - 1)];
            if (((d >> 5) & 7) == K_SUB) {
                NPS_TAKE(DIST_BITS);
                d = dt[(d >> 16) + (uint32_t)(bitbuf & ((1u << ((d >> 8) & 31)) - 1))];
            }
            NPS_TAKE(d & 31);
            if (((d >> 5) & 7) != K_BASE) return -1;
            const uint32_t dx = (d >> 8) & 31;
            const uint32_t distance = (d >> 16) + (uint32_t)(bitbuf & ((1u << dx) - 1));
            NPS_TAKE(dx);
            if ((int64_t)distance > op - out || op + length > out_end) return -1;
            const unsigned char* src = op - distance;
            unsigned char* dst = op;
            op += length;
            if (distance >= 8) {                                  // 8 bytes at a time, up to 7 past the end (slack)
                do {
                    memcpy(dst, src, 8);
                    dst += 8;
                    src += 8;
                } while (dst < op);
            } else if (distance == 1) {
                memset(dst, *src, length);
            } else {
                // period 2..7 (PNG: the pixel stride of filtered rows): the first D = distance * ceil(8 / distance) bytes one at a time,
                // the rest 8 at a time from D back - the output has period `distance`, so D back holds the same bytes as `distance` back
                const uint32_t D = distance * ((7 + distance) / distance);
                unsigned char* const head = dst + (length < D ? length : D);
                do { *dst++ = *src++; } while (dst < head);
                src = dst - D;
                while (dst < op) {
                    memcpy(dst, src, 8);
                    dst += 8;
                    src += 8;
                }
            }
        }
    }
#undef NPS_EMIT_OR
#undef NPS_LITLEN_STEP
#undef NPS_TAKE
#undef NPS_REFILL
    if (ip - (bitcnt >> 3) > in_end) return -1;                  // more bits consumed than the stream holds
    return op - out;
}

}  // namespace nps_inflate
