// Host side of the GPU JPEG decoder for a whole BATCH of files (no kernel in this file): file read, marker walk (ITU T.81 Annex B), the
// checks that keep the decoder to baseline files it handles, stuffing removal / restart split (nopesac_jpeg_prepare_scan), derived Huffman
// tables, and the launch arguments of nopesac_jpeg_huffman[_parallel] / _idct / _color for the batch - what nopesac_amd/jpeg.py does per
// file in Python (parse_markers, huffman_table_bytes, prepare_batch).  The Python form holds the interpreter lock for ~80 us per file; with
// 64 files per batch next to the thread that launches the model that was 5 of the 13 ms of interpreter time a batch cost, and the ScanNet-
// style split ran end to end at 2100 pairs/s against the model's 3800.  Here a batch is one call that runs on its own threads.
// Reference behaviour replaced: detectron2 utils.read_image -> PIL (data/planercnn_transforms.py:306-314), as for csrc/jpeg.hip.
// Semantics = jpeg.py's, checked array for array by tests/test_host_cpu.py; any file outside the supported subset fails the batch's
// fast path (status[i] != 0) and the caller falls back to the per-file Python path, which decides about PIL.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "common.h"

namespace {

constexpr int HUFF_BYTES = NOPESAC_JPEG_HUFF_BYTES, TABLES_BYTES = NOPESAC_JPEG_TABLES_BYTES, I32 = NOPESAC_JPEG_IMG_I32, I64 = NOPESAC_JPEG_IMG_I64;
constexpr int LOOK_BITS = 9;
const int ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
                        62, 63};

struct Comp {
    int id = 0, h = 0, v = 0, tq = 0, td = -1, ta = -1, bw = 0, bh = 0, dw = 0, dh = 0;
};

struct FileInfo {
    int status = -6;                      // 0 ok; -1 not a JPEG, -2 unsupported variant, -3 malformed / truncated, -6 unreadable
    int width = 0, height = 0, ncomp = 0, dri = 0, hmax = 0, vmax = 0, mcux = 0, mcuy = 0;
    Comp comps[3];
    bool have_qt[16] = {false};
    uint16_t qt[16][64];                  // natural order
    bool have_huff[2][2] = {{false, false}, {false, false}};
    std::vector<uint8_t> huff[2][2];      // BITS[16] + HUFFVAL
    std::vector<uint32_t> words;
    std::vector<int64_t> seg_off, seg_cnt, seg_bytes;
};

inline int u16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

// jpeg.py huffman_table_bytes: BITS + HUFFVAL -> look u16[512] | maxcode i32[18] | valoffset i32[18] | huffval u8[256] (T.81 Annex C codes,
// jdhuff.c's derived table).  false: bad table.
bool derived_table(const std::vector<uint8_t>& spec, uint8_t* out) {
    if (spec.size() < 16) return false;
    int total = 0;
    for (int i = 0; i < 16; ++i) total += spec[i];
    const int nval = (int)spec.size() - 16;
    if (total != nval || nval > 256) return false;
    memset(out, 0, HUFF_BYTES);
    uint16_t* look = reinterpret_cast<uint16_t*>(out);
    int32_t* maxcode = reinterpret_cast<int32_t*>(out + 1024);
    int32_t* valoff = reinterpret_cast<int32_t*>(out + 1096);
    for (int i = 0; i < 18; ++i) { maxcode[i] = -1; valoff[i] = 0; }
    int code = 0, k = 0;
    for (int ln = 1; ln <= 16; ++ln) {
        const int cnt = spec[ln - 1];
        // Kraft check BEFORE any table write: code + cnt codes of this length must fit ln bits (a crafted BITS such as {255, 0, ...}
        // passes the total == nval test and would otherwise index look[] far past its 512 entries)
        if (code + cnt > (1 << ln)) return false;
        if (cnt) {
            valoff[ln] = k - code;
            for (int j = 0; j < cnt; ++j) {
                if (ln <= LOOK_BITS) {
                    const int lo = code << (LOOK_BITS - ln), span = 1 << (LOOK_BITS - ln);
                    for (int q = 0; q < span; ++q) look[lo + q] = (uint16_t)((ln << 8) | spec[16 + k]);
                }
                ++code;
                ++k;
            }
            maxcode[ln] = code - 1;
        }
        if (code > (1 << ln)) return false;
        code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    memcpy(out + 1168, spec.data() + 16, (size_t)nval);
    return true;
}

// jpeg.py parse_markers (fast path) on one file's bytes
void parse_file(const std::vector<uint8_t>& buf, FileInfo& f) {
    const uint8_t* data = buf.data();
    const int64_t n = (int64_t)buf.size();
    f.status = -2;
    if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) { f.status = -1; return; }
    int adobe = -1;
    bool jfif = false, have_frame = false, have_scan = false;
    int64_t p = 2;
    while (p + 4 <= n) {
        if (data[p] != 0xFF) return;
        while (p < n && data[p] == 0xFF) ++p;
        if (p >= n) { f.status = -3; return; }
        const int m = data[p++];
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (m == 0xD9) break;
        if (p + 2 > n) { f.status = -3; return; }
        const int L = u16(data + p);
        if (L < 2 || p + L > n) { f.status = -3; return; }
        const uint8_t* seg = data + p + 2;
        const int sl = L - 2;
        if (m == 0xDB) {
            int s = 0;
            while (s < sl) {
                const int pq = seg[s] >> 4, tq = seg[s] & 15;
                ++s;
                if (s + (pq ? 128 : 64) > sl) { f.status = -3; return; }
                for (int k = 0; k < 64; ++k) f.qt[tq][ZIGZAG[k]] = pq ? (uint16_t)u16(seg + s + 2 * k) : seg[s + k];
                s += pq ? 128 : 64;
                f.have_qt[tq] = true;
            }
        } else if (m == 0xC4) {
            int s = 0;
            while (s < sl) {
                const int tc = seg[s] >> 4, th = seg[s] & 15;
                if (s + 17 > sl) { f.status = -3; return; }
                int cnt = 0;
                for (int i = 0; i < 16; ++i) cnt += seg[s + 1 + i];
                if (s + 17 + cnt > sl || cnt > 256) { f.status = -3; return; }
                for (int ln = 1, code = 0; ln <= 16; ++ln) {     // Kraft: the canonical codes of every length must fit that length
                    code += seg[s + ln];
                    if (code > (1 << ln)) { f.status = -3; return; }
                    code <<= 1;
                }
                if (tc <= 1 && th <= 1) {
                    f.huff[tc][th].assign(seg + s + 1, seg + s + 17 + cnt);
                    f.have_huff[tc][th] = true;
                }
                s += 17 + cnt;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || seg[0] != 8) return;
            f.height = u16(seg + 1);
            f.width = u16(seg + 3);
            const int nc = seg[5];
            if (sl < 6 + 3 * nc) { f.status = -3; return; }
            if (nc != 1 && nc != 3) return;
            f.ncomp = nc;
            for (int i = 0; i < nc; ++i) {
                f.comps[i] = Comp();
                f.comps[i].id = seg[6 + 3 * i];
                f.comps[i].h = seg[7 + 3 * i] >> 4;
                f.comps[i].v = seg[7 + 3 * i] & 15;
                f.comps[i].tq = seg[8 + 3 * i];
            }
            have_frame = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return;                                               // progressive / lossless / arithmetic coding
        } else if (m == 0xDD) {
            if (sl < 2) { f.status = -3; return; }
            f.dri = u16(seg);
        } else if (m == 0xEE && sl >= 12 && memcmp(seg, "Adobe", 5) == 0) {
            adobe = seg[11];
        } else if (m == 0xE0 && sl >= 5 && memcmp(seg, "JFIF\0", 5) == 0) {
            jfif = true;
        } else if (m == 0xDA) {
            if (!have_frame) return;
            if (sl < 1) { f.status = -3; return; }
            const int ns = seg[0];
            if (ns != f.ncomp) return;                            // more than one scan
            if (sl < 4 + 2 * ns) { f.status = -3; return; }
            for (int i = 0; i < ns; ++i) {
                Comp* c = nullptr;
                for (int j = 0; j < f.ncomp && !c; ++j)
                    if (f.comps[j].id == seg[1 + 2 * i]) c = &f.comps[j];
                if (!c) return;
                c->td = seg[2 + 2 * i] >> 4;
                c->ta = seg[2 + 2 * i] & 15;
            }
            if (seg[1 + 2 * ns] != 0 || seg[2 + 2 * ns] != 63 || seg[3 + 2 * ns] != 0) return;
            p += L;
            const int64_t rest = n - p;
            int64_t cap_segs = f.dri ? (rest / 2 + 2) : 1;
            if (cap_segs > (1 << 20)) cap_segs = 1 << 20;
            f.words.resize((size_t)(rest / 4 + 5 * cap_segs + 8));
            f.seg_off.resize((size_t)cap_segs);
            f.seg_cnt.resize((size_t)cap_segs);
            f.seg_bytes.resize((size_t)cap_segs);
            int64_t consumed = 0;
            const int64_t k = nopesac_jpeg_prepare_scan(data + p, rest, f.dri ? 1 : 0, f.words.data(), (int64_t)f.words.size(), f.seg_off.data(),
                                                        f.seg_cnt.data(), f.seg_bytes.data(), cap_segs, &consumed);
            if (k < 1) return;
            if (p + consumed + 1 >= n) { f.status = -3; return; }    // no EOI
            f.seg_off.resize((size_t)k);
            f.seg_cnt.resize((size_t)k);
            f.seg_bytes.resize((size_t)k);
            f.words.resize((size_t)(f.seg_off[k - 1] + f.seg_cnt[k - 1]));
            have_scan = true;
            break;
        }
        p += L;
    }
    if (!have_frame || !have_scan) return;
    if (f.ncomp == 1) {
        f.comps[0].h = f.comps[0].v = 1;                         // a one-component scan is never interleaved (T.81 A.2.2)
    } else {
        if (adobe >= 0 && adobe != 1) return;
        if (adobe < 0 && !jfif && f.comps[0].id == 82 && f.comps[1].id == 71 && f.comps[2].id == 66) return;      // RGB-coded file
        const Comp* c = f.comps;
        if (c[1].h != 1 || c[1].v != 1 || c[2].h != 1 || c[2].v != 1) return;
        if (!((c[0].h == 1 && c[0].v == 1) || (c[0].h == 2 && c[0].v == 1) || (c[0].h == 2 && c[0].v == 2))) return;
    }
    for (int i = 0; i < f.ncomp; ++i) {
        const Comp& c = f.comps[i];
        if (c.td < 0 || c.td > 1 || c.ta < 0 || c.ta > 1 || c.tq < 0 || c.tq > 15 || !f.have_qt[c.tq] || !f.have_huff[0][c.td] || !f.have_huff[1][c.ta]) return;
    }
    if (f.width <= 0 || f.height <= 0) return;
    f.hmax = f.comps[0].h;
    f.vmax = f.comps[0].v;
    f.mcux = (f.width + 8 * f.hmax - 1) / (8 * f.hmax);
    f.mcuy = (f.height + 8 * f.vmax - 1) / (8 * f.vmax);
    const int64_t n_mcu = (int64_t)f.mcux * f.mcuy, per = f.dri ? f.dri : n_mcu;
    if ((int64_t)f.seg_off.size() != (n_mcu + per - 1) / per) return;
    for (int i = 0; i < f.ncomp; ++i) {
        Comp& c = f.comps[i];
        c.bw = f.mcux * c.h;
        c.bh = f.mcuy * c.v;
        c.dw = (f.width * c.h + f.hmax - 1) / f.hmax;
        c.dh = (f.height * c.v + f.vmax - 1) / f.vmax;
    }
    f.status = 0;
}

struct Batch {
    std::vector<FileInfo> files;
    int64_t totals[10] = {0};              // n_words, n_segs, n_lanes, n_blocks, max_px, coef_off, plane_off, out_off, (2 spare)
    int sub_words = 0;
    int64_t parallel_min_bytes = 0;
};

}  // namespace

// Phase 1: read and parse the n files on `threads` threads.  status[n]: 0, or why file i keeps the batch off this path (-1 not a JPEG, -2
// outside the supported baseline subset, -3 malformed / truncated, -6 unreadable).  totals[8]: words, restart intervals, lanes of the
// self-synchronising decoder, 8x8 blocks, largest image in pixels, coefficient / plane / output elements of the batch.  Returns an opaque
// batch (free it with nopesac_jpeg_batch_free_host) or NULL on bad arguments; fill the launch arrays with nopesac_jpeg_batch_fill_host
// only if every status is 0.  parallel = 0: no file goes to the self-synchronising decoder.
extern "C" void* nopesac_jpeg_batch_scan_host(const char* const* paths, int n, int threads, int parallel, int* status, int64_t* totals) {
    if (!paths || n <= 0 || !status || !totals) return nullptr;
    Batch* B = new Batch();
    B->files.resize((size_t)n);
    B->sub_words = NOPESAC_JPEG_SUB_WORDS;
    B->parallel_min_bytes = 4ll * NOPESAC_JPEG_SUB_WORDS * 4;
    std::atomic<int> next(0);
    auto work = [&]() {
        std::vector<uint8_t> buf;
        while (true) {
            const int i = next.fetch_add(1);
            if (i >= n) break;
            FileInfo& f = B->files[(size_t)i];
            f.status = -6;
            FILE* fp = paths[i] ? fopen(paths[i], "rb") : nullptr;
            if (!fp) continue;
            if (fseek(fp, 0, SEEK_END) == 0) {
                const long sz = ftell(fp);
                if (sz > 0 && fseek(fp, 0, SEEK_SET) == 0) {
                    buf.resize((size_t)sz);
                    if (fread(buf.data(), 1, (size_t)sz, fp) == (size_t)sz) parse_file(buf, f);
                }
            }
            fclose(fp);
        }
    };
    const int T = threads < 1 ? 1 : (threads > n ? n : threads);
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    int64_t n_words = 0, n_segs = 0, n_lanes = 0, n_blocks = 0, max_px = 0, coef = 0, plane = 0, out = 0;
    for (int i = 0; i < n; ++i) {
        const FileInfo& f = B->files[(size_t)i];
        status[i] = f.status;
        if (f.status != 0) continue;
        n_words += (int64_t)f.words.size();
        n_segs += (int64_t)f.seg_off.size();
        for (int c = 0; c < f.ncomp; ++c) {
            const int64_t nb = (int64_t)f.comps[c].bw * f.comps[c].bh;
            n_blocks += nb;
            coef += nb * 64;
            plane += nb * 64;
        }
        const int64_t px = (int64_t)f.width * f.height;
        if (px > max_px) max_px = px;
        out += (px * 3 + 15) / 16 * 16;
        if (parallel && !f.dri && f.seg_bytes[0] >= B->parallel_min_bytes) {
            const int64_t nsub = (f.seg_bytes[0] * 8 + (int64_t)B->sub_words * 32 - 1) / ((int64_t)B->sub_words * 32);
            n_lanes += (nsub + 63) / 64 * 64;
        }
    }
    const int64_t t[8] = {n_words, n_segs, n_lanes, n_blocks, max_px, coef, plane, out};
    memcpy(totals, t, sizeof(t));
    memcpy(B->totals, t, sizeof(t));
    B->totals[8] = parallel ? 1 : 0;
    return B;
}

// Phase 2: the launch arrays of the batch (layouts: include/nopesac_hip.h NOPESAC_JPEG_*; jpeg.py prepare_batch): img32 [n][32], img64
// [n][8], tables [n][TABLES_BYTES], seg32 [n_segs][4], seg64 [n_segs][2], words [n_words], lane_img [n_lanes] (may be NULL when n_lanes = 0),
// geometry [n][2] = (height, width).  All zero-initialised by the callee.  Returns 0, -1 on bad arguments / a file with a non-zero status,
// -2 on a bad Huffman table.
extern "C" int nopesac_jpeg_batch_fill_host(void* batch, int32_t* img32, int64_t* img64, uint8_t* tables, int32_t* seg32, int64_t* seg64, uint32_t* words,
                                            int32_t* lane_img, int32_t* geometry) {
    Batch* B = (Batch*)batch;
    if (!B || !img32 || !img64 || !tables || !seg32 || !seg64 || !words || !geometry || (B->totals[2] > 0 && !lane_img)) return -1;
    const int n = (int)B->files.size();
    const bool parallel = B->totals[8] != 0;
    memset(img32, 0, (size_t)n * I32 * 4);
    memset(img64, 0, (size_t)n * I64 * 8);
    memset(tables, 0, (size_t)n * TABLES_BYTES);
    int64_t coef_off = 0, plane_off = 0, out_off = 0, word_off = 0, n_lanes = 0, n_blocks = 0, seg = 0;
    for (int i = 0; i < n; ++i) {
        const FileInfo& f = B->files[(size_t)i];
        if (f.status != 0) return -1;
        int32_t* a = img32 + (size_t)i * I32;
        int64_t* a8 = img64 + (size_t)i * I64;
        const int64_t n_mcu = (int64_t)f.mcux * f.mcuy, per = f.dri ? f.dri : n_mcu;
        a[0] = f.width; a[1] = f.height; a[2] = f.ncomp; a[3] = f.hmax; a[4] = f.vmax; a[5] = f.mcux; a[6] = f.mcuy; a[7] = (int32_t)per;
        int64_t nb = 0;
        for (int ci = 0; ci < f.ncomp; ++ci) {
            const Comp& c = f.comps[ci];
            a[8 + ci] = c.bw; a[11 + ci] = c.bh; a[14 + ci] = c.dw; a[17 + ci] = c.dh;
            a[20 + ci] = c.td; a[23 + ci] = 2 + c.ta;
            a8[ci] = coef_off;
            a8[3 + ci] = plane_off;
            coef_off += (int64_t)c.bw * c.bh * 64;
            plane_off += (int64_t)c.bw * c.bh * 64;
            nb += (int64_t)c.bw * c.bh;
            memcpy(tables + (size_t)i * TABLES_BYTES + 4 * HUFF_BYTES + 128 * ci, f.qt[c.tq], 128);
        }
        a[26] = (int32_t)n_blocks;
        a[27] = (int32_t)nb;
        n_blocks += nb;
        a8[6] = out_off;
        out_off += ((int64_t)f.width * f.height * 3 + 15) / 16 * 16;
        for (int tc = 0; tc < 2; ++tc)
            for (int th = 0; th < 2; ++th)
                if (f.have_huff[tc][th] && !derived_table(f.huff[tc][th], tables + (size_t)i * TABLES_BYTES + (size_t)(2 * tc + th) * HUFF_BYTES)) return -2;
        a8[7] = word_off;
        if (parallel && !f.dri && f.seg_bytes[0] >= B->parallel_min_bytes) {
            const int64_t nsub = (f.seg_bytes[0] * 8 + (int64_t)B->sub_words * 32 - 1) / ((int64_t)B->sub_words * 32);
            a[28] = (int32_t)n_lanes;
            a[29] = (int32_t)nsub;
            const int64_t lanes = (nsub + 63) / 64 * 64;
            for (int64_t l = 0; l < lanes; ++l) lane_img[n_lanes + l] = i;
            n_lanes += lanes;
        }
        for (size_t k = 0; k < f.seg_off.size(); ++k, ++seg) {
            const int64_t first = (int64_t)k * per, left = n_mcu - first;
            seg32[4 * seg] = i; seg32[4 * seg + 1] = (int32_t)first; seg32[4 * seg + 2] = (int32_t)(left < per ? left : per); seg32[4 * seg + 3] = 0;
            seg64[2 * seg] = word_off + f.seg_off[k];
            seg64[2 * seg + 1] = f.seg_cnt[k];
        }
        memcpy(words + word_off, f.words.data(), f.words.size() * 4);
        word_off += (int64_t)f.words.size();
        geometry[2 * i] = f.height;
        geometry[2 * i + 1] = f.width;
    }
    return 0;
}

extern "C" void nopesac_jpeg_batch_free_host(void* batch) { delete (Batch*)batch; }
