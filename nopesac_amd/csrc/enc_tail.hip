// Tail of one post-norm transformer encoder layer of the PlaneTR head in ONE launch (bf16 mode;
// transformer/transformer.py:183-199 TransformerEncoderLayer.forward_post):
//
//     s   = src + out_proj(attn)              attn = multi-head attention output (separate kernel, bf16)
//     y1  = LayerNorm1(s)
//     y2  = LayerNorm2(y1 + linear2(relu(linear1(y1))))
//     outputs: y2 (f32 residual stream), bf16(y2) and bf16(y2 + pos) (the next layer's GEMM operands)
//
// Un-fused this is 5 launches per layer (out-proj GEMM, LN, two FFN GEMMs, LN) and the 1024-wide hidden tensor makes a
// round trip through HBM.  Here a workgroup (8 waves) owns 32 consecutive tokens and the full width: the attention rows
// sit in LDS as the bf16 A tile, every GEMM is "LDS tile x fragment-major weights streamed from L2" (pwchain.hip /
// gnn_layer.hip), LayerNorm reduces across the waves through LDS, y1 stays in registers (f32) for the second residual
// and in LDS (bf16) as the FFN operand, the 32 x 1024 hidden tile never leaves the CU.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace nps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

constexpr int ET_D = 256, ET_FF = 1024, ET_BM = 32;
constexpr int ET_LD = ET_D + 8, ET_HLD = ET_FF + 8;
constexpr int ET_A = ET_BM * ET_LD;                      // elements of one [32][264] bf16 tile
// f32 output staging rows are ET_FLD floats apart: a 4-dword skew, so the eight rows one ds_write_b128 lane group touches fall on
// different banks (at 256 floats all eight hit the same four: 8-way conflicts, profiles/r3_pmc_enc_tail.json)
constexpr int ET_FLD = ET_D + 4;
constexpr size_t ET_STAGE_BYTES = 2 * (size_t)ET_BM * ET_FLD * 4 > 2 * (size_t)ET_BM * ET_HLD ? 2 * (size_t)ET_BM * ET_FLD * 4 : 2 * (size_t)ET_BM * ET_HLD;
constexpr size_t ET_LDS_BYTES = 2 * (size_t)(2 * ET_A) + ET_STAGE_BYTES + 2 * 8 * 32 * sizeof(float);

constexpr int ET_PF_MAX = 8;
struct EncTailArgs {
    const bf16_t* attn; const float* src;                 // [M][256]
    const bf16_t* wo; const float* bo; const float* g1; const float* be1;
    const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2; const float* g2; const float* be2;
    const float* pos; int pos_rows;                       // [pos_rows][256], row index = token % pos_rows (may be null)
    float* y; bf16_t* y16; bf16_t* ypos16;                // outputs [M][256] (each nullable)
    float* yn;                                            // pre-norm only: f32 copy of the normalised result (nullable)
    int M, pre_norm;
    // chained projections of the NEXT attention (nullable): pa = bf16((n + pos) Wpa^T + bpa) [M][npa], pb = bf16(n Wpb^T + bpb) [M][npb]
    // with n = the normalised result; fragment-major weights, npa / npb multiples of 32
    const bf16_t* wpa; const float* bpa; bf16_t* pa; int npa;
    const bf16_t* wpb; const float* bpb; bf16_t* pb; int npb;
    int skip_ffn;                                         // decoder self-attention half: s = src + out_proj(attn), n = LN(s), no FFN
    // weight prefetch for the NEXT launch (few workgroups - one pair per call: a tail's 0.4-1.5 MB of weights are cold in L2 and arrive
    // through each workgroup's own 128 KB of loads in flight): workgroups >= n_work do no layer work; those on an XCD the next launch
    // will run on read a slice of the listed byte ranges into that XCD's L2 and exit (as gnn_layer.hip does)
    const unsigned char* pf[ET_PF_MAX]; int pf_bytes[ET_PF_MAX]; int n_pf, n_work, pf_xcds;
};
constexpr int ET_PF_CHUNK = 512 * 16, ET_PF_PER_XCD = 16;

// prefetch workgroup of a tail launch (512 threads): its XCD = id & 7 (workgroup ids are dealt round-robin over the XCDs)
__device__ __forceinline__ void et_prefetch(const EncTailArgs& p) {
    const int xcd = blockIdx.x & 7, slot = ((int)blockIdx.x - p.n_work) >> 3;
    if (xcd >= p.pf_xcds || slot >= ET_PF_PER_XCD) return;
    unsigned acc = 0u;
    int c = slot;                                          // chunk counter over all ranges
    for (int s = 0; s < p.n_pf; ++s) {
        const int nch = (p.pf_bytes[s] + ET_PF_CHUNK - 1) / ET_PF_CHUNK;
        for (; c < nch; c += ET_PF_PER_XCD) {
            const int off = c * ET_PF_CHUNK + (int)threadIdx.x * 16;
            if (off + 16 <= p.pf_bytes[s]) {
                const uint4 v = *reinterpret_cast<const uint4*>(p.pf[s] + off);      // an ordinary load: it must allocate in L2
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
        c -= nch;
    }
    asm volatile("" ::"v"(acc));
}

struct EtRing {
    bf16x8 f[2][16];
};
__device__ __forceinline__ void et_issue(EtRing& ring, int buf, const bf16_t* __restrict__ w, int kf_total, int kf_off, int nt, int lane) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
        ring.f[buf][kk] = *reinterpret_cast<const bf16x8*>(w + ((long long)(nt * kf_total + kf_off + kk) * 64 + lane) * 8);
}
// acc += A[row][k] * W[tile][k] over the 16 k-steps held in ring.f[buf]; lane holds token l&31 x 4-channel runs
__device__ __forceinline__ void et_gemm(const EtRing& ring, int buf, const bf16_t* A, int lda, f32x16& acc, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(A + l31 * lda + kk * 16 + half * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.f[buf][kk], af, acc, 0, 0, 0);
    }
}
__device__ __forceinline__ void et_zero(f32x16& acc) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
}
// two-pass LayerNorm over 256 channels spread over the 8 waves (32 each); normalised + affine values stay in acc
__device__ __forceinline__ void et_layernorm(f32x16& acc, const float* __restrict__ gamma, const float* __restrict__ beta, float* red,
                                             int wave, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
    float mean = 0.f, rstd = 0.f;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float* rp = red + pass * 8 * 32;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float d = pass == 0 ? acc[e] : acc[e] - mean;
            s += pass == 0 ? d : d * d;
        }
        s += __shfl_xor(s, 32, 64);
        if (half == 0) rp[wave * 32 + l31] = s;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += rp[w * 32 + l31];
        if (pass == 0) mean = t / ET_D;
        else rstd = rsqrtf(t / ET_D + 1e-5f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + n), b = *reinterpret_cast<const f32x4*>(beta + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * q + e] = (acc[4 * q + e] - mean) * rstd * g[e] + b[e];
    }
}

__global__ __launch_bounds__(512, 2) void enc_tail_kernel(const EncTailArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char et_smem[];
    bf16_t* At = reinterpret_cast<bf16_t*>(et_smem);         // attention rows [32][264]
    bf16_t* Yt = At + ET_A;                                  // bf16(y1) [32][264]; later bf16(y2)
    bf16_t* Ht = Yt + ET_A;                                  // hidden [32][1032]; later the f32 / bf16 output staging
    float* red = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(Ht) + ET_STAGE_BYTES);
    if ((int)blockIdx.x >= p.n_work) { et_prefetch(p); return; }
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * ET_BM;
    const long long row = m0 + l31;
    const bool row_ok = row < p.M;
    EtRing ring;

    et_issue(ring, 0, p.wo, 16, 0, wave, lane);                                   // step 0: out-proj tile `wave`
    // (round 6: every load of this prologue is UNCONDITIONAL - rows behind M read the last row and are zeroed / never stored.  Guarded
    // loads compile to one branch + one full memory round trip each: the attention rows, the residual rows and the position rows of
    // this 24 us latency-chain kernel were seven of them in a row)
    const long long lastrow = (long long)p.M - 1;
    us8 av[ET_BM * 32 / 512];
#pragma unroll
    for (int i = 0; i < ET_BM * 32 / 512; ++i) {                                  // attention rows -> LDS
        const int c = tid + i * 512, r = c >> 5, col = (c & 31) * 8;
        av[i] = *reinterpret_cast<const us8*>(p.attn + (m0 + r < lastrow ? m0 + r : lastrow) * ET_D + col);
    }
    f32x4 srcv[4];                                                                // the residual rows: in flight under the out-projection
#pragma unroll
    for (int q = 0; q < 4; ++q)
        srcv[q] = *reinterpret_cast<const f32x4*>(p.src + (row_ok ? row : lastrow) * ET_D + wave * 32 + 8 * q + 4 * half);
#pragma unroll
    for (int i = 0; i < ET_BM * 32 / 512; ++i) {
        const int c = tid + i * 512, r = c >> 5, col = (c & 31) * 8;
        *reinterpret_cast<us8*>(At + r * ET_LD + col) = m0 + r < p.M ? av[i] : us8{};
    }
    __syncthreads();

    // ---- y1 = LN1(src + out_proj(attn)); wave owns channels wave*32 .. +32
    if (!p.skip_ffn) et_issue(ring, 1, p.w1, 16, 0, 4 * wave, lane);              // step 1: linear1 tile 4w
    f32x16 y1;
    et_zero(y1);
    et_gemm(ring, 0, At, ET_LD, y1, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bo + n);
        const f32x4 s = row_ok ? srcv[q] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) y1[4 * q + e] = (y1[4 * q + e] + b[e]) + s[e];
    }
    // post-norm (encoder): the normalised y1 is both the FFN input and the second residual;
    // pre-norm (decoder): the FFN reads LN(s) while the residual stays s
    f32x16 resid = y1;
    et_layernorm(y1, p.g1, p.be1, red, wave, lane);
    if (!p.pre_norm) resid = y1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        us4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(y1[4 * q + e]);
        *reinterpret_cast<us4*>(Yt + l31 * ET_LD + wave * 32 + 8 * q + 4 * half) = o;
    }
    __syncthreads();

    f32x16 y2;
    f32x16 u;
    if (p.skip_ffn) {                                         // (wave-uniform) the residual stream leaves as s, its norm feeds the projections
        u = resid;
        y2 = y1;
    } else {
    // ---- hidden = relu(linear1(y1)): tiles 4w .. 4w+3 of 32
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int buf = (1 + j) & 1;
        if (j < 3) et_issue(ring, buf ^ 1, p.w1, 16, 0, 4 * wave + j + 1, lane);  // steps 2..4
        else et_issue(ring, buf ^ 1, p.w2, 64, 0, wave, lane);                    // step 5: linear2, hidden 0..255
        f32x16 h;
        et_zero(h);
        et_gemm(ring, buf, Yt, ET_LD, h, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = (4 * wave + j) * 32 + 8 * q + 4 * half;
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1 + n);
            us4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = h[4 * q + e] + b[e];
                o[e] = f32_to_bf16(v > 0.f ? v : 0.f);
            }
            *reinterpret_cast<us4*>(Ht + l31 * ET_HLD + n) = o;
        }
    }
    __syncthreads();

    // ---- y2 = LN2(y1 + linear2(hidden)): K = 1024 in four ring steps
    et_zero(y2);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int buf = (5 + c) & 1;
        if (c < 3) et_issue(ring, buf ^ 1, p.w2, 64, 16 * (c + 1), wave, lane);   // steps 6..8
        et_gemm(ring, buf, Ht + 256 * c, ET_HLD, y2, lane);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) y2[4 * q + e] = (y2[4 * q + e] + b[e]) + resid[4 * q + e];
    }
    u = y2;                                                   // pre-norm: the residual stream leaves un-normalised
    et_layernorm(y2, p.g2, p.be2, red, wave, lane);           // its barriers also retire every read of Ht / Yt
    }

    // ---- outputs through LDS so that they leave as whole rows: f32 tile(s) in the hidden region, bf16 tiles in At / Yt
    float* Yf = reinterpret_cast<float*>(Ht);                 // [32][256] f32: the residual-stream output
    float* Yn = Yf + ET_BM * ET_FLD;                           // [32][256] f32: normalised copy (pre-norm, optional)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        f32x4 v = {y2[4 * q], y2[4 * q + 1], y2[4 * q + 2], y2[4 * q + 3]};
        if (p.pre_norm) {
            const f32x4 uv = {u[4 * q], u[4 * q + 1], u[4 * q + 2], u[4 * q + 3]};
            *reinterpret_cast<f32x4*>(Yf + l31 * ET_FLD + n) = uv;
            *reinterpret_cast<f32x4*>(Yn + l31 * ET_FLD + n) = v;
        } else {
            *reinterpret_cast<f32x4*>(Yf + l31 * ET_FLD + n) = v;
        }
        us4 o, op;
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (p.pos) {                                          // (wave-uniform condition; rows behind M read a valid row and are never stored)
            const f32x4 pl = *reinterpret_cast<const f32x4*>(p.pos + ((row_ok ? row : lastrow) % p.pos_rows) * ET_D + n);
            pv = row_ok ? pl : pv;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = f32_to_bf16(v[e]); op[e] = f32_to_bf16(v[e] + pv[e]); }
        *reinterpret_cast<us4*>(Yt + l31 * ET_LD + n) = o;
        *reinterpret_cast<us4*>(At + l31 * ET_LD + n) = op;
    }
    __syncthreads();
    if (p.y) {
#pragma unroll
        for (int i = 0; i < ET_BM * 64 / 512; ++i) {          // 64 16-byte chunks per f32 row
            const int c = tid + i * 512, r = c >> 6, col = (c & 63) * 4;
            if (m0 + r < p.M) *reinterpret_cast<f32x4*>(p.y + (m0 + r) * ET_D + col) = *reinterpret_cast<const f32x4*>(Yf + r * ET_FLD + col);
        }
    }
    if (p.yn && p.pre_norm) {
#pragma unroll
        for (int i = 0; i < ET_BM * 64 / 512; ++i) {
            const int c = tid + i * 512, r = c >> 6, col = (c & 63) * 4;
            if (m0 + r < p.M) *reinterpret_cast<f32x4*>(p.yn + (m0 + r) * ET_D + col) = *reinterpret_cast<const f32x4*>(Yn + r * ET_FLD + col);
        }
    }
#pragma unroll
    for (int i = 0; i < ET_BM * 32 / 512; ++i) {
        const int c = tid + i * 512, r = c >> 5, col = (c & 31) * 8;
        if (m0 + r < p.M) {
            if (p.y16) *reinterpret_cast<us8*>(p.y16 + (m0 + r) * ET_D + col) = *reinterpret_cast<const us8*>(Yt + r * ET_LD + col);
            if (p.ypos16) *reinterpret_cast<us8*>(p.ypos16 + (m0 + r) * ET_D + col) = *reinterpret_cast<const us8*>(At + r * ET_LD + col);
        }
    }
    // ---- the next attention's input projections, straight from the two bf16 tiles that are still in LDS (At = n + pos, Yt = n):
    //      column tile nt = round * 8 + wave of [Wpa ; Wpb]; K = 256 = one ring step per tile
    const int ta = p.wpa ? p.npa / 32 : 0, tb = p.wpb ? p.npb / 32 : 0, tt = ta + tb;     // (at most 32 tiles: four rounds)
    if (tt > 0) {
        auto issue_tile = [&](auto BUF, int nt) {
            constexpr int buf = decltype(BUF)::value;
            if (nt < ta) et_issue(ring, buf, p.wpa, 16, 0, nt, lane);
            else et_issue(ring, buf, p.wpb, 16, 0, nt - ta, lane);
        };
        auto do_tile = [&](auto BUF, int nt) {
            constexpr int buf = decltype(BUF)::value;
            const bool is_a = nt < ta;
            const int ct = is_a ? nt : nt - ta, ldo = is_a ? p.npa : p.npb;
            f32x16 acc;
            et_zero(acc);
            et_gemm(ring, buf, is_a ? At : Yt, ET_LD, acc, lane);
            const float* bias = is_a ? p.bpa : p.bpb;
            bf16_t* outp = (is_a ? p.pa : p.pb) + row * ldo + ct * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 8 * q + 4 * half;
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (bias) b = *reinterpret_cast<const f32x4*>(bias + ct * 32 + n);
                us4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(acc[4 * q + e] + b[e]);
                if (row_ok) *reinterpret_cast<us4*>(outp + n) = o;
            }
        };
        typedef std::integral_constant<int, 0> B0;
        typedef std::integral_constant<int, 1> B1;
        if (wave < tt) issue_tile(B0{}, wave);
        if (wave + 8 < tt) issue_tile(B1{}, wave + 8);
        if (wave < tt) do_tile(B0{}, wave);
        if (wave + 16 < tt) issue_tile(B0{}, wave + 16);
        if (wave + 8 < tt) do_tile(B1{}, wave + 8);
        if (wave + 24 < tt) issue_tile(B1{}, wave + 24);
        if (wave + 16 < tt) do_tile(B0{}, wave + 16);
        if (wave + 24 < tt) do_tile(B1{}, wave + 24);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 64-token form of the ENCODER tail (post-norm, with the optional chained projections).  The 32-token kernel above is bound by the
// latency of its weight steps - every step a wave multiplies 16 fragments it waited ~1.5 us for - and its 167-250 VGPRs allow one
// workgroup per CU: 600 workgroups x 30 us for the 19200 encoder tokens.  Here a workgroup owns 64 tokens and every fragment feeds TWO
// MFMAs (row tiles 0 and 1), so the same latency chain covers twice the tokens: half the workgroups, each a little longer.  What
// makes it fit: weight steps of K = 128 (8 fragments x 2 buffers = 64 VGPRs instead of 128) and the 1024-wide hidden tile in two
// halves of 512 (66 KB of LDS), linear2 accumulating across the halves in registers.  Same arithmetic in the same order as the
// 32-token kernel (ascending K inside every accumulator, the same LayerNorm reduction order): bit-identical results.
constexpr int E6_BM = 64, E6_HW = 512, E6_HLD = E6_HW + 8;
constexpr int E6_A = E6_BM * ET_LD;
constexpr size_t E6_LDS_BYTES = 2 * (size_t)(2 * E6_A + E6_BM * E6_HLD) + 2 * 8 * 64 * sizeof(float);
static_assert((size_t)E6_BM * ET_FLD * 4 <= 2 * (size_t)E6_BM * E6_HLD, "the f32 staging tile must fit the hidden region");

struct E6Ring {
    bf16x8 f[2][8];
};
template <int BUF>
__device__ __forceinline__ void e6_issue(E6Ring& ring, const bf16_t* __restrict__ w, int kf_total, int kf_off, int nt, int lane) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
        ring.f[BUF][kk] = *reinterpret_cast<const bf16x8*>(w + ((long long)(nt * kf_total + kf_off + kk) * 64 + lane) * 8);
}
// acc[r] += A[r*32 + row][koff*16 ..] * W over the 8 k-steps held in ring.f[BUF]
template <int BUF>
__device__ __forceinline__ void e6_gemm(const E6Ring& ring, const bf16_t* A, int lda, int koff, f32x16 (&acc)[2], int lane) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(A + (r * 32 + l31) * lda + (koff + kk) * 16 + half * 8);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.f[BUF][kk], af, acc[r], 0, 0, 0);
        }
}
__device__ __forceinline__ void e6_layernorm(f32x16 (&acc)[2], const float* __restrict__ gamma, const float* __restrict__ beta, float* red,
                                             int wave, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
    float mean[2], rstd[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float* rp = red + pass * 8 * 64;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = pass == 0 ? acc[r][e] : acc[r][e] - mean[r];
                s += pass == 0 ? d : d * d;
            }
            s += __shfl_xor(s, 32, 64);
            if (half == 0) rp[wave * 64 + r * 32 + l31] = s;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += rp[w * 64 + r * 32 + l31];
            if (pass == 0) mean[r] = t / ET_D;
            else rstd[r] = rsqrtf(t / ET_D + 1e-5f);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + n), b = *reinterpret_cast<const f32x4*>(beta + n);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][4 * q + e] = (acc[r][4 * q + e] - mean[r]) * rstd[r] * g[e] + b[e];
    }
}

__global__ __launch_bounds__(512, 1) void enc_tail64_kernel(const EncTailArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char et_smem[];
    bf16_t* At = reinterpret_cast<bf16_t*>(et_smem);         // attention rows [64][264]; later bf16(y + pos)
    bf16_t* Yt = At + E6_A;                                  // bf16(y1) [64][264]; later bf16(y)
    bf16_t* Ht = Yt + E6_A;                                  // hidden half [64][520]; later the f32 output staging
    float* red = reinterpret_cast<float*>(Ht + E6_BM * E6_HLD);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * E6_BM;
    E6Ring ring;
    typedef std::integral_constant<int, 0> B0;
    (void)sizeof(B0);

    e6_issue<0>(ring, p.wo, 16, 0, wave, lane);                                   // out-proj tile `wave`, K 0..127
#pragma unroll
    for (int i = 0; i < E6_BM * 32 / 512; ++i) {                                  // attention rows -> LDS
        const int c = tid + i * 512, r = c >> 5, col = (c & 31) * 8;
        us8 v = us8{};
        if (m0 + r < p.M) v = *reinterpret_cast<const us8*>(p.attn + (m0 + r) * ET_D + col);
        *reinterpret_cast<us8*>(At + r * ET_LD + col) = v;
    }
    __syncthreads();
    // ---- y1 = LN1(src + out_proj(attn))
    f32x16 y1[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) et_zero(y1[r]);
    e6_issue<1>(ring, p.wo, 16, 8, wave, lane);
    e6_gemm<0>(ring, At, ET_LD, 0, y1, lane);
    e6_issue<0>(ring, p.w1, 16, 0, 2 * wave, lane);                               // linear1, half 0, tile 2w, K 0..127
    e6_gemm<1>(ring, At, ET_LD, 8, y1, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bo + n);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const long long row = m0 + r * 32 + l31;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (row < p.M) s = *reinterpret_cast<const f32x4*>(p.src + row * ET_D + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) y1[r][4 * q + e] = (y1[r][4 * q + e] + b[e]) + s[e];
        }
    }
    e6_layernorm(y1, p.g1, p.be1, red, wave, lane);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            us4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(y1[r][4 * q + e]);
            *reinterpret_cast<us4*>(Yt + (r * 32 + l31) * ET_LD + wave * 32 + 8 * q + 4 * half) = o;
        }
    __syncthreads();
    // ---- FFN in two halves of 512 hidden channels: hidden_h = relu(linear1 tiles 16h .. 16h+15), y2 += hidden_h W2[:, 512h ..]
    f32x16 y2[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) et_zero(y2[r]);
    auto hidden_tile = [&](auto H, auto J) {                 // on entry ring buffer 0 holds K 0..127 of tile 16h + 2w + j
        constexpr int h = decltype(H)::value, j = decltype(J)::value;
        const int nt = 16 * h + 2 * wave + j;
        f32x16 hd[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) et_zero(hd[r]);
        e6_issue<1>(ring, p.w1, 16, 8, nt, lane);
        e6_gemm<0>(ring, Yt, ET_LD, 0, hd, lane);
        if constexpr (j == 0) e6_issue<0>(ring, p.w1, 16, 0, nt + 1, lane);      // the wave's second tile of this half
        else e6_issue<0>(ring, p.w2, 64, 32 * h, wave, lane);                     // linear2, K 512h .. +127
        e6_gemm<1>(ring, Yt, ET_LD, 8, hd, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nl = (2 * wave + j) * 32 + 8 * q + 4 * half;
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.b1 + 512 * h + nl);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                us4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = hd[r][4 * q + e] + b[e];
                    o[e] = f32_to_bf16(v > 0.f ? v : 0.f);
                }
                *reinterpret_cast<us4*>(Ht + (r * 32 + l31) * E6_HLD + nl) = o;
            }
        }
    };
    auto linear2_half = [&](auto H) {                         // on entry ring buffer 0 holds K 512h .. +127 of tile `wave`
        constexpr int h = decltype(H)::value;
        e6_issue<1>(ring, p.w2, 64, 32 * h + 8, wave, lane);
        e6_gemm<0>(ring, Ht, E6_HLD, 0, y2, lane);
        e6_issue<0>(ring, p.w2, 64, 32 * h + 16, wave, lane);
        e6_gemm<1>(ring, Ht, E6_HLD, 8, y2, lane);
        e6_issue<1>(ring, p.w2, 64, 32 * h + 24, wave, lane);
        e6_gemm<0>(ring, Ht, E6_HLD, 16, y2, lane);
        if constexpr (h == 0) e6_issue<0>(ring, p.w1, 16, 0, 16 + 2 * wave, lane);          // half 1, first tile
        e6_gemm<1>(ring, Ht, E6_HLD, 24, y2, lane);
    };
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 0> I0;
    hidden_tile(I0{}, I0{});
    hidden_tile(I0{}, I1{});
    __syncthreads();
    linear2_half(I0{});
    __syncthreads();                                          // every wave is done reading hidden half 0
    hidden_tile(I1{}, I0{});
    hidden_tile(I1{}, I1{});
    __syncthreads();
    // the first projection tile (if any) while linear2's second half runs
    const int ta = p.wpa ? p.npa / 32 : 0, tb = p.wpb ? p.npb / 32 : 0, tt = ta + tb;
    auto tile_w = [&](int nt) { return nt < ta ? p.wpa : p.wpb; };
    linear2_half(I1{});
    if (wave < tt) {                                          // first projection tile: in flight during LN2 and the output staging
        e6_issue<0>(ring, tile_w(wave), 16, 0, wave < ta ? wave : wave - ta, lane);
        e6_issue<1>(ring, tile_w(wave), 16, 8, wave < ta ? wave : wave - ta, lane);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + n);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) y2[r][4 * q + e] = (y2[r][4 * q + e] + b[e]) + y1[r][4 * q + e];
    }
    e6_layernorm(y2, p.g2, p.be2, red, wave, lane);           // its barriers also retire every read of Ht / Yt
    // ---- outputs through LDS (whole rows): f32 tile in the hidden region, bf16 tiles in Yt (y) / At (y + pos)
    float* Yf = reinterpret_cast<float*>(Ht);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = wave * 32 + 8 * q + 4 * half, rl = r * 32 + l31;
            const long long row = m0 + rl;
            const f32x4 v = {y2[r][4 * q], y2[r][4 * q + 1], y2[r][4 * q + 2], y2[r][4 * q + 3]};
            *reinterpret_cast<f32x4*>(Yf + rl * ET_FLD + n) = v;
            f32x4 pv = {0.f, 0.f, 0.f, 0.f};
            if (p.pos && row < p.M) pv = *reinterpret_cast<const f32x4*>(p.pos + (row % p.pos_rows) * ET_D + n);
            us4 o, op;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = f32_to_bf16(v[e]); op[e] = f32_to_bf16(v[e] + pv[e]); }
            *reinterpret_cast<us4*>(Yt + rl * ET_LD + n) = o;
            *reinterpret_cast<us4*>(At + rl * ET_LD + n) = op;
        }
    __syncthreads();
    if (p.y) {
#pragma unroll
        for (int i = 0; i < E6_BM * 64 / 512; ++i) {
            const int c = tid + i * 512, r = c >> 6, col = (c & 63) * 4;
            if (m0 + r < p.M) *reinterpret_cast<f32x4*>(p.y + (m0 + r) * ET_D + col) = *reinterpret_cast<const f32x4*>(Yf + r * ET_FLD + col);
        }
    }
#pragma unroll
    for (int i = 0; i < E6_BM * 32 / 512; ++i) {
        const int c = tid + i * 512, r = c >> 5, col = (c & 31) * 8;
        if (m0 + r < p.M) {
            if (p.y16) *reinterpret_cast<us8*>(p.y16 + (m0 + r) * ET_D + col) = *reinterpret_cast<const us8*>(Yt + r * ET_LD + col);
            if (p.ypos16) *reinterpret_cast<us8*>(p.ypos16 + (m0 + r) * ET_D + col) = *reinterpret_cast<const us8*>(At + r * ET_LD + col);
        }
    }
    // ---- the next attention's input projections from the two bf16 tiles (At = y + pos, Yt = y): tile nt = round * 8 + wave, K = 256
    if (tt > 0) {
        for (int nt = wave; nt < tt; nt += 8) {
            const bool is_a = nt < ta;
            const int ct = is_a ? nt : nt - ta, ldo = is_a ? p.npa : p.npb;
            f32x16 acc[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) et_zero(acc[r]);
            const int nn = nt + 8, cn = nn < ta ? nn : nn - ta;       // the next tile's fragments replace this one's as they are consumed
            e6_gemm<0>(ring, is_a ? At : Yt, ET_LD, 0, acc, lane);
            if (nn < tt) e6_issue<0>(ring, tile_w(nn), 16, 0, cn, lane);
            e6_gemm<1>(ring, is_a ? At : Yt, ET_LD, 8, acc, lane);
            if (nn < tt) e6_issue<1>(ring, tile_w(nn), 16, 8, cn, lane);
            const float* bias = is_a ? p.bpa : p.bpb;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 8 * q + 4 * half;
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (bias) b = *reinterpret_cast<const f32x4*>(bias + ct * 32 + n);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const long long row = m0 + r * 32 + l31;
                    us4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(acc[r][4 * q + e] + b[e]);
                    if (row < p.M) *reinterpret_cast<us4*>((is_a ? p.pa : p.pb) + row * ldo + ct * 32 + n) = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 128-token form of the ENCODER tail (round 6).  Why: in the 64-token kernel every 1 KB weight fragment feeds TWO MFMAs, i.e. 16 cycles of
// the CU's vector-memory path (64 B/clk) per 16 cycles of CU-level MFMA time (32 cycles on one of four SIMDs): the weight stream from L2
// and the MFMAs need the SAME time, each wave waits for its own eight fragments with one K = 128 step of look-ahead, and the kernel sat
// at 13 % MFMA-busy / 56 % of the wave cycles parked in s_waitcnt (profiles/r4_pmc_enc_tail.json).  Here a workgroup owns 128 tokens and
// every fragment feeds FOUR MFMAs (row tiles 0..3): half the L2 -> CU bytes per FLOP (1.5 MB per 128 instead of per 64 tokens), 32
// MFMAs (1024 SIMD cycles) of cover behind every weight step, 150 workgroups for the 19200 encoder tokens in ONE round instead of 300
// in two.  What makes it fit (256 registers per wave, 160 KB of LDS):
//   * the residual input is the accumulator's start value (acc = src + b_o, loaded straight into the accumulator registers while the
//     attention rows and the first weight step are in flight), and linear2 accumulates on top of y1 + b2 in the registers that held y1 -
//     no second copy of y1, no separate residual load behind the GEMM.  Sums: (src + b_o) + sum_k instead of (sum_k + b_o) + src, and
//     (y1 + b2) + sum_k instead of (sum_k + b2) + y1: f32 roundings in a different order, not bit-identical to the 32- / 64-token kernels;
//   * the 1024-wide hidden tile in four quarters of 256 = ONE column tile per wave and quarter, parked in the region that held the
//     attention rows; linear1 of quarter q + 1 is computed into registers BEFORE the barrier that frees the hidden tile, so that barrier
//     has a full linear2 + linear1 quarter (128 MFMAs per wave) between its arrivals;
//   * the f32 residual stream leaves straight from the accumulators (four 32-byte pieces of a row per wave = one 128-byte line), only
//     the bf16 operand tiles of the chained projections go back through LDS.
// NR = row tiles of 32 tokens per workgroup: 4 (128 tokens, the form described above) or 3 (96 tokens: 200 workgroups for the 19200
// encoder tokens - more CUs busy, three MFMAs per fragment)
template <int NR> struct E8 {
    static constexpr int BM = 32 * NR, A = BM * ET_LD, HALF = 16 * NR;
    // NR = 3: a THIRD tile region - the hidden quarters alternate between it and the attention-row region, so the barrier "every wave is
    // done reading the previous quarter" disappears (158 KB; at NR = 4 three regions would be 203 KB)
    static constexpr bool HDB = NR == 3;
    // + b1 (1024 floats) and b_o (256) parked behind the reduction scratch: as global loads inside the epilogues each was an exposed L2
    // round trip (and, vmcnt retiring in order, a wait for every weight fragment requested before it)
    static constexpr int VEC_FLOATS = 1024 + 256;
    static constexpr size_t LDS_BYTES = 2 * (size_t)((HDB ? 3 : 2) * A) + 2 * 8 * BM * sizeof(float) + VEC_FLOATS * sizeof(float);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert((size_t)HALF * ET_FLD * 4 <= 2 * (size_t)A, "half of the f32 rows must fit one bf16 tile region");
};

// acc[r] += A[r*32 + row][(koff + kk)*16 ..] * W over the 8 k-steps held in ring.f[BUF]; each fragment feeds four MFMAs
template <int BUF, int NR>
__device__ __forceinline__ void e8_gemm(const E6Ring& ring, const bf16_t* A, int koff, f32x16 (&acc)[NR], int lane) {
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(A + (r * 32 + l31) * ET_LD + (koff + kk) * 16 + half * 8);
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring.f[BUF][kk], af, acc[r], 0, 0, 0);
        }
}
// LayerNorm over 256 channels spread over the 8 waves (32 each), four row tiles per wave.  Two passes like every LayerNorm of the library
// (mean, then the sum of squared deviations), with the VALU work of a kernel whose LayerNorms cost 8 % of its cycles cut from 8 to 5
// operations per value: the deviations REPLACE the values in pass 2 (one subtraction, one fused multiply-add into the sum) and the
// affine step is (d * rstd) fused-multiply-added with gamma and beta.
template <int NR>
__device__ __forceinline__ void e8_layernorm(f32x16 (&acc)[NR], const float* __restrict__ gamma, const float* __restrict__ beta, float* red,
                                             int wave, int lane) {
    const int l31 = lane & 31, half = lane >> 5;
    constexpr int BM = 32 * NR;
    float mean[NR], rstd[NR];
    // gamma / beta requested in FRONT of the two reduction passes (they arrive under them; behind the passes each LayerNorm ended with
    // an exposed L2 round trip)
    f32x4 gq[4], bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        gq[q] = *reinterpret_cast<const f32x4*>(gamma + n);
        bq[q] = *reinterpret_cast<const f32x4*>(beta + n);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        float* rp = red + pass * 8 * BM;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float s = 0.f;
            if (pass == 0) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[r][e];
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[r][e] -= mean[r];
                    s = __builtin_fmaf(acc[r][e], acc[r][e], s);
                }
            }
            s += __shfl_xor(s, 32, 64);
            if (half == 0) rp[wave * BM + r * 32 + l31] = s;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) t += rp[w * BM + r * 32 + l31];
            if (pass == 0) mean[r] = t / ET_D;
            else rstd[r] = rsqrtf(t / ET_D + 1e-5f);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 g = gq[q], b = bq[q];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[r][4 * q + e] = __builtin_fmaf(acc[r][4 * q + e] * rstd[r], g[e], b[e]);
    }
}

// STAMP: tuning build - cycle stamps of every (workgroup, wave) at the phase boundaries into `dbg` [workgroups][8 waves][16] (scripts/enc_tail_stamps.py)
template <bool STAMP, int NR>
__global__ __launch_bounds__(512, 1) void enc_tail128_kernel(const EncTailArgs p, unsigned long long* dbg) {
    constexpr int E8_BM = E8<NR>::BM, E8_A = E8<NR>::A, HALF = E8<NR>::HALF;
    extern __shared__ __attribute__((aligned(16))) unsigned char et_smem[];
    bf16_t* At = reinterpret_cast<bf16_t*>(et_smem);         // attention rows [128][264]; then the hidden quarter; later bf16(y + pos)
    bf16_t* Yt = At + E8_A;                                  // bf16(y1) [128][264]; later bf16(y); last the projections' output staging
    bf16_t* Hb = E8<NR>::HDB ? Yt + E8_A : At;               // hidden quarters 1 and 3 (NR = 3: their own region)
    float* red = reinterpret_cast<float*>(Yt + (E8<NR>::HDB ? 2 : 1) * E8_A);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long m0 = (long long)blockIdx.x * E8_BM;
    E6Ring ring;
    unsigned long long ts[16];
    auto stamp = [&](int i) { if constexpr (STAMP) ts[i] = __builtin_readcyclecounter(); };
    stamp(0);
    // b1 / b_o -> LDS: the first loads of the kernel (unconditional: threads behind the 320th re-read the last piece), written with the attention rows
    float* VEC = red + 2 * 8 * E8_BM;
    const int vi = tid < 320 ? tid : 319;
    const f32x4 vec_in = *reinterpret_cast<const f32x4*>(vi < 256 ? p.b1 + 4 * vi : p.bo + 4 * (vi - 256));
    e6_issue<0>(ring, p.wo, 16, 0, wave, lane);                                   // out-proj tile `wave`, K 0..127
    // ---- the attention rows first (GEMM 1 waits for them), the residual rows behind them (they land under GEMM 1)
    // (rows behind M: the LAST row's data instead of a guarded load - a branch around a load is something the compiler's s_waitcnt
    // pass cannot count across, so the ds_writes below waited for the residual rows as well; those rows' results are never stored)
    const long long last = (long long)p.M - 1;
    us8 av[E8_BM * 32 / 512];
#pragma unroll
    for (int i = 0; i < E8_BM * 32 / 512; ++i) {
        const int c = tid + i * 512, r = c >> 5, col = (c & 31) * 8;
        const long long row = m0 + r < last ? m0 + r : last;
        av[i] = *reinterpret_cast<const us8*>(p.attn + row * ET_D + col);
    }
    f32x4 sv[NR][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const long long row = m0 + r * 32 + l31 < last ? m0 + r * 32 + l31 : last;
            sv[r][q] = *reinterpret_cast<const f32x4*>(p.src + row * ET_D + wave * 32 + 8 * q + 4 * half);
        }
#pragma unroll
    for (int i = 0; i < E8_BM * 32 / 512; ++i) {
        const int c = tid + i * 512, r = c >> 5, col = (c & 31) * 8;
        *reinterpret_cast<us8*>(At + r * ET_LD + col) = av[i];
    }
    if (tid < 320) *reinterpret_cast<f32x4*>(VEC + 4 * tid) = vec_in;
    stamp(1);
    __syncthreads();
    stamp(2);
    // ---- y1 = LN1(src + out_proj(attn))
    f32x16 y[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) et_zero(y[r]);
    e6_issue<1>(ring, p.wo, 16, 8, wave, lane);
    e8_gemm<0, NR>(ring, At, 0, y, lane);
    e6_issue<0>(ring, p.w1, 16, 0, wave, lane);                                   // linear1, quarter 0, tile `wave`, K 0..127
    e8_gemm<1, NR>(ring, At, 8, y, lane);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(VEC + 1024 + wave * 32 + 8 * q + 4 * half);
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) y[r][4 * q + e] = (y[r][4 * q + e] + b[e]) + sv[r][q][e];
    }
    stamp(3);
    f32x4 b2v[4];                                             // requested in front of LayerNorm 1, consumed behind it
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = *reinterpret_cast<const f32x4*>(p.b2 + wave * 32 + 8 * q + 4 * half);
    __builtin_amdgcn_sched_barrier(0);
    e8_layernorm<NR>(y, p.g1, p.be1, red, wave, lane);
    stamp(4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = wave * 32 + 8 * q + 4 * half;
        const f32x4 b = b2v[q];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            us4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = f32_to_bf16(y[r][4 * q + e]);
                y[r][4 * q + e] += b[e];                                          // linear2 accumulates on top of y1 + b2
            }
            *reinterpret_cast<us4*>(Yt + (r * 32 + l31) * ET_LD + n) = o;
        }
    }
    __syncthreads();                                          // bf16(y1) visible; every wave is done reading the attention rows
    stamp(5);
    // ---- the projection rounds that follow the FFN (needed here: the FFN's last quarter prefetches the first round's weights).
    // tb <= 8 (the encoder's q|k + v projections: 16 + 8 tiles): the `proj` tiles FIRST, from bf16(y); that tile is then free and every
    // round's outputs are staged in it and leave as whole rows.  Otherwise: tile nt = round * 8 + wave, stored straight from the accumulators.
    const int ta = p.wpa ? p.npa / 32 : 0, tb = p.wpb ? p.npb / 32 : 0, tt = ta + tb;
    const bool staged = tb <= 8;
    const int b_rounds = (staged && tb > 0) ? 1 : 0;
    const int nrounds = staged ? b_rounds + (ta + 7) / 8 : (tt + 7) / 8;
    auto round_tile = [&](int i) -> int {                     // this wave's tile (index into [Wpa ; Wpb]) in round i, -1: none
        if (i >= nrounds) return -1;
        if (!staged) return i * 8 + wave < tt ? i * 8 + wave : -1;
        if (b_rounds && i == 0) return wave < tb ? ta + wave : -1;
        const int nt = (i - b_rounds) * 8 + wave;
        return nt < ta ? nt : -1;
    };
    auto tile_w = [&](int nt) { return nt < ta ? p.wpa : p.wpb; };
    const int nt0 = round_tile(0);
    // ---- FFN in four quarters of 256 hidden channels: hidden_q = relu(linear1 tile 8q + wave), y += hidden_q W2[:, 256q ..]
    auto quarter = [&](auto Q) {                              // on entry ring buffer 0 holds K 0..127 of linear1 tile 8q + wave
        constexpr int q = decltype(Q)::value;
        f32x16 hd[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) et_zero(hd[r]);
        e6_issue<1>(ring, p.w1, 16, 8, 8 * q + wave, lane);
        e8_gemm<0, NR>(ring, Yt, 0, hd, lane);
        e6_issue<0>(ring, p.w2, 64, 16 * q, wave, lane);                          // linear2, K 256q .. +127
        e8_gemm<1, NR>(ring, Yt, 8, hd, lane);
        bf16_t* Hq = (q & 1) ? Hb : At;                       // this quarter's hidden tile
        if constexpr (q > 0 && !E8<NR>::HDB) __syncthreads(); // every wave is done reading hidden quarter q - 1 (two regions: not needed -
                                                              // whoever writes quarter q has passed the barrier behind quarter q - 1's
                                                              // writes, which every wave reaches only after its reads of quarter q - 2)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int nl = wave * 32 + 8 * qq + 4 * half;
            const f32x4 b = *reinterpret_cast<const f32x4*>(VEC + 256 * q + nl);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                us4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = hd[r][4 * qq + e] + b[e];
                    o[e] = f32_to_bf16(v > 0.f ? v : 0.f);
                }
                *reinterpret_cast<us4*>(Hq + (r * 32 + l31) * ET_LD + nl) = o;
            }
        }
        __syncthreads();
        e6_issue<1>(ring, p.w2, 64, 16 * q + 8, wave, lane);
        e8_gemm<0, NR>(ring, Hq, 0, y, lane);
        if constexpr (q < 3) e6_issue<0>(ring, p.w1, 16, 0, 8 * (q + 1) + wave, lane);
        e8_gemm<1, NR>(ring, Hq, 8, y, lane);
    };
    quarter(std::integral_constant<int, 0>{});
    stamp(6);
    quarter(std::integral_constant<int, 1>{});
    stamp(7);
    quarter(std::integral_constant<int, 2>{});
    stamp(8);
    quarter(std::integral_constant<int, 3>{});
    stamp(9);
    // (every per-thread address below comes from an OPAQUE copy of the thread index: the compiler otherwise shares the row / column
    // arithmetic of the prologue's residual loads with this phase and keeps ~80 registers of it alive - spilled - across the whole kernel)
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, l31b = lane2 & 31, half2 = lane2 >> 5;
    e8_layernorm<NR>(y, p.g2, p.be2, red, wave, lane2);           // its barriers also retire every read of the hidden tile / bf16(y1)
    stamp(10);
    // ---- outputs.  The normalised rows go through LDS as f32 (the first half of the rows in the first tile region, the rest in the second) and are picked
    // up again as 16-byte chunks of whole rows: the f32 residual stream leaves as 1 KB rows, the position rows arrive as whole rows, and
    // the two bf16 operand tiles of the projections (y, y + pos) are formed in that chunk layout - the accumulators are free from here on
    float* Yf0 = reinterpret_cast<float*>(At);
    float* Yf1 = reinterpret_cast<float*>(Yt);
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {y[r][4 * q], y[r][4 * q + 1], y[r][4 * q + 2], y[r][4 * q + 3]};
            const int rl = r * 32 + l31b;
            *reinterpret_cast<f32x4*>((rl < HALF ? Yf0 : Yf1) + (rl < HALF ? rl : rl - HALF) * ET_FLD + wave * 32 + 8 * q + 4 * half2) = v;
        }
    __syncthreads();
    if (nt0 >= 0) {                                           // the first projection tile's weights: they land under the chunk pass below
        e6_issue<0>(ring, tile_w(nt0), 16, 0, nt0 < ta ? nt0 : nt0 - ta, lane2);
        e6_issue<1>(ring, tile_w(nt0), 16, 8, nt0 < ta ? nt0 : nt0 - ta, lane2);
    }
    constexpr int NCH = E8_BM * 64 / 512;                     // 16 chunks of 4 channels per thread: rows (tid >> 6) + 8 i
    uint2 c16[NCH], cp16[NCH];
    // the thread's position row advances by 8 per chunk: ONE modulo, then add-and-wrap (a 32-bit modulo is ~30 VALU operations)
    const unsigned prow_n = p.pos ? (unsigned)p.pos_rows : 1u;
    unsigned prow = (unsigned)((m0 + (tid2 >> 6)) % (long long)prow_n);
    const unsigned pstep = 8u % prow_n;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid2 + i * 512, r = c >> 6, col = (c & 63) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>((r < HALF ? Yf0 : Yf1) + (r < HALF ? r : r - HALF) * ET_FLD + col);
        if (p.y && m0 + r < p.M) *reinterpret_cast<f32x4*>(p.y + (m0 + r) * ET_D + col) = v;
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (p.pos) pv = *reinterpret_cast<const f32x4*>(p.pos + (long long)prow * ET_D + col);    // (wave-uniform condition; rows behind M read a valid row)
        prow += pstep;
        prow = prow >= prow_n ? prow - prow_n : prow;
        c16[i] = uint2{f32x2_to_bf16x2(v[0], v[1]), f32x2_to_bf16x2(v[2], v[3])};
        cp16[i] = uint2{f32x2_to_bf16x2(v[0] + pv[0], v[1] + pv[1]), f32x2_to_bf16x2(v[2] + pv[2], v[3] + pv[3])};
    }
    __syncthreads();                                          // every f32 chunk has been read: the regions become the bf16 tiles
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid2 + i * 512, r = c >> 6, col = (c & 63) * 4;
        *reinterpret_cast<uint2*>(Yt + r * ET_LD + col) = c16[i];
        *reinterpret_cast<uint2*>(At + r * ET_LD + col) = cp16[i];
    }
    stamp(11);
    __syncthreads();
    stamp(12);
    if (p.y16 || p.ypos16) {
#pragma unroll
        for (int i = 0; i < E8_BM * 32 / 512; ++i) {
            const int c = tid2 + i * 512, r = c >> 5, col = (c & 31) * 8;
            if (m0 + r < p.M) {
                if (p.y16) *reinterpret_cast<us8*>(p.y16 + (m0 + r) * ET_D + col) = *reinterpret_cast<const us8*>(Yt + r * ET_LD + col);
                if (p.ypos16) *reinterpret_cast<us8*>(p.ypos16 + (m0 + r) * ET_D + col) = *reinterpret_cast<const us8*>(At + r * ET_LD + col);
            }
        }
    }
    // ---- the next attention's input projections from the two bf16 tiles (At = y + pos, Yt = y), K = 256 per tile
    for (int i = 0; i < nrounds; ++i) {
        const int nt = round_tile(i), nn = round_tile(i + 1);
        const bool is_a = nt < ta;                            // (nt = -1: an idle wave of a partial round takes part in the barriers only)
        const int ct = is_a ? nt : nt - ta;
        f32x16 acc[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) et_zero(acc[r]);
        // this round's bias in front of the GEMM: the load is unconditional (a wave without a tile / a projection without a bias reads b2
        // and selects zero) and old by the time the epilogue wants it - behind the GEMM it was an exposed L2 round trip per round
        const float* bias_v = nt >= 0 ? (is_a ? p.bpa : p.bpb) : nullptr;
        const bool has_bias = bias_v != nullptr;              // (wave-uniform)
        const float* bias_p = has_bias ? bias_v + ct * 32 : p.b2;
        f32x4 pbv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(bias_p + 8 * q + 4 * half2);
            pbv[q] = has_bias ? t : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
        if (nt >= 0) {
            e8_gemm<0, NR>(ring, is_a ? At : Yt, 0, acc, lane2);
            if (nn >= 0) e6_issue<0>(ring, tile_w(nn), 16, 0, nn < ta ? nn : nn - ta, lane2);   // the next tile's fragments replace this one's
            e8_gemm<1, NR>(ring, is_a ? At : Yt, 8, acc, lane2);
            if (nn >= 0) e6_issue<1>(ring, tile_w(nn), 16, 8, nn < ta ? nn : nn - ta, lane2);
        } else if (nn >= 0) {
            e6_issue<0>(ring, tile_w(nn), 16, 0, nn < ta ? nn : nn - ta, lane2);
            e6_issue<1>(ring, tile_w(nn), 16, 8, nn < ta ? nn : nn - ta, lane2);
        }
        if (!staged) {
            if (nt >= 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = 8 * q + 4 * half2;
                    const f32x4 b = pbv[q];
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const long long row = m0 + r * 32 + l31b;
                        us4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(acc[r][4 * q + e] + b[e]);
                        if (row < p.M) *reinterpret_cast<us4*>((is_a ? p.pa : p.pb) + row * (is_a ? p.npa : p.npb) + ct * 32 + n) = o;
                    }
                }
            }
            continue;
        }
        // staged: this round's [128][<= 256] output block goes through the bf16(y) tile (free once the `proj` round has read it) and
        // leaves as 16-byte chunks of whole rows
        const bool round_a = !(b_rounds && i == 0);
        const int base = round_a ? (i - b_rounds) * 8 : 0, width = round_a ? (ta - base < 8 ? ta - base : 8) : tb;      // tiles in this round
        __syncthreads();                                      // the previous round's chunks have left the tile / this round's reads of it are done
        if (nt >= 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 8 * q + 4 * half2;
                const f32x4 b = pbv[q];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    us4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = f32_to_bf16(acc[r][4 * q + e] + b[e]);
                    *reinterpret_cast<us4*>(Yt + (r * 32 + l31b) * ET_LD + wave * 32 + n) = o;
                }
            }
        }
        __syncthreads();
        bf16_t* outp = round_a ? p.pa + base * 32 : p.pb;
        const int ldo = round_a ? p.npa : p.npb, cpr = width * 4;             // 16-byte chunks per row
        int tid3 = tid2;                                      // (opaque per round: the chunk addresses are not loop-invariant values to keep alive - spilled - across the MFMAs)
        asm volatile("" : "+v"(tid3));
        if (cpr == 32) {
#pragma unroll
            for (int k = 0; k < E8_BM * 32 / 512; ++k) {
                const int c = tid3 + k * 512, r = c >> 5, col = (c & 31) * 8;
                if (m0 + r < p.M) *reinterpret_cast<us8*>(outp + (m0 + r) * ldo + col) = *reinterpret_cast<const us8*>(Yt + r * ET_LD + col);
            }
        } else {
            for (int c = tid3; c < E8_BM * cpr; c += 512) {
                const int r = c / cpr, col = (c - r * cpr) * 8;
                if (m0 + r < p.M) *reinterpret_cast<us8*>(outp + (m0 + r) * ldo + col) = *reinterpret_cast<const us8*>(Yt + r * ET_LD + col);
            }
        }
    }
    if constexpr (STAMP) {
        stamp(13);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stores' acknowledgements (not waited for in the product build)
        stamp(14);
        if (dbg && lane == 0)
            for (int i = 0; i < 15; ++i) dbg[((long long)blockIdx.x * 8 + wave) * 16 + i] = ts[i];
    }
}

static unsigned long long* g_et_dbg = nullptr;
extern "C" void nps_enc_tail_debug_buffer(void* buf) { g_et_dbg = (unsigned long long*)buf; }

static int et_launch(const EncTailArgs& a, hipStream_t stream) {
    // the encoder (post-norm, thousands of tokens) on the 64-token kernel; decoder forms and small inputs on the 32-token one
    // (NOPESAC_ENC_TAIL_64=1: the round-3 64-token kernel, NOPESAC_ENC_TAIL_32=1: the 32-token one - A/B runs and the bit-identity test of those two)
    if (!a.pre_norm && !a.skip_ffn && a.M >= 2048 && !getenv("NOPESAC_ENC_TAIL_32") && !getenv("NOPESAC_ENC_TAIL_64")) {
        // 96 tokens per workgroup when that fills more CUs than 128 do in one round (the encoder's 19200 tokens: 200 workgroups instead
        // of 150); NOPESAC_ENC_TAIL_ROWS=3|4 forces one form (A/B runs, tests)
        int nr = ((a.M + 127) / 128 <= 160 && (a.M + 95) / 96 <= 256) ? 3 : 4;
        if (const char* e = getenv("NOPESAC_ENC_TAIL_ROWS")) nr = atoi(e) == 3 ? 3 : 4;
        auto go = [&](auto K, size_t lds, int bm) {
            NPS_ENSURE_LDS((int)lds, K);
            hipLaunchKernelGGL(K, dim3((a.M + bm - 1) / bm), dim3(512), lds, stream, a, g_et_dbg);
        };
        if (g_et_dbg) {                                       // tuning runs only (scripts/enc_tail_stamps.py)
            if (nr == 3) go(enc_tail128_kernel<true, 3>, E8<3>::LDS_BYTES, 96); else go(enc_tail128_kernel<true, 4>, E8<4>::LDS_BYTES, 128);
        } else {
            if (nr == 3) go(enc_tail128_kernel<false, 3>, E8<3>::LDS_BYTES, 96); else go(enc_tail128_kernel<false, 4>, E8<4>::LDS_BYTES, 128);
        }
    } else if (!a.pre_norm && !a.skip_ffn && a.M >= 2048 && !getenv("NOPESAC_ENC_TAIL_32")) {
        NPS_ENSURE_LDS((int)E6_LDS_BYTES, enc_tail64_kernel);
        hipLaunchKernelGGL(enc_tail64_kernel, dim3((a.M + E6_BM - 1) / E6_BM), dim3(512), E6_LDS_BYTES, stream, a);
    } else {
        // the 32-token kernel issues its projection tiles in four unrolled rounds of 8 column tiles: 32 tiles = 1024 output columns
        NPS_CHECK_ARG((a.wpa ? a.npa : 0) + (a.wpb ? a.npb : 0) <= 1024, "transformer_tail: n_pos + n_proj > 1024 on the 32-token kernel");
        NPS_ENSURE_LDS((int)ET_LDS_BYTES, enc_tail_kernel);
        EncTailArgs b = a;
        b.n_work = (a.M + ET_BM - 1) / ET_BM;
        int extra = 0;
        if (b.n_pf > 0 && b.pf_xcds > 0 && b.n_work <= 64) extra = 8 * ET_PF_PER_XCD;      // (many workgroups share the weights in every L2 anyway)
        else b.n_pf = 0;
        hipLaunchKernelGGL(enc_tail_kernel, dim3(b.n_work + extra), dim3(512), ET_LDS_BYTES, stream, b);
    }
    return 0;
}

}  // namespace nps

extern "C" int nopesac_encoder_tail_bf16(const void* attn, const float* src, const void* wo, const float* bo, const float* ln1_g,
                                         const float* ln1_b, const void* w1, const float* b1, const void* w2, const float* b2,
                                         const float* ln2_g, const float* ln2_b, const float* pos, int pos_rows, float* y, void* y_bf16,
                                         void* ypos_bf16, int M, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(attn && src && wo && bo && ln1_g && ln1_b && w1 && b1 && w2 && b2 && ln2_g && ln2_b && M > 0, "encoder_tail: null pointer");
    NPS_CHECK_ARG(y || y_bf16 || ypos_bf16, "encoder_tail: no output requested");
    NPS_CHECK_ARG(!ypos_bf16 || (pos && pos_rows > 0), "encoder_tail: ypos needs pos");
    const void* ptrs[] = {attn, src, wo, bo, ln1_g, ln1_b, w1, b1, w2, b2, ln2_g, ln2_b, pos, y, y_bf16, ypos_bf16};
    for (const void* q : ptrs) NPS_CHECK_ARG(((uintptr_t)q & 15) == 0, "encoder_tail: pointers must be 16-byte aligned");
    EncTailArgs a;
    a.attn = (const bf16_t*)attn; a.src = src; a.wo = (const bf16_t*)wo; a.bo = bo; a.g1 = ln1_g; a.be1 = ln1_b;
    a.w1 = (const bf16_t*)w1; a.b1 = b1; a.w2 = (const bf16_t*)w2; a.b2 = b2; a.g2 = ln2_g; a.be2 = ln2_b;
    a.pos = pos; a.pos_rows = pos_rows; a.y = y; a.y16 = (bf16_t*)y_bf16; a.ypos16 = (bf16_t*)ypos_bf16; a.M = M;
    a.yn = nullptr; a.pre_norm = 0;
    a.wpa = a.wpb = nullptr; a.bpa = a.bpb = nullptr; a.pa = a.pb = nullptr; a.npa = a.npb = 0; a.skip_ffn = 0;
    a.n_pf = 0; a.n_work = 0; a.pf_xcds = 0;
    if (const int rc = et_launch(a, (hipStream_t)stream)) return rc;
    NPS_LAUNCH_RET();
}

// Pre-norm counterpart for the decoder (transformer/transformer.py:293-322 forward_pre, after the cross-attention):
//     s = tgt + out_proj(attn);  u = s + linear2(relu(linear1(LN3(s))));  n = LN_next(u)
// y = u (f32 residual stream); y_bf16 = bf16(n), ypos_bf16 = bf16(n + pos) feed the next layer's projections; yn = n in f32
// (the decoder's final norm).  Same kernel as the encoder tail (pre_norm = 1).
extern "C" int nopesac_decoder_tail_bf16(const void* attn, const float* tgt, const void* wo, const float* bo, const float* ln3_g,
                                         const float* ln3_b, const void* w1, const float* b1, const void* w2, const float* b2,
                                         const float* lnn_g, const float* lnn_b, const float* pos, int pos_rows, float* y, void* y_bf16,
                                         void* ypos_bf16, float* yn, int M, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(attn && tgt && wo && bo && ln3_g && ln3_b && w1 && b1 && w2 && b2 && lnn_g && lnn_b && M > 0, "decoder_tail: null pointer");
    NPS_CHECK_ARG(y || y_bf16 || ypos_bf16 || yn, "decoder_tail: no output requested");
    NPS_CHECK_ARG(!ypos_bf16 || (pos && pos_rows > 0), "decoder_tail: ypos needs pos");
    const void* ptrs[] = {attn, tgt, wo, bo, ln3_g, ln3_b, w1, b1, w2, b2, lnn_g, lnn_b, pos, y, y_bf16, ypos_bf16, yn};
    for (const void* q : ptrs) NPS_CHECK_ARG(((uintptr_t)q & 15) == 0, "decoder_tail: pointers must be 16-byte aligned");
    EncTailArgs a;
    a.attn = (const bf16_t*)attn; a.src = tgt; a.wo = (const bf16_t*)wo; a.bo = bo; a.g1 = ln3_g; a.be1 = ln3_b;
    a.w1 = (const bf16_t*)w1; a.b1 = b1; a.w2 = (const bf16_t*)w2; a.b2 = b2; a.g2 = lnn_g; a.be2 = lnn_b;
    a.pos = pos; a.pos_rows = pos_rows; a.y = y; a.y16 = (bf16_t*)y_bf16; a.ypos16 = (bf16_t*)ypos_bf16; a.M = M;
    a.yn = yn; a.pre_norm = 1;
    a.wpa = a.wpb = nullptr; a.bpa = a.bpb = nullptr; a.pa = a.pb = nullptr; a.npa = a.npb = 0; a.skip_ffn = 0;
    a.n_pf = 0; a.n_work = 0; a.pf_xcds = 0;
    if (const int rc = et_launch(a, (hipStream_t)stream)) return rc;
    NPS_LAUNCH_RET();
}

// General form: the tail of a transformer layer followed by the input projections of the NEXT attention, one launch.
//   pre_norm = 0 (encoder, transformer.py:183-199): n = LN2(y1 + FFN(y1)), y1 = LN1(src + out_proj(attn)); y = n
//   pre_norm = 1 (decoder, :293-322): u = s + FFN(LN_a(s)), s = src + out_proj(attn); n = LN_b(u); y = u
//   skip_ffn = 1 (the decoder's self-attention half, :300-306): s = src + out_proj(attn), n = LN_a(s), y = s  (w1 / w2 / ln_b unused)
//   proj_pos [M][n_pos] = bf16((n + pos) Wpos^T + bpos), proj [M][n_proj] = bf16(n Wp^T + bp): e.g. the next layer's q|k and v
//   (encoder / decoder self-attention) or the cross-attention's q; fragment-major weights (K = 256), n_pos / n_proj multiples of 32.
extern "C" int nopesac_transformer_tail_bf16_pf(const void* attn, const float* src, const void* wo, const float* bo, const float* lna_g,
                                                const float* lna_b, const void* w1, const float* b1, const void* w2, const float* b2,
                                                const float* lnb_g, const float* lnb_b, const float* pos, int pos_rows, float* y, void* y_bf16,
                                                void* ypos_bf16, float* yn, int pre_norm, int skip_ffn, const void* w_pos, const float* b_pos,
                                                void* proj_pos, int n_pos, const void* w_proj, const float* b_proj, void* proj, int n_proj, int M,
                                                const void* const* next_ptrs, const int64_t* next_bytes, int n_next, int next_workgroups,
                                                void* stream);

extern "C" int nopesac_transformer_tail_bf16(const void* attn, const float* src, const void* wo, const float* bo, const float* lna_g,
                                             const float* lna_b, const void* w1, const float* b1, const void* w2, const float* b2,
                                             const float* lnb_g, const float* lnb_b, const float* pos, int pos_rows, float* y, void* y_bf16,
                                             void* ypos_bf16, float* yn, int pre_norm, int skip_ffn, const void* w_pos, const float* b_pos,
                                             void* proj_pos, int n_pos, const void* w_proj, const float* b_proj, void* proj, int n_proj, int M,
                                             void* stream) {
    return nopesac_transformer_tail_bf16_pf(attn, src, wo, bo, lna_g, lna_b, w1, b1, w2, b2, lnb_g, lnb_b, pos, pos_rows, y, y_bf16, ypos_bf16, yn,
                                            pre_norm, skip_ffn, w_pos, b_pos, proj_pos, n_pos, w_proj, b_proj, proj, n_proj, M, nullptr, nullptr, 0, 0,
                                            stream);
}

extern "C" int nopesac_transformer_tail_bf16_pf(const void* attn, const float* src, const void* wo, const float* bo, const float* lna_g,
                                                const float* lna_b, const void* w1, const float* b1, const void* w2, const float* b2,
                                                const float* lnb_g, const float* lnb_b, const float* pos, int pos_rows, float* y, void* y_bf16,
                                                void* ypos_bf16, float* yn, int pre_norm, int skip_ffn, const void* w_pos, const float* b_pos,
                                                void* proj_pos, int n_pos, const void* w_proj, const float* b_proj, void* proj, int n_proj, int M,
                                                const void* const* next_ptrs, const int64_t* next_bytes, int n_next, int next_workgroups,
                                                void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(n_next >= 0 && n_next <= ET_PF_MAX && (n_next == 0 || (next_ptrs && next_bytes && next_workgroups > 0)),
                  "transformer_tail: bad prefetch list (at most %d ranges)", ET_PF_MAX);
    NPS_CHECK_ARG(attn && src && wo && bo && lna_g && lna_b && M > 0, "transformer_tail: null pointer");
    NPS_CHECK_ARG(skip_ffn || (w1 && b1 && w2 && b2 && lnb_g && lnb_b), "transformer_tail: FFN / second norm parameters missing");
    NPS_CHECK_ARG(!skip_ffn || pre_norm, "transformer_tail: skip_ffn is the pre-norm (decoder) form");
    NPS_CHECK_ARG(y || y_bf16 || ypos_bf16 || yn || proj_pos || proj, "transformer_tail: no output requested");
    NPS_CHECK_ARG((!ypos_bf16 && !proj_pos) || (pos && pos_rows > 0), "transformer_tail: ypos / proj_pos need pos");
    NPS_CHECK_ARG((!proj_pos || (w_pos && n_pos > 0 && n_pos % 32 == 0)) && (!proj || (w_proj && n_proj > 0 && n_proj % 32 == 0)),
                  "transformer_tail: projection weights / widths (multiples of 32)");
    const void* ptrs[] = {attn, src, wo, bo, lna_g, lna_b, w1, b1, w2, b2, lnb_g, lnb_b, pos, y, y_bf16, ypos_bf16, yn, w_pos, b_pos, proj_pos,
                          w_proj, b_proj, proj};
    for (const void* q : ptrs) NPS_CHECK_ARG(((uintptr_t)q & 15) == 0, "transformer_tail: pointers must be 16-byte aligned");
    EncTailArgs a;
    a.attn = (const bf16_t*)attn; a.src = src; a.wo = (const bf16_t*)wo; a.bo = bo; a.g1 = lna_g; a.be1 = lna_b;
    a.w1 = (const bf16_t*)w1; a.b1 = b1; a.w2 = (const bf16_t*)w2; a.b2 = b2; a.g2 = lnb_g; a.be2 = lnb_b;
    a.pos = pos; a.pos_rows = pos_rows; a.y = y; a.y16 = (bf16_t*)y_bf16; a.ypos16 = (bf16_t*)ypos_bf16; a.M = M;
    a.yn = yn; a.pre_norm = pre_norm ? 1 : 0; a.skip_ffn = skip_ffn ? 1 : 0;
    a.wpa = proj_pos ? (const bf16_t*)w_pos : nullptr; a.bpa = b_pos; a.pa = (bf16_t*)proj_pos; a.npa = n_pos;
    a.wpb = proj ? (const bf16_t*)w_proj : nullptr; a.bpb = b_proj; a.pb = (bf16_t*)proj; a.npb = n_proj;
    a.n_pf = n_next; a.n_work = 0; a.pf_xcds = next_workgroups < 8 ? next_workgroups : 8;
    for (int i = 0; i < n_next; ++i) {
        NPS_CHECK_ARG(next_ptrs[i] && ((uintptr_t)next_ptrs[i] & 15) == 0 && next_bytes[i] > 0 && next_bytes[i] < (1ll << 30),
                      "transformer_tail: prefetch range %d null / unaligned / too large", i);
        a.pf[i] = (const unsigned char*)next_ptrs[i]; a.pf_bytes[i] = (int)next_bytes[i];
    }
    if (const int rc = et_launch(a, (hipStream_t)stream)) return rc;
    NPS_LAUNCH_RET();
}
