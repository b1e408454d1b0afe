// Shared helpers for the gfx950 kernels of libnopesac_hip.so (not a public header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/nopesac_hip.h"

namespace nps {

typedef unsigned short bf16_t;  // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per instruction);
// the integer sequence it replaces (add 0x7fff + lsb, shift) cost 4-5 VALU operations per value in every epilogue.
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// OCP e4m3fn (gfx950's fp8): 8 floats -> 8 bytes, round to nearest even, saturating at +-448 (no NaN from overflow).
__device__ __forceinline__ uint2 f32x8_to_fp8(const float* v) {
    float c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) c[e] = fminf(fmaxf(v[e], -448.f), 448.f);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f32_to_bf16(v); }

__device__ __forceinline__ float apply_act(float v, int act) {
    // branch-free for none / ReLU / LeakyReLU (`act` is wave-uniform: the selects below are scalar); only the sigmoid branches.
    // A switch here, inlined per element into unrolled epilogue loops, compiled to thousands of scalar branches.
    if (act == NPS_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    const float neg = act == NPS_ACT_RELU ? 0.f : (act == NPS_ACT_LEAKY ? 0.01f * v : v);
    return v > 0.f ? v : neg;
}

// N values with ONE wave-uniform decision (same values as apply_act per element).  Round 6: conv3x3_c64_kernel called apply_act per element
// in its unrolled epilogue - 4.7 k lines of ISA (64 inlined sigmoid bodies behind 256 scalar branches, 37 KB of code: more than the
// instruction cache holds next to the main loop), 11.6 k of the workgroup's 29.8 k cycles (in-kernel stamps, profiles/r6_e_*).
template <int N>
__device__ __forceinline__ void apply_act_n(float (&v)[N], int act) {
    if (act == NPS_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    } else if (act == NPS_ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
    } else if (act == NPS_ACT_SIGMOID) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = 1.f / (1.f + expf(-v[e]));
    }
}

// The epilogue form: 8 values at a time, ONE wave-uniform decision per vector instead of a switch per element (a per-element
// switch inside unrolled epilogue loops compiled to thousands of scalar branches: the 256x256 conv tile's epilogue was 20 k
// lines of ISA, larger than the instruction cache, 43 k cycles per tile).  Same operations in the same order as
//   res_after ? apply_act(v, act) + r : apply_act(v + r, act)
__device__ __forceinline__ void act_residual8(float (&v)[8], const float (&r)[8], int act, int res_after) {
    if (!res_after) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (act == NPS_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    } else if (act == NPS_ACT_LEAKY) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.01f * v[e];
    } else if (act == NPS_ACT_SIGMOID) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 1.f / (1.f + expf(-v[e]));
    }
    if (res_after) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// quaternion (w,x,y,z) -> row-major 3x3 (camera_head.py:1148-1173)
__device__ __forceinline__ void quat_to_rot(const float q[4], float R[9]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * y * y - 2 * z * z; R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
    R[3] = 2 * x * y + 2 * w * z; R[4] = 1 - 2 * x * x - 2 * z * z; R[5] = 2 * y * z - 2 * w * x;
    R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = 1 - 2 * x * x - 2 * y * y;
}

// warp a view-1 plane vector p (= n*d, camera frame) into the common frame under (R,t)
// (camera_head.py:1446-1454): end = R*flip(p) + t; b = end - t; out = ((end.b)/(|b|+1e-5)^2) b
__device__ __forceinline__ void warp_plane(const float p[3], const float R[9], const float t[3], float out[3]) {
    const float f0 = p[0], f1 = -p[1], f2 = -p[2];
    float e[3], b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // keep torch's bmm order: ((R[i0]*f0 + R[i1]*f1) + R[i2]*f2), then + t
        float s = __fmul_rn(R[3 * i], f0);
        s = __fmaf_rn(R[3 * i + 1], f1, s);
        s = __fmaf_rn(R[3 * i + 2], f2, s);
        e[i] = __fadd_rn(s, t[i]);
        b[i] = __fsub_rn(e[i], t[i]);
    }
    const float dot = e[0] * b[0] + e[1] * b[1] + e[2] * b[2];
    const float nb = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]) + 1e-5f;
    const float c = dot / (nb * nb);
    out[0] = c * b[0]; out[1] = c * b[1]; out[2] = c * b[2];
}

__device__ __forceinline__ float norm3(const float v[3]) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
// F.normalize(p=2, eps=1e-12)
__device__ __forceinline__ void normalize3(const float v[3], float o[3]) {
    const float n = fmaxf(norm3(v), 1e-12f);
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}

void set_error(const char* fmt, ...);

}  // namespace nps

#define NPS_CHECK_ARG(cond, ...)               \
    do {                                       \
        if (!(cond)) {                         \
            nps::set_error(__VA_ARGS__);       \
            return NPS_E_ARG;                  \
        }                                      \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: set it once per (call site, device), not once per
// process (a process that drives several GPUs would otherwise fail to launch on the second one).  Usage:
// NPS_ENSURE_LDS(bytes, kernel<template, args>);
#define NPS_ENSURE_LDS(bytes, ...)                                                                                     \
    do {                                                                                                               \
        static std::atomic<unsigned long long> done__{0ull};                                                           \
        int dev__ = 0;                                                                                                 \
        (void)hipGetDevice(&dev__);                                                                                    \
        const unsigned long long bit__ = 1ull << (dev__ & 63);                                                         \
        if (!(done__.load(std::memory_order_relaxed) & bit__)) {                                                       \
            (void)hipFuncSetAttribute((const void*)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            done__.fetch_or(bit__, std::memory_order_relaxed);                                                         \
        }                                                                                                              \
    } while (0)

#define NPS_LAUNCH_RET()                                                   \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            nps::set_error("launch failed: %s", hipGetErrorString(e__));   \
            return (int)e__;                                               \
        }                                                                  \
        return 0;                                                          \
    } while (0)
