// Image resize of the ScanNet input path: the reference mapper calls cv2.resize(image, (640, 480)) - default INTER_LINEAR on
// uint8 - before handing the image to the model (data/planercnn_transforms.py:314).  OpenCV is a third-party dependency that
// is not in the reference tree; this kernel restates its published 8-bit INTER_LINEAR algorithm (modules/imgproc resize.cpp:
// 11-bit fixed-point coefficients, HResizeLinear then VResizeLinear with FixedPtCast):
//     sx = (dx + 0.5) * (W / OW) - 0.5, ix = floor(sx), fx = sx - ix  (clamped at the borders, fx = 0 there)
//     a1 = sat_short(rint(fx * 2048)), a0 = sat_short(rint((1 - fx) * 2048))           (same for rows: b0, b1)
//     row_k[dx] = S[k][ix] * a0 + S[k][ix + 1] * a1                                     (int32)
//     dst = ( ((b0 * (row_0 >> 4)) >> 16) + ((b1 * (row_1 >> 4)) >> 16) + 2 ) >> 2
#include "common.h"

namespace nps {

__device__ __forceinline__ void resize_coef(int d, float scale, int n, int& i0, int& i1, int& c0, int& c1) {
    float f = (float)(((double)d + 0.5) * (double)scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n - 1) { f = 0.f; s = n - 1; }
    i0 = s;
    i1 = min(s + 1, n - 1);
    c0 = (int)rintf((1.f - f) * 2048.f);
    c1 = (int)rintf(f * 2048.f);
}

__global__ void resize_bilinear_u8_kernel(const uint8_t* __restrict__ src, int H, int W, int C, uint8_t* __restrict__ dst, int OH,
                                          int OW, float scale_y, float scale_x) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= (long long)OH * OW * C) return;
    const int c = (int)(idx % C), dx = (int)((idx / C) % OW), dy = (int)(idx / ((long long)C * OW));
    int x0, x1, a0, a1, y0, y1, b0, b1;
    resize_coef(dx, scale_x, W, x0, x1, a0, a1);
    resize_coef(dy, scale_y, H, y0, y1, b0, b1);
    const int r0 = (int)src[((long long)y0 * W + x0) * C + c] * a0 + (int)src[((long long)y0 * W + x1) * C + c] * a1;
    const int r1 = (int)src[((long long)y1 * W + x0) * C + c] * a0 + (int)src[((long long)y1 * W + x1) * C + c] * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    dst[idx] = (uint8_t)min(max(v, 0), 255);
}

// n images of one size, `src_stride` bytes apart (the GPU JPEG decoder's output buffer), in ONE launch; chw = 1 writes [n][C][OH][OW] -
// the data mapper's layout - instead of [n][OH][OW][C] (the per-image path: resize, permute, contiguous = three launches per image).
// Element for element the arithmetic of resize_bilinear_u8_kernel.
__global__ void resize_bilinear_u8_batch_kernel(const uint8_t* __restrict__ src, long long src_stride, int H, int W, int C, uint8_t* __restrict__ dst,
                                                int OH, int OW, int chw, float scale_y, float scale_x) {
    const long long per = (long long)OH * OW * C;
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= per) return;
    const uint8_t* s = src + (long long)blockIdx.y * src_stride;
    int c, dx, dy;
    if (chw) { dx = (int)(idx % OW); dy = (int)((idx / OW) % OH); c = (int)(idx / ((long long)OW * OH)); }
    else { c = (int)(idx % C); dx = (int)((idx / C) % OW); dy = (int)(idx / ((long long)C * OW)); }
    int x0, x1, a0, a1, y0, y1, b0, b1;
    resize_coef(dx, scale_x, W, x0, x1, a0, a1);
    resize_coef(dy, scale_y, H, y0, y1, b0, b1);
    const int r0 = (int)s[((long long)y0 * W + x0) * C + c] * a0 + (int)s[((long long)y0 * W + x1) * C + c] * a1;
    const int r1 = (int)s[((long long)y1 * W + x0) * C + c] * a0 + (int)s[((long long)y1 * W + x1) * C + c] * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    dst[(long long)blockIdx.y * per + idx] = (uint8_t)min(max(v, 0), 255);
}

}  // namespace nps

extern "C" int nopesac_resize_bilinear_u8_batch(const uint8_t* src, int n, int64_t src_stride, int H, int W, int C, uint8_t* dst, int OH, int OW, int chw,
                                                void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(src && dst && n > 0 && n <= 65535 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0 && src_stride >= (int64_t)H * W * C, "resize_batch: bad args");
    const long long per = (long long)OH * OW * C;
    hipLaunchKernelGGL(resize_bilinear_u8_batch_kernel, dim3((unsigned)((per + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, src,
                       (long long)src_stride, H, W, C, dst, OH, OW, chw ? 1 : 0, (float)((double)H / OH), (float)((double)W / OW));
    NPS_LAUNCH_RET();
}

extern "C" int nopesac_resize_bilinear_u8(const uint8_t* src, int H, int W, int C, uint8_t* dst, int OH, int OW, void* stream) {
    using namespace nps;
    NPS_CHECK_ARG(src && dst && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "resize: bad args");
    const long long total = (long long)OH * OW * C;
    hipLaunchKernelGGL(resize_bilinear_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, H, W, C, dst,
                       OH, OW, (float)((double)H / OH), (float)((double)W / OW));
    NPS_LAUNCH_RET();
}
