// How fast can 256 persistent workgroups stream a [M][K] bf16 matrix when every step of a workgroup covers 256 rows x SEG bytes
// (SEG = 128: the K-tile of the 1x1 p8 convs; SEG = 2 K: whole rows, what the fused tails read)?  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SEG, int INFLIGHT>
__global__ __launch_bounds__(512, 1) void stream_kernel(const unsigned char* __restrict__ x, int M, int rowbytes, unsigned int* sink) {
    constexpr int LPR = SEG / 16;                 // 16-byte loads per row per step
    constexpr int ROWS_PER_PASS = 512 / LPR;      // rows covered by one load of every thread
    constexpr int PASSES = 256 / ROWS_PER_PASS;   // loads per thread per step
    const int tid = threadIdx.x;
    const int r0 = tid / LPR, c0 = (tid % LPR) * 16;
    const int ntile = M / 256, nseg = rowbytes / SEG;
    u32x4 acc = {0, 0, 0, 0};
    // XCD-aware walk like the conv kernels: XCD x owns a contiguous run of tiles
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8, per_xcd = gridDim.x / 8;
    const int tq = ntile / 8;
    for (int t = xcd * tq + slot; t < (xcd + 1) * tq; t += per_xcd) {
        const unsigned char* base = x + (size_t)t * 256 * rowbytes;
        for (int s = 0; s < nseg; s += INFLIGHT) {
            u32x4 v[INFLIGHT][PASSES];
#pragma unroll
            for (int i = 0; i < INFLIGHT; ++i)
#pragma unroll
                for (int p = 0; p < PASSES; ++p)
                    v[i][p] = *reinterpret_cast<const u32x4*>(base + (size_t)(r0 + p * ROWS_PER_PASS) * rowbytes + (size_t)(s + i) * SEG + c0);
#pragma unroll
            for (int i = 0; i < INFLIGHT; ++i)
#pragma unroll
                for (int p = 0; p < PASSES; ++p) acc ^= v[i][p];
        }
    }
    if (acc.x == 0x12345678u) sink[0] = acc.y ^ acc.z ^ acc.w;
}

template <int SEG, int INFLIGHT>
void run(const unsigned char* x, int M, int rowbytes, unsigned int* sink, const char* label) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<SEG, INFLIGHT>), dim3(256), dim3(512), 0, 0, x, M, rowbytes, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((stream_kernel<SEG, INFLIGHT>), dim3(256), dim3(512), 0, 0, x, M, rowbytes, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)(M / 2048 * 2048) * rowbytes;
    printf("%-34s rows of %4d B: %7.1f us  %5.2f TB/s\n", label, rowbytes, ms * 1000 / 20, bytes / (ms / 20 * 1e-3) / 1e12);
}

int main() {
    const int M = 76800;                          // 64 images x 30 x 40 pixels
    unsigned char* x; unsigned int* sink;
    hipMalloc(&x, (size_t)M * 4096 + 4096); hipMalloc(&sink, 64);
    hipMemset(x, 1, (size_t)M * 4096);
    for (int rowbytes : {2048, 4096, 512}) {     // 1024 / 2048 / 256 channels
        run<128, 4>(x, M, rowbytes, sink, "128 B per row per step, 4 steps");
        run<128, 8>(x, M, rowbytes, sink, "128 B per row per step, 8 steps");
        run<256, 4>(x, M, rowbytes, sink, "256 B per row per step, 4 steps");
        run<512, 2>(x, M, rowbytes, sink, "512 B per row per step, 2 steps");
        if (rowbytes >= 2048) run<2048, 1>(x, M, rowbytes, sink, "2 KB per row per step");
    }
    return 0;
}
