// Which CUs does a stream created with hipExtStreamCreateWithCUMask use on this part?  Every workgroup records (XCC, SE, SH, CU) from the
// hardware-id registers while it spins; the host counts the distinct CUs per mask.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <map>
#include <vector>

__global__ void probe(unsigned* out, long long spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static void run(hipStream_t st, const char* label, unsigned* d, int nwg) {
    std::vector<unsigned> h(2 * nwg);
    hipMemsetAsync(d, 0xff, 2 * nwg * 4, st);
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(64), 0, st, d, 40000ll);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, 2 * nwg * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    std::map<unsigned, int> per_xcc;
    for (int i = 0; i < nwg; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        if (cus.insert(key).second) per_xcc[xcc]++;
    }
    printf("%-28s distinct CUs %3zu  per XCC:", label, cus.size());
    for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
}

int main() {
    unsigned* d;
    const int nwg = 8192;
    hipMalloc(&d, 2 * nwg * 4);
    hipStream_t plain;
    hipStreamCreate(&plain);
    run(plain, "no mask", d, nwg);
    for (int k : {8, 16, 32, 64}) {
        unsigned mask[10];
        for (int variant = 0; variant < 2; ++variant) {
            memset(mask, 0, sizeof(mask));
            char label[64];
            if (variant == 0) { for (int i = 0; i < k; ++i) mask[i / 32] |= 1u << (i % 32); snprintf(label, sizeof(label), "first %d bits", k); }
            else { for (int i = 0; i < k; ++i) { const int b = 255 - i; mask[b / 32] |= 1u << (b % 32); } snprintf(label, sizeof(label), "bits 255-%d..255", k - 1); }
            hipStream_t st;
            if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("%s: create failed\n", label); continue; }
            run(st, label, d, nwg);
            hipStreamDestroy(st);
        }
    }
    return 0;
}
