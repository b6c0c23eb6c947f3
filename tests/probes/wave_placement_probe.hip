// developer probe: on which SIMD do the waves of small workgroups land?  (hipcc --offload-arch=gfx950 -O3 ... && ./a.out)
// 2-wave and 4-wave workgroups with enough LDS that 4 / 2 of them share a CU, as tridiag_panel's stages 3 / 2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out, int spin) {
    extern __shared__ char lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    volatile float x = 1.f;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.1f;   // stay resident for a while so that the CU fills up
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = hw;
    if (x == 123.f) lds[threadIdx.x] = 1;
}
int main() {
    for (int nw : {2, 4}) {
        const int nwg = 4096, lds = nw == 2 ? 36 * 1024 : 72 * 1024;
        unsigned *d;
        hipMalloc(&d, nwg * nw * sizeof(unsigned));
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(probe, dim3(nwg), dim3(64 * nw), lds, 0, d, 20000);
        std::vector<unsigned> h(nwg * nw);
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
        int cnt[4][4] = {};
        for (int b = 0; b < nwg; ++b)
            for (int w = 0; w < nw; ++w) cnt[w][(h[b * nw + w] >> 4) & 3]++;
        printf("%d-wave workgroups: SIMD histogram per wave index\n", nw);
        for (int w = 0; w < nw; ++w) printf("  wave %d: %5d %5d %5d %5d\n", w, cnt[w][0], cnt[w][1], cnt[w][2], cnt[w][3]);
        hipFree(d);
    }
    return 0;
}
