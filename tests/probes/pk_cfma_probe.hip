// developer probe: v_pk_fma_f32 with op_sel / neg modifiers as a 2-instruction complex multiply-accumulate
// (hipcc --offload-arch=gfx950 -O3 pk_cfma_probe.hip -o /tmp/pk && /tmp/pk)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_cfma(v2f acc, v2f a, v2f b) {   // acc + a b
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}
__device__ __forceinline__ v2f pk_cfma_conj(v2f acc, v2f a, v2f b) {   // acc + conj(a) b
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]"
        : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}
__global__ void k(const v2f* a, const v2f* b, v2f* o) {
    int t = threadIdx.x;
    v2f acc = {0.5f, -0.25f}, acc2 = {0.5f, -0.25f};
    acc = pk_cfma(acc, a[t], b[t]);
    acc2 = pk_cfma_conj(acc2, a[t], b[t]);
    o[t] = acc; o[t+64] = acc2;
}
int main() {
    v2f ha[64], hb[64], ho[128], *da, *db, *dout;
    for (int i = 0; i < 64; ++i) { ha[i] = v2f{0.3f * i - 7.f, 1.f + 0.11f * i}; hb[i] = v2f{2.f - 0.07f * i, 0.5f * i - 3.f}; }
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, sizeof(ho));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 64; ++i) {
        double ax = ha[i].x, ay = ha[i].y, bx = hb[i].x, by = hb[i].y;
        e1 = fmax(e1, fabs(ho[i].x - (0.5 + ax * bx - ay * by)) + fabs(ho[i].y - (-0.25 + ax * by + ay * bx)));
        e2 = fmax(e2, fabs(ho[64 + i].x - (0.5 + ax * bx + ay * by)) + fabs(ho[64 + i].y - (-0.25 + ax * by - ay * bx)));
    }
    printf("pk_cfma max err %.3g   pk_cfma_conj max err %.3g\n", e1, e2);
    return (e1 < 1e-4 && e2 < 1e-4) ? 0 : 1;
}
