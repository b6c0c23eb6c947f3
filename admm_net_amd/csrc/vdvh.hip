// vdvh.hip -- Hermitian eigen-function assembly  OUT = V diag(d) V^H  for the training route:
//   * GLayer._rebuild_definite_matrix, /root/reference/admm_net.py:336-354 (two bmm + the symmetrisation, here one pass:
//     the lower tiles are computed, the upper triangle is written as their conjugate, so OUT is exactly Hermitian);
//   * the backward of the eigenvalue-only eigh, dL/dA = V diag(dL/dw) V^H (admm_net.py:303-306: V is detached);
// and its adjoint, the quadratic forms  q_c = Re(v_c^H S v_c)  (gradient of OUT with respect to d for an incoming
// Hermitian S = (g + g^H) / 2).
// The inference path never comes here: it keeps V in LDS / in the planar image and fuses the eigenvalue network
// (rebuild_lds.h, rebuild_big.hip).  This entry takes what autograd has: V as torch.linalg.eigh lays it out
// (complex64 [B][n][n], columns = eigenvectors) and d as a float vector per matrix.
//
// One 256-thread workgroup per matrix, 32 x 32 output tiles dealt round-robin to the four waves, complex products as four
// real v_mfma_f32_32x32x2_f32 per k-step with the operands read in the matrix cores' lane layout (lane -> row of the
// tile, lane half -> k).  V of one matrix is 82 KB at the reference's n = 101 and 528 KB at n = 257: it stays in the L2
// across the tiles of its workgroup.
#include "common.h"

namespace admmnet {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int VD_THREADS = 256;

__global__ __launch_bounds__(VD_THREADS) void vdvh_kernel(int n, const float2 *__restrict__ Vg, const float *__restrict__ dg,
                                                          float2 *__restrict__ Og) {
    const int64_t b = blockIdx.x;
    const float2 *V = Vg + b * (int64_t)n * n;
    const float *d = dg + b * n;
    float2 *O = Og + b * (int64_t)n * n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r32 = lane & 31, kh = lane >> 5;
    const int nt = (n + 31) >> 5, ntri = nt * (nt + 1) / 2;
    for (int t = wave; t < ntri; t += VD_THREADS / 64) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int J = t - I * (I + 1) / 2;
        const int i = 32 * I + r32, j = 32 * J + r32;
        const bool iv = i < n, jv = j < n;
        const float2 *xi_p = V + (int64_t)(iv ? i : 0) * n, *yj_p = V + (int64_t)(jv ? j : 0) * n;
        f32x16 aRe = {0}, aIm = {0};
#pragma unroll 4
        for (int c0 = 0; c0 < n; c0 += 2) {
            const int c = c0 + kh;
            const bool cv = c < n;
            const int cc = cv ? c : 0;
            float2 x = xi_p[cc], y = yj_p[cc];
            const float dc = (cv && iv) ? d[cc] : 0.f;
            x.x *= dc;
            x.y *= dc;
            if (!(cv && jv)) y = make_float2(0.f, 0.f);
            // (x d) conj(y) = (xr yr + xi yi) + i (xi yr - xr yi)
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, aRe, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.x, aIm, 0, 0, 0);
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, aRe, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(-x.x, y.y, aIm, 0, 0, 0);
        }
        // C/D layout: column = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = 32 * I + (q & 3) + 8 * (q >> 2) + 4 * kh, gj = 32 * J + r32;
            if (gi < n && gj < n && gi >= gj) {
                if (gi == gj) {
                    O[(int64_t)gi * n + gj] = make_float2(aRe[q], 0.f);
                } else {
                    O[(int64_t)gi * n + gj] = make_float2(aRe[q], aIm[q]);
                    O[(int64_t)gj * n + gi] = make_float2(aRe[q], -aIm[q]);
                }
            }
        }
    }
}

// q[c] = Re(v_c^H S v_c) for a Hermitian S (only its lower triangle is read; the diagonal's imaginary part is ignored):
//   q_c = sum_i S_ii |V_ic|^2 + 2 Re sum_{i > j} conj(V_ic) S_ij V_jc.
// T = L V with L = strict lower triangle of S on the matrix cores (tile (I, Cb): sum over the block columns J <= I), then
// the row sums of Re(conj(V) .* T) per column c, doubled, plus the diagonal term; partial sums per (wave, tile) go to LDS
// and are added in a fixed order (no floating-point atomics: same bits every run).
__global__ __launch_bounds__(VD_THREADS) void vhsv_kernel(int n, const float2 *__restrict__ Vg, const float2 *__restrict__ Sg,
                                                          float *__restrict__ qg) {
    extern __shared__ float part[];   // [nt (row tiles)][nt * 32 (columns)]
    const int64_t b = blockIdx.x;
    const float2 *V = Vg + b * (int64_t)n * n, *S = Sg + b * (int64_t)n * n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r32 = lane & 31, kh = lane >> 5;
    const int nt = (n + 31) >> 5;
    for (int t = wave; t < nt * nt; t += VD_THREADS / 64) {
        const int I = t / nt, Cb = t - I * nt;
        const int i = 32 * I + r32, c = 32 * Cb + r32;
        const bool iv = i < n, cvv = c < n;
        f32x16 aRe = {0}, aIm = {0};
        const int jend = min(n, 32 * (I + 1));
#pragma unroll 4
        for (int j0 = 0; j0 < jend; j0 += 2) {
            const int j = j0 + kh;
            // A[m = i][k = j] = S_ij for j < i (strict lower triangle), B[k = j][n = c] = V_jc
            const bool on = iv && j < i;
            float2 s = S[(int64_t)(iv ? i : 0) * n + (j < n ? j : 0)];
            if (!on) s = make_float2(0.f, 0.f);
            float2 v = V[(int64_t)(j < n ? j : 0) * n + (cvv ? c : 0)];
            if (!(cvv && j < n)) v = make_float2(0.f, 0.f);
            // s v = (sr vr - si vi) + i (sr vi + si vr)
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(s.x, v.x, aRe, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(s.x, v.y, aIm, 0, 0, 0);
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(-s.y, v.y, aRe, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(s.y, v.x, aIm, 0, 0, 0);
        }
        // rows of this lane: gi = 32 I + (q & 3) + 8 (q >> 2) + 4 kh, column c; sum over its 16 rows, then the two lane halves
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = 32 * I + (q & 3) + 8 * (q >> 2) + 4 * kh;
            if (gi < n && cvv) {
                const float2 v = V[(int64_t)gi * n + c];
                acc = fmaf(v.x, aRe[q], fmaf(v.y, aIm[q], acc));            // Re(conj(v) t)
                // (the diagonal term S_ii |V_ic|^2 at half weight: the partial sum is doubled below)
                acc = fmaf(0.5f * S[(int64_t)gi * n + gi].x, fmaf(v.x, v.x, v.y * v.y), acc);
            }
        }
        acc += __shfl_xor(acc, 32, 64);
        if (kh == 0) part[I * (nt * 32) + 32 * Cb + r32] = 2.0f * acc;
    }
    __syncthreads();
    for (int c = tid; c < n; c += VD_THREADS) {
        float s = 0.f;
        for (int I = 0; I < nt; ++I) s += part[I * (nt * 32) + c];
        qg[b * n + c] = s;
    }
}

int launch_vdvh(int n, int64_t nb, const float2 *V, const float *d, float2 *out, hipStream_t st) {
    if (nb <= 0) return ADMMNET_OK;
    hipLaunchKernelGGL(vdvh_kernel, dim3((unsigned)nb), dim3(VD_THREADS), 0, st, n, V, d, out);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_vhsv(int n, int64_t nb, const float2 *V, const float2 *S, float *q, hipStream_t st) {
    if (nb <= 0) return ADMMNET_OK;
    const int nt = (n + 31) >> 5;
    const size_t lds = sizeof(float) * nt * nt * 32;
    hipLaunchKernelGGL(vhsv_kernel, dim3((unsigned)nb), dim3(VD_THREADS), lds, st, n, V, S, q);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
