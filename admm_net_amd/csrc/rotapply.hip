// rotapply.hip -- K3: V = Q W by replaying the QL rotation log on the rows of Q.
//
// A plane rotation of the QL sweep mixes two adjacent COLUMNS of the
// accumulated eigenvector matrix, so every ROW is independent: one thread owns
// one real row (re or im part of a row of Q) and keeps all n entries in
// registers for the whole replay -- zero LDS, zero cross-lane traffic, the
// rotation coefficients arrive as wave-uniform scalars (SGPR operands of the
// v_fma).  The register array must be statically indexed, hence the replay
// is fully unrolled over the plane index in groups of 8 and a sweep enters
// only the groups its window [l, i0] touches.
// This is the "apply Givens rotations" half of csteqr for the eigenvector
// matrix consumed by /root/reference/admm_net.py:303,349.
#include "common.h"

namespace admmnet {

template <int NMAX>
__global__ __launch_bounds__(64) void rotapply_kernel(int D, float *__restrict__ QV,
                                                      const LogRec *__restrict__ log,
                                                      const int *__restrict__ logn, int64_t cap) {
    constexpr int NG = (NMAX - 1 + 7) / 8;   // plane groups; planes 0 .. NMAX-2
    const int n = D + 1;
    const int64_t b = blockIdx.x;
    const int rho = blockIdx.y * 64 + threadIdx.x;
    const bool valid = rho < 2 * D;
    float *q = QV + b * ((int64_t)n * 2 * D);
    float z[NMAX];
    z[0] = 0.f;
#pragma unroll
    for (int c = 0; c < NMAX - 1; ++c) z[c + 1] = (c < D && valid) ? q[(int64_t)c * 2 * D + rho] : 0.f;

    const int nrec = __builtin_amdgcn_readfirstlane(logn[b * 2]);
    const LogRec *lgb = log + b * (cap + 16) + 8;
    const int2 *lg = reinterpret_cast<const int2 *>(lgb);
    const float2 *lgf = reinterpret_cast<const float2 *>(lgb);
    int pos = 0;
    while (pos < nrec) {
        const int2 hdr = lg[pos];
        const int i0 = __builtin_amdgcn_readfirstlane(hdr.x);
        const int cnt = __builtin_amdgcn_readfirstlane(hdr.y);
        const int l = i0 - cnt + 1;
        const float2 *rp = lgf + pos + 1 + i0;   // rotation of plane i is rp[-i]
#pragma unroll
        for (int g = NG - 1; g >= 0; --g) {
            if (8 * g <= i0 && 8 * g + 7 >= l) {
                float2 cs[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) cs[t] = rp[-(8 * g + t)];
#pragma unroll
                for (int t = 7; t >= 0; --t) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int i = 8 * g + t;
                    if (i + 1 < NMAX) {
                        if (i <= i0 && i >= l) {
                            const float c = cs[t].x, s = cs[t].y;
                            const float f = z[i + 1];
                            const float zi = z[i];
                            z[i + 1] = s * zi + c * f;
                            z[i] = c * zi - s * f;
                        }
                    }
                }
            }
        }
        pos += 1 + cnt;
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < NMAX; ++c)
            if (c < n) q[(int64_t)c * 2 * D + rho] = z[c];
    }
}

template <int NMAX>
static int launch_ra(int D, int64_t nb, const Ws &ws, hipStream_t st) {
    dim3 grid((unsigned)nb, (unsigned)((2 * D + 63) / 64));
    hipLaunchKernelGGL(rotapply_kernel<NMAX>, grid, dim3(64), 0, st, D, ws.QV, ws.log, ws.logn, ws.cap);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_rotapply(int D, int64_t nb, const Ws &ws, hipStream_t st) {
    if (nb <= 0) return ADMMNET_OK;
    const int n = D + 1;
    if (n <= 17) return launch_ra<17>(D, nb, ws, st);
    if (n <= 65) return launch_ra<65>(D, nb, ws, st);
    if (n <= 101) return launch_ra<101>(D, nb, ws, st);
    if (n <= 129) return launch_ra<129>(D, nb, ws, st);
    if (n <= 257) return launch_ra<257>(D, nb, ws, st);
    set_error("rotapply: n=%d unsupported", n);
    return ADMMNET_E_ARG;
}

}  // namespace admmnet
