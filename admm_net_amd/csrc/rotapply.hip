// rotapply.hip -- K3: V = Q W by replaying the QL rotation log on the rows of Q.
//
// A plane rotation of the QL sweep mixes two adjacent COLUMNS of the
// accumulated eigenvector matrix, so every ROW is independent: one thread owns
// one real row (re or im part of a row of Q) and keeps all n entries in
// registers for the whole replay -- zero LDS, zero cross-lane traffic, the
// rotation coefficients arrive as wave-uniform scalars (one aligned
// s_load_dwordx16 per group of 8 planes, SGPR operands of the VALU ops).
// The register array must be statically indexed, hence the replay is fully
// unrolled over the plane index in groups of 8; a sweep enters only the
// groups [g_lo, g_hi] named by its header, and the log pads the partial groups
// with identity rotations so that no per-plane predicate is needed.
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
    // bit 8 of the status word: the QL kernel worked on the reversed tridiagonal (T' = J T J),
    // i.e. on the rows of Q J -- the columns are loaded in reverse; the output order is unchanged.
    const bool flip = (__builtin_amdgcn_readfirstlane(logn[b * 2 + 1]) & 256) != 0;
    float z[NMAX];
#pragma unroll
    for (int j = 0; j < NMAX; ++j) {
        const int c = flip ? (D - 1 - j) : (j - 1);   // column of Q' feeding register j
        z[j] = (c >= 0 && c < D && valid) ? q[(int64_t)c * 2 * D + rho] : 0.f;
    }

    const int nrec = __builtin_amdgcn_readfirstlane(logn[b * 2]);
    const LogRec *lgb = log + b * cap;
    const int2 *lg = reinterpret_cast<const int2 *>(lgb);
    const float2 *lgf = reinterpret_cast<const float2 *>(lgb);
    int pos = 0;
    while (pos < nrec) {
        const int2 hdr = lg[pos];
        const int g_hi = __builtin_amdgcn_readfirstlane(hdr.x);
        const int g_lo = __builtin_amdgcn_readfirstlane(hdr.y);
        const float2 *rp = lgf + pos + 8;   // group g starts at rp + 8 * (g_hi - g)
#pragma unroll
        for (int g = NG - 1; g >= 0; --g) {
            if (g <= g_hi && g >= g_lo) {
                const float2 *gp = rp + 8 * (g_hi - g);
                float2 cs[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) cs[t] = gp[t];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int i = 8 * g + 7 - t;
                    if (i + 1 < NMAX) {   // static: planes above the register array hold identity
                        const float c = cs[t].x, s = cs[t].y;
                        const float f = z[i + 1];
                        const float zi = z[i];
                        z[i + 1] = s * zi + c * f;
                        z[i] = c * zi - s * f;
                    }
                }
            }
        }
        pos += 8 + 8 * (g_hi - g_lo + 1);
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
    ProfScope _prof(KC_ROTAPPLY, st);
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
