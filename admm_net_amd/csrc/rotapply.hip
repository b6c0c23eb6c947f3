// rotapply.hip -- K3: V = Q W by replaying the QL rotation log on the rows of Q.
//
// A plane rotation of the QL sweep mixes two adjacent COLUMNS of the
// accumulated eigenvector matrix, so every ROW is independent: one thread owns
// one real row (re or im part of a row of Q) and keeps all n entries in
// registers for the whole replay -- zero LDS, zero cross-lane traffic, the
// rotation coefficients arrive as wave-uniform scalars (one aligned
// s_load_dwordx16 per 64-byte log group, SGPR operands of the VALU ops).
// The register array must be statically indexed, so a sweep walks a fully
// unrolled chain over the group number and enters the groups [g_lo, g_hi] its
// header names; the log pads partial groups with identity rotations, so there is no per-plane
// predicate.  The log is consumed as a flat stream of groups (sweep headers
// included) with the next two groups always in flight in SGPRs, which hides
// the scalar-load latency at 3 waves per SIMD.  (A switch(g) inside a loop was
// tried instead of the unrolled if-chain: the merge of 129 live values doubled
// the VGPR count and cost two thirds of the occupancy.)
// This is the "apply Givens rotations" half of csteqr for the eigenvector
// matrix consumed by /root/reference/admm_net.py:303,349.
#include "common.h"

namespace admmnet {

struct Grp {   // one 64-byte log group: 8 records (c, s) or a header in record 0
    float2 r[8];
};

template <int NMAX>
__device__ __forceinline__ void rot8(float (&z)[NMAX], const Grp &q, const int G) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        constexpr int dummy = 0;
        (void)dummy;
        const int i = 8 * G + 7 - t;
        if (i + 1 < NMAX) {   // static: planes above the register array hold identity
            const float c = q.r[t].x, s = q.r[t].y;
            const float f = z[i + 1];
            const float zi = z[i];
            z[i + 1] = fmaf(s, zi, c * f);
            z[i] = fmaf(c, zi, -(s * f));
        }
    }
}

template <int NMAX>
__global__ __launch_bounds__(64) void rotapply_kernel(int D, float *__restrict__ QV,
                                                      const LogRec *__restrict__ log,
                                                      const int *__restrict__ logn, int64_t cap) {
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

    constexpr int NG = (NMAX - 1 + 7) / 8;   // plane groups; planes 0 .. NMAX-2
    const int ngrp = __builtin_amdgcn_readfirstlane(logn[b * 2]) >> 3;
    const Grp *lg = reinterpret_cast<const Grp *>(log + b * cap);
    if (ngrp > 0) {
        const int last = ngrp - 1;
        // flat stream of 64-byte groups with the next two always in flight
        Grp cur = lg[0], n1 = lg[min(1, last)], n2 = lg[min(2, last)];
        int idx = 0;
        while (idx < ngrp) {
            // `cur` is a sweep header
            const int g_hi = __float_as_int(cur.r[0].x);
            const int g_lo = __float_as_int(cur.r[0].y);
            cur = n1;
            n1 = n2;
            n2 = lg[min(idx + 3, last)];
            ++idx;
#pragma unroll
            for (int g = NG - 1; g >= 0; --g) {
                if (g <= g_hi && g >= g_lo) {
                    const Grp n3 = lg[min(idx + 3, last)];
                    rot8<NMAX>(z, cur, g);
                    cur = n1;
                    n1 = n2;
                    n2 = n3;
                    ++idx;
                }
            }
        }
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
