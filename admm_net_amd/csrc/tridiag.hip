// tridiag.hip -- K1: batched Householder tridiagonalisation + explicit Q.
//
// One workgroup per matrix.  Input is the Hermitian matrix in "arrow first"
// storage written by the prep kernel: corner c0 (real), arrow column a[D]
// (= A[1:,0]) and the trailing block M[D][D] (full storage, row-major).
// Output: T as d[n], e[n-1] (lane-transposed for the QL kernel) and the
// explicit unitary Q' (D x D, A = diag(1,Q') T diag(1,Q')^H) in the planar
// transposed layout QT[c][rho] that the rotation-replay kernel reads coalesced.
//
// Replaces the first half of torch.linalg.eigh at /root/reference/admm_net.py:303
// (LAPACK chetrd + cungtr semantics: clarfg reflectors, A <- H^H A H).
//
// For D <= 128 the D x D block lives in LDS (128 x 130 x 8 B = 133 KB of the
// 160 KB gfx950 LDS); every access walks a column of the row-major image with
// consecutive lanes on consecutive columns, so ds_read_b64 is conflict-free.
// Larger D uses the same code on the global (L2 / Infinity Cache) image.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace admmnet {

constexpr int TD_THREADS = 512;
constexpr int TD_PARTS = 4;
constexpr int TD_CW = 128;

__device__ __forceinline__ float2 td_block_sum2(float2 v, float2 *scr, int &flip) {
    v.x = wave_sum(v.x);
    v.y = wave_sum(v.y);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float2 *s = scr + flip * 8;
    flip ^= 1;
    if (lane == 0) s[wave] = v;
    __syncthreads();
    float2 r = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < TD_THREADS / 64; ++i) {
        float2 t = s[i];
        r.x += t.x;
        r.y += t.y;
    }
    return r;
}

template <bool LDSM>
__global__ __launch_bounds__(TD_THREADS) void tridiag_kernel(int D, float2 *__restrict__ Mbuf,
                                                             float *__restrict__ QV,
                                                             float *__restrict__ dT,
                                                             float *__restrict__ eT) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int cl = tid & (TD_CW - 1), part = tid / TD_CW;
    const int64_t b = blockIdx.x;
    const int n = D + 1;
    const int Dp = (D + 3) & ~3;
    float2 *Mg = Mbuf + b * ((int64_t)D * D + D + 1);
    const float2 *ag = Mg + (int64_t)D * D;

    // LDS carve
    float2 *v = reinterpret_cast<float2 *>(smem);
    float2 *w = v + Dp;
    float2 *v0s = w + Dp;
    float2 *taus = v0s + Dp;
    float2 *red = taus + Dp;                 // [4][Dp]
    float2 *scr = red + TD_PARTS * Dp;       // [2][8]
    float2 *Ms = scr + 16;
    const int P = LDSM ? (D + 2) : D;
    float2 *M = LDSM ? Ms : Mg;
    int flip = 0;

    if (LDSM) {
        for (int idx = tid; idx < D * D; idx += TD_THREADS) {
            int i = idx / D, j = idx - i * D;
            Ms[i * P + j] = Mg[idx];
        }
    }
    for (int i = tid; i < D; i += TD_THREADS) v0s[i] = ag[i];
    __syncthreads();

    float *dcol = dT + b * n;
    float *ecol = eT + b * n;

    for (int r = 0; r < D; ++r) {
        // ---- reflector r: from the arrow (r == 0) or column r-1 of M, rows r..D-1
        float2 alpha;
        float2 pn = make_float2(0.f, 0.f);
        if (r == 0) {
            alpha = v0s[0];
            for (int i = 1 + tid; i < D; i += TD_THREADS) {
                float2 x = v0s[i];
                pn.x += x.x * x.x + x.y * x.y;
            }
        } else {
            alpha = M[r * P + r - 1];
            for (int i = r + 1 + tid; i < D; i += TD_THREADS) {
                float2 x = M[i * P + r - 1];
                pn.x += x.x * x.x + x.y * x.y;
            }
        }
        const float xn2 = td_block_sum2(pn, scr, flip).x;
        float beta, tr, ti, sr, si;
        householder_c(alpha.x, alpha.y, xn2, beta, tr, ti, sr, si);
        const float2 tau = make_float2(tr, ti);
        const float2 sc = make_float2(sr, si);
        if (tid == 0) {
            ecol[r] = beta;
            dcol[r] = (r == 0) ? ag[D].x : M[(r - 1) * P + r - 1].x;
            taus[r] = tau;
        }
        for (int i = r + tid; i < D; i += TD_THREADS) {
            float2 vi;
            if (i == r) {
                vi = make_float2(1.f, 0.f);
            } else {
                float2 x = (r == 0) ? v0s[i] : M[i * P + r - 1];
                vi = cmul(x, sc);
                if (r > 0) M[i * P + r - 1] = vi;
            }
            v[i] = vi;
            if (r == 0) v0s[i] = vi;
        }
        __syncthreads();
        if (tr == 0.f && ti == 0.f) continue;   // H = I (uniform)

        // ---- p = tau * M v over indices r..D-1 (column walk, M Hermitian)
        for (int cb = 0; cb < D; cb += TD_CW) {
            const int col = cb + cl;
            if (col >= r && col < D) {
                float2 acc = make_float2(0.f, 0.f);
                for (int j = r + part; j < D; j += TD_PARTS) acc = cmacc(acc, M[j * P + col], v[j]);
                red[part * Dp + col] = acc;
            }
        }
        __syncthreads();
        float2 dotp = make_float2(0.f, 0.f);
        for (int i = r + tid; i < D; i += TD_THREADS) {
            float2 s = red[i];
#pragma unroll
            for (int q = 1; q < TD_PARTS; ++q) {
                float2 t = red[q * Dp + i];
                s.x += t.x;
                s.y += t.y;
            }
            float2 p = cmul(tau, s);
            w[i] = p;
            dotp = cmacc(dotp, p, v[i]);   // conj(p) * v
        }
        const float2 dot = td_block_sum2(dotp, scr, flip);
        float2 al = cmul(tau, dot);
        al.x *= -0.5f;
        al.y *= -0.5f;
        for (int i = r + tid; i < D; i += TD_THREADS) {
            float2 p = w[i];
            float2 t = cmul(al, v[i]);
            w[i] = make_float2(p.x + t.x, p.y + t.y);
        }
        __syncthreads();
        // ---- M -= v w^H + w v^H on indices r..D-1
        for (int cb = 0; cb < D; cb += TD_CW) {
            const int col = cb + cl;
            if (col >= r && col < D) {
                const float2 vc = v[col], wc = w[col];
                for (int j = r + part; j < D; j += TD_PARTS) {
                    const float2 vj = v[j], wj = w[j];
                    float2 m = M[j * P + col];
                    float2 t1 = cmulc(vj, wc), t2 = cmulc(wj, vc);
                    m.x -= t1.x + t2.x;
                    m.y -= t1.y + t2.y;
                    if (j == col) m.y = 0.f;
                    M[j * P + col] = m;
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        dcol[D] = M[(D - 1) * P + D - 1].x;
        ecol[D] = 0.f;
    }
    __syncthreads();

    // ---- explicit Q' = H_0 H_1 ... H_{D-1}, accumulated backwards in place
    for (int r = D - 1; r >= 0; --r) {
        const float2 tau = taus[r];
        for (int i = r + 1 + tid; i < D; i += TD_THREADS) v[i] = (r >= 1) ? M[i * P + r - 1] : v0s[i];
        __syncthreads();
        for (int cb = 0; cb < D; cb += TD_CW) {
            const int col = cb + cl;
            if (col > r && col < D) {
                float2 acc = make_float2(0.f, 0.f);
                for (int i = r + 1 + part; i < D; i += TD_PARTS) acc = cmacc(acc, v[i], M[i * P + col]);
                red[part * Dp + col] = acc;
            }
        }
        __syncthreads();
        for (int cb = 0; cb < D; cb += TD_CW) {
            const int col = cb + cl;
            if (col > r && col < D) {
                float2 z = red[col];
#pragma unroll
                for (int q = 1; q < TD_PARTS; ++q) {
                    float2 t = red[q * Dp + col];
                    z.x += t.x;
                    z.y += t.y;
                }
                const float2 tz = cmul(tau, z);
                for (int i = r + 1 + part; i < D; i += TD_PARTS) {
                    float2 t = cmul(v[i], tz);
                    float2 m = M[i * P + col];
                    m.x -= t.x;
                    m.y -= t.y;
                    M[i * P + col] = m;
                }
                if (part == 0) M[r * P + col] = make_float2(-tz.x, -tz.y);
            }
        }
        for (int i = tid; i < D; i += TD_THREADS) {
            float2 q;
            if (i < r) {
                q = make_float2(0.f, 0.f);
            } else if (i == r) {
                q = make_float2(1.f - tau.x, -tau.y);
            } else {
                float2 t = cmul(tau, v[i]);
                q = make_float2(-t.x, -t.y);
            }
            M[i * P + r] = q;
        }
        __syncthreads();
    }

    // ---- write QT[c][rho] (rho = row for re, D + row for im)
    float *q = QV + b * ((int64_t)n * 2 * D);
    for (int idx = tid; idx < D * D; idx += TD_THREADS) {
        int c = idx / D, rr = idx - c * D;
        float2 m = M[rr * P + c];
        q[(int64_t)c * 2 * D + rr] = m.x;
        q[(int64_t)c * 2 * D + D + rr] = m.y;
    }
}

static size_t td_lds_bytes(int D, bool ldsm) {
    const int Dp = (D + 3) & ~3;
    size_t sz = sizeof(float2) * ((size_t)Dp * (4 + TD_PARTS) + 16);
    if (ldsm) sz += sizeof(float2) * (size_t)D * (D + 2);
    return sz;
}

int launch_tridiag(int D, int64_t nb, const Ws &ws, hipStream_t st, const float2 *Zlow, const float2 *phi,
                   const float *h, const float *lw) {
    ProfScope _prof(KC_TRIDIAG, st);
    if (D < 1 || D > kMaxD) {
        set_error("tridiag: D=%d unsupported (1..%d)", D, kMaxD);
        return ADMMNET_E_ARG;
    }
    if (nb <= 0) return ADMMNET_OK;
    // D <= 128: register-resident kernel (tridiag_reg.hip).  ADMMNET_TRIDIAG=lds keeps the
    // LDS-resident version below selectable for A/B runs; it also serves 128 < D <= 256 (global image).
    // (A 512-thread, 4-waves-per-SIMD variant of tridiag_reg was measured 1.6x SLOWER: the O(n) per-wave
    // work of every reflector -- norm, Householder scalars, vector set-up, reductions -- is replicated in
    // each wave, and with 8 waves per matrix it outweighs the better latency hiding.)
    static int use_lds = -1;
    if (use_lds < 0) {
        const char *e = getenv("ADMMNET_TRIDIAG");
        use_lds = (e && !strcmp(e, "lds")) ? 1 : 0;
    }
    if (D <= 128 && !use_lds) return launch_tridiag_reg(D, nb, ws, st, Zlow, phi, h, lw);
    if (Zlow) {
        set_error("tridiag: the lean loader exists for the register-resident kernel only (D <= 128)");
        return ADMMNET_E_ARG;
    }
    if (D <= 256 && !use_lds) return launch_tridiag_big(D, nb, ws, st);
    const bool ldsm = td_lds_bytes(D, true) <= 160 * 1024;
    const size_t lds = td_lds_bytes(D, ldsm);
    if (ldsm) {
        ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tridiag_kernel<true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(tridiag_kernel<true>, dim3((unsigned)nb), dim3(TD_THREADS), lds, st, D, ws.Mbuf,
                           ws.QV, ws.dT, ws.eT);
    } else {
        hipLaunchKernelGGL(tridiag_kernel<false>, dim3((unsigned)nb), dim3(TD_THREADS), lds, st, D, ws.Mbuf,
                           ws.QV, ws.dT, ws.eT);
    }
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
