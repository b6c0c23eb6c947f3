// eig_core.h -- scalar numerical cores shared by the HIP kernels and the
// host-side unit-test model (tests/host_model).  Everything here is
// __host__ __device__ and free of thread cooperation.
//
// The G-layer of the reference calls torch.linalg.eigh on a dense Hermitian
// matrix (/root/reference/admm_net.py:292-308).  We split that into
//   (1) Householder tridiagonalisation A = Q T Q^H        (tridiag.hip)
//   (2) implicit-shift QL on T, one matrix per lane, that RECORDS its plane
//       rotations instead of applying them                 (tql_lane below)
//   (3) replay of the rotation log on the rows of Q held in registers
//                                                          (rotapply.hip)
// so that V = Q W never needs the eigenvectors of T to be formed separately.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

namespace admmnet {

// One 8-byte record of the rotation log.  A QL sweep writes a header
// {start plane i0 = m-1, count} followed by `count` rotations (c, s) for the
// planes i0, i0-1, ..., i0-count+1 (plane i mixes columns i and i+1).
struct LogRec {
    union {
        struct { float c, s; } r;
        struct { int32_t i0, cnt; } h;
    };
};

constexpr float kEps32 = 5.9604645e-08f;  // 2^-24, unit roundoff of binary32

HD float sign_of(float a, float b) { return b >= 0.f ? fabsf(a) : -fabsf(a); }

// Complex Householder generator, LAPACK clarfg semantics:
//   H = I - tau * v v^H, v = [1; x*scale],  H^H [alpha; x] = [beta; 0], beta real.
// Given alpha (ar, ai) and xnorm2 = ||x||^2, returns beta, tau and the scale
// to apply to x.  tau == 0 means H = I.
HD void householder_c(float ar, float ai, float xnorm2, float &beta, float &tr, float &ti,
                      float &sr, float &si) {
    if (xnorm2 == 0.f && ai == 0.f) {
        beta = ar; tr = 0.f; ti = 0.f; sr = 0.f; si = 0.f;
        return;
    }
    float nrm = sqrtf(ar * ar + ai * ai + xnorm2);
    beta = -sign_of(nrm, ar);
    tr = (beta - ar) / beta;
    ti = -ai / beta;
    // scale = 1 / (alpha - beta)
    float dr = ar - beta, di = ai;
    float den = dr * dr + di * di;
    sr = dr / den;
    si = -di / den;
}

// Implicit-shift QL (EISPACK tql2 / Numerical Recipes tqli organisation) on a
// real symmetric tridiagonal matrix; d[0..n-1] diagonal, e[i] couples i,i+1.
// Accessors: D(i), E(i) return references; Z0(i) is a length-n row vector that
// receives the same rotations (pass the first row of the identity to obtain
// the first row of the eigenvector matrix W).  `emit(rec)` appends one LogRec
// and returns false on overflow.  Returns 0 ok, 1 no convergence, 2 overflow.
template <class DA, class EA, class ZA, class Emit, class Patch>
HD int tql_lane(int n, DA D, EA E, ZA Z0, Emit emit, Patch patch, int max_sweeps, int &nsweeps) {
    nsweeps = 0;
    for (int l = 0; l < n; ++l) {
        int iter = 0;
        int m;
        do {
            for (m = l; m < n - 1; ++m) {
                float dd = fabsf(D(m)) + fabsf(D(m + 1));
                if (fabsf(E(m)) <= kEps32 * dd) break;
            }
            if (m != l) {
                if (iter++ >= max_sweeps) return 1;
                float g = (D(l + 1) - D(l)) / (2.0f * E(l));
                float r = sqrtf(g * g + 1.0f);
                g = D(m) - D(l) + E(l) / (g + sign_of(r, g));
                float s = 1.0f, c = 1.0f, p = 0.0f;
                int hdr = patch(-1, 0, 0);   // reserve header slot
                if (hdr < 0) return 2;
                int cnt = 0;
                int i;
                bool brk = false;
                for (i = m - 1; i >= l; --i) {
                    float f = s * E(i);
                    float b = c * E(i);
                    r = sqrtf(f * f + g * g);
                    E(i + 1) = r;
                    if (r == 0.0f) {
                        D(i + 1) -= p;
                        E(m) = 0.0f;
                        brk = true;
                        break;
                    }
                    s = f / r;
                    c = g / r;
                    g = D(i + 1) - p;
                    r = (D(i) - g) * s + 2.0f * c * b;
                    p = s * r;
                    D(i + 1) = g + p;
                    g = c * r - b;
                    // rotation in plane (i, i+1)
                    float zf = Z0(i + 1);
                    float zi = Z0(i);
                    Z0(i + 1) = s * zi + c * zf;
                    Z0(i) = c * zi - s * zf;
                    LogRec rec;
                    rec.r.c = c;
                    rec.r.s = s;
                    if (!emit(rec)) return 2;
                    ++cnt;
                }
                patch(hdr, m - 1, cnt);
                ++nsweeps;
                if (brk) continue;
                D(l) -= p;
                E(l) = g;
                E(m) = 0.0f;
            }
        } while (m != l);
    }
    return 0;
}

// torch.nn.functional.softplus (beta = 1, threshold = 20)
HD float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
HD float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace admmnet
