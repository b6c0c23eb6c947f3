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

// One 8-byte record of the rotation log.  Records come in 64-byte GROUPS of 8 so that the
// replay kernel fetches one group with a single aligned s_load_dwordx16.  A QL sweep over the
// planes i0 >= i >= l (plane i mixes columns i and i+1) writes
//   1 header group : record 0 = {g_hi = i0 >> 3, g_lo = l >> 3}, records 1..7 unused
//   g_hi - g_lo + 1 rotation groups, highest planes first: group g holds the planes
//     8g+7, 8g+6, ..., 8g in that order; planes outside [l, i0] (or skipped by the underflow
//     exit of the sweep) hold the identity rotation (c, s) = (1, 0).
struct LogRec {
    union {
        struct { float c, s; } r;
        struct { int32_t g_hi, g_lo; } h;
    };
};

// Log writer over a flat LogRec array (host model and device share it).
struct LogWriter {
    LogRec *lg;
    int cap;   // records available
    int pos;   // next free record (multiple of 8)
    int cur;   // next rotation slot of the open sweep
    int endp;  // one past the last slot of the open sweep
    HD bool begin(int i0, int l) {
        const int g_hi = i0 >> 3, g_lo = l >> 3;
        const int need = 8 + 8 * (g_hi - g_lo + 1);
        if (pos + need > cap) return false;
        LogRec h;
        h.h.g_hi = g_hi;
        h.h.g_lo = g_lo;
        lg[pos] = h;
        cur = pos + 8;
        endp = pos + need;
        pos = endp;
        LogRec id;
        id.r.c = 1.0f;
        id.r.s = 0.0f;
        for (int i = 8 * g_hi + 7; i > i0; --i) lg[cur++] = id;   // planes above the window
        return true;
    }
    HD void rot(float c, float s) {
        LogRec r;
        r.r.c = c;
        r.r.s = s;
        lg[cur++] = r;
    }
    HD void end() {   // identity for whatever is left (below the window / after an early exit)
        LogRec id;
        id.r.c = 1.0f;
        id.r.s = 0.0f;
        while (cur < endp) lg[cur++] = id;
    }
};

constexpr float kEps32 = 5.9604645e-08f;  // 2^-24, unit roundoff of binary32

HD float sign_of(float a, float b) { return b >= 0.f ? fabsf(a) : -fabsf(a); }

// Complex Householder generator, LAPACK clarfg semantics:
//   H = I - tau * v v^H, v = [1; x*scale],  H^H [alpha; x] = [beta; 0], beta real.
// Given alpha (ar, ai) and xnorm2 = ||x||^2, returns beta, tau and the scale
// to apply to x.  tau == 0 means H = I.
// 1 / x: v_rcp_f32 plus one Newton step on the device (well under 1 ulp; the IEEE division sequence
// costs ~10 instructions and householder_c sits on the serial path of every reflector)
HD float recip_nr(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
#else
    return 1.0f / x;
#endif
}

// 1 / sqrt(x): v_rsq_f32 plus one Newton step on the device
HD float rsqrt_nr1(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rsqf(x);
    return r * fmaf(-0.5f * x * r, r, 1.5f);
#else
    return 1.0f / sqrtf(x);
#endif
}

HD void householder_c(float ar, float ai, float xnorm2, float &beta, float &tr, float &ti,
                      float &sr, float &si) {
    if (xnorm2 == 0.f && ai == 0.f) {
        beta = ar; tr = 0.f; ti = 0.f; sr = 0.f; si = 0.f;
        return;
    }
    const float q2 = ar * ar + ai * ai + xnorm2;
#if defined(__HIP_DEVICE_COMPILE__)
    // sqrt through v_rsq_f32 + one Newton step (the IEEE sqrt sequence is ~15 dependent instructions on the
    // serial path of every reflector); outside the safe exponent range the exact routine
    float nrm = (q2 > 1e-30f && q2 < 1e30f) ? q2 * rsqrt_nr1(q2) : sqrtf(q2);
#else
    float nrm = sqrtf(q2);
#endif
    beta = -sign_of(nrm, ar);
    const float ib = recip_nr(beta);
    tr = (beta - ar) * ib;
    ti = -ai * ib;
    // scale = 1 / (alpha - beta)
    float dr = ar - beta, di = ai;
    const float iden = recip_nr(dr * dr + di * di);
    sr = dr * iden;
    si = -di * iden;
}

// Implicit-shift QL (EISPACK tql2 / Numerical Recipes tqli organisation) on a
// real symmetric tridiagonal matrix; d[0..n-1] diagonal, e[i] couples i,i+1.
// Accessors: D(i), E(i) return references; Z0(i) is a length-n row vector that
// receives the same rotations (pass the first row of the identity to obtain
// the first row of the eigenvector matrix W).  `log` is a LogWriter-like policy
// (begin / rot / end).  Returns 0 ok, 1 no convergence, 2 log overflow.
template <class DA, class EA, class ZA, class Log>
HD int tql_lane(int n, DA D, EA E, ZA Z0, Log &log, int max_sweeps, int &nsweeps) {
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
                if (!log.begin(m - 1, l)) return 2;
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
                    log.rot(c, s);
                }
                log.end();
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

// 1/sqrt(x): one v_rsq_f32 (1 ulp) on the device, 1/sqrtf on the host.
HD float rsqrt_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / sqrtf(x);
#endif
}

// Bit set over the off-diagonal indices: bit j <=> e[j] is negligible.  Lets the QL driver find
// the end of the current unreduced block in O(1) instead of re-scanning e[l..] before every
// sweep (that scan is as long as the sweep itself).  Up to 64*MW indices.
template <int MW>
struct NegMask {
    uint64_t w[MW];
    HD void clear() {
#pragma unroll
        for (int k = 0; k < MW; ++k) w[k] = 0;
    }
    HD void put(int j, bool v) {   // branch free and fully unrolled so that w[] stays in registers
        const uint64_t bit = (uint64_t)1 << (j & 63);
        const int wi = j >> 6;
#pragma unroll
        for (int k = 0; k < MW; ++k) {
            const uint64_t sel = (k == wi) ? bit : 0;
            w[k] = (w[k] & ~sel) | (v ? sel : 0);
        }
    }
    HD int first_from(int l) const {   // smallest set index >= l (one always exists: the sentinel)
        int best = 64 * MW;
#pragma unroll
        for (int k = MW - 1; k >= 0; --k) {
            const int lo = 64 * k;
            uint64_t x = w[k];
            if (l > lo) x = (l - lo >= 64) ? 0 : (x & (~(uint64_t)0 << (l - lo)));
            if (x) {
#if defined(__HIP_DEVICE_COMPILE__)
                best = lo + (int)__builtin_ctzll(x);
#else
                best = lo + __builtin_ctzll(x);
#endif
            }
        }
        return best;
    }
};

// Same algorithm as tql_lane, organised for the device:
//  * the rotation loop carries (s, c, p, g) only through registers; d[i], e[i] of the NEXT plane
//    are fetched one iteration ahead and the tracked eigenvector row keeps its running element in
//    a register, so no LDS access sits on the serial chain;
//  * sqrt + two divisions become one reciprocal square root followed by a first-order
//    renormalisation of (c, s): the plane rotation stays orthogonal to O(eps^2), which matters
//    because a biased |c|^2+|s|^2 accumulates linearly over the ~3n rotations a row sees;
//  * the deflation test of every new off-diagonal is made inside the sweep and kept in a bit
//    set (NegMask), replacing the O(n) scan before each sweep.
template <int MW, class DA, class EA, class ZA, class Log>
HD int tql_lane_pf(int n, DA D, EA E, ZA Z0, Log &log, int max_sweeps, int &nsweeps) {
    nsweeps = 0;
    NegMask<MW> neg;
    neg.clear();
    for (int j = 0; j < n - 1; ++j) neg.put(j, fabsf(E(j)) <= kEps32 * (fabsf(D(j)) + fabsf(D(j + 1))));
    neg.put(n - 1, true);   // sentinel: the block always ends at n-1
    for (int l = 0; l < n; ++l) {
        int iter = 0;
        for (;;) {
            const int m = neg.first_from(l);
            if (m == l) break;
            if (iter++ >= max_sweeps) return 1;
            const float dl = D(l), el = E(l);
            float g = (D(l + 1) - dl) / (2.0f * el);
            float r = sqrtf(g * g + 1.0f);
            g = D(m) - dl + el / (g + sign_of(r, g));
            float s = 1.0f, c = 1.0f, p = 0.0f;
            if (!log.begin(m - 1, l)) return 2;
            int i = m - 1;
            float e_i = E(i), d_i = D(i), d_ip1 = D(m);
            float zc = Z0(m);        // running element z[i+1]
            float dnew_up = 0.0f;    // new d[i+2]
            bool brk = false;
            for (; i >= l; --i) {
                const int ip = (i > l) ? i - 1 : i;   // clamped prefetch index
                const float e_nx = E(ip), d_nx = D(ip);
                const float zi = Z0(i);
                const float f = s * e_i, b = c * e_i;
                const float rr = f * f + g * g;
                if (rr == 0.0f) {      // underflow recovery of the textbook algorithm
                    E(i + 1) = 0.0f;
                    D(i + 1) = d_ip1 - p;
                    E(m) = 0.0f;
                    neg.put(i + 1, true);
                    brk = true;
                    break;
                }
                const float rinv = rsqrt_fast(rr);
                const float e_new = rr * rinv;   // new e[i+1]
                E(i + 1) = e_new;
                float s0 = f * rinv, c0 = g * rinv;
                const float h = 0.5f * fmaf(-s0, s0, fmaf(-c0, c0, 1.0f));
                s = fmaf(h, s0, s0);
                c = fmaf(h, c0, c0);
                g = d_ip1 - p;
                const float t = (d_i - g) * s + 2.0f * c * b;
                p = s * t;
                const float dnew = g + p;        // new d[i+1]
                D(i + 1) = dnew;
                g = c * t - b;
                if (i + 1 < m) neg.put(i + 1, fabsf(e_new) <= kEps32 * (fabsf(dnew) + fabsf(dnew_up)));
                dnew_up = dnew;
                Z0(i + 1) = s * zi + c * zc;
                zc = c * zi - s * zc;
                log.rot(c, s);
                d_ip1 = d_i;
                e_i = e_nx;
                d_i = d_nx;
            }
            Z0(i + 1) = zc;   // i == l-1 after a full sweep, or the plane where it stopped
            log.end();
            ++nsweeps;
            if (brk) {
                // e[m] was zeroed; re-test the untouched entries below the exit plane conservatively
                for (int j = l; j <= i; ++j)
                    neg.put(j, fabsf(E(j)) <= kEps32 * (fabsf(D(j)) + fabsf(D(j + 1))));
                continue;
            }
            const float dl_new = d_ip1 - p;   // d_ip1 holds the old d[l]
            D(l) = dl_new;
            E(l) = g;
            E(m) = 0.0f;
            neg.put(l, fabsf(g) <= kEps32 * (fabsf(dl_new) + fabsf(dnew_up)));
        }
    }
    return 0;
}

// QL converges fast and accurately when the matrix is graded small -> large from top to
// bottom (LAPACK csteqr picks QL or QR per block by comparing |d(l)| and |d(lend)|).  The
// tridiagonalisation of the layer matrices starts at the arrow column, which puts the few
// large entries FIRST, so instead of a second (QR) code path the tridiagonal matrix is
// reversed (T' = J T J, W' = J W): returns true when the top half carries more weight.
template <class DA, class EA>
HD bool choose_flip(int n, DA D, EA E) {
    float top = 0.f, bot = 0.f;
    const int hlf = n / 2;
    for (int i = 0; i < hlf; ++i) {
        top += fabsf(D(i)) + fabsf(E(i));
        bot += fabsf(D(n - 1 - i)) + (n - 2 - i >= 0 ? fabsf(E(n - 2 - i)) : 0.f);
    }
    return top > bot;
}

// torch.nn.functional.softplus (beta = 1, threshold = 20)
HD float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
HD float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace admmnet
