// dc_core.h -- scalar building blocks of the divide & conquer eigensolver for real symmetric
// tridiagonal matrices (Cuppen's method with the Gu-Eisenstat stabilisation, i.e. what LAPACK
// sstedc / slaed0-4 do).  Each function is the per-thread (or single-thread) piece of one phase;
// dc.hip strings them together with team-parallel loops, tests/host_model/dc_model.cpp runs the
// very same functions sequentially on the CPU.
//
// Why D&C on this problem: the layer matrices are (c I + tiny diagonal + low rank), i.e. all but
// a handful of eigenvalues sit in one cluster.  QL needs ~n^2 plane rotations in a serial chain
// whatever the spectrum; D&C DEFLATES clusters (work disappears), every remaining phase is
// parallel over eigenvalues, and the eigenvectors come out of GEMMs (matrix cores).
#pragma once
#include <math.h>
#include <stdint.h>

#include "eig_core.h"

namespace admmnet {

// a / b with one v_rcp_f32 (1 ulp) on the device; exact division on the host.  Used where the
// quotient only feeds an iteratively refined quantity (secular function) or a vector that is
// normalised afterwards.
HD float fdiv_fast(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}

// ---------------------------------------------------------------------------------------------
// Leaf solver: implicit QL with eigenvectors on a tiny tridiagonal (s <= 16).
// d[s], e[s] (e[i] couples i, i+1; e[s-1] ignored); Z row-major [s][ldz] receives the eigenvectors
// (columns).  A team of `kstep` lanes may share one leaf: every lane runs the (cheap) scalar
// recurrence on the shared d / e -- all lanes write identical values -- and owns the rows
// k0, k0 + kstep, ... of Z.  (k0, kstep) = (0, 1) is the single-thread form.  Returns 0 / 1.
template <class FA, class ZA>
HD int leaf_ql(int s, FA d, FA e, ZA Z, int k0 = 0, int kstep = 1) {
    for (int i = k0; i < s; i += kstep)
        for (int j = 0; j < s; ++j) Z(i, j) = (i == j) ? 1.f : 0.f;
    if (s > 0) e[s - 1] = 0.f;
    for (int l = 0; l < s; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < s - 1; ++m) {
                const float dd = fabsf(d[m]) + fabsf(d[m + 1]);
                if (fabsf(e[m]) <= kEps32 * dd) break;
            }
            if (m != l) {
                if (iter++ >= 60) return 1;
                float g = (d[l + 1] - d[l]) / (2.0f * e[l]);
                float r = sqrtf(g * g + 1.0f);
                g = d[m] - d[l] + e[l] / (g + sign_of(r, g));
                float sn = 1.0f, c = 1.0f, p = 0.0f;
                int i;
                bool brk = false;
                for (i = m - 1; i >= l; --i) {
                    float f = sn * e[i];
                    const float b = c * e[i];
                    r = sqrtf(f * f + g * g);
                    e[i + 1] = r;
                    if (r == 0.0f) {
                        d[i + 1] -= p;
                        e[m] = 0.0f;
                        brk = true;
                        break;
                    }
                    sn = f / r;
                    c = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * sn + 2.0f * c * b;
                    p = sn * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                    for (int k = k0; k < s; k += kstep) {
                        f = Z(k, i + 1);
                        Z(k, i + 1) = sn * Z(k, i) + c * f;
                        Z(k, i) = c * Z(k, i) - sn * f;
                    }
                }
                if (brk) continue;
                d[l] -= p;
                e[l] = g;
                e[m] = 0.0f;
            }
        } while (m != l);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Deflation scan of one merge (LAPACK slaed2 logic), single thread.
//   in : nn, ds[nn] ascending, zs[nn] (||z|| = 1), rho > 0
//   out: k non-deflated entries -> dl[0..k), zl[0..k), src[0..k) (position in the sorted input),
//        deflated entries at positions nn-1 down to k: dl[p] = eigenvalue, src[p] = sorted position,
//        rotations rot[0..nrot): columns (a -> b) with (c, s) to apply IN ORDER to the eigenvector
//        columns:  col_a' = c col_a - s col_b... (see apply convention below)
// Convention of a recorded rotation {pa, pb, c, s} on the columns (x = col pa, y = col pb):
//        x' = c x + s y ;  y' = -s x + c y       (LAPACK srot(x, y, c, s))
struct DcRot {
    int pa, pb;
    float c, s;
};

template <class FA, class IA, class RA>
HD void deflate_scan(int nn, float rho, FA ds, FA zs, FA dl, FA zl, IA src, RA rot, int &k_out, int &nrot_out) {
    float dmax = 0.f, zmax = 0.f;
    for (int i = 0; i < nn; ++i) {
        dmax = fmaxf(dmax, fabsf(ds[i]));
        zmax = fmaxf(zmax, fabsf(zs[i]));
    }
    const float tol = 8.0f * kEps32 * fmaxf(dmax, zmax);
    int k = 0, k2 = nn, nrot = 0;
    if (rho * zmax <= tol) {   // the rank-one term is negligible: everything deflates
        for (int j = 0; j < nn; ++j) {
            --k2;
            dl[k2] = ds[j];
            src[k2] = j;
        }
        k_out = 0;
        nrot_out = 0;
        return;
    }
    int pj = -1;
    for (int j = 0; j < nn; ++j) {
        if (rho * fabsf(zs[j]) <= tol) {   // type 1: tiny z component
            --k2;
            dl[k2] = ds[j];
            src[k2] = j;
            continue;
        }
        if (pj < 0) {
            pj = j;
            continue;
        }
        // type 2: two (nearly) equal poles -> rotate z_pj into z_j
        float s = zs[pj], c = zs[j];
        const float tau = sqrtf(c * c + s * s);
        const float t = ds[j] - ds[pj];
        c /= tau;
        s = -s / tau;
        if (fabsf(t * c * s) <= tol) {
            zs[j] = tau;
            zs[pj] = 0.f;
            DcRot r;
            r.pa = pj;
            r.pb = j;
            r.c = c;
            r.s = s;
            rot[nrot++] = r;
            const float tt = ds[pj] * c * c + ds[j] * s * s;
            ds[j] = ds[pj] * s * s + ds[j] * c * c;
            ds[pj] = tt;
            --k2;
            dl[k2] = ds[pj];
            src[k2] = pj;
            pj = j;
        } else {
            dl[k] = ds[pj];
            zl[k] = zs[pj];
            src[k] = pj;
            ++k;
            pj = j;
        }
    }
    if (pj >= 0) {
        dl[k] = ds[pj];
        zl[k] = zs[pj];
        src[k] = pj;
        ++k;
    }
    k_out = k;
    nrot_out = nrot;
}

// ---------------------------------------------------------------------------------------------
// Secular equation  f(lam) = 1 + rho * sum_i z_i^2 / (d_i - lam) = 0,  root j of k (0-based),
// d ascending and distinct, rho > 0, ||z|| = 1.  Root j lies in (d_j, d_{j+1}) (last: (d_{k-1},
// d_{k-1} + rho)).  Returns the origin pole `org` and tau with lam_j = d_org + tau; differences
// d_i - lam_j are then formed as (d_i - d_org) - tau, which keeps them accurate relative to
// themselves (the property LAPACK slaed4 is built around).  Bracketing Illinois iteration on
// the pole-free transform h(t) = f * (t - pl)(pr - t).
template <class FA>
HD void secular_root(int k, int j, float rho, FA d, FA z, int &org_out, float &tau_out) {
    if (k == 1) {
        org_out = 0;
        tau_out = rho * z[0] * z[0];
        return;
    }
    const bool last = (j == k - 1);
    // evaluate f at lam = d_org + t, also returns it for the transform
    auto feval = [&](int org, float t) -> float {
        float acc = 0.f;
        const float dorg = d[org];
        for (int i = 0; i < k; ++i) {
            const float del = (d[i] - dorg) - t;
            acc = fmaf(z[i] * z[i], fdiv_fast(1.0f, del), acc);
        }
        return 1.0f + rho * acc;
    };
    int org;
    float lo, hi;   // bracket in tau, f(lo) < 0 < f(hi) once inside
    if (last) {
        org = k - 1;
        lo = 0.f;
        hi = rho;   // f(d_{k-1} + rho ||z||^2) >= 0
        if (feval(org, hi) <= 0.f) {   // numerical corner: root at / beyond the upper bound
            org_out = org;
            tau_out = hi;
            return;
        }
    } else {
        const float gap = d[j + 1] - d[j];
        const float half = 0.5f * gap;
        const float fm = feval(j, half);
        if (fm >= 0.f) {
            org = j;
            lo = 0.f;
            hi = half;
        } else {
            org = j + 1;
            lo = -half;
            hi = 0.f;
        }
    }
    // poles adjacent to the bracket, relative to the origin
    const float pl = (org == j || last) ? 0.f : (d[j] - d[org]);           // left pole offset
    const float pr = last ? 2.0f * rho + 0.f : ((org == j) ? (d[j + 1] - d[org]) : 0.f);   // right pole offset
    auto heval = [&](float t) -> float {
        const float f = feval(org, t);
        return last ? f * (t - pl) : f * (t - pl) * (pr - t);
    };
    // bracket ends: at t = pl (resp. pr) h has a finite limit of known sign (-, +): use the limits
    // h(pl+) = -rho z_j^2 (pr - pl) < 0,  h(pr-) = +rho z_{j+1}^2 (pr - pl) > 0; interior ends evaluated.
    float a = lo, b = hi, ha, hb;
    if (last) {
        ha = -rho * z[k - 1] * z[k - 1];
        hb = heval(b);
    } else if (org == j) {
        ha = -rho * z[j] * z[j] * (pr - pl);
        hb = heval(b);
    } else {
        ha = heval(a);
        hb = rho * z[j + 1] * z[j + 1] * (pr - pl);
    }
    float t = 0.5f * (a + b);
    if (!(ha < 0.f && hb > 0.f)) {   // degenerate bracket (ha == 0 or hb == 0 numerically)
        org_out = org;
        tau_out = (ha >= 0.f) ? a : b;
        if (tau_out == 0.f) tau_out = (org == j || last) ? 1e-30f : -1e-30f;
        return;
    }
    int side = 0;
    for (int it = 0; it < 80; ++it) {
        // Illinois step, kept strictly inside the bracket; fall back to bisection when it stalls
        float tn = b - hb * (b - a) / (hb - ha);
        if (!(tn > a && tn < b)) tn = 0.5f * (a + b);
        if (tn == a || tn == b) {
            t = tn;
            break;
        }
        const float hn = heval(tn);
        t = tn;
        if (hn == 0.f) break;
        if (hn < 0.f) {
            a = tn;
            ha = hn;
            if (side == -1) hb *= 0.5f;
            side = -1;
        } else {
            b = tn;
            hb = hn;
            if (side == 1) ha *= 0.5f;
            side = 1;
        }
        // converged when the bracket is within a couple of ulps of |t|
        if ((b - a) <= 4.0f * kEps32 * fmaxf(fabsf(a), fabsf(b))) {
            t = 0.5f * (a + b);
            break;
        }
    }
    if (t == 0.f) t = (org == j || last) ? a + 0.5f * (b - a) : b - 0.5f * (b - a);
    org_out = org;
    tau_out = t;
}

// d_i - lam_j formed from the stored (org_j, tau_j)
template <class FA, class IA>
HD float dc_delta(FA d, IA org, FA tau, int i, int j) {
    return (d[i] - d[org[j]]) - tau[j];
}

// Gu / Eisenstat: z-hat_i = sign(z_i) sqrt( prod_j (lam_j - d_i) / prod_{j != i} (d_j - d_i) / rho ... )
// with all differences taken from the COMPUTED roots, which makes the eigenvectors
// u_j = (zhat_i / (d_i - lam_j))_i numerically orthogonal (LAPACK slaed3).
template <class FA, class IA>
HD float lowner_zhat(int k, int i, FA d, FA z, IA org, FA tau) {
    float w = dc_delta(d, org, tau, i, i);   // d_i - lam_i
    for (int j = 0; j < k; ++j) {
        if (j == i) continue;
        w *= dc_delta(d, org, tau, i, j) / (d[i] - d[j]);
    }
    const float r = sqrtf(fabsf(w));
    return z[i] >= 0.f ? r : -r;
}

}  // namespace admmnet
