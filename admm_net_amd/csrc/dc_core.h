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

// Leaf partition of an n x n tridiagonal: nleaf = max(1, n / 8) leaves of n / nleaf rows, the last n % nleaf of
// them one row longer (n = 129 -> 15 x 8 + 9, n = 101 -> 7 x 8 + 5 x 9: no leaf much slower than the others).
HD int dc_leaf_count(int n) { return n / 8 > 0 ? n / 8 : 1; }
HD int dc_leaf_start(int n, int nleaf, int b) {
    const int base = n / nleaf, big0 = nleaf - n % nleaf;   // leaves >= big0 have base + 1 rows
    return b * base + (b > big0 ? b - big0 : 0);
}
HD int dc_leaf_maxrows(int n, int nleaf) { return n / nleaf + (n % nleaf ? 1 : 0); }

// ---------------------------------------------------------------------------------------------
// Leaf solver: implicit QL with eigenvectors on a tiny tridiagonal (s <= 16).
// d[s], e[s] (e[i] couples i, i+1; e[s-1] ignored); Z row-major [s][ldz] receives the eigenvectors
// (columns).  A team of `kstep` lanes may share one leaf: every lane runs the (cheap) scalar
// recurrence on the shared d / e -- all lanes write identical values -- and owns the rows
// k0, k0 + kstep, ... of Z.  (k0, kstep) = (0, 1) is the single-thread form.  Returns 0 / 1.
// Plane rotation (r, sn, c) with r = hypot(f, g), sn = f / r, c = g / r.  Device: one v_rsq_f32 and a
// first-order renormalisation of (sn, c) (keeps sn^2 + c^2 = 1 to rounding, which is what the
// orthogonality of the accumulated Z needs) instead of an IEEE sqrt and two IEEE divisions on the
// serial path; values outside the safe exponent range take the exact route.
HD void plane_rot(float f, float g, float &r, float &sn, float &c) {
    const float q = f * f + g * g;
#if defined(__HIP_DEVICE_COMPILE__)
    if (q > 1e-30f && q < 1e30f) {
        const float ir = __builtin_amdgcn_rsqf(q);
        const float s0 = f * ir, c0 = g * ir;
        const float h = 0.5f * (1.0f - (s0 * s0 + c0 * c0));
        sn = fmaf(h, s0, s0);
        c = fmaf(h, c0, c0);
        r = q * ir;
        return;
    }
#endif
    r = sqrtf(q);
    if (r == 0.0f) {
        sn = 0.f;
        c = 1.f;
        return;
    }
    sn = f / r;
    c = g / r;
}

template <class FA, class ZA>
HD int leaf_ql(int s, FA d, FA e, ZA Z, int k0 = 0, int kstep = 1) {
    for (int i = k0; i < s; i += kstep)
        for (int j = 0; j < s; ++j) Z(i, j) = (i == j) ? 1.f : 0.f;
    if (s > 0) e[s - 1] = 0.f;
    for (int l = 0; l < s; ++l) {
        int iter = 0, m;
        do {
            // smallest m >= l with a negligible e[m] (no early exit: the loads pipeline)
            m = s - 1;
            for (int q = s - 2; q >= l; --q) {
                const float dd = fabsf(d[q]) + fabsf(d[q + 1]);
                if (fabsf(e[q]) <= kEps32 * dd) m = q;
            }
            if (m != l) {
                if (iter++ >= 60) return 1;
                float g = fdiv_fast(d[l + 1] - d[l], 2.0f * e[l]);
                float r = sqrtf(g * g + 1.0f);
                g = d[m] - d[l] + fdiv_fast(e[l], g + sign_of(r, g));
                float sn = 1.0f, c = 1.0f, p = 0.0f;
                // every value the sweep reads is the one from before the sweep: fetch one step ahead
                float d_up = d[m], e_i = e[m - 1], d_i = d[m - 1];
                bool brk = false;
                for (int i = m - 1; i >= l; --i) {
                    float e_n = 0.f, d_n = 0.f;
                    if (i > l) {
                        e_n = e[i - 1];
                        d_n = d[i - 1];
                    }
                    const float f = sn * e_i;
                    const float b = c * e_i;
                    plane_rot(f, g, r, sn, c);
                    e[i + 1] = r;
                    if (r == 0.0f) {
                        d[i + 1] = d_up - p;
                        e[m] = 0.0f;
                        brk = true;
                        break;
                    }
                    g = d_up - p;
                    r = (d_i - g) * sn + 2.0f * c * b;
                    p = sn * r;
                    d[i + 1] = g + p;
                    g = c * r - b;
                    for (int k = k0; k < s; k += kstep) {
                        const float zf = Z(k, i + 1);
                        Z(k, i + 1) = sn * Z(k, i) + c * zf;
                        Z(k, i) = c * Z(k, i) - sn * zf;
                    }
                    d_up = d_i;
                    e_i = e_n;
                    d_i = d_n;
                }
                if (brk) continue;
                d[l] = d_up - p;
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

// 1 / sqrt(x): v_rsq_f32 plus one Newton step on the device
HD float rsqrt_nr(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rsqf(x);
    return r * fmaf(-0.5f * x * r, r, 1.5f);
#else
    return 1.0f / sqrtf(x);
#endif
}

// The scan is a serial chain over (pj, d_pj, z_pj): that state is carried in registers and the
// next (d, z) pair is fetched one step ahead, so the chain never waits on the LDS.  `dzmax` =
// max(max |ds|, max |zs|) is supplied by the caller (team reduction on the device).
template <class FA, class IA, class RA>
HD void deflate_scan_tol(int nn, float rho, float dmax, float zmax, FA ds, FA zs, FA dl, FA zl, IA src, RA rot,
                         int &k_out, int &nrot_out) {
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
    float dpj = 0.f, zpj = 0.f;
    // one scan step on (j, d_j, z_j)
    auto step = [&](int j, float dj, float zj) {
        if (rho * fabsf(zj) <= tol) {   // type 1: tiny z component
            --k2;
            dl[k2] = dj;
            src[k2] = j;
            return;
        }
        if (pj < 0) {
            pj = j;
            dpj = dj;
            zpj = zj;
            return;
        }
        // type 2: two (nearly) equal poles -> rotate z_pj into z_j.  With c = z_j / tau, s = -z_pj / tau
        // the test |t c s| <= tol reads |t z_j z_pj| <= tol tau^2: no square root unless it deflates.
        const float q = zj * zj + zpj * zpj;
        const float t = dj - dpj;
        if (!(fabsf(t * zj * zpj) <= tol * q) || q < 1e-30f) {   // (q guard: keep v_rsq_f32 in range)
            dl[k] = dpj;
            zl[k] = zpj;
            src[k] = pj;
            ++k;
            pj = j;
            dpj = dj;
            zpj = zj;
            return;
        }
        const float itau = rsqrt_nr(q);
        const float tau = q * itau;
        const float c = zj * itau, s = -zpj * itau;
        DcRot r;
        r.pa = pj;
        r.pb = j;
        r.c = c;
        r.s = s;
        rot[nrot++] = r;
        --k2;
        dl[k2] = dpj * c * c + dj * s * s;
        src[k2] = pj;
        dpj = dpj * s * s + dj * c * c;
        zpj = tau;
        pj = j;
    };
    // The inputs are read four entries at a time, one group ahead of their use (two register sets in
    // ping-pong): with a one-step look-ahead the copy "next -> current" at the end of every trip still waited
    // for the load it had just issued.
    float da[4], za[4], db[4], zb[4];
    auto fetch = [&](int j0, float (&d4)[4], float (&z4)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = (j0 + q < nn) ? j0 + q : nn - 1;
            d4[q] = ds[j];
            z4[q] = zs[j];
        }
    };
    fetch(0, da, za);
    for (int j0 = 0; j0 < nn; j0 += 8) {
        fetch(j0 + 4, db, zb);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (j0 + q < nn) step(j0 + q, da[q], za[q]);
        fetch(j0 + 8, da, za);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (j0 + 4 + q < nn) step(j0 + 4 + q, db[q], zb[q]);
    }
    if (pj >= 0) {
        dl[k] = dpj;
        zl[k] = zpj;
        src[k] = pj;
        ++k;
    }
    k_out = k;
    nrot_out = nrot;
}

// ---------------------------------------------------------------------------------------------
// Team form of the scan: same outputs as deflate_scan_tol, bit for bit, in three phases separated by team
// barriers.  The serial chain only carries state across a rotation (the survivor's d and z change), so:
//   A  every position evaluates its own tests against its predecessor's ORIGINAL values -- exact wherever
//      the predecessor does not come out of a rotation;
//   B  one walker per run of consecutive rotation candidates replays the chain through that run (and on,
//      while the modified values keep rotating); a chain that runs into the head of another run raises
//      `conflict` and the caller falls back to the serial scan (rare);
//   C  output positions by counting: k2 slots from the number of deflation events before a step, rotation
//      slots from the rotations before it, non-deflated slots from (non-tiny positions) - (rotations).
// Work arrays of length nn: pv (previous non-tiny position), flg (DF_* bits), dvf / zvf (values a position
// carries forward), rc / rs (rotation at step j), dde (value of the pole deflated by that rotation).
constexpr int DF_TINY = 1, DF_R0 = 2, DF_ROT = 4;

template <class FA>
HD bool defl_close(float tol, float dpj, float zpj, float dj, float zj, FA) {
    const float q = zj * zj + zpj * zpj;
    const float t = dj - dpj;
    return (fabsf(t * zj * zpj) <= tol * q) && !(q < 1e-30f);
}

template <class FA, class IA>
HD void defl_par_flags(int tl, int ts, int nn, float rho, float tol, FA ds, FA zs, IA pv, IA flg, FA dvf, FA zvf) {
    for (int j = tl; j < nn; j += ts) {
        const float dj = ds[j], zj = zs[j];
        int f = (rho * fabsf(zj) <= tol) ? DF_TINY : 0;
        int p = j - 1;
        while (p >= 0 && rho * fabsf(zs[p]) <= tol) --p;   // previous non-tiny position (tiny entries are rare)
        if (!f && p >= 0 && defl_close(tol, ds[p], zs[p], dj, zj, ds)) f |= DF_R0;
        pv[j] = p;
        flg[j] = f;
        dvf[j] = dj;
        zvf[j] = zj;
    }
}

template <class FA, class IA>
HD void defl_par_walk(int tl, int ts, int nn, float tol, FA ds, FA zs, IA pv, IA flg, FA dvf, FA zvf, FA rc, FA rs,
                      FA dde, int *conflict) {
    for (int j0 = tl; j0 < nn; j0 += ts) {
        if (!(flg[j0] & DF_R0) || (flg[pv[j0]] & DF_R0)) continue;   // heads of runs only
        int pj = pv[j0], cur = j0;
        float dpj = ds[pj], zpj = zs[pj];
        for (;;) {
            const float dj = ds[cur], zj = zs[cur];
            const int fc = flg[cur];
            const bool rotd = defl_close(tol, dpj, zpj, dj, zj, ds);
            if (rotd) {
                const float q = zj * zj + zpj * zpj;
                const float itau = rsqrt_nr(q);
                const float tau = q * itau;
                const float c = zj * itau, s = -zpj * itau;
                rc[cur] = c;
                rs[cur] = s;
                dde[cur] = dpj * c * c + dj * s * s;
                dpj = dpj * s * s + dj * c * c;
                zpj = tau;
                dvf[cur] = dpj;
                zvf[cur] = zpj;
                flg[cur] = fc | DF_ROT;
            } else {
                dpj = dj;
                zpj = zj;
            }
            int nx = cur + 1;
            while (nx < nn && (flg[nx] & DF_TINY)) ++nx;
            if (nx >= nn) break;
            const bool r0c = fc & DF_R0, r0n = flg[nx] & DF_R0;
            if (!rotd && !(r0c && r0n)) break;   // clean from here (a following run has its own walker)
            if (rotd && !r0c && r0n) {           // the chain grew into the head of another run
                *conflict = 1;
                break;
            }
            pj = cur;
            cur = nx;
        }
    }
}

template <class FA, class IA, class RA>
HD void defl_par_emit(int tl, int ts, int nn, FA ds, IA pv, IA flg, FA dvf, FA zvf, FA rc, FA rs, FA dde, FA dl,
                      FA zl, IA src, RA rot, int *k_out, int *nrot_out) {
    for (int j = tl; j < nn; j += ts) {
        int ntiny = 0, nrot = 0;   // tiny entries / rotations at steps before j
#pragma unroll 4
        for (int i = 0; i < j; ++i) {
            const int f = flg[i];
            ntiny += f & DF_TINY;
            nrot += (f & DF_ROT) ? 1 : 0;
        }
        const int f = flg[j];
        const int k2 = nn - 1 - (ntiny + nrot);   // the serial scan fills the deflated slots from the top, one per event
        if (f & DF_TINY) {
            dl[k2] = ds[j];
            src[k2] = j;
            continue;
        }
        if (f & DF_ROT) {
            dl[k2] = dde[j];
            src[k2] = pv[j];
            DcRot r;
            r.pa = pv[j];
            r.pb = j;
            r.c = rc[j];
            r.s = rs[j];
            rot[nrot] = r;
        }
        int nx = j + 1;
        while (nx < nn && (flg[nx] & DF_TINY)) ++nx;
        if (nx >= nn || !(flg[nx] & DF_ROT)) {   // position j survives: slot = non-tiny positions before - rotations up to j
            const int idx = (j - ntiny) - (nrot + ((f & DF_ROT) ? 1 : 0));
            dl[idx] = dvf[j];
            zl[idx] = zvf[j];
            src[idx] = j;
        }
    }
    if (tl == 0) {
        int ntiny = 0, nrot = 0;
#pragma unroll 4
        for (int i = 0; i < nn; ++i) {
            const int f = flg[i];
            ntiny += f & DF_TINY;
            nrot += (f & DF_ROT) ? 1 : 0;
        }
        *k_out = (nn - ntiny) - nrot;
        *nrot_out = nrot;
    }
}

// Fast path of the scan: does ANY deflation trigger at sorted position p?  When no position says yes
// the serial scan would keep every entry (its running pj is always p - 1), so the caller may skip it and
// copy (ds, zs) -> (dl, zl), src = identity, k = nn, nrot = 0 in parallel.  Same tests, same tolerance.
template <class FA>
HD bool deflate_triggers(int p, float rho, float dmax, float zmax, FA ds, FA zs) {
    const float tol = 8.0f * kEps32 * fmaxf(dmax, zmax);
    const float zj = zs[p];
    if (rho * fabsf(zj) <= tol) return true;
    if (p == 0) return false;
    const float zpj = zs[p - 1];
    const float q = zj * zj + zpj * zpj;
    const float t = ds[p] - ds[p - 1];
    return (fabsf(t * zj * zpj) <= tol * q) && !(q < 1e-30f);
}

template <class FA, class IA, class RA>
HD void deflate_scan(int nn, float rho, FA ds, FA zs, FA dl, FA zl, IA src, RA rot, int &k_out, int &nrot_out) {
    float dmax = 0.f, zmax = 0.f;
    for (int i = 0; i < nn; ++i) {
        dmax = fmaxf(dmax, fabsf(ds[i]));
        zmax = fmaxf(zmax, fabsf(zs[i]));
    }
    deflate_scan_tol(nn, rho, dmax, zmax, ds, zs, dl, zl, src, rot, k_out, nrot_out);
}

// ---------------------------------------------------------------------------------------------
// Secular equation  w(lam) = 1/rho + sum_i z_i^2 / (d_i - lam) = 0,  root j of k (0-based),
// d ascending and distinct, rho > 0, ||z|| = 1.  Root j lies in (d_j, d_{j+1}) (last: (d_{k-1},
// d_{k-1} + rho)).  Returns the origin pole `org` and tau with lam_j = d_org + tau; differences
// d_i - lam_j are then formed as (d_i - d_org) - tau, which keeps them accurate relative to
// themselves (the property LAPACK slaed4 is built around).
//
// Iteration: R.-C. Li's "middle way" (the scheme inside slaed4): split w = 1/rho + psi + phi at the
// root's interval, model psi by s + p / (d_lo - lam) and phi by S + P / (d_hi - lam) matching value
// and slope at the current point, and solve the resulting quadratic for the step.  Quadratic
// convergence, no stagnating bracket end; a bracket [lo, hi] (w(lo) < 0 < w(hi)) is kept anyway and a
// step that leaves it is replaced by bisection.  Stop when |w| is below the rounding noise of its
// own evaluation (slaed4's criterion).  One evaluation = one pass over the k poles.
struct SecEval {
    float w, dpsi, dphi, err;
};

// A root may be shared by a group of G adjacent lanes: lane `sub` sums the poles sub, sub + G, ... and `red`
// adds the partial sums across the group (all lanes of the group then hold the same totals and take the same
// decisions).  Host / single lane: sub = 0, G = 1, red = identity.
struct SecNoReduce {
    HD float operator()(float x) const { return x; }
};

template <class FA, class Red = SecNoReduce>
HD SecEval secular_eval(int k, int jsplit, float rhoinv, float dorg, float t, FA d, FA z, int sub = 0, int G = 1,
                        Red red = Red()) {
    // psi: poles 0..jsplit, phi: the rest
    float sum = 0.f, asum = 0.f, dall = 0.f, dps = 0.f;
    // (unrolled: one LDS round trip per iteration would otherwise bound the loop, not the arithmetic)
#pragma unroll 4
    for (int i = sub; i < k; i += G) {
        const float del = (d[i] - dorg) - t;
        const float r = fdiv_fast(1.0f, del);
        const float term = z[i] * z[i] * r;
        const float dterm = term * r;
        sum += term;
        asum += fabsf(term);
        dall += dterm;
        dps += (i <= jsplit) ? dterm : 0.f;
    }
    sum = red(sum);
    asum = red(asum);
    dall = red(dall);
    dps = red(dps);
    SecEval e;
    e.w = rhoinv + sum;
    e.dpsi = dps;
    e.dphi = dall - dps;
    e.err = 8.0f * asum + rhoinv + fabsf(t) * dall;
    return e;
}

template <class FA, class Red = SecNoReduce>
HD void secular_root(int k, int j, float rho, FA d, FA z, int &org_out, float &tau_out, int *nit = nullptr,
                     int sub = 0, int G = 1, Red red = Red()) {
    if (nit) *nit = 0;
    if (k == 1) {
        org_out = 0;
        tau_out = rho * z[0] * z[0];
        return;
    }
    const float rhoinv = 1.0f / rho;
    const bool last = (j == k - 1);
    // the two poles the rational model keeps exact, and the psi / phi split
    const int plo = last ? k - 2 : j, phi_ = last ? k - 1 : j + 1;
    const int jsplit = plo;
    int org;
    float lo, hi, t;
    SecEval e;
    if (last) {
        org = k - 1;
        lo = 0.f;
        hi = rho;
        t = 0.5f * rho;
        e = secular_eval(k, jsplit, rhoinv, d[org], t, d, z, sub, G, red);
    } else {
        const float gap = d[j + 1] - d[j];
        const float half = 0.5f * gap;
        e = secular_eval(k, jsplit, rhoinv, d[j], half, d, z, sub, G, red);
        if (e.w >= 0.f) {   // root in the left half: measure from d_j
            org = j;
            lo = 0.f;
            hi = half;
            t = half;
        } else {            // right half: measure from d_{j+1}
            org = j + 1;
            lo = -half;
            hi = 0.f;
            t = -half;
        }
    }
    const float dorg = d[org];
    const float plo_off = d[plo] - dorg, phi_off = d[phi_] - dorg;
    for (int it = 0; it < 40; ++it) {
        if (nit) *nit = it + 1;
        if (fabsf(e.w) <= kEps32 * e.err) break;
        if (e.w < 0.f) lo = t; else hi = t;
        const float d1 = plo_off - t, d2 = phi_off - t;   // distances to the two modelled poles
        const float dw = e.dpsi + e.dphi;
        float c = e.w - d1 * e.dpsi - d2 * e.dphi;
        const float aa = (d1 + d2) * e.w - d1 * d2 * dw;
        const float bb = d1 * d2 * e.w;
        float eta;
        if (last) {
            if (c < 0.f) c = -c;
            if (c == 0.f) {
                eta = hi - t;
            } else {
                const float disc = sqrtf(fabsf(aa * aa - 4.0f * bb * c));
                eta = (aa >= 0.f) ? fdiv_fast(aa + disc, 2.0f * c) : fdiv_fast(2.0f * bb, aa - disc);
            }
        } else {
            if (c == 0.f) {
                eta = (aa != 0.f) ? fdiv_fast(bb, aa) : 0.f;
            } else {
                const float disc = sqrtf(fabsf(aa * aa - 4.0f * bb * c));
                eta = (aa <= 0.f) ? fdiv_fast(aa - disc, 2.0f * c) : fdiv_fast(2.0f * bb, aa + disc);
            }
        }
        if (!(e.w * eta < 0.f)) eta = -fdiv_fast(e.w, dw);   // wrong direction (or NaN): Newton step
        float tn = t + eta;
        if (!(tn > lo && tn < hi)) tn = 0.5f * (lo + hi);
        if (tn == t || tn == lo || tn == hi) break;           // bracket exhausted at this precision
        t = tn;
        e = secular_eval(k, jsplit, rhoinv, dorg, t, d, z, sub, G, red);
    }
    // never return a pole itself (the Loewner products divide by these differences)
    if (t == 0.f) t = (lo == 0.f) ? 0.5f * hi : 0.5f * lo;
    if (t == 0.f) t = (org == j || last) ? 1e-30f : -1e-30f;
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
// A group of G adjacent lanes may share one i: lane `sub` takes the factors j = sub, sub + G, ... and `red`
// multiplies the partial products across the group (host / single lane: sub = 0, G = 1, identity).
template <class FA, class IA, class Red = SecNoReduce>
HD float lowner_zhat(int k, int i, FA d, FA z, IA org, FA tau, int sub = 0, int G = 1, Red red = Red()) {
    float w = (sub == 0) ? dc_delta(d, org, tau, i, i) : 1.0f;   // d_i - lam_i
#pragma unroll 4
    for (int j = sub; j < k; j += G) {
        const float q = fdiv_fast(dc_delta(d, org, tau, i, j), d[i] - d[j]);
        w *= (j == i) ? 1.0f : q;
    }
    w = red(w);
    const float r = sqrtf(fabsf(w));
    return z[i] >= 0.f ? r : -r;
}

}  // namespace admmnet
