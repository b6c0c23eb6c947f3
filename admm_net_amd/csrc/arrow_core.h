// arrow_core.h -- scalar building blocks of the direct eigensolver for Hermitian ARROWHEAD matrices
//
//        A = [[ alpha, z^H ], [ z, diag(h) ]]          (arrow-first order, alpha and h real)
//
// which is what the first G-layer of the unrolled network sees: A = C - Z / rho with Z = 0, i.e. the
// plain block matrix C = [[diag h, phi], [phi^H, corner]] (/root/reference/admm_net.py:273-288; eigh at
// :303).  After the diagonal unitary scaling S = diag(1, z_i / |z_i|) the matrix is real with
// zeta_i = |z_i| >= 0, and its eigenpairs follow from one secular equation -- O(n^2) work instead of
// the O(n^3) tridiagonalisation + D&C + back-transform of the dense path:
//   * poles d (= h sorted ascending) with tiny zeta or (nearly) equal neighbours deflate exactly as in
//     the D&C merge (deflate_scan_tol in dc_core.h, with rho = 1);
//   * the k + 1 roots of  F(lam) = (lam - alpha) + sum_i zeta_i^2 / (d_i - lam)  interlace the k
//     surviving poles: lam_0 < d_0 < lam_1 < ... < d_{k-1} < lam_k;
//   * Gu & Eisenstat: recompute zeta-hat from the COMPUTED roots (the arrowhead with exactly those
//     eigenvalues), zeta-hat_i^2 = (d_i - lam_i)(lam_{i+1} - d_i) prod_{j<i} (d_i - lam_j)/(d_i - d_j)
//     prod_{j>i} (lam_{j+1} - d_i)/(d_j - d_i); eigenvector j = (1, zeta-hat_i / (lam_j - d_i))_i
//     normalised -- orthogonal to working precision whatever the accuracy of the roots.
// Shared between arrow.hip (device) and tests/host_model/arrow_model.cpp (sequential CPU model).
#pragma once
#include "dc_core.h"

namespace admmnet {

struct ArrowEval {
    float w, dpsi, dphi, err;
};

// F at lam = d_org + t, split at pole index jsplit (psi: poles 0..jsplit, phi: the rest); `c0` = d_org - alpha.
// DA / ZA are accessors (index -> value) so that the bottom root can run on the reflected problem.
// A root may be shared by a group of G adjacent lanes: lane `sub` sums the poles sub, sub + G, ... and `red`
// adds the partial sums across the group (every lane of the group gets the same totals and then takes the
// same decisions).  Host / single lane: sub = 0, G = 1, red = identity.
struct ArrowNoReduce {
    HD float operator()(float x) const { return x; }
};

template <class DA, class ZA, class Red = ArrowNoReduce>
HD ArrowEval arrow_eval(int k, int jsplit, float c0, float dorg, float t, DA d, ZA z, int sub = 0, int G = 1,
                        Red red = Red()) {
    float sum = 0.f, asum = 0.f, dall = 0.f, dps = 0.f;
    // (unrolled: one LDS round trip per iteration would otherwise bound the loop, not the arithmetic)
#pragma unroll 4
    for (int i = sub; i < k; i += G) {
        const float del = (d(i) - dorg) - t;
        const float r = fdiv_fast(1.0f, del);
        const float zi = z(i);
        const float term = zi * zi * r;
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
    ArrowEval e;
    e.w = (c0 + t) + sum;
    e.dpsi = dps;
    e.dphi = dall - dps;
    e.err = 8.0f * asum + fabsf(c0) + 2.0f * fabsf(t) + fabsf(t) * dall;
    return e;
}

// Root j (0 < j <= k) of the k-pole problem given through accessors; the caller maps j = 0 onto j = k of
// the reflected problem.  j < k: interior root in (d_{j-1}, d_j); j = k: top root in (d_{k-1}, ub).
// Returns the origin pole and tau (lam = d_org + tau), as secular_root does.
template <class DA, class ZA, class Red = ArrowNoReduce>
HD void arrow_root_upper(int k, int j, float alpha, float znorm, DA d, ZA z, int &org_out, float &tau_out,
                         int *nit = nullptr, int sub = 0, int G = 1, Red red = Red()) {
    if (nit) *nit = 0;
    const bool top = (j == k);
    int org, plo, phi_;
    float lo, hi, t;
    ArrowEval e;
    if (top) {
        org = k - 1;
        plo = (k >= 2) ? k - 2 : k - 1;
        phi_ = k - 1;
        // upper bound: the top root of the 2 x 2 problem with every pole moved up to d_{k-1}
        const float bq = d(k - 1) - alpha;
        const float disc = sqrtf(bq * bq + 4.0f * znorm * znorm);
        hi = (bq > 0.f) ? 2.0f * znorm * znorm / (bq + disc) : 0.5f * (disc - bq);
        lo = 0.f;
        if (k == 1) {   // a single pole: the bound IS the root
            org_out = org;
            tau_out = hi;
            return;
        }
        hi = hi * (1.0f + 8.0f * kEps32) + 1e-30f;
        t = hi;
        e = arrow_eval(k, plo, d(org) - alpha, d(org), t, d, z, sub, G, red);
        if (e.w <= 0.f) {   // numerical corner: the root sits at the bound
            org_out = org;
            tau_out = hi;
            return;
        }
    } else {
        plo = j - 1;
        phi_ = j;
        const float half = 0.5f * (d(j) - d(j - 1));
        e = arrow_eval(k, plo, d(j - 1) - alpha, d(j - 1), half, d, z, sub, G, red);
        if (e.w >= 0.f) {   // root in the lower half: measure from d_{j-1}
            org = j - 1;
            lo = 0.f;
            hi = half;
            t = half;
        } else {
            org = j;
            lo = -half;
            hi = 0.f;
            t = -half;   // same point seen from the new origin: e stays valid
        }
    }
    const float dorg = d(org), c0 = dorg - alpha;
    const float plo_off = d(plo) - dorg, phi_off = d(phi_) - dorg;
    for (int it = 0; it < 60; ++it) {
        if (nit) *nit = it + 1;
        if (fabsf(e.w) <= kEps32 * e.err) break;
        if (e.w < 0.f) lo = t; else hi = t;
        const float d1 = plo_off - t, d2 = phi_off - t;
        const float dw = e.dpsi + e.dphi + 1.0f;
        float c = e.w - d1 * e.dpsi - d2 * e.dphi;
        const float aa = (d1 + d2) * e.w - d1 * d2 * (e.dpsi + e.dphi);
        const float bb = d1 * d2 * e.w;
        float eta;
        if (top) {
            // outer root: keep the nearest pole AND the linear term exact, linearise the rest (psi):
            //   (c0 + t) + psi(t0) + psi'(t0)(t - t0) - zeta^2 / t = 0   ->   B t^2 + A t - zeta^2 = 0
            const float zk = z(k - 1), z2 = zk * zk;
            const float psi = e.w - (c0 + t) + fdiv_fast(z2, t);
            const float B = 1.0f + e.dpsi, A = c0 + psi - e.dpsi * t;
            const float disc = sqrtf(A * A + 4.0f * B * z2);
            const float tq = (A > 0.f) ? fdiv_fast(2.0f * z2, A + disc) : fdiv_fast(disc - A, 2.0f * B);
            eta = tq - t;
            (void)c; (void)aa; (void)bb;
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
        if (tn == t || tn == lo || tn == hi) break;
        t = tn;
        e = arrow_eval(k, plo, c0, dorg, t, d, z, sub, G, red);
    }
    if (t == 0.f) t = (lo == 0.f) ? 0.5f * hi : 0.5f * lo;
    if (t == 0.f) t = (lo == 0.f) ? 1e-30f : -1e-30f;
    org_out = org;
    tau_out = t;
}

// Root j of k + 1 (0 <= j <= k) of the arrowhead (alpha; d[k] ascending distinct; z[k] > 0).
// The bottom root is minus the top root of the reflected problem (-alpha, -d reversed); both cases run
// through the SAME code with a per-lane (sign, index map), so a wave whose lanes hold different roots
// does not execute the solver twice.
template <class FA, class Red = ArrowNoReduce>
HD void arrow_root(int k, int j, float alpha, float znorm, FA d, FA z, int &org_out, float &tau_out,
                   int *nit = nullptr, int sub = 0, int G = 1, Red red = Red()) {
    const bool refl = (j == 0);
    const float sg = refl ? -1.0f : 1.0f;
    const int km1 = k - 1;
    int orgr;
    float taur;
    arrow_root_upper(k, refl ? k : j, sg * alpha, znorm, [&](int i) { return sg * d[refl ? km1 - i : i]; },
                     [&](int i) { return z[refl ? km1 - i : i]; }, orgr, taur, nit, sub, G, red);
    org_out = refl ? km1 - orgr : orgr;
    tau_out = sg * taur;
}

// d_i - lam_j from the stored origin value lamd_j = d[org_j] and tau_j, roots j = 0..k
template <class FA>
HD float arrow_delta(FA d, FA lamd, FA tau, int i, int j) {
    return (d[i] - lamd[j]) - tau[j];
}

// Gu / Eisenstat zeta-hat_i (>= 0) from the computed roots
// (sub, G, red): a group of G adjacent lanes shares one i, `red` multiplies the partial products across it.
template <class FA, class Red = ArrowNoReduce>
HD float arrow_zhat(int k, int i, FA d, FA lamd, FA tau, int sub = 0, int G = 1, Red red = Red()) {
    const float di = d[i];
    float w = (sub == 0) ? ((di - lamd[i]) - tau[i]) * -((di - lamd[i + 1]) - tau[i + 1])   // (d_i - lam_i)(lam_{i+1} - d_i)
                         : 1.0f;
#pragma unroll 4
    for (int j = sub; j < k; j += G) {
        const int jr = (j < i) ? j : j + 1;   // root paired with pole j
        const float q = fdiv_fast((di - lamd[jr]) - tau[jr], di - d[j]);
        w *= (j == i) ? 1.0f : q;
    }
    return sqrtf(fabsf(red(w)));
}

}  // namespace admmnet
