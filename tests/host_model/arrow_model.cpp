// Host (CPU) model of the direct arrowhead eigensolver -- TEST HARNESS.
// Runs the per-thread building blocks of admm_net_amd/csrc/arrow_core.h (and the deflation scan of
// dc_core.h) sequentially, phase by phase, in the order the device kernel (arrow.hip) runs them.
//   g++ -O2 -shared -fPIC -I admm_net_amd/csrc tests/host_model/arrow_model.cpp -o tests/host_model/libarrow_model.so
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

#include "arrow_core.h"

using namespace admmnet;
typedef std::complex<float> cf;

extern "C" {

// A = [[alpha, z^H],[z, diag(h)]], z[D] interleaved complex, h[D].
// Outputs: lam[n] ascending, V[n*n] complex row-major (V[row][col], col = eigenvalue index),
// stats[4] = {k, deflated, rotations, max secular iterations}.
int arrow_solve(int D, float alpha, const float *z_ri, const float *h, float *lam_out, float *V_ri, int *stats) {
    const int n = D + 1;
    std::vector<float> zeta(D), phr(D), phi(D);
    for (int i = 0; i < D; ++i) {
        const float re = z_ri[2 * i], im = z_ri[2 * i + 1];
        const float a = std::sqrt(re * re + im * im);
        zeta[i] = a;
        phr[i] = a > 0.f ? re / a : 1.f;
        phi[i] = a > 0.f ? im / a : 0.f;
    }
    // sort h ascending (stable by rank counting, as the device does)
    std::vector<int> perm(D);
    for (int i = 0; i < D; ++i) {
        int r = 0;
        for (int q = 0; q < D; ++q) r += (h[q] < h[i]) || (h[q] == h[i] && q < i);
        perm[r] = i;
    }
    std::vector<float> ds(D), zs(D), dl(D), zl(D);
    std::vector<int> src(D);
    std::vector<DcRot> rot(D);
    float dmax = std::fabs(alpha), zmax = 0.f;
    for (int p = 0; p < D; ++p) {
        ds[p] = h[perm[p]];
        zs[p] = zeta[perm[p]];
        dmax = std::max(dmax, std::fabs(ds[p]));
        zmax = std::max(zmax, zs[p]);
    }
    int k = 0, nrot = 0;
    deflate_scan_tol(D, 1.0f, dmax, zmax, ds.data(), zs.data(), dl.data(), zl.data(), src.data(), rot.data(), k, nrot);
    // roots
    std::vector<int> org(k + 1);
    std::vector<float> tau(k + 1), lamd(k + 1), vals(n);
    float zn2 = 0.f;
    for (int i = 0; i < k; ++i) zn2 += zl[i] * zl[i];
    const float znorm = std::sqrt(zn2);
    int itmax = 0;
    if (k == 0) {
        vals[0] = alpha;
    } else {
        for (int j = 0; j <= k; ++j) {
            int nit = 0;
            arrow_root(k, j, alpha, znorm, dl.data(), zl.data(), org[j], tau[j], &nit);
            itmax = std::max(itmax, nit);
            lamd[j] = dl[org[j]];
            vals[j] = lamd[j] + tau[j];
        }
    }
    for (int p = k; p < D; ++p) vals[p + 1] = dl[p];
    // zeta-hat, norms
    std::vector<float> zh(k), x0(k + 1, 1.f);
    for (int i = 0; i < k; ++i) zh[i] = arrow_zhat(k, i, dl.data(), lamd.data(), tau.data());
    for (int j = 0; j <= k && k > 0; ++j) {
        float nrm = 1.f;
        for (int i = 0; i < k; ++i) {
            const float v = zh[i] / (-arrow_delta(dl.data(), lamd.data(), tau.data(), i, j));
            nrm += v * v;
        }
        x0[j] = 1.0f / std::sqrt(nrm);
    }
    // ranks (ascending, stable)
    std::vector<int> rnk(n);
    for (int s = 0; s < n; ++s) {
        int r = 0;
        for (int q = 0; q < n; ++q) r += (vals[q] < vals[s]) || (vals[q] == vals[s] && q < s);
        rnk[s] = r;
    }
    // eigenvectors in the sorted, rotated, real basis: X[row][slot], row 0 = arrow, row 1 + p = sorted pole p
    std::vector<float> X((size_t)n * n, 0.f);
    for (int j = 0; j <= k; ++j) {
        X[(size_t)0 * n + j] = x0[j];
        for (int i = 0; i < k; ++i)
            X[(size_t)(1 + src[i]) * n + j] = zh[i] * x0[j] / (-arrow_delta(dl.data(), lamd.data(), tau.data(), i, j));
    }
    for (int p = k; p < D; ++p) X[(size_t)(1 + src[p]) * n + (p + 1)] = 1.f;
    // undo the deflation rotations (reverse order): v = G^T v'
    for (int r = nrot - 1; r >= 0; --r) {
        float *xa = &X[(size_t)(1 + rot[r].pa) * n], *xb = &X[(size_t)(1 + rot[r].pb) * n];
        const float c = rot[r].c, s = rot[r].s;
        for (int q = 0; q < n; ++q) {
            const float a = xa[q], b = xb[q];
            xa[q] = c * a - s * b;
            xb[q] = s * a + c * b;
        }
    }
    // unsort, phases, order by eigenvalue
    for (int s = 0; s < n; ++s) {
        const int c = rnk[s];
        lam_out[c] = vals[s];
        V_ri[2 * ((size_t)0 * n + c)] = X[s];
        V_ri[2 * ((size_t)0 * n + c) + 1] = 0.f;
        for (int p = 0; p < D; ++p) {
            const int i = perm[p];
            const float x = X[(size_t)(1 + p) * n + s];
            V_ri[2 * ((size_t)(1 + i) * n + c)] = x * phr[i];
            V_ri[2 * ((size_t)(1 + i) * n + c) + 1] = x * phi[i];
        }
    }
    if (stats) {
        stats[0] = k;
        stats[1] = D - k;
        stats[2] = nrot;
        stats[3] = itmax;
    }
    return 0;
}
}
