// Host (CPU) model of the divide & conquer tridiagonal eigensolver -- TEST HARNESS.
// Runs the per-thread building blocks of admm_net_amd/csrc/dc_core.h sequentially, phase by
// phase, exactly in the order the device kernel (dc.hip) runs them with team-parallel loops.
//   g++ -O2 -shared -fPIC -I admm_net_amd/csrc tests/host_model/dc_model.cpp -o tests/host_model/libdc_model.so
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "dc_core.h"

using namespace admmnet;

namespace {

constexpr int LS = 8;   // nominal leaf size (dc_leaf_start spreads the remainder; a single leaf has < 2*LS rows)

struct Ctx {
    int n;
    std::vector<float> lam, WA, WB;   // WT layout: W?[j * n + i] = W[i][j]
    std::vector<float> e0;            // original off-diagonals
    int stat_defl = 0, stat_k = 0, stat_iter = 0, stat_itmax = 0;
};

// merge blocks [a, b) and [b, c); reads `src`, writes `dst` (both WT layout), updates lam
int merge(Ctx &cx, int a, int b, int c, const std::vector<float> &srcW, std::vector<float> &dstW) {
    const int n = cx.n, nn = c - a, n1 = b - a;
    const float beta = cx.e0[b - 1];
    const float rho = 2.0f * std::fabs(beta);
    const float sgn = beta >= 0.f ? 1.f : -1.f;
    const float isq2 = 0.70710678118654752f;
    std::vector<float> z(nn), ds(nn), zs(nn), dl(nn), zl(nn);
    std::vector<int> perm(nn), src(nn);
    std::vector<DcRot> rot(nn);
    for (int i = 0; i < nn; ++i)
        z[i] = (i < n1 ? srcW[(size_t)(a + i) * n + (b - 1)] : sgn * srcW[(size_t)(a + i) * n + b]) * isq2;
    // rank of every entry in the merged order (two ascending lists, ties: left first)
    for (int i = 0; i < nn; ++i) {
        const float v = cx.lam[a + i];
        int r;
        if (i < n1) {
            r = i;
            for (int q = n1; q < nn; ++q) r += (cx.lam[a + q] < v);
        } else {
            r = i - n1;
            for (int q = 0; q < n1; ++q) r += (cx.lam[a + q] <= v);
        }
        perm[r] = i;
    }
    for (int p = 0; p < nn; ++p) {
        ds[p] = cx.lam[a + perm[p]];
        zs[p] = z[perm[p]];
    }
    int k = 0, nrot = 0;
    deflate_scan(nn, rho, ds.data(), zs.data(), dl.data(), zl.data(), src.data(), rot.data(), k, nrot);
    cx.stat_defl += nn - k;
    cx.stat_k += k;
    // working copy of the source columns (rotations modify them)
    std::vector<float> cols((size_t)nn * nn);
    for (int p = 0; p < nn; ++p)
        for (int i = 0; i < nn; ++i) cols[(size_t)p * nn + i] = srcW[(size_t)(a + perm[p]) * n + a + i];
    for (int r = 0; r < nrot; ++r) {
        float *x = &cols[(size_t)rot[r].pa * nn], *y = &cols[(size_t)rot[r].pb * nn];
        const float cc = rot[r].c, ss = rot[r].s;
        for (int i = 0; i < nn; ++i) {
            const float xi = x[i], yi = y[i];
            x[i] = cc * xi + ss * yi;
            y[i] = cc * yi - ss * xi;
        }
    }
    std::vector<float> tau(k), zh(k), U((size_t)k * k), vals(nn);
    std::vector<int> org(k);
    for (int j = 0; j < k; ++j) {
        int nit = 0;
        secular_root(k, j, rho, dl.data(), zl.data(), org[j], tau[j], &nit);
        cx.stat_iter += nit;
        cx.stat_itmax = std::max(cx.stat_itmax, nit);
        vals[j] = dl[org[j]] + tau[j];
    }
    for (int i = 0; i < k; ++i) zh[i] = lowner_zhat(k, i, dl.data(), zl.data(), org.data(), tau.data());
    for (int j = 0; j < k; ++j) {
        float nrm = 0.f;
        for (int i = 0; i < k; ++i) {
            const float u = zh[i] / dc_delta(dl.data(), org.data(), tau.data(), i, j);
            U[(size_t)i * k + j] = u;
            nrm += u * u;
        }
        const float inv = 1.0f / std::sqrt(nrm);
        for (int i = 0; i < k; ++i) U[(size_t)i * k + j] *= inv;
    }
    for (int p = k; p < nn; ++p) vals[p] = dl[p];
    // final positions: ascending, stable
    std::vector<int> rank(nn);
    for (int p = 0; p < nn; ++p) {
        int r = 0;
        for (int q = 0; q < nn; ++q) r += (vals[q] < vals[p]) || (vals[q] == vals[p] && q < p);
        rank[p] = r;
    }
    std::vector<float> newlam(nn);
    for (int p = 0; p < nn; ++p) {
        float *out = &dstW[(size_t)(a + rank[p]) * n + a];
        if (p < k) {
            for (int i = 0; i < nn; ++i) {
                float acc = 0.f;
                for (int kk = 0; kk < k; ++kk) acc += U[(size_t)kk * k + p] * cols[(size_t)src[kk] * nn + i];
                out[i] = acc;
            }
        } else {
            for (int i = 0; i < nn; ++i) out[i] = cols[(size_t)src[p] * nn + i];
        }
        newlam[rank[p]] = vals[p];
    }
    for (int p = 0; p < nn; ++p) cx.lam[a + p] = newlam[p];
    return 0;
}

}  // namespace

extern "C" {

// Team form of the deflation scan (defl_par_* in dc_core.h), emulated with `ts` sequential "threads" per phase,
// against the serial scan: returns 0 when every output is identical (bit for bit), 1 otherwise, 2 when the team
// form reported a conflict (the device then falls back to the serial scan).  out[0..1] = k, nrot of the serial scan.
int deflate_compare(int nn, float rho, const float *ds_in, const float *zs_in, int ts, int *out) {
    std::vector<float> ds(ds_in, ds_in + nn), zs(zs_in, zs_in + nn);
    float dmax = 0.f, zmax = 0.f;
    for (int i = 0; i < nn; ++i) {
        dmax = std::fmax(dmax, std::fabs(ds[i]));
        zmax = std::fmax(zmax, std::fabs(zs[i]));
    }
    std::vector<float> dl(nn, -7.f), zl(nn, -7.f), dl2(nn, -7.f), zl2(nn, -7.f);
    std::vector<int> src(nn, -7), src2(nn, -7);
    std::vector<DcRot> rot(nn), rot2(nn);
    int k = 0, nr = 0, k2 = 0, nr2 = 0;
    deflate_scan_tol(nn, rho, dmax, zmax, ds.data(), zs.data(), dl.data(), zl.data(), src.data(), rot.data(), k, nr);
    out[0] = k;
    out[1] = nr;
    const float tol = 8.0f * kEps32 * std::fmax(dmax, zmax);
    std::vector<int> pv(nn), flg(nn);
    std::vector<float> dvf(nn), zvf(nn), rc(nn), rs(nn), dde(nn);
    int conflict = 0;
    for (int tl = 0; tl < ts; ++tl)
        defl_par_flags(tl, ts, nn, rho, tol, ds.data(), zs.data(), pv.data(), flg.data(), dvf.data(), zvf.data());
    for (int tl = ts - 1; tl >= 0; --tl)   // (any order: walkers are independent)
        defl_par_walk(tl, ts, nn, tol, ds.data(), zs.data(), pv.data(), flg.data(), dvf.data(), zvf.data(), rc.data(),
                      rs.data(), dde.data(), &conflict);
    if (conflict) return 2;
    for (int tl = 0; tl < ts; ++tl)
        defl_par_emit(tl, ts, nn, ds.data(), pv.data(), flg.data(), dvf.data(), zvf.data(), rc.data(), rs.data(),
                      dde.data(), dl2.data(), zl2.data(), src2.data(), rot2.data(), &k2, &nr2);
    if (k != k2 || nr != nr2) return 1;
    for (int i = 0; i < nn; ++i) {
        if (std::memcmp(&dl[i], &dl2[i], 4) || src[i] != src2[i]) return 1;
        if (i < k && std::memcmp(&zl[i], &zl2[i], 4)) return 1;
    }
    for (int i = 0; i < nr; ++i)
        if (rot[i].pa != rot2[i].pa || rot[i].pb != rot2[i].pb || std::memcmp(&rot[i].c, &rot2[i].c, 4) ||
            std::memcmp(&rot[i].s, &rot2[i].s, 4))
            return 1;
    return 0;
}

// d[n], e[n] (e[i] couples i, i+1) -> lam[n] ascending, WT[n*n] with WT[j*n+i] = W[i][j].
// stats[4]: deflated count, secular roots solved, secular iterations (sum, max).  Returns 0 or the leaf status.
int dc_solve(int n, const float *d_in, const float *e_in, float *lam_out, float *WT_out, int *stats) {
    Ctx cx;
    cx.n = n;
    cx.lam.assign(n, 0.f);
    cx.WA.assign((size_t)n * n, 0.f);
    cx.WB.assign((size_t)n * n, 0.f);
    cx.e0.assign(e_in, e_in + n);
    std::vector<float> d(d_in, d_in + n);
    // sstedc's slascl, as dc_kernel does it: the power of two that brings max(|d|, |e|) into [0.5, 1)
    float orgnrm = 0.f, unscale = 1.f;
    for (int i = 0; i < n; ++i) orgnrm = std::fmax(orgnrm, std::fmax(std::fabs(d[i]), i < n - 1 ? std::fabs(cx.e0[i]) : 0.f));
    if (orgnrm > 0.f) {
        int ex;
        (void)std::frexp(orgnrm, &ex);
        ex = std::max(-120, std::min(120, ex));
        const float sc = std::ldexp(1.f, -ex);
        unscale = std::ldexp(1.f, ex);
        for (int i = 0; i < n; ++i) {
            d[i] *= sc;
            cx.e0[i] *= sc;
        }
    }
    const int nblk = dc_leaf_count(n);
    std::vector<int> bnd(nblk + 1);
    for (int b = 0; b < nblk; ++b) bnd[b] = dc_leaf_start(n, nblk, b);
    bnd[nblk] = n;
    for (int b = 1; b < nblk; ++b) {
        const int k = bnd[b];
        const float rho = std::fabs(cx.e0[k - 1]);
        d[k - 1] -= rho;
        d[k] -= rho;
    }
    for (int b = 0; b < nblk; ++b) {
        const int a = bnd[b], s = bnd[b + 1] - a;
        float dd[2 * LS], ee[2 * LS], Z[2 * LS * 2 * LS];
        for (int i = 0; i < s; ++i) {
            dd[i] = d[a + i];
            ee[i] = (i < s - 1) ? cx.e0[a + i] : 0.f;
        }
        auto Zacc = [&](int i, int j) -> float & { return Z[i * 2 * LS + j]; };
        const int st = leaf_ql(s, dd, ee, Zacc);
        if (st) return st;
        int idx[2 * LS];
        for (int i = 0; i < s; ++i) idx[i] = i;
        std::stable_sort(idx, idx + s, [&](int x, int y) { return dd[x] < dd[y]; });
        for (int j = 0; j < s; ++j) {
            cx.lam[a + j] = dd[idx[j]];
            for (int i = 0; i < s; ++i) cx.WA[(size_t)(a + j) * n + a + i] = Zacc(i, idx[j]);
        }
    }
    std::vector<int> cur(bnd);
    std::vector<float> *src = &cx.WA, *dst = &cx.WB;
    while ((int)cur.size() > 2) {
        std::vector<int> nxt;
        nxt.push_back(cur[0]);
        size_t i = 0;
        for (; i + 2 < cur.size(); i += 2) {
            merge(cx, cur[i], cur[i + 1], cur[i + 2], *src, *dst);
            nxt.push_back(cur[i + 2]);
        }
        if (i + 1 < cur.size()) {   // unpaired block: copy through
            const int a = cur[i], c = cur[i + 1];
            for (int j = a; j < c; ++j)
                for (int r = a; r < c; ++r) (*dst)[(size_t)j * n + r] = (*src)[(size_t)j * n + r];
            nxt.push_back(c);
        }
        cur.swap(nxt);
        std::swap(src, dst);
    }
    for (int i = 0; i < n; ++i) lam_out[i] = cx.lam[i] * unscale;
    std::memcpy(WT_out, src->data(), sizeof(float) * (size_t)n * n);
    if (stats) {
        stats[0] = cx.stat_defl;
        stats[1] = cx.stat_k;
        stats[2] = cx.stat_iter;
        stats[3] = cx.stat_itmax;
    }
    return 0;
}
}
