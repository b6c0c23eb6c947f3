// Host (CPU) model of the device eigensolver pipeline -- TEST HARNESS.
// Compiles admm_net_amd/csrc/eig_core.h with g++ and runs the same scalar
// cores (Householder generator, QL-with-rotation-log) sequentially, so the
// log format and the replay logic can be validated without a GPU.
//   g++ -O2 -shared -fPIC -I admm_net_amd/csrc tests/host_model/eigh_model.cpp -o tests/host_model/libeigh_model.so
#include <complex>
#include <cstring>
#include <vector>

#include "eig_core.h"

using cf = std::complex<float>;
using namespace admmnet;

static int g_variant = 1;

extern "C" {

void hm_set_variant(int v) { g_variant = v; }

// A: n x n row-major Hermitian (lower triangle read).  Outputs d[n], e[n] (e[n-1]=0),
// Q n x n row-major explicit unitary with A = Q T Q^H.
int hm_tridiag(int n, const cf *Ain, float *d, float *e, cf *Q) {
    std::vector<cf> A(Ain, Ain + (size_t)n * n);
    // fill upper from lower
    for (int i = 0; i < n; ++i) {
        A[i * n + i] = cf(A[i * n + i].real(), 0.f);
        for (int j = i + 1; j < n; ++j) A[i * n + j] = std::conj(A[j * n + i]);
    }
    std::vector<cf> taus(n, cf(0, 0));
    std::vector<cf> p(n), w(n), v(n);
    for (int i = 0; i < n - 1; ++i) {
        int m = n - i - 1;  // reflector length
        cf alpha = A[(i + 1) * n + i];
        float xn2 = 0.f;
        for (int r = i + 2; r < n; ++r) xn2 += std::norm(A[r * n + i]);
        float beta, tr, ti, sr, si;
        householder_c(alpha.real(), alpha.imag(), xn2, beta, tr, ti, sr, si);
        cf tau(tr, ti), sc(sr, si);
        e[i] = beta;
        d[i] = A[i * n + i].real();
        taus[i] = tau;
        v[0] = cf(1, 0);
        for (int r = 1; r < m; ++r) {
            v[r] = A[(i + 1 + r) * n + i] * sc;
            A[(i + 1 + r) * n + i] = v[r];
        }
        if (tr != 0.f || ti != 0.f) {
            for (int r = 0; r < m; ++r) {
                cf acc(0, 0);
                for (int c = 0; c < m; ++c) acc += A[(i + 1 + r) * n + (i + 1 + c)] * v[c];
                p[r] = tau * acc;
            }
            cf dot(0, 0);
            for (int r = 0; r < m; ++r) dot += std::conj(p[r]) * v[r];
            cf al = -0.5f * tau * dot;
            for (int r = 0; r < m; ++r) w[r] = p[r] + al * v[r];
            for (int r = 0; r < m; ++r)
                for (int c = 0; c < m; ++c)
                    A[(i + 1 + r) * n + (i + 1 + c)] -= v[r] * std::conj(w[c]) + w[r] * std::conj(v[c]);
        }
    }
    d[n - 1] = A[(n - 1) * n + (n - 1)].real();
    e[n - 1] = 0.f;
    // Q = H(0) H(1) ... H(n-2), backward accumulation
    for (int i = 0; i < n * n; ++i) Q[i] = cf(0, 0);
    for (int i = 0; i < n; ++i) Q[i * n + i] = cf(1, 0);
    for (int i = n - 2; i >= 0; --i) {
        int m = n - i - 1;
        v[0] = cf(1, 0);
        for (int r = 1; r < m; ++r) v[r] = A[(i + 1 + r) * n + i];
        cf tau = taus[i];
        for (int c = i + 1; c < n; ++c) {
            cf z(0, 0);
            for (int r = 0; r < m; ++r) z += std::conj(v[r]) * Q[(i + 1 + r) * n + c];
            z *= tau;
            for (int r = 0; r < m; ++r) Q[(i + 1 + r) * n + c] -= v[r] * z;
        }
    }
    return 0;
}

// d,e in/out (d -> eigenvalues), z0[n] in/out, log[cap] records, returns status.
int hm_tql(int n, float *d, float *e, float *z0, LogRec *log, int cap, int *nrec, int *nsweeps) {
    auto D = [&](int i) -> float & { return d[i]; };
    auto E = [&](int i) -> float & { return e[i]; };
    auto Z = [&](int i) -> float & { return z0[i]; };
    LogWriter lw{log, cap, 0, 0, 0};
    int ns = 0;
    // g_variant 0: textbook organisation (tql_lane); 1: the prefetching variant the device runs
    int st = g_variant ? tql_lane_pf<5>(n, D, E, Z, lw, 60, ns) : tql_lane(n, D, E, Z, lw, 60, ns);
    *nrec = lw.pos;
    *nsweeps = ns;
    return st;
}

// Replay the log on the rows of X (rows x n, row-major float, in place).
void hm_replay(int n, int rows, float *X, const LogRec *log, int nrec) {
    int pos = 0;
    while (pos < nrec) {
        const int g_hi = log[pos].h.g_hi, g_lo = log[pos].h.g_lo;
        pos += 8;
        for (int g = g_hi; g >= g_lo; --g) {
            for (int t = 0; t < 8; ++t) {
                const int i = 8 * g + 7 - t;
                const float c = log[pos].r.c, s = log[pos].r.s;
                ++pos;
                if (i + 1 >= n) continue;   // slot above the matrix: always identity
                for (int r = 0; r < rows; ++r) {
                    float *z = X + (size_t)r * n;
                    float f = z[i + 1];
                    z[i + 1] = s * z[i] + c * f;
                    z[i] = c * z[i] - s * f;
                }
            }
        }
    }
}

// Full pipeline: A -> w (unsorted), V (n x n row-major).  Returns tql status.
int hm_eigh(int n, const cf *A, float *w, cf *V, int *nrec_out, int *nsweeps_out) {
    std::vector<float> d(n), e(n), z0(n, 0.f);
    std::vector<cf> Q((size_t)n * n);
    hm_tridiag(n, A, d.data(), e.data(), Q.data());
    int cap = (3 * n * n + 64 * n + 64 + 7) & ~7;   // = the device's capacity (api.hip carve_chunk)
    std::vector<LogRec> log(cap);
    int nrec = 0, ns = 0;
    auto Dg = [&](int i) -> float { return d[i]; };
    auto Eg = [&](int i) -> float { return e[i]; };
    const bool flip0 = choose_flip(n, Dg, Eg);
    std::vector<float> dd(n), ee(n);
    int st = 1;
    bool flip = flip0;
    for (int attempt = 0; attempt < 2 && st != 0; ++attempt) {
        flip = flip0 ^ (attempt == 1);
        for (int i = 0; i < n; ++i) {
            dd[i] = d[flip ? n - 1 - i : i];
            ee[i] = (i < n - 1) ? (flip ? e[n - 2 - i] : e[i]) : 0.f;
            z0[i] = (i == (flip ? n - 1 : 0)) ? 1.f : 0.f;
        }
        st = hm_tql(n, dd.data(), ee.data(), z0.data(), log.data(), cap, &nrec, &ns);
    }
    d = dd;
    if (nrec_out) *nrec_out = nrec;
    if (nsweeps_out) *nsweeps_out = ns;
    if (st) return st;
    // rows of Q (columns reversed when flipped) as 2n real rows
    std::vector<float> X((size_t)2 * n * n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            const int sc = flip ? n - 1 - c : c;
            X[(size_t)r * n + c] = Q[r * n + sc].real();
            X[(size_t)(n + r) * n + c] = Q[r * n + sc].imag();
        }
    hm_replay(n, 2 * n, X.data(), log.data(), nrec);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) V[r * n + c] = cf(X[(size_t)r * n + c], X[(size_t)(n + r) * n + c]);
    // consistency of the separately tracked first row of W with row 0 of V (Q row 0 = e_0)
    for (int c = 0; c < n; ++c)
        if (std::abs(V[c].real() - z0[c]) > 1e-4f || std::abs(V[c].imag()) > 1e-4f) return 7;
    for (int c = 0; c < n; ++c) w[c] = d[c];
    return 0;
}
}
