// Sanitizer driver (the GPU pool offers no device sanitizer, so the scalar eigensolver cores --
// eig_core.h / dc_core.h / arrow_core.h, the code the HIP kernels share -- run under AddressSanitizer +
// UndefinedBehaviorSanitizer on the CPU).  Built by tests/test_host_logic.py::test_host_models_under_sanitizers as
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I admm_net_amd/csrc sanitize_main.cpp
//       eigh_model.cpp dc_model.cpp arrow_model.cpp
// Exercises every extern "C" entry of the three host models on seeded inputs, checks the results loosely and returns 0.
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

typedef std::complex<float> cf;
extern "C" {
void hm_set_variant(int v);
int hm_eigh(int n, const cf *A, float *w, cf *V, int *nrec_out, int *nsweeps_out);
int dc_solve(int n, const float *d_in, const float *e_in, float *lam_out, float *WT_out, int *stats);
int arrow_solve(int D, float alpha, const float *z_ri, const float *h, float *lam_out, float *V_ri, int *stats);
}

int main() {
    std::mt19937 gen(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int variant = 0; variant < 2; ++variant) {
        hm_set_variant(variant);
        for (int n : {2, 3, 17, 33, 65}) {
            std::vector<cf> A((size_t)n * n), V((size_t)n * n);
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) {
                    const cf x(nd(gen), i == j ? 0.f : nd(gen));
                    A[(size_t)i * n + j] = x;
                    A[(size_t)j * n + i] = std::conj(x);
                }
            std::vector<float> w(n);
            int nrec = 0, ns = 0;
            if (hm_eigh(n, A.data(), w.data(), V.data(), &nrec, &ns)) return 10;
            float tr = 0.f, sw = 0.f;
            for (int i = 0; i < n; ++i) {
                tr += A[(size_t)i * n + i].real();
                sw += w[i];
            }
            if (std::fabs(tr - sw) > 1e-3f * n) return 11;
        }
    }
    for (int n : {8, 9, 31, 64, 129, 257}) {
        std::vector<float> d(n), e(n, 0.f), lam(n), WT((size_t)n * n);
        for (int i = 0; i < n; ++i) d[i] = nd(gen);
        for (int i = 0; i + 1 < n; ++i) e[i] = nd(gen);
        if (n == 31)
            for (int i = 0; i < n; ++i) d[i] = 0.25f;   // a cluster: deflation paths
        int stats[4] = {0, 0, 0, 0};
        if (dc_solve(n, d.data(), e.data(), lam.data(), WT.data(), stats)) return 20;
        for (int i = 0; i + 1 < n; ++i)
            if (!(lam[i] <= lam[i + 1])) return 21;
    }
    for (int D : {1, 2, 16, 100, 128, 256}) {
        std::vector<float> z(2 * D), h(D), lam(D + 1), V((size_t)2 * (D + 1) * (D + 1));
        for (int i = 0; i < D; ++i) {
            z[2 * i] = 0.1f * nd(gen);
            z[2 * i + 1] = 0.1f * nd(gen);
            h[i] = std::fabs(nd(gen)) + 0.05f;
        }
        if (D >= 16) {
            h[3] = h[2];          // equal poles: rotation deflation
            z[10] = z[11] = 0.f;  // zero coupling: trivial deflation
        }
        int stats[4] = {0, 0, 0, 0};
        if (arrow_solve(D, 9.9f, z.data(), h.data(), lam.data(), V.data(), stats)) return 30;
        for (int i = 0; i < D; ++i)
            if (!(lam[i] <= lam[i + 1])) return 31;
    }
    std::puts("sanitize ok");
    return 0;
}
