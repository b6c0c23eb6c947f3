// spectrum.hip -- batched delay-Doppler spectrum |phi^H kron(s(f), conj d(tau))|^2.
//   /root/reference/utils/peakSearchUtils.py:9-33 (peak_search_func) evaluated on
//   a whole (tau, f) grid (peak_search :37-60, a double Python loop with one
//   kron + dot per grid point in the reference).  The steering vectors are
//   Vandermonde (utils/mathUtils.py:4-21), so the D-long inner product factors:
//     U[ks][ix] = sum_kd conj(phi[ks][kd]) conj(d_ix[kd]),  z[iy][ix] = sum_ks s_iy[ks] U[ks][ix].
// Evaluated in float64 like the reference (numpy complex128) so that the
// regional-maxima index search that follows is decided by the same ordering of
// grid values.  Cost is O(D nx + ny ybase nx) per signal -- negligible next to
// the eigensolver; VALU fp64 code, tables L2 resident.
#include "common.h"

namespace admmnet {

__global__ void steer_table_kernel(const double *__restrict__ freqs, int nf, int base,
                                   double2 *__restrict__ tab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nf * base) return;
    const int i = idx / base, k = idx - i * base;
    // vander_vec(0, (base-1)*x, base): linspace(0, (base-1) x, base)[k]
    const double stop = (double)(base - 1) * freqs[i];
    double fre = (base > 1) ? (double)k * (stop / (double)(base - 1)) : 0.0;
    if (base > 1 && k == base - 1) fre = stop;   // numpy.linspace pins the end point
    double s, c;
    sincos(2.0 * 3.14159265358979323846 * fre, &s, &c);
    tab[idx] = make_double2(c, s);
}

constexpr int SP_THREADS = 256;
constexpr int SP_XCHUNK = 64;

__global__ __launch_bounds__(SP_THREADS) void spectrum_kernel(const float2 *__restrict__ phi, int xbase,
                                                              int ybase, const double2 *__restrict__ tabD,
                                                              int nx, const double2 *__restrict__ tabS,
                                                              int ny, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2 *ph = reinterpret_cast<double2 *>(smem);   // [D]
    double2 *U = ph + xbase * ybase;                   // [ybase][SP_XCHUNK]
    const int D = xbase * ybase;
    const int64_t b = blockIdx.x;
    for (int i = threadIdx.x; i < D; i += SP_THREADS) {
        const float2 p = phi[b * D + i];
        ph[i] = make_double2((double)p.x, (double)p.y);
    }
    __syncthreads();
    for (int x0 = 0; x0 < nx; x0 += SP_XCHUNK) {
        const int xw = min(SP_XCHUNK, nx - x0);
        for (int p = threadIdx.x; p < ybase * xw; p += SP_THREADS) {
            const int ks = p / xw, xl = p - ks * xw;
            const double2 *d = tabD + (int64_t)(x0 + xl) * xbase;
            double ur = 0.0, ui = 0.0;
            for (int kd = 0; kd < xbase; ++kd) {
                // conj(phi) * conj(d) = conj(phi * d)
                const double2 a = ph[ks * xbase + kd], e = d[kd];
                ur += a.x * e.x - a.y * e.y;
                ui -= a.x * e.y + a.y * e.x;
            }
            U[ks * SP_XCHUNK + xl] = make_double2(ur, ui);
        }
        __syncthreads();
        for (int p = threadIdx.x; p < ny * xw; p += SP_THREADS) {
            const int iy = p / xw, xl = p - iy * xw;
            const double2 *s = tabS + (int64_t)iy * ybase;
            double zr = 0.0, zi = 0.0;
            for (int ks = 0; ks < ybase; ++ks) {
                const double2 a = s[ks], u = U[ks * SP_XCHUNK + xl];
                zr += a.x * u.x - a.y * u.y;
                zi += a.x * u.y + a.y * u.x;
            }
            out[(b * ny + iy) * (int64_t)nx + x0 + xl] = zr * zr + zi * zi;
        }
        __syncthreads();
    }
}

int launch_spectrum_tables(const double *taus, int nx, int xbase, const double *fs, int ny, int ybase,
                           double2 *tabD, double2 *tabS, hipStream_t st) {
    hipLaunchKernelGGL(steer_table_kernel, dim3((nx * xbase + 255) / 256), dim3(256), 0, st, taus, nx, xbase,
                       tabD);
    hipLaunchKernelGGL(steer_table_kernel, dim3((ny * ybase + 255) / 256), dim3(256), 0, st, fs, ny, ybase,
                       tabS);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_spectrum_main(const float2 *phi, int64_t B, int xbase, int ybase, const double2 *tabD, int nx,
                         const double2 *tabS, int ny, double *out, hipStream_t st) {
    ProfScope _prof(KC_SPECTRUM, st);
    if (B <= 0) return ADMMNET_OK;
    const size_t lds = sizeof(double2) * ((size_t)xbase * ybase + (size_t)ybase * SP_XCHUNK);
    if (lds > 150 * 1024) {
        set_error("spectrum: xbase*ybase too large for LDS");
        return ADMMNET_E_ARG;
    }
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(spectrum_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spectrum_kernel, dim3((unsigned)B), dim3(SP_THREADS), lds, st, phi, xbase, ybase, tabD,
                       nx, tabS, ny, out);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
