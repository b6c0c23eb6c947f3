// peaks.hip -- batched grid peak search on the delay-Doppler spectrum of phi: regional maxima of the
// coarse grid + local refinement rounds, one workgroup per signal.
//   /root/reference/utils/peakSearchUtils.py:63-173 (alt_peak_search): coarse grid -> skimage
//   local_maxima(connectivity=2) (:118) -> coordinates in np.where row-major order (:119-126) -> `iter`
//   refinement rounds (:136-171): step *= reducefactor, window +-step clipped to [min, max - step],
//   np.arange grid, first arg-max in row-major order.  Rows of the result: (x = tau, y = f, height).
// The coarse spectrum comes from spectrum.hip (float64, same arithmetic as the local evaluations here);
// this kernel is O(grid) integer / compare work per signal plus a handful of D-term inner products per peak.
#include "common.h"

// numpy evaluates start + i * delta, k * step, ... with one rounding per operation: no FMA contraction in this
// file, so that grid coordinates (and hence the returned peak positions) are bit-identical to the reference's
#pragma clang fp contract(off)

namespace admmnet {

constexpr int PK_THREADS = 256;

// exp(j 2 pi fre_k), fre = numpy.linspace(0, (base - 1) x, base)[k]  (utils/mathUtils.py:4-21); the same
// expression as steer_table_kernel in spectrum.hip
__device__ __forceinline__ double2 pk_steer(double x, int k, int base) {
    const double stop = (double)(base - 1) * x;
    double fre = (base > 1) ? (double)k * (stop / (double)(base - 1)) : 0.0;
    if (base > 1 && k == base - 1) fre = stop;
    double s, c;
    sincos(2.0 * 3.14159265358979323846 * fre, &s, &c);
    return make_double2(c, s);
}

// |phi^H kron(s(y), conj d(x))|^2 at one point, separable form of spectrum.hip.  The delay steering vector
// is evaluated once per point when it fits 16 registers pairs (every geometry of the reference), else per use.
__device__ double pk_point(const double2 *ph, int xbase, int ybase, double x, double y) {
    double2 e[16];
    const bool cached = xbase <= 16;
    if (cached) {
#pragma unroll
        for (int kd = 0; kd < 16; ++kd) e[kd] = (kd < xbase) ? pk_steer(x, kd, xbase) : make_double2(0.0, 0.0);
    }
    double zr = 0.0, zi = 0.0;
    for (int ks = 0; ks < ybase; ++ks) {
        double ur = 0.0, ui = 0.0;
        if (cached) {
#pragma unroll
            for (int kd = 0; kd < 16; ++kd) {
                if (kd < xbase) {
                    const double2 a = ph[ks * xbase + kd];
                    ur += a.x * e[kd].x - a.y * e[kd].y;
                    ui -= a.x * e[kd].y + a.y * e[kd].x;
                }
            }
        } else {
            for (int kd = 0; kd < xbase; ++kd) {
                const double2 a = ph[ks * xbase + kd], ee = pk_steer(x, kd, xbase);
                ur += a.x * ee.x - a.y * ee.y;
                ui -= a.x * ee.y + a.y * ee.x;
            }
        }
        const double2 s = pk_steer(y, ks, ybase);
        zr += s.x * ur - s.y * ui;
        zi += s.x * ui + s.y * ur;
    }
    return zr * zr + zi * zi;
}

struct PeakOpts {
    double xmin, xmax, xstep, ymin, ymax, ystep, reduce;
    int iters, max_peaks;
};

__global__ __launch_bounds__(PK_THREADS) void peaks_kernel(const float2 *__restrict__ phi, int xbase, int ybase,
                                                           const double *__restrict__ Z, int nx, int ny,
                                                           const double *__restrict__ axis_x,
                                                           const double *__restrict__ axis_y, PeakOpts o,
                                                           double *__restrict__ peaks, int32_t *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int npix = nx * ny, D = xbase * ybase;
    double *img = reinterpret_cast<double *>(smem);                 // [ny][nx]
    double2 *ph = reinterpret_cast<double2 *>(img + npix);          // [D]
    int *plist = reinterpret_cast<int *>(ph + D);                   // [max_peaks] pixel index of peak k
    int *scan = plist + o.max_peaks;                                // [PK_THREADS + 1]
    unsigned char *cand = reinterpret_cast<unsigned char *>(scan + PK_THREADS + 1);   // [2][npix]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    const double *Zb = Z + b * (int64_t)npix;
    for (int i = tid; i < npix; i += PK_THREADS) img[i] = Zb[i];
    for (int i = tid; i < D; i += PK_THREADS) {   // (phi == nullptr: maxima of a caller-supplied image, no refinement)
        const float2 p = phi ? phi[b * D + i] : make_float2(0.f, 0.f);
        ph[i] = make_double2((double)p.x, (double)p.y);
    }
    __syncthreads();
    // ---- regional maxima, 8-connected, plateau aware (skimage local_maxima(connectivity=2)):
    //      a pixel is a candidate if no neighbour is larger; a plateau survives only if all of its pixels do
    const double v0 = img[0];
    int notflat = 0;
    for (int i = tid; i < npix; i += PK_THREADS) notflat |= (img[i] != v0);
    const int any_diff = __syncthreads_or(notflat);
    unsigned char *cur = cand, *nxt = cand + npix;
    for (int i = tid; i < npix; i += PK_THREADS) {
        const int r = i / nx, c = i - r * nx;
        const double v = img[i];
        bool ok = any_diff != 0;   // a constant image has no regional maximum
        for (int dr = -1; dr <= 1; ++dr)
            for (int dc = -1; dc <= 1; ++dc) {
                const int rr = r + dr, cc = c + dc;
                if ((dr | dc) == 0 || rr < 0 || rr >= ny || cc < 0 || cc >= nx) continue;
                ok = ok && (v >= img[rr * nx + cc]);
            }
        cur[i] = ok ? 1 : 0;
    }
    __syncthreads();
    for (int guard = 0; guard < npix; ++guard) {   // rejection spreads over plateaus until nothing changes
        int changed = 0;
        for (int i = tid; i < npix; i += PK_THREADS) {
            unsigned char keep = cur[i];
            if (keep) {
                const int r = i / nx, c = i - r * nx;
                const double v = img[i];
                for (int dr = -1; dr <= 1; ++dr)
                    for (int dc = -1; dc <= 1; ++dc) {
                        const int rr = r + dr, cc = c + dc;
                        if ((dr | dc) == 0 || rr < 0 || rr >= ny || cc < 0 || cc >= nx) continue;
                        const int q = rr * nx + cc;
                        if (img[q] == v && !cur[q]) keep = 0;
                    }
                changed |= !keep;
            }
            nxt[i] = keep;
        }
        unsigned char *t = cur;
        cur = nxt;
        nxt = t;
        if (!__syncthreads_or(changed)) break;
    }
    // ---- peaks in row-major order (np.where): each thread owns a contiguous run of pixels
    const int per = (npix + PK_THREADS - 1) / PK_THREADS;
    const int i0 = min(tid * per, npix), i1 = min(i0 + per, npix);
    int mine = 0;
    for (int i = i0; i < i1; ++i) mine += cur[i];
    scan[tid + 1] = mine;
    if (tid == 0) scan[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int t = 1; t <= PK_THREADS; ++t) scan[t] += scan[t - 1];
    __syncthreads();
    {
        int pos = scan[tid];
        for (int i = i0; i < i1; ++i)
            if (cur[i]) {
                if (pos < o.max_peaks) plist[pos] = i;
                ++pos;
            }
    }
    const int total = scan[PK_THREADS];
    if (tid == 0) counts[b] = total;
    __syncthreads();
    // ---- refinement: one wave per peak, lanes over the points of the local grid
    const int npk = min(total, o.max_peaks);
    double *out = peaks + b * (int64_t)o.max_peaks * 3;
    for (int k = wave; k < npk; k += PK_THREADS / 64) {
        const int pix = plist[k];
        const int r = pix / nx, c = pix - r * nx;
        double px = axis_x ? axis_x[c] : (double)c, py = axis_y ? axis_y[r] : (double)r, height = 0.0;
        double lx = o.xstep, ly = o.ystep;
        for (int it = 0; it < o.iters; ++it) {
            lx = o.reduce * lx;
            ly = o.reduce * ly;
            const double x0 = fmax(o.xmin, px - lx), x1 = fmin(o.xmax - lx, px + lx);
            const double y0 = fmax(o.ymin, py - ly), y1 = fmin(o.ymax - ly, py + ly);
            if (x0 >= x1 || y0 >= y1) continue;
            // numpy.arange(start, stop, step): len = ceil((stop - start) / step), values start + i * ((start + step) - start)
            const int nxl = (int)ceil((x1 - x0) / lx), nyl = (int)ceil((y1 - y0) / ly);
            if (nxl <= 0 || nyl <= 0) continue;
            const double dx = (x0 + lx) - x0, dy = (y0 + ly) - y0;
            double best = -1.0;
            int bidx = 0x7fffffff;
            for (int p = lane; p < nxl * nyl; p += 64) {
                const int pr = p / nxl, pc = p - pr * nxl;
                const double z = pk_point(ph, xbase, ybase, x0 + pc * dx, y0 + pr * dy);
                if (z > best) {   // first arg-max in row-major order: strictly greater replaces, ties keep the lower index
                    best = z;
                    bidx = p;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bidx, off, 64);
                if (ob > best || (ob == best && oi < bidx)) {
                    best = ob;
                    bidx = oi;
                }
            }
            const int pr = bidx / nxl, pc = bidx - pr * nxl;
            px = x0 + pc * dx;
            py = y0 + pr * dy;
            height = best;
        }
        if (lane == 0) {
            out[3 * k + 0] = px;
            out[3 * k + 1] = py;
            out[3 * k + 2] = height;
        }
    }
}

size_t peaks_lds_bytes(int npix, int D, int max_peaks) {
    return sizeof(double) * npix + sizeof(double2) * D + sizeof(int) * (max_peaks + PK_THREADS + 1) + 2 * (size_t)npix + 16;
}

int launch_peaks(const float2 *phi, int64_t B, int xbase, int ybase, const double *Z, int nx, int ny,
                 const double *axis_x, const double *axis_y, const double *opt7, int iters, int max_peaks,
                 double *peaks, int32_t *counts, hipStream_t st) {
    ProfScope _prof(KC_SPECTRUM, st);   // (bench accounting: the post-processing of cfg5 belongs to the "spectrum" class)
    if (B <= 0) return ADMMNET_OK;
    const size_t lds = peaks_lds_bytes(nx * ny, xbase * ybase, max_peaks);
    if (lds > 160 * 1024) {
        set_error("peak search: grid %d x %d (+ %d peaks) does not fit the LDS", nx, ny, max_peaks);
        return ADMMNET_E_ARG;
    }
    PeakOpts o{opt7[0], opt7[1], opt7[2], opt7[3], opt7[4], opt7[5], opt7[6], iters, max_peaks};
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(peaks_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(peaks_kernel, dim3((unsigned)B), dim3(PK_THREADS), lds, st, phi, xbase, ybase, Z, nx, ny,
                       axis_x, axis_y, o, peaks, counts);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
