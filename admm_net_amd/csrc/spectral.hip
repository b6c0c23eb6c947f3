// spectral.hip -- evaluation of the G-layer as a MATRIX FUNCTION instead of through an eigendecomposition (the default route of
// admmnet_layer_front; ADMMNET_SPECTRAL=0 turns it off).  Same result as GLayer.forward (/root/reference/admm_net.py:237-354: eigh, per-eigenvalue map f,
// V f(L) V^H) to fp32 rounding, for the matrices this network produces:
//
//   A = [[diag h, phi], [phi^H, corner]] - Z / rho  has all but TWO of its n eigenvalues in a bulk of relative width ~1e-4
//   (measured on the reference's own forward at every layer >= 1, default and perturbed weights, 10 x 10 .. 16 x 16:
//   tests/proto_spectral_shortcut.py, profiles/r03/spectral_shortcut_prototype.log) -- the arrowhead's pair of outliers,
//   and Z = sum of steps alpha (G - C) that are each "scalar x I + low rank + small".  With the outliers (lam_k, v_k) deflated,
//       E = A - c I - sum_k (lam_k - c) v_k v_k^H                    (c = mean of the bulk, ||E|| ~ 5e-3),
//       f(A) = f(c) (I - sum_k v_k v_k^H) + a1 E + a2 E^2 + sum_k f(lam_k) v_k v_k^H + O(f''' ||E||^3),
//   i.e. a two-vector subspace iteration (it converges like (||E|| / |lam_k - c|)^steps ~ 1e-3 per step, from the arrowhead's
//   own outlier pair as the start) and ONE Hermitian matrix product on the matrix cores replace tridiagonalisation,
//   divide & conquer, back-transform and rebuild (24 n^3 canonical flops, mostly latency-bound) by ~4 n^3 of GEMM.
//
// Safety: nothing is assumed -- every matrix is CHECKED and falls back to the eigensolver pipeline if
//   * the subspace iteration did not converge (residual of an outlier pair),
//   * the quadratic model of f on [c - d, c + d], d = ||E^2||_F^(1/2) >= ||E||_2 (rigorous), misses f at interior sample points by
//     more than the fp32 rounding of the result (this catches a wide bulk, more than two outliers, |lam| = 0 or a ReLU kink of
//     value_net inside the bulk, ...),
//   * anything is non-finite.
// The fallback needs no compaction: the eigen-pipeline kernels take the per-matrix flag array and the workgroups of matrices
// that are already done leave at once (Ws::skip).
//
// Layout: everything here is in the ORIGINAL index order of the state (arrow row / column last), full n x n complex row-major
// scratch matrices in the chunk buffers that the eigen-pipeline uses afterwards (Wdc: A then E, VT: E^2).
#include <stdlib.h>

#include "common.h"

namespace admmnet {

constexpr int SP_THREADS = 256;
constexpr int SP_ITERS = 4;        // matrix-vector passes of the subspace iteration (3 power steps + the final Rayleigh-Ritz)

bool use_spectral() {   // on by default; ADMMNET_SPECTRAL=0 sends every matrix through the eigensolver pipeline
    static const bool on = !(getenv("ADMMNET_SPECTRAL") && atoi(getenv("ADMMNET_SPECTRAL")) == 0);
    return on;
}

// ---- F1: A = C_g - inv_rho Z as a full Hermitian n x n matrix (Z: lower triangle valid if `lower`, else full) ------------
__global__ __launch_bounds__(SP_THREADS) void sp_build_kernel(int D, const float *__restrict__ lw, const float2 *__restrict__ phi,
                                                              const float *__restrict__ h, const float2 *__restrict__ Zg,
                                                              float2 *__restrict__ Ag) {
    __shared__ float2 tile[32][33];
    const int n = D + 1;
    const float corner = lw[S_CORNER_G], inv_rho = lw[S_INV_RHO_G];
    const int64_t b = blockIdx.x;
    const float2 *Z = Zg + b * (int64_t)n * n;
    float2 *A = Ag + b * (int64_t)n * n;
    const float2 *ph = phi + b * D;
    const float *hh = h + b * D;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int nt = (n + 31) >> 5;
    for (int I = 0; I < nt; ++I)
        for (int J = 0; J <= I; ++J) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 32 * I + ty + 8 * q, j = 32 * J + tx;
                if (i < n && j < n && j <= i) {
                    const float2 z = Z[(int64_t)i * n + j];
                    float2 c;
                    if (i < D) c = make_float2(i == j ? hh[i] : 0.f, 0.f);
                    else if (j == D) c = make_float2(corner, 0.f);
                    else c = make_float2(ph[j].x, -ph[j].y);                 // C[D][j] = conj(phi_j)
                    float2 a = make_float2(c.x - inv_rho * z.x, c.y - inv_rho * z.y);
                    if (i == j) a.y = 0.f;
                    A[(int64_t)i * n + j] = a;
                    tile[ty + 8 * q][tx] = a;
                } else {
                    tile[ty + 8 * q][tx] = make_float2(0.f, 0.f);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // the mirror: A[j][i] = conj(A[i][j]), written along i
                const int j = 32 * J + ty + 8 * q, i = 32 * I + tx;
                if (i < n && j < n && j < i) {
                    const float2 a = tile[tx][ty + 8 * q];
                    A[(int64_t)j * n + i] = make_float2(a.x, -a.y);
                }
            }
        }
}

// ---- F2: the two outlier eigenpairs by subspace iteration -------------------------------------------------------------
// vals[b][8] (double): lam0, lam1, c, res0, res1, trace, -, -;  vecs[b][2][n] complex64
__device__ __forceinline__ float sp_block_sum(float v, float *red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(SP_THREADS) void sp_outlier_kernel(int D, const float2 *__restrict__ Ag, const float2 *__restrict__ phi,
                                                                float2 *__restrict__ vecs, double *__restrict__ vals) {
    extern __shared__ float2 sm[];   // X[2][NP], Y[2][NP]
    __shared__ float red[4];
    __shared__ double sc[12];
    const int n = D + 1, NP = (n + 3) & ~3;
    float2 *X0 = sm, *X1 = sm + NP, *Y0 = sm + 2 * NP, *Y1 = sm + 3 * NP;
    const int64_t b = blockIdx.x;
    const float2 *A = Ag + b * (int64_t)n * n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // start: the outlier pair of the pure arrowhead lives in span{e_D, (phi, 0)}
    float pn = 0.f;
    for (int i = tid; i < D; i += SP_THREADS) {
        const float2 p = phi[b * D + i];
        pn += p.x * p.x + p.y * p.y;
    }
    pn = sp_block_sum(pn, red);
    const float ipn = pn > 0.f ? rsqrtf(pn) : 0.f;
    for (int i = tid; i < n; i += SP_THREADS) {
        X0[i] = make_float2(i == D ? 1.f : 0.f, 0.f);
        const float2 p = i < D ? phi[b * D + i] : make_float2(0.f, 0.f);
        // (phi = 0: any second direction will do; e_0 keeps the pair orthonormal)
        X1[i] = pn > 0.f ? make_float2(p.x * ipn, p.y * ipn) : make_float2(i == 0 ? 1.f : 0.f, 0.f);
    }
    __syncthreads();
    double trace = 0.0, cc = 0.0;
    for (int it = 0; it < SP_ITERS; ++it) {
        // Y = A X: one wave per row, lanes over the columns
        float tr_loc = 0.f;
        for (int i = wave; i < n; i += SP_THREADS / 64) {
            const float2 *row = A + (int64_t)i * n;
            float ar = 0.f, ai = 0.f, br = 0.f, bi = 0.f;
            for (int j = lane; j < n; j += 64) {
                const float2 a = row[j], x0 = X0[j], x1 = X1[j];
                ar = fmaf(a.x, x0.x, fmaf(-a.y, x0.y, ar));
                ai = fmaf(a.x, x0.y, fmaf(a.y, x0.x, ai));
                br = fmaf(a.x, x1.x, fmaf(-a.y, x1.y, br));
                bi = fmaf(a.x, x1.y, fmaf(a.y, x1.x, bi));
            }
            ar = wave_sum(ar); ai = wave_sum(ai); br = wave_sum(br); bi = wave_sum(bi);
            if (lane == 0) {
                Y0[i] = make_float2(ar, ai);
                Y1[i] = make_float2(br, bi);
                if (it == 0) tr_loc += row[i].x;
            }
        }
        if (it == 0) {
            const float t = sp_block_sum(lane == 0 ? tr_loc : 0.f, red);
            trace = (double)t;
        }
        __syncthreads();
        // H = X^H Y (2 x 2 Hermitian): h00, h11 real, h01 complex
        float h00 = 0.f, h11 = 0.f, h01r = 0.f, h01i = 0.f;
        for (int i = tid; i < n; i += SP_THREADS) {
            const float2 x0 = X0[i], x1 = X1[i], y0 = Y0[i], y1 = Y1[i];
            h00 += x0.x * y0.x + x0.y * y0.y;
            h11 += x1.x * y1.x + x1.y * y1.y;
            h01r += x0.x * y1.x + x0.y * y1.y;          // conj(x0) y1
            h01i += x0.x * y1.y - x0.y * y1.x;
        }
        h00 = sp_block_sum(h00, red);
        h11 = sp_block_sum(h11, red);
        h01r = sp_block_sum(h01r, red);
        h01i = sp_block_sum(h01i, red);
        if (tid == 0) {   // closed-form eigen-decomposition of [[h00, h01], [conj(h01), h11]] in double
            const double a = h00, d = h11, br_ = h01r, bi_ = h01i;
            const double ab = sqrt(br_ * br_ + bi_ * bi_);
            const double dif = 0.5 * (a - d), rad = sqrt(dif * dif + ab * ab);
            const double l0 = 0.5 * (a + d) - rad, l1 = 0.5 * (a + d) + rad;
            // eigenvector of l1: (cos t, e^{-i arg} sin t) with tan(2 t) = |b| / dif; of l0: (-sin t, e^{-i arg} cos t)
            double ct = 1.0, st = 0.0, er = 1.0, ei = 0.0;
            if (ab > 0.0) {
                const double th = 0.5 * atan2(ab, dif);
                ct = cos(th);
                st = sin(th);
                er = br_ / ab;
                ei = -bi_ / ab;     // conj(b) / |b|
            }
            sc[0] = l0; sc[1] = l1;
            // column 0 (l0): s00 = -st, s10 = (er, ei) ct ; column 1 (l1): s01 = ct, s11 = (er, ei) st
            sc[2] = ct; sc[3] = st; sc[4] = er; sc[5] = ei;
            sc[6] = (trace - l0 - l1) / (double)(n - 2);
        }
        __syncthreads();
        const float ct = (float)sc[2], st = (float)sc[3], er = (float)sc[4], ei = (float)sc[5];
        const float l0 = (float)sc[0], l1 = (float)sc[1];
        cc = sc[6];
        const float cf = (float)cc;
        // rotate X, Y into the Ritz basis; residuals; next X = Y - c X (not yet orthonormal)
        float r0 = 0.f, r1 = 0.f;
        for (int i = tid; i < n; i += SP_THREADS) {
            const float2 x0 = X0[i], x1 = X1[i], y0 = Y0[i], y1 = Y1[i];
            const float2 ex1 = make_float2(er * x1.x - ei * x1.y, er * x1.y + ei * x1.x);
            const float2 ey1 = make_float2(er * y1.x - ei * y1.y, er * y1.y + ei * y1.x);
            const float2 nx0 = make_float2(-st * x0.x + ct * ex1.x, -st * x0.y + ct * ex1.y);
            const float2 nx1 = make_float2(ct * x0.x + st * ex1.x, ct * x0.y + st * ex1.y);
            const float2 ny0 = make_float2(-st * y0.x + ct * ey1.x, -st * y0.y + ct * ey1.y);
            const float2 ny1 = make_float2(ct * y0.x + st * ey1.x, ct * y0.y + st * ey1.y);
            const float2 d0 = make_float2(ny0.x - l0 * nx0.x, ny0.y - l0 * nx0.y);
            const float2 d1 = make_float2(ny1.x - l1 * nx1.x, ny1.y - l1 * nx1.y);
            r0 += d0.x * d0.x + d0.y * d0.y;
            r1 += d1.x * d1.x + d1.y * d1.y;
            if (it + 1 < SP_ITERS) {
                X0[i] = make_float2(ny0.x - cf * nx0.x, ny0.y - cf * nx0.y);
                X1[i] = make_float2(ny1.x - cf * nx1.x, ny1.y - cf * nx1.y);
            } else {
                X0[i] = nx0;
                X1[i] = nx1;
            }
        }
        r0 = sp_block_sum(r0, red);
        r1 = sp_block_sum(r1, red);
        if (tid == 0) {
            sc[7] = sqrt((double)r0);
            sc[8] = sqrt((double)r1);
        }
        if (it + 1 < SP_ITERS) {   // Gram-Schmidt on the power step
            float n0 = 0.f;
            for (int i = tid; i < n; i += SP_THREADS) n0 += X0[i].x * X0[i].x + X0[i].y * X0[i].y;
            n0 = sp_block_sum(n0, red);
            const float in0 = n0 > 0.f ? rsqrtf(n0) : 0.f;
            float pr = 0.f, pi = 0.f;
            for (int i = tid; i < n; i += SP_THREADS) {
                const float2 x0 = make_float2(X0[i].x * in0, X0[i].y * in0), x1 = X1[i];
                X0[i] = x0;
                pr += x0.x * x1.x + x0.y * x1.y;      // conj(x0) x1
                pi += x0.x * x1.y - x0.y * x1.x;
            }
            pr = sp_block_sum(pr, red);
            pi = sp_block_sum(pi, red);
            float n1 = 0.f;
            for (int i = tid; i < n; i += SP_THREADS) {
                const float2 x0 = X0[i];
                const float2 x1 = make_float2(X1[i].x - (pr * x0.x - pi * x0.y), X1[i].y - (pr * x0.y + pi * x0.x));
                X1[i] = x1;
                n1 += x1.x * x1.x + x1.y * x1.y;
            }
            n1 = sp_block_sum(n1, red);
            const float in1 = n1 > 0.f ? rsqrtf(n1) : 0.f;
            for (int i = tid; i < n; i += SP_THREADS) X1[i] = make_float2(X1[i].x * in1, X1[i].y * in1);
            __syncthreads();
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += SP_THREADS) {
        vecs[(b * 2 + 0) * n + i] = X0[i];
        vecs[(b * 2 + 1) * n + i] = X1[i];
    }
    if (tid == 0) {
        double *v = vals + b * 8;
        v[0] = sc[0]; v[1] = sc[1]; v[2] = cc; v[3] = sc[7]; v[4] = sc[8]; v[5] = trace; v[6] = 0.0; v[7] = 0.0;
    }
}

// ---- F3: E = A - c I - sum_k (lam_k - c) v_k v_k^H, in place ---------------------------------------------------------
__global__ __launch_bounds__(SP_THREADS) void sp_deflate_kernel(int n, float2 *__restrict__ Ag, const float2 *__restrict__ vecs,
                                                                const double *__restrict__ vals) {
    extern __shared__ float2 sv[];   // v0[n], v1[n]
    const int64_t b = blockIdx.x;
    float2 *A = Ag + b * (int64_t)n * n;
    for (int i = threadIdx.x; i < 2 * n; i += SP_THREADS) sv[i] = vecs[b * 2 * n + i];
    __syncthreads();
    const float c = (float)vals[b * 8 + 2];
    const float m0 = (float)(vals[b * 8 + 0] - vals[b * 8 + 2]), m1 = (float)(vals[b * 8 + 1] - vals[b * 8 + 2]);
    const float2 *v0 = sv, *v1 = sv + n;
    for (int idx = threadIdx.x; idx < n * n; idx += SP_THREADS) {
        const int i = idx / n, j = idx - i * n;
        float2 a = A[idx];
        const float2 p0 = cmulc(v0[i], v0[j]), p1 = cmulc(v1[i], v1[j]);   // v[i] conj(v[j])
        a.x -= m0 * p0.x + m1 * p1.x;
        a.y -= m0 * p0.y + m1 * p1.y;
        if (i == j) {
            a.x -= c;
            a.y = 0.f;
        }
        A[idx] = a;
    }
}

// ---- F4: the lower triangle of E^2 = E^H E for the Hermitian E, and ||E^2||_F^2 ----------------------------------------------
// O[i][j] = sum_c conj(E[c][i]) E[c][j]: both operands are read along rows of E (coalesced), 32 x 32 tiles of the lower
// triangle dealt round-robin to the four waves, four real v_mfma_f32_32x32x2_f32 per complex k-step.
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ __launch_bounds__(SP_THREADS) void sp_square_kernel(int n, const float2 *__restrict__ Eg, float2 *__restrict__ Og,
                                                               double *__restrict__ vals) {
    __shared__ float red[4];
    const int64_t b = blockIdx.x;
    const float2 *E = Eg + b * (int64_t)n * n;
    float2 *O = Og + b * (int64_t)n * n;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r32 = lane & 31, kh = lane >> 5;
    const int nt = (n + 31) >> 5, ntri = nt * (nt + 1) / 2;
    float fro = 0.f;
    for (int t = wave; t < ntri; t += SP_THREADS / 64) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int J = t - I * (I + 1) / 2;
        const int i = 32 * I + r32, j = 32 * J + r32;
        const bool iv = i < n, jv = j < n;
        f32x16 aRe = {0}, aIm = {0};
#pragma unroll 4
        for (int c0 = 0; c0 < n; c0 += 2) {
            const int c = c0 + kh;
            const bool cv = c < n;
            const float2 *row = E + (int64_t)(cv ? c : 0) * n;
            float2 ei = row[iv ? i : 0], ej = row[jv ? j : 0];
            if (!(cv && iv)) ei = make_float2(0.f, 0.f);
            if (!(cv && jv)) ej = make_float2(0.f, 0.f);
            // conj(ei) ej = (er yr + ei yi) + i (er yi - ei yr)
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(ei.x, ej.x, aRe, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(ei.x, ej.y, aIm, 0, 0, 0);
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(ei.y, ej.y, aRe, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(-ei.y, ej.x, aIm, 0, 0, 0);
        }
        // C/D layout: column = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = 32 * I + (q & 3) + 8 * (q >> 2) + 4 * kh, gj = 32 * J + r32;
            if (gi < n && gj < n && gi >= gj) {
                const float2 o = make_float2(aRe[q], gi == gj ? 0.f : aIm[q]);
                O[(int64_t)gi * n + gj] = o;
                fro += (gi == gj ? 1.f : 2.f) * (o.x * o.x + o.y * o.y);
            }
        }
    }
    fro = sp_block_sum(fro, red);
    if (tid == 0) vals[b * 8 + 6] = (double)fro;
}

// ---- F5: checks, then G = a0 I + a1 E + a2 E^2 + sum_k (f(lam_k) - a0) v_k v_k^H and ||G - C_z||_F ------------------------
__device__ inline double sp_eig_map(double w, double thr, const float *vn) {   // rebuild_lds.h: br_eig_map, in double
    const double x = w - thr;
    const double base = x > 20.0 ? x : log1p(exp(x));
    const double a = fabs(w);
    double acc = vn[48];
    for (int j = 0; j < 16; ++j) {
        const double pre = (double)vn[j] * a + (double)vn[16 + j];
        acc += (double)vn[32 + j] * (pre > 0.0 ? pre : 0.0);
    }
    return base / (1.0 + exp(-acc));
}

__global__ __launch_bounds__(SP_THREADS) void sp_assemble_kernel(int D, const float *__restrict__ lw, const float2 *__restrict__ Eg,
                                                                 const float2 *__restrict__ E2g, const float2 *__restrict__ vecs,
                                                                 const double *__restrict__ vals, const float2 *__restrict__ phi,
                                                                 const float *__restrict__ h, float2 *__restrict__ G,
                                                                 float *__restrict__ rn, int *__restrict__ flag,
                                                                 int32_t *__restrict__ status, int lower_only, float tol) {
    extern __shared__ float2 sv[];   // v0[n], v1[n]
    __shared__ float red[4];
    __shared__ float coef[8];
    __shared__ int ok;
    const int n = D + 1;
    const int64_t b = blockIdx.x;
    const float2 *E = Eg + b * (int64_t)n * n, *E2 = E2g + b * (int64_t)n * n;
    const int tid = threadIdx.x;
    for (int i = tid; i < 2 * n; i += SP_THREADS) sv[i] = vecs[b * 2 * n + i];
    if (tid == 0) {
        const double fro = vals[b * 8 + 6];                    // ||E^2||_F^2 from sp_square_kernel
        const LayerLayout L{D};
        const float *vn = lw + L.off_vn();
        const double thr = lw[S_THR];
        const double *v = vals + b * 8;
        const double l0 = v[0], l1 = v[1], c = v[2];
        const double delta = sqrt(sqrt(fro));                  // ||E||_2 <= ||E^2||_F^(1/2)
        bool good = isfinite(l0) && isfinite(l1) && isfinite(c) && isfinite(delta) && delta > 0.0;
        int why = good ? 0 : 8;
        // the outlier pairs must have converged: residual against the gap to the bulk
        const double g0 = fabs(l0 - c), g1 = fabs(l1 - c);
        if (good && !(v[3] <= 1e-5 * fmax(g0, 1e-30) && v[4] <= 1e-5 * fmax(g1, 1e-30))) good = false, why = 1;
        // ... and be outliers indeed (the bulk must be narrow against its distance to them)
        if (good && !(delta < 0.05 * fmin(g0, g1))) good = false, why = 2;
        double a0 = 0, a1 = 0, a2 = 0, f0k = 0, f1k = 0;
        if (good) {
            const double fm = sp_eig_map(c - delta, thr, vn), fc = sp_eig_map(c, thr, vn), fp = sp_eig_map(c + delta, thr, vn);
            a0 = fc;
            a1 = (fp - fm) / (2.0 * delta);
            a2 = (fp - 2.0 * fc + fm) / (2.0 * delta * delta);
            f0k = sp_eig_map(l0, thr, vn);
            f1k = sp_eig_map(l1, thr, vn);
            // the quadratic through (c - d, c, c + d) against f at interior points: a smooth f misses it by ~ f''' d^3 / 16 there
            // (the truncation error of the series on the bulk is of that size); a kink of f (|lam| = 0, a ReLU of value_net)
            // or a bulk too wide for two terms shows as a miss
            const double scale = fmax(fmax(fabs(fc), fmax(fabs(f0k), fabs(f1k))), 1e-6);
            double miss = 0.0;
            const double ts[6] = {-0.75, -0.5, -0.25, 0.25, 0.5, 0.75};
            for (int q = 0; q < 6; ++q) {
                const double t = ts[q] * delta;
                miss = fmax(miss, fabs(sp_eig_map(c + t, thr, vn) - (a0 + a1 * t + a2 * t * t)));
            }
            if (!(miss <= (double)tol * scale)) good = false, why = 4;
        }
        coef[0] = (float)a0; coef[1] = (float)a1; coef[2] = (float)a2; coef[3] = (float)(f0k - a0); coef[4] = (float)(f1k - a0);
        ok = good ? 1 : 0;
        flag[b] = good ? 0 : why;   // (non-zero: the eigen-pipeline runs this matrix; the value says which check failed)
        if (status) {
            atomicAdd(status + (good ? 2 : 1), 1);
            if (why == 4) atomicAdd(status + 3, 1);   // [3]: of [1], those where f is not a quadratic on the bulk
        }
    }
    __syncthreads();
    if (!ok) return;
    const float a0 = coef[0], a1 = coef[1], a2 = coef[2], d0 = coef[3], d1 = coef[4];
    const float2 *v0 = sv, *v1 = sv + n;
    const LayerLayout L{D};
    const float corner_z = lw[S_CORNER_Z];
    float2 *Gb = G + b * (int64_t)n * n;
    float acc = 0.f;
    for (int idx = tid; idx < n * n; idx += SP_THREADS) {   // the lower triangle; the mirror is written where G is kept in full
        const int i = idx / n, j = idx - i * n;
        if (j > i) continue;
        const float2 e = E[idx], e2 = E2[idx];
        const float2 p0 = cmulc(v0[i], v0[j]), p1 = cmulc(v1[i], v1[j]);
        float2 g = make_float2(a1 * e.x + a2 * e2.x + d0 * p0.x + d1 * p1.x, a1 * e.y + a2 * e2.y + d0 * p0.y + d1 * p1.y);
        if (i == j) {
            g.x += a0;
            g.y = 0.f;
        }
        Gb[idx] = g;
        if (!lower_only && j < i) Gb[(int64_t)j * n + i] = make_float2(g.x, -g.y);
        {   // residual against C_z = [[diag h, phi], [phi^H, corner_z]] over the lower triangle, off-diagonal twice
            float2 c;
            if (i < D) c = make_float2(i == j ? h[b * D + i] : 0.f, 0.f);
            else if (j == D) c = make_float2(corner_z, 0.f);
            else c = make_float2(phi[b * D + j].x, -phi[b * D + j].y);
            const float dr = g.x - c.x, di = g.y - c.y;
            acc += (i == j ? 1.f : 2.f) * (dr * dr + di * di);
        }
    }
    acc = sp_block_sum(acc, red);
    if (tid == 0) rn[b] = sqrtf(acc);
}

// Scratch of its own (Ws::spec_*, carved only when the path is on).  Returns with flag[] filled; the caller runs the
// eigen-pipeline with Ws::skip = flag.
int launch_spectral(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *Z, float2 *G,
                    float *rn, const Ws &ws, int32_t *status, hipStream_t st, bool lower_only, const float *alpha,
                    const float2 *phi_prev, const float *h_prev, const float *lw_prev, int update_mode) {
    ProfScope _prof(KC_GFUNC, st);
    if (nb <= 0) return ADMMNET_OK;
    const int n = D + 1;
    // model tolerance: the quadratic may miss f on the bulk by 1e-6 of the result's scale -- below the eigensolver route's own
    // rounding per layer (~2e-6) and without effect on the distance to the float64 oracle (3e-7, 1e-6 and 3e-6 measured the same:
    // tests/gpu_spectral_check.py); at K = 32 the tighter 3e-7 rejected 3.0 % of the matrix-layers, this one 0.4 %
    static const float tol = getenv("ADMMNET_SPECTRAL_TOL") ? (float)atof(getenv("ADMMNET_SPECTRAL_TOL")) : 1e-6f;
    if (use_spectral_fused()) {
        if (!ws.spec_flag || !lower_only) {
            set_error("spectral: the fused kernel needs the flag buffer and the lower-triangle state");
            return ADMMNET_E_WORKSPACE;
        }
        return launch_spectral_fused(D, nb, lw, phi, h, Z, G, rn, ws.spec_flag, status, tol, alpha, phi_prev, h_prev, lw_prev,
                                     update_mode, st);
    }
    if (update_mode) {
        set_error("spectral: the multi-kernel form does not apply the Z update");
        return ADMMNET_E_ARG;
    }
    if (!ws.spec_flag || !ws.spec_vec || !ws.spec_val || !ws.spec_mat) {
        set_error("spectral: workspace without the fast-path buffers");
        return ADMMNET_E_WORKSPACE;
    }
    float2 *A = ws.spec_mat;                                  // [chunk][n][n]: A, then E in place
    float2 *E2 = ws.spec_mat + (int64_t)ws.chunk * n * n;     // [chunk][n][n]
    hipLaunchKernelGGL(sp_build_kernel, dim3((unsigned)nb), dim3(SP_THREADS), 0, st, D, lw, phi, h, Z, A);
    ADMM_HIP(hipGetLastError());
    const int NP = (n + 3) & ~3;
    hipLaunchKernelGGL(sp_outlier_kernel, dim3((unsigned)nb), dim3(SP_THREADS), sizeof(float2) * 4 * NP, st, D, A, phi,
                       ws.spec_vec, ws.spec_val);
    ADMM_HIP(hipGetLastError());
    hipLaunchKernelGGL(sp_deflate_kernel, dim3((unsigned)nb), dim3(SP_THREADS), sizeof(float2) * 2 * n, st, n, A, ws.spec_vec,
                       ws.spec_val);
    ADMM_HIP(hipGetLastError());
    hipLaunchKernelGGL(sp_square_kernel, dim3((unsigned)nb), dim3(SP_THREADS), 0, st, n, A, E2, ws.spec_val);
    ADMM_HIP(hipGetLastError());
    hipLaunchKernelGGL(sp_assemble_kernel, dim3((unsigned)nb), dim3(SP_THREADS), sizeof(float2) * 2 * n, st, D, lw, A, E2,
                       ws.spec_vec, ws.spec_val, phi, h, G, rn, ws.spec_flag, status, lower_only ? 1 : 0, tol);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
