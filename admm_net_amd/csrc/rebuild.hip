// rebuild.hip -- K4: learned eigenvalue map + G = V f(Lambda) V^H + ||G - C||_F.
//
//   f(lambda) = softplus(lambda - sigmoid(threshold)) * value_net(|lambda|)
//       /root/reference/admm_net.py:310-334  (one MLP 1->16->1 per eigenvalue,
//       a Python loop of 2n addmm launches in the reference)
//   G = V diag(f) V^H, G = (G + G^H)/2            admm_net.py:336-354
//   r = ||G - [[diag h, phi],[phi^H, corner_z]]||_F  admm_net.py:400-403,454
//
// One workgroup (4 waves) per matrix.  The D x D block is computed on the
// f32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 products) as 32x32
// tiles of the LOWER triangle only (the Hermitian mirror is written from the
// same accumulators, which also makes G exactly Hermitian as the reference's
// explicit symmetrisation does).  Complex product in the 4M form:
//   Re = (Xr f) Yr^T + (Xi f) Yi^T,  Im = (Xi f) Yr^T - (Xr f) Yi^T.
// Operands come straight from the planar transposed VT[c][rho] image, which
// is exactly the MFMA A/B lane layout (lane&31 = row, lane>>5 = k), so loads
// are 128-byte coalesced segments with no LDS staging.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace admmnet {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int RB_THREADS = 256;

__device__ __forceinline__ float eig_map(float w, float thr, const float *vn) {
    // vn: w1[16] b1[16] w2[16] b2[1]
    const float base = softplus_f(w - thr);
    const float a = fabsf(w);
    float acc = vn[48];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = fmaf(vn[32 + j], fmaxf(fmaf(vn[j], a, vn[16 + j]), 0.f), acc);
    return base * sigmoid_f(acc);
}

__global__ __launch_bounds__(RB_THREADS) void rebuild_kernel(int D, const float *__restrict__ lw,
                                                             const float *__restrict__ QV,
                                                             const float *__restrict__ wv,
                                                             const float *__restrict__ w0v,
                                                             const float2 *__restrict__ phi,
                                                             const float *__restrict__ h,
                                                             float2 *__restrict__ G, float *__restrict__ rn,
                                                             int lower_only) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = D + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b = blockIdx.x;
    float *fs = reinterpret_cast<float *>(smem);   // [n+1] f(lambda)
    float *w0f = fs + ((n + 4) & ~3);              // [n+1] w0 * f
    float *z0s = w0f + ((n + 4) & ~3);             // [n+1] w0
    float *rowb = z0s + ((n + 4) & ~3);            // [2D] last-row staging
    float *redb = rowb + 2 * D;                    // [8]
    const LayerLayout L{D};
    const float thr = lw[S_THR];
    const float *vn = lw + L.off_vn();
    for (int c = tid; c <= n; c += RB_THREADS) {
        float f = 0.f, z0 = 0.f;
        if (c < n) {
            f = eig_map(wv[b * n + c], thr, vn);
            z0 = w0v[b * n + c];
        }
        fs[c] = f;
        w0f[c] = z0 * f;
        z0s[c] = z0;
    }
    __syncthreads();

    const float *VT = QV + b * ((int64_t)n * 2 * D);
    float2 *Gb = G + b * (int64_t)n * n;
    const int pitch = 2 * D;
    float acc2 = 0.f;

    // ---- D x D block on the matrix cores
    const int NT = (D + 31) / 32;
    const int ntiles = NT * (NT + 1) / 2;
    const int r = lane & 31, kh = lane >> 5;
    for (int t = wave; t < ntiles; t += RB_THREADS / 64) {
        // decode lower-triangular tile index t -> (I, J), I >= J
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int J = t - I * (I + 1) / 2;
        const int i0 = 32 * I, j0 = 32 * J;
        const bool iv = (i0 + r) < D, jv = (j0 + r) < D;
        const int xo = iv ? (i0 + r) : 0, yo = jv ? (j0 + r) : 0;
        f32x16 aRe = {0}, aIm = {0};
#pragma unroll 4
        for (int kk = 0; kk < n; kk += 2) {
            const int c = kk + kh;
            const bool cv = c < n;
            const int cc = cv ? c : 0;
            const float fc = cv ? fs[cc] : 0.f;
            const float *row = VT + (int64_t)cc * pitch;
            float xr = row[xo], xi = row[D + xo];
            float yr = row[yo], yi = row[D + yo];
            xr = iv ? xr * fc : 0.f;
            xi = iv ? xi * fc : 0.f;
            yr = jv ? yr : 0.f;
            yi = jv ? yi : 0.f;
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, yr, aRe, 0, 0, 0);
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, yi, aRe, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, yr, aIm, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(-xr, yi, aIm, 0, 0, 0);
        }
        // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = i0 + (q & 3) + 8 * (q >> 2) + 4 * kh;
            const int gj = j0 + r;
            if (gi < D && gj < D && gi >= gj) {
                float re = aRe[q], im = aIm[q];
                if (gi == gj) {
                    im = 0.f;
                    Gb[(int64_t)gi * n + gj] = make_float2(re, 0.f);
                    const float d = re - h[b * D + gi];
                    acc2 += d * d;
                } else {
                    Gb[(int64_t)gi * n + gj] = make_float2(re, im);
                    if (!lower_only) Gb[(int64_t)gj * n + gi] = make_float2(re, -im);
                    acc2 += 2.f * (re * re + im * im);
                }
            }
        }
    }

    // ---- arrow row (perm row 0 = original row D): G'[0][j] = sum_c w0_c f_c conj(V[j][c])
    for (int rho = tid; rho < 2 * D; rho += RB_THREADS) {
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < n; ++c) a = fmaf(w0f[c], VT[(int64_t)c * pitch + rho], a);
        rowb[rho] = a;
    }
    __syncthreads();
    for (int o = tid; o < D; o += RB_THREADS) {
        const float gr = rowb[o], gim = -rowb[D + o];     // G[D][o]
        Gb[(int64_t)D * n + o] = make_float2(gr, gim);
        if (!lower_only) Gb[(int64_t)o * n + D] = make_float2(gr, -gim);
        const float2 p = phi[b * D + o];                  // C[D][o] = conj(phi_o)
        const float dr = gr - p.x, di = gim + p.y;
        acc2 += 2.f * (dr * dr + di * di);
    }
    if (wave == 0) {   // corner: G'[0][0] = sum_c f_c w0_c^2
        float g00 = 0.f;
        for (int c = lane; c < n; c += 64) g00 = fmaf(w0f[c], z0s[c], g00);
        g00 = wave_sum(g00);
        if (lane == 0) {
            Gb[(int64_t)D * n + D] = make_float2(g00, 0.f);
            const float d = g00 - lw[S_CORNER_Z];
            acc2 += d * d;
        }
    }
    acc2 = wave_sum(acc2);
    if (lane == 0) redb[wave] = acc2;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < RB_THREADS / 64; ++i) s += redb[i];
        rn[b] = sqrtf(s);
    }
}

// Generic eigh output: V[b][row][col] row-major complex from VT / w0; used by admmnet_eigh_c64.
__global__ void vout_kernel(int n, const float *__restrict__ QV, const float *__restrict__ w0v,
                            float2 *__restrict__ V) {
    const int D = n - 1;
    const int64_t b = blockIdx.x;
    const float *VT = QV + b * ((int64_t)n * 2 * D);
    float2 *Vb = V + b * (int64_t)n * n;
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        const int c = idx / n, rr = idx - c * n;   // consecutive threads -> consecutive rows (coalesced reads)
        float2 v;
        if (rr == 0) v = make_float2(w0v[b * n + c], 0.f);
        else v = make_float2(VT[(int64_t)c * 2 * D + rr - 1], VT[(int64_t)c * 2 * D + D + rr - 1]);
        Vb[(int64_t)rr * n + c] = v;
    }
}

// Eigenvalues (and, for the generic eigh entry, eigenvectors) of the layer matrix out of a PADDED image (api.hip: A' =
// diag(A, 0) in the D = 256 pipeline): an eigenvector of the padding is a unit vector there, i.e. all of its weight sits in
// the rows Da .. Dimg - 1 of the image; the others have none.  One workgroup per matrix finds them, numbers the rest in
// ascending order (the order they come in) and writes w [na] and, if asked, V [na][na] row-major.
__global__ void unpad_kernel(int Da, int Dimg, const float *__restrict__ VTg, const float *__restrict__ wv,
                             const float *__restrict__ w0v, float *__restrict__ w_out, float2 *__restrict__ V) {
    __shared__ int slot[260];
    const int na = Da + 1, n = Dimg + 1;
    const int64_t b = blockIdx.x;
    const float *VT = VTg + b * ((int64_t)n * 2 * Dimg);
    for (int c = threadIdx.x; c < n; c += blockDim.x) {
        float s = 0.f;
        for (int r = Da; r < Dimg; ++r) {
            const float x = VT[(int64_t)c * 2 * Dimg + r], y = VT[(int64_t)c * 2 * Dimg + Dimg + r];
            s = fmaf(x, x, fmaf(y, y, s));
        }
        slot[c] = s > 0.5f ? -1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int k = 0;
        for (int c = 0; c < n; ++c)
            if (slot[c] == 0) slot[c] = (k < na) ? k++ : -1;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += blockDim.x)
        if (slot[c] >= 0 && w_out) w_out[b * na + slot[c]] = wv[b * n + c];
    if (!V) return;
    float2 *Vb = V + b * (int64_t)na * na;
    for (int idx = threadIdx.x; idx < n * na; idx += blockDim.x) {
        const int c = idx / na, rr = idx - c * na;   // consecutive threads -> consecutive rows (coalesced reads)
        const int k = slot[c];
        if (k < 0) continue;
        float2 v;
        if (rr == 0) v = make_float2(w0v[b * n + c], 0.f);
        else v = make_float2(VT[(int64_t)c * 2 * Dimg + rr - 1], VT[(int64_t)c * 2 * Dimg + Dimg + rr - 1]);
        Vb[(int64_t)rr * na + k] = v;
    }
}

int launch_rebuild(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *G,
                   float *rn, float *w_out, const Ws &ws, hipStream_t st, bool lower_only, int image_dim) {
    ProfScope _prof(KC_REBUILD, st);
    if (nb <= 0) return ADMMNET_OK;
    const int n = D + 1;
    if (image_dim <= 0) image_dim = D;
    // image of dimension 256: every tile resident, V^T read once (rebuild_big.hip); ADMMNET_REBUILD=tiles keeps the kernel below
    static const bool tiles = getenv("ADMMNET_REBUILD") && !strcmp(getenv("ADMMNET_REBUILD"), "tiles");
    if ((!tiles || image_dim != D) && rebuild_big_supported(image_dim)) {
        int rc = launch_rebuild_big(D, nb, lw, phi, h, G, rn, ws, st, lower_only);
        if (rc) return rc;
        if (w_out && image_dim == D) ADMM_HIP(hipMemcpyAsync(w_out, ws.w, sizeof(float) * nb * n, hipMemcpyDeviceToDevice, st));
        if (w_out && image_dim != D) {
            hipLaunchKernelGGL(unpad_kernel, dim3((unsigned)nb), dim3(256), 0, st, D, image_dim, ws.VT, ws.w, ws.w0, w_out,
                               (float2 *)nullptr);
            ADMM_HIP(hipGetLastError());
        }
        return ADMMNET_OK;
    }
    if (image_dim != D) {
        set_error("rebuild: no kernel for a %d-image holding D=%d", image_dim, D);
        return ADMMNET_E_ARG;
    }
    if (ws.skip) {
        set_error("rebuild: the per-tile kernel has no per-matrix filter (ADMMNET_SPECTRAL=1 with ADMMNET_REBUILD=tiles)");
        return ADMMNET_E_ARG;
    }
    const size_t lds = sizeof(float) * (3 * ((n + 4) & ~3) + 2 * D + 8);
    hipLaunchKernelGGL(rebuild_kernel, dim3((unsigned)nb), dim3(RB_THREADS), lds, st, D, lw, ws.VT, ws.w,
                       ws.w0, phi, h, G, rn, lower_only ? 1 : 0);
    ADMM_HIP(hipGetLastError());
    if (w_out) ADMM_HIP(hipMemcpyAsync(w_out, ws.w, sizeof(float) * nb * n, hipMemcpyDeviceToDevice, st));
    return ADMMNET_OK;
}

int launch_vout(int n, int64_t nb, float2 *V, float *w, const Ws &ws, hipStream_t st) {
    if (nb <= 0) return ADMMNET_OK;
    if (eig_dim(n - 1) != n - 1) {   // padded route: drop the eigenpairs of the padding
        hipLaunchKernelGGL(unpad_kernel, dim3((unsigned)nb), dim3(256), 0, st, n - 1, eig_dim(n - 1), ws.VT, ws.w, ws.w0, w, V);
        ADMM_HIP(hipGetLastError());
        return ADMMNET_OK;
    }
    hipLaunchKernelGGL(vout_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, ws.VT, ws.w0, V);
    ADMM_HIP(hipGetLastError());
    if (w) ADMM_HIP(hipMemcpyAsync(w, ws.w, sizeof(float) * nb * n, hipMemcpyDeviceToDevice, st));
    return ADMMNET_OK;
}

}  // namespace admmnet
