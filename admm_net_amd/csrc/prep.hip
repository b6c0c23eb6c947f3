// prep.hip -- K0: front end of one unrolled layer, one workgroup per signal.
//
//   (lazy)  Z <- Z + alpha_b (G - C_prev)        ZLayer.forward      admm_net.py:400-412
//   phi = |b|^2/(1+rho|b|^2) (y/(b+eps) + rho g + zeta)  PhiLayer    admm_net.py:88-103
//   t = Re diag(G + Z/(rho+eps)); h = soft projection of t + 0.1 MLP(t)
//                                                HLayer.forward      admm_net.py:143-194
//   A = [[diag h, phi],[phi^H, 1/lambda^2]] - Z/(rho+eps)
//                                                GLayer._build_block_matrix  admm_net.py:262-290
// The Z update of the PREVIOUS layer is applied here because its step alpha_b
// depends on the batch mean of the residual norms (admm_net.py:459), which is
// only known after every signal's G has been rebuilt.  A is written in the
// "arrow first" order the tridiagonalisation wants (corner, arrow column, D x D
// block); the similarity permutation is undone when G is written back.
#include "common.h"

namespace admmnet {

constexpr int PR_THREADS = 256;

__device__ __forceinline__ float2 cdiv_smith(float2 x, float2 d) {
    // numpy / c10::complex division (Smith); y / (b + eps) at admm_net.py:101
    const float a = x.x, b = x.y, c = d.x, e = d.y;
    if (fabsf(c) >= fabsf(e)) {
        if (c == 0.f && e == 0.f) return make_float2(a / fabsf(c), b / fabsf(e));
        const float rat = e / c, scl = 1.0f / (c + e * rat);
        return make_float2((a + b * rat) * scl, (b - a * rat) * scl);
    }
    const float rat = c / e, scl = 1.0f / (c * rat + e);
    return make_float2((a * rat + b) * scl, (b * rat - a) * scl);
}

__device__ __forceinline__ float pr_block_reduce(float v, float *scr, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) scr[wave] = v;
    __syncthreads();
    float r = scr[0];
#pragma unroll
    for (int i = 1; i < PR_THREADS / 64; ++i) r = is_max ? fmaxf(r, scr[i]) : (r + scr[i]);
    return r;
}

// mode bits
constexpr int PM_FIRST = 1;      // layer 0: G = Z = 0, nothing is read
constexpr int PM_ZZERO = 2;      // layer 1: stored Z is still zero (never written)
constexpr int PM_PHI_ONLY = 4;   // last layer: only phi is needed (admm_net.py:757-764)
constexpr int PM_NO_MATRIX = 8;  // layer 0 on the arrowhead path (arrow.hip): phi and h only, A is never formed
constexpr int PM_HALF = 32;      // D = 256 (tridiag_panel.hip): G / Z streamed as lower triangles, the image written for the lower
                                 // 16-block triangle only (diagonal blocks in full) -- the tiles that kernel loads
constexpr int PM_NOIMG = 64;     // with PM_HALF: only the lazy Z update streams, no image -- the G-layer is evaluated as a matrix function
                                 // straight from Z (spectral_fused.hip); the matrices it rejects get their image from half_image_kernel
constexpr int PM_SMALL = 128;    // phi and h only: the lazy Z update is folded into the first sweep of the matrix-function kernel
constexpr int PM_LEAN = 16;      // G / Z kept as lower triangles, A built by the tridiagonalisation's own loader
                                 // (tridiag_reg.hip): only the lazy Z update streams here, 24 n^2 / 2 bytes per signal

// (TRI: the D = 256 triangle walk with its 16 loads in flight compiled in -- 98 VGPRs; the other instance keeps the 54
//  registers and 8 waves per SIMD the lean D <= 128 stream wants)
// Zero padding of an eig_dim x eig_dim image around its D x D matrix (full storage: both triangles), arrow included.
__device__ __forceinline__ void pad_image(float2 *Mg, int D, int Dimg, int tid, int nthreads) {
    if (Dimg <= D) return;
    const float2 zero2 = make_float2(0.f, 0.f);
    for (int t = tid; t < Dimg * Dimg; t += nthreads) {
        const int i = t / Dimg, j = t - i * Dimg;
        if (i >= D || j >= D) Mg[t] = zero2;
    }
    for (int j = D + tid; j < Dimg; j += nthreads) Mg[(int64_t)Dimg * Dimg + j] = zero2;
}

// One element (i >= j) of the half image of A = C_g - Z / rho from the state element zn = Z[i][j]: the lower 16-block triangle, diagonal
// blocks in full, the arrow column as the conjugate of the arrow row (what tridiag_panel_kernel loads).
__device__ __forceinline__ void half_image_store(float2 *Mg, int D, int Dimg, int i, int j, float2 zn, float corner_g,
                                                 float inv_rho_g, float2 phj, float hi) {
    if (i == D) {
        if (j == D) Mg[(int64_t)Dimg * Dimg + Dimg] = make_float2(corner_g - inv_rho_g * zn.x, -inv_rho_g * zn.y);
        else Mg[(int64_t)Dimg * Dimg + j] = make_float2(phj.x - inv_rho_g * zn.x, phj.y + inv_rho_g * zn.y);   // A[j][D] = conj(A[D][j])
    } else {
        const float2 a = make_float2((i == j ? hi : 0.f) - inv_rho_g * zn.x, -inv_rho_g * zn.y);
        Mg[(int64_t)i * Dimg + j] = a;
        if (i != j && (i >> 4) == (j >> 4)) Mg[(int64_t)j * Dimg + i] = make_float2(a.x, -a.y);
    }
}
// The zero padding of the half image on the padded route (Dimg > D): rows D .. Dimg - 1 of the lower 16-block triangle (whole
// diagonal blocks: the block that holds row D also gets its columns >= D), the arrow entries behind D.
__device__ __forceinline__ void half_image_pad(float2 *Mg, int D, int Dimg, int tid, int nthreads) {
    const float2 zero2 = make_float2(0.f, 0.f);
    const int pad = Dimg - D;
    for (int t = tid; t < pad * Dimg; t += nthreads) {
        const int i = D + t / Dimg, j = t - (i - D) * Dimg;
        if ((j >> 4) <= (i >> 4)) Mg[(int64_t)i * Dimg + j] = zero2;
    }
    const int i0 = D & ~15;                      // rows of the block that holds row D, above it
    for (int t = tid; t < (D - i0) * 16; t += nthreads) {
        const int i = i0 + (t >> 4), j = i0 + (t & 15);
        if (j >= D) Mg[(int64_t)i * Dimg + j] = zero2;
    }
    for (int j = D + tid; j < Dimg; j += nthreads) Mg[(int64_t)Dimg * Dimg + j] = zero2;
}

// The half image for the matrices the matrix-function route hands to the eigensolver (skip[s] != 0), from the already updated Z.
__global__ __launch_bounds__(PR_THREADS) void half_image_kernel(int D, int Dimg, const float *__restrict__ lw,
                                                                const float2 *__restrict__ phi, const float *__restrict__ h,
                                                                const float2 *__restrict__ Z, float2 *__restrict__ Mbuf,
                                                                const int *__restrict__ skip) {
    const int64_t s = blockIdx.x;
    if (skip && skip[s] == 0) return;
    const int n = D + 1, tid = threadIdx.x;
    const float corner_g = lw[S_CORNER_G], inv_rho_g = lw[S_INV_RHO_G];
    const float2 *Zs = Z + s * (int64_t)n * n;
    float2 *Mg = Mbuf + s * ((int64_t)Dimg * Dimg + Dimg + 1);
    const int ntri = n * (n + 1) / 2;
    for (int t = tid; t < ntri; t += PR_THREADS) {
        int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        if (i * (i + 1) / 2 > t) --i;
        if ((i + 1) * (i + 2) / 2 <= t) ++i;
        const int j = t - i * (i + 1) / 2;
        const float2 zn = Zs[(int64_t)i * n + j];
        half_image_store(Mg, D, Dimg, i, j, zn, corner_g, inv_rho_g, (i == D && j < D) ? phi[s * D + j] : make_float2(0.f, 0.f),
                         i < D ? h[s * D + i] : 0.f);
    }
    if (Dimg > D) half_image_pad(Mg, D, Dimg, tid, PR_THREADS);
}

template <bool TRI>
__global__ __launch_bounds__(PR_THREADS) void prep_kernel(
    int D, int mode, const float *__restrict__ lw, const float *__restrict__ lw_prev,
    const float2 *__restrict__ y, const float2 *__restrict__ bsym, const float *__restrict__ sigma,
    float2 *__restrict__ G, float2 *__restrict__ Z, const float2 *__restrict__ phi_prev,
    const float *__restrict__ h_prev, const float *__restrict__ alpha, float2 *__restrict__ phi_out,
    float *__restrict__ h_out, float2 *__restrict__ Mbuf, int Dimg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = D + 1;
    const int tid = threadIdx.x;
    const int64_t s = blockIdx.x;
    float2 *phis = reinterpret_cast<float2 *>(smem);   // [D] new phi
    float2 *phip = phis + D;                          // [D] previous phi
    float *ts = reinterpret_cast<float *>(phip + D);   // [D] t, then tc
    float *hs = ts + D;                                // [D] new h
    float *hp = hs + D;                                // [D] previous h
    float *hid = hp + D;                               // [64]
    float *scr = hid + kHid;                           // [8]
    const LayerLayout L{D};
    const bool first = mode & PM_FIRST, zzero = mode & PM_ZZERO, phi_only = mode & PM_PHI_ONLY;
    const float al = first ? 0.f : alpha[s];
    const float rho_phi = lw[S_RHO_PHI], rho_h_eps = lw[S_RHO_H_EPS];
    float2 *Gs = G + s * (int64_t)n * n;
    float2 *Zs = Z + s * (int64_t)n * n;

    for (int i = tid; i < D; i += PR_THREADS) {
        float2 g = make_float2(0.f, 0.f), zeta = make_float2(0.f, 0.f);
        float t = 0.f;
        if (!first) {
            const float2 pp = phi_prev[s * D + i];
            const float hpv = h_prev[s * D + i];
            phip[i] = pp;
            hp[i] = hpv;
            const float2 Gl = Gs[(int64_t)D * n + i];                    // G[D][i] = conj(G[i][D])
            const float2 Zl = zzero ? make_float2(0.f, 0.f) : Zs[(int64_t)D * n + i];
            // Z_new[D][i] = Z[D][i] + alpha (G[D][i] - conj(phi_prev_i))
            const float znr = Zl.x + al * (Gl.x - pp.x);
            const float zni = Zl.y + al * (Gl.y + pp.y);
            g = make_float2(Gl.x, -Gl.y);
            zeta = make_float2(znr, -zni);
            const float Gd = Gs[(int64_t)i * n + i].x;
            const float Zd = zzero ? 0.f : Zs[(int64_t)i * n + i].x;
            const float Znd = Zd + al * (Gd - hpv);
            t = Gd + Znd / rho_h_eps;
        }
        const float2 bv = bsym[s * D + i];
        const float ab = hypotf(bv.x, bv.y);
        const float b_sq = ab * ab + kEpsRef;
        const float wgt = b_sq / (1.0f + rho_phi * b_sq);
        const float2 yb = cdiv_smith(y[s * D + i], make_float2(bv.x + kEpsRef, bv.y));
        const float2 rt = make_float2(yb.x + rho_phi * g.x + zeta.x, yb.y + rho_phi * g.y + zeta.y);
        const float2 ph = make_float2(wgt * rt.x, wgt * rt.y);
        phis[i] = ph;
        phi_out[s * D + i] = ph;
        ts[i] = t;
    }
    if (phi_only) return;
    __syncthreads();

    // ---- H layer: correction MLP D -> 64 -> D, soft projection
    if (tid < kHid) {
        const float *w1t = lw + L.off_w1t();
        float a = lw[L.off_b1() + tid];
        for (int i = 0; i < D; ++i) a = fmaf(w1t[i * kHid + tid], ts[i], a);
        hid[tid] = fmaxf(a, 0.f);
    }
    __syncthreads();
    float lmax = 0.f, lsum = 0.f;
    for (int i = tid; i < D; i += PR_THREADS) {
        const float *w2t = lw + L.off_w2t();
        float a = lw[L.off_b2() + i];
#pragma unroll 8
        for (int j = 0; j < kHid; ++j) a = fmaf(w2t[j * D + i], hid[j], a);
        const float tc = ts[i] + 0.1f * tanhf(a);
        hs[i] = tc;
        lmax = fmaxf(lmax, fabsf(tc));
        lsum += tc;
    }
    const float linf = pr_block_reduce(lmax, scr, true);
    const float tr = pr_block_reduce(lsum, scr, false);
    const float sg = sigma[s];
    const float Acoef = lw[S_A_COEF] * sg + sg * sg;
    const float cval = Acoef * linf + tr;
    float scale = lw[S_SIG_PW] / (cval + kEpsRef);
    scale = fminf(scale, 1.0f);
    for (int i = tid; i < D; i += PR_THREADS) {
        const float hv = hs[i] * scale;
        hs[i] = hv;
        h_out[s * D + i] = hv;
    }
    __syncthreads();

    if (mode & (PM_NO_MATRIX | PM_SMALL)) return;
    if (mode & PM_LEAN) {
        // Z <- Z + alpha (G - C_prev) on the lower triangle (row D = arrow row).  Two rows per trip, one per
        // half of the workgroup: row r (r + 1 entries) and row n - 1 - r, so both halves stay busy.
        if (first) return;
        const float corner_zp = lw_prev[S_CORNER_Z];
        const int half = tid >> 7, t7 = tid & 127;
        for (int r = 0; 2 * r < n; ++r) {
            const int i = half ? (n - 1 - r) : r;
            if (half && i == r) break;   // middle row of an odd n: the first half takes it
            for (int j = t7; j <= i; j += PR_THREADS / 2) {
                const int64_t idx = (int64_t)i * n + j;
                const float2 gij = Gs[idx];
                const float2 zij = zzero ? make_float2(0.f, 0.f) : Zs[idx];
                float2 c;
                if (i < D) c = make_float2(i == j ? hp[i] : 0.f, 0.f);
                else if (j == D) c = make_float2(corner_zp, 0.f);
                else c = make_float2(phip[j].x, -phip[j].y);   // C[D][j] = conj(phi_prev_j)
                Zs[idx] = make_float2(zij.x + al * (gij.x - c.x), zij.y + al * (gij.y - c.y));
            }
        }
        return;
    }
    // ---- stream the matrix: finish the lazy Z update, build A (arrow-first order)
    const float corner_g = lw[S_CORNER_G], inv_rho_g = lw[S_INV_RHO_G];
    const float corner_zp = first ? 0.f : lw_prev[S_CORNER_Z];
    // (Dimg = eig_dim(D): the image is Dimg x Dimg with the matrix in its leading D x D block, the arrow and the corner
    //  behind it as always; rows / columns D .. Dimg - 1 are zero -- the padded route of api.hip)
    float2 *Mg = Mbuf + s * ((int64_t)Dimg * Dimg + Dimg + 1);
    if (TRI && (mode & PM_HALF)) {
        // Lower triangle only (G, Z and C are Hermitian; row D = the arrow row lies in it): half the G / Z streams,
        // about half the image.  Entries inside a diagonal 16-block also write their mirror, the arrow COLUMN of the
        // image is the conjugate of the arrow row.
        // The triangle is walked by its own linear index t = i (i + 1) / 2 + j (no skipped trips, no division: i from a
        // float square root, exact below 2^24, with a one-step correction), eight elements per trip with their sixteen loads
        // issued before the first use.
        constexpr int PM_EPT = 8;
        const int ntri = n * (n + 1) / 2;
        const float2 zero2 = make_float2(0.f, 0.f);
        for (int t0 = tid; t0 < ntri; t0 += PM_EPT * PR_THREADS) {
            int ii[PM_EPT], jj[PM_EPT];
            float2 gv[PM_EPT], zv[PM_EPT];
#pragma unroll
            for (int q = 0; q < PM_EPT; ++q) {
                const int t = t0 + q * PR_THREADS;
                int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
                if (i * (i + 1) / 2 > t) --i;
                if ((i + 1) * (i + 2) / 2 <= t) ++i;
                ii[q] = i;
                jj[q] = t - i * (i + 1) / 2;
                const bool on = t < ntri && !first;
                const int64_t idx = (int64_t)i * n + jj[q];
                gv[q] = on ? Gs[idx] : zero2;
                zv[q] = (on && !zzero) ? Zs[idx] : zero2;
            }
#pragma unroll
            for (int q = 0; q < PM_EPT; ++q) {
                if (t0 + q * PR_THREADS >= ntri) continue;
                const int i = ii[q], j = jj[q];
                const int64_t idx = (int64_t)i * n + j;
                float2 zn = zero2;
                if (!first) {
                    float2 c;
                    if (i < D) c = make_float2(i == j ? hp[i] : 0.f, 0.f);
                    else if (j == D) c = make_float2(corner_zp, 0.f);
                    else c = make_float2(phip[j].x, -phip[j].y);   // C[D][j] = conj(phi_prev_j)
                    zn = make_float2(zv[q].x + al * (gv[q].x - c.x), zv[q].y + al * (gv[q].y - c.y));
                    Zs[idx] = zn;
                }
                if (!(mode & PM_NOIMG))
                    half_image_store(Mg, D, Dimg, i, j, zn, corner_g, inv_rho_g, (i == D && j < D) ? phis[j] : zero2, i < D ? hs[i] : 0.f);
            }
        }
        if (Dimg > D && !(mode & PM_NOIMG)) half_image_pad(Mg, D, Dimg, tid, PR_THREADS);   // (uniform)
        return;
    }
    for (int idx = tid; idx < n * n; idx += PR_THREADS) {
        const int i = idx / n, j = idx - i * n;
        float2 zn = make_float2(0.f, 0.f);
        if (!first) {
            const float2 gij = Gs[idx];
            const float2 zij = zzero ? make_float2(0.f, 0.f) : Zs[idx];
            float2 c;
            if (i < D && j < D) c = make_float2(i == j ? hp[i] : 0.f, 0.f);
            else if (i == D && j == D) c = make_float2(corner_zp, 0.f);
            else if (j == D) c = phip[i];
            else c = make_float2(phip[j].x, -phip[j].y);
            zn = make_float2(zij.x + al * (gij.x - c.x), zij.y + al * (gij.y - c.y));
            Zs[idx] = zn;
        }
        if (i == D && j < D) continue;   // mirror of the arrow column
        float2 c;
        if (i < D && j < D) c = make_float2(i == j ? hs[i] : 0.f, 0.f);
        else if (i == D) c = make_float2(corner_g, 0.f);
        else c = phis[i];
        const float2 a = make_float2(c.x - inv_rho_g * zn.x, c.y - inv_rho_g * zn.y);
        if (i < D && j < D) Mg[(int64_t)i * Dimg + j] = a;
        else if (i < D) Mg[(int64_t)Dimg * Dimg + i] = a;
        else Mg[(int64_t)Dimg * Dimg + Dimg] = a;
    }
    pad_image(Mg, D, Dimg, tid, PR_THREADS);
}

// Generic: Hermitian A[b][n][n] (lower triangle read) -> arrow-first storage, no permutation.
__global__ void build_generic_kernel(int n, const float2 *__restrict__ A, float2 *__restrict__ Mbuf, int Dimg) {
    const int D = n - 1;
    const int64_t b = blockIdx.x;
    const float2 *Ab = A + b * (int64_t)n * n;
    float2 *Mg = Mbuf + b * ((int64_t)Dimg * Dimg + Dimg + 1);
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        const int i = idx / n, j = idx - i * n;
        float2 a;
        if (i >= j) {
            a = Ab[(int64_t)i * n + j];
            if (i == j) a.y = 0.f;
        } else {
            a = Ab[(int64_t)j * n + i];
            a.y = -a.y;
        }
        if (i >= 1 && j >= 1) Mg[(int64_t)(i - 1) * Dimg + (j - 1)] = a;
        else if (j == 0 && i >= 1) Mg[(int64_t)Dimg * Dimg + (i - 1)] = a;
        else if (i == 0 && j == 0) Mg[(int64_t)Dimg * Dimg + Dimg] = a;
    }
    pad_image(Mg, D, Dimg, threadIdx.x, blockDim.x);
}

// Unit-test / building-block entry: A = [[diag h, phi],[phi^H, corner]] - inv_rho Z in
// arrow-first storage (GLayer._build_block_matrix, admm_net.py:262-290).  Z may be null.
__global__ void build_block_kernel(int D, float corner, float inv_rho, const float2 *__restrict__ phi,
                                   const float *__restrict__ h, const float2 *__restrict__ Z,
                                   float2 *__restrict__ Mbuf, int Dimg) {
    const int n = D + 1;
    const int64_t s = blockIdx.x;
    float2 *Mg = Mbuf + s * ((int64_t)Dimg * Dimg + Dimg + 1);
    for (int idx = threadIdx.x; idx < n * n; idx += blockDim.x) {
        const int i = idx / n, j = idx - i * n;
        if (i == D && j < D) continue;
        const float2 zn = Z ? Z[s * (int64_t)n * n + idx] : make_float2(0.f, 0.f);
        float2 c;
        if (i < D && j < D) c = make_float2(i == j ? h[s * D + i] : 0.f, 0.f);
        else if (i == D) c = make_float2(corner, 0.f);
        else c = phi[s * D + i];
        const float2 a = make_float2(c.x - inv_rho * zn.x, c.y - inv_rho * zn.y);
        if (i < D && j < D) Mg[(int64_t)i * Dimg + j] = a;
        else if (i < D) Mg[(int64_t)Dimg * Dimg + i] = a;
        else Mg[(int64_t)Dimg * Dimg + Dimg] = a;
    }
    pad_image(Mg, D, Dimg, threadIdx.x, blockDim.x);
}

int launch_build_block(int D, int64_t nb, float corner, float inv_rho, const float2 *phi, const float *h,
                       const float2 *Z, const Ws &ws, hipStream_t st) {
    if (nb <= 0) return ADMMNET_OK;
    hipLaunchKernelGGL(build_block_kernel, dim3((unsigned)nb), dim3(256), 0, st, D, corner, inv_rho, phi, h, Z,
                       ws.Mbuf, eig_dim(D));
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_prep(const admmnet_cfg *cfg, const float *lw_all, int k, const float2 *y, const float2 *b,
                const float *sigma, int64_t b0, int64_t nb, const Ws &ws, bool phi_only, hipStream_t st,
                bool no_matrix, bool lean, bool no_image, bool small) {
    ProfScope _prof(KC_PREP, st);
    if (nb <= 0) return ADMMNET_OK;
    const int D = cfg->M * cfg->N, n = D + 1;
    const LayerLayout L{D};
    const float *lw = lw_all + (int64_t)k * L.size();
    const float *lwp = k > 0 ? lw_all + (int64_t)(k - 1) * L.size() : lw;
    int mode = 0;
    if (k == 0) mode |= PM_FIRST;
    if (k == 1) mode |= PM_ZZERO;
    if (phi_only) mode |= PM_PHI_ONLY;
    if (no_matrix && k == 0) mode |= PM_NO_MATRIX;
    if (lean) mode |= (D > 128) ? PM_HALF : PM_LEAN;
    if (no_image && (mode & PM_HALF)) mode |= PM_NOIMG;
    if (small) mode |= PM_SMALL;
    const int cur = k & 1, prv = cur ^ 1;
    const size_t lds = sizeof(float2) * 2 * D + sizeof(float) * (3 * D + kHid + 8);
    auto kern = (mode & PM_HALF) ? prep_kernel<true> : prep_kernel<false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(PR_THREADS), lds, st, D, mode, lw, lwp,
                       y + b0 * D, b + b0 * D, sigma + b0, ws.G + b0 * (int64_t)n * n,
                       ws.Z + b0 * (int64_t)n * n, ws.phi[prv] + b0 * D, ws.h[prv] + b0 * D, ws.alpha + b0,
                       ws.phi[cur] + b0 * D, ws.h[cur] + b0 * D, ws.Mbuf, eig_dim(D));
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_half_image(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, const float2 *Z, const Ws &ws,
                      hipStream_t st) {
    ProfScope _prof(KC_PREP, st);
    if (nb <= 0) return ADMMNET_OK;
    hipLaunchKernelGGL(half_image_kernel, dim3((unsigned)nb), dim3(PR_THREADS), 0, st, D, eig_dim(D), lw, phi, h, Z, ws.Mbuf, ws.skip);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_build_generic(int n, int64_t nb, const float2 *A, const Ws &ws, hipStream_t st) {
    if (nb <= 0) return ADMMNET_OK;
    hipLaunchKernelGGL(build_generic_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, A, ws.Mbuf, eig_dim(n - 1));
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
