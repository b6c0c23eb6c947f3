// rebuild_big.hip -- K4 for D = 256: learned eigenvalue map + G = V f(Lambda) V^H + ||G - C||_F with ALL 36 lower
// 32 x 32 tiles of the D x D block resident as matrix-core accumulators (rebuild.hip's formulas, see there:
// /root/reference/admm_net.py:310-354, :400-403, :454).
//
// rebuild_kernel (rebuild.hip) walks the tiles one after the other and streams both operands of every tile from the
// V^T image in global memory: at n = 257 that image (526 KB per matrix) does not stay in a CU's share of the L2, so it
// came from HBM ~7 times per matrix (measured FETCH_SIZE: 3.7 MB per matrix).  Here the eigenvector index is the
// OUTER loop: 8 waves hold 4-5 tiles each (<= 160 accumulator registers), a slab of 8 eigenvectors (8 rows of the
// image = 16 KB) is staged through LDS once and feeds every tile, so V^T is read exactly once and G written once.
// The arrow row (sum_c w0_c f_c conj(V[j][c])) rides along on the staged slabs.
#include "common.h"

namespace admmnet {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int RG_D = 256;
constexpr int RG_THREADS = 512;
constexpr int RG_KS = 8;                    // eigenvectors per slab
constexpr int RG_PITCH = 2 * RG_D + 32;     // floats per staged row: the two k-halves of a wave read rows 32 banks apart
constexpr int RG_NT = 8;                    // 32-blocks per dimension
constexpr int RG_TILES = RG_NT * (RG_NT + 1) / 2;   // 36

__device__ __forceinline__ float rg_eig_map(float w, float thr, const float *vn) {   // rebuild.hip: eig_map
    const float base = softplus_f(w - thr);
    const float a = fabsf(w);
    float acc = vn[48];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = fmaf(vn[32 + j], fmaxf(fmaf(vn[j], a, vn[16 + j]), 0.f), acc);
    return base * sigmoid_f(acc);
}

__global__ __launch_bounds__(RG_THREADS, 2) void rebuild_big_kernel(const float *__restrict__ lw,
                                                                    const float *__restrict__ VTg,
                                                                    const float *__restrict__ wv,
                                                                    const float *__restrict__ w0v,
                                                                    const float2 *__restrict__ phi,
                                                                    const float *__restrict__ h, float2 *__restrict__ G,
                                                                    float *__restrict__ rn, int lower_only, int Da,
                                                                    const int *__restrict__ skip) {
    __shared__ float slab[2][RG_KS][RG_PITCH];
    __shared__ float fs[264], w0f[264], z0s[264];
    __shared__ float redb[8];
    // D, n: the image (eigenvector index c < n, rows rho < D); Da, na: the layer matrix inside it (Da < D on the padded
    // route of api.hip: rows / columns Da .. D - 1 of G' = diag(G, f(0) I) are not stored, the padded eigenvectors have
    // exact zeros in the stored rows and a zero first component, so they drop out of every stored entry)
    constexpr int D = RG_D, n = D + 1;
    const int na = Da + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, kh = lane >> 5;
    const int64_t b = blockIdx.x;
    if (skip && skip[b] == 0) return;   // (uniform) this matrix' G is already there: spectral.hip
    const LayerLayout L{Da};
    const float thr = lw[S_THR];
    const float *vn = lw + L.off_vn();
    for (int c = tid; c < 264; c += RG_THREADS) {
        float f = 0.f, z0 = 0.f;
        if (c < n) {
            f = rg_eig_map(wv[b * n + c], thr, vn);
            z0 = w0v[b * n + c];
        }
        fs[c] = f;
        w0f[c] = z0 * f;
        z0s[c] = z0;
    }
    const float *VT = VTg + b * ((int64_t)n * 2 * D);
    float2 *Gb = G + b * (int64_t)na * na;

    // tiles of this wave: t = wave + 8 s, s = 0 .. 4 (t < 36), t -> (I, J) of the lower triangle
    int tI[5], tJ[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int t = wave + 8 * s;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        tI[s] = (t < RG_TILES) ? I : 0;
        tJ[s] = (t < RG_TILES) ? t - I * (I + 1) / 2 : 0;
    }
    const int nmine = (wave + 32 < RG_TILES) ? 5 : 4;   // (uniform) waves 0 .. 3 own five tiles
    f32x16 aRe[5], aIm[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) aRe[s] = aIm[s] = f32x16{0};
    float arow = 0.f;                                   // arrow row entry rho' = tid (re plane | im plane)

    // slab staging: 8 rows x 512 floats = 1024 float4, two per thread
    float4 stage[2];
    auto gload = [&](int c0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + q * RG_THREADS, row = idx >> 7, col4 = idx & 127;
            const int c = c0 + row;
            stage[q] = (c < n) ? *reinterpret_cast<const float4 *>(VT + (int64_t)c * 2 * D + 4 * col4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + q * RG_THREADS, row = idx >> 7, col4 = idx & 127;
            const int o = 4 * col4;
            *reinterpret_cast<float4 *>(&slab[buf][row][(o >= D) ? D + 32 + (o - D) : o]) = stage[q];
        }
    };
    constexpr int NSLAB = (n + RG_KS - 1) / RG_KS;      // 33
    gload(0);
    lstore(0);
    __syncthreads();
    for (int sl = 0; sl < NSLAB; ++sl) {
        const int buf = sl & 1, c0 = sl * RG_KS;
        if (sl + 1 < NSLAB) gload(c0 + RG_KS);
        // arrow row: every thread owns one column rho' of the image
        {
            const int o = (tid >= D) ? D + 32 + (tid - D) : tid;
#pragma unroll
            for (int k = 0; k < RG_KS; ++k) arow = fmaf(w0f[c0 + k], slab[buf][k][o], arow);   // (w0f = 0 beyond n)
        }
#pragma unroll
        for (int kk = 0; kk < RG_KS; kk += 2) {
            const int c = c0 + kk + kh;                  // this lane half's eigenvector of the k-step
            const float fc = fs[c];
            const float *row = &slab[buf][kk + kh][0];
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                if (s < nmine) {   // (uniform)
                    const int xo = 32 * tI[s] + r32, yo = 32 * tJ[s] + r32;
                    const float xr = row[xo] * fc, xi = row[D + 32 + xo] * fc;
                    const float yr = row[yo], yi = row[D + 32 + yo];
                    aRe[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, yr, aRe[s], 0, 0, 0);
                    aIm[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, yr, aIm[s], 0, 0, 0);
                    aRe[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, yi, aRe[s], 0, 0, 0);
                    aIm[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(-xr, yi, aIm[s], 0, 0, 0);
                }
            }
        }
        if (sl + 1 < NSLAB) lstore(buf ^ 1);
        __syncthreads();
    }
    // ---- epilogue: C/D layout col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5)
    float acc2 = 0.f;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        if (s < nmine) {
            const int i0 = 32 * tI[s], j0 = 32 * tJ[s];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int gi = i0 + (q & 3) + 8 * (q >> 2) + 4 * kh, gj = j0 + r32;
                if (gi >= gj && gi < Da) {
                    const float re = aRe[s][q], im = aIm[s][q];
                    if (gi == gj) {
                        Gb[(int64_t)gi * na + gj] = make_float2(re, 0.f);
                        const float d = re - h[b * Da + gi];
                        acc2 += d * d;
                    } else {
                        Gb[(int64_t)gi * na + gj] = make_float2(re, im);
                        if (!lower_only) Gb[(int64_t)gj * na + gi] = make_float2(re, -im);
                        acc2 += 2.f * (re * re + im * im);
                    }
                }
            }
        }
    }
    // arrow row G[D][o] = (arow[o], -arow[D + o]): exchange the two planes through LDS
    float *rowb = &slab[0][0][0];
    rowb[tid] = arow;
    __syncthreads();
    if (tid < Da) {
        const int o = tid;
        const float gr = rowb[o], gim = -rowb[D + o];
        Gb[(int64_t)Da * na + o] = make_float2(gr, gim);
        if (!lower_only) Gb[(int64_t)o * na + Da] = make_float2(gr, -gim);
        const float2 p = phi[b * Da + o];                 // C[D][o] = conj(phi_o)
        const float dr = gr - p.x, di = gim + p.y;
        acc2 += 2.f * (dr * dr + di * di);
    }
    if (wave == 7) {   // corner: G[D][D] = sum_c f_c w0_c^2
        float g00 = 0.f;
        for (int c = lane; c < n; c += 64) g00 = fmaf(w0f[c], z0s[c], g00);
        g00 = wave_sum(g00);
        if (lane == 0) {
            Gb[(int64_t)Da * na + Da] = make_float2(g00, 0.f);
            const float d = g00 - lw[S_CORNER_Z];
            acc2 += d * d;
        }
    }
    acc2 = wave_sum(acc2);
    if (lane == 0) redb[wave] = acc2;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += redb[i];
        rn[b] = sqrtf(s);
    }
}

bool rebuild_big_supported(int image_dim) { return image_dim == RG_D; }

int launch_rebuild_big(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *G, float *rn,
                       const Ws &ws, hipStream_t st, bool lower_only) {
    if (D < 1 || D > RG_D) {
        set_error("rebuild_big: D=%d does not fit the 256 image", D);
        return ADMMNET_E_ARG;
    }
    hipLaunchKernelGGL(rebuild_big_kernel, dim3((unsigned)nb), dim3(RG_THREADS), 0, st, lw, ws.VT, ws.w, ws.w0, phi, h, G,
                       rn, lower_only ? 1 : 0, D, ws.skip);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
