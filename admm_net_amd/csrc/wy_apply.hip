// wy_apply.hip -- K3'' for D = 256: back-transform of the divide & conquer eigenvectors WITHOUT forming Q,
//   V = diag(1, Q') W,   Q' = H_0 H_1 ... H_{D-1},   H_u = I - tau_u v_u v_u^H
// (second half of torch.linalg.eigh at /root/reference/admm_net.py:303), by the block reflectors of the panels the
// tridiagonalisation worked in (tridiag_panel.hip; LAPACK cunmtr / clarfb, forward columnwise):
//   X <- (I - Y T Y^H) X,   X = W[1:, :],  panels from the last to the first,
// every product on the matrix cores (v_mfma_f32_16x16x4_f32).  Replaces ungtr_big_kernel (explicit Q, 16/3 n^3 flops on
// the vector ALUs) + vgemm_big_kernel (Q W).
//
// One wave owns 16 columns of X for all 256 rows: sixteen 16 x 16 accumulator tiles (128 VGPRs), and keeps them in
// registers through all 17 panels.  Per panel and wave:
//   Z  = Y^H X    the X tiles ARE the B operands (accumulator layout = B-operand layout: lane -> column, the four
//                 lane groups -> k), the A operands conj(Y) come from the LDS copy of the panel
//   Zt = T Z      Z, just produced as an accumulator tile, is the B operand again
//   X -= Y Zt     Zt as B operand, Y rows as A operands, accumulated straight into the X tiles
// so no X / Z / Zt element ever moves between lanes or through memory.  A workgroup = 4 waves = 64 columns shares the
// panel (Y: 256 x 16 complex = 36 KB with padding, T: 2 KB) through LDS; 4 workgroups per matrix cover columns 0 .. 255.
// The 257th eigenvector would cost a fifth workgroup with one live wave and 1 / 16 live columns in it (measured: 18 %
// of the kernel); wy_lastcol_kernel applies the 256 reflectors to that one vector directly, one wave per matrix.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "lane_reduce.h"

namespace admmnet {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int WY_D = 256;
constexpr int WY_THREADS = 256;
constexpr int WY_PITCH = 18;    // float2 per LDS row of the panel (as tridiag_panel.hip)

struct WyShared {
    float2 Y[WY_D][WY_PITCH];
    float2 T[16][WY_PITCH];
};

__global__ __launch_bounds__(WY_THREADS, 3) void wy_apply_kernel(const float2 *__restrict__ Mbuf,
                                                                 const float2 *__restrict__ Tfac,
                                                                 const float *__restrict__ Wbuf,
                                                                 const int2 *__restrict__ Wmap, float *__restrict__ VT,
                                                                 int nb, const int *__restrict__ skip) {
    __shared__ WyShared sh;
    constexpr int D = WY_D, n = D + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g = lane >> 4;
    // The four column slabs of a matrix read the same reflectors: workgroups are dealt to the 8 XCDs round-robin by
    // their linear id, so the slabs are made consecutive WITHIN an XCD (same L2, about the same time) -- dealt naively
    // they sit on four XCDs and each fetches the image from HBM for itself (measured 1.6 MB per matrix instead of 0.6).
    const int xcd = blockIdx.x & 7, iq = blockIdx.x >> 3;
    const int64_t bm = (int64_t)(iq >> 2) * 8 + xcd;
    if (bm >= nb) return;                          // (uniform; the grid is padded to a multiple of 8 matrices)
    if (skip && skip[bm] == 0) return;             // (uniform) this matrix' G is already there: spectral.hip
    const int cb = 4 * (iq & 3) + wave;            // column block of this wave: eigenvectors 16 cb .. 16 cb + 15
    const int col = 16 * cb + c16;
    const float2 *Mg = Mbuf + bm * ((int64_t)D * D + D + 1);
    const float2 *Tg = Tfac + bm * 17 * 256;
    // eigenvector `col` of T through the column map of the divide & conquer's top-level merge (dc.hip): a float offset
    // into the matrix' buffer block and the rows that exist there (the others are zero)
    int2 wm = Wmap[bm * n + col];
    wm.x = min(max(wm.x, 0), 3 * n * n - n);                   // (whatever the map holds, the reads stay inside the block)
    const float *WT = Wbuf + bm * (int64_t)3 * n * n + wm.x;   // WT[i] = W[i][col]
    const int rlo = wm.y & 0xffff, rhi = min(wm.y >> 16, n);
    float *Vb = VT + bm * ((int64_t)n * 2 * D);

    // X = W[1:, 16 cb .. 16 cb + 15]: tile I, register q of lane (c16, g) = X[16 I + 4 g + q][col]  (real to begin with)
    f32x4 xr[16], xi[16];
#pragma unroll
    for (int I = 0; I < 16; ++I) {
        const int r0 = 1 + 16 * I + 4 * g;
        const float *src = WT + r0;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 >= rlo && r0 + 3 < rhi) {
            v = make_float4(src[0], src[1], src[2], src[3]);
        } else {
            v.x = (r0 >= rlo && r0 < rhi) ? src[0] : 0.f;
            v.y = (r0 + 1 >= rlo && r0 + 1 < rhi) ? src[1] : 0.f;
            v.z = (r0 + 2 >= rlo && r0 + 2 < rhi) ? src[2] : 0.f;
            v.w = (r0 + 3 >= rlo && r0 + 3 < rhi) ? src[3] : 0.f;
        }
        xr[I] = f32x4{v.x, v.y, v.z, v.w};
        xi[I] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // the panel image travels global -> registers -> LDS at the head of each panel.  (Issuing the NEXT panel's loads
    // before this panel's products hid their latency at two waves per SIMD, 7.5 -> 6.7 ms; the 34 registers of that
    // prefetch are what stood between the kernel and THREE waves per SIMD, which hides it as well and keeps the matrix
    // cores busier: 5.13 -> 5.04 ms.)
    // (in two halves of eight reflectors: 17 staging registers instead of 34 -- at three waves per SIMD the budget is
    //  168 and the whole panel in flight spilled 14 registers to scratch, +0.37 MB of HBM traffic per matrix)
    float2 ypre[8], tpre;
    auto gload = [&](int pp, int h) {
        const int u0 = 16 * (pp - 1) + 1, I0 = (u0 < 0 ? 0 : u0) >> 4;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int uu = u0 + 8 * h + jj;
            ypre[jj] = (uu >= 0 && uu < D && tid >= 16 * I0) ? Mg[(int64_t)uu * D + tid] : make_float2(0.f, 0.f);
        }
        if (h == 0) tpre = Tg[pp * 256 + tid];
    };
    for (int pp = 16; pp >= 0; --pp) {
        const int u0 = 16 * (pp - 1) + 1;          // reflector of slot jj: u0 + jj (absent outside 0 .. D - 1)
        const int I0 = (u0 < 0 ? 0 : u0) >> 4;     // first block row the panel touches
        gload(pp, 0);
        __syncthreads();                           // the previous panel's LDS image is no longer read
        // ---- the panel: Y[r][jj] = v_{u0 + jj}[r] (reflector row u of the image), T
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) sh.Y[tid][jj] = ypre[jj];
        sh.T[tid >> 4][tid & 15] = tpre;
        gload(pp, 1);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) sh.Y[tid][8 + jj] = ypre[jj];
        __syncthreads();
        // ---- Z = Y^H X  (16 reflectors x 16 columns):  Zr = Yr Xr + Yi Xi,  Zi = Yr Xi - Yi Xr
        //      THREE real products per complex one (the "3M" form of cgemm3m) -- the kernel is bound by the matrix cores:
        //          T1 = Yr Xr,  T2 = Yi Xi,  T3 = (Yr + Yi)(Xi - Xr)  ->  Zr = T1 + T2,  Zi = T3 + T1 - T2
        //      (normwise the same error bound as the four-product form: |error| <= c eps |Y| |X|.)
        //      (Splitting this 256-term accumulation into block-local sums added pairwise was tried for accuracy: the
        //      distance of V to the float64 back-transform of the same reflectors moved from 9.3e-7 to 8.7e-7 only, for
        //      +28 % time -- the chain length is not what separates this kernel from the explicit-Q pair's 3.8e-7.)
        f32x4 t1 = f32x4{0.f, 0.f, 0.f, 0.f}, t2 = t1, t3 = t1;
#pragma unroll
        for (int I = 0; I < 16; ++I) {
            if (I >= I0) {   // (uniform)
                // step q: B = X register q (k = lane group g <-> row 16 I + 4 g + q), A[m = jj][k = g] = conj(Y[that row][jj])
                const float xrq[4] = {xr[I].x, xr[I].y, xr[I].z, xr[I].w}, xiq[4] = {xi[I].x, xi[I].y, xi[I].z, xi[I].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float2 y = sh.Y[16 * I + 4 * g + q][c16];
                    t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y.x, xrq[q], t1, 0, 0, 0);
                    t2 = __builtin_amdgcn_mfma_f32_16x16x4f32(y.y, xiq[q], t2, 0, 0, 0);
                    t3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y.x + y.y, xiq[q] - xrq[q], t3, 0, 0, 0);
                }
            }
        }
        const f32x4 zr = t1 + t2, zi = t3 + t1 - t2;
        // ---- Zt = T Z:  step q: B = Z register q (k = g <-> reflector 4 g + q), A[m][k = g] = T[m][4 g + q]
        f32x4 tr = f32x4{0.f, 0.f, 0.f, 0.f}, ti = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float zrq[4] = {zr.x, zr.y, zr.z, zr.w}, ziq[4] = {zi.x, zi.y, zi.z, zi.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float2 t = sh.T[c16][4 * g + q];
                tr = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, zrq[q], tr, 0, 0, 0);
                ti = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, ziq[q], ti, 0, 0, 0);
                tr = __builtin_amdgcn_mfma_f32_16x16x4f32(-t.y, ziq[q], tr, 0, 0, 0);
                ti = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, zrq[q], ti, 0, 0, 0);
            }
        }
        // ---- X -= Y Zt:  step q: B = Zt register q (k = g <-> reflector 4 g + q), A[m = row][k = g] = Y[16 I + m][4 g + q]
        //      3M again:  P1 = Yr Zr,  P2 = Yi Zi,  P3 = (Yr + Yi)(Zr + Zi)  ->  Xr -= P1 - P2,  Xi -= P3 - P1 - P2
        {
            const float trq[4] = {tr.x, tr.y, tr.z, tr.w}, tiq[4] = {ti.x, ti.y, ti.z, ti.w};
#pragma unroll
            for (int I = 0; I < 16; ++I) {
                if (I >= I0) {   // (uniform)
                    f32x4 p1 = f32x4{0.f, 0.f, 0.f, 0.f}, p2 = p1, im = xi[I];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float2 y = sh.Y[16 * I + c16][4 * g + q];
                        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y.x, trq[q], p1, 0, 0, 0);
                        p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(y.y, tiq[q], p2, 0, 0, 0);
                        im = __builtin_amdgcn_mfma_f32_16x16x4f32(-(y.x + y.y), trq[q] + tiq[q], im, 0, 0, 0);
                    }
                    xr[I] = xr[I] - p1 + p2;
                    xi[I] = im + p1 + p2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- V^T image for the rebuild: VT[c][rho] = Re V[rho][c], VT[c][D + rho] = Im V[rho][c]  (pitch 2 D)
    {
        float *dst = Vb + (int64_t)col * 2 * D + 4 * g;
#pragma unroll
        for (int I = 0; I < 16; ++I) {
            *reinterpret_cast<float4 *>(dst + 16 * I) = make_float4(xr[I].x, xr[I].y, xr[I].z, xr[I].w);
            *reinterpret_cast<float4 *>(dst + D + 16 * I) = make_float4(xi[I].x, xi[I].y, xi[I].z, xi[I].w);
        }
    }
}

// Column 256 of V: x <- H_0 H_1 ... H_{D-1} x, reflector by reflector (H_u = I - tau_u v_u v_u^H, v_u = image row u, zero
// above its unit row; taus in the consumed arrow slot).  One wave per matrix, lane l holds rows l, l + 64, l + 128, l + 192
// (the layout of a coalesced row load); eight reflectors per batch, the next batch's loads in flight behind the current
// batch's dot -> wave sum -> update chains.
constexpr int WL_BATCH = 8;

__global__ __launch_bounds__(64) void wy_lastcol_kernel(const float2 *__restrict__ Mbuf, const float *__restrict__ Wbuf,
                                                        const int2 *__restrict__ Wmap, float *__restrict__ VT,
                                                        const int *__restrict__ skip) {
    constexpr int D = WY_D, n = D + 1;
    const int lane = threadIdx.x;
    const int64_t bm = blockIdx.x;
    if (skip && skip[bm] == 0) return;
    const float2 *Mg = Mbuf + bm * ((int64_t)D * D + D + 1);
    const float2 *taus = Mg + (int64_t)D * D;
    int2 wm = Wmap[bm * n + D];                                  // eigenvector 256 through the column map (see wy_apply_kernel)
    wm.x = min(max(wm.x, 0), 3 * n * n - n);
    const float *WT = Wbuf + bm * (int64_t)3 * n * n + wm.x;
    const int rlo = wm.y & 0xffff, rhi = min(wm.y >> 16, n);
    float *Vb = VT + bm * ((int64_t)n * 2 * D);
    v2f xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = 1 + lane + 64 * k;
        xv[k] = v2f{(r >= rlo && r < rhi) ? WT[r] : 0.f, 0.f};
    }
    float2 va[WL_BATCH][4], vb[WL_BATCH][4], ta[WL_BATCH], tb[WL_BATCH];
    auto gload = [&](float2(&buf)[WL_BATCH][4], float2(&tt)[WL_BATCH], int u0) {   // reflectors u0, u0 - 1, ...
#pragma unroll
        for (int i = 0; i < WL_BATCH; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k)   // (uniform test: v_u is zero above row u -- half of the image is never read)
                buf[i][k] = (64 * k + 63 >= u0 - i) ? Mg[(int64_t)(u0 - i) * D + lane + 64 * k] : make_float2(0.f, 0.f);
            tt[i] = taus[u0 - i];
        }
    };
    auto apply = [&](const float2(&buf)[WL_BATCH][4], const float2(&tt)[WL_BATCH]) {
#pragma unroll
        for (int i = 0; i < WL_BATCH; ++i) {
            v2f d0 = pk_cfma_conj(v2f{0.f, 0.f}, pk2(buf[i][0]), xv[0]), d1 = pk_cfma_conj(v2f{0.f, 0.f}, pk2(buf[i][1]), xv[1]);
            d0 = pk_cfma_conj(d0, pk2(buf[i][2]), xv[2]);
            d1 = pk_cfma_conj(d1, pk2(buf[i][3]), xv[3]);
            float dx = pn_row16_sum(d0.x + d1.x), dy = pn_row16_sum(d0.y + d1.y);
            pn_group_sum2(dx, dy);
            const float2 sc = cmul(tt[i], make_float2(dx, dy));   // tau (v^H x)
            const v2f nsc = v2f{-sc.x, -sc.y};
#pragma unroll
            for (int k = 0; k < 4; ++k) xv[k] = pk_cfma(xv[k], nsc, pk2(buf[i][k]));   // x -= tau (v^H x) v
        }
    };
    gload(va, ta, D - 1);
    for (int b = 0; b < D / WL_BATCH; b += 2) {
        gload(vb, tb, D - 1 - WL_BATCH * (b + 1));
        apply(va, ta);
        if (b + 2 < D / WL_BATCH) gload(va, ta, D - 1 - WL_BATCH * (b + 2));
        apply(vb, tb);
    }
    float *dst = Vb + (int64_t)D * 2 * D;   // row c = 256 of the V^T image
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        dst[lane + 64 * k] = xv[k].x;
        dst[D + lane + 64 * k] = xv[k].y;
    }
}

// The block-reflector back-transform needs the T factors the panel tridiagonalisation writes; ADMMNET_BACK=q keeps the
// explicit Q (ungtr_big_kernel) + vgemm_big_kernel pair for A/B runs.
bool use_wy_back(int D) {
    // (every switch that takes the tridiagonalisation or the tridiagonal solver off the panel / D&C route turns it off)
    static const bool off = (getenv("ADMMNET_BACK") && !strcmp(getenv("ADMMNET_BACK"), "q")) ||
                            (getenv("ADMMNET_TRIDIAG_BIG") && !strcmp(getenv("ADMMNET_TRIDIAG_BIG"), "sweep")) ||
                            (getenv("ADMMNET_TRIDIAG") && !strcmp(getenv("ADMMNET_TRIDIAG"), "lds"));
    return !off && use_dc() && tridiag_panel_supported(D);
}

int launch_wy_apply(int D, int64_t nb, const Ws &ws, hipStream_t st) {
    ProfScope _prof(KC_ROTAPPLY, st);
    if (nb <= 0) return ADMMNET_OK;
    if (D != WY_D || !ws.Tfac || !ws.Wdc || !ws.Wmap) {
        set_error("wy_apply: D=%d unsupported (256 with the panel tridiagonalisation only)", D);
        return ADMMNET_E_ARG;
    }
    hipLaunchKernelGGL(wy_apply_kernel, dim3((unsigned)(4 * ((nb + 7) & ~(int64_t)7))), dim3(WY_THREADS), 0, st, ws.Mbuf,
                       ws.Tfac, ws.Wdc, ws.Wmap, ws.VT, (int)nb, ws.skip);
    ADMM_HIP(hipGetLastError());
    hipLaunchKernelGGL(wy_lastcol_kernel, dim3((unsigned)nb), dim3(64), 0, st, ws.Mbuf, ws.Wdc, ws.Wmap, ws.VT, ws.skip);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
