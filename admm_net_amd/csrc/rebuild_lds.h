// rebuild_lds.h -- G = V f(Lambda) V^H, arrow row, corner and ||G - C||_F with V resident in LDS.
// Shared tail of backrebuild.hip (V from the back-transform) and arrow.hip (V from the arrowhead
// solver); same arithmetic as rebuild.hip (/root/reference/admm_net.py:336-354, 400-403, 454).
//
// LDS inputs (all written before the call, followed by a barrier):
//   VTl[c][rho'] (pitch VP): eigenvector c, rho' = row index in the real plane | Dp + row index in
//                the imaginary plane, planes padded to Dp = 32 ceil(D / 32) with zeros
//   fs[c] = f(lambda_c) with fs[n] = 0, w0f[c] = w0_c f_c, z0s[c] = w0_c (w0 = arrow-row entries of V)
#pragma once
#include "common.h"

namespace admmnet {

struct BrGeom {
    int D, n, NT, Dp, BP, VP;
    __host__ __device__ explicit BrGeom(int D_) : D(D_), n(D_ + 1), NT((D_ + 31) / 32), Dp(32 * ((D_ + 31) / 32)) {
        BP = 2 * Dp;       // slab B row pitch: real plane | imaginary plane, each padded to 32
        VP = 2 * Dp + 4;   // VT row pitch in LDS
    }
    __host__ __device__ size_t vt_floats() const { return (size_t)n * VP; }
    // fs, w0f, z0s, rowb, redb
    __host__ __device__ size_t small_floats() const { return (size_t)3 * ((n + 4) & ~3) + 2 * Dp + 8; }
};

// learned eigenvalue map  f(lambda) = softplus(lambda - sigmoid(thr)) * value_net(|lambda|)
// (/root/reference/admm_net.py:310-334); vn: w1[16] b1[16] w2[16] b2[1], thr already sigmoid-ed
__device__ __forceinline__ float br_eig_map(float w, float thr, const float *vn) {
    const float base = softplus_f(w - thr);
    const float a = fabsf(w);
    float acc = vn[48];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = fmaf(vn[32 + j], fmaxf(fmaf(vn[j], a, vn[16 + j]), 0.f), acc);
    return base * sigmoid_f(acc);
}

template <class Mark>
__device__ __forceinline__ void rebuild_from_lds(const BrGeom &g, int64_t b, const float *__restrict__ lw,
                                                 const float *VTl, const float *fs, const float *w0f,
                                                 const float *z0s, float *rowb, float *redb,
                                                 const float2 *__restrict__ phi, const float *__restrict__ h,
                                                 float2 *__restrict__ G, float *__restrict__ rn, Mark mark,
                                                 int lower_only = 0) {
    using f32x16 = __attribute__((ext_vector_type(16))) float;
    constexpr int BR_THREADS = 256;
    const int D = g.D, n = g.n, NT = g.NT, Dp = g.Dp, VP = g.VP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, kh = lane >> 5;
    // ---------------- phase C: G = V f V^H, lower-triangle tiles, operands from LDS -------------------
    float2 *Gb = G + b * (int64_t)n * n;
    float acc2 = 0.f;
    const int ntiles = NT * (NT + 1) / 2;
    for (int t = wave; t < ntiles; t += BR_THREADS / 64) {
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        const int J = t - I * (I + 1) / 2;
        const int i0 = 32 * I, j0 = 32 * J;
        f32x16 aRe = {0}, aIm = {0};
        // K-steps in groups of 4, software pipelined: the 20 LDS reads of group g + 1 are issued before
        // the 16 MFMAs of group g.  No predicates: f(lambda) is zero-padded (fs[n] = 0) and the row index
        // is clamped, so the steps beyond n contribute exact zeros.
        constexpr int U = 4;
        const int nks = (n + 1) / 2;
        float fx[U], xr[U], xi[U], yr[U], yi[U];
        // u in [U0, U1) of one group of 4 K-steps
        auto lds_part = [&](int ks0, int U0, int U1, float (&f_)[U], float (&xr_)[U], float (&xi_)[U],
                            float (&yr_)[U], float (&yi_)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < U0 || u >= U1) continue;
                const int c = min(2 * (ks0 + u) + kh, n);
                const float *row = VTl + min(c, n - 1) * VP + l32;
                f_[u] = fs[c];
                xr_[u] = row[i0];
                xi_[u] = row[Dp + i0];
                yr_[u] = row[j0];
                yi_[u] = row[Dp + j0];
            }
        };
        auto mfma_part = [&](int U0, int U1, const float (&f_)[U], const float (&xr_)[U], const float (&xi_)[U],
                             const float (&yr_)[U], const float (&yi_)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < U0 || u >= U1) continue;
                const float ar = xr_[u] * f_[u], ai = xi_[u] * f_[u];
                aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(ar, yr_[u], aRe, 0, 0, 0);
                aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, yr_[u], aIm, 0, 0, 0);
                aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(ai, yi_[u], aRe, 0, 0, 0);
                aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(-ar, yi_[u], aIm, 0, 0, 0);
            }
        };
        // Two register sets in ping-pong, no copies.  The reads of the NEXT group are issued ahead of the MFMAs
        // of the current one, ten at a time: lgkmcnt is a 4-bit counter, so with more than 15 younger reads in
        // flight "wait for the older ones" is not expressible and the compiler falls back to lgkmcnt(0) -- a
        // full LDS round trip in front of every group.  sched_barrier pins the order.
        float fb[U], xrb[U], xib[U], yrb[U], yib[U];
        const int ngrp = (nks + U - 1) / U;
        lds_part(0, 0, U, fx, xr, xi, yr, yi);
        int gq = 0;
        for (; gq + 1 < ngrp; gq += 2) {
            lds_part((gq + 1) * U, 0, 2, fb, xrb, xib, yrb, yib);
            __builtin_amdgcn_sched_barrier(0);
            mfma_part(0, 2, fx, xr, xi, yr, yi);
            __builtin_amdgcn_sched_barrier(0);
            lds_part((gq + 1) * U, 2, 4, fb, xrb, xib, yrb, yib);
            __builtin_amdgcn_sched_barrier(0);
            mfma_part(2, 4, fx, xr, xi, yr, yi);
            __builtin_amdgcn_sched_barrier(0);
            lds_part((gq + 2) * U, 0, 2, fx, xr, xi, yr, yi);   // (past the end: clamped, exact zeros, unused)
            __builtin_amdgcn_sched_barrier(0);
            mfma_part(0, 2, fb, xrb, xib, yrb, yib);
            __builtin_amdgcn_sched_barrier(0);
            lds_part((gq + 2) * U, 2, 4, fx, xr, xi, yr, yi);
            __builtin_amdgcn_sched_barrier(0);
            mfma_part(2, 4, fb, xrb, xib, yrb, yib);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (gq < ngrp) mfma_part(0, U, fx, xr, xi, yr, yi);
        // epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = i0 + (q & 3) + 8 * (q >> 2) + 4 * kh;
            const int gj = j0 + l32;
            if (gi < D && gj < D && gi >= gj) {
                const float re = aRe[q], im = aIm[q];
                if (gi == gj) {
                    Gb[(int64_t)gi * n + gj] = make_float2(re, 0.f);
                    const float d = re - h[b * D + gi];
                    acc2 += d * d;
                } else {
                    Gb[(int64_t)gi * n + gj] = make_float2(re, im);
                    if (!lower_only) Gb[(int64_t)gj * n + gi] = make_float2(re, -im);   // (state kept as lower triangle)
                    acc2 += 2.f * (re * re + im * im);
                }
            }
        }
    }

    mark(4);
    // ---------------- arrow row (perm row 0 = original row D): G'[0][j] = sum_c w0_c f_c conj(V[j][c])
    //   When the tiles do not divide evenly over the four waves (10 tiles at NT = 4: 3, 3, 2, 2) the waves with
    //   one tile fewer take this row, while the others are still in their last tile.
    {
        const int heavy = ntiles & 3;                                   // waves [0, heavy) had one tile more
        const int first = heavy ? heavy : 0, nl = BR_THREADS / 64 - first;
        if (wave >= first) {
            for (int rp = tid - 64 * first; rp < 2 * Dp; rp += 64 * nl) {
                float a = 0.f;
#pragma unroll 8
                for (int c = 0; c < n; ++c) a = fmaf(w0f[c], VTl[c * VP + rp], a);
                rowb[rp] = a;
            }
        }
    }
    __syncthreads();
    for (int o = tid; o < D; o += BR_THREADS) {
        const float gr = rowb[o], gim = -rowb[Dp + o];     // G[D][o]
        Gb[(int64_t)D * n + o] = make_float2(gr, gim);
        if (!lower_only) Gb[(int64_t)o * n + D] = make_float2(gr, -gim);
        const float2 p = phi[b * D + o];                   // C[D][o] = conj(phi_o)
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
    mark(5);
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < BR_THREADS / 64; ++i) s += redb[i];
        rn[b] = sqrtf(s);
    }
}

}  // namespace admmnet
