// backrebuild.hip -- K3' + K4 fused for D <= 128: back-transform V = Q W and rebuild
// G = V f(Lambda) V^H + ||G - C||_F in ONE kernel, with V never leaving the chip.
//
//   reference: the second half of torch.linalg.eigh (V) and GLayer's eigenvalue map + reconstruction,
//   /root/reference/admm_net.py:303, 310-354, and ZLayer's residual norm :400-403,454.
//
// One 256-thread workgroup per matrix, one workgroup per CU (it owns the LDS and the register file):
//   A. VT[c][rho] = sum_r W[1 + r][c] QT[r][rho] on the f32 matrix cores.  K (= r) is streamed in slabs
//      of 16 rows through LDS (register-staged double buffering: the next slab's global loads fly
//      during the MFMAs of the current one).  Wave w owns the columns rho of row-tile w of V (real
//      and imaginary planes) for all eigenvector tiles: 5 x 2 accumulator tiles = 160 registers, and
//      every k-step feeds 10 MFMAs from 7 LDS reads.
//   B. The accumulators are written to LDS as VT[c][rho] (134 KB, over the dead slab buffers).
//   C. G's lower-triangle 32 x 32 tiles (4M complex product, same arithmetic as rebuild.hip) with
//      all four operand streams read from LDS; arrow row, corner and the residual norm as in
//      rebuild.hip.
// Against the unfused pair (dc.hip vgemm_kernel + rebuild.hip) this removes the VT round trip
// through HBM (2 x 132 KB per matrix) and replaces L2-latency-bound operand loads by LDS reads.
#include "common.h"

namespace admmnet {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BR_THREADS = 256;
constexpr int BR_KS = 16;          // K rows per slab
constexpr int BR_NCT = 5;          // eigenvector tiles: n <= 129 + padding
constexpr int BR_AP = 32 * BR_NCT; // slab A row pitch (floats)

__device__ __forceinline__ float br_eig_map(float w, float thr, const float *vn) {
    // vn: w1[16] b1[16] w2[16] b2[1]   (same as rebuild.hip eig_map)
    const float base = softplus_f(w - thr);
    const float a = fabsf(w);
    float acc = vn[48];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = fmaf(vn[32 + j], fmaxf(fmaf(vn[j], a, vn[16 + j]), 0.f), acc);
    return base * sigmoid_f(acc);
}

struct BrGeom {
    int D, n, NT, Dp, BP, VP;
    __host__ __device__ explicit BrGeom(int D_) : D(D_), n(D_ + 1), NT((D_ + 31) / 32), Dp(32 * ((D_ + 31) / 32)) {
        BP = 2 * Dp;       // slab B row pitch: real plane | imaginary plane, each padded to 32
        VP = 2 * Dp + 4;   // VT row pitch in LDS
    }
    __host__ __device__ size_t slab_floats() const { return (size_t)2 * BR_KS * (BR_AP + BP); }
    __host__ __device__ size_t vt_floats() const { return (size_t)n * VP; }
    __host__ __device__ size_t small_floats() const { return (size_t)3 * ((n + 4) & ~3) + 2 * Dp + 8; }
    __host__ __device__ size_t lds_bytes() const {
        const size_t big = slab_floats() > vt_floats() ? slab_floats() : vt_floats();
        return sizeof(float) * (big + small_floats());
    }
};

__global__ __launch_bounds__(BR_THREADS, 1) void back_rebuild_kernel(
    int D, const float *__restrict__ lw, const float *__restrict__ Wbuf, const float *__restrict__ QT,
    const float *__restrict__ wv, const float *__restrict__ w0v, const float2 *__restrict__ phi,
    const float *__restrict__ h, float2 *__restrict__ G, float *__restrict__ rn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const BrGeom g(D);
    const int n = g.n, NT = g.NT, Dp = g.Dp, BP = g.BP, VP = g.VP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, kh = lane >> 5;
    const int64_t b = blockIdx.x;
    float *big = reinterpret_cast<float *>(smem);
    const size_t bigf = g.slab_floats() > g.vt_floats() ? g.slab_floats() : g.vt_floats();
    float *fs = big + bigf;                          // [n+1] f(lambda)
    float *w0f = fs + ((n + 4) & ~3);                // [n+1] w0 * f
    float *z0s = w0f + ((n + 4) & ~3);               // [n+1] w0
    float *rowb = z0s + ((n + 4) & ~3);              // [2 Dp] arrow-row staging
    float *redb = rowb + 2 * Dp;                     // [8]
    float *slabA = big;                              // [2][KS][AP]
    float *slabB = big + (size_t)2 * BR_KS * BR_AP;  // [2][KS][BP]
    float *VTl = big;                                // [n][VP] (phase B onwards)

    const float *Wr = Wbuf + b * (int64_t)3 * n * n + (int64_t)2 * n * n;   // W[i][j] row-major (dc.hip)
    const float *Q = QT + b * ((int64_t)n * 2 * D);                         // QT[r][rho], pitch 2D

    {   // eigenvalue map (independent of phase A: overlaps its first loads)
        const LayerLayout L{D};
        const float thr = lw[S_THR];
        const float *vn = lw + L.off_vn();
        for (int c = tid; c <= n; c += BR_THREADS) {
            float f = 0.f, z0 = 0.f;
            if (c < n) {
                f = br_eig_map(wv[b * n + c], thr, vn);
                z0 = w0v[b * n + c];
            }
            fs[c] = f;
            w0f[c] = z0 * f;
            z0s[c] = z0;
        }
    }

    // ---------------- phase A: VT = W^T-rows x QT on the matrix cores, K streamed through LDS ------
    // staging registers of one slab: thread t carries column t of every slab row (A: t < 160 columns c,
    // B: t < BP columns rho'), so global loads and LDS stores are contiguous across the workgroup
    float ra[BR_KS], rb[BR_KS];
    const bool bim = tid >= Dp;
    const int bo = bim ? tid - Dp : tid;                   // offset inside the real / imaginary plane
    const bool bval = tid < BP && bo < D, aval = tid < n;
    const float *qcol = Q + (bim ? D : 0) + (bval ? bo : 0);
    const float *wcol = Wr + n + (aval ? tid : 0);         // row 1 + r of W
    auto gload = [&](int r0) {
#pragma unroll
        for (int q = 0; q < BR_KS; ++q) {
            const int r = r0 + q;
            ra[q] = (aval && r < D) ? wcol[(int64_t)r * n] : 0.f;
            rb[q] = (bval && r < D) ? qcol[(int64_t)r * 2 * D] : 0.f;
        }
    };
    auto lstore = [&](int buf) {
        float *sa = slabA + (size_t)buf * BR_KS * BR_AP, *sb = slabB + (size_t)buf * BR_KS * BP;
#pragma unroll
        for (int q = 0; q < BR_KS; ++q) {
            if (tid < BR_AP) sa[q * BR_AP + tid] = ra[q];
            if (tid < BP) sb[q * BP + tid] = rb[q];
        }
    };
    f32x16 accR[BR_NCT], accI[BR_NCT];
#pragma unroll
    for (int ct = 0; ct < BR_NCT; ++ct) {
        accR[ct] = f32x16{0};
        accI[ct] = f32x16{0};
    }
    const int nct = (n + 31) / 32;
    const bool wact = wave < NT;   // this wave owns row-tile `wave` of V
    const int nslab = (D + BR_KS - 1) / BR_KS;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload((s + 1) * BR_KS);
        if (wact) {
            const float *sa = slabA + (size_t)buf * BR_KS * BR_AP, *sb = slabB + (size_t)buf * BR_KS * BP;
#pragma unroll
            for (int kk = 0; kk < BR_KS / 2; ++kk) {
                const int row = 2 * kk + kh;
                const float bR = sb[row * BP + 32 * wave + l32];
                const float bI = sb[row * BP + Dp + 32 * wave + l32];
#pragma unroll
                for (int ct = 0; ct < BR_NCT; ++ct) {
                    if (ct < nct) {
                        const float a = sa[row * BR_AP + 32 * ct + l32];
                        accR[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bR, accR[ct], 0, 0, 0);
                        accI[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bI, accI[ct], 0, 0, 0);
                    }
                }
            }
        }
        if (s + 1 < nslab) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---------------- phase B: accumulators -> VT[c][rho'] in LDS ------------------------------------
    if (wact) {
#pragma unroll
        for (int ct = 0; ct < BR_NCT; ++ct) {
            if (ct < nct) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int c = 32 * ct + (q & 3) + 8 * (q >> 2) + 4 * kh;
                    if (c < n) {
                        VTl[c * VP + 32 * wave + l32] = accR[ct][q];
                        VTl[c * VP + Dp + 32 * wave + l32] = accI[ct][q];
                    }
                }
            }
        }
    }
    __syncthreads();

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
#pragma unroll 4
        for (int kk = 0; kk < n; kk += 2) {
            const int c = kk + kh;
            const bool cv = c < n;
            const int cc = cv ? c : 0;
            const float fc = cv ? fs[cc] : 0.f;
            const float *row = VTl + cc * VP;
            const float xr = row[i0 + l32] * fc, xi = row[Dp + i0 + l32] * fc;
            float yr = row[j0 + l32], yi = row[Dp + j0 + l32];
            yr = cv ? yr : 0.f;
            yi = cv ? yi : 0.f;
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(xr, yr, aRe, 0, 0, 0);
            aRe = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, yi, aRe, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, yr, aIm, 0, 0, 0);
            aIm = __builtin_amdgcn_mfma_f32_32x32x2f32(-xr, yi, aIm, 0, 0, 0);
        }
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
                    Gb[(int64_t)gj * n + gi] = make_float2(re, -im);
                    acc2 += 2.f * (re * re + im * im);
                }
            }
        }
    }

    // ---------------- arrow row (perm row 0 = original row D): G'[0][j] = sum_c w0_c f_c conj(V[j][c])
    for (int rp = tid; rp < 2 * Dp; rp += BR_THREADS) {
        float a = 0.f;
#pragma unroll 8
        for (int c = 0; c < n; ++c) a = fmaf(w0f[c], VTl[c * VP + rp], a);
        rowb[rp] = a;
    }
    __syncthreads();
    for (int o = tid; o < D; o += BR_THREADS) {
        const float gr = rowb[o], gim = -rowb[Dp + o];     // G[D][o]
        Gb[(int64_t)D * n + o] = make_float2(gr, gim);
        Gb[(int64_t)o * n + D] = make_float2(gr, -gim);
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
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < BR_THREADS / 64; ++i) s += redb[i];
        rn[b] = sqrtf(s);
    }
}

bool back_rebuild_supported(int D) { return D >= 1 && D <= 128; }

int launch_back_rebuild(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *G,
                        float *rn, float *w_out, const Ws &ws, hipStream_t st) {
    ProfScope _prof(KC_REBUILD, st);
    if (nb <= 0) return ADMMNET_OK;
    if (!back_rebuild_supported(D) || !ws.Wdc) {
        set_error("back_rebuild: D=%d unsupported", D);
        return ADMMNET_E_ARG;
    }
    const BrGeom g(D);
    const size_t lds = g.lds_bytes();
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(back_rebuild_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(back_rebuild_kernel, dim3((unsigned)nb), dim3(BR_THREADS), lds, st, D, lw, ws.Wdc, ws.QV,
                       ws.w, ws.w0, phi, h, G, rn);
    ADMM_HIP(hipGetLastError());
    const int n = D + 1;
    if (w_out) ADMM_HIP(hipMemcpyAsync(w_out, ws.w, sizeof(float) * nb * n, hipMemcpyDeviceToDevice, st));
    return ADMMNET_OK;
}

}  // namespace admmnet
