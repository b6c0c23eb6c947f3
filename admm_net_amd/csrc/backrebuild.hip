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
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "rebuild_lds.h"

namespace admmnet {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BR_THREADS = 256;
constexpr int BR_KS = 16;          // K rows per slab
constexpr int BR_NCT = 5;          // eigenvector tiles: n <= 129 + padding
constexpr int BR_AP = 32 * BR_NCT + 1; // slab A row pitch (floats): odd, the transposing LDS stores of the staging spread over the banks

// slab double buffer of phase A
__host__ __device__ inline size_t br_slab_floats(const BrGeom &g) { return (size_t)2 * BR_KS * (BR_AP + g.BP); }
__host__ __device__ inline size_t br_lds_bytes(const BrGeom &g) {
    const size_t big = br_slab_floats(g) > g.vt_floats() ? br_slab_floats(g) : g.vt_floats();
    return sizeof(float) * (big + g.small_floats());
}

// NCT = ceil(n / 32) eigenvector tiles (compile time: a run-time tile count puts a branch and an exposed LDS
// wait in front of every MFMA pair)
// NSLAB > 0: compile-time slab count (the slab loop is fully unrolled: with a loop back-edge the 160
// accumulators are carried in VGPRs and copied to and from the AGPRs around every slab); 0: run-time count.
template <int NCT, int NSLAB, bool VEC>
__global__ __launch_bounds__(BR_THREADS, 1) void back_rebuild_kernel(
    int D, const float *__restrict__ lw, const float *__restrict__ Wbuf, const float *__restrict__ QT,
    const float *__restrict__ wv, const float *__restrict__ w0v, const float2 *__restrict__ phi,
    const float *__restrict__ h, float2 *__restrict__ G, float *__restrict__ rn,
    unsigned long long *__restrict__ ptime, int64_t wt_off, int lower_only, const int *__restrict__ skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // developer phase timer (ADMMNET_BR_TIMING=1): cycles of thread 0 between marks
    long long t_prev = ptime ? clock64() : 0;
    auto mark = [&](int id) {
        if (ptime && threadIdx.x == 0) {
            const long long t_now = clock64();
            atomicAdd(&ptime[id], (unsigned long long)(t_now - t_prev));
            t_prev = t_now;
        }
    };
    const BrGeom g(D);
    const int n = g.n, NT = g.NT, Dp = g.Dp, BP = g.BP, VP = g.VP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, kh = lane >> 5;
    const int64_t b = blockIdx.x;
    if (skip && skip[b] == 0) return;   // (uniform) this matrix' G is already there: spectral.hip
    float *big = reinterpret_cast<float *>(smem);
    const size_t bigf = br_slab_floats(g) > g.vt_floats() ? br_slab_floats(g) : g.vt_floats();
    float *fs = big + bigf;                          // [n+1] f(lambda)
    float *w0f = fs + ((n + 4) & ~3);                // [n+1] w0 * f
    float *z0s = w0f + ((n + 4) & ~3);               // [n+1] w0
    float *rowb = z0s + ((n + 4) & ~3);              // [2 Dp] arrow-row staging
    float *redb = rowb + 2 * Dp;                     // [8]
    float *slabA = big;                              // [2][KS][AP]
    float *slabB = big + (size_t)2 * BR_KS * BR_AP;  // [2][KS][BP]
    float *VTl = big;                                // [n][VP] (phase B onwards)

    const float *WT = Wbuf + b * (int64_t)3 * n * n + wt_off;   // WT[c][i] = W[i][c], c = eigenvalue (dc.hip)
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
    // Slab staging.  A[c][k = r] = W[1 + r][c] = WT[c][1 + r]: the D&C kernel leaves the eigenvectors
    // transposed (WT[c][.] contiguous), so a 16-row slab is 16 consecutive floats of every WT row c: element
    // e = 16 c + rl is loaded by thread e (mod 256) -- 64-byte runs, 4 rows per wave-load -- and stored
    // transposed into the slab (row pitch 161: odd, the 16 lanes of a run hit 16 different banks).
    // D % 4 == 0 (every geometry of the reference): the 16 x 2D B slab (QT rows, 16-byte aligned) moves as
    // float4 chunks, 4 per thread at D = 128.  Otherwise thread t carries column t of every B row.
    constexpr bool vec = VEC;   // D % 4 == 0
    constexpr int NAS = (BR_KS * 129 + BR_THREADS - 1) / BR_THREADS;   // 9: slab elements per thread (n <= 129)
    constexpr int NBV = BR_KS * (2 * 128 / 4) / BR_THREADS;            // 4 at D = 128
    float sa_reg[NAS];
    float4 vb[NBV];
    float rb[BR_KS];
    const bool bim = tid >= Dp;
    const int bo = bim ? tid - Dp : tid;                   // offset inside the real / imaginary plane
    const bool bval = tid < BP && bo < D;
    const float *qcol = Q + (bim ? D : 0) + (bval ? bo : 0);
    const int cpr = D / 2;                                 // B chunks per row (2 D / 4)
    const int a_rl = tid & (BR_KS - 1), a_c0 = tid / BR_KS;   // thread -> (row of the slab, first eigenvector c)
    auto gload = [&](int r0) {
        const bool rok = r0 + a_rl < D;
        const float *src = WT + 1 + r0 + a_rl;
#pragma unroll
        for (int q = 0; q < NAS; ++q) {
            const int c = a_c0 + q * (BR_THREADS / BR_KS);
            sa_reg[q] = (rok && c < n) ? src[(int64_t)c * n] : 0.f;
        }
        if constexpr (vec) {
#pragma unroll
            for (int q = 0; q < NBV; ++q) {
                const int idx = tid + q * BR_THREADS;
                const int row = idx / cpr, gq = idx - row * cpr;
                const int r = r0 + row;
                vb[q] = (row < BR_KS && r < D) ? *reinterpret_cast<const float4 *>(Q + (int64_t)r * 2 * D + 4 * gq)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int q = 0; q < BR_KS; ++q) rb[q] = (bval && r0 + q < D) ? qcol[(int64_t)(r0 + q) * 2 * D] : 0.f;
        }
    };
    auto lstore = [&](int buf) {
        float *sa = slabA + (size_t)buf * BR_KS * BR_AP, *sb = slabB + (size_t)buf * BR_KS * BP;
#pragma unroll
        for (int q = 0; q < NAS; ++q) {
            const int c = a_c0 + q * (BR_THREADS / BR_KS);
            if (c < n) sa[a_rl * BR_AP + c] = sa_reg[q];   // columns [n, 161) stay zero (cleared once below)
        }
        if constexpr (vec) {
#pragma unroll
            for (int q = 0; q < NBV; ++q) {
                const int idx = tid + q * BR_THREADS;
                const int row = idx / cpr, gq = idx - row * cpr;
                if (row < BR_KS) {
                    const int o = 4 * gq;                       // float offset inside the 2 D global row
                    const int lo = (o >= D) ? Dp + (o - D) : o;   // real plane | imaginary plane (padded to Dp)
                    *reinterpret_cast<float4 *>(sb + row * BP + lo) = vb[q];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < BR_KS; ++q)
                if (tid < BP) sb[q * BP + tid] = rb[q];
        }
    };
    // the padding columns of both A slab buffers (c in [n, 160)) and, when D % 32 != 0, of the B planes are
    // never written by the staging: clear the slab area once
    for (int i = tid; i < (int)br_slab_floats(g); i += BR_THREADS) big[i] = 0.f;
    __syncthreads();
    f32x16 accR[NCT], accI[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        accR[ct] = f32x16{0};
        accI[ct] = f32x16{0};
    }
    const bool wact = wave < NT;   // this wave owns row-tile `wave` of V
    const int nslab = NSLAB > 0 ? NSLAB : (D + BR_KS - 1) / BR_KS;
    mark(0);
    gload(0);
    lstore(0);
    __syncthreads();
    mark(1);
#pragma unroll
    for (int s = 0; s < (NSLAB > 0 ? NSLAB : nslab); ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload((s + 1) * BR_KS);
        if (wact) {
            const float *sa = slabA + (size_t)buf * BR_KS * BR_AP, *sb = slabB + (size_t)buf * BR_KS * BP;
            // operands of k-step kk + 1 are read from LDS before the MFMAs of k-step kk are issued
            // (sched_barrier pins that order: left alone, the scheduler sinks every read to just before
            // its first use and each MFMA pair then waits out a full LDS latency)
            const float *sap = sa + kh * BR_AP + l32, *sbp = sb + kh * BP + 32 * wave + l32;
            float a_cur[NCT], a_nxt[NCT], bR_cur, bI_cur, bR_nxt = 0.f, bI_nxt = 0.f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) a_cur[ct] = sap[32 * ct];
            bR_cur = sbp[0];
            bI_cur = sbp[Dp];
#pragma unroll
            for (int kk = 0; kk < BR_KS / 2; ++kk) {
                if (kk + 1 < BR_KS / 2) {
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) a_nxt[ct] = sap[2 * (kk + 1) * BR_AP + 32 * ct];
                    bR_nxt = sbp[2 * (kk + 1) * BP];
                    bI_nxt = sbp[2 * (kk + 1) * BP + Dp];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    accR[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ct], bR_cur, accR[ct], 0, 0, 0);
                    accI[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ct], bI_cur, accI[ct], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) a_cur[ct] = a_nxt[ct];
                bR_cur = bR_nxt;
                bI_cur = bI_nxt;
            }
        }
        if (s + 1 < nslab) lstore(buf ^ 1);
        __syncthreads();
    }

    mark(2);
    // ---------------- phase B: accumulators -> VT[c][rho'] in LDS ------------------------------------
    if (wact) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
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
    __syncthreads();
    mark(3);

    rebuild_from_lds(g, b, lw, VTl, fs, w0f, z0s, rowb, redb, phi, h, G, rn, mark, lower_only);
}

bool back_rebuild_supported(int D) { return D >= 1 && D <= 128; }

int launch_back_rebuild(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *G,
                        float *rn, float *w_out, const Ws &ws, hipStream_t st, bool lower_only) {
    ProfScope _prof(KC_REBUILD, st);
    if (nb <= 0) return ADMMNET_OK;
    if (!back_rebuild_supported(D) || !ws.Wdc) {
        set_error("back_rebuild: D=%d unsupported", D);
        return ADMMNET_E_ARG;
    }
    const BrGeom g(D);
    const size_t lds = br_lds_bytes(g);
    const int nct = (D + 1 + 31) / 32;
    const bool v4 = (D & 3) == 0;   // 16-byte slab staging
    auto kern = nct == 1   ? (v4 ? back_rebuild_kernel<1, 0, true> : back_rebuild_kernel<1, 0, false>)
                : nct == 2 ? (v4 ? back_rebuild_kernel<2, 0, true> : back_rebuild_kernel<2, 0, false>)
                : nct == 3 ? (v4 ? back_rebuild_kernel<3, 0, true> : back_rebuild_kernel<3, 0, false>)
                : D == 100 ? back_rebuild_kernel<4, 7, true>   // the reference's default 10 x 10 grid: unrolled slab loop too
                : nct == 4 ? (v4 ? back_rebuild_kernel<4, 0, true> : back_rebuild_kernel<4, 0, false>)
                : D == 128 ? back_rebuild_kernel<5, 8, true>
                           : (v4 ? back_rebuild_kernel<5, 0, true> : back_rebuild_kernel<5, 0, false>);
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    static const bool timing = getenv("ADMMNET_BR_TIMING") != nullptr;   // developer aid, never on by default
    unsigned long long *ptime = nullptr;
    if (timing) {
        ADMM_HIP(hipMalloc(&ptime, 16 * sizeof(unsigned long long)));
        ADMM_HIP(hipMemsetAsync(ptime, 0, 16 * sizeof(unsigned long long), st));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(BR_THREADS), lds, st, D, lw, ws.Wdc, ws.QV,
                       ws.w, ws.w0, phi, h, G, rn, ptime, dc_final_offset(D + 1), lower_only ? 1 : 0, ws.skip);
    ADMM_HIP(hipGetLastError());
    if (timing) {
        unsigned long long hb[16];
        ADMM_HIP(hipMemcpyAsync(hb, ptime, sizeof(hb), hipMemcpyDeviceToHost, st));
        ADMM_HIP(hipStreamSynchronize(st));
        ADMM_HIP(hipFree(ptime));
        static const char *nm[6] = {"eig map", "first slab", "phase A", "phase B", "phase C", "arrow+norm"};
        fprintf(stderr, "[back_rebuild timing] D=%d nb=%lld  mean cycles per workgroup:\n", D, (long long)nb);
        for (int i = 0; i < 6; ++i) fprintf(stderr, "   %-12s %10.0f\n", nm[i], (double)hb[i] / (double)nb);
    }
    const int n = D + 1;
    if (w_out) ADMM_HIP(hipMemcpyAsync(w_out, ws.w, sizeof(float) * nb * n, hipMemcpyDeviceToDevice, st));
    return ADMMNET_OK;
}

}  // namespace admmnet
