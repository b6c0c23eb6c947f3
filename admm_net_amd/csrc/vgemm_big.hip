// vgemm_big.hip -- K3' for 128 < D <= 256: back-transform of the divide & conquer eigenvectors,
//   VT[c][rho] = sum_r W[1 + r][c] QT[r][rho]      (V = diag(1, Q') W, planar transposed layout)
// i.e. the second half of torch.linalg.eigh at /root/reference/admm_net.py:303 for the matrices too large
// for the fused kernel (backrebuild.hip keeps V in LDS up to D = 128).
//
// Same scheme as phase A of backrebuild.hip, scaled to the larger matrix: one 256-thread workgroup per
// (matrix, group of eigenvector tiles), one wave per SIMD.  K (= r) is streamed in slabs of 16 rows through
// LDS with register-staged double buffering; wave w owns row-tiles {w, w + 4} of V, real and imaginary plane,
// for the (up to 4) eigenvector tiles of its group: up to 16 accumulator tiles, and every k-step feeds
// 4 NCH MFMAs from NCH + 4 LDS reads.  The A operand is read from the TRANSPOSED eigenvector image the
// D&C kernel leaves behind (dc_final_offset), so that kernel's final transpose is skipped on this path.
#include "common.h"

namespace admmnet {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int VB_THREADS = 256;
constexpr int VB_KS = 16;              // K rows per slab
constexpr int VB_DP = 256;             // plane width of the B slab (real | imaginary)
constexpr int VB_BP = 2 * VB_DP + 32;  // B row pitch: the two k-halves of a wave read rows 32 banks apart

__host__ __device__ constexpr int vb_ap(int nch) { return 32 * nch + 1; }   // odd: transposing stores spread over the banks
__host__ __device__ constexpr size_t vb_lds_bytes(int nch) {
    return sizeof(float) * 2 * VB_KS * (size_t)(vb_ap(nch) + VB_BP);
}

// NCH eigenvector tiles starting at eigenvector `cbase`; NSLAB > 0: compile-time slab count (fully unrolled
// slab loop: with a back-edge the accumulators are shuffled between VGPRs and AGPRs around every slab)
template <int NCH, int NSLAB, bool VEC>
__global__ __launch_bounds__(VB_THREADS, 1) void vgemm_big_kernel(int D, int cbase, const float *__restrict__ Wbuf,
                                                                  const float *__restrict__ QT,
                                                                  float *__restrict__ VT, int64_t wt_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int AP = vb_ap(NCH);
    const int n = D + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, kh = lane >> 5;
    const int64_t b = blockIdx.x;
    float *slabA = reinterpret_cast<float *>(smem);          // [2][KS][AP]
    float *slabB = slabA + (size_t)2 * VB_KS * AP;           // [2][KS][BP]
    const float *WT = Wbuf + b * (int64_t)3 * n * n + wt_off;   // WT[c][i] = W[i][c]
    const float *Q = QT + b * ((int64_t)n * 2 * D);             // QT[r][rho], pitch 2 D
    float *V = VT + b * ((int64_t)n * 2 * D);

    // slab staging: A element (rl, c') <- WT[cbase + c'][1 + r0 + rl], thread -> rl = tid & 15 (64-byte runs),
    // stored transposed; B rows as 16-byte chunks (D % 4 == 0) or two columns per thread
    constexpr int NAS = 2 * NCH;                         // 32 NCH x 16 / 256
    constexpr int NBV = VB_KS * (2 * 256 / 4) / VB_THREADS;   // 8 at D = 256
    float sa_reg[NAS];
    float4 vb[VEC ? NBV : 1];
    float rb[VEC ? 1 : 2 * VB_KS];
    const int a_rl = tid & (VB_KS - 1), a_c0 = tid / VB_KS;
    const int cpr = D / 2;                               // B chunks per row
    auto gload = [&](int r0) {
        const bool rok = r0 + a_rl < D;
        const float *src = WT + 1 + r0 + a_rl;
#pragma unroll
        for (int q = 0; q < NAS; ++q) {
            const int c = cbase + a_c0 + q * (VB_THREADS / VB_KS);
            sa_reg[q] = (rok && c < n) ? src[(int64_t)c * n] : 0.f;
        }
        if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < NBV; ++q) {
                const int idx = tid + q * VB_THREADS;
                const int row = idx / cpr, gq = idx - row * cpr;
                const int r = r0 + row;
                vb[q] = (row < VB_KS && r < D) ? *reinterpret_cast<const float4 *>(Q + (int64_t)r * 2 * D + 4 * gq)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int q = 0; q < VB_KS; ++q) {
                const bool rv = r0 + q < D;
                const float *qr = Q + (int64_t)(rv ? r0 + q : 0) * 2 * D;
                rb[2 * q] = (rv && tid < 2 * D) ? qr[tid] : 0.f;
                rb[2 * q + 1] = (rv && tid + VB_THREADS < 2 * D) ? qr[tid + VB_THREADS] : 0.f;
            }
        }
    };
    auto lstore = [&](int buf) {
        float *sa = slabA + (size_t)buf * VB_KS * AP, *sb = slabB + (size_t)buf * VB_KS * VB_BP;
#pragma unroll
        for (int q = 0; q < NAS; ++q) sa[a_rl * AP + a_c0 + q * (VB_THREADS / VB_KS)] = sa_reg[q];
        if constexpr (VEC) {
#pragma unroll
            for (int q = 0; q < NBV; ++q) {
                const int idx = tid + q * VB_THREADS;
                const int row = idx / cpr, gq = idx - row * cpr;
                if (row < VB_KS) {
                    const int o = 4 * gq;
                    const int lo = (o >= D) ? VB_DP + (o - D) : o;
                    *reinterpret_cast<float4 *>(sb + row * VB_BP + lo) = vb[q];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < VB_KS; ++q) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int o = tid + u * VB_THREADS;
                    if (o < 2 * D) sb[q * VB_BP + ((o >= D) ? VB_DP + (o - D) : o)] = rb[2 * q + u];
                }
            }
        }
    };
    // plane columns [D, 256) of the B slabs are never written by the staging: clear everything once
    for (int i = tid; i < 2 * VB_KS * (AP + VB_BP); i += VB_THREADS) slabA[i] = 0.f;
    __syncthreads();
    f32x16 acc[NCH][4];
#pragma unroll
    for (int ct = 0; ct < NCH; ++ct)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[ct][p] = f32x16{0};
    const int nslab = NSLAB > 0 ? NSLAB : (D + VB_KS - 1) / VB_KS;
    gload(0);
    lstore(0);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < (NSLAB > 0 ? NSLAB : nslab); ++s) {
        const int buf = s & 1;
        if (s + 1 < nslab) gload((s + 1) * VB_KS);
        {
            const float *sa = slabA + (size_t)buf * VB_KS * AP, *sb = slabB + (size_t)buf * VB_KS * VB_BP;
            // operands of k-step kk + 1 are read before the MFMAs of k-step kk are issued (see backrebuild.hip)
            const float *sap = sa + kh * AP + l32, *sbp = sb + kh * VB_BP + 32 * wave + l32;
            float a_cur[NCH], a_nxt[NCH], b_cur[4], b_nxt[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ct = 0; ct < NCH; ++ct) a_cur[ct] = sap[32 * ct];
            b_cur[0] = sbp[0];
            b_cur[1] = sbp[VB_DP];
            b_cur[2] = sbp[128];
            b_cur[3] = sbp[VB_DP + 128];
#pragma unroll
            for (int kk = 0; kk < VB_KS / 2; ++kk) {
                if (kk + 1 < VB_KS / 2) {
#pragma unroll
                    for (int ct = 0; ct < NCH; ++ct) a_nxt[ct] = sap[2 * (kk + 1) * AP + 32 * ct];
                    b_nxt[0] = sbp[2 * (kk + 1) * VB_BP];
                    b_nxt[1] = sbp[2 * (kk + 1) * VB_BP + VB_DP];
                    b_nxt[2] = sbp[2 * (kk + 1) * VB_BP + 128];
                    b_nxt[3] = sbp[2 * (kk + 1) * VB_BP + VB_DP + 128];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < NCH; ++ct)
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        acc[ct][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[ct], b_cur[p], acc[ct][p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ct = 0; ct < NCH; ++ct) a_cur[ct] = a_nxt[ct];
#pragma unroll
                for (int p = 0; p < 4; ++p) b_cur[p] = b_nxt[p];
            }
        }
        if (s + 1 < nslab) lstore(buf ^ 1);
        __syncthreads();
    }
    // accumulators -> VT[c][plane D + o] (128-byte runs per tile row)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int o = 32 * wave + 128 * (p >> 1) + l32;
        if (o >= D) continue;
        float *vcol = V + (p & 1) * D + o;
#pragma unroll
        for (int ct = 0; ct < NCH; ++ct) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int c = cbase + 32 * ct + (q & 3) + 8 * (q >> 2) + 4 * kh;
                if (c < n) vcol[(int64_t)c * 2 * D] = acc[ct][p][q];
            }
        }
    }
}

bool vgemm_big_supported(int D) { return D > 128 && D <= 256; }

template <int NCH>
static int vb_launch_half(int D, int cbase, int64_t nb, const Ws &ws, hipStream_t st) {
    const bool v4 = (D & 3) == 0;
    auto kern = (D == 256) ? vgemm_big_kernel<NCH, 16, true>
                : v4       ? vgemm_big_kernel<NCH, 0, true>
                           : vgemm_big_kernel<NCH, 0, false>;
    const size_t lds = vb_lds_bytes(NCH);
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(VB_THREADS), lds, st, D, cbase, ws.Wdc, ws.QV, ws.VT,
                       dc_final_offset(D + 1));
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_vgemm_big(int D, int64_t nb, const Ws &ws, hipStream_t st) {
    ProfScope _prof(KC_ROTAPPLY, st);
    if (nb <= 0) return ADMMNET_OK;
    if (!vgemm_big_supported(D) || !ws.Wdc) {
        set_error("vgemm_big: D=%d unsupported", D);
        return ADMMNET_E_ARG;
    }
    // eigenvector tiles in groups of at most 4 (16 accumulator tiles per wave fill the AGPRs): two launches, plus a
    // third for the single leftover tile when n = 257
    const int nct = (D + 1 + 31) / 32;
    const int nc0 = nct >= 8 ? 4 : (nct + 1) / 2, nc1 = nct >= 8 ? 4 : nct - nc0, nc2 = nct - nc0 - nc1;
    auto go = [&](int nch, int cbase) {
        switch (nch) {
            case 1: return vb_launch_half<1>(D, cbase, nb, ws, st);
            case 2: return vb_launch_half<2>(D, cbase, nb, ws, st);
            case 3: return vb_launch_half<3>(D, cbase, nb, ws, st);
            default: return vb_launch_half<4>(D, cbase, nb, ws, st);
        }
    };
    int rc;
    if ((rc = go(nc0, 0))) return rc;
    if ((rc = go(nc1, 32 * nc0))) return rc;
    return nc2 > 0 ? go(nc2, 32 * (nc0 + nc1)) : ADMMNET_OK;
}

}  // namespace admmnet
