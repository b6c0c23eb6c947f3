// tridiag_big.hip -- K1 for 128 < D <= 256 (BASELINE cfg 3/4/5: D = 256, n = 257).
//
// A 256 x 256 complex matrix is 512 KB -- exactly the whole VGPR file of a CU -- so the
// register-resident scheme of tridiag_reg.hip is applied to the HERMITIAN HALF: a 1024-thread
// workgroup (32 x 32 grid, 2D-cyclic with period 32) keeps the block-lower triangle
// (diagonal blocks in full): 36 complex block slots per thread at NA = 8.  The 128-VGPR budget of
// a 1024-thread workgroup cannot hold them all (344 spills), so the diagonal and first
// sub-diagonal block slots (15) live in LDS -- 120 KB, lane-contiguous, conflict free -- and the
// other 21 in VGPRs.
//   p = M v needs two reductions now: the row part along tj (the 32 fast lanes, DPP) and the
//   mirrored part along ti (in-wave swap + one 32 KB LDS pass);  3 barriers per reflector.
// The explicit Q (not Hermitian: 512 KB) is formed by a second kernel, ungtr_big_kernel, as
// P = Q^H with its ROWS split over two workgroups (rows of P are independent under
// P <- P H^H), barrier free, in packed (v_pk_fma_f32) complex arithmetic.  (The packed form was
// also tried in the tridiagonalisation kernel below: its extra live vectors raised the spill
// count from 57 to 178 VGPRs and made it 30 % slower, so that kernel keeps scalar FMAs.)  Same mathematics as tridiag_reg.hip (LAPACK chetd2 + cung2l,
// first half of torch.linalg.eigh at /root/reference/admm_net.py:303).
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace admmnet {

constexpr int TB_THREADS = 1024;

__device__ __forceinline__ float row32_sum(float x) {   // all-reduce over the 32 lanes of a half wave
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xF, 0xF, false));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xF, 0xF, false));
    // every lane of a 16-lane row now holds its row's sum; add the partner row (0 <-> 1, 2 <-> 3) with
    // v_permlane16_swap (gfx950: odd rows of the first operand <-> even rows of the second) instead of a
    // ds_bpermute round trip through the LDS pipeline:  (a, b) = (x, x) -> a = [x0 x0 x2 x2], b = [x1 x1 x3 x3]
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
// x + (x of the lane 32 away): v_permlane32_swap exchanges the upper half of the first operand with the lower
// half of the second
__device__ __forceinline__ float half_pair_sum(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float wave_sum_fast(float x) { return half_pair_sum(row32_sum(x)); }   // all 64 lanes

typedef float v2 __attribute__((ext_vector_type(2)));   // packed complex arithmetic, see tridiag_reg.hip
__device__ __forceinline__ v2 b_tov2(float2 a) { return v2{a.x, a.y}; }
__device__ __forceinline__ float2 b_tof2(v2 a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ v2 b_rot(v2 v) { return v2{-v.y, v.x}; }
__device__ __forceinline__ v2 b_rotc(v2 v) { return v2{v.y, -v.x}; }
__device__ __forceinline__ v2 b_cmac(v2 acc, v2 m, v2 v, v2 vj) {   // acc + m v, vj = rot(v)
    acc = __builtin_elementwise_fma(m.xx, v, acc);
    return __builtin_elementwise_fma(m.yy, vj, acc);
}
__device__ __forceinline__ v2 b_cmacc(v2 acc, v2 a, v2 aj, v2 b) {   // acc + a conj(b), aj = rotc(a)
    acc = __builtin_elementwise_fma(b.xx, a, acc);
    return __builtin_elementwise_fma(b.yy, aj, acc);
}

// operand-select forms (see tridiag_reg.hip): complex products straight from (re, im) pairs, no rotated copies.
// Their results are invisible to the compiler's hazard recogniser: never feed one directly into a DPP read.
__device__ __forceinline__ v2 b_cmac_sel(v2 acc, v2 m, v2 v) {   // acc + m v
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "+v"(acc)
        : "v"(m), "v"(v));
    return acc;
}
__device__ __forceinline__ v2 b_cmsubc_sel(v2 acc, v2 a, v2 b) {   // acc - a conj(b)
    asm("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,1,0]"
        : "+v"(acc)
        : "v"(a), "v"(b));
    return acc;
}

template <int NA>
struct TbShared {
    float2 colbuf[2][32 * NA];   // x = column below the unit position, ZERO at i <= u and i >= D
    float2 head[2];              // alpha = column entry at the unit position
    float dprev[2];              // finished diagonal entry d[u]
    float2 prow[32 * NA];
    float2 pfull[32 * NA];
    float2 cpart[16][32 * (NA - 1)];
    float2 dotbuf[16];
};

// Block (a, b), a >= b: the diagonal (a == b) and first sub-diagonal (a == b + 1) block slots live
// in LDS (lds slot a, resp. NA + b), the rest in registers.
constexpr __host__ __device__ int tb_slot(int a, int b) { return (a - 1) * (a - 2) / 2 + b; }   // a >= b + 2
constexpr int tb_nslot(int na) { return (na - 1) * (na - 2) / 2 > 0 ? (na - 1) * (na - 2) / 2 : 1; }
#define TB_LDS(a, b) dg[((a) == (b) ? (a) : NA + (b)) * TB_THREADS + threadIdx.x]
#define TB_GET(a, b) ((a) - (b) <= 1 ? TB_LDS(a, b) : m[tb_slot((a), (b))])
#define TB_SET(a, b, val)                         \
    do {                                          \
        if ((a) - (b) <= 1) TB_LDS(a, b) = (val); \
        else m[tb_slot((a), (b))] = (val);        \
    } while (0)

template <int NA, int A0>
__device__ __forceinline__ void tb_step(float2 (&m)[tb_nslot(NA)], float2 *__restrict__ dg, TbShared<NA> &sh,
                                        int u, int D, float corner, float2 *__restrict__ Mg, float *__restrict__ dcol,
                                        float *__restrict__ ecol) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tj = tid & 31, ti = tid >> 5;
    const int par = u & 1;
    if (u == 0) {
        for (int i = tid; i < 32 * NA; i += TB_THREADS) {
            const float2 x = (i < D) ? Mg[(int64_t)D * D + i] : make_float2(0.f, 0.f);
            sh.colbuf[0][i] = (i > 0) ? x : make_float2(0.f, 0.f);
            if (i == 0) sh.head[0] = x;
        }
    } else {
        const int k = u - 1;
        if (tj == (k & 31)) {
            auto put = [&](int a, float2 x) {
                const int i = 32 * a + ti;
                sh.colbuf[par][i] = (i > u) ? x : make_float2(0.f, 0.f);
                if (i == u) sh.head[par] = x;
                if (i == k) sh.dprev[par] = x.x;
            };
            if ((k >> 5) == A0) {
#pragma unroll
                for (int a = A0; a < NA; ++a) put(a, TB_GET(a, A0));
            } else {
                constexpr int B = A0 > 0 ? A0 - 1 : 0;
#pragma unroll
                for (int a = B; a < NA; ++a) put(a, TB_GET(a, B));
            }
        }
    }
    __syncthreads();   // (A)
    const float2 *col = sh.colbuf[par];
    float pn = 0.f;
    for (int i = lane; i < 32 * NA; i += 64) {
        const float2 x = col[i];
        pn += x.x * x.x + x.y * x.y;
    }
    const float xn2 = wave_sum_fast(pn);
    const float2 alpha = sh.head[par];
    float beta, tr, tim, sr, si;
    householder_c(alpha.x, alpha.y, xn2, beta, tr, tim, sr, si);
    // UNNORMALISED reflector (as in tridiag_reg.hip): H = I - tau v v^H with v = s u is I - gamma u u^H,
    // gamma = tau |s|^2, u = (alpha - beta at the unit position, x below): no per-entry scaling anywhere
    const float g2 = sr * sr + si * si;
    const float2 tau = make_float2(tr * g2, tim * g2);
    const float2 hu = make_float2(alpha.x - beta, alpha.y);
    auto vat = [&](int i) -> float2 {   // component i of the reflector (the column buffer is zero above the unit entry / beyond D)
        return (i == u) ? hu : col[i];
    };
    if (tid == 0) {
        ecol[u] = beta;
        dcol[u] = (u == 0) ? corner : sh.dprev[par];
    }
    if (tid < D) Mg[(int64_t)u * D + tid] = vat(tid);          // reflector row u for the Q kernel
    if (tid == 0) Mg[(int64_t)D * D + u] = tau;                 // taus live in the consumed arrow slot
    if (tr == 0.f && tim == 0.f) return;                        // H = I (uniform)

    // ---- p = M v with the Hermitian half, in two register-lean passes:
    //      row part  sum_{j <= blk(i)} M_ij v_j  (reduce over tj, the 32 fast lanes), then the
    //      mirrored part  sum_{i > blk(j)} conj(M_ij) v_i  (reduce over ti: in-wave swap + LDS)
    {
        float2 vc[NA];
#pragma unroll
        for (int b = A0; b < NA; ++b) vc[b] = vat(32 * b + tj);
#pragma unroll
        for (int a = A0; a < NA; ++a) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int b = A0; b <= a; ++b) {
                const float2 x = TB_GET(a, b);
                acc.x = fmaf(x.x, vc[b].x, fmaf(-x.y, vc[b].y, acc.x));
                acc.y = fmaf(x.x, vc[b].y, fmaf(x.y, vc[b].x, acc.y));
            }
            acc.x = row32_sum(acc.x);
            acc.y = row32_sum(acc.y);
            if (tj == a) sh.prow[32 * a + ti] = acc;
            __builtin_amdgcn_sched_barrier(0);   // keep the block-row loads inside their iteration (VGPR budget)
        }
    }
    {
        float2 vr[NA];
#pragma unroll
        for (int a = A0; a < NA; ++a) vr[a] = vat(32 * a + ti);
#pragma unroll
        for (int b = A0; b < NA - 1; ++b) {
            float2 t = make_float2(0.f, 0.f);
#pragma unroll
            for (int a = b + 1; a < NA; ++a) t = cmacc(t, TB_GET(a, b), vr[a]);   // conj(M_ij) v_i -> row j
            t.x = half_pair_sum(t.x);
            t.y = half_pair_sum(t.y);
            if (lane < 32) sh.cpart[wave][32 * b + tj] = t;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();   // (B)
    if (tid < 256) {
        float2 dotp = make_float2(0.f, 0.f);
        if (tid < 32 * NA && tid >= 32 * A0) {
            float2 s = sh.prow[tid];
            if ((tid >> 5) < NA - 1) {
#pragma unroll
                for (int w = 0; w < 16; ++w) {
                    const float2 t = sh.cpart[w][tid];
                    s.x += t.x;
                    s.y += t.y;
                }
            }
            const float2 p = cmul(tau, s);
            sh.pfull[tid] = p;
            dotp = cmacc(dotp, p, vat(tid));
        }
        dotp.x = wave_sum_fast(dotp.x);
        dotp.y = wave_sum_fast(dotp.y);
        if (lane == 0) sh.dotbuf[wave] = dotp;
    }
    __syncthreads();   // (C)
    float2 dot = sh.dotbuf[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        dot.x += sh.dotbuf[q].x;
        dot.y += sh.dotbuf[q].y;
    }
    float2 al = cmul(tau, dot);
    al.x *= -0.5f;
    al.y *= -0.5f;
    auto wat = [&](int i, float2 v) -> float2 {
        if (!(i >= u && i < D)) return make_float2(0.f, 0.f);
        const float2 p = sh.pfull[i], t = cmul(al, v);
        return make_float2(p.x + t.x, p.y + t.y);
    };
    float2 vc[NA], wc[NA];
#pragma unroll
    for (int b = A0; b < NA; ++b) {
        vc[b] = vat(32 * b + tj);
        wc[b] = wat(32 * b + tj, vc[b]);
    }
#pragma unroll
    for (int a = A0; a < NA; ++a) {
        const float2 vra = vat(32 * a + ti);
        const float2 wra = wat(32 * a + ti, vra);
#pragma unroll
        for (int b = A0; b <= a; ++b) {
            const float2 t1 = cmulc(vra, wc[b]), t2 = cmulc(wra, vc[b]);
            float2 x = TB_GET(a, b);
            x.x -= t1.x + t2.x;
            x.y -= t1.y + t2.y;
            if (a == b && ti == tj) x.y = 0.f;
            TB_SET(a, b, x);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NA, int A0>
struct TbPhases {
    static __device__ __forceinline__ void run(float2 (&m)[tb_nslot(NA)], float2 *dg, TbShared<NA> &sh, int D,
                                               float corner, float2 *Mg, float *dcol, float *ecol) {
        const int hi = min(32 * (A0 + 1), D);
        for (int u = 32 * A0; u < hi; ++u) tb_step<NA, A0>(m, dg, sh, u, D, corner, Mg, dcol, ecol);
        if constexpr (A0 + 1 < NA) TbPhases<NA, A0 + 1>::run(m, dg, sh, D, corner, Mg, dcol, ecol);
    }
};

template <int NA>
__global__ __launch_bounds__(TB_THREADS) void tridiag_big_kernel(int D, float2 *__restrict__ Mbuf,
                                                                 float *__restrict__ dT, float *__restrict__ eT) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TbShared<NA> &sh = *reinterpret_cast<TbShared<NA> *>(smem);
    float2 *dg = reinterpret_cast<float2 *>(smem + sizeof(TbShared<NA>));   // [2 NA - 1][1024] LDS block slots
    const int tid = threadIdx.x;
    const int tj = tid & 31, ti = tid >> 5;
    const int64_t bm = blockIdx.x;
    const int n = D + 1;
    float2 *Mg = Mbuf + bm * ((int64_t)D * D + D + 1);
    float *dcol = dT + bm * n, *ecol = eT + bm * n;
    float2 m[tb_nslot(NA)];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            const int i = 32 * a + ti, j = 32 * b + tj;
            const float2 x = (i < D && j < D) ? Mg[(int64_t)i * D + j] : make_float2(0.f, 0.f);
            TB_SET(a, b, x);
        }
    const float corner = Mg[(int64_t)D * D + D].x;
    // cpart rows of the last block are never written: keep them zero-free by construction (guarded reads)
    __syncthreads();
    TbPhases<NA, 0>::run(m, dg, sh, D, corner, Mg, dcol, ecol);
    {
        const int k = D - 1, ka = k >> 5;
        if (ti == (k & 31) && tj == (k & 31)) {
#pragma unroll
            for (int a = 0; a < NA; ++a)
                if (a == ka) {
                    dcol[D] = dg[a * TB_THREADS + tid].x;
                    ecol[D] = 0.f;
                }
        }
    }
}

// P = Q^H, the 32-row blocks {g, g + 2, g + 4, g + 6} handled by workgroup g (interleaved: the work per row grows
// with the row index):  P <- P (I - conj(tau_u) v_u v_u^H), u = D-1 .. 0.
// Two reflectors per pass (u, then u - 1):  P (I - c1 v1 v1^H)(I - c2 v2 v2^H) with c = conj(tau):
//   y1 = P v1, y2 = P v2 in ONE sweep over the registers, P' v2 = y2 - c1 y1 (v1^H v2), then one rank-2 update.
// Same flops as two single steps, half the serialized reduce -> broadcast -> update round trips.
template <int NB, int B0>
__device__ __forceinline__ void ub_step2(float2 (&p)[4][NB], int u, int D, const float2 *__restrict__ Mg, int g) {
    const int tj = threadIdx.x & 31;
    // (a reflector with tau = 0 -- H = I -- needs no special case: its t below is zero.)  The last pass of a block
    // with an odd number of reflectors has no second one: tau2 = 0.
    const bool has2 = u - 1 >= 32 * B0;
    const int u2 = has2 ? u - 1 : u;
    const float2 tau1 = Mg[(int64_t)D * D + u];
    const float2 tau2 = has2 ? Mg[(int64_t)D * D + u2] : make_float2(0.f, 0.f);
    if (tau1.x == 0.f && tau1.y == 0.f && tau2.x == 0.f && tau2.y == 0.f) return;   // uniform
    v2 va[NB], vb[NB];
#pragma unroll
    for (int b = B0; b < NB; ++b) {
        const int j = 32 * b + tj;
        va[b] = (j < D) ? b_tov2(Mg[(int64_t)u * D + j]) : v2{0.f, 0.f};
        vb[b] = (j < D) ? b_tov2(Mg[(int64_t)u2 * D + j]) : v2{0.f, 0.f};
    }
    // s12 = v1^H v2 (every lane of a 32-lane row ends up with the full sum)
    v2 s12 = {0.f, 0.f};
#pragma unroll
    for (int b = B0; b < NB; ++b) s12 = b_cmacc(s12, vb[b], b_rotc(vb[b]), va[b]);
    s12.x = row32_sum(s12.x);
    s12.y = row32_sum(s12.y);
    const v2 c1 = v2{tau1.x, -tau1.y}, c2 = v2{tau2.x, -tau2.y};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        // row r of P is still e_r while u > r (v_u vanishes above its unit entry): row blocks above the
        // reflector's block have nothing to do (uniform per workgroup)
        if (2 * a + g < B0) continue;
        v2 y1 = {0.f, 0.f}, y2 = {0.f, 0.f};
#pragma unroll
        for (int b = B0; b < NB - 1; ++b) {
            y1 = b_cmac_sel(y1, b_tov2(p[a][b]), va[b]);
            y2 = b_cmac_sel(y2, b_tov2(p[a][b]), vb[b]);
        }
        // last term by the compiler: the DPP reduction reads y1 / y2 next
        y1 = b_cmac(y1, b_tov2(p[a][NB - 1]), va[NB - 1], b_rot(va[NB - 1]));
        y2 = b_cmac(y2, b_tov2(p[a][NB - 1]), vb[NB - 1], b_rot(vb[NB - 1]));
        y1.x = row32_sum(y1.x);
        y1.y = row32_sum(y1.y);
        y2.x = row32_sum(y2.x);
        y2.y = row32_sum(y2.y);
        const v2 t1 = b_tov2(cmul(b_tof2(c1), b_tof2(y1)));
        const float2 cor = cmul(b_tof2(t1), b_tof2(s12));
        const v2 t2 = b_tov2(cmul(b_tof2(c2), make_float2(y2.x - cor.x, y2.y - cor.y)));
#pragma unroll
        for (int b = B0; b < NB; ++b) {
            v2 x = b_tov2(p[a][b]);
            x = b_cmsubc_sel(x, t1, va[b]);
            x = b_cmsubc_sel(x, t2, vb[b]);
            p[a][b] = b_tof2(x);
        }
    }
}

template <int NB, int B0>
struct UbPhases {
    static __device__ __forceinline__ void run(float2 (&p)[4][NB], int D, const float2 *Mg, int g) {
        if constexpr (B0 + 1 < NB) UbPhases<NB, B0 + 1>::run(p, D, Mg, g);
        const int hi = min(32 * (B0 + 1), D);
        for (int u = hi - 1; u >= 32 * B0; u -= 2) ub_step2<NB, B0>(p, u, D, Mg, g);
    }
};

template <int NB>
__global__ __launch_bounds__(TB_THREADS) void ungtr_big_kernel(int D, const float2 *__restrict__ Mbuf,
                                                               float *__restrict__ QV) {
    const int tid = threadIdx.x;
    const int tj = tid & 31, ti = tid >> 5;
    const int64_t bm = blockIdx.x;
    const int g = blockIdx.y;
    const int n = D + 1;
    const float2 *Mg = Mbuf + bm * ((int64_t)D * D + D + 1);
    float2 p[4][NB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) p[a][b] = make_float2((32 * (2 * a + g) + ti == 32 * b + tj) ? 1.f : 0.f, 0.f);
    UbPhases<NB, 0>::run(p, D, Mg, g);
    float *q = QV + bm * ((int64_t)n * 2 * D);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c = 32 * (2 * a + g) + ti, rho = 32 * b + tj;
            if (c < D && rho < D) {
                q[(int64_t)c * 2 * D + rho] = p[a][b].x;
                q[(int64_t)c * 2 * D + D + rho] = -p[a][b].y;
            }
        }
}

// D = 256: the panel-blocked kernel of tridiag_panel.hip (trailing updates on the matrix cores) produces the same
// (d, e, reflector rows, taus); ADMMNET_TRIDIAG_BIG=sweep keeps the per-reflector register sweep below for A/B runs.
static bool use_panel(int D) {
    static const bool sweep = getenv("ADMMNET_TRIDIAG_BIG") && !strcmp(getenv("ADMMNET_TRIDIAG_BIG"), "sweep");
    return !sweep && tridiag_panel_supported(D);
}

template <int NA>
static int launch_tb(int D, int64_t nb, const Ws &ws, hipStream_t st) {
    if (use_panel(D)) {
        int rc = launch_tridiag_panel(D, nb, ws, st);
        if (rc) return rc;
    } else {
        const size_t lds = sizeof(TbShared<NA>) + sizeof(float2) * (2 * NA - 1) * TB_THREADS;
        ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tridiag_big_kernel<NA>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(tridiag_big_kernel<NA>, dim3((unsigned)nb), dim3(TB_THREADS), lds, st, D, ws.Mbuf, ws.dT,
                           ws.eT);
        ADMM_HIP(hipGetLastError());
    }
    if (use_panel(D) && use_wy_back(D)) return ADMMNET_OK;   // the back-transform applies the block reflectors itself
    hipLaunchKernelGGL(ungtr_big_kernel<NA>, dim3((unsigned)nb, (unsigned)((D + 127) / 128)), dim3(TB_THREADS), 0, st,
                       D, ws.Mbuf, ws.QV);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_tridiag_big(int D, int64_t nb, const Ws &ws, hipStream_t st) {
    const int na = (D + 31) / 32;
    switch (na) {
        case 5: return launch_tb<5>(D, nb, ws, st);
        case 6: return launch_tb<6>(D, nb, ws, st);
        case 7: return launch_tb<7>(D, nb, ws, st);
        case 8: return launch_tb<8>(D, nb, ws, st);
        default:
            set_error("tridiag_big: D=%d unsupported (129..256)", D);
            return ADMMNET_E_ARG;
    }
}

}  // namespace admmnet
