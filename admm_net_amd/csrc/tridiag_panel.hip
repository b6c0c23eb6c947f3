// tridiag_panel.hip -- K1 for D = 256 (BASELINE cfg 3/4/5, n = 257): panel-blocked Householder
// tridiagonalisation with the trailing updates on the matrix cores.
//
// First half of torch.linalg.eigh at /root/reference/admm_net.py:303, LAPACK chetrd / clatrd (lower) organisation
// (host model: tests/host_model/latrd_model.py):
//   * panels of NB = 16 reflectors.  Inside a panel each reflector costs ONE Hermitian matrix-vector product with
//     the panel-start matrix plus skinny corrections with the panel's (V, W) columns; the rank-2 updates of the
//     trailing matrix -- half of all the flops -- are deferred to one rank-32 product per panel,
//         M[16(p+1):, 16(p+1):] -= V W^H + W V^H,
//     issued as v_mfma_f32_16x16x4_f32 straight into the resident tiles.
//   * the Hermitian half of M lives in REGISTERS in the matrix cores' own accumulator layout: 136 lower-triangular
//     16 x 16 tiles (diagonal tiles in full), tile (I, J) = 8 VGPRs of one wave (lane -> column l & 15, registers ->
//     rows 4 (l >> 4) + q).  512 threads = 8 waves, wave w owns the block rows {w, 15 - w}: 17 tiles = 136 VGPRs.
//     So the MFMA update needs no data movement at all, and the matrix-vector product reads every stored element
//     once from registers for both triangle halves:
//         y_I += T_IJ v_J      (row form: per-lane products, ONE 16-lane DPP reduction per block row)
//         y_J += T_IJ^H v_I    (column form: per-lane sums over the 4 rows a lane holds + a 4-group reduction)
//   * V, W panels (2 x 36 KB), the column / reflector vectors and the partial sums live in LDS.
// Four barriers per reflector:
//     F+B  w of the previous reflector; the column (looked ahead, see below) + that reflector's term, norm partials
//     C    reflector scalars (branch-free, so the panel dots W^H v, V^H v -- taken over x below the unit row, which is
//          known one phase before v -- run in the shadow of their dependent chain)
//     D    y = M v from the register tiles; the unit row's term of the panel dots
//     E    row waves: y - V g1 - W g2, p = tau y, p^H v;  the waves that hold no rows meanwhile bring the NEXT column up
//          to date with the reflectors 0 .. j-1 (look-ahead) and append column j of the block reflector's T factor
// instead of the rank-2 update's register sweep per reflector of tridiag_big.hip; outputs (d, e, reflector rows, taus)
// in that kernel's format, so ungtr_big_kernel / the D&C / the back-transform are unchanged consumers.
//   * Every phase is a latency chain (LDS round trip -> ~100 dependent instructions -> lane reduction -> LDS -> barrier),
//     ~9 k cycles per reflector however small the trailing matrix is, and the register-resident half of the 256 x 256
//     matrix admits one workgroup per CU.  So the reduction runs in three STAGES (one kernel template): panels 0..7 on
//     8 waves (one matrix per CU), panels 8..11 on the trailing 128 x 128 matrix with 4 waves (69 KB of LDS: two
//     matrices per CU), panels 12..15 on the trailing 64 x 64 with 2 waves (four per CU); the trailing tiles travel
//     between the stages in the accumulator layout itself (Ws::Tail, 74 KB per matrix).  ADMMNET_PN_SPLIT=0 | 8 select
//     one or two stages for A/B runs.
//
// Index conventions (arrow-first order, as every tridiagonalisation here): F = [[corner, a^H], [a, M]], M is D x D.
// Reflector u (0 <= u < D) has its unit position at M-row u, annihilates F column u below it (for u >= 1 that is
// column c = u - 1 of M, for u = 0 the arrow a), d[u] = F[u][u], e[u] = beta_u.  The prologue reflector u = 0 is
// handled as column 15 of a virtual panel -1 whose other columns are zero.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "lane_reduce.h"

namespace admmnet {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int PN_D = 256;
constexpr int PN_PITCH = 18;       // float2 per panel row: 16 columns + 2 pad (144 B rows: 16-byte aligned, spread over banks)
constexpr int PN_TAIL_TILES = 36;  // lower block triangle of the 128 x 128 trailing matrix handed from stage to stage

// One STAGE of the reduction works on the trailing DL x DL matrix (DL = 16 NT rows, M-rows R0 = D - DL ...) with NT / 2
// waves.  The first stage (NT = 16, HEAD) starts from the image, later ones from the tile set the previous stage left.
template <int NT>
struct PnShared {
    static constexpr int DL = 16 * NT, NW = NT / 2;
    float2 Vp[DL][PN_PITCH];       // panel reflectors (unnormalised), row r = local M-row
    float2 Wp[DL][PN_PITCH];       // panel w vectors
    float2 Ap[DL][PN_PITCH];       // block column p of the panel-start matrix (the 16 columns the panel reduces)
    float2 colbuf[DL];             // the arrow: column of the prologue reflector
    float2 vbuf[DL];               // current reflector (zero above its unit position)
    float2 xbuf[DL];               // ... with its unit entry zeroed (known one phase earlier: the panel dots start from it)
    float2 hu;                     // the unit entry
    float2 xnext[DL];              // the NEXT column, brought up to date with all panel reflectors but the current one
    float2 yrow[DL];               // row-form part of M v (written by the owner wave of each block row)
    float2 ycol[NW][DL];           // column-form partials per wave
    float2 g[32];                  // g[jj] = W_jj^H v, g[16 + jj] = V_jj^H v
    float2 red2[8];
    float2 pu;                     // p[u] of the current reflector
    float2 Gp[16][16];             // Gp[k][i] = V_k^H v_i (k < i): strict upper triangle of the panel's Gram matrix
    float2 Tl[16][16];             // T factor of the panel's block reflector, built one column per reflector
    int skip;                      // the current reflector is the identity
    float dbuf[DL + 4], ebuf[DL + 4];       // d, e and the taus are gathered here and written out once: a global store on
    float2 taubuf[DL];                      // the per-reflector path makes the next barrier wait for its completion
    float red[8];
    float2 alpha;
};

// x + (x of the neighbouring lane l ^ 1): DPP quad_perm [1, 0, 3, 2]
__device__ __forceinline__ float pn_pair_sum(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float pn_wave_sum(float x) { return pn_group_sum(pn_row16_sum(x)); }

__device__ __forceinline__ float2 pn_fma_c(float2 acc, float2 a, float2 b) {      // acc + a b
    acc.x = fmaf(a.x, b.x, fmaf(-a.y, b.y, acc.x));
    acc.y = fmaf(a.x, b.y, fmaf(a.y, b.x, acc.y));
    return acc;
}
__device__ __forceinline__ float2 pn_fms_cc(float2 acc, float2 a, float2 b) {    // acc - a conj(b)
    acc.x = fmaf(-a.x, b.x, fmaf(-a.y, b.y, acc.x));
    acc.y = fmaf(a.x, b.y, fmaf(-a.y, b.x, acc.y));
    return acc;
}
__device__ __forceinline__ float2 pn_fms_c(float2 acc, float2 a, float2 b) {     // acc - a b
    acc.x = fmaf(-a.x, b.x, fmaf(a.y, b.y, acc.x));
    acc.y = fmaf(-a.x, b.y, fmaf(-a.y, b.x, acc.y));
    return acc;
}

// householder_c (eig_core.h) without branches, so that the scheduler can run its dependent chain in the shadow of other
// work of the same basic block: same arithmetic in the normal range (q2 rsqrt(q2) with one Newton step), power-of-two
// pre / post scaling instead of the exact-sqrt fallback outside it, the H = I case by selects at the end.
__device__ __forceinline__ void pn_householder(float ar, float ai, float xnorm2, float &beta, float &tr, float &ti,
                                               float &sr, float &si) {
    const bool ident = (xnorm2 == 0.f && ai == 0.f);
    const float q2 = ar * ar + ai * ai + xnorm2;
    const float sc = (q2 < 1e-30f) ? 0x1p+64f : ((q2 > 1e30f) ? 0x1p-64f : 1.0f);
    const float ps = (q2 < 1e-30f) ? 0x1p-32f : ((q2 > 1e30f) ? 0x1p+32f : 1.0f);
    const float q2s = ident ? 1.0f : q2 * sc;
    const float nrm = q2s * rsqrt_nr1(q2s) * ps;
    const float b = -sign_of(nrm, ar);
    const float ib = recip_nr(b);
    const float dr = ar - b, di = ai;
    const float iden = recip_nr(dr * dr + di * di);
    beta = ident ? ar : b;
    tr = ident ? 0.f : (b - ar) * ib;
    ti = ident ? 0.f : -ai * ib;
    sr = ident ? 0.f : dr * iden;
    si = ident ? 0.f : -di * iden;
}

__device__ __forceinline__ v2f pn_lo(f32x4 a) { return v2f{a.x, a.y}; }
__device__ __forceinline__ v2f pn_hi(f32x4 a) { return v2f{a.z, a.w}; }
// one 16 x 16 tile (tre, tim: rows 4 g + q of column c16) in the matrix-vector product, packed FMAs over row pairs:
//   ROW form   P[q] += T[q][c] vJ[c]            (Pr, Pi: rows 01 | 23; a diagonal tile is stored at half its value, so
//                                                 that its two forms add up to T v: see the load)
//   COL form   C    += sum_q conj(T[q][c]) vI[q]  (vIr, vIi: the four row entries of v, planar pairs; Cr, Ci pairs)
__device__ __forceinline__ void pn_tile_mv(f32x4 tre, f32x4 tim, float2 vJ, v2f vIr01, v2f vIr23, v2f vIi01,
                                           v2f vIi23, v2f &Pr01, v2f &Pr23, v2f &Pi01, v2f &Pi23, v2f &Cr, v2f &Ci) {
    const v2f r01 = pn_lo(tre), r23 = pn_hi(tre), i01 = pn_lo(tim), i23 = pn_hi(tim);
    const v2f jx = v2f{vJ.x, vJ.x}, jy = v2f{vJ.y, vJ.y};
    Pr01 = __builtin_elementwise_fma(r01, jx, Pr01);
    Pi01 = __builtin_elementwise_fma(r01, jy, Pi01);
    Pr23 = __builtin_elementwise_fma(r23, jx, Pr23);
    Pi23 = __builtin_elementwise_fma(r23, jy, Pi23);
    Pr01 = __builtin_elementwise_fma(-i01, jy, Pr01);
    Pi01 = __builtin_elementwise_fma(i01, jx, Pi01);
    Pr23 = __builtin_elementwise_fma(-i23, jy, Pr23);
    Pi23 = __builtin_elementwise_fma(i23, jx, Pi23);
    // conj(T) v = (Tr vr + Ti vi) + i (Tr vi - Ti vr)
    Cr = __builtin_elementwise_fma(r01, vIr01, Cr);
    Ci = __builtin_elementwise_fma(r01, vIi01, Ci);
    Cr = __builtin_elementwise_fma(i01, vIi01, Cr);
    Ci = __builtin_elementwise_fma(-i01, vIr01, Ci);
    Cr = __builtin_elementwise_fma(r23, vIr23, Cr);
    Ci = __builtin_elementwise_fma(r23, vIi23, Ci);
    Cr = __builtin_elementwise_fma(i23, vIi23, Cr);
    Ci = __builtin_elementwise_fma(-i23, vIr23, Ci);
}

// Four per-lane partial vectors (one per block column, each to be summed over the four 16-lane rows of the wave)
// reduced together: 3 swaps + 3 adds instead of 8 + 8.  Result: row 0 holds the total of x0, row 1 of x2, row 2 of x1,
// row 3 of x3 (v_permlane32_swap: upper half of the first operand <-> lower half of the second; v_permlane16_swap: odd
// rows of the first <-> even rows of the second).
__device__ __forceinline__ float pn_quad_group_sum(float x0, float x1, float x2, float x3) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x2), "+v"(x3));
    float a = x0 + x1, b = x2 + x3;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// ... two such quadruples (re / im) in the same two asm blocks
__device__ __forceinline__ void pn_quad_group_sum2(const float (&x)[4], const float (&y)[4], float &tx, float &ty) {
    float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3], y0 = y[0], y1 = y[1], y2 = y[2], y3 = y[3];
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\t"
                 "v_permlane32_swap_b32 %4, %5\n\tv_permlane32_swap_b32 %6, %7\n\ts_nop 1"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
    float a = x0 + x1, b = x2 + x3, c = y0 + y1, d = y2 + y3;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    tx = a + b;
    ty = c + d;
}

// One step of the transposing butterfly sum: this lane keeps `lo` (hi lanes: `hi`), its DPP partner sends the value it
// does not keep; returns kept + received.
template <int CTRL>
__device__ __forceinline__ float pn_keep_add(bool hi_lane, float lo, float hi) {
    const float keep = hi_lane ? hi : lo, send = hi_lane ? lo : hi;
    return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), CTRL, 0xF, 0xF, false));
}

// slot s of wave w: tile (IB, s) for s <= IB, tile (IA, NT - s) otherwise   (IA = w, IB = NT - 1 - w)
#define PN_SLOT_IJ(s, I, J)            \
    int I, J;                          \
    if ((s) <= IB) { I = IB; J = (s); } \
    else { I = IA; J = NT - (s); }

// tile t(I, J) = I (I + 1) / 2 + J of the hand-over set, register q of lane l at ((t * 4 + q) * 64 + l): the accumulator
// layout itself, so both sides move whole 512-byte wave rows
__device__ __forceinline__ int64_t pn_tail_at(int I, int J, int q, int lane) {
    return ((int64_t)(I * (I + 1) / 2 + J) * 4 + q) * 64 + lane;
}

template <int NT, bool HEAD, bool TIMING>
__global__ __launch_bounds__(32 * NT, 2) void tridiag_panel_kernel(float2 *__restrict__ Mbuf, float *__restrict__ dT,
                                                                   float *__restrict__ eT, float2 *__restrict__ Tfac,
                                                                   float2 *__restrict__ Tail, int pstop, int zfill,
                                                                   unsigned long long *__restrict__ tdbg,
                                                                   const int *__restrict__ skip) {
    static_assert(NT % 4 == 0 && NT <= 16 && (HEAD == (NT == 16)), "stage geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Shared = PnShared<NT>;
    Shared &sh = *reinterpret_cast<Shared *>(smem);
    constexpr int D = PN_D, n = D + 1;
    constexpr int DL = 16 * NT, NW = NT / 2, RW = DL / 64, THREADS = 64 * NW;   // RW row waves, NW tile waves
    constexpr int R0 = D - DL, P0 = 16 - NT;                                    // first M-row / panel of this stage
    constexpr bool ALLW = (NT == 16);   // every wave runs the reflector scalars (see phase C)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: tile ownership tests below are uniform branches
    const int c16_0 = lane & 15, g_0 = lane >> 4;
    int c16 = c16_0, g = g_0;
    int IA = wave, IB = NT - 1 - wave;
    const int64_t bm = blockIdx.x;
    if (skip && skip[bm] == 0) return;   // (uniform) this matrix' G is already there: spectral.hip
    float2 *Mg = Mbuf + bm * ((int64_t)D * D + D + 1);
    float *dcol = dT + bm * n, *ecol = eT + bm * n;
    float2 *tail = Tail + bm * (PN_TAIL_TILES * 256);

    // ---- load the Hermitian half into the accumulator layout
    f32x4 tr[NT + 1], ti[NT + 1];
#pragma unroll
    for (int s = 0; s < NT + 1; ++s) {
        PN_SLOT_IJ(s, I, J)
        if constexpr (HEAD) {
            const float2 *src = Mg + (int64_t)(16 * I + 4 * g) * D + 16 * J + c16;
            const float2 e0 = src[0], e1 = src[D], e2 = src[2 * D], e3 = src[3 * D];
            // DIAGONAL tiles are kept at HALF their value: the matrix-vector phase then runs both of its forms on every
            // tile without masks (T v / 2 from the row form + T^H v / 2 from the column form), 3 VALU instructions per
            // tile and reflector less; the trailing update adds half of its term there, and the two readers of true
            // values (the panel-column copy, the next stage's first panel) double it back.  Powers of two: exact.
            const float dsc = (I == J) ? 0.5f : 1.0f;
            tr[s] = f32x4{e0.x, e1.x, e2.x, e3.x} * dsc;
            ti[s] = f32x4{e0.y, e1.y, e2.y, e3.y} * dsc;
        } else {
            const float2 e0 = tail[pn_tail_at(I, J, 0, lane)], e1 = tail[pn_tail_at(I, J, 1, lane)],
                         e2 = tail[pn_tail_at(I, J, 2, lane)], e3 = tail[pn_tail_at(I, J, 3, lane)];
            tr[s] = f32x4{e0.x, e1.x, e2.x, e3.x};
            ti[s] = f32x4{e0.y, e1.y, e2.y, e3.y};
            if (J == 0) {   // (uniform) the first panel's columns (the hand-over set holds the diagonal tiles halved)
                const float cs = (I == 0) ? 2.0f : 1.0f;
                float2 *dst = &sh.Ap[16 * I + 4 * g][c16];
                dst[0] = make_float2(e0.x * cs, e0.y * cs);
                dst[PN_PITCH] = make_float2(e1.x * cs, e1.y * cs);
                dst[2 * PN_PITCH] = make_float2(e2.x * cs, e2.y * cs);
                dst[3 * PN_PITCH] = make_float2(e3.x * cs, e3.y * cs);
            }
        }
    }
    float corner = 0.f;
    if constexpr (HEAD) corner = Mg[(int64_t)D * D + D].x;
    for (int i = tid; i < DL * PN_PITCH; i += THREADS) {
        (&sh.Vp[0][0])[i] = make_float2(0.f, 0.f);
        (&sh.Wp[0][0])[i] = make_float2(0.f, 0.f);
    }
    for (int i = tid; i < 256; i += THREADS) {
        (&sh.Gp[0][0])[i] = make_float2(0.f, 0.f);
        (&sh.Tl[0][0])[i] = make_float2(0.f, 0.f);
    }
    if constexpr (HEAD)
        if (tid < D) sh.colbuf[tid] = Mg[(int64_t)D * D + tid];   // the arrow: column of the prologue reflector
    // (no barrier yet: the first one of the step loop orders these stores before any reader, and every global
    //  store below comes after at least one barrier, i.e. after every wave's matrix loads have been issued AND
    //  their results consumed into registers by the slot loop above)
    __syncthreads();

    const int r = tid;                 // row owned by the threads of waves 0..3 (one wave per SIMD; splitting the rows
                                       // over lane pairs of all eight waves was measured slower: same instruction total)
    // carried from one reflector to the next (same panel): its v and w entries of this thread's row, and -- known to
    // every thread -- the two entries at its unit row, (hu, wu) = (v, w)[u].  With them the next column is brought up to
    // date without waiting for the panel stores of the previous step (no barrier between the steps).
    float2 xcol = make_float2(0.f, 0.f), vreg = make_float2(0.f, 0.f), wreg = make_float2(0.f, 0.f),
           preg = make_float2(0.f, 0.f), hu = make_float2(0.f, 0.f), wu = make_float2(0.f, 0.f);
    unsigned long long tmark = TIMING ? __builtin_amdgcn_s_memtime() : 0ull, tacc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto mark = [&](int id) {
        if constexpr (TIMING) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            tacc[id] += t - tmark;
            tmark = t;
        }
    };

    // Look-ahead, on the waves that hold no rows (idle while the row waves assemble y): column j + 1 of the panel with the
    // corrections of the reflectors 0 .. j - 1, so that the next step's column phase only adds reflector j's term (from
    // registers).  x = A[:, j+1] - sum_k V[:, k] conj(W[c+1][k]) + W[:, k] conj(V[c+1][k]), rows >= c + 1.
    auto lookahead = [&](int p, int j) {
        if (tid >= DL && p >= 0 && j + 1 < 16) {   // (uniform per wave)
            const int r2 = tid - DL, c2 = 16 * p + j + 1;
            float2 x = sh.Ap[r2][j + 1];
            if (r2 >= c2) {
                v2f acc = v2f{0.f, 0.f}, acc2 = v2f{0.f, 0.f};
                for (int j0 = 0; j0 < j; j0 += 4) {   // four columns per pass: all 16 loads in flight together
                    float2 vr[4], wr[4], vc[4], wc[4];    // (columns >= j of the panel are zero: no masks)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        vr[q] = sh.Vp[r2][j0 + q];
                        wr[q] = sh.Wp[r2][j0 + q];
                        vc[q] = sh.Vp[c2][j0 + q];
                        wc[q] = sh.Wp[c2][j0 + q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc = pk_cfma_conj(acc, pk2(wc[q]), pk2(vr[q]));     // v conj(w_c)
                        acc2 = pk_cfma_conj(acc2, pk2(vc[q]), pk2(wr[q]));   // w conj(v_c)
                    }
                }
                x.x -= acc.x + acc2.x;
                x.y -= acc.y + acc2.y;
            }
            sh.xnext[r2] = x;
        }
    };

    for (int p = HEAD ? -1 : 0; p < NT; ++p) {
        if (p >= 0) {   // the panel's columns start from zero: the skinny sums below run over whole groups of four columns
            for (int i = tid; i < DL * 16; i += THREADS) {
                sh.Vp[i >> 4][i & 15] = make_float2(0.f, 0.f);
                sh.Wp[i >> 4][i & 15] = make_float2(0.f, 0.f);
            }
            if (tid < 32) sh.g[tid] = make_float2(0.f, 0.f);
        }
        for (int j = (p < 0) ? 15 : 0; j < 16; ++j) {
            const int c = 16 * p + j, u = c + 1;
            // the lane coordinates are re-derived per step from an opaque copy: otherwise every LDS address of the
            // unrolled tile loops below is hoisted out of the reflector loop and kept alive (~90 VGPRs, spills)
            c16 = c16_0;
            g = g_0;
            asm volatile("" : "+v"(c16), "+v"(g));
            {   // the same for the wave's block rows: the ownership tests are re-derived (scalar compares) per step
                int wv = wave;   // instead of ~100 precomputed conditions parked in spill lanes
                asm volatile("" : "+s"(wv));
                IA = wv;
                IB = NT - 1 - wv;
            }
            // ---- B: bring the column up to date with the panel's earlier reflectors; d, alpha, |x|^2
            if (tid < DL) {
                float2 x = (p >= 0) ? (j > 0 ? sh.xnext[r] : sh.Ap[r][j]) : sh.colbuf[r];
                if (r >= c && p >= 0 && j > 0) {   // reflector j - 1 from registers: (V, W)[c][j - 1] = (hu, wu)
                    v2f a = pk_cfma_conj(v2f{0.f, 0.f}, pk2(wu), pk2(vreg));
                    a = pk_cfma_conj(a, pk2(hu), pk2(wreg));
                    x.x -= a.x;
                    x.y -= a.y;
                }
                xcol = x;
                {   // everything of the reflector but its unit entry is known here already
                    const float2 xm = (r > u) ? x : make_float2(0.f, 0.f);
                    sh.vbuf[r] = xm;
                    sh.xbuf[r] = xm;
                }
                if (r == c) sh.dbuf[u] = x.x;
                if (HEAD && p < 0 && r == 0) sh.dbuf[0] = corner;
                if (r == u) sh.alpha = x;
                float pn = (r > u) ? (x.x * x.x + x.y * x.y) : 0.f;
                pn = pn_wave_sum(pn);
                if (lane == 0) sh.red[wave] = pn;
            }
            if (u >= DL) break;        // c = DL - 1: only d[D] was due (uniform)
            mark(0);
            __syncthreads();   // (B2)
            mark(8);
            // ---- C: the reflector scalars -- a ~25-deep dependent chain (rsq, rcp, Newton steps) -- and, in the same
            //      instruction stream so that they fill its latency slots, the panel dots W^H v, V^H v taken over
            //      the rows below the unit position (xbuf); the unit row's term  conj(X[u][jj]) hu  is added in phase D.
            //      dot id q = q0 + 4 wave + g : q < 16 -> W_q^H v, else V_{q-16}^H v; the 16 lanes of a group stride the
            //      rows (x is zero above, so all row blocks are summed: fixed trip count, all loads in flight together)
            float2 tau = make_float2(0.f, 0.f);
            {
                float2 dacc[(32 + 4 * NW - 1) / (4 * NW)];
                auto dots = [&]() {
#pragma unroll
                    for (int q0 = 0, ps = 0; q0 < 32; q0 += 4 * NW, ++ps) {   // (one pass with eight waves)
                        const int q = (q0 + 4 * wave + g) & 31, jj = q & 15;
                        v2f acc = v2f{0.f, 0.f}, acc2 = v2f{0.f, 0.f};
                        // 8-wave stage: only the dots of the panel's existing columns touch LDS (both panels + x are 64 KB
                        // per reflector if every lane group reads: -2 %); the smaller stages lose more from the branch
                        // between the reflector chain and these loads than they gain (+7 %)
                        if (!ALLW || (p >= 0 && jj < j)) {
                            const float2(*X)[PN_PITCH] = (q < 16) ? sh.Wp : sh.Vp;
#pragma unroll
                            for (int i = 0; i < NT; i += 2) {
                                acc = pk_cfma_conj(acc, pk2(X[16 * i + c16][jj]), pk2(sh.xbuf[16 * i + c16]));
                                acc2 = pk_cfma_conj(acc2, pk2(X[16 * i + 16 + c16][jj]), pk2(sh.xbuf[16 * i + 16 + c16]));
                            }
                        }
                        dacc[ps] = make_float2(acc.x + acc2.x, acc.y + acc2.y);
                    }
                };
                // (the scalars are needed by the row waves only, and the others share their SIMDs; measured: skipping them
                //  there gains 6 % in the 4-wave stage and LOSES 4 % in the 8-wave stage, which keeps the common path)
                if (ALLW || wave < RW) {   // (uniform)
                    float xn2;
                    if constexpr (RW == 4) xn2 = (sh.red[0] + sh.red[1]) + (sh.red[2] + sh.red[3]);
                    else if constexpr (RW == 2) xn2 = sh.red[0] + sh.red[1];
                    else if constexpr (RW == 1) xn2 = sh.red[0];
                    else xn2 = sh.red[0] + sh.red[1] + sh.red[2];
                    const float2 alpha = sh.alpha;
                    float beta, tre, tim, sr, si;
                    pn_householder(alpha.x, alpha.y, xn2, beta, tre, tim, sr, si);
                    dots();
                    const float g2 = sr * sr + si * si;
                    tau = make_float2(tre * g2, tim * g2);   // unnormalised reflector: H = I - tau v v^H, v = (alpha - beta, x)
                    hu = make_float2(alpha.x - beta, alpha.y);
                    if (tid < DL) {
                        vreg = (r == u) ? hu : (r > u ? xcol : make_float2(0.f, 0.f));
                        if (r == u) sh.vbuf[r] = hu;
                    }
                    if (tid == 0) {
                        sh.ebuf[u] = beta;
                        sh.taubuf[u] = tau;
                        sh.hu = hu;
                        sh.skip = (tre == 0.f && tim == 0.f) ? 1 : 0;
                    }
                } else {
                    dots();
                }
#pragma unroll
                for (int q0 = 0, ps = 0; q0 < 32; q0 += 4 * NW, ++ps) {
                    const int qq = q0 + 4 * wave + g, jj = qq & 15;
                    const float sx = pn_row16_sum(dacc[ps].x), sy = pn_row16_sum(dacc[ps].y);
                    if (c16 == 0 && qq < 32 && jj < j && p >= 0) sh.g[qq] = make_float2(sx, sy);
                }
            }
            mark(1);
            __syncthreads();   // (B3)
            mark(9);
            if (__builtin_amdgcn_readfirstlane(sh.skip)) {        // H = I (uniform): v = 0, w = 0
                hu = make_float2(0.f, 0.f);
                wu = make_float2(0.f, 0.f);
                wreg = make_float2(0.f, 0.f);
                if (tid < DL) {
                    sh.Vp[r][j] = make_float2(0.f, 0.f);
                    sh.Wp[r][j] = make_float2(0.f, 0.f);
                }
                if (tid < 16) sh.Tl[tid][j] = make_float2(0.f, 0.f);
                lookahead(p, j);
                __syncthreads();
                continue;
            }
            // ---- D: y = M v with the resident half (+ the panel dots W^H v, V^H v)
            {
                const int J0 = u >> 4;
                v2f Ar01, Ar23, Ai01, Ai23, Br01, Br23, Bi01, Bi23;   // v at the rows of block rows IA / IB (planar pairs)
                {
                    const float2 *va = &sh.vbuf[16 * IA + 4 * g], *vb = &sh.vbuf[16 * IB + 4 * g];
                    const float2 a0 = va[0], a1 = va[1], a2 = va[2], a3 = va[3];
                    const float2 b0 = vb[0], b1 = vb[1], b2 = vb[2], b3 = vb[3];
                    Ar01 = v2f{a0.x, a1.x}; Ar23 = v2f{a2.x, a3.x}; Ai01 = v2f{a0.y, a1.y}; Ai23 = v2f{a2.y, a3.y};
                    Br01 = v2f{b0.x, b1.x}; Br23 = v2f{b2.x, b3.x}; Bi01 = v2f{b0.y, b1.y}; Bi23 = v2f{b2.y, b3.y};
                }
                const v2f z2 = v2f{0.f, 0.f};
                v2f PAr01 = z2, PAr23 = z2, PAi01 = z2, PAi23 = z2, PBr01 = z2, PBr23 = z2, PBi01 = z2, PBi23 = z2;
#pragma unroll
                for (int JQ = 0; JQ < NT; JQ += 4) {
                    if (JQ + 3 >= J0 && JQ <= IB) {   // (uniform) four block columns per pass: one joint lane reduction
                        float cx[4], cy[4];
                        float2 vJq[4];   // the pass's four column slices of v, all in flight before the first product
#pragma unroll
                        for (int k = 0; k < 4; ++k) vJq[k] = sh.vbuf[16 * (JQ + k) + c16];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int J = JQ + k;
                            v2f Cr = z2, Ci = z2;
                            if (J >= J0 && J <= IB) {   // (uniform)
                                const float2 vJ = vJq[k];
                                pn_tile_mv(tr[J], ti[J], vJ, Br01, Br23, Bi01, Bi23, PBr01, PBr23, PBi01, PBi23, Cr, Ci);
                                if (J <= IA)
                                    pn_tile_mv(tr[NT - J], ti[NT - J], vJ, Ar01, Ar23, Ai01, Ai23, PAr01, PAr23, PAi01,
                                               PAi23, Cr, Ci);
                            }
                            cx[k] = Cr.x + Cr.y;
                            cy[k] = Ci.x + Ci.y;
                        }
                        float tx, ty;
                        pn_quad_group_sum2(cx, cy, tx, ty);
                        const int Jl = JQ + ((g & 1) << 1) + (g >> 1);   // lane row g holds block column JQ + {0, 2, 1, 3}[g]
                        if (Jl >= J0 && Jl <= IB) sh.ycol[wave][16 * Jl + c16] = make_float2(tx, ty);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // keep each pass's loads inside it (VGPR budget)
                }
                mark(2);
                // Row sums of the 8 (or 16) partial values per lane over the 16 lanes of a row, as a transposing butterfly:
                // at each step a lane keeps one half of its values, hands the other half to its mirror partner and adds
                // what it receives (2 selects + 1 DPP add per pair; mirror over 16 / 8 / 4 / 2 lanes) -- 15 pairs = 45
                // instructions for 16 values instead of 64 dependent DPP adds, and lane c16 ends up with the total of
                // value c16 = 8 (block row IA?) + 2 q + (imaginary?), which it stores itself.
                // (8-wave stage only: -2 % there, +6 % in the 4-wave stage, which keeps the plain DPP sums)
                if constexpr (ALLW) {
                    const bool b3 = c16 & 8, b2 = c16 & 4, b1 = c16 & 2, b0 = c16 & 1;
                    float total;
                    const float vB[8] = {PBr01.x, PBi01.x, PBr01.y, PBi01.y, PBr23.x, PBi23.x, PBr23.y, PBi23.y};
                    if (IA >= J0) {   // (uniform) both block rows live
                        const float vA[8] = {PAr01.x, PAi01.x, PAr01.y, PAi01.y, PAr23.x, PAi23.x, PAr23.y, PAi23.y};
                        float w8[8], w4[4], w2[2];
#pragma unroll
                        for (int m = 0; m < 8; ++m) w8[m] = pn_keep_add<0x140>(b3, vB[m], vA[m]);   // row_mirror
#pragma unroll
                        for (int m = 0; m < 4; ++m) w4[m] = pn_keep_add<0x141>(b2, w8[m], w8[m + 4]);   // row_half_mirror
#pragma unroll
                        for (int m = 0; m < 2; ++m) w2[m] = pn_keep_add<0x1B>(b1, w4[m], w4[m + 2]);   // quad_perm [3,2,1,0]
                        total = pn_keep_add<0xB1>(b0, w2[0], w2[1]);                                 // quad_perm [1,0,3,2]
                        const int q = (c16 >> 1) & 3, I = b3 ? IA : IB;
                        reinterpret_cast<float *>(&sh.yrow[16 * I + 4 * g + q])[c16 & 1] = total;
                    } else if (IB >= J0) {   // (uniform) one block row: value (c16 & 7) on lane pairs {c16, c16 ^ 8}
                        float w4[4], w2[2];
#pragma unroll
                        for (int m = 0; m < 4; ++m) w4[m] = pn_keep_add<0x141>(b2, vB[m], vB[m + 4]);
#pragma unroll
                        for (int m = 0; m < 2; ++m) w2[m] = pn_keep_add<0x1B>(b1, w4[m], w4[m + 2]);
                        total = pn_keep_add<0xB1>(b0, w2[0], w2[1]);
                        total += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, total), 0x128,
                                                                                        0xF, 0xF, false));   // + lane c16 ^ 8 (row_ror:8)
                        const int q = (c16 >> 1) & 3;
                        if (!b3) reinterpret_cast<float *>(&sh.yrow[16 * IB + 4 * g + q])[c16 & 1] = total;
                    }
                } else {
                    if (IB >= J0) {
                        float2 P[4] = {make_float2(PBr01.x, PBi01.x), make_float2(PBr01.y, PBi01.y),
                                       make_float2(PBr23.x, PBi23.x), make_float2(PBr23.y, PBi23.y)};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            P[q].x = pn_row16_sum(P[q].x);
                            P[q].y = pn_row16_sum(P[q].y);
                        }
                        if (c16 == 0) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) sh.yrow[16 * IB + 4 * g + q] = P[q];
                        }
                    }
                    if (IA >= J0) {
                        float2 P[4] = {make_float2(PAr01.x, PAi01.x), make_float2(PAr01.y, PAi01.y),
                                       make_float2(PAr23.x, PAi23.x), make_float2(PAr23.y, PAi23.y)};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            P[q].x = pn_row16_sum(P[q].x);
                            P[q].y = pn_row16_sum(P[q].y);
                        }
                        if (c16 == 0) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) sh.yrow[16 * IA + 4 * g + q] = P[q];
                        }
                    }
                }
                mark(3);
                // the unit row's term of the panel dots (wave NW - 1 has the fewest tiles); Gp keeps V_jj^H v_j for the T factor
                if (wave == NW - 1 && lane < 32 && p >= 0) {
                    const int jj = lane & 15;
                    const float2(*X)[PN_PITCH] = (lane < 16) ? sh.Wp : sh.Vp;
                    const float2 t = cmacc(sh.g[lane], X[u][jj], sh.hu);
                    if (jj < j) {
                        sh.g[lane] = t;
                        if (lane >= 16) sh.Gp[jj][j] = t;
                    }
                }
            }
            mark(4);
            __syncthreads();   // (B4)
            mark(10);
            // ---- E: assemble y, corrections, p = tau y, p^H v  (row waves; the others look ahead, see above)
            lookahead(p, j);
            //      ... and append column j to the panel's T factor (LAPACK clarft, forward /
            //      columnwise):  T[j][j] = tau_j,  T[0:j, j] = -tau_j T[0:j, 0:j] (Y[:, 0:j]^H y_j) -- the Gram entries are the
            //      panel dots left in Gp.  Lane group g = row m of T (four rows per wave and pass), lane c16 = term k.
            if (wave >= RW && p >= 0) {   // (uniform)
                const float2 gam = sh.taubuf[u];
#pragma unroll
                for (int ps = 0; ps < 4 / RW; ++ps) {
                    const int m = 4 * ((wave - RW) + RW * ps) + g, k = c16;
                    const float2 tk = sh.Tl[m][k], gk = sh.Gp[k][j];
                    float2 acc = (k >= m && k < j) ? cmul(tk, gk) : make_float2(0.f, 0.f);
                    acc.x = pn_row16_sum(acc.x);
                    acc.y = pn_row16_sum(acc.y);
                    const float2 t = cmul(gam, acc);
                    if (c16 == 0) sh.Tl[m][j] = (m == j) ? gam : (m < j ? make_float2(-t.x, -t.y) : make_float2(0.f, 0.f));
                }
            }
            if (tid < DL) {
                float2 y = make_float2(0.f, 0.f);
                if (r >= u) {
                    const int J = r >> 4;
                    y = sh.yrow[r];
                    const int wmax = min(NW - 1, NT - 1 - J);
                    float2 t[NW];
#pragma unroll
                    for (int w = 0; w < NW; ++w) t[w] = sh.ycol[w][r];      // (slots above wmax hold stale finite values)
#pragma unroll
                    for (int w = 0; w < NW; ++w) {
                        if (w <= wmax) {
                            y.x += t[w].x;
                            y.y += t[w].y;
                        }
                    }
                    mark(6);
                    if (p >= 0) {
                        v2f acc = v2f{0.f, 0.f}, acc2 = v2f{0.f, 0.f};
                        for (int j0 = 0; j0 < j; j0 += 4) {
                            float2 vr[4], wr[4], g1[4], g2[4];   // (columns >= j of the panel and of g are zero)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                vr[q] = sh.Vp[r][j0 + q];
                                wr[q] = sh.Wp[r][j0 + q];
                                g1[q] = sh.g[j0 + q];
                                g2[q] = sh.g[16 + j0 + q];
                            }
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc = pk_cfma(acc, pk2(vr[q]), pk2(g1[q]));
                                acc2 = pk_cfma(acc2, pk2(wr[q]), pk2(g2[q]));
                            }
                        }
                        y.x -= acc.x + acc2.x;
                        y.y -= acc.y + acc2.y;
                    }
                    {
                        const v2f ty = pk_cfma(v2f{0.f, 0.f}, pk2(tau), pk2(y));
                        y = make_float2(ty.x, ty.y);
                    }
                    mark(12);
                }
                preg = y;
                if (r == u) sh.pu = y;
                float2 dp = cmacc(make_float2(0.f, 0.f), y, vreg);   // conj(p) v
                dp.x = pn_row16_sum(dp.x);
                dp.y = pn_row16_sum(dp.y);
                pn_group_sum2(dp.x, dp.y);
                if (lane == 0) sh.red2[wave] = dp;
            }
            mark(5);
            __syncthreads();   // (B5)
            mark(11);
            // ---- F: w = p - (tau / 2)(p^H v) v ; store the panel column (read again only behind later barriers)
            if (ALLW || wave < RW) {   // (uniform)
                float2 dot = sh.red2[0];
#pragma unroll
                for (int q = 1; q < RW; ++q) {
                    dot.x += sh.red2[q].x;
                    dot.y += sh.red2[q].y;
                }
                const v2f alv = pk_cfma(v2f{0.f, 0.f}, pk2(tau), pk2(dot)) * -0.5f;
                {
                    const v2f t = pk_cfma(pk2(sh.pu), alv, pk2(hu));
                    wu = make_float2(t.x, t.y);
                }
                if (tid < DL) {
                    const v2f wv = pk_cfma(pk2(preg), alv, pk2(vreg));
                    wreg = (r >= u) ? make_float2(wv.x, wv.y) : make_float2(0.f, 0.f);
                    sh.Vp[r][j] = vreg;
                    sh.Wp[r][j] = wreg;
                }
            }
        }
        __syncthreads();
        // the panel's reflectors (rows u = 16 p + 1 + jj of the image, the layout the Q kernel reads) leave in one go
        if (tid < DL) {
            for (int jj = (p < 0) ? 15 : 0; jj < 16; ++jj) {
                const int uu = 16 * p + 1 + jj;
                if (uu < DL) Mg[(int64_t)(R0 + uu) * D + R0 + tid] = sh.Vp[tid][jj];
            }
        }
        if constexpr (!HEAD) {   // the reflectors are zero above this stage's rows: only the explicit-Q consumer
            if (zfill) {         // (ADMMNET_BACK=q) reads there
                for (int i = tid; i < 16 * R0; i += THREADS) {
                    const int uu = 16 * p + 1 + i / R0;
                    if (uu < DL) Mg[(int64_t)(R0 + uu) * D + i % R0] = make_float2(0.f, 0.f);
                }
            }
        }
        // ... and the T factor of the panel's block reflector  H_u0 H_u0+1 ... = I - Y T Y^H  (built column by column
        // during the panel, see phase E; the prologue "panel" holds the single reflector u = 0 in slot 15)
        if (Tfac != nullptr && wave == NW - 1 && lane < 16) {
            float2 *dst = Tfac + (bm * 17 + (P0 + p + 1)) * 256 + lane * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float2 t = sh.Tl[lane][i];
                if (p < 0) t = (lane == 15 && i == 15) ? sh.taubuf[0] : make_float2(0.f, 0.f);
                if (16 * p + 1 + i >= DL) t = make_float2(0.f, 0.f);  // slot without a reflector (u = D)
                dst[i] = t;
            }
        }
        if (p == NT - 1) break;
        // ---- trailing update on the matrix cores: tiles (I, J), I >= J >= p + 1:  T -= V_I W_J^H + W_I V_J^H
        //      re -= Vr Wr' + Vi Wi' + Wr Vr' + Wi Vi' ;  im -= Vi Wr' - Vr Wi' + Wi Vr' - Wr Vi'   (' = block column J)
        //      A operands: rows of block row I (lane (m = c16, g) supplies k' = 4 g + s at step s), B operands from LDS.
        //      One block row at a time: 24 operand registers live across the column loop.
        c16 = c16_0;
        g = g_0;
        asm volatile("" : "+v"(c16), "+v"(g));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int P1 = p + 1;
            const int I = half ? IA : IB;
            // (compiler-level memory barrier: without it the second block row's B-operand loads are merged with the
            //  first one's -- same LDS addresses -- and all sixteen block columns' operands stay live: 85 VGPR spills)
            asm volatile("" ::: "memory");
            if (I < P1) continue;   // (uniform)
            // Three real products per complex one (3M, as in wy_apply.hip): with ' = block column J,
            //     S1 = Vr Wr' + Wr Vr',  S2 = Vi Wi' + Wi Vi',  S3 = (Vr + Vi)(Wr' - Wi') + (Wr + Wi)(Vr' - Vi')
            //     re -= S1 + S2,   im -= S3 - S1 + S2
            // 6 instead of 8 matrix-core instructions per k-step.
            float aVr[4], aVi[4], nVs[4], aWr[4], aWi[4], nWs[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float2 v = sh.Vp[16 * I + c16][4 * g + s], w = sh.Wp[16 * I + c16][4 * g + s];
                aVr[s] = v.x; aVi[s] = v.y; nVs[s] = v.x + v.y;
                aWr[s] = w.x; aWi[s] = w.y; nWs[s] = w.x + w.y;
            }
#pragma unroll
            for (int J = 0; J < NT; ++J) {
                if (J >= P1 && J <= I) {   // (uniform)
                    float2 bV[4], bW[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        bV[s] = sh.Vp[16 * J + c16][4 * g + s];
                        bW[s] = sh.Wp[16 * J + c16][4 * g + s];
                    }
                    f32x4 re = half ? tr[NT - J] : tr[J], im = half ? ti[NT - J] : ti[J];
                    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aVr[s], bW[s].x, s1, 0, 0, 0);
                        s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(aVi[s], bW[s].y, s2, 0, 0, 0);
                        s3 = __builtin_amdgcn_mfma_f32_16x16x4f32(nVs[s], bW[s].x - bW[s].y, s3, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(aWr[s], bV[s].x, s1, 0, 0, 0);
                        s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(aWi[s], bV[s].y, s2, 0, 0, 0);
                        s3 = __builtin_amdgcn_mfma_f32_16x16x4f32(nWs[s], bV[s].x - bV[s].y, s3, 0, 0, 0);
                    }
                    const float usc = (J == I) ? 0.5f : 1.0f;   // (a diagonal tile is held at half its value)
                    re = re - (s1 + s2) * usc;
                    im = im + ((s1 - s2) - s3) * usc;
                    if (half) {
                        tr[NT - J] = re;
                        ti[NT - J] = im;
                    } else {
                        tr[J] = re;
                        ti[J] = im;
                    }
                    if (J == P1) {   // (uniform) the next panel's columns, up to date: rows 16 I + 4 g + q, column c16
                        const float cs = (J == I) ? 2.0f : 1.0f;
                        float2 *dst = &sh.Ap[16 * I + 4 * g][c16];
                        dst[0] = make_float2(re.x * cs, im.x * cs);
                        dst[PN_PITCH] = make_float2(re.y * cs, im.y * cs);
                        dst[2 * PN_PITCH] = make_float2(re.z * cs, im.z * cs);
                        dst[3 * PN_PITCH] = make_float2(re.w * cs, im.w * cs);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // operands of block column J + 1 are not hoisted above these MFMAs
            }
        }
        mark(7);
        if (p == pstop - 1) {   // (uniform) hand the trailing matrix to the next stage: tiles (I, J), I >= J >= pstop
#pragma unroll
            for (int s = 0; s < NT + 1; ++s) {
                PN_SLOT_IJ(s, I, J)
                if (J >= pstop) {   // (uniform)
                    const f32x4 re = tr[s], im = ti[s];
                    tail[pn_tail_at(I - pstop, J - pstop, 0, lane)] = make_float2(re.x, im.x);
                    tail[pn_tail_at(I - pstop, J - pstop, 1, lane)] = make_float2(re.y, im.y);
                    tail[pn_tail_at(I - pstop, J - pstop, 2, lane)] = make_float2(re.z, im.z);
                    tail[pn_tail_at(I - pstop, J - pstop, 3, lane)] = make_float2(re.w, im.w);
                }
            }
            break;
        }
        __syncthreads();   // the next panel's columns (Ap) are complete; Vp / Wp may be rewritten
    }
    if constexpr (TIMING) {
        if (lane == 0)
            for (int i = 0; i < 14; ++i) atomicAdd(&tdbg[14 * wave + i], tacc[i]);
    }
    __syncthreads();
    // d, e, taus of the columns this stage reduced: local indices (HEAD: 0, else 1) .. min(16 pstop, DL)
    const int hi = (pstop < NT) ? 16 * pstop : DL;
    for (int i = tid + (HEAD ? 0 : 1); i <= hi; i += THREADS) {
        dcol[R0 + i] = sh.dbuf[i];
        ecol[R0 + i] = (i < DL) ? sh.ebuf[i] : 0.f;
        if (i < DL) Mg[(int64_t)D * D + R0 + i] = sh.taubuf[i];   // taus live in the consumed arrow slot
    }
}

bool tridiag_panel_supported(int D) { return D == PN_D; }
int64_t tridiag_panel_tail_elems() { return PN_TAIL_TILES * 256; }

// Stage split: the first kernel (8 waves, one workgroup per CU: the register-resident half of the 256 x 256 matrix)
// reduces panels 0 .. 7, the second (4 waves, 69 KB of LDS: two workgroups per CU) the trailing 128 x 128 matrix.  Every
// reflector is a chain of five barrier-separated latency-bound phases, so two independent matrices per CU overlap where
// one leaves the CU idle.  ADMMNET_PN_SPLIT=0 runs the whole reduction in the first kernel.
static int pn_split() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("ADMMNET_PN_SPLIT");
        v = (e && !strcmp(e, "0")) ? 0 : (e && !strcmp(e, "8")) ? 8 : 84;
    }
    return v;
}

template <int NT, bool HEAD, bool TIMING>
static int pn_launch_stage(int64_t nb, const Ws &ws, int pstop, unsigned long long *tdbg, hipStream_t st) {
    const size_t lds = sizeof(PnShared<NT>);
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tridiag_panel_kernel<NT, HEAD, TIMING>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((tridiag_panel_kernel<NT, HEAD, TIMING>), dim3((unsigned)nb), dim3(32 * NT), lds, st, ws.Mbuf,
                       ws.dT, ws.eT, ws.Tfac, ws.Tail, pstop, use_wy_back(PN_D) ? 0 : 1, tdbg, ws.skip);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_tridiag_panel(int D, int64_t nb, const Ws &ws, hipStream_t st) {
    if (!tridiag_panel_supported(D)) {
        set_error("tridiag_panel: D=%d unsupported (256 only)", D);
        return ADMMNET_E_ARG;
    }
    const bool split = pn_split() != 0 && ws.Tail != nullptr, split3 = split && pn_split() == 84;
    static const bool timing = getenv("ADMMNET_PN_TIMING") != nullptr;   // developer aid, never on by default
    if (timing) {
        unsigned long long *ptime = nullptr, hb[3 * 112];
        ADMM_HIP(hipMalloc(&ptime, sizeof(hb)));
        ADMM_HIP(hipMemsetAsync(ptime, 0, sizeof(hb), st));
        int rc = pn_launch_stage<16, true, true>(nb, ws, split ? 8 : 16, ptime, st);
        if (rc == ADMMNET_OK && split) rc = pn_launch_stage<8, false, true>(nb, ws, split3 ? 4 : 8, ptime + 112, st);
        if (rc == ADMMNET_OK && split3) rc = pn_launch_stage<4, false, true>(nb, ws, 4, ptime + 224, st);
        if (rc != ADMMNET_OK) return rc;
        ADMM_HIP(hipMemcpyAsync(hb, ptime, sizeof(hb), hipMemcpyDeviceToHost, st));
        ADMM_HIP(hipStreamSynchronize(st));
        ADMM_HIP(hipFree(ptime));
        static const char *nm[14] = {"w | column+norm", "reflector+dots", "matvec tiles", "row flush", "dot fix-up", "E: dot | look-ahead+T", "E: assemble y", "panel end + mfma",
                                     "wait B2", "wait B3", "wait B4", "wait B5", "E: corrections", ""};
        static const int order[13] = {0, 8, 1, 9, 2, 3, 4, 10, 6, 12, 5, 11, 7};
        fprintf(stderr, "[tridiag_panel timing] nb=%lld  mean kilocycles per matrix and wave\n", (long long)nb);
        for (int stg = 0; stg < 3; ++stg) {
            const int nw = stg == 0 ? 8 : (stg == 1 ? 4 : 2);
            if ((stg == 1 && !split) || (stg == 2 && !split3)) continue;
            fprintf(stderr, " stage %d\n", stg + 1);
            for (int ii = 0; ii < 13; ++ii) {
                const int i = order[ii];
                fprintf(stderr, "   %-22s", nm[i]);
                for (int w = 0; w < nw; ++w) fprintf(stderr, " %8.1f", (double)hb[112 * stg + 14 * w + i] / (double)nb / 1e3);
                fprintf(stderr, "\n");
            }
        }
        return ADMMNET_OK;
    }
    int rc = pn_launch_stage<16, true, false>(nb, ws, split ? 8 : 16, nullptr, st);
    if (rc == ADMMNET_OK && split) rc = pn_launch_stage<8, false, false>(nb, ws, split3 ? 4 : 8, nullptr, st);
    if (rc == ADMMNET_OK && split3) rc = pn_launch_stage<4, false, false>(nb, ws, 4, nullptr, st);
    return rc;
}

}  // namespace admmnet
