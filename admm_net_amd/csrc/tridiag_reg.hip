// tridiag_reg.hip -- K1 for D <= 128: Householder tridiagonalisation + explicit Q with the
// matrix held IN REGISTERS (LDS only carries vectors).
//
// A 256-thread workgroup owns one matrix.  Thread (ti, tj) = (tid >> 4, tid & 15) holds the
// 2D-cyclic slice  m[a][b] = M[16 a + ti][16 b + tj]  (NA x NA complex, 128 VGPRs at NA = 8), so
//   * the work stays balanced while the active trailing block shrinks (a cyclic slice of it);
//   * every reduction the algorithm needs runs along tj = the 16 fast lanes of a DPP row
//     (4 row_ror adds, no LDS, no barrier);
//   * LDS per matrix is ~5 KB, so two workgroups share a CU and hide each other's latencies
//     (the LDS-resident version in tridiag.hip needs 140 KB and a barrier-bound 512 threads).
// Per reflector: 2 barriers (column broadcast, p / dot broadcast).  The explicit Q is
// accumulated afterwards as P = Q^H (P <- P H^H), which puts ITS reduction on the fast lanes
// too and needs no barrier at all.  Same mathematics as tridiag.hip / LAPACK chetd2 + cung2l
// (first half of torch.linalg.eigh, /root/reference/admm_net.py:303).
#include <cstdlib>

#include <type_traits>

#include "common.h"

namespace admmnet {

constexpr int TR_THREADS = 256;

// DPP lane exchanges: no LDS traffic, no barrier.  CTRL 0x121/2/4/8 = row_ror:1/2/4/8 (rotate inside
// each row of 16 lanes), 0x142 / 0x143 = row_bcast:15 / row_bcast:31 (last lane of a row -> next row(s)).
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ float dpp_get(float x) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xF, false));
}
// sum over the 16 lanes of a DPP row; every lane of the row receives the total
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_get<0x128>(x);
    x += dpp_get<0x124>(x);
    x += dpp_get<0x122>(x);
    x += dpp_get<0x121>(x);
    return x;
}
// sum over the whole wave as a wave-uniform value (row totals chained through row_bcast, then a readlane)
__device__ __forceinline__ float wave_sum_dpp(float x) {
    x = row16_sum(x);
    x += dpp_get<0x142, 0xA>(x);   // rows 1, 3 += row 0, 2
    x += dpp_get<0x143, 0xC>(x);   // rows 2, 3 += rows 0 + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// two wave sums at once: the chains interleave, so no DPP hazard stalls between dependent steps
__device__ __forceinline__ void wave_sum_dpp2(float &x, float &y) {
    x += dpp_get<0x128>(x); y += dpp_get<0x128>(y);
    x += dpp_get<0x124>(x); y += dpp_get<0x124>(y);
    x += dpp_get<0x122>(x); y += dpp_get<0x122>(y);
    x += dpp_get<0x121>(x); y += dpp_get<0x121>(y);
    x += dpp_get<0x142, 0xA>(x); y += dpp_get<0x142, 0xA>(y);
    x += dpp_get<0x143, 0xC>(x); y += dpp_get<0x143, 0xC>(y);
    x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
    y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y), 63));
}

// Complex arithmetic on (re, im) register pairs written so that hipcc emits ONE v_pk_fma_f32 per
// half of a complex multiply-add (op_sel broadcasts, no shuffles): f32 vector peak on gfx950 needs
// the packed form.  `rot(v)` = (-v.y, v.x) = i v and `rotc(v)` = (v.y, -v.x) = -i v are formed once
// per vector element and reused across a whole block row / column.
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 tov2(float2 a) { return v2{a.x, a.y}; }
__device__ __forceinline__ float2 tof2(v2 a) { return make_float2(a.x, a.y); }
__device__ __forceinline__ v2 rot(v2 v) { return v2{-v.y, v.x}; }
__device__ __forceinline__ v2 rotc(v2 v) { return v2{v.y, -v.x}; }
// acc + m * v, given vj = rot(v)
__device__ __forceinline__ v2 pk_cmac(v2 acc, v2 m, v2 v, v2 vj) {
    acc = __builtin_elementwise_fma(m.xx, v, acc);
    return __builtin_elementwise_fma(m.yy, vj, acc);
}
// acc + a * conj(b), given aj = rotc(a):  a conj(b) = b.x (a.x, a.y) + b.y (a.y, -a.x)
__device__ __forceinline__ v2 pk_cmacc(v2 acc, v2 a, v2 aj, v2 b) {
    acc = __builtin_elementwise_fma(b.xx, a, acc);
    return __builtin_elementwise_fma(b.yy, aj, acc);
}
// The same two products with the i * v / -i * a rotations folded into the operand selects and negate
// bits of v_pk_fma_f32 (the compiler materialises the rotated pair with a v_mov + v_xor per use):
//   acc + m * v        = fma(m.xx, (v.x, v.y), acc), then fma(m.yy, (-v.y, v.x), .)
//   acc + a * conj(b)  = fma(b.xx, (a.x, a.y), acc), then fma(b.yy, (a.y, -a.x), .)
// The hazard recogniser does not see inside inline asm: a value that a DPP / lane instruction reads next
// must come out of a compiler-emitted VALU op (see the last term of the HEMV loops).
__device__ __forceinline__ v2 pk_cmac_sel(v2 acc, v2 m, v2 v) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "+v"(acc)
        : "v"(m), "v"(v));
    return acc;
}
__device__ __forceinline__ v2 pk_cmacc_sel(v2 acc, v2 a, v2 b) {
    asm("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]"
        : "+v"(acc)
        : "v"(a), "v"(b));
    return acc;
}
// m * v (no accumulator to clear first)
__device__ __forceinline__ v2 pk_cmul_sel(v2 m, v2 v) {
    v2 acc;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(acc)
        : "v"(m), "v"(v));
    return acc;
}
// acc - a * conj(b): the same pair with the broadcast operand negated (neg_lo / neg_hi on src0)
__device__ __forceinline__ v2 pk_cmsubc_sel(v2 acc, v2 a, v2 b) {
    asm("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,1,0]"
        : "+v"(acc)
        : "v"(a), "v"(b));
    return acc;
}
// stage-wise row reduction of NA complex values: the NA chains interleave, so no DPP hazard stalls
template <int NA, int A0>
__device__ __forceinline__ void row16_sum_all(v2 (&acc)[NA]) {
#pragma unroll
    for (int a = A0; a < NA; ++a) { acc[a].x += dpp_get<0x128>(acc[a].x); acc[a].y += dpp_get<0x128>(acc[a].y); }
#pragma unroll
    for (int a = A0; a < NA; ++a) { acc[a].x += dpp_get<0x124>(acc[a].x); acc[a].y += dpp_get<0x124>(acc[a].y); }
#pragma unroll
    for (int a = A0; a < NA; ++a) { acc[a].x += dpp_get<0x122>(acc[a].x); acc[a].y += dpp_get<0x122>(acc[a].y); }
#pragma unroll
    for (int a = A0; a < NA; ++a) { acc[a].x += dpp_get<0x121>(acc[a].x); acc[a].y += dpp_get<0x121>(acc[a].y); }
}

// LDS vectors of one workgroup (double-buffered by reflector parity: one barrier separates a write
// from the reads of the previous use)
template <int NA>
struct TrShared {
    float2 colbuf[2][16 * NA];   // x = column below the unit position, ZERO at i <= u and i >= D
    float2 pbuf[2][16 * NA];     // p = tau M v
    float2 dotbuf[2][4];
    float2 head[2];              // alpha = column entry at the unit position
    float dprev[2];              // finished diagonal entry d[u]
    float2 taus[16 * NA];
};

// One Householder step with the unit entry at index u, for 16*A0 <= u < 16*(A0+1): every loop over
// the cyclic blocks starts at the compile-time bound A0, so the work shrinks with the trailing block
// (a run-time bound turned into select instructions and kept the full 8 x 8 cost per step).
template <int NA, int A0>
__device__ __forceinline__ void tr_step(float2 (&m)[NA][NA], TrShared<NA> &sh, int u, int D, float corner,
                                        const float2 *__restrict__ ag, float2 *__restrict__ Mg,
                                        float *__restrict__ dcol, float *__restrict__ ecol) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tj = tid & 15, ti = tid >> 4;
    const int par = u & 1;
    if (u == 0) {
        for (int i = tid; i < 16 * NA; i += TR_THREADS) {
            const float2 x = (i < D) ? ag[i] : make_float2(0.f, 0.f);
            sh.colbuf[0][i] = (i > 0) ? x : make_float2(0.f, 0.f);
            if (i == 0) sh.head[0] = x;
        }
    } else {
        const int k = u - 1;
        if (tj == (k & 15)) {
            // column k lives in block (k >> 4), which is A0 or (for u = 16 A0) A0 - 1
            constexpr int AP = A0 > 0 ? A0 - 1 : 0;
            auto publish = [&](auto blk) {   // blk: compile-time column block holding column k
                constexpr int B = decltype(blk)::value;
#pragma unroll
                for (int a = AP; a < NA; ++a) {
                    const int i = 16 * a + ti;
                    const float2 x = m[a][B];
                    sh.colbuf[par][i] = (i > u) ? x : make_float2(0.f, 0.f);
                    if (i == u) sh.head[par] = x;
                    if (i == k) sh.dprev[par] = x.x;
                }
            };
            if ((k >> 4) == A0) publish(std::integral_constant<int, A0>{});   // (uniform branch: no per-entry selects)
            else publish(std::integral_constant<int, AP>{});
        }
    }
    __syncthreads();   // (A) column visible
    const float2 *col = sh.colbuf[par];
    float pn = 0.f;
#pragma unroll
    for (int q = 0; q < (16 * NA + 63) / 64; ++q) {
        const int i = lane + 64 * q;
        if ((16 * NA) % 64 == 0 || i < 16 * NA) {
            const float2 x = col[i];
            pn += x.x * x.x + x.y * x.y;
        }
    }
    const float xn2 = wave_sum_dpp(pn);
    const float2 alpha = sh.head[par];
    float beta, tr, tim, sr, si;
    householder_c(alpha.x, alpha.y, xn2, beta, tr, tim, sr, si);
    // UNNORMALISED reflector: H = I - tau v v^H with v = s u, u = (alpha - beta at the unit position, x below)
    // is the same operator as I - gamma u u^H, gamma = tau |s|^2.  Working with (gamma, u) saves the scaling of
    // every vector entry in both passes (x is already in LDS / memory; only the head changes).
    const float g2 = sr * sr + si * si;
    const float2 tau = make_float2(tr * g2, tim * g2);
    const v2 hu = v2{alpha.x - beta, alpha.y};
    if (tid == 0) {
        ecol[u] = beta;
        dcol[u] = (u == 0) ? corner : sh.dprev[par];
        sh.taus[u] = tau;
    }
    // keep the reflector for the Q accumulation: row u of the (consumed) global image
    if (tid < D) {
        v2 x = tov2(col[tid]);
        if (tid == u) x = hu;
        Mg[(int64_t)u * D + tid] = tof2(x);
    }
    if (tr == 0.f && tim == 0.f) return;   // H = I (uniform)

    v2 vr[NA], vc[NA];
#pragma unroll
    for (int a = A0; a < NA; ++a) {
        vr[a] = tov2(col[16 * a + ti]);
        vc[a] = tov2(col[16 * a + tj]);
        if (a == A0) {   // the unit position can only sit in the first active block
            if (16 * a + ti == u) vr[a] = hu;
            if (16 * a + tj == u) vc[a] = hu;
        }
    }
    const v2 vcj = rot(vc[NA - 1]);
    // p = tau * M v : partial over my columns, summed along the 16 fast lanes (all lanes of a row
    // of the thread grid end up with the p of their matrix rows)
    v2 pr[NA];
#pragma unroll
    for (int a = A0; a < NA; ++a) {
        if constexpr (A0 < NA - 1) {
            v2 acc = pk_cmul_sel(tov2(m[a][A0]), vc[A0]);
#pragma unroll
            for (int b = A0 + 1; b < NA - 1; ++b) acc = pk_cmac_sel(acc, tov2(m[a][b]), vc[b]);
            pr[a] = pk_cmac(acc, tov2(m[a][NA - 1]), vc[NA - 1], vcj);   // last term by the compiler (DPP reads pr next)
        } else {
            pr[a] = pk_cmac(v2{0.f, 0.f}, tov2(m[a][NA - 1]), vc[NA - 1], vcj);
        }
    }
    row16_sum_all<NA, A0>(pr);
    const v2 tauv = tov2(tau), tauj = rot(tauv);
    v2 dotp = {0.f, 0.f}, psel = {0.f, 0.f};
#pragma unroll
    for (int a = A0; a < NA; ++a) {
        pr[a] = pk_cmul_sel(pr[a], tauv);   // tau * acc
        // conj(p) v = v.x (p.x, -p.y) + v.y (p.y, p.x); every lane of the 16 adds the same term,
        // the total is scaled by 1/16 below (exact)
        if (a < NA - 1) {
            dotp = pk_cmacc_sel(dotp, vr[a], pr[a]);   // + v_i conj(p_i)
        } else {   // last term by the compiler (DPP reads dotp next)
            dotp = __builtin_elementwise_fma(vr[a].xx, v2{pr[a].x, -pr[a].y}, dotp);
            dotp = __builtin_elementwise_fma(vr[a].yy, v2{pr[a].y, pr[a].x}, dotp);
        }
        psel = (tj == a) ? pr[a] : psel;
    }
    if (tj >= A0 && tj < NA) sh.pbuf[par][16 * tj + ti] = tof2(psel);   // lane a of a row publishes p of block row a
    {
        float dx = dotp.x, dy = dotp.y;
        wave_sum_dpp2(dx, dy);
        dotp = v2{dx, dy};
    }
    if (lane == 0) sh.dotbuf[par][wave] = make_float2(dotp.x * 0.0625f, dotp.y * 0.0625f);
    __syncthreads();   // (B) p and the dot partials visible
    float2 dot = sh.dotbuf[par][0];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        dot.x += sh.dotbuf[par][q].x;
        dot.y += sh.dotbuf[par][q].y;
    }
    float2 al = cmul(tau, dot);
    const v2 alv = v2{-0.5f * al.x, -0.5f * al.y}, alj = rot(alv);
    // M -= v w^H + w v^H on the active block:  x += (-v_i) conj(w_j) + (-w_i) conj(v_j)
    // (w = p + alpha v, zero on the finished rows / columns i < u, which only block A0 can hold)
    v2 wc[NA];
#pragma unroll
    for (int b = A0; b < NA; ++b) {
        const int j = 16 * b + tj;
        v2 w = pk_cmac_sel(tov2(sh.pbuf[par][j]), vc[b], alv);   // p_j + alpha v_j
        if (b == A0) w = (j >= u) ? w : v2{0.f, 0.f};
        wc[b] = w;
    }
#pragma unroll
    for (int a = A0; a < NA; ++a) {
        const int i = 16 * a + ti;
        v2 w = pk_cmac_sel(pr[a], vr[a], alv);   // p_i + alpha v_i
        if (a == A0) w = (i >= u) ? w : v2{0.f, 0.f};
        const v2 wra = w;
        const v2 vra = vr[a];
#pragma unroll
        for (int b = A0; b < NA; ++b) {
            v2 x = tov2(m[a][b]);
            x = pk_cmsubc_sel(x, vra, wc[b]);
            x = pk_cmsubc_sel(x, wra, vc[b]);
            if (a == b && ti == tj) x.y = 0.f;
            m[a][b] = tof2(x);
        }
    }
}

// reflector u restricted to my columns (blocks b >= B0), from the global image
// (float2 storage: arrays of ext_vector types that live across calls end up in scratch)
// FULL (D == 16 NA): no padding columns, the loads need no predicate (u = -1, past the last step,
// reads row 0 and is never used).
template <int NA, int B0, bool FULL>
__device__ __forceinline__ void q_load(float2 (&vn)[NA], int u, int D, const float2 *__restrict__ Mg) {
    const int tj = threadIdx.x & 15;
    const float2 *row = Mg + (int64_t)(u < 0 ? 0 : u) * D + tj;
#pragma unroll
    for (int b = B0; b < NA; ++b) {
        if constexpr (FULL) {
            vn[b] = row[16 * b];
        } else if (b < NA - 1) {
            vn[b] = row[16 * b];   // 16 (NA - 1) < D: only the last block has padding columns
        } else {
            // unconditional load from a clamped address, value selected afterwards: a load inside a branch waits
            // for the one before it, and the step then pays one memory round trip per block
            const int j = 16 * b + tj;
            const float2 x = row[min(j, D - 1) - tj];
            vn[b] = (j < D) ? x : make_float2(0.f, 0.f);
        }
    }
}

// P <- P (I - conj(tau) v v^H) for the reflector with unit entry u, 16*A0 <= u < 16*(A0+1).  `vc`
// holds reflector u on entry and reflector u - 1 on exit (its loads fly during this step).
template <int NA, int A0, bool FULL>
__device__ __forceinline__ void q_step(float2 (&m)[NA][NA], const TrShared<NA> &sh, int u, int D,
                                       const float2 *__restrict__ Mg, float2 (&vcs)[NA]) {
    constexpr int AP = A0 > 0 ? A0 - 1 : 0;
    float2 vn[NA];
    q_load<NA, AP, FULL>(vn, u - 1, D, Mg);
    const float2 tau = sh.taus[u];
    if (!(tau.x == 0.f && tau.y == 0.f)) {
        v2 vc[NA], y[NA];
#pragma unroll
        for (int b = A0; b < NA; ++b) vc[b] = tov2(vcs[b]);
        const v2 vjl = rot(vc[NA - 1]);
#pragma unroll
        for (int a = A0; a < NA; ++a) {
            if constexpr (A0 < NA - 1) {
                v2 acc = pk_cmul_sel(tov2(m[a][A0]), vc[A0]);
#pragma unroll
                for (int b = A0 + 1; b < NA - 1; ++b) acc = pk_cmac_sel(acc, tov2(m[a][b]), vc[b]);
                y[a] = pk_cmac(acc, tov2(m[a][NA - 1]), vc[NA - 1], vjl);   // last term by the compiler (DPP reads y next)
            } else {
                y[a] = pk_cmac(v2{0.f, 0.f}, tov2(m[a][NA - 1]), vc[NA - 1], vjl);
            }
        }
        row16_sum_all<NA, A0>(y);
        const v2 nct = v2{-tau.x, tau.y}, nctj = rot(nct);   // -conj(tau)
#pragma unroll
        for (int a = A0; a < NA; ++a) {
            const v2 nty = pk_cmul_sel(y[a], nct);
#pragma unroll
            for (int b = A0; b < NA; ++b) m[a][b] = tof2(pk_cmacc_sel(tov2(m[a][b]), nty, vc[b]));   // -= conj(tau) y conj(v_b)
        }
    }
#pragma unroll
    for (int b = AP; b < NA; ++b) vcs[b] = vn[b];
}

template <int NA, int A0>
struct TrPhases {
    static __device__ __forceinline__ void forward(float2 (&m)[NA][NA], TrShared<NA> &sh, int D, float corner,
                                                   const float2 *ag, float2 *Mg, float *dcol, float *ecol) {
        const int hi = min(16 * (A0 + 1), D);
        for (int u = 16 * A0; u < hi; ++u) tr_step<NA, A0>(m, sh, u, D, corner, ag, Mg, dcol, ecol);
        if constexpr (A0 + 1 < NA) TrPhases<NA, A0 + 1>::forward(m, sh, D, corner, ag, Mg, dcol, ecol);
    }
    template <bool FULL>
    static __device__ __forceinline__ void backward(float2 (&m)[NA][NA], const TrShared<NA> &sh, int D,
                                                    const float2 *Mg, float2 (&vc)[NA]) {
        if constexpr (A0 + 1 < NA) TrPhases<NA, A0 + 1>::template backward<FULL>(m, sh, D, Mg, vc);
        const int hi = min(16 * (A0 + 1), D);
        for (int u = hi - 1; u >= 16 * A0; --u) q_step<NA, A0, FULL>(m, sh, u, D, Mg, vc);
    }
};

// LEAN: the matrix A = C - Z / rho is formed HERE from the layer's own inputs -- the lower triangle of Z
// (original index order, arrow row last), phi, h and two scalars -- instead of being read from an image that
// the prep kernel would have to write first (A is Hermitian: 36 of the 64 register blocks are loaded, the
// rest are their conjugate transposes, fetched from the owning threads through LDS).  Per signal and layer
// this removes a 131 KB write and a 131 KB read of the image and halves the G / Z streams of the prep kernel.
template <int NA, bool LEAN, int OCC = 2>
__global__ __launch_bounds__(TR_THREADS, OCC) void tridiag_reg_kernel(int D, float2 *__restrict__ Mbuf,
                                                                    float *__restrict__ QV,
                                                                    float *__restrict__ dT,
                                                                    float *__restrict__ eT,
                                                                    const float2 *__restrict__ Zlow,
                                                                    const float2 *__restrict__ phi,
                                                                    const float *__restrict__ hvec,
                                                                    const float *__restrict__ lw,
                                                                    const int *__restrict__ skip) {
    __shared__ TrShared<NA> sh;
    const int tid = threadIdx.x;
    const int tj = tid & 15, ti = tid >> 4;
    const int64_t bm = blockIdx.x;
    if (skip && skip[bm] == 0) return;   // (uniform) this matrix' G is already there: spectral.hip
    const int n = D + 1;
    float2 *Mg = Mbuf + bm * ((int64_t)D * D + D + 1);
    const float2 *ag = Mg + (int64_t)D * D;
    float *dcol = dT + bm * n;
    float *ecol = eT + bm * n;

    float2 m[NA][NA];
    float corner;
    if constexpr (LEAN) {
        constexpr int NBLK = NA * (NA + 1) / 2, RB = (NBLK + 1) / 2;   // lower blocks, blocks per exchange round
        __shared__ float2 xch[RB][16][17];
        const float inv_rho = lw[S_INV_RHO_G], corner_g = lw[S_CORNER_G];
        const float2 *Zm = Zlow + bm * (int64_t)n * n;
        const float2 *ph = phi + bm * D;
        const float *hh = hvec + bm * D;
        // all loads unconditional (clamped address, value selected afterwards): inside per-block branches
        // every load waited for the previous one and the loader took 36 memory round trips
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NA; ++b) m[a][b] = make_float2(0.f, 0.f);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                const int i = 16 * a + ti, j = 16 * b + tj;
                const bool ok = i < D && j < D && i >= j;
                const float2 z = Zm[ok ? (int64_t)i * n + j : 0];
                float2 v = make_float2(-inv_rho * z.x, -inv_rho * z.y);
                if (a == b) {   // diagonal block: h on the diagonal, which is real
                    const float hv = hh[min(i, D - 1)];
                    if (ti == tj) v = make_float2(hv + v.x, 0.f);
                }
                m[a][b] = ok ? v : make_float2(0.f, 0.f);
            }
        // arrow column a_i = phi_i - conj(Z[D][i]) / rho and the corner go through the image's arrow slot
        if (tid < D) {
            const float2 z = Zm[(int64_t)D * n + tid], p = ph[tid];
            Mg[(int64_t)D * D + tid] = make_float2(p.x - inv_rho * z.x, p.y + inv_rho * z.y);
        }
        if (tid == 0) Mg[(int64_t)D * D + D] = make_float2(corner_g - inv_rho * Zm[(int64_t)D * n + D].x, 0.f);
        // conjugate transposes of the lower blocks (and of the lower halves of the diagonal blocks)
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            __syncthreads();
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int blk = a * (a + 1) / 2 + b;
                    if (blk / RB == round) xch[blk % RB][ti][tj] = m[a][b];
                }
            __syncthreads();
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b <= a; ++b) {
                    const int blk = a * (a + 1) / 2 + b;
                    if (blk / RB == round) {
                        const float2 t = xch[blk % RB][tj][ti];   // element (16 a + tj, 16 b + ti) of the lower block
                        if (a != b) m[b][a] = make_float2(t.x, -t.y);
                        else if (ti < tj) m[a][a] = make_float2(t.x, -t.y);
                    }
                }
        }
        __threadfence_block();
        __syncthreads();   // arrow slot visible to the whole workgroup
        corner = ag[D].x;
    } else {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NA; ++b) {
                const int i = 16 * a + ti, j = 16 * b + tj;
                const float2 x = Mg[(int64_t)min(i, D - 1) * D + min(j, D - 1)];   // clamped, unconditional (see q_load)
                m[a][b] = (i < D && j < D) ? x : make_float2(0.f, 0.f);
            }
        corner = ag[D].x;
        __syncthreads();   // all loads done before Mg rows are overwritten with reflectors
    }

    // ---------------- tridiagonalisation: reflector u has its unit entry at index u ----------
    TrPhases<NA, 0>::forward(m, sh, D, corner, ag, Mg, dcol, ecol);
    // last diagonal entry d[D] = Re M[D-1][D-1] (select chain: a run-time index would push m[][] to scratch)
    {
        const int k = D - 1, ka = k >> 4;
        float dl = 0.f;
#pragma unroll
        for (int a = 0; a < NA; ++a) dl = (a == ka) ? m[a][a].x : dl;
        if (ti == (k & 15) && tj == (k & 15)) {
            dcol[D] = dl;
            ecol[D] = 0.f;
        }
    }
    __syncthreads();   // reflector rows in Mg and taus[] complete

    // ---------------- explicit Q, accumulated as P = Q^H:  P <- P (I - conj(tau) v v^H) -------
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NA; ++b) m[a][b] = make_float2((a == b && ti == tj) ? 1.f : 0.f, 0.f);
    float2 vc[NA];
    if (D == 16 * NA) {   // uniform
        q_load<NA, 0, true>(vc, D - 1, D, Mg);
        TrPhases<NA, 0>::template backward<true>(m, sh, D, Mg, vc);
    } else {
        q_load<NA, 0, false>(vc, D - 1, D, Mg);
        TrPhases<NA, 0>::template backward<false>(m, sh, D, Mg, vc);
    }
    // ---------------- QT[c][rho] = Q[rho][c] = conj(P[c][rho]) ---------------------------------
    float *q = QV + bm * ((int64_t)n * 2 * D);
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NA; ++b) {
            const int c = 16 * a + ti, rho = 16 * b + tj;
            if (c < D && rho < D) {
                q[(int64_t)c * 2 * D + rho] = m[a][b].x;
                q[(int64_t)c * 2 * D + D + rho] = -m[a][b].y;
            }
        }
}

template <int NA>
static int launch_tr(int D, int64_t nb, const Ws &ws, hipStream_t st, const float2 *Zlow, const float2 *phi,
                     const float *h, const float *lw) {
    // developer knob: ADMMNET_TR_PAD_LDS=<bytes> of unused dynamic LDS per workgroup (e.g. 100000 leaves one
    // workgroup per CU: tells latency-bound from issue-bound)
    static const int pad = getenv("ADMMNET_TR_PAD_LDS") ? atoi(getenv("ADMMNET_TR_PAD_LDS")) : 0;
    if (pad > 0) {
        ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tridiag_reg_kernel<NA, true>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, pad));
        ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tridiag_reg_kernel<NA, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, pad));
    }
    // NA = 7 (97 <= D <= 112, the reference's 10 x 10 geometry): compiled for THREE workgroups per CU (168 registers, a few
    // values in scratch) -- the kernel is a latency chain per reflector and a third matrix per CU hides more of it than the
    // spills cost; ADMMNET_TR_OCC=2 keeps the two-workgroup build for A/B runs.
    static const int occ3 = !(getenv("ADMMNET_TR_OCC") && atoi(getenv("ADMMNET_TR_OCC")) == 2);
    if constexpr (NA == 7) {
        if (occ3 && pad == 0) {
            if (Zlow)
                hipLaunchKernelGGL((tridiag_reg_kernel<NA, true, 3>), dim3((unsigned)nb), dim3(TR_THREADS), 0, st, D, ws.Mbuf,
                                   ws.QV, ws.dT, ws.eT, Zlow, phi, h, lw, ws.skip);
            else
                hipLaunchKernelGGL((tridiag_reg_kernel<NA, false, 3>), dim3((unsigned)nb), dim3(TR_THREADS), 0, st, D, ws.Mbuf,
                                   ws.QV, ws.dT, ws.eT, Zlow, phi, h, lw, ws.skip);
            ADMM_HIP(hipGetLastError());
            return ADMMNET_OK;
        }
    }
    if (Zlow)
        hipLaunchKernelGGL((tridiag_reg_kernel<NA, true>), dim3((unsigned)nb), dim3(TR_THREADS), pad, st, D, ws.Mbuf,
                           ws.QV, ws.dT, ws.eT, Zlow, phi, h, lw, ws.skip);
    else
        hipLaunchKernelGGL((tridiag_reg_kernel<NA, false>), dim3((unsigned)nb), dim3(TR_THREADS), pad, st, D, ws.Mbuf,
                           ws.QV, ws.dT, ws.eT, Zlow, phi, h, lw, ws.skip);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

// Zlow != nullptr: "lean" loader (the kernel forms A = C - Z / rho itself, see tridiag_reg_kernel)
int launch_tridiag_reg(int D, int64_t nb, const Ws &ws, hipStream_t st, const float2 *Zlow, const float2 *phi,
                       const float *h, const float *lw) {
    const int na = (D + 15) / 16;
    switch (na) {
        case 1: return launch_tr<1>(D, nb, ws, st, Zlow, phi, h, lw);
        case 2: return launch_tr<2>(D, nb, ws, st, Zlow, phi, h, lw);
        case 3: return launch_tr<3>(D, nb, ws, st, Zlow, phi, h, lw);
        case 4: return launch_tr<4>(D, nb, ws, st, Zlow, phi, h, lw);
        case 5: return launch_tr<5>(D, nb, ws, st, Zlow, phi, h, lw);
        case 6: return launch_tr<6>(D, nb, ws, st, Zlow, phi, h, lw);
        case 7: return launch_tr<7>(D, nb, ws, st, Zlow, phi, h, lw);
        case 8: return launch_tr<8>(D, nb, ws, st, Zlow, phi, h, lw);
        default:
            set_error("tridiag_reg: D=%d unsupported", D);
            return ADMMNET_E_ARG;
    }
}

}  // namespace admmnet
