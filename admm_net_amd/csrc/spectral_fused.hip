// spectral_fused.hip -- the matrix-function evaluation of the G-layer (spectral.hip has the mathematics and the checks) as ONE
// kernel per chunk: a 768-thread workgroup per matrix reads the lower triangle of the state Z (six times, five of them from the
// L2 / the memory-side cache), and writes the lower triangle of G once.  Nothing else goes through HBM -- no n x n scratch matrix.
//
//   P1  the two outlier eigenpairs by subspace iteration, A x formed from the lower triangle of Z on the fly
//       (A = [[diag h, phi], [phi^H, corner]] - Z / rho is never stored): wave w owns the rows i = w mod 12, lane l the columns
//       l + 64 m; the row part sum_{j <= i} A_ij x_j is reduced across the wave, the mirrored part y_j += conj(A_ij) x_i stays in
//       lane-private accumulators until the end of the pass.  Both vectors ride the same pass.  fp32 throughout.
//   P2  E = A - c I - sum_k mu_k v_k v_k^H in slabs of 32 rows, rounded to bf16 and written to LDS TRANSPOSED (ET[j][k]), so that
//       both operands of  O = E^H E  (O_ij = sum_k conj(E_ki) E_kj) are k-contiguous 16-byte LDS reads in the matrix cores' lane
//       layout.  bf16 is enough HERE AND ONLY HERE: a2 E^2 is the second-order term, |a2| ||E||^2 <= 1e-4 of the result's scale is
//       checked per matrix (else: eigen-pipeline), so the 2^-9 relative rounding of the operands moves G by < 4e-7 of its scale.
//   P3  v_mfma_f32_32x32x16_bf16 on the resident accumulators: the 32 x 32 tiles of the lower triangle of the D x D interior are
//       dealt to the twelve waves (three per wave at D = 256, consecutive ones: they share a tile row) and stay in registers over all slabs; the border row (index D, the
//       arrow) of E^2 is accumulated by the vector ALUs from the same slabs.
//   P4  ||E^2||_F -> delta, the quadratic model of f and its checks (spectral.hip), then
//       G = (a0 - a1 c) I + a1 A + a2 E^2 + sum_k (f(lam_k) - a0 - a1 mu_k) v_k v_k^H  straight from the accumulator layout
//       (A once more from Z), and ||G - C_z||_F for the Z-layer.
// Matrices that fail a check are flagged and leave G untouched: the eigen-pipeline runs them (Ws::skip).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "lane_reduce.h"

namespace admmnet {

constexpr int SF_THREADS = 768, SF_WAVES = 12;   // three waves per SIMD, 168 registers each: every phase is latency-bound
constexpr int SF_PITCH = 40;   // bf16 per slab row: 32 k-values + 8 of padding (80 bytes: 16-byte aligned, spreads the banks)
constexpr int SF_MAXM = 4;     // column chunks of 64 per row: columns 0 .. 255; the one column beyond them -- the corner element of
                               // row 256 at n = 257 -- is handled by a single lane (a fifth chunk would cost 2 registers in every row buffer)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ unsigned sf_pack_bf16(float a, float b) {   // (a -> low half, b -> high half), round to nearest even
    const f32x2v p = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2));
}
__device__ __forceinline__ float sf_bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float sf_bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// Four wave-wide sums for the price of ~one: returns t with t(row 0) = sum a, t(row 1) = sum b, t(row 2) = sum c,
// t(row 3) = sum d (rows = the four 16-lane groups of the wave; every lane of a row holds the sum).
__device__ __forceinline__ float sf_reduce4(float a, float b, float c, float d) {
    // halves: lanes 0..31 keep (a, b), lanes 32..63 keep (c, d)
    float a2 = a, c2 = c, b2 = b, d2 = d;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(a2), "+v"(c2), "+v"(b2), "+v"(d2));
    float x = a2 + c2, y = b2 + d2;   // x: [a lo+hi | c lo+hi], y: [b | d]
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    return pn_row16_sum(x + y);       // rows: a, b, c, d
}

// block-wide sums of four values: every thread gets all four (through LDS; two barriers)
template <int W>
__device__ __forceinline__ void sf_block_sum4(float &a, float &b, float &c, float &d, float *red /* [4][W] */) {
    const float t = sf_reduce4(a, b, c, d);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if ((lane & 15) == 0) red[(lane >> 4) * W + wave] = t;
    __syncthreads();
    float s[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) v += red[q * W + w];
        s[q] = v;
    }
    a = s[0]; b = s[1]; c = s[2]; d = s[3];
}

// six values in one round (two barriers)
template <int W>
__device__ __forceinline__ void sf_block_sum8(float &a, float &b, float &c, float &d, float &e, float &f, float *red /* [8][W] */) {
    const float t = sf_reduce4(a, b, c, d), u = sf_reduce4(e, f, 0.f, 0.f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if ((lane & 15) == 0) {
        red[(lane >> 4) * W + wave] = t;
        red[(4 + (lane >> 4)) * W + wave] = u;
    }
    __syncthreads();
    float s[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) v += red[q * W + w];
        s[q] = v;
    }
    a = s[0]; b = s[1]; c = s[2]; d = s[3]; e = s[4]; f = s[5];
}

__device__ inline double sf_eig_map(double w, double thr, const float *vn) {   // rebuild_lds.h: br_eig_map, in double
    const double x = w - thr;
    const double base = x > 20.0 ? x : log1p(exp(x));
    const double a = fabs(w);
    double acc = vn[48];
    for (int j = 0; j < 16; ++j) {
        const double pre = (double)vn[j] * a + (double)vn[16 + j];
        acc += (double)vn[32 + j] * (pre > 0.0 ? pre : 0.0);
    }
    return base / (1.0 + exp(-acc));
}

struct SfCarve {
    int NP, NJ;
    float2 *X0, *X1, *Y0, *Y1, *ph, *Ob;
    float *hh, *hp, *red, *coef;
    double *sc;
    int *flags;
    float2 *part;          // [waves][2][NP]      (P1)
    unsigned short *ETre;  // [NJ][SF_PITCH]      (P2 .. P3, aliases part)
    unsigned short *ETim;
    __host__ __device__ static int np_of(int n) { return (n + 7) & ~7; }
    __host__ __device__ static int nj_of(int n) {
        const int D = n - 1, nt = (D + 31) >> 5;
        const int a = 32 * nt, b = (n + 7) & ~7;
        return a > b ? a : b;
    }
    __host__ __device__ static size_t bytes(int n, int W) {
        const size_t NP = np_of(n), NJ = nj_of(n);
        const size_t fixed = sizeof(float2) * 6 * NP + sizeof(float) * 2 * NP + sizeof(float) * 96 + sizeof(float) * 16 +
                             sizeof(double) * 16 + sizeof(int) * 4;
        const size_t part = sizeof(float2) * W * 2 * NP;
        const size_t slab = sizeof(unsigned short) * 2 * NJ * SF_PITCH;
        return fixed + (part > slab ? part : slab) + 64;
    }
    __device__ SfCarve(char *smem, int n) {
        NP = np_of(n);
        NJ = nj_of(n);
        sc = reinterpret_cast<double *>(smem);                       // 16 doubles
        X0 = reinterpret_cast<float2 *>(sc + 16);
        X1 = X0 + NP; Y0 = X1 + NP; Y1 = Y0 + NP; ph = Y1 + NP; Ob = ph + NP;
        hh = reinterpret_cast<float *>(Ob + NP);
        hp = hh + NP;                                               // previous layer's h (lazy Z update)
        red = hp + NP;                                              // 96 floats
        coef = red + 96;                                            // 16 floats
        flags = reinterpret_cast<int *>(coef + 16);                 // 4 ints
        // (offsets, not pointer arithmetic through integers: the compiler must keep seeing LDS addresses -- a round trip through
        //  uintptr_t turned every slab access into a flat_load with a 64-bit address that it then spilled around the MFMAs)
        const int fixed = (int)(sizeof(double) * 16 + sizeof(float2) * 6 * NP + sizeof(float) * 2 * NP + sizeof(float) * 112 +
                                sizeof(int) * 4);
        char *u = smem + ((fixed + 15) & ~15);
        part = reinterpret_cast<float2 *>(u);
        ETre = reinterpret_cast<unsigned short *>(u);
        ETim = ETre + NJ * SF_PITCH;
    }
};

// The lazy Z update of the previous layer, folded into the first sweep (prep_kernel then only computes phi and h):
//   Z <- Z + alpha_b (G - C_prev),  C_prev = [[diag h_prev, phi_prev], [phi_prev^H, corner_z of the previous layer]]   (admm_net.py:400-412)
// mode 0: Z is current (prep streamed it); 1: update; 2: update with the stored Z still zero (layer 1: never written).
struct SfUpdate {
    const float *alpha;        // [B] step of the previous layer
    const float2 *phi_prev;    // [B][D]
    const float *h_prev;       // [B][D]
    const float *lw_prev;      // packed weights of the previous layer (S_CORNER_Z)
    int mode;
};

// A_ij (i >= j) of the layer matrix from the state element z = Z_ij
__device__ __forceinline__ float2 sf_a_elem(int i, int j, int D, float2 z, float ir, float corner, float hi, float2 phj) {
    float2 a = make_float2(-ir * z.x, -ir * z.y);
    if (i < D) {
        if (i == j) a = make_float2(hi - ir * z.x, 0.f);
    } else {
        a = (j == D) ? make_float2(corner - ir * z.x, 0.f) : make_float2(phj.x - ir * z.x, -phj.y - ir * z.y);
    }
    return a;
}

// W waves (64 W threads) per matrix: 12 for n > 129 (one workgroup per CU), 4 for n <= 129 -- three matrices per CU then overlap
// each other's latencies (at those sizes the triangle stays in the L2 and the kernel is bound by its dependent phases)
template <int TPW, int W>
__global__ __launch_bounds__(64 * W, W == 12 ? 1 : 3) void sp_fused_kernel(int D, const float *__restrict__ lw, const float2 *__restrict__ phi,
                                                                 const float *__restrict__ h, float2 *Zg,
                                                                 float2 *G, float *__restrict__ rn,
                                                                 int *__restrict__ flag, int32_t *__restrict__ status, float tol,
                                                                 int iters, unsigned long long *__restrict__ ptime, SfUpdate up) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // developer phase timer (ADMMNET_SF_TIMING=1): cycles of thread 0 between marks
    long long t_prev = ptime ? clock64() : 0;
    auto mark = [&](int id) {
        if (ptime && threadIdx.x == 0) {
            const long long t_now = clock64();
            atomicAdd(&ptime[id], (unsigned long long)(t_now - t_prev));
            t_prev = t_now;
        }
    };
    const int n = D + 1;
    const SfCarve cv(smem, n);
    const int NP = cv.NP;
    const int64_t b = blockIdx.x;
    float2 *Z = Zg + b * (int64_t)n * n;
    const float2 *Gold = G + b * (int64_t)n * n;   // (read only in the first pass, with up.mode: G of the previous layer)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float ir = lw[S_INV_RHO_G], corner = lw[S_CORNER_G];
    const int NT = (D + 31) >> 5, ntri = NT * (NT + 1) / 2;
    const int MM = (n + 63) >> 6;   // column chunks in use

    // ---- load phi, h; start vectors (the outlier pair of the arrowhead lives in span{e_D, (phi, 0)}) ----------------------
    float pn = 0.f;
    if (tid < NP) {
        const float2 p = tid < D ? phi[b * D + tid] : make_float2(0.f, 0.f);
        cv.ph[tid] = p;
        cv.hh[tid] = tid < D ? h[b * D + tid] : 0.f;
        pn = p.x * p.x + p.y * p.y;
        if (up.mode) {   // C of the previous layer for the lazy Z update (Ob is free until the border rows are written)
            cv.Ob[tid] = tid < D ? up.phi_prev[b * D + tid] : make_float2(0.f, 0.f);
            cv.hp[tid] = tid < D ? up.h_prev[b * D + tid] : 0.f;
        }
    }
    const float al = up.mode ? up.alpha[b] : 0.f;
    const float corner_zp = up.mode ? up.lw_prev[S_CORNER_Z] : 0.f;
    {
        float z1 = 0.f, z2 = 0.f, z3 = 0.f;
        sf_block_sum4<W>(pn, z1, z2, z3, cv.red);
    }
    const float ipn = pn > 0.f ? rsqrtf(pn) : 0.f;
    if (tid < NP) {
        const float2 p = cv.ph[tid];
        cv.X0[tid] = make_float2(tid == D ? 1.f : 0.f, 0.f);
        cv.X1[tid] = pn > 0.f ? make_float2(p.x * ipn, p.y * ipn) : make_float2(tid == 0 ? 1.f : 0.f, 0.f);
    }
    __syncthreads();

    mark(0);
    // ---- P1: subspace iteration ---------------------------------------------------------------------------------------------
    double trace = 0.0;
    float l0 = 0.f, l1 = 0.f, cf = 0.f, res0 = 0.f, res1 = 0.f;
    // at least two passes, then until both Ritz pairs have residuals below 2e-6 of their gap to the bulk (the acceptance test of
    // P4 asks for 1e-5), at most `iters`: the arrowhead start is so good that two or three passes are the rule
    bool last = false;
    for (int it = 0; it < iters; ++it) {
        v2f x0[SF_MAXM], x1[SF_MAXM], c0[SF_MAXM], c1[SF_MAXM];
#pragma unroll
        for (int m = 0; m < SF_MAXM; ++m) {
            const int j = lane + 64 * m;
            const bool v = m < MM && j < n;
            x0[m] = v ? pk2(cv.X0[j]) : v2f{0.f, 0.f};
            x1[m] = v ? pk2(cv.X1[j]) : v2f{0.f, 0.f};
            c0[m] = v2f{0.f, 0.f};
            c1[m] = v2f{0.f, 0.f};
        }
        float trl = 0.f;
        // (the sweep in two instances: the first pass with the folded Z update carries the rows of the old G as well; the others
        //  must not pay for its registers)
        auto run_pass = [&](auto updc) {
        constexpr bool UPD = decltype(updc)::value;
        float2 za[SF_MAXM], zb[SF_MAXM], zc[SF_MAXM], zd[SF_MAXM];
        float2 ga[SF_MAXM], gb[SF_MAXM], gc[SF_MAXM], gd[SF_MAXM];   // (first pass with the Z update: the rows of the old G)
        auto load_row = [&](int i, float2(&z)[SF_MAXM], float2(&g)[SF_MAXM]) {
#pragma unroll
            for (int m = 0; m < SF_MAXM; ++m) {
                const int j = lane + 64 * m;
                const bool v = i < n && j <= i;
                z[m] = (v && !(UPD && up.mode == 2)) ? Z[(int64_t)i * n + j] : make_float2(0.f, 0.f);
                if (UPD) g[m] = v ? Gold[(int64_t)i * n + j] : make_float2(0.f, 0.f);
            }
        };
        auto proc_row = [&](int i, float2(&zc)[SF_MAXM], const float2(&gq)[SF_MAXM]) {
            if (UPD) {   // Z <- Z + alpha (G - C_prev) for this row, written back; the rest of the kernel reads the new Z
                const float hpi = cv.hp[i];
#pragma unroll
                for (int m = 0; m < SF_MAXM; ++m) {
                    if (64 * m <= i) {
                        const int j = lane + 64 * m;
                        if (j <= i) {
                            float2 c;
                            if (i < D) c = make_float2(i == j ? hpi : 0.f, 0.f);
                            else if (j == D) c = make_float2(corner_zp, 0.f);
                            else {
                                const float2 pp = cv.Ob[j];
                                c = make_float2(pp.x, -pp.y);   // C[D][j] = conj(phi_prev_j)
                            }
                            const float2 zn = make_float2(zc[m].x + al * (gq[m].x - c.x), zc[m].y + al * (gq[m].y - c.y));
                            zc[m] = zn;
                            Z[(int64_t)i * n + j] = zn;
                        }
                    }
                }
            }
            const v2f xi0 = pk2(cv.X0[i]), xi1 = pk2(cv.X1[i]);
            const float hi = cv.hh[i];
            v2f r0 = {0.f, 0.f}, r1 = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < SF_MAXM; ++m) {
                if (64 * m <= i) {   // (wave-uniform)
                    const int j = lane + 64 * m;
                    float2 a = sf_a_elem(i, j, D, zc[m], ir, corner, hi, (i == D && j < D) ? cv.ph[j] : make_float2(0.f, 0.f));
                    if (j > i) a = make_float2(0.f, 0.f);
                    if (j == i) trl += a.x;
                    const v2f av = pk2(a);
                    r0 = pk_cfma(r0, av, x0[m]);
                    r1 = pk_cfma(r1, av, x1[m]);
                    const v2f ac = (j < i) ? av : v2f{0.f, 0.f};
                    c0[m] = pk_cfma_conj(c0[m], ac, xi0);
                    c1[m] = pk_cfma_conj(c1[m], ac, xi1);
                }
            }
            if (i == 64 * SF_MAXM && lane == 0) {   // (n = 257, row 256) the corner element Z[256][256]: no chunk holds it
                const int64_t idx = (int64_t)i * n + i;
                float2 z = (UPD && up.mode == 2) ? make_float2(0.f, 0.f) : Z[idx];
                if (UPD) {
                    const float2 g = Gold[idx];
                    z = make_float2(z.x + al * (g.x - corner_zp), z.y + al * g.y);
                    Z[idx] = z;
                }
                const float a = corner - ir * z.x;          // (i = D here: D = 256 is the only geometry with a column 256)
                trl += a;
                const float2 xa = cv.X0[i], xb = cv.X1[i];
                r0.x += a * xa.x; r0.y += a * xa.y;
                r1.x += a * xb.x; r1.y += a * xb.y;
            }
            const float t = sf_reduce4(r0.x, r0.y, r1.x, r1.y);
            if ((lane & 15) == 0) {   // rows of the wave: Y0.re, Y0.im, Y1.re, Y1.im of row i (this wave owns row i)
                float *dst = reinterpret_cast<float *>((lane & 32) ? cv.Y1 : cv.Y0);
                dst[2 * i + ((lane >> 4) & 1)] = t;
            }
        };
        // two rows per trip, the next two in flight behind them (the loads are what bounds this phase)
        load_row(wave, za, ga);
        load_row(wave + W, zb, gb);
        for (int i = wave; i < n; i += 2 * W) {
            load_row(i + 2 * W, zc, gc);
            load_row(i + 3 * W, zd, gd);
            proc_row(i, za, ga);
            if (i + W < n) proc_row(i + W, zb, gb);
#pragma unroll
            for (int m = 0; m < SF_MAXM; ++m) {
                za[m] = zc[m];
                zb[m] = zd[m];
                ga[m] = gc[m];
                gb[m] = gd[m];
            }
        }
        };
        if (up.mode != 0 && it == 0) run_pass(std::true_type{});   // (uniform)
        else run_pass(std::false_type{});
        // the mirrored part: per-wave partial sums, then one add per column
#pragma unroll
        for (int m = 0; m < SF_MAXM; ++m) {
            const int j = lane + 64 * m;
            if (m < MM && j < NP) {
                cv.part[(wave * 2 + 0) * NP + j] = make_float2(c0[m].x, c0[m].y);
                cv.part[(wave * 2 + 1) * NP + j] = make_float2(c1[m].x, c1[m].y);
            }
        }
        if (it == 0) {
            float z1 = 0.f, z2 = 0.f, z3 = 0.f;
            sf_block_sum4<W>(trl, z1, z2, z3, cv.red);   // (barriers inside)
            trace = (double)trl;
        }
        __syncthreads();
        mark(1);
        float2 x0e = make_float2(0.f, 0.f), x1e = x0e, y0e = x0e, y1e = x0e;
        if (tid < n) {
            y0e = cv.Y0[tid];
            y1e = cv.Y1[tid];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                if (tid >= 64 * SF_MAXM) break;   // (column 256 has no mirrored part: nothing lies below the corner)
                const float2 p0 = cv.part[(w * 2 + 0) * NP + tid], p1 = cv.part[(w * 2 + 1) * NP + tid];
                y0e.x += p0.x; y0e.y += p0.y;
                y1e.x += p1.x; y1e.y += p1.y;
            }
            x0e = cv.X0[tid];
            x1e = cv.X1[tid];
        }
        // H = X^H Y
        float h00 = x0e.x * y0e.x + x0e.y * y0e.y, h11 = x1e.x * y1e.x + x1e.y * y1e.y;
        float h01r = x0e.x * y1e.x + x0e.y * y1e.y, h01i = x0e.x * y1e.y - x0e.y * y1e.x;
        sf_block_sum4<W>(h00, h11, h01r, h01i, cv.red);
        // closed-form eigen-decomposition of [[h00, h01], [conj(h01), h11]] (every thread; fp32 like the sums it is made of)
        float ct = 1.f, st = 0.f, er = 1.f, ei = 0.f;
        {
            const float ab = sqrtf(h01r * h01r + h01i * h01i);
            const float dif = 0.5f * (h00 - h11), rad = sqrtf(dif * dif + ab * ab);
            l0 = 0.5f * (h00 + h11) - rad;
            l1 = 0.5f * (h00 + h11) + rad;
            // eigenvector of the larger eigenvalue of [[a, |b|], [|b|, d]]: (cos t, sin t), the better-conditioned of its two forms
            const float vx = dif >= 0.f ? dif + rad : ab, vy = dif >= 0.f ? ab : rad - dif;
            const float nrm = sqrtf(vx * vx + vy * vy);
            if (nrm > 0.f) {
                ct = vx / nrm;
                st = vy / nrm;
            }
            if (ab > 0.f) {
                er = h01r / ab;
                ei = -h01i / ab;
            }
            cf = (float)((trace - (double)l0 - (double)l1) / (double)(n - 2));
            if (tid == 0) {
                cv.sc[0] = l0; cv.sc[1] = l1; cv.sc[2] = cf;
            }
        }
        // rotate into the Ritz basis, residuals, the power step and its Gram-Schmidt sums in one round
        const float2 ex1 = make_float2(er * x1e.x - ei * x1e.y, er * x1e.y + ei * x1e.x);
        const float2 ey1 = make_float2(er * y1e.x - ei * y1e.y, er * y1e.y + ei * y1e.x);
        const float2 nx0 = make_float2(-st * x0e.x + ct * ex1.x, -st * x0e.y + ct * ex1.y);
        const float2 nx1 = make_float2(ct * x0e.x + st * ex1.x, ct * x0e.y + st * ex1.y);
        const float2 ny0 = make_float2(-st * y0e.x + ct * ey1.x, -st * y0e.y + ct * ey1.y);
        const float2 ny1 = make_float2(ct * y0e.x + st * ey1.x, ct * y0e.y + st * ey1.y);
        const float2 d0 = make_float2(ny0.x - l0 * nx0.x, ny0.y - l0 * nx0.y);
        const float2 d1 = make_float2(ny1.x - l1 * nx1.x, ny1.y - l1 * nx1.y);
        float r0s = d0.x * d0.x + d0.y * d0.y, r1s = d1.x * d1.x + d1.y * d1.y;
        const float2 p0 = make_float2(ny0.x - cf * nx0.x, ny0.y - cf * nx0.y);   // power step
        const float2 p1 = make_float2(ny1.x - cf * nx1.x, ny1.y - cf * nx1.y);
        float n0 = p0.x * p0.x + p0.y * p0.y, n1 = p1.x * p1.x + p1.y * p1.y;
        float pr = p0.x * p1.x + p0.y * p1.y, pi = p0.x * p1.y - p0.y * p1.x;   // conj(p0) p1
        sf_block_sum8<W>(r0s, r1s, n0, n1, pr, pi, cv.red);
        res0 = sqrtf(r0s);
        res1 = sqrtf(r1s);
        last = it + 1 >= iters ||
               (it >= 1 && res0 <= 2e-6f * fabsf(l0 - cf) && res1 <= 2e-6f * fabsf(l1 - cf));   // (uniform: block-wide sums)
        if (!last) {
            const float in0 = n0 > 0.f ? rsqrtf(n0) : 0.f;
            const float2 q0 = make_float2(p0.x * in0, p0.y * in0);
            const float gr = pr * in0, gi = pi * in0;   // q0^H p1
            float2 q1 = make_float2(p1.x - (gr * q0.x - gi * q0.y), p1.y - (gr * q0.y + gi * q0.x));
            const float n1o = n1 - (gr * gr + gi * gi);
            const float in1 = n1o > 0.f ? rsqrtf(n1o) : 0.f;
            q1.x *= in1;
            q1.y *= in1;
            if (tid < NP) {
                cv.X0[tid] = tid < n ? q0 : make_float2(0.f, 0.f);
                cv.X1[tid] = tid < n ? q1 : make_float2(0.f, 0.f);
            }
        } else if (tid < NP) {
            cv.X0[tid] = tid < n ? nx0 : make_float2(0.f, 0.f);
            cv.X1[tid] = tid < n ? nx1 : make_float2(0.f, 0.f);
        }
        __syncthreads();
        mark(2);
        if (last) break;
    }
    const float mu0 = l0 - cf, mu1 = l1 - cf;

    // ---- P2 + P3: E in bf16 slabs through LDS, E^H E on the matrix cores ------------------------------------------------------
    int tI[TPW], tJ[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        const int t = wave * TPW + s;
        int I = 0;
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        tI[s] = I;
        tJ[s] = t - I * (I + 1) / 2;
    }
    f32x16 accRe[TPW], accIm[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            accRe[s][q] = 0.f;
            accIm[s][q] = 0.f;
        }
    }
    float2 ob = make_float2(0.f, 0.f);   // border row of E^2: thread j accumulates sum_k conj(E[k][D]) E[k][j]
    unsigned *ETre32 = reinterpret_cast<unsigned *>(cv.ETre), *ETim32 = reinterpret_cast<unsigned *>(cv.ETim);
    constexpr int P32 = SF_PITCH / 2;   // slab pitch in dwords
    // rows j >= n of the slab are operands of ignored outputs only: zero them once so that nothing non-finite is multiplied
    for (int idx = tid; idx < (cv.NJ - n) * P32; idx += (64 * W)) {
        ETre32[n * P32 + idx] = 0u;
        ETim32[n * P32 + idx] = 0u;
    }
    const int nslab = (n + 31) >> 5;
    for (int K = 0; K < nslab; ++K) {
        const int k0 = 32 * K;
        __syncthreads();   // the previous slab's readers are done
        // (opaque copy of the thread index: everything derived from it is recomputed per slab instead of being hoisted out of
        //  the loop and spilled -- a reload from scratch is a memory round trip)
        int tl = tid;
        asm volatile("" : "+v"(tl));
        // element pair (j; k, k + 1) of E^T: e_q = E[k + q][j].  j <= k: from rows k, k + 1 of Z (lanes over j);
        // j >= k + 1: conj of E[j][k + q] from row j of Z (lanes over k)
        // (the vector elements every element of a thread shares -- its column in the L region, its row pair in the U region -- are read
        //  from LDS once per slab, the pair-wise ones as one 16-byte read: the staging was bound by the LDS pipe, 14 LDS
        //  instructions per element pair; now 5)
        auto deflate = [&](float2 a, float2 vr0, float2 vc0, float2 vr1, float2 vc1, bool diag) {   // E[row][col] from A[row][col]
            const float2 q0 = cmulc(vr0, vc0), q1 = cmulc(vr1, vc1);
            a.x -= mu0 * q0.x + mu1 * q1.x;
            a.y -= mu0 * q0.y + mu1 * q1.y;
            if (diag) {
                a.x -= cf;
                a.y = 0.f;
            }
            return a;
        };
        auto pair_of = [&](const float2 *v, int k, float2 &lo, float2 &hi) {   // v[k], v[k + 1] (k even: one 16-byte read)
            const float4 t = *reinterpret_cast<const float4 *>(v + k);
            lo = make_float2(t.x, t.y);
            hi = make_float2(t.z, t.w);
        };
        {   // L region: three row pairs per sweep, 256 columns each; the loads of all sweeps in flight together
            constexpr int RP = (64 * W) / 256, LQ = (16 + RP - 1) / RP;
            const int j = tl & 255;
            const float2 vj0 = cv.X0[j], vj1 = cv.X1[j], phj = cv.ph[j];   // (j <= 255 < NP)
            constexpr int LB = 3;
#pragma unroll
            for (int qb = 0; qb < LQ; qb += LB) {
            float2 zl0[LB], zl1[LB];
#pragma unroll
            for (int uu = 0; uu < LB; ++uu) {
                const int kp = RP * (qb + uu) + (tl >> 8);
                const int k = k0 + 2 * kp;
                const bool v = kp < 16 && j <= k && j < n;
                zl0[uu] = (v && k < n) ? Z[(int64_t)k * n + j] : make_float2(0.f, 0.f);
                zl1[uu] = (v && k + 1 < n) ? Z[(int64_t)(k + 1) * n + j] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int kp = RP * (qb + u) + (tl >> 8);
                const int k = k0 + 2 * kp;
                if (kp < 16 && j <= k && j < n) {
                    float2 vk0, vk0b, vk1, vk1b;
                    pair_of(cv.X0, k, vk0, vk0b);     // (k + 1 <= 257 + 30 < the padded length only for k < NP - 1: see below)
                    pair_of(cv.X1, k, vk1, vk1b);
                    const float2 hk = *reinterpret_cast<const float2 *>(cv.hh + k);
                    float2 e0 = make_float2(0.f, 0.f), e1 = e0;
                    if (k < n) e0 = deflate(sf_a_elem(k, j, D, zl0[u], ir, corner, hk.x, phj), vk0, vj0, vk1, vj1, k == j);
                    if (k + 1 < n) e1 = deflate(sf_a_elem(k + 1, j, D, zl1[u], ir, corner, hk.y, phj), vk0b, vj0, vk1b, vj1, k + 1 == j);
                    ETre32[j * P32 + kp] = sf_pack_bf16(e0.x, e1.x);
                    ETim32[j * P32 + kp] = sf_pack_bf16(e0.y, e1.y);
                }
            }
            }
        }
        if (tl < 16 && D >= 256) {   // (the L sweep covers j <= 255: the corner element j = k = D = 256 of the last slab)
            const int kp = tl, k = k0 + 2 * kp, j = 256;
            if (j <= k && j < n) {   // (k = 256: the corner; k > 256: padding of the k dimension)
                float2 e0 = make_float2(0.f, 0.f);
                if (k < n)
                    e0 = deflate(sf_a_elem(k, j, D, Z[(int64_t)k * n + j], ir, corner, cv.hh[k], cv.ph[j]), cv.X0[k], cv.X0[j], cv.X1[k],
                                 cv.X1[j], k == j);
                ETre32[j * P32 + kp] = sf_pack_bf16(e0.x, 0.f);
                ETim32[j * P32 + kp] = sf_pack_bf16(e0.y, 0.f);
            }
        }
        {   // U region: 48 rows of Z per sweep (16 lanes per row), the loads of three sweeps in flight
            constexpr int RPS = (64 * W) / 16, UB = 2;
            const int kp = tl & 15, k = k0 + 2 * kp;
            const int kc = k < NP - 1 ? k : 0;   // (slab positions beyond the matrix: their pairs are never used)
            float2 vk0, vk0b, vk1, vk1b, phk, phkb;
            pair_of(cv.X0, kc, vk0, vk0b);
            pair_of(cv.X1, kc, vk1, vk1b);
            pair_of(cv.ph, kc, phk, phkb);
            for (int jb = k0 + 1; jb < n; jb += UB * RPS) {
                float2 zu0[UB], zu1[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int j = jb + RPS * u + (tl >> 4);
                    const bool v = j < n && j >= k + 1;
                    zu0[u] = v ? Z[(int64_t)j * n + k] : make_float2(0.f, 0.f);
                    zu1[u] = (v && k + 1 < n) ? Z[(int64_t)j * n + k + 1] : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int j = jb + RPS * u + (tl >> 4);
                    if (j < n && j >= k + 1) {
                        const float hj = cv.hh[j];
                        const float2 vj0 = cv.X0[j], vj1 = cv.X1[j];
                        float2 e0 = deflate(sf_a_elem(j, k, D, zu0[u], ir, corner, hj, phk), vj0, vk0, vj1, vk1, false);
                        float2 e1 = make_float2(0.f, 0.f);
                        if (k + 1 < n)   // (k + 1 <= j; equal: the diagonal element)
                            e1 = deflate(sf_a_elem(j, k + 1, D, zu1[u], ir, corner, hj, phkb), vj0, vk0b, vj1, vk1b, j == k + 1);
                        ETre32[j * P32 + kp] = sf_pack_bf16(e0.x, e1.x);     // E[k][j] = conj(E[j][k])
                        ETim32[j * P32 + kp] = sf_pack_bf16(-e0.y, -e1.y);
                    }
                }
            }
        }
        __syncthreads();
        mark(3);
        // matrix cores: O_IJ += sum_k conj(E[k][i]) E[k][j]
        {
            int r32 = lane & 31, kh = lane >> 5;
            asm volatile("" : "+v"(r32), "+v"(kh));   // (see above: no hoisted, spilled operand addresses)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int off = 8 * ks + 4 * kh;   // dwords: 16 k-values per step, 8 per lane half
                uint4 are = make_uint4(0, 0, 0, 0), aim = are;
#pragma unroll
                for (int s = 0; s < TPW; ++s) {
                    if (wave * TPW + s < ntri) {   // (wave-uniform)
                        const int ia = 32 * tI[s] + r32, jb_ = 32 * tJ[s] + r32;
                        if (s == 0 || tI[s] != tI[s - 1]) {   // (uniform) consecutive tiles share the row operand
                            are = *reinterpret_cast<const uint4 *>(ETre32 + ia * P32 + off);
                            aim = *reinterpret_cast<const uint4 *>(ETim32 + ia * P32 + off);
                        }
                        const uint4 bre = *reinterpret_cast<const uint4 *>(ETre32 + jb_ * P32 + off);
                        const uint4 bim = *reinterpret_cast<const uint4 *>(ETim32 + jb_ * P32 + off);
                        const uint4 nbre = make_uint4(bre.x ^ 0x80008000u, bre.y ^ 0x80008000u, bre.z ^ 0x80008000u,
                                                      bre.w ^ 0x80008000u);
                        const bf16x8 Are = __builtin_bit_cast(bf16x8, are), Aim = __builtin_bit_cast(bf16x8, aim);
                        const bf16x8 Bre = __builtin_bit_cast(bf16x8, bre), Bim = __builtin_bit_cast(bf16x8, bim);
                        const bf16x8 nBre = __builtin_bit_cast(bf16x8, nbre);
                        // conj(a) b = (ar br + ai bi) + i (ar bi - ai br)
                        accRe[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Are, Bre, accRe[s], 0, 0, 0);
                        accIm[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Are, Bim, accIm[s], 0, 0, 0);
                        accRe[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aim, Bim, accRe[s], 0, 0, 0);
                        accIm[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Aim, nBre, accIm[s], 0, 0, 0);
                    }
                }
            }
        }
        mark(4);
        // (a barrier of its own in front of the border rows: with the vector-ALU phase of one wave running beside the
        //  matrix-core phase of another, ~1 matrix in 10^4 came out with the real part of ONE border element of E^2 different from
        //  run to run -- tests/gpu_spectral_determinism2.py; a wait + delay at this point instead of the barrier does not cure it)
        __syncthreads();
        // border row of E^2 on the vector ALUs
        if (tid < n) {
            const unsigned *dre = ETre32 + D * P32, *dim = ETim32 + D * P32;
            const unsigned *jre = ETre32 + tid * P32, *jim = ETim32 + tid * P32;
#pragma unroll 2
            for (int kp = 0; kp < 16; ++kp) {
                const unsigned ur = dre[kp], ui = dim[kp], vr = jre[kp], vi = jim[kp];
                const float ar0 = sf_bf16_lo(ur), ai0 = sf_bf16_lo(ui), br0 = sf_bf16_lo(vr), bi0 = sf_bf16_lo(vi);
                const float ar1 = sf_bf16_hi(ur), ai1 = sf_bf16_hi(ui), br1 = sf_bf16_hi(vr), bi1 = sf_bf16_hi(vi);
                {   // SCALAR fused multiply-adds, spelled out: the compiler packs the plain C expression into v_pk_mul_f32 / v_pk_fma_f32
                    // with op_sel / neg modifiers, and with those this accumulation was NOT reproducible run to run whenever another
                    // wave of the CU was in its bf16 matrix-core phase (one border element's real part in ~1 matrix of 10^4; first seen
                    // inside a workgroup, cured by the barrier above; back with three workgroups per CU; gone with this form: 0 of 32
                    // comparisons of 16384 matrices against 16 of 32 -- tests/gpu_spectral_determinism2.py)
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ob.x) : "v"(ar0), "v"(br0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ob.x) : "v"(ai0), "v"(bi0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ob.x) : "v"(ar1), "v"(br1));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ob.x) : "v"(ai1), "v"(bi1));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ob.y) : "v"(ar0), "v"(bi0));
                    asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(ob.y) : "v"(ai0), "v"(br0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ob.y) : "v"(ar1), "v"(bi1));
                    asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(ob.y) : "v"(ai1), "v"(br1));
                }
            }
        }
        mark(5);
    }
    __syncthreads();
    if (tid < NP) cv.Ob[tid] = tid < n ? ob : make_float2(0.f, 0.f);

    // ---- P4: ||E^2||_F, the model of f, the checks ----------------------------------------------------------------------------
    float fro = 0.f;
    {
        const int r32 = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
            if (wave * TPW + s < ntri) {
                const float wgt = tI[s] == tJ[s] ? 1.f : 2.f;   // a diagonal tile holds both triangles
                const int gj = 32 * tJ[s] + r32;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int gi = 32 * tI[s] + (q & 3) + 8 * (q >> 2) + 4 * kh;
                    if (gi < D && gj < D) fro += wgt * (accRe[s][q] * accRe[s][q] + accIm[s][q] * accIm[s][q]);
                }
            }
        }
        if (tid < n) fro += (tid < D ? 2.f : 1.f) * (ob.x * ob.x + ob.y * ob.y);
        float z1 = 0.f, z2 = 0.f, z3 = 0.f;
        sf_block_sum4<W>(fro, z1, z2, z3, cv.red);
    }
    const LayerLayout L{D};
    const float *vn = lw + L.off_vn();
    if (wave == 0) {   // the 11 evaluations of f on 11 lanes, in double
        const double thr = lw[S_THR];
        const double dl0 = cv.sc[0], dl1 = cv.sc[1], c = cv.sc[2];
        const double delta = sqrt(sqrt((double)fro));   // ||E||_2 <= ||E^2||_F^(1/2)
        const double ts[9] = {-1.0, 0.0, 1.0, -0.75, -0.5, -0.25, 0.25, 0.5, 0.75};
        double arg = c;
        if (lane < 9) arg = c + ts[lane] * delta;
        else if (lane == 9) arg = dl0;
        else if (lane == 10) arg = dl1;
        const double fv = sf_eig_map(arg, thr, vn);
        auto bcast = [&](int src) {
            const long long bits = __builtin_bit_cast(long long, fv);
            const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), src);
            const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src);
            return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned long long)(unsigned)lo);
        };
        const double fm = bcast(0), fc = bcast(1), fp = bcast(2), f0k = bcast(9), f1k = bcast(10);
        const double a0 = fc, a1 = (fp - fm) / (2.0 * delta), a2 = (fp - 2.0 * fc + fm) / (2.0 * delta * delta);
        const double scale = fmax(fmax(fabs(fc), fmax(fabs(f0k), fabs(f1k))), 1e-6);
        double miss = 0.0;
        if (lane >= 3 && lane < 9) {
            const double t = ts[lane] * delta;
            miss = fabs(fv - (a0 + a1 * t + a2 * t * t));
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) miss = fmax(miss, __shfl_xor(miss, o, 64));   // lanes 0..15
        if (lane == 0) {
            int why = 0;
            const double g0 = fabs(dl0 - c), g1 = fabs(dl1 - c);
            if (!(isfinite(dl0) && isfinite(dl1) && isfinite(c) && isfinite(delta) && delta > 0.0 && isfinite(miss))) why = 8;
            else if (!((double)res0 <= 1e-5 * fmax(g0, 1e-30) && (double)res1 <= 1e-5 * fmax(g1, 1e-30))) why = 1;
            else if (!(delta < 0.05 * fmin(g0, g1))) why = 2;
            else if (!(miss <= (double)tol * scale)) why = 4;
            else if (!(fabs(a2) * delta * delta <= 1e-4 * scale)) why = 16;   // the bf16 operands of E^2 must not matter
            cv.coef[0] = (float)(a0 - a1 * c);
            cv.coef[1] = (float)a1;
            cv.coef[2] = (float)a2;
            cv.coef[3] = (float)(f0k - a0 - a1 * (dl0 - c));
            cv.coef[4] = (float)(f1k - a0 - a1 * (dl1 - c));
            cv.flags[0] = why;
            flag[b] = why;
            if (status) {
                atomicAdd(status + (why ? 1 : 2), 1);
                if (why == 4) atomicAdd(status + 3, 1);
            }
        }
    }
    __syncthreads();
    mark(6);
    if (cv.flags[0]) return;   // (uniform) the eigen-pipeline takes this matrix

    // ---- G from the accumulator layout ---------------------------------------------------------------------------------------
    const float k0c = cv.coef[0], a1f = cv.coef[1], a2f = cv.coef[2], g0f = cv.coef[3], g1f = cv.coef[4];
    const float corner_z = lw[S_CORNER_Z];
    float2 *Gb = G + b * (int64_t)n * n;
    float acc = 0.f;
    {
        const int r32 = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int s = 0; s < TPW; ++s) {
            if (wave * TPW + s < ntri) {
                const int gj = 32 * tJ[s] + r32;
                const float2 vj0 = cv.X0[gj < n ? gj : 0], vj1 = cv.X1[gj < n ? gj : 0];
                constexpr int AB = 16;
#pragma unroll
                for (int qb = 0; qb < 16; qb += AB) {
                    float2 zq[AB];
#pragma unroll
                    for (int u = 0; u < AB; ++u) {   // (the loads of the tile first)
                        const int q = qb + u, gi = 32 * tI[s] + (q & 3) + 8 * (q >> 2) + 4 * kh;
                        zq[u] = (gi < D && gj <= gi) ? Z[(int64_t)gi * n + gj] : make_float2(0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < AB; ++u) {
                        const int q = qb + u, gi = 32 * tI[s] + (q & 3) + 8 * (q >> 2) + 4 * kh;
                        if (gi < D && gj <= gi) {
                            const float hi = cv.hh[gi];
                            const float2 a = sf_a_elem(gi, gj, D, zq[u], ir, corner, hi, make_float2(0.f, 0.f));
                            const float2 p0 = cmulc(cv.X0[gi], vj0), p1 = cmulc(cv.X1[gi], vj1);
                            float2 g = make_float2(a1f * a.x + a2f * accRe[s][q] + g0f * p0.x + g1f * p1.x,
                                                   a1f * a.y + a2f * accIm[s][q] + g0f * p0.y + g1f * p1.y);
                            float cz = 0.f;
                            if (gi == gj) {
                                g.x += k0c;
                                g.y = 0.f;
                                cz = hi;
                            }
                            Gb[(int64_t)gi * n + gj] = g;
                            const float dr = g.x - cz;
                            acc += (gi == gj ? 1.f : 2.f) * (dr * dr + g.y * g.y);
                        }
                    }
                }
            }
        }
    }
    if (tid < n) {   // the border row
        const int j = tid;
        const float2 a = sf_a_elem(D, j, D, Z[(int64_t)D * n + j], ir, corner, 0.f, cv.ph[j]);
        const float2 o = cv.Ob[j];
        const float2 p0 = cmulc(cv.X0[D], cv.X0[j]), p1 = cmulc(cv.X1[D], cv.X1[j]);
        float2 g = make_float2(a1f * a.x + a2f * o.x + g0f * p0.x + g1f * p1.x, a1f * a.y + a2f * o.y + g0f * p0.y + g1f * p1.y);
        float2 cz = make_float2(cv.ph[j].x, -cv.ph[j].y);
        if (j == D) {
            g.x += k0c;
            g.y = 0.f;
            cz = make_float2(corner_z, 0.f);
        }
        Gb[(int64_t)D * n + j] = g;
        const float dr = g.x - cz.x, di = g.y - cz.y;
        acc += (j == D ? 1.f : 2.f) * (dr * dr + di * di);
    }
    {
        float z1 = 0.f, z2 = 0.f, z3 = 0.f;
        sf_block_sum4<W>(acc, z1, z2, z3, cv.red);
    }
    if (tid == 0) rn[b] = sqrtf(acc);
    mark(7);
}

bool use_spectral_fused() {
    static const bool on = !(getenv("ADMMNET_SPECTRAL_FUSED") && atoi(getenv("ADMMNET_SPECTRAL_FUSED")) == 0);
    return on;
}

template <int TPW, int W>
static int sf_launch(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *Z, float2 *G, float *rn,
                     int *flag, int32_t *status, float tol, int iters, const SfUpdate &up, hipStream_t st) {
    const size_t lds = SfCarve::bytes(D + 1, W);
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sp_fused_kernel<TPW, W>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    static const bool timing = getenv("ADMMNET_SF_TIMING") != nullptr;   // developer aid, never on by default
    unsigned long long *ptime = nullptr;
    if (timing) {
        ADMM_HIP(hipMalloc(&ptime, 16 * sizeof(unsigned long long)));
        ADMM_HIP(hipMemsetAsync(ptime, 0, 16 * sizeof(unsigned long long), st));
    }
    hipLaunchKernelGGL((sp_fused_kernel<TPW, W>), dim3((unsigned)nb), dim3(64 * W), lds, st, D, lw, phi, h, Z, G, rn, flag, status,
                       tol, iters, ptime, up);
    ADMM_HIP(hipGetLastError());
    if (timing) {
        unsigned long long hb[16];
        ADMM_HIP(hipMemcpyAsync(hb, ptime, sizeof(hb), hipMemcpyDeviceToHost, st));
        ADMM_HIP(hipStreamSynchronize(st));
        ADMM_HIP(hipFree(ptime));
        fprintf(stderr, "[sf timing D=%d nb=%lld] cycles per workgroup: setup %.0f | matvec %.0f ritz %.0f (all passes) | stage %.0f "
                "mfma %.0f border %.0f (all slabs) | model %.0f | assemble %.0f\n", D, (long long)nb, hb[0] / (double)nb,
                hb[1] / (double)nb, hb[2] / (double)nb, (hb[3]) / (double)nb, hb[4] / (double)nb, hb[5] / (double)nb,
                hb[6] / (double)nb, hb[7] / (double)nb);
    }
    return ADMMNET_OK;
}

int launch_spectral_fused(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *Z, float2 *G,
                          float *rn, int *flag, int32_t *status, float tol, const float *alpha, const float2 *phi_prev,
                          const float *h_prev, const float *lw_prev, int update_mode, hipStream_t st) {
    const SfUpdate up{alpha, phi_prev, h_prev, lw_prev, update_mode};
    if (D < 2 || D > 256) {
        set_error("spectral: D=%d outside 2..256", D);
        return ADMMNET_E_ARG;
    }
    static const int iters = getenv("ADMMNET_SPECTRAL_ITERS") ? atoi(getenv("ADMMNET_SPECTRAL_ITERS")) : 5;   // (upper bound)
    const int NT = (D + 31) >> 5, ntri = NT * (NT + 1) / 2;
    static const bool small_wg = !(getenv("ADMMNET_SF_SMALLWG") && atoi(getenv("ADMMNET_SF_SMALLWG")) == 0);
    // 256 threads per matrix, three matrices per CU -- once there are more than two matrices per CU to overlap (measured at 10 x 10,
    // K = 10: 1024 signals 3.45 vs 3.63 ms per forward, 4096 signals 8.3 vs 10.4 ms; but 256 signals 2.42 vs 2.03 ms and a single
    // signal 0.64 vs 0.53 ms: a lone matrix is served faster by twelve waves)
    if (D <= 128 && small_wg && nb > 512) {
        switch ((ntri + 3) / 4) {
            case 1: return sf_launch<1, 4>(D, nb, lw, phi, h, Z, G, rn, flag, status, tol, iters, up, st);
            case 2: return sf_launch<2, 4>(D, nb, lw, phi, h, Z, G, rn, flag, status, tol, iters, up, st);
            default: return sf_launch<3, 4>(D, nb, lw, phi, h, Z, G, rn, flag, status, tol, iters, up, st);
        }
    }
    switch ((ntri + SF_WAVES - 1) / SF_WAVES) {
        case 1: return sf_launch<1, SF_WAVES>(D, nb, lw, phi, h, Z, G, rn, flag, status, tol, iters, up, st);
        case 2: return sf_launch<2, SF_WAVES>(D, nb, lw, phi, h, Z, G, rn, flag, status, tol, iters, up, st);
        default: return sf_launch<3, SF_WAVES>(D, nb, lw, phi, h, Z, G, rn, flag, status, tol, iters, up, st);
    }
}

}  // namespace admmnet
