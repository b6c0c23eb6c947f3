// dc.hip -- K2': divide & conquer eigensolver for the real symmetric tridiagonal matrices,
// one 256-thread workgroup per matrix; K3': V = Q W on the matrix cores.
//
// Replaces the QL + rotation-replay pair (tql.hip / rotapply.hip) for the second half of
// torch.linalg.eigh (/root/reference/admm_net.py:303): LAPACK sstedc semantics -- tear T into
// leaves of 8, solve the leaves, then merge pairs level by level: sort, deflate (slaed2), solve
// the secular equation per root (slaed4's job, here a bracketed Illinois iteration on the
// pole-free transform), rebuild z by Loewner's formula (slaed3) and multiply the eigenvector
// blocks (v_mfma_f32_32x32x2_f32).  Every phase except the O(n) deflation scan is parallel over
// eigenvalues; clusters deflate instead of costing QL sweeps; the eigenvectors come out sorted.
// The scalar pieces live in dc_core.h and are shared with tests/host_model/dc_model.cpp.
//
// Layouts: eigenvector blocks are kept transposed, WT[j][i] = W[i][j] (j = eigenvalue index), in
// two ping-pong n x n global buffers per matrix; the rank-one eigenvectors U of a merge exist only in
// registers (regenerated under the MFMA); a third n x n region takes W row-major for the consumers that want it.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "dc_core.h"

namespace admmnet {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int DC_THREADS = 256;
constexpr int DC_LS = 8;          // nominal leaf size (dc_leaf_start in dc_core.h spreads the remainder)
constexpr int DC_MAXLEAF = 33;    // n <= 8 * 33 + 7
constexpr int DC_MAXLS = 2 * DC_LS;   // a single leaf (n < 16) has up to 15 rows
constexpr int DC_RA = 4;              // deflation rotations whose operands are loaded ahead (one L2 round trip per batch)
constexpr int DC_KB = 8;              // K-steps (of two eigenvector entries each) whose operands are loaded ahead in the merge products
constexpr int DC_FULL = 1 << 30;      // flag on a source-column index: a deflation rotation has filled all its rows

// leaf scratch: Z of every leaf as [maxrows][maxrows | 1] (odd pitch: the team's row-per-lane accesses spread
// over the banks)
__host__ __device__ inline size_t dc_leafz_floats(int n) {
    const int nleaf = dc_leaf_count(n), mr = dc_leaf_maxrows(n, nleaf);
    return (size_t)nleaf * mr * (mr | 1);
}

struct DcShared {
    int bnd[2][DC_MAXLEAF + 2];
    int kk[DC_MAXLEAF];      // non-deflated count per merge
    int nrot[DC_MAXLEAF];
    int conf[DC_MAXLEAF];    // team scan: a rotation chain reached the next run -> serial scan for this merge
    int mx[DC_MAXLEAF][2];   // max |d|, max |z| of a merge as float bit patterns (non-negative: integer order)
    int fail;
    int nrm;                 // max |d|, |e| of the whole matrix (float bits)
    int kc[DC_MAXLEAF][2];   // per merge: non-deflated source columns with rows in the first / second block
};

// LDS carve of one workgroup: the bookkeeping block, then float / int arrays of length NP each.  Built from (smem, n)
// wherever it is needed, so that the phases compiled as functions of their own (below) take no LDS pointers as
// arguments -- a pointer passed through a call would lose its address space.
struct DcCarve {
    DcShared *sh;
    int NP;
    float *lam, *e0, *zv, *ds, *zs, *un, *dl, *zl, *tau, *zh, *vals, *lamn;
    int *perm, *src, *org, *rnk, *cidx;
    DcRot *rot;
    float *leafZ, *leafD;
    __device__ __forceinline__ DcCarve(char *smem, int n) {
        sh = reinterpret_cast<DcShared *>(smem);
        NP = (n + 3) & ~3;
        lam = reinterpret_cast<float *>(smem + ((sizeof(DcShared) + 15) & ~(size_t)15));
        e0 = lam + NP;
        zv = e0 + NP;
        ds = zv + NP;
        zs = ds + NP;
        un = zs;            // (after the deflation scan) 1 / ||u_j|| of the merge's rank-one eigenvectors
        dl = zs + NP;
        zl = dl + NP;
        tau = zl + NP;
        zh = tau + NP;
        vals = zh + NP;
        lamn = vals + NP;
        perm = reinterpret_cast<int *>(lamn + NP);
        src = perm + NP;
        org = src + NP;
        rnk = org + NP;
        cidx = rnk + NP;                                      // source column (block-local) of merged position p
        rot = reinterpret_cast<DcRot *>(cidx + NP);           // [NP]
        leafZ = reinterpret_cast<float *>(rot + NP);          // [nleaf][maxrows][maxrows | 1]
        leafD = leafZ + dc_leafz_floats(n);                   // [nleaf][2 * DC_MAXLS]
    }
};

// The eigenvector products of one level: WTdst[rank(j)][i] = sum_kk U[kk][j] WTsrc[col(src[kk])][i] on the matrix cores,
// waves take tiles.  A function of its own (never inlined): inside dc_kernel its 32 + 16 accumulator / operand registers met
// the live ranges of every other phase at the 128-register budget of 4 workgroups per CU, and the allocator parked ~26
// values in scratch around EVERY tile (reloaded before, spilled after: two L2 round trips per tile).
template <int OCC, bool BLK>
static __device__ __attribute__((noinline)) void dc_level_gemm(int n, int nm, int cb, const float *__restrict__ Ws,
                                                      float *__restrict__ Wd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DcCarve cv(smem, n);
    DcShared &sh = *cv.sh;
    const int NP = cv.NP, tid = threadIdx.x;
    const int *bn = sh.bnd[cb];
    const float *dl = cv.dl, *tau = cv.tau, *un = cv.un, *zh = cv.zh;
    const int *org = cv.org, *rnk = cv.rnk, *cidx = cv.cidx;
    const DcRot *rot = cv.rot;
{
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, kh = lane >> 5;
    int gbase = 0;   // tiles of all merges of the level are dealt round-robin to the 4 waves
    for (int mm = 0; mm < nm; ++mm) {
        const int ma = bn[2 * mm], mc = bn[2 * mm + 2];
        const int mnn = mc - ma, mk = sh.kk[mm], mn1 = bn[2 * mm + 1] - ma;
        // work unit: two stacked 32 x 32 tiles (64 roots j) x 32 rows i; the B operand
        // (source columns, the only memory stream) is shared by the pair, the A operand
        // U[kk][j] = zh_kk / ((d_kk - d_org(j)) - tau_j) / ||u_j|| is generated in registers
        const int tm2 = (mk + 63) >> 6, tn = (mnn + 31) >> 5;
        const int first = (wave - gbase) & 3;
        gbase += tm2 * tn;
        for (int t = first; t < tm2 * tn; t += DC_THREADS / 64) {
            const int j0 = (t / tn) * 64, i0 = (t % tn) * 32;
            const bool two = j0 + 32 < mk;
            const bool jv0 = (j0 + r) < mk, jv1 = (j0 + 32 + r) < mk, iv = (i0 + r) < mnn;
            const int ja = ma + (jv0 ? j0 + r : 0), jb = ma + (jv1 ? j0 + 32 + r : 0);
            const float dorg0 = dl[ma + org[ja]], tau0 = tau[ja], inv0 = jv0 ? un[ja] : 0.f;
            const float dorg1 = dl[ma + org[jb]], tau1 = tau[jb], inv1 = jv1 ? un[jb] : 0.f;
            const int io = iv ? i0 + r : 0;
            f32x16 acc0 = {0}, acc1 = {0};
            // row tile inside one block: only that block's source columns (list), no masks; a tile that straddles
            // the block boundary (one per merge unless n1 is a multiple of 32) takes every column and masks
            const bool blk1 = BLK && min(i0 + 32, mnn) <= mn1, blk2 = BLK && i0 >= mn1;
            const int *kl = reinterpret_cast<const int *>(rot) + ma + (blk2 ? NP : 0);
            const int kcnt = blk1 ? sh.kc[mm][0] : (blk2 ? sh.kc[mm][1] : mk);
            const bool listed = blk1 || blk2;
            for (int k0 = 0; k0 < kcnt; k0 += 2 * DC_KB) {   // DC_KB K-steps per batch: all loads first
                float bv[DC_KB];
#pragma unroll
                for (int s16 = 0; s16 < DC_KB; ++s16) {
                    const int ki = k0 + 2 * s16 + kh;
                    const bool kv = ki < kcnt;
                    const int kq = listed ? kl[kv ? ki : 0] : (kv ? ki : 0);
                    const int cc = cidx[ma + kq], col = cc & ~DC_FULL;
                    const float b_ = Ws[(ma + col) * n + ma + io];
                    const bool ok = !BLK || listed || (cc & DC_FULL) || ((col < mn1) == (io < mn1));
                    bv[s16] = (kv && iv && ok) ? b_ : 0.f;
                }
#pragma unroll
                for (int s16 = 0; s16 < DC_KB; ++s16) {
                    const int ki = k0 + 2 * s16 + kh;
                    const bool kv = ki < kcnt;
                    const int kc = ma + (listed ? kl[kv ? ki : 0] : (kv ? ki : 0));
                    const float zk = kv ? zh[kc] : 0.f, dk = dl[kc];
                    const float a0 = kv ? fdiv_fast(zk, (dk - dorg0) - tau0) * inv0 : 0.f;
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[s16], acc0, 0, 0, 0);
                    if (two) {
                        const float a1 = kv ? fdiv_fast(zk, (dk - dorg1) - tau1) * inv1 : 0.f;
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[s16], acc1, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int j = j0 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                const int i = i0 + r;
                if (j < mk && i < mnn) Wd[(ma + rnk[ma + j]) * n + ma + i] = acc0[q];
                if (j + 32 < mk && i < mnn) Wd[(ma + rnk[ma + j + 32]) * n + ma + i] = acc1[q];
            }
        }
    }
}
}

// Deflation scan of one merge by ONE WAVE with the inputs in registers (levels whose teams are a wave or more).
// Same chain, same tests and arithmetic as deflate_scan_tol (dc_core.h).  There the single walker fetches (d_j, z_j)
// from LDS, branches four ways and stores its outputs as it goes: 600 cycles per position, 156 k cycles for the 257
// positions of the top-level merge -- and the team form does not help on the layer matrices, where one cluster is one
// run of rotation candidates, i.e. one walker.  Here
//   A  lane l holds the entries l + 64 m; every lane replays the chain redundantly, so all state is wave-uniform: the
//      operands arrive by v_readlane, the four cases are selects (no branch, no store, no wait), and what position j
//      decided -- its case, the survivor it displaced, the rotation -- is left in lane j's registers;
//   B  the outputs are placed by counting (ballots + prefix popcounts), all lanes storing at once.
// A function of its own (never inlined; OCC only carries the caller's register budget over): see dc_level_gemm.
__device__ __forceinline__ float dc_readlane(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
__device__ __forceinline__ float dc_uniform(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
constexpr int DC_WSM = 5;   // 64-entry register groups: n <= 8 * DC_MAXLEAF + 7 = 271 <= 320
// case of a position: tiny z (deflates), first non-tiny (becomes the survivor), push (the survivor before it is final: a
// non-deflated pole; the position becomes the survivor), rotate (the survivor before it deflates into the position)
constexpr int DW_TINY = 0, DW_FIRST = 1, DW_PUSH = 2, DW_ROT = 3;

struct DwGroup {   // per lane: the record of position lane + 64 m
    int typ, pold;         // case, survivor position before the step
    float a, b, c;         // push: (d, z) of the displaced survivor;  rotate: (c, s, deflated pole)
};

template <int OCC>
static __device__ __attribute__((noinline)) void deflate_scan_wave(int n, int team, int a, int nn_, float rho_, float dmax_,
                                                                 float zmax_) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const DcCarve cv(smem, n);
    const float *ds = cv.ds + a, *zs = cv.zs + a;
    float *dl = cv.dl + a, *zl = cv.zl + a;
    int *src = cv.src + a;
    DcRot *rot = cv.rot + a;
    int *k_out = &cv.sh->kk[team], *nrot_out = &cv.sh->nrot[team];
    const int lane = threadIdx.x & 63;
    const int nn = __builtin_amdgcn_readfirstlane(nn_);
    const float rho = dc_uniform(rho_), dmax = dc_uniform(dmax_), zmax = dc_uniform(zmax_);
    const float tol = 8.0f * kEps32 * fmaxf(dmax, zmax);
    if (rho * zmax <= tol) {   // the rank-one term is negligible: everything deflates (slot nn - 1 - j, as the serial scan)
        for (int j = lane; j < nn; j += 64) {
            dl[nn - 1 - j] = ds[j];
            src[nn - 1 - j] = j;
        }
        if (lane == 0) {
            *k_out = 0;
            *nrot_out = 0;
        }
        return;
    }
    float dr[DC_WSM], zr[DC_WSM];
#pragma unroll
    for (int m = 0; m < DC_WSM; ++m) {
        const int i = lane + 64 * m;
        dr[m] = (i < nn) ? ds[i] : 0.f;
        zr[m] = (i < nn) ? zs[i] : 0.f;
    }
    // ---- A: the chain
    int pj = -1;
    float dpj = 0.f, zpj = 0.f;
    auto chain = [&](int m, float dreg, float zreg, DwGroup &rec) {
        rec.typ = DW_TINY;
        rec.pold = -1;
        rec.a = rec.b = rec.c = 0.f;
        const int cnt = min(64, nn - 64 * m);   // (uniform; <= 0 for absent groups)
        for (int jj = 0; jj < cnt; ++jj) {
            const int j = 64 * m + jj;
            const float dj = dc_readlane(dreg, jj), zj = dc_readlane(zreg, jj);
            const bool tiny = rho * fabsf(zj) <= tol;                 // type 1: tiny z component
            const bool have = pj >= 0;
            // type 2: two (nearly) equal poles -> rotate z_pj into z_j (tests as deflate_scan_tol)
            const float q = zj * zj + zpj * zpj;
            const float t = dj - dpj;
            const bool close = (fabsf(t * zj * zpj) <= tol * q) && !(q < 1e-30f);
            const float itau = rsqrt_nr(q);
            const float tau = q * itau;
            const float c = zj * itau, sn = -zpj * itau;
            const float dde = dpj * c * c + dj * sn * sn;            // the pole the rotation deflates
            const float drot = dpj * sn * sn + dj * c * c;           // the survivor's new pole
            const int typ = tiny ? DW_TINY : (!have ? DW_FIRST : (close ? DW_ROT : DW_PUSH));
            const bool rotd = typ == DW_ROT;
            const bool mine = lane == jj;   // (per-lane selects: position j's record stays in its own lane)
            rec.typ = mine ? typ : rec.typ;
            rec.pold = mine ? pj : rec.pold;
            rec.a = mine ? (rotd ? c : dpj) : rec.a;
            rec.b = mine ? (rotd ? sn : zpj) : rec.b;
            rec.c = mine ? dde : rec.c;
            dpj = tiny ? dpj : (rotd ? drot : dj);
            zpj = tiny ? zpj : (rotd ? tau : zj);
            pj = tiny ? pj : j;
        }
    };
    static_assert(DC_WSM == 5, "one call per register group");
    DwGroup r0, r1, r2, r3, r4;   // (separate objects, no array: no register is indexed by a loop variable)
    chain(0, dr[0], zr[0], r0);
    chain(1, dr[1], zr[1], r1);
    chain(2, dr[2], zr[2], r2);
    chain(3, dr[3], zr[3], r3);
    chain(4, dr[4], zr[4], r4);
    // ---- B: emission by counting.  Before position j: ne deflation events (tiny or rotate; the serial scan fills the
    //      deflated slots from the top, one per event), np pushes (non-deflated slots from the bottom), nr rotations.
    int ne = 0, np = 0, nr = 0;
    auto emit = [&](int m, float dreg, const DwGroup &rec) {
        const int j = lane + 64 * m;
        const bool in = j < nn;
        const bool ev = in && (rec.typ == DW_TINY || rec.typ == DW_ROT), ps = in && rec.typ == DW_PUSH,
                   rt = in && rec.typ == DW_ROT;
        const unsigned long long bev = __ballot(ev), bps = __ballot(ps), brt = __ballot(rt);
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int e = ne + __popcll(bev & below), pu = np + __popcll(bps & below), ro = nr + __popcll(brt & below);
        if (ev) {
            dl[nn - 1 - e] = rt ? rec.c : dreg;
            src[nn - 1 - e] = rt ? rec.pold : j;
        }
        if (rt) {
            DcRot r;
            r.pa = rec.pold;
            r.pb = j;
            r.c = rec.a;
            r.s = rec.b;
            rot[ro] = r;
        }
        if (ps) {
            dl[pu] = rec.a;
            zl[pu] = rec.b;
            src[pu] = rec.pold;
        }
        ne += __popcll(bev);
        np += __popcll(bps);
        nr += __popcll(brt);
    };
    emit(0, dr[0], r0);
    emit(1, dr[1], r1);
    emit(2, dr[2], r2);
    emit(3, dr[3], r3);
    emit(4, dr[4], r4);
    if (lane == 0) {
        int k = np;
        if (pj >= 0) {   // the last survivor
            dl[k] = dpj;
            zl[k] = zpj;
            src[k] = pj;
            ++k;
        }
        *k_out = k;
        *nrot_out = nr;
    }
}

// BLK: the block-structured variant (no zero-fill, masked reads of source columns, merge products per block) -- worth
// its extra registers and bookkeeping at n = 257 (needs the 128 registers of 4 waves per SIMD; the LDS admits 4
// workgroups per CU there anyway); at n <= 129 the plain variant at 5 waves per SIMD is faster (cfg2: 7.5 vs 8.5 ms).
template <int OCC, bool BLK>
__global__ __launch_bounds__(DC_THREADS, OCC) void dc_kernel(int n, const float *__restrict__ dT,
                                                        const float *__restrict__ eT, float *__restrict__ Wbuf,
                                                        float *__restrict__ wout, float *__restrict__ w0out,
                                                        int *__restrict__ logn, int32_t *__restrict__ status,
                                                        unsigned long long *__restrict__ ptime, int rowmajor, int poison,
                                                        int2 *__restrict__ wmap, const int *__restrict__ skip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int64_t bm = blockIdx.x;
    if (skip && skip[bm] == 0) return;   // (uniform) this matrix' G is already there: spectral.hip
    const DcCarve cv(smem, n);
    DcShared &sh = *cv.sh;
    const int NP = cv.NP;
    // developer phase timer (ADMMNET_DC_TIMING=1): cycles of workgroup thread 0 between barriers, accumulated in
    // LDS and flushed once at the end (an atomic per mark would sit in front of the next barrier's vmcnt(0) and
    // charge its own -- contended -- latency to every phase)
    __shared__ unsigned int tacc[64], tstat[24];
    if (ptime && tid < 64) tacc[tid] = 0;
    if (ptime && tid < 24) tstat[tid] = 0;
    long long t_prev = ptime ? clock64() : 0;
    int lvl = 0;
    auto mark = [&](int id) {
        if (ptime && tid == 0) {
            const long long t_now = clock64();
            tacc[id] += (unsigned int)(t_now - t_prev);
            if (id >= 2 && id <= 8) tacc[16 + 8 * min(lvl, 5) + id - 2] += (unsigned int)(t_now - t_prev);
            if (id == 11) tacc[16 + 8 * min(lvl, 5) + 7] += (unsigned int)(t_now - t_prev);   // GEMM of this level
            t_prev = t_now;
        }
    };
    float *lam = cv.lam, *e0 = cv.e0, *zv = cv.zv, *ds = cv.ds, *zs = cv.zs, *un = cv.un, *dl = cv.dl, *zl = cv.zl,
          *tau = cv.tau, *zh = cv.zh, *vals = cv.vals, *lamn = cv.lamn;
    int *perm = cv.perm, *src = cv.src, *org = cv.org, *rnk = cv.rnk, *cidx = cv.cidx;
    DcRot *rot = cv.rot;
    float *leafZ = cv.leafZ, *leafD = cv.leafD;
    (void)zv;

    float *WA = Wbuf + bm * (int64_t)3 * n * n;
    float *WB = WA + (int64_t)n * n;
    float *U = WB + (int64_t)n * n;
    const float *dg = dT + bm * n, *eg = eT + bm * n;

    // ---- leaves
    const int nleaf = dc_leaf_count(n);
    if (tid == 0) {
        for (int b = 0; b < nleaf; ++b) sh.bnd[0][b] = dc_leaf_start(n, nleaf, b);
        sh.bnd[0][nleaf] = n;
        sh.fail = 0;
        sh.nrm = 0;
    }
    for (int i = tid; i < 2 * DC_MAXLEAF; i += DC_THREADS) (&sh.mx[0][0])[i] = 0;
    int bad = 0;
    float amax = 0.f;
    for (int i = tid; i < n; i += DC_THREADS) {
        lam[i] = dg[i];
        e0[i] = (i < n - 1) ? eg[i] : 0.f;
        bad |= !(isfinite(lam[i]) && isfinite(e0[i]));
        amax = fmaxf(amax, fmaxf(fabsf(lam[i]), fabsf(e0[i])));
    }
    // A non-finite tridiagonal (NaN / Inf in the input): report and leave.  torch.linalg.eigh raises on such input, and
    // the rank-by-counting / deflation bookkeeping below would leave permutation slots unwritten and index with them.
    if (__syncthreads_or(bad)) {
        for (int i = tid; i < n; i += DC_THREADS) {
            wout[bm * n + i] = 0.f;
            w0out[bm * n + i] = 0.f;
        }
        for (int64_t i = tid; i < (int64_t)3 * n * n; i += DC_THREADS) WA[i] = 0.f;
        if (wmap)   // the consumer follows the column map whatever happened here: point it at the zeroed first buffer
            for (int i = tid; i < n; i += DC_THREADS) wmap[bm * n + i] = make_int2(i * n, n << 16);
        if (tid == 0) {
            logn[bm * 2 + 1] = 1;
            if (status) atomicAdd(status, 1);
        }
        return;
    }
    // The ping-pong buffers are NOT cleared (0.53 MB of zero writes per matrix at n = 257): every level writes its
    // diagonal blocks in full, and a reader of a source column takes the rows outside that column's own block as zero
    // unless a deflation rotation has filled them (DC_FULL on the column index).  ADMMNET_DC_POISON=1 (tests) fills the
    // buffers with NaN instead, so that any read of a never-written element shows.
    if (!BLK || poison) {
        const float fill = BLK ? __int_as_float(0x7fc00000) : 0.f;   // (plain variant: the buffers start at zero)
        for (int64_t i = tid; i < (int64_t)2 * n * n; i += DC_THREADS) WA[i] = fill;
    }
    // LAPACK sstedc scales T to unit max-norm first (slascl): the deflation tests compare rho |z_j| (z normalised) with
    // 8 eps max(|d|, |z|), which means "negligible against T" only on a matrix of norm ~ 1.  Scaled here by the power
    // of two that brings max(|d|, |e|) into [0.5, 1) -- exact, so a matrix that already is of that size takes the very
    // same arithmetic as before -- and the eigenvalues are scaled back on the way out.
    atomicMax(&sh.nrm, __float_as_int(amax));
    __syncthreads();
    float unscale = 1.f;
    {
        const float orgnrm = __int_as_float(sh.nrm);
        if (orgnrm > 0.f) {
            int ex;
            (void)frexpf(orgnrm, &ex);
            ex = max(-120, min(120, ex));
            const float sc = ldexpf(1.f, -ex);
            unscale = ldexpf(1.f, ex);
            for (int i = tid; i < n; i += DC_THREADS) {
                lam[i] *= sc;
                e0[i] *= sc;
            }
        }
    }
    __syncthreads();
    mark(0);
    // tear: d[k-1] -= |e[k-1]|, d[k] -= |e[k-1]| at every leaf boundary k
    for (int b = 1 + tid; b < nleaf; b += DC_THREADS) {
        const int k = dc_leaf_start(n, nleaf, b);
        const float r = fabsf(e0[k - 1]);
        lam[k - 1] -= r;   // each boundary touches its own two entries (leaves have >= 8 rows)
        lam[k] -= r;
    }
    __syncthreads();
    {   // a team of lanes per leaf: shared scalar recurrence, the rows of Z split over the lanes
        int lt = 16;   // <= 16 lanes: a team never straddles a wave (its lanes run in lock-step)
        while (lt * nleaf > DC_THREADS) lt >>= 1;
        const int leaf = tid / lt, k0 = tid - leaf * lt;
        if (leaf < nleaf) {
            const int a = sh.bnd[0][leaf], s = sh.bnd[0][leaf + 1] - a;
            float *dd = leafD + (size_t)leaf * 2 * DC_MAXLS, *ee = dd + DC_MAXLS;
            const int mr = dc_leaf_maxrows(n, nleaf), ldz = mr | 1;
            float *Z = leafZ + (size_t)leaf * mr * ldz;
            for (int i = 0; i < s; ++i) {   // identical values from every lane of the team
                dd[i] = lam[a + i];
                ee[i] = (i < s - 1) ? e0[a + i] : 0.f;
            }
            auto Zacc = [&](int i, int j) -> float & { return Z[i * ldz + j]; };
            if (leaf_ql(s, dd, ee, Zacc, k0, lt) && k0 == 0) atomicAdd(&sh.fail, 1);
            // ascending order by selection (s <= 15), then emit WT[j][i] = Z(i, idx_j)
            for (int j = 0; j < s; ++j) {
                int best = j;
                for (int q = j + 1; q < s; ++q)
                    if (dd[q] < dd[best]) best = q;
                const float dj = dd[j], db = dd[best];
                dd[j] = db;
                dd[best] = dj;
                for (int i = k0; i < s; i += lt) {
                    const float u = Zacc(i, j);
                    Zacc(i, j) = Zacc(i, best);
                    Zacc(i, best) = u;
                    WA[(int64_t)(a + j) * n + a + i] = Zacc(i, j);
                }
                if (k0 == 0) lam[a + j] = db;
            }
        }
    }
    __syncthreads();
    mark(1);

    // ---- merge levels
    int nblk = nleaf, cb = 0;
    float *Ws = WA, *Wd = WB;
    while (nblk > 1) {
        const int nm = nblk >> 1;                 // merges at this level
        const bool odd = nblk & 1;                // last block passes through
        int ts = DC_THREADS;                      // team size: largest power of two with nm teams
        while (ts * nm > DC_THREADS) ts >>= 1;
        const int team = tid / ts, tl = tid - team * ts;
        const bool act = team < nm;
        const int *bn = sh.bnd[cb];
        int a = 0, b = 0, c = 0;
        if (act) {
            a = bn[2 * team];
            b = bn[2 * team + 1];
            c = bn[2 * team + 2];
        }
        const int nn = c - a, n1 = b - a;
        float rho = 0.f;
        // P1: z, merged order
        if (act) {
            const float beta = e0[b - 1];
            rho = 2.0f * fabsf(beta);
            const float sg = (beta >= 0.f ? 1.f : -1.f) * 0.70710678118654752f;
            for (int i = tl; i < nn; i += ts) {
                const float z = (i < n1) ? Ws[(int64_t)(a + i) * n + (b - 1)] * 0.70710678118654752f
                                         : Ws[(int64_t)(a + i) * n + b] * sg;
                const float v = lam[a + i];
                int r;
                if (i < n1) {
                    r = i;
#pragma unroll 4
                    for (int q = n1; q < nn; ++q) r += (lam[a + q] < v);
                } else {
                    r = i - n1;
#pragma unroll 4
                    for (int q = 0; q < n1; ++q) r += (lam[a + q] <= v);
                }
                perm[a + r] = i;
                ds[a + r] = v;
                zs[a + r] = z;
                atomicMax(&sh.mx[team][0], __float_as_int(fabsf(v)));
                atomicMax(&sh.mx[team][1], __float_as_int(fabsf(z)));
            }
        }
        __syncthreads();
        mark(2);
        // P2: deflation scan, team form (defl_par_* in dc_core.h: flags, chain walkers, emission by counting; the
        //     arrays of the later phases serve as scratch).  A chain that grows into the next run of candidates
        //     is left to the serial scan.
        {
            const float dmx = act ? __int_as_float(sh.mx[team][0]) : 0.f, zmx = act ? __int_as_float(sh.mx[team][1]) : 0.f;
            const float tol = 8.0f * kEps32 * fmaxf(dmx, zmx);
            float *dde = reinterpret_cast<float *>(cidx);
            // (BLK = the n = 257 route: on its layer matrices 60 - 70 % of the merges of the two top levels used to fall out
            //  of the team scan -- one cluster is one run of rotation candidates -- so the wave scan is the primary there;
            //  at n <= 129 the team scan rarely conflicts and is faster (cfg2: 7.7 vs 8.1 ms), the wave scan is its fallback)
            if (BLK && ts >= 64) {   // (uniform) a wave or more per merge: the register-resident scan on the team's first wave
                if (act && tl < 64) {
                    deflate_scan_wave<OCC>(n, team, a, nn, rho, dmx, zmx);
                    if (tl == 0) sh.conf[team] = 0;
                }
            } else {
            if (act) {
                defl_par_flags(tl, ts, nn, rho, tol, ds + a, zs + a, org + a, rnk + a, tau + a, zh + a);
                if (tl == 0) sh.conf[team] = 0;
            }
            __syncthreads();
            if (act)
                defl_par_walk(tl, ts, nn, tol, ds + a, zs + a, org + a, rnk + a, tau + a, zh + a, lamn + a, vals + a,
                              dde + a, &sh.conf[team]);
            __syncthreads();
            if (act) {
                if (sh.conf[team]) {
                    if (ts >= 64) {   // (uniform per level)
                        if (tl < 64) deflate_scan_wave<OCC>(n, team, a, nn, rho, dmx, zmx);
                    } else if (tl == 0) {
                        int k = 0, nr = 0;
                        deflate_scan_tol(nn, rho, dmx, zmx, ds + a, zs + a, dl + a, zl + a, src + a, rot + a, k, nr);
                        sh.kk[team] = k;
                        sh.nrot[team] = nr;
                    }
                } else {
                    defl_par_emit(tl, ts, nn, ds + a, org + a, rnk + a, tau + a, zh + a, lamn + a, vals + a, dde + a,
                                  dl + a, zl + a, src + a, rot + a, &sh.kk[team], &sh.nrot[team]);
                }
            }
            }
        }
        __syncthreads();
        mark(3);
        const int k = act ? sh.kk[team] : 0;
        if (act)
            for (int p = k + tl; p < nn; p += ts) vals[a + p] = dl[a + p];   // eigenvalues of the deflated poles
        if (ptime && act && tl == 0) {   // developer statistics: merge sizes, non-deflated counts, rotations per level
            atomicAdd(&tstat[4 * min(lvl, 5) + 0], (unsigned int)nn);
            atomicAdd(&tstat[4 * min(lvl, 5) + 1], (unsigned int)k);
            atomicAdd(&tstat[4 * min(lvl, 5) + 2], (unsigned int)sh.nrot[team]);
            atomicAdd(&tstat[4 * min(lvl, 5) + 3], 1u);
            if (sh.conf[team]) atomicAdd(&tacc[56 + min(lvl, 5)], 1u);   // merges left to the serial scan
        }
        // P3: deflation rotations on the source columns (thread-private rows i) + secular roots
        if (act) {
            // A chain of rotations (pa, pb) hands column pb on as the next pa: each thread keeps its
            // rows of that running column in a register, so the chain never round-trips through memory;
            // the other operand of every rotation is a column no earlier rotation touched, loaded
            // DC_RA rotations ahead.
            const int nr = sh.nrot[team];
            for (int i = tl; i < nn; i += ts) {
                float carry = 0.f;
                int cpb = -1;
                for (int r0 = 0; r0 < nr; r0 += DC_RA) {
                    DcRot rr[DC_RA];
                    float xv[DC_RA], yv[DC_RA];
                    int xo[DC_RA], yo[DC_RA];   // 32-bit element offsets: registers are what bounds the occupancy here
#pragma unroll
                    for (int q = 0; q < DC_RA; ++q) {
                        rr[q] = rot[a + min(r0 + q, nr - 1)];
                        // (a column is read from memory before any rotation has written it: its rows outside its own
                        //  block are zero by definition, not by content; the flag may be set concurrently, hence masked)
                        const int ca = perm[a + rr[q].pa] & ~DC_FULL, cb2 = perm[a + rr[q].pb] & ~DC_FULL;
                        xo[q] = (a + ca) * n + a + i;
                        yo[q] = (a + cb2) * n + a + i;
                        xv[q] = (!BLK || (ca < n1) == (i < n1)) ? Ws[xo[q]] : 0.f;
                        yv[q] = (!BLK || (cb2 < n1) == (i < n1)) ? Ws[yo[q]] : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < DC_RA; ++q) {
                        if (r0 + q < nr) {
                            const float xi = (rr[q].pa == cpb) ? carry : xv[q], yi = yv[q];
                            Ws[xo[q]] = rr[q].c * xi + rr[q].s * yi;
                            carry = rr[q].c * yi - rr[q].s * xi;
                            Ws[yo[q]] = carry;
                            cpb = rr[q].pb;
                        }
                    }
                }
            }
            // both columns of a rotation now hold all nn rows (written above for every i)
            if constexpr (BLK) {
                for (int r = tl; r < nr; r += ts) {
                    atomicOr(&perm[a + rot[a + r].pa], DC_FULL);
                    atomicOr(&perm[a + rot[a + r].pb], DC_FULL);
                }
            }
            {   // when the team has two lanes per root, adjacent lanes share one: each sums every other pole
                // and one DPP swap adds the halves (one code path: G = 1 makes the swap a no-op)
                const int G = (2 * k <= ts) ? 2 : 1;
                const int sub = (G == 2) ? (tl & 1) : 0;
                auto red = [G](float x) {
                    const float y = __builtin_bit_cast(
                        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
                    return G == 2 ? x + y : x;
                };
                for (int j = (G == 2) ? (tl >> 1) : tl; j < k; j += ts / G) {
                    int o;
                    float t;
                    secular_root(k, j, rho, dl + a, zl + a, o, t, nullptr, sub, G, red);
                    if (sub == 0) {
                        org[a + j] = o;
                        tau[a + j] = t;
                        vals[a + j] = dl[a + o] + t;
                    }
                }
            }
        }
        __syncthreads();
        mark(4);
        // P4: Loewner z-hat, final (ascending, stable) positions
        if (act) {
            {   // two adjacent lanes per pole while the team has them (as for the roots): half the serial product each
                const int G = (2 * k <= ts) ? 2 : 1;
                const int sub = (G == 2) ? (tl & 1) : 0;
                auto redm = [G](float x) {
                    const float y = __builtin_bit_cast(
                        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
                    return G == 2 ? x * y : x;
                };
                for (int i = (G == 2) ? (tl >> 1) : tl; i < k; i += ts / G) {
                    const float v = lowner_zhat(k, i, dl + a, zl + a, org + a, tau + a, sub, G, redm);
                    if (sub == 0) zh[a + i] = v;
                }
            }
            for (int p = tl; p < nn; p += ts) {
                const float v = vals[a + p];
                int r = 0;
#pragma unroll 4
                for (int q = 0; q < nn; ++q) {
                    const float u = vals[a + q];
                    r += (u < v) || (u == v && q < p);
                }
                rnk[a + p] = r;
                lamn[a + r] = v;
                cidx[a + p] = perm[a + src[a + p]];
            }
        }
        __syncthreads();
        mark(5);
        // P5: norms of the eigenvectors of the rank-one update, u_j = (zh_i / (d_i - lam_j))_i.  The
        //     vectors themselves are regenerated inside the GEMM (two subtractions, a reciprocal and two
        //     products per entry, hidden under the MFMA) instead of making a round trip through memory.
        if (act) {
            // The merged eigenvector block is [W1 0; 0 W2] U: the rows of the first block only see the source columns
            // of the first block (and the columns a deflation rotation filled), likewise the second -- as slaed3's two
            // GEMMs.  kl1 / kl2 list the non-deflated merged positions by that criterion, in ascending order (placed by
            // counting: deterministic), in the memory of the rotation list (dead since P3).
            if constexpr (BLK) {
                int *kl1 = reinterpret_cast<int *>(rot) + a, *kl2 = kl1 + NP;
                for (int p = tl; p < k; p += ts) {
                    const int cc = cidx[a + p];
                    const bool t1 = (cc & DC_FULL) || (cc & ~DC_FULL) < n1, t2 = (cc & DC_FULL) || (cc & ~DC_FULL) >= n1;
                    int r1 = 0, r2 = 0;
#pragma unroll 4
                    for (int q = 0; q < p; ++q) {
                        const int cq = cidx[a + q];
                        const bool f = cq & DC_FULL, lo = (cq & ~DC_FULL) < n1;
                        r1 += (f || lo);
                        r2 += (f || !lo);
                    }
                    if (t1) kl1[r1] = p;
                    if (t2) kl2[r2] = p;
                    if (p == k - 1) {
                        sh.kc[team][0] = r1 + (t1 ? 1 : 0);
                        sh.kc[team][1] = r2 + (t2 ? 1 : 0);
                    }
                }
                if (k == 0 && tl == 0) sh.kc[team][0] = sh.kc[team][1] = 0;
            }
            const int G = (2 * k <= ts) ? 2 : 1;   // two lanes per root: each sums every other entry
            const int sub = (G == 2) ? (tl & 1) : 0;
            for (int j = (G == 2) ? (tl >> 1) : tl; j < k; j += ts / G) {
                float nrm = 0.f;
                const float dorg = dl[a + org[a + j]], tj = tau[a + j];
#pragma unroll 4
                for (int i = sub; i < k; i += G) {
                    const float u = fdiv_fast(zh[a + i], (dl[a + i] - dorg) - tj);
                    nrm = fmaf(u, u, nrm);
                }
                if (G == 2)
                    nrm += __builtin_bit_cast(
                        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, nrm), 0xB1, 0xF, 0xF, false));
                un[a + j] = 1.0f / sqrtf(nrm);
            }
        }
        __syncthreads();
        mark(6);
        // P6: new blocks.  Deflated columns are copied, the others come from the GEMM
        //     WTdst[rank(j)][i] = sum_kk U[kk][j] * WTsrc[col(src[kk])][i]   (MFMA, waves take tiles)
        // The top-level merge with a column map (D = 256 route): its deflated columns -- 230 of 257 on the layer matrices
        // -- stay where they are (rotated in place above), the products land in the other buffer as always, and the
        // consumer is told where eigenvector j lives and which of its rows exist (a column no rotation filled has rows
        // in its own block only; the others are zero by definition, not by content).
        const bool mapped = wmap != nullptr && nblk == 2;
        if (mapped) {
            int2 *wm = wmap + bm * n;
            for (int p = tl; p < nn; p += ts) {
                const int rk = rnk[a + p];
                if (p < k) {
                    wm[rk] = make_int2((int)(Wd - WA) + (a + rk) * n + a, nn << 16);
                } else {
                    const int cc = cidx[a + p], col = cc & ~DC_FULL;
                    const bool full = !BLK || (cc & DC_FULL);
                    const int lo = full ? 0 : (col < n1 ? 0 : n1), hi = full ? nn : (col < n1 ? n1 : nn);
                    wm[rk] = make_int2((int)(Ws - WA) + (a + col) * n + a, lo | (hi << 16));
                }
            }
        }
        if (act && !mapped) {
            const int cw = ts >= 32 ? 32 : ts;   // lanes per column copy
            for (int p = k + (tl / cw); p < nn; p += max(1, ts / cw)) {
                const int cc = cidx[a + p], col = cc & ~DC_FULL;
                const bool full = cc & DC_FULL;
                const float *xs = Ws + (int64_t)(a + col) * n + a;
                float *xd = Wd + (int64_t)(a + rnk[a + p]) * n + a;
                // nine loads in flight per lane: one trip per column at n = 257 (the loop is bound by the L2 round trip,
                // and most columns of the layer matrices deflate -- 230 of 257 at the top level)
                for (int i = tl & (cw - 1); i < nn; i += 9 * cw) {
                    float v[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        const int iq = i + q * cw;
                        v[q] = (iq < nn && (!BLK || full || ((col < n1) == (iq < n1)))) ? xs[iq] : 0.f;
                    }
#pragma unroll
                    for (int q = 0; q < 9; ++q)
                        if (i + q * cw < nn) xd[i + q * cw] = v[q];
                }
            }
        }
        if (odd) {   // pass the unpaired block through
            const int pa = bn[nblk - 1], pc = bn[nblk];
            const int w = pc - pa;
            for (int idx = tid; idx < w * w; idx += DC_THREADS) {
                const int j = idx / w, i = idx - j * w;
                Wd[(int64_t)(pa + j) * n + pa + i] = Ws[(int64_t)(pa + j) * n + pa + i];
            }
            for (int i = tid; i < w; i += DC_THREADS) lamn[pa + i] = lam[pa + i];
        }
        mark(10);
        dc_level_gemm<OCC, BLK>(n, nm, cb, Ws, Wd);
        mark(11);
        __syncthreads();
        mark(7);
        // P7: commit eigenvalues and block boundaries
        for (int i = tid; i < n; i += DC_THREADS) lam[i] = lamn[i];
        for (int i = tid; i < 2 * DC_MAXLEAF; i += DC_THREADS) (&sh.mx[0][0])[i] = 0;
        if (tid == 0) {
            int o = 0;
            for (int q = 0; q < nm; ++q) sh.bnd[cb ^ 1][o++] = bn[2 * q];
            if (odd) sh.bnd[cb ^ 1][o++] = bn[nblk - 1];
            sh.bnd[cb ^ 1][o] = n;
        }
        __syncthreads();
        mark(8);
        nblk = nm + (odd ? 1 : 0);
        cb ^= 1;
        ++lvl;
        float *tmp = Ws;
        Ws = Wd;
        Wd = tmp;
    }
    // ---- outputs: eigenvalues (ascending), first row of W, status, and -- unless the consumer reads the
    //      transposed image itself (backrebuild.hip; dc_final_offset() tells it which ping-pong buffer) -- W
    //      row-major (W[i][j], j = eigenvalue) in the third region: the orientation vgemm_kernel reads coalesced
    const bool mapped_out = wmap != nullptr && nleaf > 1;   // (Ws / Wd were swapped after the last level)
    for (int i = tid; i < n; i += DC_THREADS) {
        wout[bm * n + i] = lam[i] * unscale;
        if (mapped_out) {
            const int2 e = wmap[bm * n + i];
            w0out[bm * n + i] = ((e.y & 0xffff) == 0) ? WA[e.x] : 0.f;   // row 0 exists iff the column's rows start at 0
        } else {
            w0out[bm * n + i] = Ws[(int64_t)i * n];
        }
    }
    if (rowmajor) {
        float *tile = leafZ;   // 32 x 33 floats (the leaf scratch is dead by now)
        const int tt = (n + 31) >> 5;
        const int lx = tid & 31, ly = tid >> 5;   // 32 x 8 threads
        for (int t = 0; t < tt * tt; ++t) {
            const int j0 = (t / tt) * 32, i0 = (t % tt) * 32;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jj = ly + 8 * q;
                tile[jj * 33 + lx] = (j0 + jj < n && i0 + lx < n) ? Ws[(int64_t)(j0 + jj) * n + i0 + lx] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ii = ly + 8 * q;
                if (i0 + ii < n && j0 + lx < n) U[(int64_t)(i0 + ii) * n + j0 + lx] = tile[lx * 33 + ii];
            }
        }
    }
    mark(9);
    if (ptime) {
        __syncthreads();
        if (tid < 64 && tacc[tid]) atomicAdd(&ptime[tid], (unsigned long long)tacc[tid]);
        if (tid < 24 && tstat[tid]) atomicAdd(&ptime[64 + tid], (unsigned long long)tstat[tid]);
    }
    if (tid == 0) {
        logn[bm * 2 + 0] = 0;
        logn[bm * 2 + 1] = sh.fail ? 1 : 0;
        if (sh.fail && status) atomicAdd(status, 1);
    }
}

// K3': VT[c][rho] = sum_r W[1 + r][c] * QT[r][rho]   (V = diag(1, Q') W in the planar transposed layout)
//   W: [n][n] row-major from dc_kernel (third buffer), QT: [D][2D] from the tridiagonalisation,
//   VT: [n][2D].  One workgroup (4 waves) per matrix; the 32 x 32 output tiles are dealt round-robin
//   to the waves; both operands are read straight from L2 in the MFMA lane layout (coalesced rows).
__global__ __launch_bounds__(256) void vgemm_kernel(int D, const float *__restrict__ Wbuf,
                                                    const float *__restrict__ QT, float *__restrict__ VT) {
    const int n = D + 1;
    const int64_t bm = blockIdx.x;
    const float *Wr = Wbuf + bm * (int64_t)3 * n * n + (int64_t)2 * n * n;
    const float *Q = QT + bm * ((int64_t)n * 2 * D);
    float *V = VT + bm * ((int64_t)n * 2 * D);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, kh = lane >> 5;
    const int tm = (n + 31) >> 5, tn = (2 * D + 31) >> 5;
    for (int t = wave; t < tm * tn; t += 4) {
        const int c0 = 32 * (t / tn), p0 = 32 * (t % tn);
        const bool cv = (c0 + r) < n, pv = (p0 + r) < 2 * D;
        const int co = cv ? c0 + r : 0, po = pv ? p0 + r : 0;
        f32x16 acc = {0};
#pragma unroll 8
        for (int k0 = 0; k0 < D; k0 += 2) {
            const int rr = k0 + kh;
            const bool kv = rr < D;
            const int rc = kv ? rr : 0;
            float av = Wr[(int64_t)(1 + rc) * n + co];
            float bv = Q[(int64_t)rc * 2 * D + po];
            av = (kv && cv) ? av : 0.f;
            bv = (kv && pv) ? bv : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = c0 + (q & 3) + 8 * (q >> 2) + 4 * kh;
            if (c < n && pv) V[(int64_t)c * 2 * D + p0 + r] = acc[q];
        }
    }
}

size_t dc_lds_bytes(int n) {
    const int NP = (n + 3) & ~3;
    const size_t nleaf = (size_t)dc_leaf_count(n);
    size_t leaf = dc_leafz_floats(n) + nleaf * 2 * DC_MAXLS;
    if (leaf < 32 * 33) leaf = 32 * 33;   // the final transpose reuses the leaf scratch as a tile
    return ((sizeof(DcShared) + 15) & ~(size_t)15) + sizeof(float) * 11 * NP + sizeof(int) * 5 * NP + sizeof(DcRot) * NP +
           sizeof(float) * leaf;
}

// float offset (inside one matrix' 3 n^2 block) of the ping-pong buffer that holds the final WT[j][i]: the
// merge loop swaps buffers once per level
int64_t dc_final_offset(int n) {
    int nblk = dc_leaf_count(n), levels = 0;
    while (nblk > 1) {
        nblk = (nblk >> 1) + (nblk & 1);
        ++levels;
    }
    return (levels & 1) ? (int64_t)n * n : 0;
}

int launch_dc(int n, int64_t nb, const Ws &ws, int32_t *status, hipStream_t st, bool rowmajor, bool colmap) {
    ProfScope _prof(KC_TQL, st);
    if (nb <= 0) return ADMMNET_OK;
    if (n / DC_LS > DC_MAXLEAF) {
        set_error("dc: n=%d unsupported", n);
        return ADMMNET_E_ARG;
    }
    const size_t lds = dc_lds_bytes(n);
    static const int env_occ = getenv("ADMMNET_DC_OCC") ? atoi(getenv("ADMMNET_DC_OCC")) : 0;   // tuning knob
    const bool blk = n > 129 && !(getenv("ADMMNET_DC_BLOCKS") && !strcmp(getenv("ADMMNET_DC_BLOCKS"), "0"));
    const int occ = env_occ > 0 ? env_occ : (blk ? 4 : 5);
    auto kern = blk ? (occ >= 8 ? dc_kernel<8, true> : occ == 6 ? dc_kernel<6, true> : occ == 5 ? dc_kernel<5, true> : dc_kernel<4, true>)
                    : (occ >= 8 ? dc_kernel<8, false> : occ == 6 ? dc_kernel<6, false> : occ == 4 ? dc_kernel<4, false> : dc_kernel<5, false>);
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    static const bool timing = getenv("ADMMNET_DC_TIMING") != nullptr;   // developer aid, never on by default
    static const bool poison = getenv("ADMMNET_DC_POISON") != nullptr;   // tests: NaN in every never-written element
    unsigned long long *ptime = nullptr;
    if (timing) {
        ADMM_HIP(hipMalloc(&ptime, 96 * sizeof(unsigned long long)));
        ADMM_HIP(hipMemsetAsync(ptime, 0, 96 * sizeof(unsigned long long), st));
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(DC_THREADS), lds, st, n, ws.dT, ws.eT, ws.Wdc, ws.w,
                       ws.w0, ws.logn, status, ptime, rowmajor ? 1 : 0, poison ? 1 : 0,
                       (colmap && ws.Wmap && dc_leaf_count(n) > 1) ? ws.Wmap : nullptr, ws.skip);
    ADMM_HIP(hipGetLastError());
    if (timing) {
        unsigned long long h[96];
        ADMM_HIP(hipMemcpyAsync(h, ptime, sizeof(h), hipMemcpyDeviceToHost, st));
        ADMM_HIP(hipStreamSynchronize(st));
        ADMM_HIP(hipFree(ptime));
        static const char *nm[12] = {"init", "leaves", "P1 sort", "P2 scan", "P3 rot+secular", "P4 zhat+rank",
                                     "P5 U", "P6 barrier wait", "P7 commit", "final transpose", "P6 copy", "P6 gemm"};
        fprintf(stderr, "[dc timing] n=%d nb=%lld  mean cycles per workgroup:\n", n, (long long)nb);
        for (int i = 0; i < 12; ++i) fprintf(stderr, "   %-16s %10.0f\n", nm[i], (double)h[i] / (double)nb);
        for (int l = 0; l < 6; ++l)
            if (h[64 + 4 * l + 3])
                fprintf(stderr, "   level %d merges: mean size %.1f, non-deflated %.1f, rotations %.1f, left to the serial scan %.3f\n", l,
                        (double)h[64 + 4 * l] / h[64 + 4 * l + 3], (double)h[64 + 4 * l + 1] / h[64 + 4 * l + 3],
                        (double)h[64 + 4 * l + 2] / h[64 + 4 * l + 3], (double)h[56 + l] / h[64 + 4 * l + 3]);
        for (int l = 0; l < 6; ++l) {
            fprintf(stderr, "   level %d:", l);
            for (int q = 0; q < 8; ++q) fprintf(stderr, " %8.0f", (double)h[16 + 8 * l + q] / (double)nb);   // P1..P7, GEMM
            fprintf(stderr, "\n");
        }
    }
    return ADMMNET_OK;
}

int launch_vgemm(int D, int64_t nb, const Ws &ws, hipStream_t st) {
    ProfScope _prof(KC_ROTAPPLY, st);
    if (nb <= 0) return ADMMNET_OK;
    hipLaunchKernelGGL(vgemm_kernel, dim3((unsigned)nb), dim3(256), 0, st, D, ws.Wdc, ws.QV, ws.VT);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
