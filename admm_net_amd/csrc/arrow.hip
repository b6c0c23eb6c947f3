// arrow.hip -- first G-layer (Z = 0): direct eigensolver for the Hermitian arrowhead
//     A = C = [[diag h, phi], [phi^H, corner]]          (/root/reference/admm_net.py:273-288)
// fused with the rebuild G = V f(Lambda) V^H + ||G - C||_F (admm_net.py:303-354, 400-403, 454), D <= 128.
//
// The dense path (tridiagonalise, divide & conquer, back-transform) costs O(n^3) per matrix; an
// arrowhead's eigenpairs follow from one secular equation in O(n^2) (arrow_core.h).  One 256-thread
// workgroup per matrix:
//   P1 sort h (rank counting), |phi| and phases          P5 zeta-hat (Loewner), ranks of all eigenvalues
//   P2 deflation scan (one thread, as in the D&C merge)  P6 norms, eigenvalue map f(lambda)
//   P4 secular roots, one thread per root                P7 eigenvectors written straight into LDS as VT[c][rho]
//   P8 deflation rotations undone, phases applied        then rebuild_from_lds (shared with backrebuild.hip)
// V never exists in memory; the only HBM traffic is phi, h in and G out.
#include <cstdio>
#include <cstdlib>

#include "common.h"

#include "arrow_core.h"
#include "rebuild_lds.h"

namespace admmnet {

constexpr int AR_THREADS = 256;

struct ArShared {
    int k, nrot, conf;
    int mx[2];
    float znorm;
};

__host__ __device__ inline size_t ar_np(int D) { return (size_t)((D + 1 + 3) & ~3); }
// float arrays of length NP: hraw zraw phr phim ds zs dl zl tau zh vals x0 (12) ; int arrays: perm src org rnk kidx (5)
__host__ __device__ inline size_t ar_lds_bytes(const BrGeom &g) {
    const size_t NP = ar_np(g.D);
    return sizeof(float) * (g.vt_floats() + g.small_floats() + 12 * NP) + sizeof(int) * 5 * NP + sizeof(DcRot) * NP;
}

// BIG (128 < D <= 256): the eigenvector image does not fit the LDS; it is built in the global VT buffer of the
// dense path (rows c, planes at 0 / D, pitch 2 D) together with w and w0, and rebuild.hip's kernel consumes it.
template <bool BIG>
__global__ __launch_bounds__(AR_THREADS, 1) void arrow_rebuild_kernel(
    int D, const float *__restrict__ lw, const float2 *__restrict__ phi, const float *__restrict__ h,
    float2 *__restrict__ G, float *__restrict__ rn, float *__restrict__ w_out, int32_t *__restrict__ status,
    unsigned long long *__restrict__ ptime, int lower_only, float *VTg, float *__restrict__ w0g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ ArShared sh;
    // developer phase timer (ADMMNET_AR_TIMING=1): cycles of thread 0 between marks
    long long t_prev = ptime ? clock64() : 0;
    auto mark = [&](int id) {
        if (ptime && threadIdx.x == 0) {
            const long long t_now = clock64();
            atomicAdd(&ptime[id], (unsigned long long)(t_now - t_prev));
            t_prev = t_now;
        }
    };
    const BrGeom g(D);
    const int n = g.n;
    const int Dp = BIG ? D : g.Dp, VP = BIG ? 2 * D : g.VP;   // plane width / row pitch of the eigenvector image
    const int NP = (int)ar_np(D);
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const float alpha = lw[S_CORNER_G];   // corner of C = 1 / (lambda^2 + eps), admm_net.py:271
    float *VTl = BIG ? VTg + b * ((int64_t)n * 2 * D) : reinterpret_cast<float *>(smem);
    float *fs = reinterpret_cast<float *>(smem) + (BIG ? 0 : g.vt_floats());
    float *w0f = fs + ((n + 4) & ~3);
    float *z0s = w0f + ((n + 4) & ~3);
    float *rowb = z0s + ((n + 4) & ~3);
    float *redb = rowb + 2 * Dp;
    float *hraw = redb + 8;
    float *zraw = hraw + NP;
    float *phr = zraw + NP;
    float *phim = phr + NP;
    float *ds = phim + NP;
    float *zs = ds + NP;
    float *dl = zs + NP;
    float *zl = dl + NP;
    float *tau = zl + NP;
    float *zh = tau + NP;
    float *vals = zh + NP;
    float *x0 = vals + NP;
    float *lamd = zs;   // (after the deflation scan) d[org_j]: origin pole value of root j
    int *perm = reinterpret_cast<int *>(x0 + NP);
    int *src = perm + NP;
    int *org = src + NP;
    int *rnk = org + NP;
    int *kidx = rnk + NP;
    DcRot *rot = reinterpret_cast<DcRot *>(kidx + NP);

    // ---- P0: load, moduli and phases
    if (tid == 0) {
        sh.mx[0] = __float_as_int(fabsf(alpha));
        sh.mx[1] = 0;
    }
    int bad = 0;   // non-finite input or eigenvalue: reported through status[0] (torch.linalg.eigh raises on such input)
    for (int i = tid; i < D; i += AR_THREADS) {
        const float2 z = phi[b * D + i];
        const float a = sqrtf(z.x * z.x + z.y * z.y);
        const float ia = a > 0.f ? 1.0f / a : 0.f;
        hraw[i] = h[b * D + i];
        bad |= !(isfinite(a) && isfinite(hraw[i]));
        zraw[i] = a;
        phr[i] = a > 0.f ? z.x * ia : 1.f;
        phim[i] = a > 0.f ? z.y * ia : 0.f;
    }
    if (__syncthreads_or(bad)) {   // NaN / Inf input: the rank sort below would leave permutation slots unwritten
        if (tid == 0 && status) atomicAdd(status, 1);
        return;
    }
    // ---- P1: sort h ascending (stable rank counting)
    for (int i = tid; i < D; i += AR_THREADS) {
        const float v = hraw[i];
        int r = 0;
#pragma unroll 8
        for (int q = 0; q < D; ++q) {
            const float u = hraw[q];
            r += (u < v) || (u == v && q < i);
        }
        perm[r] = i;
        ds[r] = v;
        zs[r] = zraw[i];
        atomicMax(&sh.mx[0], __float_as_int(fabsf(v)));
        atomicMax(&sh.mx[1], __float_as_int(zraw[i]));
    }
    __syncthreads();
    // ---- P2: deflation.  Usual case: nothing deflates (distinct h, non-zero phi) -- checked in parallel;
    //      otherwise the serial scan of the D&C merge (one thread)
    {
        const float dmax = __int_as_float(sh.mx[0]), zmax = __int_as_float(sh.mx[1]);
        int trig = 0;
        for (int p = tid; p < D; p += AR_THREADS) trig |= deflate_triggers(p, 1.0f, dmax, zmax, ds, zs) ? 1 : 0;
        if (__syncthreads_or(trig)) {   // team form of the scan (dc_core.h), later phases' arrays as scratch
            const float tol = 8.0f * kEps32 * fmaxf(dmax, zmax);
            float *dde = reinterpret_cast<float *>(kidx);
            defl_par_flags(tid, AR_THREADS, D, 1.0f, tol, ds, zs, org, rnk, tau, zh);
            if (tid == 0) sh.conf = 0;
            __syncthreads();
            defl_par_walk(tid, AR_THREADS, D, tol, ds, zs, org, rnk, tau, zh, vals, x0, dde, &sh.conf);
            __syncthreads();
            if (sh.conf) {   // a rotation chain grew into the next run of candidates: serial scan
                if (tid == 0) {
                    int k = 0, nr = 0;
                    deflate_scan_tol(D, 1.0f, dmax, zmax, ds, zs, dl, zl, src, rot, k, nr);
                    sh.k = k;
                    sh.nrot = nr;
                }
            } else {
                defl_par_emit(tid, AR_THREADS, D, ds, org, rnk, tau, zh, vals, x0, dde, dl, zl, src, rot, &sh.k, &sh.nrot);
            }
        } else {
            for (int p = tid; p < D; p += AR_THREADS) {
                dl[p] = ds[p];
                zl[p] = zs[p];
                src[p] = p;
            }
            if (tid == 0) {
                sh.k = D;
                sh.nrot = 0;
            }
        }
    }
    __syncthreads();
    mark(0);
    const int k = sh.k, nrot = sh.nrot;
    // slots: 0..k roots, p + 1 for the deflated pole at scan position p in [k, D)
    for (int p = k + tid; p < D; p += AR_THREADS) vals[p + 1] = dl[p];
    for (int p = tid; p < D; p += AR_THREADS) kidx[src[p]] = (p < k) ? p : -(p + 1) - 1;   // sorted position -> pole / slot
    if (tid < 64) {   // ||zeta|| of the surviving poles
        float s = 0.f;
        for (int i = tid; i < k; i += 64) s = fmaf(zl[i], zl[i], s);
        s = wave_sum(s);
        if (tid == 0) sh.znorm = sqrtf(s);
    }
    __syncthreads();
    // ---- P4: secular roots
    if (k == 0) {
        if (tid == 0) vals[0] = alpha;
    } else {
        const float znorm = sh.znorm;
        auto put = [&](int jr, int o, float t) {
            org[jr] = o;
            tau[jr] = t;
            lamd[jr] = dl[o];
            vals[jr] = dl[o] + t;
        };
        // Lanes per root: two adjacent lanes (each sums every other pole) while the roots fit, one otherwise; roots
        // beyond the lanes' first pass -- root 128 of D = 128, root 256 of D = 256 -- are solved by the whole of
        // wave 0 (each lane two to four poles) instead of a second, nearly empty pass.
        const int first = (k + 1 <= AR_THREADS / 2 + 1) ? min(k + 1, AR_THREADS / 2) : min(k + 1, AR_THREADS);
        if (first <= AR_THREADS / 2) {
            const int jr = tid >> 1, sub = tid & 1;
            if (jr < first) {
                int o;
                float t;
                arrow_root(k, jr, alpha, znorm, dl, zl, o, t, nullptr, sub, 2, [](float x) {
                    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1,
                                                                                       0xF, 0xF, false));   // quad_perm [1,0,3,2]
                });
                if (sub == 0) put(jr, o, t);
            }
        } else if (tid < first) {
            int o;
            float t;
            arrow_root(k, tid, alpha, znorm, dl, zl, o, t);
            put(tid, o, t);
        }
        if (tid < 64) {
            for (int jr = first; jr <= k; ++jr) {
                int o;
                float t;
                arrow_root(k, jr, alpha, znorm, dl, zl, o, t, nullptr, tid, 64, [](float x) { return wave_sum(x); });
                if (tid == 0) put(jr, o, t);
            }
        }
    }
    __syncthreads();
    for (int p = tid; p < n; p += AR_THREADS) bad |= !isfinite(vals[p]);
    mark(1);
    // ---- P5: zeta-hat, final (ascending, stable) positions of all n eigenvalues
    {   // two adjacent lanes per pole (k <= 128): half of the serial product each
        const int G = (2 * k <= AR_THREADS) ? 2 : 1;
        const int sub = (G == 2) ? (tid & 1) : 0;
        auto redm = [G](float x) {
            const float y = __builtin_bit_cast(
                float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));
            return G == 2 ? x * y : x;
        };
        for (int i = (G == 2) ? (tid >> 1) : tid; i < k; i += AR_THREADS / G) {
            const float v = arrow_zhat(k, i, dl, lamd, tau, sub, G, redm);
            if (sub == 0) zh[i] = v;
        }
    }
    for (int s = tid; s < n; s += AR_THREADS) {
        const float v = vals[s];
        int r = 0;
#pragma unroll 8
        for (int q = 0; q < n; ++q) {
            const float u = vals[q];
            r += (u < v) || (u == v && q < s);
        }
        rnk[s] = r;
    }
    __syncthreads();
    // ---- P6: norms; eigenvalue map and arrow-row entries in final order
    {
        const LayerLayout L{D};
        const float thr = lw[S_THR];
        const float *vn = lw + L.off_vn();
        for (int s = tid; s < n; s += AR_THREADS) {
            float xa = 0.f;   // arrow component of eigenvector s
            if (s <= k) {
                float nrm = 1.f;
                if (k > 0) {
                    const float dorg = lamd[s], ts = tau[s];
#pragma unroll 4
                    for (int i = 0; i < k; ++i) {
                        const float v = fdiv_fast(zh[i], (dorg - dl[i]) + ts);   // zhat_i / (lam_s - d_i)
                        nrm = fmaf(v, v, nrm);
                    }
                }
                xa = 1.0f / sqrtf(nrm);
            }
            x0[s] = xa;
            const int c = rnk[s];
            const float lam = vals[s];
            const float f = br_eig_map(lam, thr, vn);
            fs[c] = f;
            w0f[c] = xa * f;
            z0s[c] = xa;
            if (w_out) w_out[b * n + c] = lam;
            if (BIG) w0g[b * n + c] = xa;
        }
        if (tid == 0) {
            fs[n] = 0.f;
            w0f[n] = 0.f;
            z0s[n] = 0.f;
        }
    }
    __syncthreads();
    mark(2);
    // ---- P7: eigenvectors in the rotated real basis, straight into VT[c][column of ORIGINAL index]
    //      thread = original index i (both planes' padding columns are zeroed as well)
    int *ipos = reinterpret_cast<int *>(hraw);   // inverse permutation (hraw is dead since P1)
    for (int p = tid; p < D; p += AR_THREADS) ipos[perm[p]] = p;
    __syncthreads();
    // (when the plane is narrower than the workgroup, AR_THREADS / Dp threads share a column and split its
    //  entries s: at D = 128 two threads per column instead of 128 idle ones)
    const int nparts = (AR_THREADS / Dp > 0) ? AR_THREADS / Dp : 1;
    const int sper = (n + nparts - 1) / nparts;
    for (int it = tid; it < Dp * nparts; it += AR_THREADS) {
        const int i = it % Dp, part = it / Dp;
        const int s0 = part * sper, s1 = min(n, s0 + sper);
        float *colr = VTl + i, *coli = VTl + Dp + i;
        if (i >= D) {
            for (int c = s0; c < s1; ++c) {
                colr[c * VP] = 0.f;
                coli[c * VP] = 0.f;
            }
            continue;
        }
        const int kd = kidx[ipos[i]];
        if (kd >= 0) {   // surviving pole kd: component zhat x0_s / (lam_s - d) in every root's eigenvector
            const float zi = zh[kd], di = dl[kd];
#pragma unroll 4
            for (int s = s0; s < min(s1, k + 1); ++s) {
                const float v = fdiv_fast(zi, (lamd[s] - di) + tau[s]) * x0[s];
                colr[rnk[s] * VP] = v;
            }
            for (int s = max(s0, k + 1); s < s1; ++s) colr[rnk[s] * VP] = 0.f;
        } else {         // deflated pole: unit vector of slot -(kd) - 1
            const int slot = -kd - 1;
#pragma unroll 4
            for (int s = s0; s < s1; ++s) colr[rnk[s] * VP] = (s == slot) ? 1.f : 0.f;
        }
    }
    __syncthreads();
    // ---- P8: undo the deflation rotations (reverse order, v = G^T v'), thread = eigenvector row c
    for (int c = tid; c < n; c += AR_THREADS) {
        float *row = VTl + c * VP;
        for (int r = nrot - 1; r >= 0; --r) {
            const DcRot rr = rot[r];
            const int ia = perm[rr.pa], ib = perm[rr.pb];
            const float a = row[ia], bb = row[ib];
            row[ia] = rr.c * a - rr.s * bb;
            row[ib] = rr.s * a + rr.c * bb;
        }
    }
    __syncthreads();
    // phases: (re, im) planes
    for (int it = tid; it < Dp * nparts; it += AR_THREADS) {
        const int i = it % Dp, part = it / Dp;
        if (i >= D) continue;
        float *colr = VTl + i, *coli = VTl + Dp + i;
        const float pr = phr[i], pi = phim[i];
#pragma unroll 8
        for (int c = part * sper; c < min(n, (part + 1) * sper); ++c) {
            const float x = colr[c * VP];
            colr[c * VP] = x * pr;
            coli[c * VP] = x * pi;
        }
    }
    __syncthreads();
    mark(3);
    if (__syncthreads_or(bad) && tid == 0 && status) atomicAdd(status, 1);
    if constexpr (BIG) return;
    else rebuild_from_lds(g, b, lw, VTl, fs, w0f, z0s, rowb, redb, phi, h, G, rn, [&](int id) { mark(id); }, lower_only);
}

bool arrow_rebuild_supported(int D) { return D >= 1 && D <= 256; }

int launch_arrow_rebuild(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *G, float *rn,
                         float *w_out, int32_t *status, const Ws &ws, hipStream_t st, bool lower_only) {
    if (nb <= 0) return ADMMNET_OK;
    if (!arrow_rebuild_supported(D)) {
        set_error("arrow_rebuild: D=%d unsupported", D);
        return ADMMNET_E_ARG;
    }
    const BrGeom g(D);
    if (D > 128) {   // eigenvectors to the global image, then the dense path's rebuild kernel
        {
            ProfScope _prof(KC_REBUILD, st);
            const size_t lds = ar_lds_bytes(g) - sizeof(float) * g.vt_floats();
            ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(arrow_rebuild_kernel<true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(arrow_rebuild_kernel<true>, dim3((unsigned)nb), dim3(AR_THREADS), lds, st, D, lw, phi, h,
                               G, rn, ws.w, status, (unsigned long long *)nullptr, lower_only ? 1 : 0, ws.VT, ws.w0);
            ADMM_HIP(hipGetLastError());
        }
        return launch_rebuild(D, nb, lw, phi, h, G, rn, w_out, ws, st, lower_only, D);   // (image laid out for D itself)
    }
    ProfScope _prof(KC_REBUILD, st);
    const size_t lds = ar_lds_bytes(g);
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(arrow_rebuild_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const bool timing = getenv("ADMMNET_AR_TIMING") != nullptr;   // developer aid, never on by default
    unsigned long long *ptime = nullptr;
    if (timing) {
        ADMM_HIP(hipMalloc(&ptime, 16 * sizeof(unsigned long long)));
        ADMM_HIP(hipMemsetAsync(ptime, 0, 16 * sizeof(unsigned long long), st));
    }
    hipLaunchKernelGGL(arrow_rebuild_kernel<false>, dim3((unsigned)nb), dim3(AR_THREADS), lds, st, D, lw, phi, h, G, rn,
                       w_out, status, ptime, lower_only ? 1 : 0, (float *)nullptr, (float *)nullptr);
    ADMM_HIP(hipGetLastError());
    if (timing) {
        unsigned long long hb[16];
        ADMM_HIP(hipMemcpyAsync(hb, ptime, sizeof(hb), hipMemcpyDeviceToHost, st));
        ADMM_HIP(hipStreamSynchronize(st));
        ADMM_HIP(hipFree(ptime));
        static const char *nm[6] = {"sort+deflate", "roots", "zhat+rank+norm", "vectors", "G tiles", "arrow+norm"};
        fprintf(stderr, "[arrow_rebuild timing] D=%d nb=%lld  mean cycles per workgroup:\n", D, (long long)nb);
        for (int i = 0; i < 6; ++i) fprintf(stderr, "   %-14s %10.0f\n", nm[i], (double)hb[i] / (double)nb);
    }
    return ADMMNET_OK;
}

}  // namespace admmnet
