// tql.hip -- K2: implicit-shift QL on the tridiagonal matrices, ONE WAVE PER MATRIX.
//
// The QL recurrence is a serial scalar chain, so throughput comes from running thousands of
// independent chains, four per SIMD.  One wave owns one matrix: d, e and the running first row
// of W live in LDS (1.5 KB per wave), all control flow is wave-uniform (scalar branches, no
// exec-mask divergence bookkeeping), the serial chain runs redundantly in every lane, and the
// lanes are used where the algorithm IS parallel: the deflation test of all off-diagonals is one
// vector compare + ballot per 64 entries, and "end of the unreduced block" is a scalar
// find-first-bit.  (A first version ran one matrix per LANE: 10x slower per rotation, because
// divergent loop control and per-lane LDS round trips sat on the serial chain.)
// Every plane rotation is appended to the matrix' rotation log (64-byte groups, eig_core.h);
// rotapply.hip replays it on the rows of Q.  Second half of torch.linalg.eigh
// (/root/reference/admm_net.py:303): LAPACK csteqr semantics with the EISPACK deflation test;
// the sweep direction is chosen per matrix (choose_flip) with one retry the other way.
// The algorithm is the one of tql_lane_pf (eig_core.h), which the host model runs.
#include <stdlib.h>

#include "common.h"

namespace admmnet {

__device__ __forceinline__ float ufloat(float x) {   // value is wave-uniform: move it to an SGPR
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}

constexpr int TQ_MAXBLK = 5;   // n <= 320

__global__ __launch_bounds__(64) void tql_wave_kernel(int n, const float *__restrict__ dT,
                                                      const float *__restrict__ eT,
                                                      float *__restrict__ wout, float *__restrict__ w0out,
                                                      LogRec *__restrict__ log, int *__restrict__ logn,
                                                      int64_t cap, int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NP = (n + 63) & ~63;
    float *ds = reinterpret_cast<float *>(smem);   // [NP + 1]
    float *es = ds + NP + 1;
    float *zs = es + NP + 1;
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    const float *dg = dT + b * n, *eg = eT + b * n;
    LogRec *lg = log + b * cap;
    const int nblk = NP >> 6;
    const LogRec ident = [] { LogRec r; r.r.c = 1.0f; r.r.s = 0.0f; return r; }();

    // direction: reversed when the top half of T carries more weight (eig_core.h choose_flip)
    float top = 0.f, bot = 0.f;
    const int hlf = n / 2;
    for (int i = lane; i < hlf; i += 64) {
        top += fabsf(dg[i]) + fabsf(eg[i]);
        bot += fabsf(dg[n - 1 - i]) + (n - 2 - i >= 0 ? fabsf(eg[n - 2 - i]) : 0.f);
    }
    const bool flip0 = wave_sum(top) > wave_sum(bot);

    int st = 1, pos = 0;
    bool flip = flip0;
    for (int attempt = 0; attempt < 2 && st != 0; ++attempt) {
        flip = flip0 ^ (attempt == 1);
        for (int i = lane; i <= NP; i += 64) {
            const int src = flip ? n - 1 - i : i;
            ds[i] = (i < n) ? dg[src] : 0.f;
            es[i] = (i < n - 1) ? (flip ? eg[n - 2 - i] : eg[i]) : 0.f;
            zs[i] = (i == (flip ? n - 1 : 0)) ? 1.f : 0.f;
        }
        pos = 0;
        st = 0;
        uint64_t mask[TQ_MAXBLK];
        for (int l = 0; l < n && st == 0; ++l) {
            int iter = 0;
            for (;;) {
                // deflation test of every off-diagonal at once; bit n-1 is the sentinel
#pragma unroll
                for (int k = 0; k < TQ_MAXBLK; ++k) {
                    mask[k] = 0;
                    if (k < nblk) {
                        const int j = 64 * k + lane;
                        const float dj = ds[j], dj1 = ds[j + 1], ej = es[j];
                        const bool neg = (j < n - 1) ? (fabsf(ej) <= kEps32 * (fabsf(dj) + fabsf(dj1))) : (j == n - 1);
                        mask[k] = __ballot(neg);
                    }
                }
                int m = n - 1;
#pragma unroll
                for (int k = TQ_MAXBLK - 1; k >= 0; --k) {
                    if (k < nblk) {
                        const int lo = 64 * k;
                        uint64_t x = mask[k];
                        if (l > lo) x = (l - lo >= 64) ? 0 : (x & (~(uint64_t)0 << (l - lo)));
                        if (x) m = lo + (int)__builtin_ctzll(x);
                    }
                }
                if (m == l) break;
                if (iter++ >= 60) {
                    st = 1;
                    break;
                }
                // ---- one QL sweep over planes m-1 .. l
                const int g_hi = (m - 1) >> 3, g_lo = l >> 3;
                const int need = 8 + 8 * (g_hi - g_lo + 1);
                if (pos + need > cap) {
                    st = 2;
                    break;
                }
                int cur = pos + 8;
                const int endp = pos + need;
                if (lane == 0) {
                    LogRec h;
                    h.h.g_hi = g_hi;
                    h.h.g_lo = g_lo;
                    lg[pos] = h;
                }
                pos = endp;
                {   // identity records for the planes of the top group above the window
                    const int npad = 8 * g_hi + 7 - (m - 1);
                    if (lane < npad) lg[cur + lane] = ident;
                    cur += npad;
                }
                const float dl = ufloat(ds[l]), el = ufloat(es[l]);
                float g = (ufloat(ds[l + 1]) - dl) / (2.0f * el);
                float r = sqrtf(g * g + 1.0f);
                g = ufloat(ds[m]) - dl + el / (g + sign_of(r, g));
                float s = 1.0f, c = 1.0f, p = 0.0f;
                int i = m - 1;
                float e_i = es[i], d_i = ds[i], d_ip1 = ds[m];
                float zc = zs[m];
                bool brk = false;
                for (; i >= l; --i) {
                    const int ip = (i > l) ? i - 1 : i;
                    const float e_nx = es[ip], d_nx = ds[ip];
                    const float zi = zs[i];
                    const float f = s * e_i, bq = c * e_i;
                    const float rr = f * f + g * g;
                    if (__builtin_amdgcn_readfirstlane(__float_as_int(rr)) == 0) {   // underflow exit
                        if (lane == 0) {
                            es[i + 1] = 0.0f;
                            ds[i + 1] = d_ip1 - p;
                            es[m] = 0.0f;
                        }
                        brk = true;
                        break;
                    }
                    const float rinv = rsqrt_fast(rr);
                    const float e_new = rr * rinv;
                    const float s0 = f * rinv, c0 = g * rinv;
                    const float h = 0.5f * fmaf(-s0, s0, fmaf(-c0, c0, 1.0f));
                    s = fmaf(h, s0, s0);
                    c = fmaf(h, c0, c0);
                    g = d_ip1 - p;
                    const float t = (d_i - g) * s + 2.0f * c * bq;
                    p = s * t;
                    const float dnew = g + p;
                    g = c * t - bq;
                    const float znew = s * zi + c * zc;
                    zc = c * zi - s * zc;
                    if (lane == 0) {
                        es[i + 1] = e_new;
                        ds[i + 1] = dnew;
                        zs[i + 1] = znew;
                        LogRec rec;
                        rec.r.c = c;
                        rec.r.s = s;
                        lg[cur] = rec;
                    }
                    ++cur;
                    d_ip1 = d_i;
                    e_i = e_nx;
                    d_i = d_nx;
                }
                if (lane == 0) zs[i + 1] = zc;
                {   // identity for the slots that are left (below the window / after an early exit)
                    const int left = endp - cur;
                    if (lane < left) lg[cur + lane] = ident;
                }
                if (!brk && lane == 0) {
                    ds[l] = d_ip1 - p;
                    es[l] = g;
                    es[m] = 0.0f;
                }
            }
        }
    }
    if (lane == 0) {
        logn[b * 2 + 0] = (st == 0) ? pos : 0;
        logn[b * 2 + 1] = st | (flip ? 256 : 0);
        if (st != 0 && status) atomicAdd(status, 1);
    }
    for (int i = lane; i < n; i += 64) {
        wout[b * n + i] = ds[i];
        w0out[b * n + i] = zs[i];
    }
}

int launch_tql(int n, int64_t nb, const Ws &ws, int32_t *status, hipStream_t st) {
    ProfScope _prof(KC_TQL, st);
    if (nb <= 0) return ADMMNET_OK;
    if (n > 64 * TQ_MAXBLK) {
        set_error("tql: n=%d unsupported (max %d)", n, 64 * TQ_MAXBLK);
        return ADMMNET_E_ARG;
    }
    const int NP = (n + 63) & ~63;
    const size_t lds = sizeof(float) * 3 * (NP + 1);
    hipLaunchKernelGGL(tql_wave_kernel, dim3((unsigned)nb), dim3(64), lds, st, n, ws.dT, ws.eT, ws.w, ws.w0,
                       ws.log, ws.logn, ws.cap, status);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
