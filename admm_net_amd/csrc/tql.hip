// tql.hip -- K2: implicit-shift QL on the tridiagonal matrices, ONE MATRIX PER
// LANE.  The QL recurrence is a serial scalar chain per matrix, so instead of
// one workgroup idling 511 threads behind it, 64 (or 32) matrices advance in
// lock-step in the lanes of one wave.  d, e and the running first row of W live
// in LDS as [i][lane] (conflict-free).  Every plane rotation is appended to the
// matrix' rotation log in global memory; rotapply.hip replays that log on the
// rows of Q.  Second half of torch.linalg.eigh (/root/reference/admm_net.py:303),
// LAPACK csteqr semantics (QL branch) with the EISPACK deflation test.
#include <stdlib.h>

#include "common.h"

namespace admmnet {

__global__ __launch_bounds__(64) void tql_kernel(int n, int64_t nb, int lanes,
                                                 const float *__restrict__ dT,
                                                 const float *__restrict__ eT, float *__restrict__ wout,
                                                 float *__restrict__ w0out, LogRec *__restrict__ log,
                                                 int *__restrict__ logn, int64_t cap,
                                                 int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *ds = reinterpret_cast<float *>(smem);   // [n][lanes]
    float *es = ds + (size_t)n * lanes;
    float *zs = es + (size_t)n * lanes;
    const int lane = threadIdx.x;
    const int64_t b = (int64_t)blockIdx.x * lanes + lane;
    const bool active = lane < lanes && b < nb;
    // no cross-lane LDS traffic: each lane touches only its own column
    if (!active) return;
    const float *dcol = dT + ((b >> 6) * n) * 64 + (b & 63);
    const float *ecol = eT + ((b >> 6) * n) * 64 + (b & 63);
    LogRec *lg = log + b * cap;
    auto Dacc = [&](int i) -> float & { return ds[i * lanes + lane]; };
    auto Eacc = [&](int i) -> float & { return es[i * lanes + lane]; };
    auto Zacc = [&](int i) -> float & { return zs[i * lanes + lane]; };
    LogWriter lw{lg, (int)cap, 0, 0, 0};
    auto Dg = [&](int i) -> float { return dcol[i * 64]; };
    auto Eg = [&](int i) -> float { return ecol[i * 64]; };
    const bool flip0 = choose_flip(n, Dg, Eg);
    int st = 1;
    bool flip = flip0;
    for (int attempt = 0; attempt < 2 && st != 0; ++attempt) {
        flip = flip0 ^ (attempt == 1);   // second try: the other direction
        for (int i = 0; i < n; ++i) {
            const int src = flip ? n - 1 - i : i;
            ds[i * lanes + lane] = dcol[src * 64];
            float ev = 0.f;
            if (i < n - 1) ev = flip ? ecol[(n - 2 - i) * 64] : ecol[i * 64];
            es[i * lanes + lane] = ev;
            zs[i * lanes + lane] = (i == (flip ? n - 1 : 0)) ? 1.f : 0.f;
        }
        lw.pos = 0;
        int nsweeps = 0;
        st = tql_lane_pf(n, Dacc, Eacc, Zacc, lw, 60, nsweeps);
    }
    logn[b * 2 + 0] = (st == 0) ? lw.pos : 0;
    logn[b * 2 + 1] = st | (flip ? 256 : 0);
    if (st != 0 && status) atomicAdd(status, 1);
    for (int i = 0; i < n; ++i) {
        wout[b * n + i] = ds[i * lanes + lane];
        w0out[b * n + i] = zs[i * lanes + lane];
    }
}

int launch_tql(int n, int64_t nb, const Ws &ws, int32_t *status, hipStream_t st) {
    ProfScope _prof(KC_TQL, st);
    if (nb <= 0) return ADMMNET_OK;
    // Matrices per wave: the QL recurrence is latency bound, so few lanes per wave (more waves,
    // less lock-step divergence) beats dense packing; ADMMNET_TQL_LANES overrides for experiments.
    static int lanes_cfg = -1;
    if (lanes_cfg < 0) {
        const char *e = getenv("ADMMNET_TQL_LANES");
        lanes_cfg = e ? atoi(e) : 16;
        if (lanes_cfg < 1 || lanes_cfg > 64) lanes_cfg = 16;
    }
    int lanes = lanes_cfg;
    while ((size_t)3 * n * lanes * sizeof(float) > 150 * 1024 && lanes > 1) lanes >>= 1;
    const size_t lds = (size_t)3 * n * lanes * sizeof(float);
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tql_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (nb + lanes - 1) / lanes;
    hipLaunchKernelGGL(tql_kernel, dim3((unsigned)blocks), dim3(lanes), lds, st, n, nb, lanes, ws.dT, ws.eT,
                       ws.w, ws.w0, ws.log, ws.logn, ws.cap, status);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
