// tql.hip -- K2: implicit-shift QL on the tridiagonal matrices, ONE MATRIX PER
// LANE.  The QL recurrence is a serial scalar chain per matrix, so instead of
// one workgroup idling 511 threads behind it, 64 (or 32) matrices advance in
// lock-step in the lanes of one wave.  d, e and the running first row of W live
// in LDS as [i][lane] (conflict-free).  Every plane rotation is appended to the
// matrix' rotation log in global memory; rotapply.hip replays that log on the
// rows of Q.  Second half of torch.linalg.eigh (/root/reference/admm_net.py:303),
// LAPACK csteqr semantics (QL branch) with the EISPACK deflation test.
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
    if (active) {
        const float *dcol = dT + ((b >> 6) * n) * 64 + (b & 63);
        const float *ecol = eT + ((b >> 6) * n) * 64 + (b & 63);
        for (int i = 0; i < n; ++i) {
            ds[i * lanes + lane] = dcol[i * 64];
            es[i * lanes + lane] = ecol[i * 64];
            zs[i * lanes + lane] = (i == 0) ? 1.f : 0.f;
        }
    }
    // no cross-lane LDS traffic: each lane touches only its own column
    if (!active) return;
    LogRec *lg = log + b * (cap + 16) + 8;   // 8 pad records on both sides
    int pos = 0;
    auto Dacc = [&](int i) -> float & { return ds[i * lanes + lane]; };
    auto Eacc = [&](int i) -> float & { return es[i * lanes + lane]; };
    auto Zacc = [&](int i) -> float & { return zs[i * lanes + lane]; };
    auto emit = [&](const LogRec &r) -> bool {
        if (pos >= cap) return false;
        lg[pos++] = r;
        return true;
    };
    auto patch = [&](int at, int i0, int cnt) -> int {
        if (at < 0) {
            if (pos >= cap) return -1;
            return pos++;
        }
        LogRec h;
        h.h.i0 = i0;
        h.h.cnt = cnt;
        lg[at] = h;
        return at;
    };
    int nsweeps = 0;
    const int st = tql_lane(n, Dacc, Eacc, Zacc, emit, patch, 60, nsweeps);
    logn[b * 2 + 0] = (st == 0) ? pos : 0;
    logn[b * 2 + 1] = st;
    if (st != 0 && status) atomicAdd(status, 1);
    for (int i = 0; i < n; ++i) {
        wout[b * n + i] = ds[i * lanes + lane];
        w0out[b * n + i] = zs[i * lanes + lane];
    }
}

int launch_tql(int n, int64_t nb, const Ws &ws, int32_t *status, hipStream_t st) {
    if (nb <= 0) return ADMMNET_OK;
    int lanes = 64;
    while ((size_t)3 * n * lanes * sizeof(float) > 150 * 1024 && lanes > 1) lanes >>= 1;
    const size_t lds = (size_t)3 * n * lanes * sizeof(float);
    ADMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tql_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (nb + lanes - 1) / lanes;
    hipLaunchKernelGGL(tql_kernel, dim3((unsigned)blocks), dim3(64), lds, st, n, nb, lanes, ws.dT, ws.eT,
                       ws.w, ws.w0, ws.log, ws.logn, ws.cap, status);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
