// zstep.hip -- K5: batch-coupled adaptive dual step of the Z layer.
//   r_b / (mean_b r + eps) -> MLP 3 -> 32 -> 1 -> sigmoid -> 0.5 + 1.5 s -> * softplus(rho)
//   /root/reference/admm_net.py:443-474 (ZLayer._compute_adaptive_step).
// The batch mean is the only cross-signal (and cross-GPU) coupling of the whole
// forward: the local sum is produced in fp64 by one deterministic workgroup and
// the caller may all-reduce it before the step kernel runs.
#include "common.h"

namespace admmnet {

__global__ __launch_bounds__(1024) void rn_sum_kernel(int64_t B, const float *__restrict__ rn,
                                                      double *__restrict__ sum) {
    __shared__ double sh[1024];
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < B; i += 1024) a += (double)rn[i];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sum[0] = sh[0];
        sum[1] = (double)B;   // (the pair the sharded protocol all-reduces: local sum, local count)
    }
}

__global__ void mean_kernel(const double *__restrict__ sum, int64_t B, float *__restrict__ mean) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mean[0] = (float)(sum[0] / (double)B);
}

__global__ __launch_bounds__(256) void zstep_kernel(int D, int64_t B, const float *__restrict__ lw,
                                                    const float *__restrict__ rn,
                                                    const float *__restrict__ mean,
                                                    float *__restrict__ alpha) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const LayerLayout L{D};
    const float *rs = lw + L.off_rs();   // W1[32][3] b1[32] w2[32] b2[1]
    const float rho = lw[S_RHO_Z];
    const float f0 = lw[S_KNORM], f1 = rho, f2 = rn[i] / (mean[0] + kEpsRef);
    float acc = rs[160];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        float hj = rs[96 + j];
        hj = fmaf(rs[3 * j + 0], f0, hj);
        hj = fmaf(rs[3 * j + 1], f1, hj);
        hj = fmaf(rs[3 * j + 2], f2, hj);
        acc = fmaf(rs[128 + j], fmaxf(hj, 0.f), acc);
    }
    const float sf = 0.5f + 1.5f * sigmoid_f(acc);
    alpha[i] = rho * sf;
}

int launch_rn_sum(int64_t B, const float *rn, double *sum, hipStream_t st) {
    hipLaunchKernelGGL(rn_sum_kernel, dim3(1), dim3(1024), 0, st, B, rn, sum);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

__global__ void mean_pair_kernel(const double *__restrict__ sc, float *__restrict__ mean) {
    if (threadIdx.x == 0 && blockIdx.x == 0) mean[0] = (float)(sc[0] / sc[1]);
}

int launch_mean_from_pair(const double *sum_count, float *mean, hipStream_t st) {
    hipLaunchKernelGGL(mean_pair_kernel, dim3(1), dim3(64), 0, st, sum_count, mean);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_mean_from_sum(const double *sum, int64_t B, float *mean, hipStream_t st) {
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, st, sum, B, mean);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

int launch_zstep(const float *lw, int D, int64_t B, const float *rn, const float *mean, float *alpha,
                 hipStream_t st) {
    ProfScope _prof(KC_ZSTEP, st);
    if (B <= 0) return ADMMNET_OK;
    hipLaunchKernelGGL(zstep_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, D, B, lw, rn, mean,
                       alpha);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
