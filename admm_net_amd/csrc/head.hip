// head.hip -- learned peak-search head of ADMMNet (eval mode).
//   /root/reference/admm_net.py:570-630 (PeakSearchLayer.forward):
//   MLP(2D->128->128) -> 1-query 4-head attention over D position tokens ->
//   residual -> MLP 128->64->32->16 -> L x (tau, f) regressors + shared confidence.
// The position tokens are batch independent (the reference repeats them B
// times, :594-595); their K / V projections are computed once per forward by
// headkv_kernel.  <1 % of the forward's flops: plain VALU code, weights stay
// L2 resident, one 128-thread workgroup per signal.
#include "common.h"

namespace admmnet {

constexpr int HH = 128;   // hidden_dim
constexpr int NH = 4;     // heads
constexpr int HD_ = 32;   // head dim

__global__ __launch_bounds__(HH) void headkv_kernel(int D, int L, const float *__restrict__ hw,
                                                    float *__restrict__ kv) {
    __shared__ float pos[HH];
    const HeadLayout H{D, L};
    const int t = blockIdx.x, o = threadIdx.x;
    const float p0 = hw[H.off_pos() + 2 * t], p1 = hw[H.off_pos() + 2 * t + 1];
    pos[o] = fmaf(hw[H.off_ppw() + 2 * o], p0, fmaf(hw[H.off_ppw() + 2 * o + 1], p1, hw[H.off_ppb() + o]));
    __syncthreads();
    const float *inw = hw + H.off_inw();   // [128][384] transposed
    float ak = hw[H.off_inb() + HH + o], av = hw[H.off_inb() + 2 * HH + o];
    for (int j = 0; j < HH; ++j) {
        ak = fmaf(inw[j * 384 + HH + o], pos[j], ak);
        av = fmaf(inw[j * 384 + 2 * HH + o], pos[j], av);
    }
    kv[(int64_t)t * HH + o] = ak;
    kv[(int64_t)D * HH + (int64_t)t * HH + o] = av;
}

__device__ __forceinline__ float dense(const float *wT, int ostride, int o, const float *bias,
                                       const float *in, int nin) {
    float a = bias[o];
    for (int i = 0; i < nin; ++i) a = fmaf(wT[i * ostride + o], in[i], a);
    return a;
}

__global__ __launch_bounds__(HH) void head_kernel(int D, int L, const float *__restrict__ hw,
                                                  const float *__restrict__ kv, int64_t B,
                                                  const float2 *__restrict__ phi, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *feat = reinterpret_cast<float *>(smem);   // [2D]
    float *x1 = feat + 2 * D;                        // [128]
    float *x = x1 + HH;                              // [128]
    float *qv = x + HH;                              // [128]
    float *ctx = qv + HH;                            // [128]
    float *sc = ctx + HH;                            // [4][D]
    float *red = sc + NH * D;                        // [8]
    float *t0 = red + 8;                             // [64]
    float *t1 = t0 + 64;                             // [32]
    float *xp = t1 + 32;                             // [16]
    float *hidr = xp + 16;                           // [3][32]
    const HeadLayout H{D, L};
    const int o = threadIdx.x;
    const int64_t b = blockIdx.x;
    for (int i = o; i < D; i += HH) {
        const float2 p = phi[b * D + i];
        feat[i] = p.x;
        feat[D + i] = p.y;
    }
    __syncthreads();
    x1[o] = fmaxf(dense(hw + H.off_fe0w(), HH, o, hw + H.off_fe0b(), feat, 2 * D), 0.f);
    __syncthreads();
    x[o] = fmaxf(dense(hw + H.off_fe2w(), HH, o, hw + H.off_fe2b(), x1, HH), 0.f);
    __syncthreads();
    qv[o] = dense(hw + H.off_inw(), 384, o, hw + H.off_inb(), x, HH);
    __syncthreads();
    // scores: (head, token) pairs, scale 1/sqrt(32)
    const float scale = 0.17677669529663687f;
    for (int p = o; p < NH * D; p += HH) {
        const int hh = p / D, t = p - hh * D;
        const float *kr = kv + (int64_t)t * HH + hh * HD_;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < HD_; ++d) a = fmaf(qv[hh * HD_ + d], kr[d], a);
        sc[p] = a * scale;
    }
    __syncthreads();
    // softmax per head: wave w (2 waves) handles heads w, w+2
    {
        const int lane = o & 63, wave = o >> 6;
        for (int hh = wave; hh < NH; hh += HH / 64) {
            float m = -INFINITY;
            for (int t = lane; t < D; t += 64) m = fmaxf(m, sc[hh * D + t]);
            m = wave_max(m);
            float ssum = 0.f;
            for (int t = lane; t < D; t += 64) {
                const float e = expf(sc[hh * D + t] - m);
                sc[hh * D + t] = e;
                ssum += e;
            }
            ssum = wave_sum(ssum);
            if (lane == 0) red[hh] = ssum;
        }
    }
    __syncthreads();
    {
        const int hh = o / HD_;
        const float *vv = kv + (int64_t)D * HH;
        float a = 0.f;
        for (int t = 0; t < D; ++t) a = fmaf(sc[hh * D + t], vv[(int64_t)t * HH + o], a);
        ctx[o] = a / red[hh];
    }
    __syncthreads();
    const float xf = x[o] + dense(hw + H.off_outw(), HH, o, hw + H.off_outb(), ctx, HH);
    __syncthreads();
    x1[o] = xf;   // reuse x1 as x_fixed
    __syncthreads();
    if (o < 64) t0[o] = fmaxf(dense(hw + H.off_pe0w(), 64, o, hw + H.off_pe0b(), x1, HH), 0.f);
    __syncthreads();
    if (o < 32) t1[o] = fmaxf(dense(hw + H.off_pe2w(), 32, o, hw + H.off_pe2b(), t0, 64), 0.f);
    __syncthreads();
    if (o < 16) xp[o] = fmaxf(dense(hw + H.off_pe4w(), 16, o, hw + H.off_pe4b(), t1, 32), 0.f);
    __syncthreads();
    for (int t = 0; t < L; ++t) {
        const float off = (float)((double)t / (double)L);
        const float *rg = hw + H.off_reg(t);
        // three 16 -> {32, 32, 16} hidden layers: threads 0..31 tau, 32..63 f, 64..79 confidence
        if (o < 80) {
            const float *w;
            const float *bb;
            int ostride, oo;
            if (o < 32) { w = rg; bb = rg + 512; ostride = 32; oo = o; }
            else if (o < 64) { w = rg + 577; bb = rg + 577 + 512; ostride = 32; oo = o - 32; }
            else { w = hw + H.off_conf(); bb = w + 256; ostride = 16; oo = o - 64; }
            float a = bb[oo];
#pragma unroll
            for (int i = 0; i < 16; ++i) a = fmaf(w[i * ostride + oo], xp[i] + off, a);
            hidr[o] = fmaxf(a, 0.f);
        }
        __syncthreads();
        if (o < 3) {
            const float *w2;
            const float *hsrc;
            int nh;
            if (o == 0) { w2 = rg + 544; hsrc = hidr; nh = 32; }
            else if (o == 1) { w2 = rg + 577 + 544; hsrc = hidr + 32; nh = 32; }
            else { w2 = hw + H.off_conf() + 272; hsrc = hidr + 64; nh = 16; }
            float a = w2[nh];   // bias follows the nh weights
            for (int j = 0; j < nh; ++j) a = fmaf(w2[j], hsrc[j], a);
            const float v = (o == 1) ? tanhf(a) : sigmoid_f(a);
            out[((int64_t)o * B + b) * L + t] = v;
        }
        __syncthreads();
    }
}

int launch_head(const admmnet_cfg *cfg, const float *hw, int64_t B, const float2 *phi, float *kv,
                float *out, hipStream_t st) {
    ProfScope _prof(KC_HEAD, st);
    if (B <= 0) return ADMMNET_OK;
    const int D = cfg->M * cfg->N, L = cfg->L;
    hipLaunchKernelGGL(headkv_kernel, dim3(D), dim3(HH), 0, st, D, L, hw, kv);
    ADMM_HIP(hipGetLastError());
    const size_t lds = sizeof(float) * (2 * D + 4 * HH + NH * D + 8 + 64 + 32 + 16 + 96);
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)B), dim3(HH), lds, st, D, L, hw, kv, B, phi, out);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
