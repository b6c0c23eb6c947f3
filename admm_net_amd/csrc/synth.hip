// synth.hip -- batched OFDM-radar scene synthesis + classical-solver labels on the device.
//
// One workgroup per sample restates /root/reference/generate_data.py:133-221 (_generate_single_sample /
// _generate_communication_symbols): L targets tau ~ U(0.1, 0.9), f ~ U(-0.4, 0.4), C ~ N(0, 0.7^2) + j N(0, 0.7^2);
// Psi = kr(S, conj D) C; QPSK symbols, 7 dB demodulation noise, b = mod(demod(.)), e = sig - b; y = diag(b + e) Psi + w
// at snr ~ U(lo, hi) dB; sigma = ||e / b|| + 1; casts to complex64 / float32 (:196-201).  The phi label of
// DatasetGeneratorCreatePhi (:410-463) is admm_for_us(y, b, ...), which as written collapses to the recursion
//   phi_k = W (y / b + rho phi_{k-1}),  W = (diag(1 / |b|^2) + rho 1 1^T)^-1,  stopped at min_iter = 5
// (admm.py:77-79 with the 'rho * np.ones(len)' broadcast) -- O(D) per iteration through
// Sherman-Morrison, evaluated here in float64 like the reference's complex128.
// The reference seeds nothing; the generator here is counter based (splitmix64 of (seed, sample, stream, index)), so a
// batch is reproducible and independent of the launch geometry.  Arithmetic of the scene in float64, as numpy's.
#include "common.h"

namespace admmnet {

constexpr int SY_THREADS = 256;
constexpr int SY_MAXL = 8;

__device__ __forceinline__ unsigned long long sy_mix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ unsigned long long sy_bits(unsigned long long seed, long long sample, int stream, int idx) {
    return sy_mix(sy_mix(sy_mix(seed ^ 0xA5A5A5A55A5A5A5Aull) + (unsigned long long)sample) ^
                  (((unsigned long long)stream << 40) | (unsigned int)idx));
}
__device__ __forceinline__ double sy_uniform(unsigned long long bits) {   // (0, 1)
    return ((double)(bits >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
// two independent standard normals (Box-Muller) from one 64-bit draw + its successor
__device__ __forceinline__ double2 sy_normal2(unsigned long long seed, long long sample, int stream, int idx) {
    const double u1 = sy_uniform(sy_bits(seed, sample, stream, 2 * idx)), u2 = sy_uniform(sy_bits(seed, sample, stream, 2 * idx + 1));
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincos(2.0 * 3.14159265358979323846 * u2, &s, &c);
    return make_double2(r * c, r * s);
}

enum { SY_TAU = 1, SY_F, SY_C, SY_DATA, SY_DEMOD, SY_SNR, SY_NOISE };

__global__ __launch_bounds__(SY_THREADS) void synth_kernel(int Nb, int Nd, int L, unsigned long long seed, double snr_lo,
                                                           double snr_hi, double snr_e, double rho, int label_iters,
                                                           float2 *__restrict__ y_out, float2 *__restrict__ b_out,
                                                           float *__restrict__ sigma_out, float *__restrict__ tau_out,
                                                           float *__restrict__ f_out, float2 *__restrict__ C_out,
                                                           float2 *__restrict__ phi_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = Nb * Nd;
    double2 *yv = reinterpret_cast<double2 *>(smem);     // [D] y (float64)
    double2 *bv = yv + D;                                // [D] b
    double2 *ph = bv + D;                                // [D] label recursion
    double *red = reinterpret_cast<double *>(ph + D);    // [8]
    __shared__ double tgt[SY_MAXL][4];                   // tau, f, Re C, Im C
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long s = blockIdx.x;
    if (tid < L) {
        tgt[tid][0] = 0.1 + 0.8 * sy_uniform(sy_bits(seed, s, SY_TAU, tid));     // generate_data.py:30,138
        tgt[tid][1] = -0.4 + 0.8 * sy_uniform(sy_bits(seed, s, SY_F, tid));      // :31,139
        const double2 g = sy_normal2(seed, s, SY_C, tid);                         // :142-144
        tgt[tid][2] = 0.7 * g.x;
        tgt[tid][3] = 0.7 * g.y;
        tau_out[s * L + tid] = (float)tgt[tid][0];
        f_out[s * L + tid] = (float)tgt[tid][1];
        C_out[s * L + tid] = make_float2((float)tgt[tid][2], (float)tgt[tid][3]);
    }
    __syncthreads();
    const double p_noise = 1.0 / pow(10.0, snr_e / 10.0);   // |sig| = 1: awgn(sig, snr_e), mathUtils.py:93-111
    double acc_y2 = 0.0, acc_eb = 0.0;
    for (int i = tid; i < D; i += SY_THREADS) {
        const int ib = i / Nd, id = i - ib * Nd;           // kr(S, conj D): index = i_b * Nd + i_d
        double pr = 0.0, pi = 0.0;
        for (int l = 0; l < L; ++l) {
            double sn, cs;
            sincos(2.0 * 3.14159265358979323846 * ((double)ib * tgt[l][1] - (double)id * tgt[l][0]), &sn, &cs);
            pr += tgt[l][2] * cs - tgt[l][3] * sn;
            pi += tgt[l][2] * sn + tgt[l][3] * cs;
        }
        // QPSK: sig = exp(j (2 pi data / 4 + pi / 4)); demodulate the noisy symbol; b = mod(demod)   (:205-221)
        const int data = (int)(sy_bits(seed, s, SY_DATA, i) >> 62);
        double sgs, sgc;
        sincos(2.0 * 3.14159265358979323846 * data / 4.0 + 3.14159265358979323846 / 4.0, &sgs, &sgc);
        const double2 nz = sy_normal2(seed, s, SY_DEMOD, i);
        const double nr = sgc + sqrt(p_noise / 2.0) * nz.x, ni = sgs + sqrt(p_noise / 2.0) * nz.y;
        double ang = atan2(ni, nr) - 3.14159265358979323846 / 4.0;
        ang = fmod(ang + 3.14159265358979323846 / 4.0, 2.0 * 3.14159265358979323846);
        if (ang < 0.0) ang += 2.0 * 3.14159265358979323846;                       // np.mod: result in [0, 2 pi)
        const int dd = ((int)floor(ang * 4.0 / (2.0 * 3.14159265358979323846))) & 3;
        double bs, bc;
        sincos(2.0 * 3.14159265358979323846 * dd / 4.0 + 3.14159265358979323846 / 4.0, &bs, &bc);
        const double er = sgc - bc, ei = sgs - bs;
        // real_y = (b + e) psi = sig psi
        const double yr = sgc * pr - sgs * pi, yi = sgc * pi + sgs * pr;
        yv[i] = make_double2(yr, yi);
        bv[i] = make_double2(bc, bs);
        acc_y2 += yr * yr + yi * yi;
        acc_eb += er * er + ei * ei;                       // |e / b|^2 = |e|^2 (|b| = 1)
    }
    // block sums
    for (int o = 32; o > 0; o >>= 1) {
        acc_y2 += __shfl_xor(acc_y2, o, 64);
        acc_eb += __shfl_xor(acc_eb, o, 64);
    }
    if (lane == 0) {
        red[wave] = acc_y2;
        red[4 + wave] = acc_eb;
    }
    __syncthreads();
    const double y2 = (red[0] + red[1]) + (red[2] + red[3]), eb = (red[4] + red[5]) + (red[6] + red[7]);
    const double snr_w = snr_lo + (snr_hi - snr_lo) * sy_uniform(sy_bits(seed, s, SY_SNR, 0));   // :164
    const double w_std = sqrt(y2 / (pow(10.0, snr_w / 10.0) * (double)D));                        // :166-169
    for (int i = tid; i < D; i += SY_THREADS) {
        const double2 nz = sy_normal2(seed, s, SY_NOISE, i);
        double2 v = yv[i];
        v.x += w_std * 0.70710678118654752440 * nz.x;
        v.y += w_std * 0.70710678118654752440 * nz.y;
        yv[i] = v;
        y_out[s * D + i] = make_float2((float)v.x, (float)v.y);
        b_out[s * D + i] = make_float2((float)bv[i].x, (float)bv[i].y);
        ph[i] = make_double2(0.0, 0.0);
    }
    if (tid == 0) sigma_out[s] = (float)(sqrt(eb) + 1.0);                                          // :171
    if (phi_out == nullptr) return;   // (uniform)
    // ---- label: phi_k = W (y / b + rho phi_{k-1}),  W r = d r - d (rho sum(d r)) / (1 + rho sum d),  d = |b|^2
    __syncthreads();
    for (int it = 0; it < label_iters; ++it) {
        double sr = 0.0, si = 0.0, sd = 0.0;
        for (int i = tid; i < D; i += SY_THREADS) {
            const double2 bb = bv[i], yy = yv[i], pp = ph[i];
            const double d = bb.x * bb.x + bb.y * bb.y;
            // y / b = y conj(b) / |b|^2
            const double qr = (yy.x * bb.x + yy.y * bb.y) / d + rho * pp.x, qi = (yy.y * bb.x - yy.x * bb.y) / d + rho * pp.y;
            sr += d * qr;
            si += d * qi;
            sd += d;
        }
        for (int o = 32; o > 0; o >>= 1) {
            sr += __shfl_xor(sr, o, 64);
            si += __shfl_xor(si, o, 64);
            sd += __shfl_xor(sd, o, 64);
        }
        __syncthreads();              // the previous iteration's readers of red are done
        if (lane == 0) {
            red[wave] = sr;
            red[4 + wave] = si;
        }
        __shared__ double sdw[4];
        if (lane == 0) sdw[wave] = sd;
        __syncthreads();
        const double Sr = (red[0] + red[1]) + (red[2] + red[3]), Si = (red[4] + red[5]) + (red[6] + red[7]);
        const double Sd = (sdw[0] + sdw[1]) + (sdw[2] + sdw[3]);
        const double kr = rho * Sr / (1.0 + rho * Sd), ki = rho * Si / (1.0 + rho * Sd);
        for (int i = tid; i < D; i += SY_THREADS) {
            const double2 bb = bv[i], yy = yv[i], pp = ph[i];
            const double d = bb.x * bb.x + bb.y * bb.y;
            const double qr = (yy.x * bb.x + yy.y * bb.y) / d + rho * pp.x, qi = (yy.y * bb.x - yy.x * bb.y) / d + rho * pp.y;
            ph[i] = make_double2(d * qr - d * kr, d * qi - d * ki);
        }
        __syncthreads();
    }
    for (int i = tid; i < D; i += SY_THREADS) phi_out[s * D + i] = make_float2((float)ph[i].x, (float)ph[i].y);
}

int launch_synth(int64_t B, int Nb, int Nd, int L, unsigned long long seed, double snr_lo, double snr_hi, double snr_e,
                 double rho, int label_iters, float2 *y, float2 *b, float *sigma, float *tau, float *f, float2 *C,
                 float2 *phi_label, hipStream_t st) {
    if (B <= 0) return ADMMNET_OK;
    const int D = Nb * Nd;
    const size_t lds = sizeof(double2) * 3 * D + sizeof(double) * 8;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)B), dim3(SY_THREADS), lds, st, Nb, Nd, L, seed, snr_lo, snr_hi, snr_e,
                       rho, label_iters, y, b, sigma, tau, f, C, phi_label);
    ADMM_HIP(hipGetLastError());
    return ADMMNET_OK;
}

}  // namespace admmnet
