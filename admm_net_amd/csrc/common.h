// common.h -- shared host/device declarations of the gfx950 ADMM-Net library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/admmnet.h"
#include "eig_core.h"

namespace admmnet {

constexpr float kEpsRef = 1e-8f;   // epsilon of every reference layer (admm_net.py:74,114,211,360)
constexpr int kHid = 64;           // HLayer.correction_net hidden width (admm_net.py:128)
constexpr int kMaxD = 256;         // largest supported signal length D = M*N
constexpr int kLogCapMul = 3;      // rotation-log capacity = kLogCapMul * n * n records per matrix

void set_error(const char *fmt, ...);
#define ADMM_HIP(call)                                                              \
    do {                                                                            \
        hipError_t _e = (call);                                                     \
        if (_e != hipSuccess) {                                                     \
            admmnet::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                 \
            return ADMMNET_E_HIP;                                                   \
        }                                                                           \
    } while (0)

// ---- packed per-layer weights (floats) -------------------------------------
// scalars, resolved on the host by admmnet_pack_weights
enum LayerScalar {
    S_RHO_PHI = 0,    // softplus(phiLayers.k.rho)                      admm_net.py:97
    S_RHO_H_EPS = 1,  // softplus(hLayers.k.rho) + eps                  admm_net.py:148,151
    S_SIG_PW = 2,     // sigmoid(hLayers.k.projection_weight)          admm_net.py:188
    S_CORNER_G = 3,   // 1 / (softplus(gLayers.k.lambda_param)^2 + eps) admm_net.py:269-271
    S_INV_RHO_G = 4,  // 1 / (softplus(gLayers.k.rho) + eps)            admm_net.py:287-288
    S_THR = 5,        // sigmoid(gLayers.k.threshold)                  admm_net.py:321
    S_RHO_Z = 6,      // softplus(zLayers.k.rho)                        admm_net.py:406
    S_CORNER_Z = 7,   // 1 / (softplus(zLayers.k.lambda_param)^2 + eps) admm_net.py:424-426
    S_KNORM = 8,      // (float)(k / 10.0)                              admm_net.py:457
    S_A_COEF = 9,     // 2 * sqrtf((float)(M*N))                        admm_net.py:159
    S_COUNT = 16
};

struct LayerLayout {
    int D;
    __host__ __device__ int off_scal() const { return 0; }
    __host__ __device__ int off_w1t() const { return S_COUNT; }               // [D][64]  (transposed correction_net.0.weight)
    __host__ __device__ int off_b1() const { return off_w1t() + D * kHid; }   // [64]
    __host__ __device__ int off_w2t() const { return off_b1() + kHid; }       // [64][D]  (transposed correction_net.2.weight)
    __host__ __device__ int off_b2() const { return off_w2t() + kHid * D; }   // [D]
    __host__ __device__ int off_vn() const { return off_b2() + D; }           // value_net: w1[16] b1[16] w2[16] b2[1]
    __host__ __device__ int off_rs() const { return off_vn() + 49; }          // residual_scale_net: W1[32][3] b1[32] w2[32] b2[1]
    __host__ __device__ int size() const { return (off_rs() + 161 + 3) & ~3; }
};

// head (PeakSearchLayer) packed layout, hidden = 128, heads = 4
struct HeadLayout {
    int D, L;
    __host__ __device__ int off_pos() const { return 0; }                          // position_encoder [D][2]
    __host__ __device__ int off_fe0w() const { return off_pos() + 2 * D; }         // [2D][128] transposed
    __host__ __device__ int off_fe0b() const { return off_fe0w() + 2 * D * 128; }
    __host__ __device__ int off_fe2w() const { return off_fe0b() + 128; }          // [128][128] transposed
    __host__ __device__ int off_fe2b() const { return off_fe2w() + 128 * 128; }
    __host__ __device__ int off_ppw() const { return off_fe2b() + 128; }           // position_projection.weight [128][2]
    __host__ __device__ int off_ppb() const { return off_ppw() + 256; }
    __host__ __device__ int off_inw() const { return off_ppb() + 128; }            // in_proj_weight [384][128] -> stored transposed [128][384]
    __host__ __device__ int off_inb() const { return off_inw() + 384 * 128; }
    __host__ __device__ int off_outw() const { return off_inb() + 384; }           // out_proj.weight transposed [128][128]
    __host__ __device__ int off_outb() const { return off_outw() + 128 * 128; }
    __host__ __device__ int off_pe0w() const { return off_outb() + 128; }          // [128][64] transposed
    __host__ __device__ int off_pe0b() const { return off_pe0w() + 128 * 64; }
    __host__ __device__ int off_pe2w() const { return off_pe0b() + 64; }           // [64][32] transposed
    __host__ __device__ int off_pe2b() const { return off_pe2w() + 64 * 32; }
    __host__ __device__ int off_pe4w() const { return off_pe2b() + 32; }           // [32][16] transposed
    __host__ __device__ int off_pe4b() const { return off_pe4w() + 32 * 16; }
    // per target t: tau.0 w[16][32]T b[32], tau.2 w[32] b[1], f.0 w[16][32]T b[32], f.2 w[32] b[1]
    __host__ __device__ int off_reg(int t) const { return off_pe4b() + 16 + t * reg_size(); }
    __host__ __device__ int reg_size() const { return 2 * (16 * 32 + 32 + 32 + 1); }
    __host__ __device__ int off_conf() const { return off_reg(L); }                // c.0 w[16][16]T b[16], c.2 w[16] b[1]
    __host__ __device__ int size() const { return (off_conf() + 16 * 16 + 16 + 16 + 1 + 3) & ~3; }
};

// ---- workspace carve (all offsets in bytes, 256-byte aligned) ------------
struct Ws {
    float2 *G, *Z;            // [B][n][n] state, original index order, full storage
    float2 *phi[2];           // [B][D] double buffered (layer k uses k&1)
    float *h[2];              // [B][D]
    float *alpha;             // [B] adaptive step of the previous layer
    float *rn;                // [B] ||G - C||_F of the current layer
    double *sum;              // [2] scratch for the batch sum
    float *mean;              // [1] batch mean (single-rank path)
    float *headkv;            // [2][D][128] batch independent K / V projections of the head
    // chunk-sized eigensolver buffers
    float2 *Mbuf;             // [chunk][D*D + D + 1]: M row-major, arrow a[D], corner (re only)
    float *QV;                // [chunk][n][2D] planar transposed: QT (D columns) then VT (n columns)
    float *dT, *eT;           // [chunk][n] tridiagonal (diagonal, off-diagonal)
    float *w, *w0;            // [chunk][n] eigenvalues, first row of W
    float *Wdc;               // [chunk][3][n][n] divide & conquer: two WT ping-pong buffers + U (null: QL path)
    int2 *Wmap;               // [chunk][n] D = 256 path: where eigenvector j of T lives after the top-level merge (float offset
                              // into the matrix' Wdc block, valid rows lo | hi << 16) -- deflated columns are not copied there
    float *VT;                // [chunk][n][2D] eigenvectors for the rebuild (= QV on the QL path)
    float2 *Tfac;             // [chunk][17][16][16] T factors of the panel block reflectors (D = 256 path, else null)
    float2 *Tail;             // [chunk][36][4][64] trailing 128 x 128 tile set between the stages of tridiag_panel
    LogRec *log;              // [chunk][cap], 64-byte groups (eig_core.h)
    int *logn;                // [chunk][2]: records, status
    int64_t chunk;            // signals per chunk
    int64_t cap;              // log records per matrix (multiple of 8)
    int64_t total_bytes;
    int64_t set2_offset;      // byte offset of a second set of chunk buffers (two chunks in flight, api.hip), 0 = none
    // matrix-function fast path of the G-layer (spectral.hip, ADMMNET_SPECTRAL=1; null otherwise)
    float2 *spec_mat;         // [2][chunk][n][n]: A (then E in place), E^2
    float2 *spec_vec;         // [chunk][2][n] the two outlier eigenvectors
    double *spec_val;         // [chunk][8] lam0, lam1, c, residuals, trace
    int *spec_flag;           // [chunk] 1 = this matrix takes the eigen-pipeline after all
    const int *skip;          // per-matrix filter of the eigen-pipeline kernels: workgroups of matrices with skip[b] == 0 leave
                              // at once (null: every matrix runs)
};

int64_t eig_chunk_bytes(int D, int64_t chunk);
int carve_workspace(const admmnet_cfg *cfg, int64_t B, void *base, int64_t bytes, Ws *ws, bool state);
int64_t pick_chunk(const admmnet_cfg *cfg, int64_t B);

// ---- kernel launchers (each enqueues on `st`, returns ADMMNET_* code) ----------
// prep.hip
int launch_prep(const admmnet_cfg *cfg, const float *lw, int k, const float2 *y, const float2 *b,
                const float *sigma, int64_t b0, int64_t nb, const Ws &ws, bool phi_only, hipStream_t st,
                bool no_matrix = false, bool lean = false, bool no_image = false, bool small = false);
// (small: phi and h only -- the Z update of the previous layer is then the matrix-function kernel's, launch_spectral's `update`)
// (no_image, D > 128 lean route: only the Z update streams; launch_half_image then builds the image of the matrices with
//  ws.skip[s] != 0 from the updated Z)
int launch_half_image(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, const float2 *Z, const Ws &ws,
                      hipStream_t st);
// (image builders: the matrix lands in an eig_dim x eig_dim image, zero outside its own D x D block)
int launch_build_generic(int n, int64_t nb, const float2 *A, const Ws &ws, hipStream_t st);
int launch_build_block(int D, int64_t nb, float corner, float inv_rho, const float2 *phi, const float *h,
                       const float2 *Z, const Ws &ws, hipStream_t st);
// tridiag.hip
int launch_tridiag(int D, int64_t nb, const Ws &ws, hipStream_t st, const float2 *Zlow = nullptr,
                   const float2 *phi = nullptr, const float *h = nullptr, const float *lw = nullptr);
int launch_tridiag_reg(int D, int64_t nb, const Ws &ws, hipStream_t st, const float2 *Zlow = nullptr,
                       const float2 *phi = nullptr, const float *h = nullptr,
                       const float *lw = nullptr);   // tridiag_reg.hip, D <= 128
int launch_tridiag_big(int D, int64_t nb, const Ws &ws, hipStream_t st);   // tridiag_big.hip, 128 < D <= 256
bool tridiag_panel_supported(int D);                                      // tridiag_panel.hip, D == 256
// Dimension the eigen-pipeline works in: D itself, or 256 for 128 < D < 256 -- those geometries are embedded in the
// D = 256 pipeline as diag(A, 0) (api.hip, "padded route"); every chunk buffer is laid out for eig_dim(D).
int eig_dim(int D);
int launch_tridiag_panel(int D, int64_t nb, const Ws &ws, hipStream_t st);
int64_t tridiag_panel_tail_elems();                                       // float2 per matrix of Ws::Tail
bool use_wy_back(int D);                                                  // wy_apply.hip: V = Q W without forming Q
int launch_wy_apply(int D, int64_t nb, const Ws &ws, hipStream_t st);     // wy_apply.hip
// tql.hip
int launch_tql(int n, int64_t nb, const Ws &ws, int32_t *status, hipStream_t st);
// rotapply.hip
int launch_rotapply(int D, int64_t nb, const Ws &ws, hipStream_t st);
// rebuild.hip
int launch_dc(int n, int64_t nb, const Ws &ws, int32_t *status, hipStream_t st, bool rowmajor = true,
              bool colmap = false);   // dc.hip (colmap: leave the top level's deflated columns in place, write Ws::Wmap)
int64_t dc_final_offset(int n);                                                            // dc.hip
int launch_vgemm(int D, int64_t nb, const Ws &ws, hipStream_t st);                       // dc.hip
bool vgemm_big_supported(int D);                                                          // vgemm_big.hip
int launch_vgemm_big(int D, int64_t nb, const Ws &ws, hipStream_t st);                   // vgemm_big.hip (reads WT)
int launch_peaks(const float2 *phi, int64_t B, int xbase, int ybase, const double *Z, int nx, int ny,
                 const double *axis_x, const double *axis_y, const double *opt7, int iters, int max_peaks,
                 double *peaks, int32_t *counts, hipStream_t st);                          // peaks.hip
bool arrow_rebuild_supported(int D);                                                       // arrow.hip
int launch_arrow_rebuild(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *G, float *rn,
                         float *w_out, int32_t *status, const Ws &ws, hipStream_t st, bool lower_only = false);   // arrow.hip
bool back_rebuild_supported(int D);                                                        // backrebuild.hip
int launch_back_rebuild(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *G,
                        float *rn, float *w_out, const Ws &ws, hipStream_t st,
                        bool lower_only = false);                                        // backrebuild.hip
bool use_dc();                                                                          // api.hip
// image_dim: the dimension the eigenvector image ws.VT is laid out for (eig_dim(D) behind the dense pipeline, D itself
// behind the arrowhead solver of the first layer)
int launch_rebuild(int D, int64_t nb, const float *lw, const float2 *phi, const float *h,
                   float2 *G, float *rn, float *w_out, const Ws &ws, hipStream_t st, bool lower_only = false,
                   int image_dim = 0);
int launch_vout(int n, int64_t nb, float2 *V, float *w, const Ws &ws, hipStream_t st);
bool rebuild_big_supported(int D);                                                        // rebuild_big.hip, image dimension 256
int launch_rebuild_big(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *G, float *rn,
                       const Ws &ws, hipStream_t st, bool lower_only);
// zstep.hip
int launch_rn_sum(int64_t B, const float *rn, double *sum, hipStream_t st);
int launch_mean_from_sum(const double *sum, int64_t B, float *mean, hipStream_t st);
int launch_mean_from_pair(const double *sum_count, float *mean, hipStream_t st);   // mean = sum_count[0] / sum_count[1]
int launch_zstep(const float *lw, int D, int64_t B, const float *rn, const float *mean, float *alpha, hipStream_t st);
// head.hip
int launch_head(const admmnet_cfg *cfg, const float *hw, int64_t B, const float2 *phi, float *kv,
                float *out, hipStream_t st);
// spectrum.hip
int launch_spectrum_tables(const double *taus, int nx, int xbase, const double *fs, int ny, int ybase,
                           double2 *tabD, double2 *tabS, hipStream_t st);
int launch_spectrum_main(const float2 *phi, int64_t B, int xbase, int ybase, const double2 *tabD, int nx,
                         const double2 *tabS, int ny, double *out, hipStream_t st);

// spectral.hip
bool use_spectral();
bool use_spectral_fused();   // spectral_fused.hip: the whole evaluation in one kernel (default when the path is on)
int launch_spectral_fused(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *Z, float2 *G,
                          float *rn, int *flag, int32_t *status, float tol, const float *alpha, const float2 *phi_prev,
                          const float *h_prev, const float *lw_prev, int update_mode, hipStream_t st);
// update_mode 0: Z is current; 1 / 2: the fused kernel applies Z <- Z + alpha (G - C_prev) in its first sweep (2: stored Z still zero)
int launch_spectral(int D, int64_t nb, const float *lw, const float2 *phi, const float *h, float2 *Z, float2 *G,
                    float *rn, const Ws &ws, int32_t *status, hipStream_t st, bool lower_only, const float *alpha = nullptr,
                    const float2 *phi_prev = nullptr, const float *h_prev = nullptr, const float *lw_prev = nullptr,
                    int update_mode = 0);
// vdvh.hip (training route)
int launch_vdvh(int n, int64_t nb, const float2 *V, const float *d, float2 *out, hipStream_t st);
int launch_vhsv(int n, int64_t nb, const float2 *V, const float2 *S, float *q, hipStream_t st);

// synth.hip
int launch_synth(int64_t B, int Nb, int Nd, int L, unsigned long long seed, double snr_lo, double snr_hi, double snr_e,
                 double rho, int label_iters, float2 *y, float2 *b, float *sigma, float *tau, float *f, float2 *C,
                 float2 *phi_label, hipStream_t st);

// ---- optional per-kernel-class HIP-event profiler (bench.py roofline leg) ------
enum KernelClass { KC_PREP = 0, KC_TRIDIAG, KC_TQL, KC_ROTAPPLY, KC_REBUILD, KC_ZSTEP, KC_HEAD, KC_SPECTRUM, KC_GFUNC, KC_COUNT };
struct ProfScope {   // records start/stop events on `st` around a launcher body when profiling is on
    int slot;
    hipStream_t st;
    ProfScope(int kclass, hipStream_t s);
    ~ProfScope();
};

// ---- small device helpers ---------------------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cmacc(float2 acc, float2 a, float2 b) {  // acc + conj(a) * b
    acc.x = fmaf(a.x, b.x, fmaf(a.y, b.y, acc.x));
    acc.y = fmaf(a.x, b.y, fmaf(-a.y, b.x, acc.y));
    return acc;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace admmnet
