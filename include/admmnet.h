/*
 * admmnet.h -- C ABI of the MI355X (gfx950) ADMM-Net forward path.
 *
 * This is the drop-in boundary for ONE hot path of E-J408/admm-net: the
 * K-layer unrolled ADMM-Net forward.  Every entry point names the reference
 * interface it replaces (file:line under /root/reference).  The reference is
 * pure Python/torch with no FFI of its own, so the "binding a maintainer would
 * add" is the ctypes stub shown in INTEGRATION.md (admm_net_amd/_lib.py is that
 * stub, shipped).
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every call returns 0 on
 *     success or a negative ADMMNET_E_* code, message via admmnet_last_error()
 *     (thread-local).
 *   - the CALLER owns every buffer (device and host); nothing here allocates
 *     device memory.  All device work is enqueued on the caller's stream and is
 *     asynchronous; no call synchronises the device.
 *   - device = the caller's current HIP device.
 *   - complex64 = interleaved (re, im) float pairs, as torch.complex64.
 *   - D = M*N (signal length), n = D + 1 (state dimension), B = batch,
 *     K = number of unrolled layers.
 */
#ifndef ADMMNET_H
#define ADMMNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADMMNET_ABI_VERSION 1

enum {
    ADMMNET_OK = 0,
    ADMMNET_E_ARG = -1,        /* bad argument / unsupported shape            */
    ADMMNET_E_HIP = -2,        /* a HIP runtime call failed                   */
    ADMMNET_E_WORKSPACE = -3,  /* workspace too small                         */
    ADMMNET_E_NOCONV = -4      /* eigensolver did not converge / log overflow  */
};

/* Model geometry: admm_net.py:726-741 / :770-789 (ctor arguments). */
typedef struct admmnet_cfg {
    int32_t M;          /* reference "M" (= Nb)                                */
    int32_t N;          /* reference "N" (= Nd)                                */
    int32_t L;          /* max targets of the PeakSearchLayer head (3)         */
    int32_t K;          /* num_layers                                          */
    int32_t has_head;   /* 1: ADMMNet (PeakSearchLayer), 0: PhiEstADMMNet       */
    int32_t chunk;      /* signals per eigensolver work chunk (0 = auto)       */
    int32_t reserved[2];
} admmnet_cfg;

int         admmnet_abi_version(void);
const char *admmnet_last_error(void);

/* ---- weights -------------------------------------------------------------
 * Raw (host) order, all float32, concatenated:
 *   for k in 0..K-1:
 *     phiLayers.k.rho
 *     hLayers.k.rho, hLayers.k.projection_weight,
 *     hLayers.k.correction_net.0.weight[64][D], .0.bias[64],
 *     hLayers.k.correction_net.2.weight[D][64], .2.bias[D]
 *     gLayers.k.lambda_param, gLayers.k.rho, gLayers.k.threshold,
 *     gLayers.k.value_net.0.weight[16], .0.bias[16], .2.weight[16], .2.bias[1]
 *     zLayers.k.rho, zLayers.k.lambda_param,
 *     zLayers.k.residual_scale_net.0.weight[32][3], .0.bias[32], .2.weight[32], .2.bias[1]
 *   if has_head: peakSearchLayer.* in this order:
 *     position_encoder[D][2], feature_extractor.0.{weight[128][2D],bias[128]},
 *     feature_extractor.2.{weight[128][128],bias[128]},
 *     position_projection.{weight[128][2],bias[128]},
 *     attention.in_proj_weight[384][128], in_proj_bias[384],
 *     attention.out_proj.{weight[128][128],bias[128]},
 *     peak_extractor.0.{weight[64][128],bias[64]}, .2.{[32][64],[32]}, .4.{[16][32],[16]},
 *     for t in 0..L-1: tau_regressor.t.0.{[32][16],[32]}, .2.{[32],[1]},
 *                      f_regressor.t.0.{[32][16],[32]},   .2.{[32],[1]}
 *     confidence_net.0.{[16][16],[16]}, .2.{[16],[1]}
 * (zLayers.k.step_adjust_net is in the state_dict but never used by forward,
 *  admm_net.py:381-386, and is not part of the raw buffer.)
 *
 * admmnet_pack_weights resolves every softplus / sigmoid / reciprocal scalar
 * on the host (replaces the three .item() host syncs per layer at
 * admm_net.py:271,426,458) and lays the MLPs out for coalesced device reads.
 */
int64_t admmnet_raw_weight_count(const admmnet_cfg *cfg);     /* floats */
int64_t admmnet_packed_weight_count(const admmnet_cfg *cfg);  /* floats */
int     admmnet_pack_weights(const admmnet_cfg *cfg, const float *raw_host,
                             float *packed_host);

/* ---- workspace -------------------------------------------------------------*/
int64_t admmnet_workspace_bytes(const admmnet_cfg *cfg, int64_t B);

/* ---- whole forward ----------------------------------------------------------
 * Replaces PhiEstADMMNet.forward (admm_net.py:742-764) and ADMMNet.forward
 * (admm_net.py:791-816) with the batch mean of ZLayer (admm_net.py:459) taken
 * over the B signals of this call.
 *   y, b      device complex64 [B][D]
 *   sigma     device float32   [B]
 *   phi_out   device complex64 [B][D]
 *   head_out  device float32   [3][B][L] (tau, f, confidence) or NULL
 *   status    device int32     [4] or NULL, zeroed by the call: [0] = #matrices whose eigensolver
 *             failed (must be 0).  The G-layer (admm_net.py:237-354: eigh, eigenvalue map f, V f(L) V^H) is
 *             evaluated as a matrix function wherever the per-matrix checks of csrc/spectral.hip allow it
 *             (ADMMNET_SPECTRAL=0: never) and through the eigensolver otherwise; [1] = #matrix-layers that
 *             went through the eigensolver after a rejection, [2] = #matrix-layers evaluated as a matrix
 *             function, [3] = of [1], those rejected because f is not a quadratic on the bulk of the spectrum.
 */
int admmnet_forward_f32(const admmnet_cfg *cfg, const float *weights_dev,
                        const void *y, const void *b, const float *sigma,
                        int64_t B, void *phi_out, float *head_out,
                        void *workspace, int64_t workspace_bytes,
                        int32_t *status, void *stream);

/* ---- layer-at-a-time API (multi-GPU "global" batch-mean scope) --------------
 * layer_front(k): lazy Z update with the step of layer k-1, phi/H/G of layer k
 *   (admm_net.py:806-810) and r_b = ||G_b - C_b||_F; writes the LOCAL sum of
 *   r_b to sum_out[0] and the local count B to sum_out[1] (device float64 [2]).
 *   For k == K-1 only phi is produced.
 * layer_back(k): ZLayer step (admm_net.py:443-474) from a caller-supplied
 *   batch mean (device float, e.g. all-reduced sum / global B).
 * begin() zeroes the per-forward state; finish() writes phi_out (+ head).
 */
int admmnet_begin(const admmnet_cfg *cfg, int64_t B, void *workspace,
                  int64_t workspace_bytes, int32_t *status, void *stream);
int admmnet_layer_front(const admmnet_cfg *cfg, const float *weights_dev, int32_t k,
                        const void *y, const void *b, const float *sigma, int64_t B,
                        void *workspace, double *sum_out, int32_t *status, void *stream);
int admmnet_layer_back(const admmnet_cfg *cfg, const float *weights_dev, int32_t k,
                       int64_t B, void *workspace, const float *mean_dev, void *stream);
/* The same from the (all-reduced) pair the protocol carries: sum_count_dev = device float64 [2] = (sum of r_b, number of
 * signals) over all ranks -- the mean of admm_net.py:459 is formed on the device, no host arithmetic between the calls. */
int admmnet_layer_back_pair(const admmnet_cfg *cfg, const float *weights_dev, int32_t k,
                            int64_t B, void *workspace, const double *sum_count_dev, void *stream);
int admmnet_finish(const admmnet_cfg *cfg, const float *weights_dev, int64_t B,
                   void *workspace, void *phi_out, float *head_out, void *stream);

/* ---- building blocks (exported for unit tests and reuse) --------------------
 * Batched Hermitian eigen-function  G = V f(Lambda) V^H  of the block matrix
 *   A = [[diag(h), phi],[phi^H, corner]] - inv_rho * Z
 * i.e. GLayer.forward (admm_net.py:237-354).  Z may be NULL (treated as 0).
 *   G_out  device complex64 [B][n][n];  w_out device float [B][n] (eigenvalues,
 *   unsorted) or NULL;  rn_out device float [B] = ||G - [[diag h, phi],[phi^H,
 *   corner_z]]||_F or NULL.
 *   layer_weights: packed weights of ONE layer (admmnet_layer_weight_offset).
 */
int64_t admmnet_layer_weight_offset(const admmnet_cfg *cfg, int32_t k);   /* floats */
int64_t admmnet_glayer_workspace_bytes(const admmnet_cfg *cfg, int64_t B);
int admmnet_glayer_f32(const admmnet_cfg *cfg, const float *layer_weights,
                       const void *phi, const float *h, const void *Z, int64_t B,
                       void *G_out, float *w_out, float *rn_out,
                       void *workspace, int64_t workspace_bytes,
                       int32_t *status, void *stream);

/* Batched Hermitian eigendecomposition of arbitrary complex64 Hermitian
 * matrices (torch.linalg.eigh at admm_net.py:303).  A [B][n][n] (only the
 * lower triangle is read); w [B][n] unsorted; V [B][n][n] row-major,
 * columns = eigenvectors. */
int64_t admmnet_eigh_workspace_bytes(int32_t n, int64_t B);
int admmnet_eigh_c64(int32_t n, int64_t B, const void *A, float *w, void *V,
                     void *workspace, int64_t workspace_bytes,
                     int32_t *status, void *stream);

/* Training route (trainPhi.py / train.py call forward in train mode).
 * admmnet_vdvh_c64: out = V diag(d) V^H, exactly Hermitian -- GLayer._rebuild_definite_matrix (admm_net.py:336-354: two
 *   bmm + the symmetrisation) and the backward of the eigenvalue-only eigh, dL/dA = V diag(dL/dw) V^H (admm_net.py:303-306,
 *   V detached).   V device complex64 [B][n][n] (columns = eigenvectors), d device float [B][n], out complex64 [B][n][n].
 * admmnet_vhsv_f32: its adjoint, q[c] = Re(v_c^H S v_c) for a Hermitian S (lower triangle read): the gradient of out with
 *   respect to d for an incoming S = (g + g^H) / 2.   S device complex64 [B][n][n], q device float [B][n]. */
int admmnet_vdvh_c64(int32_t n, int64_t B, const void *V, const float *d, void *out, void *stream);
int admmnet_vhsv_f32(int32_t n, int64_t B, const void *V, const void *S, float *q, void *stream);

/* Spectrum |phi^H kron(s(f), conj d(tau))|^2 on a (tau, f) grid:
 * peak_search_func / peak_search, utils/peakSearchUtils.py:9-60, evaluated in
 * float64 like the reference.
 *   phi device complex64 [B][ybase*xbase] (index ks*xbase + kd);
 *   taus [nx], fs [ny] device float64;  out device float64 [B][ny][nx];
 *   workspace: device scratch of admmnet_spectrum_workspace_bytes() bytes. */
int64_t admmnet_spectrum_workspace_bytes(int32_t xbase, int32_t ybase, int32_t nx, int32_t ny);
int admmnet_spectrum_f64(const void *phi, int64_t B, int32_t xbase, int32_t ybase,
                         const double *taus, int32_t nx, const double *fs, int32_t ny,
                         double *out, void *workspace, int64_t workspace_bytes, void *stream);

/* Batched grid peak search on that spectrum: alt_peak_search, utils/peakSearchUtils.py:63-173
 * (coarse grid -> regional maxima, skimage local_maxima(connectivity=2) semantics -> `iters`
 * refinement rounds), one signal per workgroup.
 *   axis_x [nx], axis_y [ny]: the coarse grid (np.arange(xmin, xmax - xstep, xstep) and the
 *     reference's np.arange(ymin, ymax - xstep, ystep), :105-106), device float64;
 *   opts7 (host): xmin, xmax, xstep, ymin, ymax, ystep, reducefactor;
 *   peaks device float64 [B][max_peaks][3] = (x = tau, y = f, height) in np.where row-major order
 *     of the coarse maxima (callers sort by height, main_for_net.py:119); rows >= count are not written;
 *   counts device int32 [B]: number of regional maxima found (may exceed max_peaks: truncated);
 *   workspace: device scratch of admmnet_peak_search_workspace_bytes() bytes (tables + coarse spectra). */
int64_t admmnet_peak_search_workspace_bytes(int32_t xbase, int32_t ybase, int32_t nx, int32_t ny, int64_t B);
int admmnet_peak_search_f64(const void *phi, int64_t B, int32_t xbase, int32_t ybase,
                            const double *axis_x, int32_t nx, const double *axis_y, int32_t ny,
                            const double *opts7, int32_t iters, int32_t max_peaks, double *peaks,
                            int32_t *counts, void *workspace, int64_t workspace_bytes, void *stream);

/* The regional-maxima stage of that kernel alone, on caller-supplied images: skimage.morphology.local_maxima(
 * connectivity=2) as used at utils/peakSearchUtils.py:118 (8-connected, plateau aware, borders allowed, a constant
 * image has none; the reference's own example input is the plateau matrix at :427-432).
 *   Z device float64 [B][ny][nx];  peaks device float64 [B][max_peaks][3] = (column, row, 0) of every maximum
 *   pixel in np.where row-major order;  counts device int32 [B]. */
int admmnet_regional_maxima_f64(const double *Z, int64_t B, int32_t nx, int32_t ny, int32_t max_peaks,
                                double *peaks, int32_t *counts, void *stream);

/* Batched scene synthesis + classical-solver labels on the device: generate_data.py:133-221 (_generate_single_sample,
 * _generate_communication_symbols) and the phi label of DatasetGeneratorCreatePhi (:410-463 = admm_for_us(y, b, ...),
 * which as written is the recursion phi_k = W (y / b + rho phi_{k-1}) stopped at min_iter).
 *   outputs (device): y, b complex64 [B][Nb*Nd]; sigma float [B]; tau, f float [B][L]; C complex64 [B][L];
 *   phi_label complex64 [B][Nb*Nd] or NULL.  snr_lo..snr_hi: range of the per-sample SNR in dB (equal = fixed);
 *   snr_e: demodulation SNR (7); rho, label_iters: the classical solver's rho (1) and its iteration count (5).
 *   Counter-based generator: (seed, sample index) fixes a sample regardless of B or launch geometry. */
int admmnet_synth_batch(int64_t B, int32_t Nb, int32_t Nd, int32_t L, uint64_t seed, double snr_lo, double snr_hi,
                        double snr_e, double rho, int32_t label_iters, void *y, void *b, float *sigma, float *tau,
                        float *f, void *C, void *phi_label, void *stream);

/* ---- measurement hooks (bench.py roofline leg) ---------------------------------
 * When enabled, every kernel launcher brackets its launch with HIP events on the
 * caller's stream.  admmnet_profile_read synchronises those events, returns the
 * summed milliseconds and launch counts per kernel class and clears the buffer.
 * Classes: 0 prep, 1 tridiag, 2 tridiagonal eigensolver, 3 back-transform (V = Q W),
 * 4 rebuild, 5 zstep, 6 head, 7 spectrum, 8 G-layer as a matrix function (ADMMNET_KERNEL_CLASSES entries).
 * The event pool grows with the number of launches between two reads; a launch is only left out when an event
 * cannot be created, and admmnet_profile_dropped() returns how many were (0 in any healthy run; bench.py refuses
 * to print per-step sums otherwise).  Guarded by one mutex; off by default. */
#define ADMMNET_KERNEL_CLASSES 9
int admmnet_profile_enable(int32_t on);
int admmnet_profile_read(double *ms_total, int64_t *launches, int32_t nclasses);
int64_t admmnet_profile_dropped(void);

#ifdef __cplusplus
}
#endif
#endif /* ADMMNET_H */
