// api.hip -- the extern "C" boundary declared in include/admmnet.h: host-side
// weight packing, workspace carving and the per-layer launch sequence.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace admmnet {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiler -------------------------------------------------------------------
// Event pairs live in a pool that grows with the number of scopes recorded since the last read (the bench records
// ~600 scopes per cfg3 step; an earlier fixed pool of 8192 silently dropped everything after step 13).  A scope is
// only ever dropped when an event cannot be created; drops are counted and reported by admmnet_profile_dropped().
struct ProfState {
    std::mutex mu;
    bool on = false;
    std::vector<hipEvent_t> ev0, ev1;   // created events (reused across reads)
    std::vector<int> kclass;            // class of the scopes recorded since the last read
    int64_t dropped = 0;
};
static ProfState &prof() {
    static ProfState p;
    return p;
}
static const size_t kProfHardCap = (size_t)1 << 22;   // 4 M scopes between two reads: a runaway guard, not a budget

ProfScope::ProfScope(int kclass, hipStream_t s) : slot(-1), st(s) {
    ProfState &p = prof();
    if (!p.on) return;
    std::lock_guard<std::mutex> lk(p.mu);
    const size_t i = p.kclass.size();
    if (i >= p.ev0.size()) {
        hipEvent_t a, b;
        if (i >= kProfHardCap || hipEventCreate(&a) != hipSuccess) {
            ++p.dropped;
            return;
        }
        if (hipEventCreate(&b) != hipSuccess) {
            (void)hipEventDestroy(a);
            ++p.dropped;
            return;
        }
        p.ev0.push_back(a);
        p.ev1.push_back(b);
    }
    p.kclass.push_back(kclass);
    slot = (int)i;
    (void)hipEventRecord(p.ev0[i], st);
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    ProfState &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    (void)hipEventRecord(p.ev1[slot], st);
}

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

static int check_cfg(const admmnet_cfg *cfg) {
    if (!cfg) {
        set_error("cfg is NULL");
        return ADMMNET_E_ARG;
    }
    const int64_t D = (int64_t)cfg->M * cfg->N;
    if (cfg->M < 1 || cfg->N < 1 || D < 1 || D > kMaxD) {
        set_error("unsupported geometry M=%d N=%d (need 1 <= M*N <= %d)", cfg->M, cfg->N, kMaxD);
        return ADMMNET_E_ARG;
    }
    if (cfg->K < 1 || cfg->K > 1024) {
        set_error("unsupported num_layers K=%d", cfg->K);
        return ADMMNET_E_ARG;
    }
    if (cfg->has_head && (cfg->L < 1 || cfg->L > 16)) {
        set_error("unsupported L=%d", cfg->L);
        return ADMMNET_E_ARG;
    }
    return ADMMNET_OK;
}

// Tridiagonal eigensolver: divide & conquer (default) or QL + rotation replay (ADMMNET_EIG=ql).
bool use_dc() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("ADMMNET_EIG");
        v = (e && !strcmp(e, "ql")) ? 0 : 1;
    }
    return v == 1;
}

int64_t pick_chunk(const admmnet_cfg *cfg, int64_t B) {
    int64_t c = cfg->chunk > 0 ? cfg->chunk : 8192;
    if (c > B) c = B;
    if (c < 1) c = 1;
    return c;
}

// carve helper
struct Carver {
    char *base;
    int64_t off = 0;
    template <class T>
    T *take(int64_t count) {
        T *p = reinterpret_cast<T *>(base + off);
        off = align_up(off + (int64_t)sizeof(T) * count, 256);
        return p;
    }
};

// 128 < D < 256 ("padded route"): the layer matrix is embedded in the D = 256 pipeline as A' = diag(A, 0) -- arrow-first
// order puts the padding behind the last row of the D x D block.  The reflectors of A have exact zeros in the padded rows,
// the padded columns reduce to identity reflectors (tau = 0), so T' = diag(T, 0) exactly; the padded poles deflate (z = 0)
// with unit eigenvectors, the block reflectors leave those alone, and G' = V' f(L') V'^H = diag(G, f(0) I): the rebuild
// stores the leading block.  Costs the D = 256 flops whatever D is, still several times faster than the per-reflector
// sweep (tridiag_big.hip) it replaces as the default; any switch that leaves the panel / D&C / block-reflector route
// (ADMMNET_TRIDIAG_BIG=sweep, ADMMNET_BACK=q, ADMMNET_TRIDIAG=lds, ADMMNET_EIG=ql) also leaves the padding.
// Below kPadMin the sweep at the geometry's own size is faster than 256-sized work (measured on MI355X, K = 16, 4096
// signals, padded vs sweep per forward: D = 160 306 vs 280 ms, D = 176 312 vs 368 ms, D = 192 319 vs 396 ms) -- ADMMNET_PAD_MIN
// moves the switch.
static int pad_min() {
    // (with the matrix-function route on -- the default -- the eigen-pipeline only sees the matrices it rejects, and the route needs
    //  the lower-triangle state of the padded pipeline: every 128 < D < 256 is then padded, the crossover no longer matters)
    static const int v = getenv("ADMMNET_PAD_MIN") ? atoi(getenv("ADMMNET_PAD_MIN")) : (use_spectral() ? 129 : 176);
    return v;
}
int eig_dim(int D) { return (D > 128 && D >= pad_min() && D < 256 && use_wy_back(256)) ? 256 : D; }

static void carve_chunk(Carver &c, int Dact, int64_t chunk, Ws *ws) {
    const int D = eig_dim(Dact);
    const int64_t n = D + 1;
    ws->chunk = chunk;
    ws->cap = ((int64_t)kLogCapMul * n * n + 64 * n + 64 + 7) & ~(int64_t)7;   // whole 64-byte groups
    ws->Mbuf = c.take<float2>(chunk * ((int64_t)D * D + D + 1));
    ws->QV = c.take<float>(chunk * n * 2 * D);
    const int64_t groups = (chunk + 63) / 64;
    ws->dT = c.take<float>(groups * n * 64);
    ws->eT = c.take<float>(groups * n * 64);
    ws->w = c.take<float>(chunk * n);
    ws->w0 = c.take<float>(chunk * n);
    ws->logn = c.take<int>(chunk * 2);
    ws->Tfac = nullptr;
    ws->Tail = nullptr;
    ws->Wmap = nullptr;
    if (use_dc()) {
        ws->Wdc = c.take<float>(chunk * 3 * n * n);
        ws->VT = c.take<float>(chunk * n * 2 * D);
        if (tridiag_panel_supported(D)) {
            ws->Tfac = c.take<float2>(chunk * 17 * 256);
            ws->Tail = c.take<float2>(chunk * tridiag_panel_tail_elems());
            ws->Wmap = c.take<int2>(chunk * n);
        }
        ws->log = nullptr;
    } else {
        ws->log = c.take<LogRec>(chunk * ws->cap);
        ws->Wdc = nullptr;
        ws->VT = ws->QV;   // the rotation replay works in place
    }
    ws->spec_mat = nullptr;
    ws->spec_vec = nullptr;
    ws->spec_val = nullptr;
    ws->spec_flag = nullptr;
    ws->skip = nullptr;
    if (use_spectral()) {   // (in the layer's own dimension: the fast path never sees the padded image)
        const int64_t na = Dact + 1;
        if (!use_spectral_fused()) {   // (the multi-kernel form keeps A / E and E^2 in memory)
            ws->spec_mat = c.take<float2>(2 * chunk * na * na);
            ws->spec_vec = c.take<float2>(chunk * 2 * na);
            ws->spec_val = c.take<double>(chunk * 8);
        }
        ws->spec_flag = c.take<int>(chunk);
    }
}

int64_t eig_chunk_bytes(int D, int64_t chunk) {
    Carver c{nullptr};
    Ws ws;
    carve_chunk(c, D, chunk, &ws);
    return c.off;
}

// Two chunks in flight (ADMMNET_STREAMS=2): the per-chunk kernel sequence prep -> tridiagonalisation -> D&C -> back-transform ->
// rebuild of consecutive chunks alternates between two internal streams and two sets of chunk buffers, so that the
// vector-ALU-bound kernels of one chunk and the matrix-core-bound kernels of the other can share the CUs wherever their
// registers and LDS admit both.  Only for batches of at least two chunks; the state (G, Z, phi, h, rn) is shared -- the
// chunks touch disjoint slices of it.
static bool two_streams() {
    static const bool on = getenv("ADMMNET_STREAMS") && atoi(getenv("ADMMNET_STREAMS")) == 2;
    return on;
}

struct ChunkStreams {   // per device, created on first use (non-blocking streams: they order against the caller's by events)
    std::mutex mu;
    hipStream_t s[2] = {nullptr, nullptr};
    int dev = -1;
};
static int chunk_streams(hipStream_t out[2]) {
    static ChunkStreams cs[16];
    int dev = 0;
    ADMM_HIP(hipGetDevice(&dev));
    ChunkStreams &c = cs[dev & 15];
    std::lock_guard<std::mutex> lk(c.mu);
    if (c.s[0] == nullptr || c.dev != dev) {
        ADMM_HIP(hipStreamCreateWithFlags(&c.s[0], hipStreamNonBlocking));
        ADMM_HIP(hipStreamCreateWithFlags(&c.s[1], hipStreamNonBlocking));
        c.dev = dev;
    }
    out[0] = c.s[0];
    out[1] = c.s[1];
    return ADMMNET_OK;
}

static void carve_chunk(Carver &c, int Dact, int64_t chunk, Ws *ws);

// the second set of chunk buffers (same layout as the first; state pointers copied from `ws`)
static void carve_second_set(Carver &c, int D, const Ws &ws, Ws *ws2) {
    *ws2 = ws;
    carve_chunk(c, D, ws.chunk, ws2);
}

int carve_workspace(const admmnet_cfg *cfg, int64_t B, void *base, int64_t bytes, Ws *ws, bool state) {
    const int D = cfg->M * cfg->N;
    const int64_t n = D + 1;
    Carver c{reinterpret_cast<char *>(base)};
    memset(ws, 0, sizeof(*ws));
    if (state) {
        ws->G = c.take<float2>(B * n * n);
        ws->Z = c.take<float2>(B * n * n);
        for (int i = 0; i < 2; ++i) ws->phi[i] = c.take<float2>(B * D);
        for (int i = 0; i < 2; ++i) ws->h[i] = c.take<float>(B * D);
        ws->alpha = c.take<float>(B);
        ws->rn = c.take<float>(B);
        ws->sum = c.take<double>(2);
        ws->mean = c.take<float>(4);
        ws->headkv = c.take<float>((int64_t)2 * D * 128);
    }
    carve_chunk(c, D, pick_chunk(cfg, B), ws);
    ws->set2_offset = 0;
    if (state && two_streams() && B > ws->chunk) {   // room for a second chunk in flight
        ws->set2_offset = c.off;
        Ws tmp;
        carve_second_set(c, D, *ws, &tmp);
    }
    ws->total_bytes = c.off;
    if (base && bytes < c.off) {
        set_error("workspace too small: %lld < %lld bytes", (long long)bytes, (long long)c.off);
        return ADMMNET_E_WORKSPACE;
    }
    return ADMMNET_OK;
}

// The G-layer can take V = Q W inside its rebuild kernel (backrebuild.hip, D <= 128, D&C path):
// then the eigen-solve stops at (Q, W) and V never goes through memory.  ADMMNET_FUSE_BACK=0 keeps
// the separate kernels (tuning / debugging aid).
static bool fuse_back(int D, const Ws &ws) {
    static const bool on = !(getenv("ADMMNET_FUSE_BACK") && atoi(getenv("ADMMNET_FUSE_BACK")) == 0);
    return on && ws.Wdc && back_rebuild_supported(D);
}

// The first G-layer (Z = 0) sees a plain arrowhead matrix: arrow.hip solves it directly in O(n^2)
// (fused with the rebuild for D <= 128, through the global eigenvector image above).  ADMMNET_ARROW=0 sends it
// down the dense path like every other layer.
static bool use_arrow(int D) {
    static const bool on = !(getenv("ADMMNET_ARROW") && atoi(getenv("ADMMNET_ARROW")) == 0);
    return on && arrow_rebuild_supported(D);
}

// "Lean" state (D <= 128, register-resident tridiagonalisation, arrowhead first layer): G and Z are kept as
// lower triangles and the tridiagonalisation forms A = C - Z / rho itself, so the prep kernel only streams the
// lazy Z update (no A image is written or read).  ADMMNET_LEAN=0 keeps full storage + the image.
static bool use_lean(int D) {
    static const bool on = !(getenv("ADMMNET_LEAN") && atoi(getenv("ADMMNET_LEAN")) == 0);
    static const bool lds = getenv("ADMMNET_TRIDIAG") && !strcmp(getenv("ADMMNET_TRIDIAG"), "lds");
    static const bool sweep = getenv("ADMMNET_TRIDIAG_BIG") && !strcmp(getenv("ADMMNET_TRIDIAG_BIG"), "sweep");
    // D = 256 on the panel tridiagonalisation: the same idea in its "half image" form (prep.hip PM_HALF): lower-triangle
    // G / Z, image of the lower 16-block triangle -- exactly the tiles tridiag_panel_kernel loads
    if (D > 128) return on && !lds && !sweep && use_dc() && tridiag_panel_supported(eig_dim(D)) && use_arrow(D);
    return on && !lds && use_arrow(D);
}

static int eig_chunk(int Dact, int64_t nb, const Ws &ws, int32_t *status, hipStream_t st, bool with_v = true,
                     const float2 *Zlow = nullptr, const float2 *phi = nullptr, const float *h = nullptr,
                     const float *lw = nullptr) {
    const int D = eig_dim(Dact);   // (the image, T, W and the eigenvector image are all of this dimension)
    int rc;
    if ((rc = launch_tridiag(D, nb, ws, st, Zlow, phi, h, lw))) return rc;
    if (ws.Wdc) {   // divide & conquer + V = Q W on the matrix cores
        // the fused consumer (backrebuild.hip) and the large back-transform read the transposed image WT themselves
        const bool big = vgemm_big_supported(D);
        // (the block-reflector back-transform reads the eigenvectors through the column map of the top-level merge)
        const bool wy = big && use_wy_back(D) && with_v && ws.Wmap;
        if ((rc = launch_dc(D + 1, nb, ws, status, st, with_v && !big, wy))) return rc;
        if (!with_v) return ADMMNET_OK;
        if (wy) return launch_wy_apply(D, nb, ws, st);   // block reflectors applied to W: no explicit Q
        return big ? launch_vgemm_big(D, nb, ws, st) : launch_vgemm(D, nb, ws, st);
    }
    if ((rc = launch_tql(D + 1, nb, ws, status, st))) return rc;
    return launch_rotapply(D, nb, ws, st);
}

}  // namespace admmnet

using namespace admmnet;

extern "C" {

int admmnet_abi_version(void) { return ADMMNET_ABI_VERSION; }
const char *admmnet_last_error(void) { return g_err; }

int64_t admmnet_raw_weight_count(const admmnet_cfg *cfg) {
    if (check_cfg(cfg)) return -1;
    const int64_t D = (int64_t)cfg->M * cfg->N;
    const int64_t per_layer = 1 + 2 + 64 * D + 64 + D * 64 + D + 3 + 49 + 2 + 161;
    int64_t total = per_layer * cfg->K;
    if (cfg->has_head) {
        const int64_t L = cfg->L;
        total += 2 * D + 128 * 2 * D + 128 + 128 * 128 + 128 + 256 + 128 + 384 * 128 + 384 + 128 * 128 + 128 +
                 64 * 128 + 64 + 32 * 64 + 32 + 16 * 32 + 16 + L * 2 * (32 * 16 + 32 + 32 + 1) +
                 16 * 16 + 16 + 16 + 1;
    }
    return total;
}

int64_t admmnet_layer_weight_offset(const admmnet_cfg *cfg, int32_t k) {
    if (check_cfg(cfg)) return -1;
    const LayerLayout L{cfg->M * cfg->N};
    return (int64_t)k * L.size();
}

int64_t admmnet_packed_weight_count(const admmnet_cfg *cfg) {
    if (check_cfg(cfg)) return -1;
    const int D = cfg->M * cfg->N;
    const LayerLayout L{D};
    int64_t total = (int64_t)cfg->K * L.size();
    if (cfg->has_head) total += HeadLayout{D, cfg->L}.size();
    return total;
}

// transpose src[rows][cols] -> dst[cols][rows]
static void tr(const float *src, int rows, int cols, float *dst) {
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) dst[(size_t)c * rows + r] = src[(size_t)r * cols + c];
}

int admmnet_pack_weights(const admmnet_cfg *cfg, const float *raw, float *out) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!raw || !out) {
        set_error("pack_weights: NULL buffer");
        return ADMMNET_E_ARG;
    }
    const int D = cfg->M * cfg->N;
    const LayerLayout L{D};
    memset(out, 0, sizeof(float) * (size_t)admmnet_packed_weight_count(cfg));
    const float *p = raw;
    for (int k = 0; k < cfg->K; ++k) {
        float *o = out + (size_t)k * L.size();
        const float rho_phi = *p++;
        const float rho_h = *p++, pw = *p++;
        const float *w1 = p; p += 64 * D;
        const float *b1 = p; p += 64;
        const float *w2 = p; p += D * 64;
        const float *b2 = p; p += D;
        const float lam_g = *p++, rho_g = *p++, thr = *p++;
        const float *vn = p; p += 49;
        const float rho_z = *p++, lam_z = *p++;
        const float *rs = p; p += 161;
        // fp32 arithmetic, mirroring the reference tensors (admm_net.py:97,148,188,269-271,287-288,321,406,424-426,457)
        o[S_RHO_PHI] = softplus_f(rho_phi);
        o[S_RHO_H_EPS] = softplus_f(rho_h) + kEpsRef;
        o[S_SIG_PW] = sigmoid_f(pw);
        {
            const float lv = softplus_f(lam_g);
            o[S_CORNER_G] = 1.0f / (lv * lv + kEpsRef);
        }
        o[S_INV_RHO_G] = 1.0f / (softplus_f(rho_g) + kEpsRef);
        o[S_THR] = sigmoid_f(thr);
        o[S_RHO_Z] = softplus_f(rho_z);
        {
            const float lv = softplus_f(lam_z);
            o[S_CORNER_Z] = 1.0f / (lv * lv + kEpsRef);
        }
        o[S_KNORM] = (float)((double)k / 10.0);
        o[S_A_COEF] = 2.0f * sqrtf((float)D);
        tr(w1, 64, D, o + L.off_w1t());           // [64][D] -> [D][64]
        memcpy(o + L.off_b1(), b1, sizeof(float) * 64);
        tr(w2, D, 64, o + L.off_w2t());           // [D][64] -> [64][D]
        memcpy(o + L.off_b2(), b2, sizeof(float) * D);
        memcpy(o + L.off_vn(), vn, sizeof(float) * 49);
        memcpy(o + L.off_rs(), rs, sizeof(float) * 161);
    }
    if (cfg->has_head) {
        const HeadLayout H{D, cfg->L};
        float *o = out + (size_t)cfg->K * L.size();
        memcpy(o + H.off_pos(), p, sizeof(float) * 2 * D); p += 2 * D;
        tr(p, 128, 2 * D, o + H.off_fe0w()); p += 128 * 2 * D;
        memcpy(o + H.off_fe0b(), p, sizeof(float) * 128); p += 128;
        tr(p, 128, 128, o + H.off_fe2w()); p += 128 * 128;
        memcpy(o + H.off_fe2b(), p, sizeof(float) * 128); p += 128;
        memcpy(o + H.off_ppw(), p, sizeof(float) * 256); p += 256;
        memcpy(o + H.off_ppb(), p, sizeof(float) * 128); p += 128;
        tr(p, 384, 128, o + H.off_inw()); p += 384 * 128;
        memcpy(o + H.off_inb(), p, sizeof(float) * 384); p += 384;
        tr(p, 128, 128, o + H.off_outw()); p += 128 * 128;
        memcpy(o + H.off_outb(), p, sizeof(float) * 128); p += 128;
        tr(p, 64, 128, o + H.off_pe0w()); p += 64 * 128;
        memcpy(o + H.off_pe0b(), p, sizeof(float) * 64); p += 64;
        tr(p, 32, 64, o + H.off_pe2w()); p += 32 * 64;
        memcpy(o + H.off_pe2b(), p, sizeof(float) * 32); p += 32;
        tr(p, 16, 32, o + H.off_pe4w()); p += 16 * 32;
        memcpy(o + H.off_pe4b(), p, sizeof(float) * 16); p += 16;
        for (int t = 0; t < cfg->L; ++t) {
            float *r = o + H.off_reg(t);
            for (int q = 0; q < 2; ++q) {      // tau then f
                tr(p, 32, 16, r); p += 32 * 16;                     // [32][16] -> [16][32]
                memcpy(r + 512, p, sizeof(float) * 32); p += 32;    // b1
                memcpy(r + 544, p, sizeof(float) * 32); p += 32;    // w2
                r[576] = *p++;                                      // b2
                r += 577;
            }
        }
        float *c = o + H.off_conf();
        tr(p, 16, 16, c); p += 256;
        memcpy(c + 256, p, sizeof(float) * 16); p += 16;
        memcpy(c + 272, p, sizeof(float) * 16); p += 16;
        c[288] = *p++;
    }
    if (p - raw != admmnet_raw_weight_count(cfg)) {
        set_error("pack_weights: internal count mismatch %lld vs %lld", (long long)(p - raw),
                  (long long)admmnet_raw_weight_count(cfg));
        return ADMMNET_E_ARG;
    }
    return ADMMNET_OK;
}

int64_t admmnet_workspace_bytes(const admmnet_cfg *cfg, int64_t B) {
    if (check_cfg(cfg) || B < 1) return -1;
    Ws ws;
    carve_workspace(cfg, B, nullptr, 0, &ws, true);
    return ws.total_bytes;
}

int admmnet_begin(const admmnet_cfg *cfg, int64_t B, void *workspace, int64_t workspace_bytes,
                  int32_t *status, void *stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (B < 1 || !workspace) {
        set_error("begin: bad B or workspace");
        return ADMMNET_E_ARG;
    }
    Ws ws;
    if ((rc = carve_workspace(cfg, B, workspace, workspace_bytes, &ws, true))) return rc;
    if (status) ADMM_HIP(hipMemsetAsync(status, 0, 4 * sizeof(int32_t), (hipStream_t)stream));
    return ADMMNET_OK;
}

int admmnet_layer_front(const admmnet_cfg *cfg, const float *W, int32_t k, const void *y, const void *b,
                        const float *sigma, int64_t B, void *workspace, double *sum_out, int32_t *status,
                        void *stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (k < 0 || k >= cfg->K || B < 1 || !W || !y || !b || !sigma || !workspace) {
        set_error("layer_front: bad argument");
        return ADMMNET_E_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    Ws ws;
    carve_workspace(cfg, B, workspace, INT64_MAX, &ws, true);
    const int D = cfg->M * cfg->N;
    const int64_t n = D + 1;
    const LayerLayout L{D};
    const float2 *yy = (const float2 *)y, *bb = (const float2 *)b;
    if (k == cfg->K - 1) return launch_prep(cfg, W, k, yy, bb, sigma, 0, B, ws, true, st);
    const float *lw = W + (int64_t)k * L.size();
    const int cur = k & 1;
    // chunk buffers and streams: one set on the caller's stream, or two sets on two internal streams (two_streams())
    Ws sets[2] = {ws, ws};
    hipStream_t ss[2] = {st, st};
    const bool dual = ws.set2_offset != 0;
    if (dual) {
        Carver c2{reinterpret_cast<char *>(workspace) + ws.set2_offset};
        carve_second_set(c2, D, ws, &sets[1]);
        if ((rc = chunk_streams(ss))) return rc;
        hipEvent_t e0;
        ADMM_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
        ADMM_HIP(hipEventRecord(e0, st));                  // everything the caller has enqueued so far ...
        ADMM_HIP(hipStreamWaitEvent(ss[0], e0, 0));        // ... happens before the chunks
        ADMM_HIP(hipStreamWaitEvent(ss[1], e0, 0));
        ADMM_HIP(hipEventDestroy(e0));
    }
    int ci = 0;
    for (int64_t b0 = 0; b0 < B; b0 += ws.chunk, ++ci) {
        const int64_t nb = (B - b0 < ws.chunk) ? (B - b0) : ws.chunk;
        const Ws &wc = sets[ci & 1];
        hipStream_t sc = ss[ci & 1];
        const bool lean = use_lean(D);
        const float2 *phk = ws.phi[cur] + b0 * D;
        const float *hk = ws.h[cur] + b0 * D;
        float2 *Gk = ws.G + b0 * n * n;
        if (k == 0 && use_arrow(D)) {   // Z = 0: arrowhead, no matrix is ever formed
            if ((rc = launch_prep(cfg, W, k, yy, bb, sigma, b0, nb, wc, false, sc, true))) return rc;
            if ((rc = launch_arrow_rebuild(D, nb, lw, phk, hk, Gk, ws.rn + b0, nullptr, status, wc, sc, lean))) return rc;
            continue;
        }
        const bool fused = fuse_back(D, wc);
        // G as a matrix function where the spectrum allows it (checked per matrix, spectral.hip); the eigen-pipeline below then
        // only runs the matrices it flagged
        const bool spec = use_spectral() && wc.spec_flag && lean && D >= 8 && (D > 128 || fused);
        const bool late_image = spec && D > 128 && use_spectral_fused();   // (the image only for the flagged matrices, afterwards)
        // the lazy Z update of the previous layer rides the first sweep of the fused kernel: prep then only computes phi and h
        static const bool fold_env = !(getenv("ADMMNET_SF_FOLD") && atoi(getenv("ADMMNET_SF_FOLD")) == 0);
        const bool fold = spec && use_spectral_fused() && fold_env && k >= 1;
        if ((rc = launch_prep(cfg, W, k, yy, bb, sigma, b0, nb, wc, false, sc, false, lean, late_image, fold))) return rc;
        Ws wf = wc;
        if (spec) {
            const int prv = cur ^ 1;
            const float *lwp = k >= 1 ? W + (int64_t)(k - 1) * L.size() : lw;
            if ((rc = launch_spectral(D, nb, lw, phk, hk, ws.Z + b0 * n * n, Gk, ws.rn + b0, wc, status, sc, true,
                                      fold ? ws.alpha + b0 : nullptr, fold ? ws.phi[prv] + b0 * D : nullptr,
                                      fold ? ws.h[prv] + b0 * D : nullptr, fold ? lwp : nullptr, fold ? (k == 1 ? 2 : 1) : 0)))
                return rc;
            wf.skip = wc.spec_flag;
            if (late_image && (rc = launch_half_image(D, nb, lw, phk, hk, ws.Z + b0 * n * n, wf, sc))) return rc;
        }
        // (D <= 128: the tridiagonalisation's own loader forms A from the lower triangle of Z; D = 256 reads the half image)
        if ((rc = eig_chunk(D, nb, wf, status, sc, !fused, (lean && D <= 128) ? ws.Z + b0 * n * n : nullptr, phk, hk, lw)))
            return rc;
        rc = fused ? launch_back_rebuild(D, nb, lw, phk, hk, Gk, ws.rn + b0, nullptr, wf, sc, lean)
                   : launch_rebuild(D, nb, lw, phk, hk, Gk, ws.rn + b0, nullptr, wf, sc, lean, eig_dim(D));
        if (rc) return rc;
    }
    if (dual) {   // the caller's stream continues behind both chunk streams
        for (int q = 0; q < 2; ++q) {
            hipEvent_t e1;
            ADMM_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            ADMM_HIP(hipEventRecord(e1, ss[q]));
            ADMM_HIP(hipStreamWaitEvent(st, e1, 0));
            ADMM_HIP(hipEventDestroy(e1));
        }
    }
    return launch_rn_sum(B, ws.rn, sum_out ? sum_out : ws.sum, st);
}

int admmnet_layer_back(const admmnet_cfg *cfg, const float *W, int32_t k, int64_t B, void *workspace,
                       const float *mean_dev, void *stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (k < 0 || k >= cfg->K - 1 || !mean_dev) {
        set_error("layer_back: bad argument (k=%d)", k);
        return ADMMNET_E_ARG;
    }
    Ws ws;
    carve_workspace(cfg, B, workspace, INT64_MAX, &ws, true);
    const int D = cfg->M * cfg->N;
    const LayerLayout L{D};
    return launch_zstep(W + (int64_t)k * L.size(), D, B, ws.rn, mean_dev, ws.alpha, (hipStream_t)stream);
}

int admmnet_layer_back_pair(const admmnet_cfg *cfg, const float *W, int32_t k, int64_t B, void *workspace,
                            const double *sum_count_dev, void *stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (k < 0 || k >= cfg->K - 1 || !sum_count_dev) {
        set_error("layer_back_pair: bad argument (k=%d)", k);
        return ADMMNET_E_ARG;
    }
    Ws ws;
    carve_workspace(cfg, B, workspace, INT64_MAX, &ws, true);
    if ((rc = launch_mean_from_pair(sum_count_dev, ws.mean, (hipStream_t)stream))) return rc;
    const int D = cfg->M * cfg->N;
    const LayerLayout L{D};
    return launch_zstep(W + (int64_t)k * L.size(), D, B, ws.rn, ws.mean, ws.alpha, (hipStream_t)stream);
}

int admmnet_finish(const admmnet_cfg *cfg, const float *W, int64_t B, void *workspace, void *phi_out,
                   float *head_out, void *stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    Ws ws;
    carve_workspace(cfg, B, workspace, INT64_MAX, &ws, true);
    const int D = cfg->M * cfg->N;
    const float2 *phi = ws.phi[(cfg->K - 1) & 1];
    if (phi_out)
        ADMM_HIP(hipMemcpyAsync(phi_out, phi, sizeof(float2) * B * D, hipMemcpyDeviceToDevice, st));
    if (head_out) {
        if (!cfg->has_head) {
            set_error("finish: head_out given but cfg.has_head == 0");
            return ADMMNET_E_ARG;
        }
        const LayerLayout L{D};
        return launch_head(cfg, W + (int64_t)cfg->K * L.size(), B, phi, ws.headkv, head_out, st);
    }
    return ADMMNET_OK;
}

int admmnet_forward_f32(const admmnet_cfg *cfg, const float *W, const void *y, const void *b,
                        const float *sigma, int64_t B, void *phi_out, float *head_out, void *workspace,
                        int64_t workspace_bytes, int32_t *status, void *stream) {
    int rc;
    if ((rc = admmnet_begin(cfg, B, workspace, workspace_bytes, status, stream))) return rc;
    Ws ws;
    carve_workspace(cfg, B, workspace, workspace_bytes, &ws, true);
    hipStream_t st = (hipStream_t)stream;
    for (int k = 0; k < cfg->K; ++k) {
        if ((rc = admmnet_layer_front(cfg, W, k, y, b, sigma, B, workspace, ws.sum, status, stream))) return rc;
        if (k < cfg->K - 1) {
            if ((rc = launch_mean_from_sum(ws.sum, B, ws.mean, st))) return rc;
            if ((rc = admmnet_layer_back(cfg, W, k, B, workspace, ws.mean, stream))) return rc;
        }
    }
    return admmnet_finish(cfg, W, B, workspace, phi_out, head_out, stream);
}

int admmnet_glayer_f32(const admmnet_cfg *cfg, const float *lw, const void *phi, const float *h,
                       const void *Z, int64_t B, void *G_out, float *w_out, float *rn_out, void *workspace,
                       int64_t workspace_bytes, int32_t *status, void *stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (B < 1 || !lw || !phi || !h || !G_out || !workspace) {
        set_error("glayer: bad argument");
        return ADMMNET_E_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const int D = cfg->M * cfg->N;
    const int64_t n = D + 1;
    // workspace: [rn scratch B floats | chunk buffers]; weights scalars are needed on the host
    float sc[S_COUNT];
    ADMM_HIP(hipMemcpyAsync(sc, lw, sizeof(sc), hipMemcpyDeviceToHost, st));
    ADMM_HIP(hipStreamSynchronize(st));   // test/utility entry point only
    admmnet_cfg c2 = *cfg;
    Ws ws;
    char *base = (char *)workspace;
    const int64_t rn_bytes = align_up(sizeof(float) * B, 256);
    if ((rc = carve_workspace(&c2, B, base + rn_bytes, workspace_bytes - rn_bytes, &ws, false))) return rc;
    float *rn_tmp = (float *)base;
    if (status) ADMM_HIP(hipMemsetAsync(status, 0, 4 * sizeof(int32_t), st));
    for (int64_t b0 = 0; b0 < B; b0 += ws.chunk) {
        const int64_t nb = (B - b0 < ws.chunk) ? (B - b0) : ws.chunk;
        const float2 *ph = (const float2 *)phi + b0 * D;
        const float2 *Zc = Z ? (const float2 *)Z + b0 * n * n : nullptr;
        if (!Zc && use_arrow(D)) {
            if ((rc = launch_arrow_rebuild(D, nb, lw, ph, h + b0 * D, (float2 *)G_out + b0 * n * n,
                                           rn_out ? rn_out + b0 : rn_tmp + b0, w_out ? w_out + b0 * n : nullptr,
                                           status, ws, st)))
                return rc;
            continue;
        }
        if ((rc = launch_build_block(D, nb, sc[S_CORNER_G], sc[S_INV_RHO_G], ph, h + b0 * D, Zc, ws, st))) return rc;
        const bool fused = fuse_back(D, ws);
        if ((rc = eig_chunk(D, nb, ws, status, st, !fused))) return rc;
        float2 *Go = (float2 *)G_out + b0 * n * n;
        float *rno = rn_out ? rn_out + b0 : rn_tmp + b0, *wo = w_out ? w_out + b0 * n : nullptr;
        rc = fused ? launch_back_rebuild(D, nb, lw, ph, h + b0 * D, Go, rno, wo, ws, st)
                   : launch_rebuild(D, nb, lw, ph, h + b0 * D, Go, rno, wo, ws, st, false, eig_dim(D));
        if (rc) return rc;
    }
    return ADMMNET_OK;
}

int64_t admmnet_eigh_workspace_bytes(int32_t n, int64_t B) {
    if (n < 2 || n - 1 > kMaxD || B < 1) return -1;
    admmnet_cfg cfg = {n - 1, 1, 3, 1, 0, 0, {0, 0}};
    Ws ws;
    carve_workspace(&cfg, B, nullptr, 0, &ws, false);
    return ws.total_bytes;
}

int64_t admmnet_glayer_workspace_bytes(const admmnet_cfg *cfg, int64_t B) {
    if (check_cfg(cfg) || B < 1) return -1;
    Ws ws;
    carve_workspace(cfg, B, nullptr, 0, &ws, false);
    return ws.total_bytes + align_up(sizeof(float) * B, 256);
}

int admmnet_eigh_c64(int32_t n, int64_t B, const void *A, float *w, void *V, void *workspace,
                     int64_t workspace_bytes, int32_t *status, void *stream) {
    if (n < 2 || n - 1 > kMaxD || B < 1 || !A || !w || !V || !workspace) {
        set_error("eigh: bad argument (n=%d)", n);
        return ADMMNET_E_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    admmnet_cfg cfg = {n - 1, 1, 3, 1, 0, 0, {0, 0}};
    Ws ws;
    int rc;
    if ((rc = carve_workspace(&cfg, B, workspace, workspace_bytes, &ws, false))) return rc;
    if (status) ADMM_HIP(hipMemsetAsync(status, 0, 4 * sizeof(int32_t), st));
    const int D = n - 1;
    for (int64_t b0 = 0; b0 < B; b0 += ws.chunk) {
        const int64_t nb = (B - b0 < ws.chunk) ? (B - b0) : ws.chunk;
        if ((rc = launch_build_generic(n, nb, (const float2 *)A + b0 * n * n, ws, st))) return rc;
        if ((rc = eig_chunk(D, nb, ws, status, st))) return rc;
        if ((rc = launch_vout(n, nb, (float2 *)V + b0 * (int64_t)n * n, w + b0 * n, ws, st))) return rc;
    }
    return ADMMNET_OK;
}

int admmnet_vdvh_c64(int32_t n, int64_t B, const void *V, const float *d, void *out, void *stream) {
    if (n < 1 || n - 1 > kMaxD || B < 1 || !V || !d || !out) {
        set_error("vdvh: bad argument (n=%d)", n);
        return ADMMNET_E_ARG;
    }
    return launch_vdvh(n, B, (const float2 *)V, d, (float2 *)out, (hipStream_t)stream);
}

int admmnet_vhsv_f32(int32_t n, int64_t B, const void *V, const void *S, float *q, void *stream) {
    if (n < 1 || n - 1 > kMaxD || B < 1 || !V || !S || !q) {
        set_error("vhsv: bad argument (n=%d)", n);
        return ADMMNET_E_ARG;
    }
    return launch_vhsv(n, B, (const float2 *)V, (const float2 *)S, q, (hipStream_t)stream);
}

int admmnet_profile_enable(int32_t on) {
    ProfState &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    p.on = on != 0;
    p.kclass.clear();
    p.dropped = 0;
    return ADMMNET_OK;
}

int64_t admmnet_profile_dropped(void) {
    ProfState &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    return p.dropped;
}

int admmnet_profile_read(double *ms_total, int64_t *launches, int32_t nclasses) {
    if (!ms_total || !launches || nclasses < KC_COUNT) {
        set_error("profile_read: need %d classes", (int)KC_COUNT);
        return ADMMNET_E_ARG;
    }
    for (int i = 0; i < nclasses; ++i) { ms_total[i] = 0.0; launches[i] = 0; }
    ProfState &p = prof();
    std::lock_guard<std::mutex> lk(p.mu);
    for (size_t i = 0; i < p.kclass.size(); ++i) {
        ADMM_HIP(hipEventSynchronize(p.ev1[i]));
        float ms = 0.f;
        ADMM_HIP(hipEventElapsedTime(&ms, p.ev0[i], p.ev1[i]));
        ms_total[p.kclass[i]] += ms;
        launches[p.kclass[i]] += 1;
    }
    p.kclass.clear();
    return ADMMNET_OK;
}

int64_t admmnet_spectrum_workspace_bytes(int32_t xbase, int32_t ybase, int32_t nx, int32_t ny) {
    if (xbase < 1 || ybase < 1 || nx < 1 || ny < 1) return -1;
    return align_up(sizeof(double2) * (int64_t)nx * xbase, 256) + align_up(sizeof(double2) * (int64_t)ny * ybase, 256);
}

int admmnet_spectrum_f64(const void *phi, int64_t B, int32_t xbase, int32_t ybase, const double *taus,
                         int32_t nx, const double *fs, int32_t ny, double *out, void *workspace,
                         int64_t workspace_bytes, void *stream) {
    const int64_t need = admmnet_spectrum_workspace_bytes(xbase, ybase, nx, ny);
    if (need < 0 || B < 1 || !phi || !taus || !fs || !out || !workspace) {
        set_error("spectrum: bad argument");
        return ADMMNET_E_ARG;
    }
    if (workspace_bytes < need) {
        set_error("spectrum: workspace too small");
        return ADMMNET_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    double2 *tabD = (double2 *)workspace;
    double2 *tabS = (double2 *)((char *)workspace + align_up(sizeof(double2) * (int64_t)nx * xbase, 256));
    int rc;
    if ((rc = launch_spectrum_tables(taus, nx, xbase, fs, ny, ybase, tabD, tabS, st))) return rc;
    return launch_spectrum_main((const float2 *)phi, B, xbase, ybase, tabD, nx, tabS, ny, out, st);
}

int64_t admmnet_peak_search_workspace_bytes(int32_t xbase, int32_t ybase, int32_t nx, int32_t ny, int64_t B) {
    const int64_t t = admmnet_spectrum_workspace_bytes(xbase, ybase, nx, ny);
    if (t < 0 || B < 1) return -1;
    return align_up(t, 256) + align_up((int64_t)sizeof(double) * B * nx * ny, 256);
}

int admmnet_peak_search_f64(const void *phi, int64_t B, int32_t xbase, int32_t ybase, const double *axis_x,
                            int32_t nx, const double *axis_y, int32_t ny, const double *opts7, int32_t iters,
                            int32_t max_peaks, double *peaks, int32_t *counts, void *workspace,
                            int64_t workspace_bytes, void *stream) {
    const int64_t need = admmnet_peak_search_workspace_bytes(xbase, ybase, nx, ny, B);
    if (need < 0 || !phi || !axis_x || !axis_y || !opts7 || !peaks || !counts || !workspace || nx < 1 || ny < 1 ||
        iters < 0 || max_peaks < 1) {
        set_error("peak search: bad argument");
        return ADMMNET_E_ARG;
    }
    if (workspace_bytes < need) {
        set_error("peak search: workspace too small");
        return ADMMNET_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t tb = align_up(admmnet_spectrum_workspace_bytes(xbase, ybase, nx, ny), 256);
    double *Z = (double *)((char *)workspace + tb);
    int rc;
    if ((rc = admmnet_spectrum_f64(phi, B, xbase, ybase, axis_x, nx, axis_y, ny, Z, workspace, tb, stream))) return rc;
    return launch_peaks((const float2 *)phi, B, xbase, ybase, Z, nx, ny, axis_x, axis_y, opts7, iters, max_peaks,
                        peaks, counts, st);
}

int admmnet_regional_maxima_f64(const double *Z, int64_t B, int32_t nx, int32_t ny, int32_t max_peaks,
                                double *peaks, int32_t *counts, void *stream) {
    if (!Z || B < 1 || nx < 1 || ny < 1 || max_peaks < 1 || !peaks || !counts) {
        set_error("regional maxima: bad argument");
        return ADMMNET_E_ARG;
    }
    const double o7[7] = {0, 0, 0, 0, 0, 0, 0};
    // the maxima stage of the peak-search kernel alone: no phi, no axes (positions = pixel column / row), no rounds
    return launch_peaks(nullptr, B, 1, 1, Z, nx, ny, nullptr, nullptr, o7, 0, max_peaks, peaks, counts,
                        (hipStream_t)stream);
}

int admmnet_synth_batch(int64_t B, int32_t Nb, int32_t Nd, int32_t L, uint64_t seed, double snr_lo, double snr_hi,
                        double snr_e, double rho, int32_t label_iters, void *y, void *b, float *sigma, float *tau, float *f,
                        void *C, void *phi_label, void *stream) {
    if (B < 1 || Nb < 1 || Nd < 1 || (int64_t)Nb * Nd > 4096 || L < 1 || L > 8 || !y || !b || !sigma || !tau || !f || !C ||
        label_iters < 0) {
        set_error("synth_batch: bad argument");
        return ADMMNET_E_ARG;
    }
    return launch_synth(B, Nb, Nd, L, seed, snr_lo, snr_hi, snr_e, rho, label_iters, (float2 *)y, (float2 *)b, sigma, tau,
                        f, (float2 *)C, (float2 *)phi_label, (hipStream_t)stream);
}

}  // extern "C"
