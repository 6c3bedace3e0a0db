/*
 * orb_api.hip -- host side of the C-ABI for the ORB extractor (include/orbslam_amd.h).
 *
 * Owns the handle (device buffers sized for max_w x max_h x max_batch, pinned staging, one HIP
 * stream), restates the reference constructor's tables (src/ORBextractor.cpp:360-420), derives the
 * per-level geometry / FAST cell grid (:720-762) / cv::resize coefficient tables, and enqueues the
 * kernel pipeline:  pyramid (1 copy + nlevels-1 resizes) -> FAST cells -> quadtree -> blur -> describe.
 * There is no CPU fallback anywhere in this file: without a HIP device create() fails.
 */
#include "orb_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static thread_local std::string g_err;
static void set_err(const char *what, hipError_t e, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_err = buf;
}
#define HIPCHK(expr)                                       \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) {                            \
            set_err(#expr, _e, __FILE__, __LINE__);        \
            return ORB_E_HIP;                              \
        }                                                  \
    } while (0)

extern "C" const char *orb_last_error(void) { return g_err.c_str(); }
void orb_set_last_error(const char *msg) { g_err = msg ? msg : ""; }

/* ORBSLAM_AMD_TRACE_EXIT=1 (tools/exit_hang_probe.sh): stderr markers that bracket this library's share of the exit
 * handlers.  orb_exit_marker_after_modules is registered from a priority constructor, i.e. BEFORE the hipcc module
 * constructors register their __hip_module_dtor, so it runs AFTER the library's code objects were unregistered;
 * orb_trace_exit_late() is called by ba_create and registers a handler that runs BEFORE them. */
static void orb_exit_marker_after_modules() { fprintf(stderr, "[liborbslam_amd] exit: code objects unregistered (all __hip_module_dtor done)\n"); }
static void orb_exit_marker_before_modules() { fprintf(stderr, "[liborbslam_amd] exit: handler registered at the first create() runs now\n"); }
__attribute__((constructor(101))) static void orb_register_exit_markers() {
    if (getenv("ORBSLAM_AMD_TRACE_EXIT")) atexit(orb_exit_marker_after_modules);
}
void orb_trace_exit_late() {
    static bool done = false;
    if (!done && getenv("ORBSLAM_AMD_TRACE_EXIT")) atexit(orb_exit_marker_before_modules);
    done = true;
}

extern "C" int orb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static inline int cv_round(double v) { return (int)std::lrint(v); } /* cvRound: round half to even */
static inline int cv_floor(double v) {
    int i = (int)v;
    return i - (i > v);
}
static inline int cv_ceil(double v) {
    int i = (int)v;
    return i + (i < v);
}
static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct OrbExtractor {
    OrbConfig cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t last_launch_stream = nullptr; /* stream of the most recent orb_extract_batch_device call */
    int max_w = 0, max_h = 0, max_batch = 0;
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> nfeat;
    int umax[16];
    double scale_factor_d = 0; /* include/ORBextractor.h:84: `double scaleFactor` initialised from a float */

    OrbGeom geom{};
    bool geom_valid = false;
    std::vector<CellDesc> h_cells;
    std::vector<int32_t> h_tab_i;
    std::vector<int16_t> h_tab_c;

    /* device allocations and their capacities */
    OrbBuffers buf{};
    CellDesc *d_cells = nullptr;
    int32_t *d_tab_i = nullptr;
    int16_t *d_tab_c = nullptr;
    size_t cap_pyr = 0, cap_blur = 0, cap_slab = 0, cap_cellcnt = 0, cap_keysA = 0, cap_keysB = 0, cap_lvlkp = 0,
           cap_cand = 0, cap_lvlcnt = 0, cap_cells = 0, cap_tab_i = 0, cap_tab_c = 0;

    /* staging for the host-pointer API */
    uint8_t *d_in = nullptr, *h_in = nullptr;
    size_t cap_in = 0;
    OrbKeyPoint *d_kps = nullptr, *h_kps = nullptr;
    uint8_t *d_desc = nullptr, *h_desc = nullptr;
    int32_t *d_counts = nullptr, *h_counts = nullptr;
    int out_cap = 0;
    int last_frames = 0;
    bool profile = false;
    static const int kProfRing = 64;
    hipEvent_t ev[64][8] = {};
    long prof_calls = 0;
    /* the blur only needs the pyramid, so it runs on an auxiliary stream beside FAST + quadtree */
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap = true;
    /* HIP graph of the whole host-pointer call (H2D, 12 kernels incl. the fork/join, 3 x D2H): captured on the
       second call with the same (w, h, n_frames) and replayed afterwards -- one launch instead of 16 */
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    long graph_gen = -1;       /* value of `generation` the graph was captured at */
    int graph_w = 0, graph_h = 0, graph_n = 0, graph_seen = 0;
    long generation = 0;       /* bumped whenever a device / staging buffer or the geometry changes */
    bool use_graph = true;
};

/* reference src/ORBextractor.cpp:360-420 */
static void build_tables(OrbExtractor *h) {
    const int nlevels = h->cfg.nlevels;
    const double scaleFactor = (double)h->cfg.scale_factor;
    h->scale_factor_d = scaleFactor;
    h->scale.assign(nlevels, 1.f);
    h->sigma2.assign(nlevels, 1.f);
    for (int i = 1; i < nlevels; i++) {
        h->scale[i] = (float)(h->scale[i - 1] * scaleFactor);
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    h->inv_scale.resize(nlevels);
    h->inv_sigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
        h->inv_scale[i] = 1.0f / h->scale[i];
        h->inv_sigma2[i] = 1.0f / h->sigma2[i];
    }
    h->nfeat.assign(nlevels, 0);
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = h->cfg.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        h->nfeat[l] = cv_round(nDesired);
        sum += h->nfeat[l];
        nDesired *= factor;
    }
    h->nfeat[nlevels - 1] = std::max(h->cfg.nfeatures - sum, 0);
    /* circular patch rows */
    const int HP = 15;
    int v, v0, vmax = cv_floor(HP * std::sqrt(2.f) / 2 + 1);
    int vmin = cv_ceil(HP * std::sqrt(2.f) / 2);
    const double hp2 = HP * HP;
    for (v = 0; v <= vmax; ++v) h->umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (v = HP, v0 = 0; v >= vmin; --v) {
        while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
        h->umax[v] = v0;
        ++v0;
    }
}

static void level_size(const OrbExtractor *h, int w, int hgt, int l, int &lw, int &lh) {
    float s = h->inv_scale[l]; /* src/ORBextractor.cpp:1075-1076 */
    lw = cv_round((float)w * s);
    lh = cv_round((float)hgt * s);
}

/* cv::resize INTER_LINEAR coefficient tables (OpenCV 2.4/3.x imgwarp.cpp, 11-bit fixed point) */
static void resize_tables(int ssize, int dsize, int32_t *ofs, int16_t *coef, bool clamp_x) {
    double inv_scale = (double)dsize / ssize, scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (clamp_x) {
            if (s < 0) {
                f = 0;
                s = 0;
            }
            if (s >= ssize - 1) {
                f = 0;
                s = ssize - 1;
            }
        }
        ofs[d] = s;
        int c0 = cv_round((1.f - f) * 2048), c1 = cv_round(f * 2048);
        coef[2 * d] = (int16_t)std::min(std::max(c0, -32768), 32767);
        coef[2 * d + 1] = (int16_t)std::min(std::max(c1, -32768), 32767);
    }
}

static int build_geom(OrbExtractor *h, int w, int hgt) {
    OrbGeom g{};
    g.nlevels = h->cfg.nlevels;
    g.w = w;
    g.h = hgt;
    g.ini_th = h->cfg.ini_th_fast;
    g.min_th = h->cfg.min_th_fast;
    h->h_cells.clear();
    h->h_tab_i.clear();
    h->h_tab_c.clear();
    int64_t pyr_off = 0, key_off = 0;
    int slab_off = 0, out_base = 0;
    for (int l = 0; l < g.nlevels; l++) {
        LevelGeom &L = g.lv[l];
        level_size(h, w, hgt, l, L.w, L.h);
        if (L.w < 1 || L.h < 1) {
            g_err = "pyramid level degenerates to zero size";
            return ORB_E_INVALID;
        }
        if (L.w - 32 > 4095 || L.h - 32 > 4095) {
            g_err = "level larger than 4127 px: packed 12-bit key coordinates would overflow";
            return ORB_E_INVALID;
        }
        L.pitch = (int)align_up(L.w, 64);
        L.plane_off = pyr_off;
        pyr_off += align_up((int64_t)L.pitch * L.h, 256);
        L.maxBX = L.w - ORB_EDGE_THRESHOLD + 3;
        L.maxBY = L.h - ORB_EDGE_THRESHOLD + 3;
        L.scale = h->scale[l];
        L.patch_size = (float)(int)(31 * h->scale[l]);
        L.quota = h->nfeat[l];
        L.cell_base = (int)h->h_cells.size();
        L.key_off = key_off;
        L.out_base = out_base;
        /* resize tables for l >= 1 */
        if (l > 0) {
            const LevelGeom &S = g.lv[l - 1];
            L.xofs_off = (int)h->h_tab_i.size();
            h->h_tab_i.resize(h->h_tab_i.size() + L.w);
            h->h_tab_c.resize(h->h_tab_c.size() + 2 * L.w);
            resize_tables(S.w, L.w, &h->h_tab_i[L.xofs_off], &h->h_tab_c[2 * L.xofs_off], true);
            L.yofs_off = (int)h->h_tab_i.size();
            h->h_tab_i.resize(h->h_tab_i.size() + L.h);
            h->h_tab_c.resize(h->h_tab_c.size() + 2 * L.h);
            resize_tables(S.h, L.h, &h->h_tab_i[L.yofs_off], &h->h_tab_c[2 * L.yofs_off], false);
            /* LDS window of the largest 128 x 32 output tile (k_resize stages exactly this much) */
            const int32_t *xo = &h->h_tab_i[L.xofs_off], *yo = &h->h_tab_i[L.yofs_off];
            int max_dw = 0, max_rows = 0;
            for (int tx0 = 0; tx0 < L.w; tx0 += 128) {
                const int x_last = std::min(tx0 + 128, L.w) - 1;
                const int lo = xo[tx0] & ~3, hi = std::min(xo[x_last] + 1, S.w - 1);
                max_dw = std::max(max_dw, (hi - lo) / 4 + 1);
            }
            for (int ty0 = 0; ty0 < L.h; ty0 += 32) {
                const int y_last = std::min(ty0 + 32, L.h) - 1;
                const int lo = std::min(std::max(yo[ty0], 0), S.h - 1), hi = std::min(std::max(yo[y_last] + 1, 0), S.h - 1);
                max_rows = std::max(max_rows, hi - lo + 1);
            }
            L.rs_lw = max_dw * 4;
            L.rs_lh = max_rows;
            if ((int64_t)L.rs_lw * L.rs_lh > 40 * 1024) L.rs_lw = L.rs_lh = 0; /* huge scale step: gather path */
        }
        /* FAST cell grid, src/ORBextractor.cpp:729-762 */
        const int minB = ORB_MIN_BORDER;
        const float width = (float)(L.maxBX - minB), height = (float)(L.maxBY - minB);
        const float W = 30;
        L.nCols = (width > 0) ? (int)(width / W) : 0;
        L.nRows = (height > 0) ? (int)(height / W) : 0;
        L.n_cells = 0;
        L.key_cap = 0;
        L.nIni = 0;
        L.hX = 1.f;
        if (L.nCols >= 1 && L.nRows >= 1) {
            L.wCell = (int)std::ceil(width / L.nCols);
            L.hCell = (int)std::ceil(height / L.nRows);
            for (int i = 0; i < L.nRows; i++) {
                const float iniY = (float)(minB + i * L.hCell);
                float maxY = iniY + L.hCell + 6;
                if (iniY >= L.maxBY - 3) continue;
                if (maxY > L.maxBY) maxY = (float)L.maxBY;
                for (int j = 0; j < L.nCols; j++) {
                    const float iniX = (float)(minB + j * L.wCell);
                    float maxX = iniX + L.wCell + 6;
                    if (iniX >= L.maxBX - 6) continue;
                    if (maxX > L.maxBX) maxX = (float)L.maxBX;
                    CellDesc c{};
                    c.level = (short)l;
                    c.iniX = (short)iniX;
                    c.iniY = (short)iniY;
                    c.rw = (short)((int)maxX - (int)iniX);
                    c.rh = (short)((int)maxY - (int)iniY);
                    c.offX = (short)(j * L.wCell);
                    c.offY = (short)(i * L.hCell);
                    int iw = std::max(c.rw - 6, 0), ih = std::max(c.rh - 6, 0);
                    c.slab_cap = ((iw + 1) / 2) * ((ih + 1) / 2); /* NMS keeps <= 1 px per 2x2 block */
                    c.slab_off = slab_off;
                    slab_off += c.slab_cap;
                    L.key_cap += c.slab_cap;
                    if (c.rw > ORB_CELL_MAX - 1 || c.rh > ORB_CELL_MAX - 1) {
                        g_err = "FAST cell larger than the LDS tile";
                        return ORB_E_INVALID;
                    }
                    h->h_cells.push_back(c);
                    L.n_cells++;
                    g.cell_max_rw = std::max<int>(g.cell_max_rw, c.rw);
                    g.cell_max_rh = std::max<int>(g.cell_max_rh, c.rh);
                }
            }
            /* quadtree roots, src/ORBextractor.cpp:493-495 */
            L.nIni = (int)std::round(static_cast<float>(L.maxBX - minB) / (L.maxBY - minB));
            if (L.nIni < 1) {
                g_err = "image too tall: quadtree root count rounds to 0 (reference divides by zero)";
                return ORB_E_ASPECT;
            }
            L.hX = static_cast<float>(L.maxBX - minB) / L.nIni;
        }
        L.out_cap = std::max(L.quota + 4, 4 * L.nIni);
        if (L.out_cap > ORB_QT_MAX_NODES) {
            g_err = "per-level feature quota exceeds the quadtree list capacity (2044)";
            return ORB_E_INVALID;
        }
        key_off += align_up(L.key_cap, 64);
        out_base += L.out_cap;
    }
    g.total_cells = (int)h->h_cells.size();
    g.total_out_cap = out_base;
    g.pyr_frame_bytes = pyr_off;
    g.key_frame_count = std::max<int64_t>(key_off, 64);
    g.slab_frame_count = std::max(slab_off, 64);
    h->geom = g;
    return ORB_OK;
}

static thread_local long *g_generation_sink = nullptr; /* set by the caller of grow(); per host thread (one handle per thread) */
template <typename T> static int grow(T *&p, size_t &cap, size_t need) {
    if (need <= cap && p) return ORB_OK;
    if (g_generation_sink) (*g_generation_sink)++;
    if (p) HIPCHK(hipFree(p));
    p = nullptr;
    cap = 0;
    HIPCHK(hipMalloc((void **)&p, std::max<size_t>(need * sizeof(T), 256)));
    cap = need;
    return ORB_OK;
}

static int ensure_buffers(OrbExtractor *h, int n_frames) {
    g_generation_sink = &h->generation;
    const OrbGeom &g = h->geom;
    const size_t F = (size_t)std::max(n_frames, h->max_batch);
    int rc;
    if ((rc = grow(h->buf.pyr, h->cap_pyr, F * g.pyr_frame_bytes))) return rc;
    if ((rc = grow(h->buf.blur, h->cap_blur, F * g.pyr_frame_bytes))) return rc;
    if ((rc = grow(h->buf.slab, h->cap_slab, F * g.slab_frame_count))) return rc;
    if ((rc = grow(h->buf.cell_cnt, h->cap_cellcnt, F * std::max(g.total_cells, 1)))) return rc;
    if ((rc = grow(h->buf.keysA, h->cap_keysA, F * g.key_frame_count))) return rc;
    if ((rc = grow(h->buf.keysB, h->cap_keysB, F * g.key_frame_count))) return rc;
    if ((rc = grow(h->buf.lvl_kp, h->cap_lvlkp, F * g.total_out_cap))) return rc;
    if ((rc = grow(h->buf.cand_cnt, h->cap_cand, F * ORB_MAX_LEVELS))) return rc;
    if ((rc = grow(h->buf.lvl_cnt, h->cap_lvlcnt, F * ORB_MAX_LEVELS))) return rc;
    return ORB_OK;
}

static int set_geometry(OrbExtractor *h, int w, int hgt) {
    if (h->geom_valid && h->geom.w == w && h->geom.h == hgt) return ORB_OK;
    h->geom_valid = false;
    h->generation++;
    g_generation_sink = &h->generation;
    int rc = build_geom(h, w, hgt);
    if (rc) return rc;
    /* previous launches may still read the old tables -- on the handle's stream or on whatever stream the caller of
       orb_extract_batch_device last passed (nothing else orders the blocking table copies below against those) */
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->last_launch_stream && h->last_launch_stream != h->stream) HIPCHK(hipStreamSynchronize(h->last_launch_stream));
    if ((rc = grow(h->d_cells, h->cap_cells, std::max<size_t>(h->h_cells.size(), 1)))) return rc;
    if ((rc = grow(h->d_tab_i, h->cap_tab_i, std::max<size_t>(h->h_tab_i.size(), 1)))) return rc;
    if ((rc = grow(h->d_tab_c, h->cap_tab_c, std::max<size_t>(h->h_tab_c.size(), 2)))) return rc;
    /* (on the handle's own stream: a synchronous hipMemcpy runs on the legacy stream, and ROCm 7.2 fails legacy-stream
       operations -- or the capture -- while ANOTHER host thread's extractor captures its launch graph) */
    if (!h->h_cells.empty())
        HIPCHK(hipMemcpyAsync(h->d_cells, h->h_cells.data(), h->h_cells.size() * sizeof(CellDesc), hipMemcpyHostToDevice, h->stream));
    if (!h->h_tab_i.empty()) {
        HIPCHK(hipMemcpyAsync(h->d_tab_i, h->h_tab_i.data(), h->h_tab_i.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipMemcpyAsync(h->d_tab_c, h->h_tab_c.data(), h->h_tab_c.size() * 2, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    h->buf.cells = h->d_cells;
    h->buf.tab_i = h->d_tab_i;
    h->buf.tab_c = h->d_tab_c;
    h->geom_valid = true;
    return ORB_OK;
}

static void free_staging(OrbExtractor *h) {
    (void)hipFree(h->d_in);
    (void)hipFree(h->d_kps);
    (void)hipFree(h->d_desc);
    (void)hipFree(h->d_counts);
    if (h->h_in) (void)hipHostFree(h->h_in);
    if (h->h_kps) (void)hipHostFree(h->h_kps);
    if (h->h_desc) (void)hipHostFree(h->h_desc);
    if (h->h_counts) (void)hipHostFree(h->h_counts);
    h->d_in = h->h_in = nullptr;
    h->d_kps = h->h_kps = nullptr;
    h->d_desc = h->h_desc = nullptr;
    h->d_counts = h->h_counts = nullptr;
}

extern "C" int orb_max_keypoints(const OrbExtractor *h);

/* pinned + device staging of the host-pointer API, sized for batch frames of w x hgt */
static int alloc_staging(OrbExtractor *h, int w, int hgt, int batch) {
    free_staging(h);
    h->generation++;
    h->out_cap = orb_max_keypoints(h);
    const size_t in_bytes = (size_t)batch * align_up(w, 64) * hgt;
    h->cap_in = in_bytes;
    if (hipMalloc((void **)&h->d_in, in_bytes) != hipSuccess || hipHostMalloc((void **)&h->h_in, in_bytes) != hipSuccess ||
        hipMalloc((void **)&h->d_kps, (size_t)batch * h->out_cap * sizeof(OrbKeyPoint)) != hipSuccess ||
        hipHostMalloc((void **)&h->h_kps, (size_t)batch * h->out_cap * sizeof(OrbKeyPoint)) != hipSuccess ||
        hipMalloc((void **)&h->d_desc, (size_t)batch * h->out_cap * 32) != hipSuccess ||
        hipHostMalloc((void **)&h->h_desc, (size_t)batch * h->out_cap * 32) != hipSuccess ||
        hipMalloc((void **)&h->d_counts, (size_t)batch * 4) != hipSuccess ||
        hipHostMalloc((void **)&h->h_counts, (size_t)batch * 4) != hipSuccess) {
        g_err = "staging allocation failed";
        return ORB_E_HIP;
    }
    h->max_w = w;
    h->max_h = hgt;
    h->max_batch = batch;
    return ORB_OK;
}

extern "C" int orb_max_keypoints(const OrbExtractor *h) {
    if (!h) return 0;
    /* per level the quadtree stops at >= quota nodes and can overshoot by at most 3 -- but it never returns fewer
       than the children of its nIni = round(W / H) root nodes allow, so a wide frame with a small budget can
       yield up to 4 * nIni keypoints per level (src/ORBextractor.cpp:493-534).  The bound is that of the frame
       size the handle currently holds (the creation size until a call brings another one). */
    int tot = 0;
    for (int l = 0; l < h->cfg.nlevels; l++) tot += h->nfeat[l] + 4;
    if (h->geom_valid) tot = std::max(tot, h->geom.total_out_cap);
    return tot;
}

extern "C" int orb_create(const OrbConfig *cfg, int max_w, int max_h, int max_batch, int device_id,
                          OrbExtractor **out) {
    if (!cfg || !out || max_w <= 0 || max_h <= 0 || max_batch <= 0) {
        g_err = "orb_create: bad argument";
        return ORB_E_INVALID;
    }
    if (cfg->nlevels < 1 || cfg->nlevels > ORB_MAX_LEVELS || cfg->nfeatures < 1 || !(cfg->scale_factor > 1.0f) ||
        cfg->ini_th_fast < 1 || cfg->min_th_fast < 1 || cfg->min_th_fast > cfg->ini_th_fast ||
        cfg->ini_th_fast > 254) {
        g_err = "orb_create: unsupported ORBextractor configuration";
        return ORB_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) {
        g_err = "orb_create: no usable HIP device (this library has no CPU fallback)";
        return ORB_E_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device_id));
    OrbExtractor *h = new OrbExtractor();
    h->cfg = *cfg;
    h->device = device_id;
    h->max_w = max_w;
    h->max_h = max_h;
    h->max_batch = max_batch;
    build_tables(h);
    int rc;
    auto fail = [&](int code) {
        orb_destroy(h);
        return code;
    };
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        g_err = "hipStreamCreate failed";
        return fail(ORB_E_HIP);
    }
    h->overlap = getenv("ORBSLAM_AMD_NO_OVERLAP") == nullptr;
    h->use_graph = getenv("ORBSLAM_AMD_NO_GRAPH") == nullptr;
    if (orb_upload_constants(h->device, h->stream) != hipSuccess) {
        g_err = "constant upload failed";
        return fail(ORB_E_HIP);
    }
    if ((rc = set_geometry(h, max_w, max_h))) return fail(rc);
    if ((rc = ensure_buffers(h, max_batch))) return fail(rc);
    if ((rc = alloc_staging(h, max_w, max_h, max_batch))) return fail(rc);
    *out = h;
    return ORB_OK;
}

extern "C" void orb_destroy(OrbExtractor *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->buf.pyr);
    (void)hipFree(h->buf.blur);
    (void)hipFree(h->buf.slab);
    (void)hipFree(h->buf.cell_cnt);
    (void)hipFree(h->buf.keysA);
    (void)hipFree(h->buf.keysB);
    (void)hipFree(h->buf.cand_cnt);
    (void)hipFree(h->buf.lvl_kp);
    (void)hipFree(h->buf.lvl_cnt);
    (void)hipFree(h->d_cells);
    (void)hipFree(h->d_tab_i);
    (void)hipFree(h->d_tab_c);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    free_staging(h);
    for (int r = 0; r < OrbExtractor::kProfRing; r++)
        for (int i = 0; i < 8; i++)
            if (h->ev[r][i]) (void)hipEventDestroy(h->ev[r][i]);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->aux) (void)hipStreamDestroy(h->aux);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int orb_get_levels(const OrbExtractor *h) { return h ? h->cfg.nlevels : 0; }
extern "C" float orb_get_scale_factor(const OrbExtractor *h) { return h ? (float)h->scale_factor_d : 0.f; }
extern "C" int orb_get_tables(const OrbExtractor *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                              int32_t *fpl, int32_t *umax16) {
    if (!h) return ORB_E_INVALID;
    for (int i = 0; i < h->cfg.nlevels; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->inv_scale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
        if (fpl) fpl[i] = h->nfeat[i];
    }
    if (umax16)
        for (int i = 0; i < 16; i++) umax16[i] = h->umax[i];
    return ORB_OK;
}
extern "C" int orb_get_level_sizes(const OrbExtractor *h, int w, int hgt, int32_t *ws, int32_t *hs) {
    if (!h || !ws || !hs) return ORB_E_INVALID;
    for (int l = 0; l < h->cfg.nlevels; l++) {
        int lw, lh;
        level_size(h, w, hgt, l, lw, lh);
        ws[l] = lw;
        hs[l] = lh;
    }
    return ORB_OK;
}
extern "C" int orb_set_profiling(OrbExtractor *h, int enable) {
    if (!h) return ORB_E_INVALID;
    HIPCHK(hipSetDevice(h->device));
    if (enable && !h->ev[0][0])
        for (int r = 0; r < OrbExtractor::kProfRing; r++)
            for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&h->ev[r][i]));
    h->profile = enable != 0;
    h->prof_calls = 0;
    return ORB_OK;
}
/* average stage durations over the (at most 64 most recent) calls recorded since profiling was enabled */
extern "C" int orb_get_stage_ms(OrbExtractor *h, float *ms5) {
    if (!h || !ms5 || !h->ev[0][0] || h->prof_calls == 0) return ORB_E_INVALID;
    HIPCHK(hipSetDevice(h->device));
    const int n = (int)std::min<long>(h->prof_calls, OrbExtractor::kProfRing);
    for (int i = 0; i < 5; i++) ms5[i] = 0.f;
    for (int r = 0; r < n; r++) {
        HIPCHK(hipEventSynchronize(h->ev[r][5]));
        HIPCHK(hipEventSynchronize(h->ev[r][7]));
        static const int a[5] = {0, 1, 2, 6, 4}, b[5] = {1, 2, 3, 7, 5}; /* pyramid, FAST, quadtree, blur, describe */
        for (int i = 0; i < 5; i++) {
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, h->ev[r][a[i]], h->ev[r][b[i]]));
            ms5[i] += t / n;
        }
    }
    h->prof_calls = 0;
    return ORB_OK;
}
extern "C" void *orb_stream(OrbExtractor *h) { return h ? (void *)h->stream : nullptr; }
extern "C" int orb_synchronize(OrbExtractor *h) {
    if (!h) return ORB_E_INVALID;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return ORB_OK;
}

extern "C" int orb_extract_batch_device(OrbExtractor *h, const uint8_t *d_imgs, int n_frames, int w, int hgt, int stride,
                                        int64_t frame_stride, OrbKeyPoint *d_kps, uint8_t *d_desc, int cap,
                                        int32_t *d_counts, void *stream) {
    if (!h || !d_imgs || !d_kps || !d_desc || !d_counts || n_frames <= 0 || w <= 0 || hgt <= 0 || stride < w) {
        g_err = "orb_extract_batch_device: bad argument";
        return ORB_E_INVALID;
    }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if ((rc = set_geometry(h, w, hgt))) return rc;
    if (cap < orb_max_keypoints(h)) { /* the bound of THIS frame size (wide frames: 4 * nIni per level) */
        g_err = "orb_extract_batch_device: cap must be >= orb_max_keypoints() for this frame size";
        return ORB_E_CAPACITY;
    }
    if ((rc = ensure_buffers(h, n_frames))) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    h->last_launch_stream = st;
    const OrbGeom &g = h->geom;
    const bool prof = h->profile;
    hipEvent_t *ev = h->ev[h->prof_calls % OrbExtractor::kProfRing];
    /* events: 0 start | 1 pyramid done | 2 FAST done | 3 quadtree done | 4 describe start | 5 describe done |
               6 blur start (aux stream) | 7 blur done (aux stream) */
    if (prof) HIPCHK(hipEventRecord(ev[0], st));
    HIPCHK(launch_pyramid(g, h->buf, d_imgs, stride, frame_stride, n_frames, st));
    if (prof) HIPCHK(hipEventRecord(ev[1], st));
    HIPCHK(launch_fast(g, h->buf, n_frames, st));
    if (prof) HIPCHK(hipEventRecord(ev[2], st));
    if (h->overlap) {
        /* fork: the blur (throughput-bound, needs only the pyramid) runs on the auxiliary stream BESIDE the
           quadtree (latency-bound: few resident wavefronts, mostly waiting), and is joined before describe */
        HIPCHK(hipEventRecord(h->ev_fork, st));
        HIPCHK(hipStreamWaitEvent(h->aux, h->ev_fork, 0));
        if (prof) HIPCHK(hipEventRecord(ev[6], h->aux));
        HIPCHK(launch_blur(g, h->buf, n_frames, h->aux));
        if (prof) HIPCHK(hipEventRecord(ev[7], h->aux));
        HIPCHK(hipEventRecord(h->ev_join, h->aux));
    }
    HIPCHK(launch_quadtree(g, h->buf, n_frames, st));
    if (prof) HIPCHK(hipEventRecord(ev[3], st));
    if (h->overlap) {
        HIPCHK(hipStreamWaitEvent(st, h->ev_join, 0)); /* join */
    } else {
        if (prof) HIPCHK(hipEventRecord(ev[6], st));
        HIPCHK(launch_blur(g, h->buf, n_frames, st));
        if (prof) HIPCHK(hipEventRecord(ev[7], st));
    }
    if (prof) HIPCHK(hipEventRecord(ev[4], st));
    HIPCHK(launch_describe(g, h->buf, n_frames, d_kps, d_desc, cap, d_counts, st));
    if (prof) HIPCHK(hipEventRecord(ev[5], st));
    if (prof) h->prof_calls++;
    h->last_frames = n_frames;
    return ORB_OK;
}

extern "C" int orb_extract_batch(OrbExtractor *h, const uint8_t *const *imgs, int n_frames, int w, int hgt, int stride,
                                 OrbKeyPoint *kps_out, uint8_t *desc_out, int cap, int32_t *n_out) {
    if (!h || !n_out || n_frames < 0) {
        g_err = "orb_extract_batch: bad argument";
        return ORB_E_INVALID;
    }
    for (int f = 0; f < n_frames; f++) n_out[f] = 0;
    if (n_frames == 0 || !imgs || w <= 0 || hgt <= 0) return ORB_OK; /* empty image: silent return (:1004-1005) */
    if (stride < w || !kps_out || !desc_out) {
        g_err = "orb_extract_batch: bad argument";
        return ORB_E_INVALID;
    }
    if (n_frames > h->max_batch) {
        g_err = "orb_extract_batch: batch larger than the handle was created for";
        return ORB_E_CAPACITY;
    }
    HIPCHK(hipSetDevice(h->device));
    int rc;
    if (w > h->max_w || hgt > h->max_h) { /* a larger frame than announced: grow the staging once */
        HIPCHK(hipStreamSynchronize(h->stream));
        if ((rc = alloc_staging(h, std::max(w, h->max_w), std::max(hgt, h->max_h), h->max_batch))) return rc;
    }
    if ((rc = set_geometry(h, w, hgt))) return rc;
    if (orb_max_keypoints(h) > h->out_cap) { /* this frame shape can return more keypoints than the staging holds */
        HIPCHK(hipStreamSynchronize(h->stream));
        if ((rc = alloc_staging(h, h->max_w, h->max_h, h->max_batch))) return rc;
    }
    const int pitch = h->geom.lv[0].pitch;
    const size_t fbytes = (size_t)pitch * hgt;
    for (int f = 0; f < n_frames; f++)
        if (!imgs[f]) {
            g_err = "orb_extract_batch: null frame";
            return ORB_E_INVALID;
        }
    {
        /* frames -> pinned staging block (pitch-aligned rows); a batch is copied by a few host threads: one core
           moves ~10 GB/s, the 64-frame batch of the bench is 20 MB against 57 GB/s of PCIe behind it */
        auto stage = [&](int t, int nt) {
            for (int f = t; f < n_frames; f += nt) {
                uint8_t *dst = h->h_in + f * fbytes;
                if (stride == pitch && w == pitch)
                    memcpy(dst, imgs[f], fbytes);
                else
                    for (int y = 0; y < hgt; y++) memcpy(dst + (size_t)y * pitch, imgs[f] + (size_t)y * stride, w);
            }
        };
        const int nt = std::max(1, std::min<int>(n_frames / 4, std::min<int>(8, (int)std::thread::hardware_concurrency())));
        if (nt == 1)
            stage(0, 1);
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < nt; t++) pool.emplace_back(stage, t, nt);
            for (auto &th : pool) th.join();
        }
    }
    if ((rc = ensure_buffers(h, n_frames))) return rc;
    auto enqueue = [&]() -> int {
        HIPCHK(hipMemcpyAsync(h->d_in, h->h_in, fbytes * n_frames, hipMemcpyHostToDevice, h->stream));
        int r = orb_extract_batch_device(h, h->d_in, n_frames, w, hgt, pitch, (int64_t)fbytes, h->d_kps, h->d_desc,
                                         h->out_cap, h->d_counts, h->stream);
        if (r) return r;
        HIPCHK(hipMemcpyAsync(h->h_counts, h->d_counts, (size_t)n_frames * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(h->h_kps, h->d_kps, (size_t)n_frames * h->out_cap * sizeof(OrbKeyPoint), hipMemcpyDeviceToHost,
                              h->stream));
        HIPCHK(hipMemcpyAsync(h->h_desc, h->d_desc, (size_t)n_frames * h->out_cap * 32, hipMemcpyDeviceToHost, h->stream));
        return ORB_OK;
    };
    /* graphs pay off where launch overhead dominates (a few frames per call); big batches stay eager */
    const bool want_graph = h->use_graph && !h->profile && n_frames <= 8;
    const bool same = h->graph_exec && h->graph_gen == h->generation && h->graph_w == w && h->graph_h == hgt &&
                      h->graph_n == n_frames;
    if (want_graph && same) {
        HIPCHK(hipGraphLaunch(h->graph_exec, h->stream)); /* replay */
        h->last_frames = n_frames;
    } else if (want_graph && h->graph_seen && h->graph_gen == h->generation && h->graph_w == w &&
               h->graph_h == hgt && h->graph_n == n_frames) {
        /* second call with this shape (the first ran eagerly and did every one-time setup): capture */
        if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
        if (h->graph) (void)hipGraphDestroy(h->graph);
        h->graph_exec = nullptr;
        h->graph = nullptr;
        /* RELAXED capture mode: only this handle's (non-blocking) stream takes part in the capture, and its dependencies
           are explicit events.  In thread-local mode ROCm 7.2 still fails calls of OTHER host threads while this one
           captures (a second extractor's hipMemcpyAsync came back with "would make the legacy stream depend on a
           capturing blocking stream", about once in ten runs of three extractors starting up on three threads). */
        HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed));
        rc = enqueue();
        hipError_t ce = hipStreamEndCapture(h->stream, &h->graph);
        if (rc || ce != hipSuccess) (void)hipGetLastError(); /* (the eager fallback below starts from a clean error state) */
        if (rc || ce != hipSuccess || hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0) != hipSuccess) {
            /* capture not possible here: fall back to eager launches for good (still the same kernels) */
            h->use_graph = false;
            if (h->graph) (void)hipGraphDestroy(h->graph);
            h->graph = nullptr;
            h->graph_exec = nullptr;
            if ((rc = enqueue())) return rc;
        } else {
            HIPCHK(hipGraphLaunch(h->graph_exec, h->stream));
            h->last_frames = n_frames;
        }
    } else {
        if ((rc = enqueue())) return rc;
        h->graph_seen = 1;
        h->graph_gen = h->generation;
        h->graph_w = w;
        h->graph_h = hgt;
        h->graph_n = n_frames;
        if (h->graph_exec) { /* shape changed: drop the stale graph */
            (void)hipGraphExecDestroy(h->graph_exec);
            (void)hipGraphDestroy(h->graph);
            h->graph_exec = nullptr;
            h->graph = nullptr;
        }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    int worst = ORB_OK;
    for (int f = 0; f < n_frames; f++) {
        int n = h->h_counts[f];
        n_out[f] = n;
        if (n > cap) {
            worst = ORB_E_CAPACITY;
            g_err = "orb_extract_batch: output capacity too small";
            continue;
        }
        memcpy(kps_out + (size_t)f * cap, h->h_kps + (size_t)f * h->out_cap, (size_t)n * sizeof(OrbKeyPoint));
        memcpy(desc_out + (size_t)f * cap * 32, h->h_desc + (size_t)f * h->out_cap * 32, (size_t)n * 32);
    }
    return worst;
}

extern "C" int orb_extract(OrbExtractor *h, const uint8_t *gray, int w, int hgt, int stride, OrbKeyPoint *kps_out,
                           uint8_t *desc_out, int cap, int32_t *n_out) {
    if (!h || !n_out) {
        g_err = "orb_extract: bad argument";
        return ORB_E_INVALID;
    }
    *n_out = 0;
    if (!gray || w <= 0 || hgt <= 0) return ORB_OK;
    const uint8_t *imgs[1] = {gray};
    return orb_extract_batch(h, imgs, 1, w, hgt, stride, kps_out, desc_out, cap, n_out);
}

static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

extern "C" int orb_download_level(OrbExtractor *h, int frame, int level, int blurred, uint8_t *out) {
    if (!h || !out || !h->geom_valid || frame < 0 || frame >= h->last_frames || level < 0 || level >= h->cfg.nlevels) {
        g_err = "orb_download_level: bad argument";
        return ORB_E_INVALID;
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const LevelGeom &L = h->geom.lv[level];
    std::vector<uint8_t> tight((size_t)L.w * L.h);
    const uint8_t *src = level_ptr(blurred ? h->buf.blur : h->buf.pyr, h->geom, frame, level);
    HIPCHK(hipMemcpy2D(tight.data(), L.w, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
    if (blurred) {
        memcpy(out, tight.data(), tight.size());
        return ORB_OK;
    }
    /* copyMakeBorder(.., 19,19,19,19, BORDER_REFLECT_101) -- src/ORBextractor.cpp:1086,:1091 */
    const int B = ORB_EDGE_THRESHOLD, W = L.w + 2 * B;
    for (int y = -B; y < L.h + B; y++) {
        const uint8_t *row = tight.data() + (size_t)reflect101(y, L.h) * L.w;
        uint8_t *o = out + (size_t)(y + B) * W;
        for (int x = -B; x < L.w + B; x++) o[x + B] = row[reflect101(x, L.w)];
    }
    return ORB_OK;
}

extern "C" int orb_download_candidates(OrbExtractor *h, int frame, int level, OrbKeyPoint *out, int cap, int32_t *n_out) {
    if (!h || !n_out || !h->geom_valid || frame < 0 || frame >= h->last_frames || level < 0 || level >= h->cfg.nlevels) {
        g_err = "orb_download_candidates: bad argument";
        return ORB_E_INVALID;
    }
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const OrbGeom &g = h->geom;
    const LevelGeom &L = g.lv[level];
    std::vector<int32_t> cnt(std::max(L.n_cells, 1));
    if (L.n_cells)
    {
        HIPCHK(hipMemcpyAsync(cnt.data(), h->buf.cell_cnt + (size_t)frame * g.total_cells + L.cell_base, (size_t)L.n_cells * 4,
                              hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    int n = 0;
    std::vector<uint32_t> keys;
    for (int c = 0; c < L.n_cells; c++) {
        const CellDesc &cd = h->h_cells[L.cell_base + c];
        if (cnt[c] < 0 || cnt[c] > cd.slab_cap) {
            g_err = "orb_download_candidates: corrupt cell count";
            return ORB_E_HIP;
        }
        if (cnt[c] == 0) continue;
        keys.resize(cnt[c]);
        HIPCHK(hipMemcpyAsync(keys.data(), h->buf.slab + (size_t)frame * g.slab_frame_count + cd.slab_off, (size_t)cnt[c] * 4,
                              hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int k = 0; k < cnt[c]; k++, n++) {
            if (out && n < cap) {
                OrbKeyPoint kp;
                kp.x = (float)KEY_X(keys[k]);
                kp.y = (float)KEY_Y(keys[k]);
                kp.size = 7.f;
                kp.angle = -1.f;
                kp.response = (float)KEY_R(keys[k]);
                kp.octave = 0;
                kp.class_id = -1;
                out[n] = kp;
            }
        }
    }
    *n_out = n;
    return (out && n > cap) ? ORB_E_CAPACITY : ORB_OK;
}
