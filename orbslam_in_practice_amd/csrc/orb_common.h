/* orb_common.h -- shared host/device structures of the gfx950 ORB front-end. */
#ifndef ORB_COMMON_H
#define ORB_COMMON_H

#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <stdint.h>

#include "../../include/orbslam_amd.h"

#define ORB_MAX_LEVELS 12
#define ORB_EDGE_THRESHOLD 19
#define ORB_MIN_BORDER 16      /* EDGE_THRESHOLD - 3, reference src/ORBextractor.cpp:729 */
#define ORB_CELL_MAX 66        /* FAST cell ROI is at most (59+6) px on a side            */
#define ORB_QT_MAX_NODES 2048  /* quadtree list capacity per (frame, level): k_quadtree<2048>  */
#define ORB_QT_THREADS 512

/* packed FAST candidate: window-relative x (12 bit) | y (12 bit) << 12 | score << 24 */
#define KEY_X(k) ((int)((k) & 0xFFFu))
#define KEY_Y(k) ((int)(((k) >> 12) & 0xFFFu))
#define KEY_R(k) ((int)((k) >> 24))
#define KEY_PACK(x, y, r) (((uint32_t)(x)) | ((uint32_t)(y) << 12) | ((uint32_t)(r) << 24))

struct LevelGeom {
    int w, h, pitch;          /* level size; bytes per row in the pyramid buffers                  */
    int64_t plane_off;        /* byte offset of this level inside one frame's pyramid block        */
    int maxBX, maxBY;         /* maxBorderX/Y (src/ORBextractor.cpp:731-732); minBorder = 16       */
    int nCols, nRows, wCell, hCell; /* FAST cell grid (:740-743); nCols==0 -> level skipped         */
    int cell_base;            /* first cell id of this level                                       */
    int n_cells;              /* cells actually scanned on this level (row-major)                  */
    int quota;                /* mnFeaturesPerLevel[level]                                         */
    int out_cap;              /* quadtree output capacity                                          */
    int out_base;             /* first output slot of this level in the per-frame level-kp array   */
    int key_cap;              /* capacity of the per-level candidate arrays (NMS bound)            */
    int64_t key_off;          /* offset (in keys) of this level inside one frame's key block       */
    int nIni;                 /* quadtree root nodes (:493)                                        */
    float hX;                 /* root node width (:495)                                            */
    float scale;              /* mvScaleFactor[level]                                              */
    float patch_size;         /* (int)(31*scale) as float (:793)                                   */
    /* resize tables (level l from level l-1): offsets into the coefficient table buffer */
    int xofs_off, yofs_off;   /* int index tables                                                  */
    int rs_lw, rs_lh;         /* resize: LDS source window (bytes per row, rows) of the largest 128x32 tile; 0 = gather path */
};

struct OrbGeom {
    int nlevels, w, h;
    int ini_th, min_th;
    int total_cells;
    int cell_max_rw, cell_max_rh; /* largest FAST cell ROI of this geometry (sizes the LDS tile) */
    int total_out_cap;        /* sum of out_cap                                                     */
    int64_t pyr_frame_bytes;  /* bytes of one frame's pyramid block (all levels)                   */
    int64_t key_frame_count;  /* keys of one frame's candidate block (all levels)                  */
    int64_t slab_frame_count; /* keys of one frame's cell-slab block                               */
    LevelGeom lv[ORB_MAX_LEVELS];
};

/* one FAST cell (reference src/ORBextractor.cpp:745-786) */
struct CellDesc {
    short level;
    short rw, rh;     /* ROI size (incl. the 3-px rim FAST ignores)       */
    short iniX, iniY; /* ROI origin in level coordinates                   */
    short offX, offY; /* j*wCell, i*hCell: added to ROI-relative kp coords */
    short pad;
    int slab_off;     /* first slot of this cell in the frame's slab block */
    int slab_cap;
};

struct OrbBuffers {
    uint8_t *pyr;        /* [frame][pyr_frame_bytes]           un-blurred levels            */
    uint8_t *blur;       /* same layout, blurred levels                                       */
    uint32_t *slab;      /* [frame][slab_frame_count]          per-cell FAST output (packed) */
    int32_t *cell_cnt;   /* [frame][total_cells]                                              */
    uint32_t *keysA;     /* [frame][key_frame_count]           quadtree key buffers          */
    uint32_t *keysB;
    int32_t *cand_cnt;   /* [frame][nlevels]  candidates per level                            */
    uint32_t *lvl_kp;    /* [frame][total_out_cap]  packed keys kept by the quadtree          */
    int32_t *lvl_cnt;    /* [frame][nlevels]                                                  */
    const CellDesc *cells;
    const int32_t *tab_i;   /* xofs / yofs tables      */
    const int16_t *tab_c;   /* ialpha / ibeta tables (2 per entry, same indexing *2) */
};

__host__ __device__ inline const uint8_t *level_ptr(const uint8_t *base, const OrbGeom &g, int frame, int level) {
    return base + (int64_t)frame * g.pyr_frame_bytes + g.lv[level].plane_off;
}
__host__ __device__ inline uint8_t *level_ptr(uint8_t *base, const OrbGeom &g, int frame, int level) {
    return base + (int64_t)frame * g.pyr_frame_bytes + g.lv[level].plane_off;
}

/* XCD-aware work-item order (MI355X: 8 XCDs, each with its own L2; workgroups are dealt round-robin, so ids
 * b and b+8 share an L2).  Maps the hardware's linear workgroup id to a work-item index such that
 * CONSECUTIVE work items (x-adjacent cells / tiles, which share 128-B lines) land on the SAME XCD at about
 * the same time.  Bijective for any nwg (guide: cdna_hip_programming.md T1).  Speed only, never correctness. */
__device__ __forceinline__ int xcd_remap(int lin, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7, slot = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

/* launchers (one per kernel family); all enqueue on `st` and return hipGetLastError() */
hipError_t launch_pyramid(const OrbGeom &g, const OrbBuffers &b, const uint8_t *d_imgs, int stride,
                          int64_t frame_stride, int n_frames, hipStream_t st);
hipError_t launch_blur(const OrbGeom &g, const OrbBuffers &b, int n_frames, hipStream_t st);
hipError_t launch_fast(const OrbGeom &g, const OrbBuffers &b, int n_frames, hipStream_t st);
hipError_t launch_quadtree(const OrbGeom &g, const OrbBuffers &b, int n_frames, hipStream_t st);
hipError_t orb_upload_constants(int device, hipStream_t stream);
hipError_t launch_describe(const OrbGeom &g, const OrbBuffers &b, int n_frames, OrbKeyPoint *d_kps,
                           uint8_t *d_desc, int cap, int32_t *d_counts, hipStream_t st);


/* hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember the largest size granted on each.  The
 * check and the attribute call sit under one mutex, so a second host thread can never see the size as "already
 * raised" and launch before the first thread's hipFuncSetAttribute has taken effect. */
struct PerDeviceLdsLimit {
    size_t set[64] = {};
    std::mutex mu;
    hipError_t ensure(const void *func, size_t bytes) {
        int d = 0;
        hipError_t e = hipGetDevice(&d);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(mu);
        if (d >= 0 && d < 64 && bytes <= set[d]) return hipSuccess;
        e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess && d >= 0 && d < 64) set[d] = bytes;
        return e;
    }
};


/* one error text per host thread for the whole library (orb_last_error()): the matcher and BA translation units
 * forward theirs through this slot */
void orb_set_last_error(const char *msg);
struct OrbErrSlot {
    OrbErrSlot &operator=(const std::string &s) {
        orb_set_last_error(s.c_str());
        return *this;
    }
    OrbErrSlot &operator=(const char *s) {
        orb_set_last_error(s);
        return *this;
    }
};

#endif
