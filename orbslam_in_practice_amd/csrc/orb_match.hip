/*
 * orb_match.hip -- ORBmatcher on gfx950: all-pairs Hamming distances as i8 MFMA products (xor / popcount
 * scans as the fallback), wavefront min-reductions, speculative rounds for the sequential walks.
 *
 *   k_hamming_knn2   brute-force 2-NN (north_star / SURVEY a12): one query descriptor per lane in
 *                    8 VGPRs, the train set streamed through LDS in 1024-descriptor tiles and read
 *                    back as wave-wide broadcasts (conflict free), ties -> lowest index.
 *   k_hamming_knn2_mfma  the same result from an i8 MFMA product (see k_match_knn_mfma): the default.
 *   k_match_knn      phase 1 of ORBmatcher::SearchForInitialization (reference src/ORBmatcher.cpp:9-126
 *                    with Frame::GetFeaturesInArea, src/Frame.cpp:219-271): fully parallel, the 8
 *                    smallest (distance, grid order, index) keys of every query over all its window
 *                    candidates, ignoring the greedy filter.
 *   k_match_knn_mfma phase 1 on the matrix cores: all-pairs Hamming distance as a matrix product of
 *                    0/1 and +-1 operand bytes (v_mfma_i32_32x32x32_i8), branch-free scan of the
 *                    results against the largest distance that can matter; the default when the
 *                    32-bit key fits (k_match_knn otherwise).
 *   k_match_init     phase 2: ONE wavefront per frame pair.  The reference loop over i1 carries a
 *                    dependency (a later i1 may only take an i2 whose recorded match distance is
 *                    larger, :49-50); the filtered candidate set is a subset of the unfiltered one, so
 *                    the answer of a query is read off its phase-1 list, and 64 queries at a time are
 *                    resolved in speculative rounds that commit exactly the prefix the sequential walk
 *                    would have produced (see the kernel); a query whose list is exhausted is rescanned
 *                    by the whole wave with a (distance, grid order) min-reduction.
 *
 * Candidate set and order of GetFeaturesInArea without materialising the 64x48 grid: an F2 keypoint is
 * a candidate iff it was assigned to the grid (round((p-min)*inv) inside the grid, src/Frame.cpp:
 * 161-168 with the miMaxY typo corrected), passes the octave filter and lies strictly inside the
 * square window; the cell range [floor((x-r)*inv), ceil((x+r)*inv)] can never exclude such a point.
 * The reference visits candidates by (cell x, cell y, insertion order); only the FIRST minimum
 * depends on that order, so each candidate carries the key (dist << 32 | cell << 16 | index).
 */
#include "orb_common.h"
#include <type_traits>

#include <algorithm>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#define GRID_COLS 64
#define GRID_ROWS 48
#define HISTO_LENGTH 30
#define TH_LOW 50

__device__ __forceinline__ int hamming256(const uint32_t a[8], const uint4 b0, const uint4 b1) {
    return __popc(a[0] ^ b0.x) + __popc(a[1] ^ b0.y) + __popc(a[2] ^ b0.z) + __popc(a[3] ^ b0.w) +
           __popc(a[4] ^ b1.x) + __popc(a[5] ^ b1.y) + __popc(a[6] ^ b1.z) + __popc(a[7] ^ b1.w);
}

#define KNN_TILE 1024
__global__ __launch_bounds__(256) void k_hamming_knn2(const uint8_t *d1, int n1, const uint8_t *d2, int n2,
                                                       int32_t *idx, uint16_t *best, uint16_t *second) {
    __shared__ uint4 s_d2[KNN_TILE * 2];
    const int q = blockIdx.x * 256 + threadIdx.x;
    uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (q < n1) {
        const uint4 *p = (const uint4 *)(d1 + (size_t)q * 32);
        uint4 v0 = p[0], v1 = p[1];
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
        a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
    }
    int b1 = 0x7FFFFFFF, b2 = 0x7FFFFFFF, bi = -1;
    for (int t0 = 0; t0 < n2; t0 += KNN_TILE) {
        const int nt = min(KNN_TILE, n2 - t0);
        __syncthreads();
        const uint4 *src = (const uint4 *)(d2 + (size_t)t0 * 32);
        for (int i = threadIdx.x; i < nt * 2; i += 256) s_d2[i] = src[i]; /* 16 B per lane, coalesced */
        __syncthreads();
        for (int j = 0; j < nt; j++) {
            const int d = hamming256(a, s_d2[2 * j], s_d2[2 * j + 1]);
            if (d < b1) {
                b2 = b1;
                b1 = d;
                bi = t0 + j;
            } else if (d < b2)
                b2 = d;
        }
    }
    if (q < n1) {
        idx[q] = bi;
        best[q] = (uint16_t)(b1 == 0x7FFFFFFF ? 0xFFFF : b1);
        second[q] = (uint16_t)(b2 == 0x7FFFFFFF ? 0xFFFF : b2);
    }
}

struct MatchArgs {
    const OrbKeyPoint *kps;  /* frame slabs: kps[frame*cap + i]           */
    const uint8_t *desc;     /* desc[(frame*cap + i)*32]                  */
    const int32_t *counts;   /* counts[frame]                             */
    int cap;
    const int32_t *pair_a, *pair_b; /* NULL: pair p = (2p, 2p+1)           */
    const float *prev_xy;    /* NULL: prev matched = F1 keypoint positions; else [pair][cap][2] */
    float *prev_out;         /* optional updated prev matched              */
    float min_x, max_x, min_y, max_y;
    int window;              /* < 0: brute force variant                   */
    float nnratio;
    int check_ori;
    int32_t *matches12;      /* [pair][cap]                                */
    int32_t *nmatches;       /* [pair]                                     */
    uint8_t *bin_scratch;    /* [pair][cap] rotation bin of each i1        */
    int desc_in_lds;         /* F2 descriptors staged in LDS for the (rare) rescans */
    ulonglong2 *knn;         /* [pair][cap][MK_TOP/2] phase-1 result: the smallest keys (dist<<32 | ord), ~0 = none */
};

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint64_t t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ uint64_t rdlane64(uint64_t v, int lane_uniform) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane_uniform);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane_uniform);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int t = __shfl_xor(v, o);
        v = t < v ? t : v;
    }
    return v;
}


/* candidate order key of one F2 keypoint (see file header); 0xFFFFFFFF = never a candidate */
__device__ __forceinline__ uint32_t f2_order_key(const OrbKeyPoint &kp, int j, bool brute, float min_x, float min_y,
                                                 float invW, float invH) {
    if (brute) return (uint32_t)j;
    if (kp.octave != 0) return 0xFFFFFFFFu; /* GetFeaturesInArea(.., level1, level1) with level1 == 0 */
    const int ix = (int)round(((double)kp.x - (double)min_x) * (double)invW);
    const int iy = (int)round(((double)kp.y - (double)min_y) * (double)invH);
    if (ix >= 0 && ix < GRID_COLS && iy >= 0 && iy < GRID_ROWS) return ((uint32_t)(ix * GRID_ROWS + iy) << 16) | (uint32_t)j;
    return 0xFFFFFFFFu;
}

/* PHASE 1 (fully parallel): for every query i1 the MK_TOP smallest (dist, order) keys over ALL its window
 * candidates, ignoring the greedy vMatchedDistance filter.  One query per lane; F2 is streamed through LDS in
 * 512-keypoint tiles from which only the keypoints that can be a candidate of ANY query are kept (windowed
 * variant: octave 0 and inside the grid -- about a fifth of a frame), compacted with an LDS counter: the
 * result is a set of unique keys, so the staging order is irrelevant.  Every LDS read in the scan is a
 * wave-wide broadcast.  Workgroups without a single active query (all octaves > 0) leave before staging.
 * KEY32: the (dist, cell, index) key packed into 32 bits (9 + 12 + 11, or dist << 16 | j for the brute-force
 * variant) so that the sorted insert works on single registers; it is widened to the 64-bit format of phase 2
 * on the way out.  Used whenever the slab capacity allows (cap <= 2048, resp. <= 65536). */
#define MK_TILE 512
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#define MK_TOP 8 /* keys kept per query; phase 2 rescans only if fewer than two of them are still allowed */
template <bool KEY32>
__global__ __launch_bounds__(256) void k_match_knn(MatchArgs A) {
    __shared__ uint4 s_d[MK_TILE * 2];
    __shared__ float s_x[MK_TILE], s_y[MK_TILE];
    __shared__ uint32_t s_o[MK_TILE];
    __shared__ int s_nc;
    typedef typename std::conditional<KEY32, uint32_t, uint64_t>::type key_t;
    const key_t KNONE = (key_t)~(key_t)0;
    const int pair = blockIdx.y;
    const int fa = A.pair_a ? A.pair_a[pair] : 2 * pair, fb = A.pair_b ? A.pair_b[pair] : 2 * pair + 1;
    const int n1 = A.counts[fa], n2 = A.counts[fb];
    if ((int)(blockIdx.x * 256) >= n1) return; /* whole block beyond the query count (uniform) */
    const OrbKeyPoint *k1 = A.kps + (size_t)fa * A.cap, *k2 = A.kps + (size_t)fb * A.cap;
    const uint8_t *D1 = A.desc + (size_t)fa * A.cap * 32, *D2 = A.desc + (size_t)fb * A.cap * 32;
    const bool brute = A.window < 0;
    const float invW = (float)GRID_COLS / (A.max_x - A.min_x), invH = (float)GRID_ROWS / (A.max_y - A.min_y);
    const float r = (float)A.window;
    const int i1 = blockIdx.x * 256 + threadIdx.x;
    uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float px = 0, py = 0;
    bool active = false;
    if (i1 < n1) {
        const OrbKeyPoint kp1 = k1[i1];
        active = brute || kp1.octave == 0; /* :26-28 */
        px = kp1.x;
        py = kp1.y;
        if (A.prev_xy) {
            px = A.prev_xy[((size_t)pair * A.cap + i1) * 2];
            py = A.prev_xy[((size_t)pair * A.cap + i1) * 2 + 1];
        }
        const uint4 *p = (const uint4 *)(D1 + (size_t)i1 * 32);
        uint4 v0 = p[0], v1 = p[1];
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
        a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
    }
    key_t bk[MK_TOP]; /* the MK_TOP smallest keys, ascending */
#pragma unroll
    for (int k = 0; k < MK_TOP; k++) bk[k] = KNONE;
    const bool any_active = __syncthreads_or(active) != 0;
    for (int t0 = 0; any_active && t0 < n2; t0 += MK_TILE) {
        const int nt = min(MK_TILE, n2 - t0);
        __syncthreads();
        if (threadIdx.x == 0) s_nc = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += 256) {
            /* keypoint and descriptor are fetched together, before the candidate test: one memory round trip per
               step instead of two dependent ones */
            const OrbKeyPoint kp = k2[t0 + i];
            const uint4 *src = (const uint4 *)(D2 + (size_t)(t0 + i) * 32);
            const uint4 dlo = src[0], dhi = src[1];
            const uint32_t ord = f2_order_key(kp, t0 + i, brute, A.min_x, A.min_y, invW, invH);
            if (ord != 0xFFFFFFFFu) {
                const int slot = atomicAdd(&s_nc, 1);
                s_d[2 * slot] = dlo;
                s_d[2 * slot + 1] = dhi;
                s_x[slot] = kp.x;
                s_y[slot] = kp.y;
                /* 32-bit form: cell << 11 | index (windowed), index (brute force) */
                s_o[slot] = (KEY32 && !brute) ? ((ord >> 16) << 11 | (ord & 0x7FFu)) : ord;
            }
        }
        __syncthreads();
        const int nc = s_nc;
        if (active) {
            for (int j = 0; j < nc; j++) {
                const uint32_t ord = s_o[j];
                bool cand = true;
                if (!brute) cand = fabsf(s_x[j] - px) < r && fabsf(s_y[j] - py) < r;
                const int d = hamming256(a, s_d[2 * j], s_d[2 * j + 1]);
                /* A candidate farther than TH_LOW can only ever act as the SECOND best, and it changes the ratio test
                   (:65-67: best <= TH_LOW && best < second * ratio) only if TH_LOW >= d * ratio.  Beyond that it cannot
                   influence any decision of phase 2 -- with it or without it the query is accepted, or has no candidate
                   within TH_LOW -- so it is not kept: the list then ends in KNONE, which phase 2 reads as "no further
                   candidate that matters".  Random descriptors sit ~128 bits apart: nearly every step of the scan skips
                   the sorted insert wave-wide (a third of this loop's instructions). */
                cand = cand && !(d > TH_LOW && (float)TH_LOW < (float)d * A.nnratio);
                if (!cand) continue;
                key_t key;
                if (KEY32)
                    key = ((uint32_t)d << (brute ? 16 : 23)) | ord;
                else
                    key = (key_t)(((uint64_t)(uint32_t)d << 32) | ord);
                if (KEY32) {
                    /* branch-free sorted insert: with a[k-1] <= a[k], the new a[k] is median(a[k-1], key, a[k])
                       (one v_med3_u32 each) -- a no-op for key = KNONE */
#pragma unroll
                    for (int k = MK_TOP - 1; k > 0; k--)
                        bk[k] = (key_t)umed3((uint32_t)bk[k - 1], (uint32_t)key, (uint32_t)bk[k]);
                    bk[0] = key < bk[0] ? key : bk[0];
                } else if (key < bk[MK_TOP - 1]) { /* rare once the list has warmed up (keys are unique) */
#pragma unroll
                    for (int k = MK_TOP - 1; k > 0; k--) bk[k] = key < bk[k - 1] ? bk[k - 1] : (key < bk[k] ? key : bk[k]);
                    bk[0] = key < bk[0] ? key : bk[0];
                }
            }
        }
    }
    if (i1 < n1) {
        uint64_t out[MK_TOP];
#pragma unroll
        for (int k = 0; k < MK_TOP; k++) {
            if (!KEY32) {
                out[k] = (uint64_t)bk[k];
            } else if (bk[k] == KNONE) {
                out[k] = ~0ull;
            } else if (brute) {
                out[k] = ((uint64_t)((uint32_t)bk[k] >> 16) << 32) | ((uint32_t)bk[k] & 0xFFFFu);
            } else {
                const uint32_t k32 = (uint32_t)bk[k];
                out[k] = ((uint64_t)(k32 >> 23) << 32) | (uint64_t)((((k32 >> 11) & 0xFFFu) << 16) | (k32 & 0x7FFu));
            }
        }
#pragma unroll
        for (int k = 0; k < MK_TOP / 2; k++)
            A.knn[((size_t)pair * A.cap + i1) * (MK_TOP / 2) + k] = make_ulonglong2(out[2 * k], out[2 * k + 1]);
    }
}

/* PHASE 1 on the matrix cores (the KEY32 case).  All-pairs Hamming distance IS a matrix product: with the F2
 * descriptor bits t_k in {0, 1} and the query bits as q'_k = 1 - 2 q_k in {+1, -1},
 *       sum_k t_k q'_k = popc(t) - 2 popc(t & q)   ==>   d(t, q) = popc(t ^ q) = popc(q) + sum_k t_k q'_k
 * so one v_mfma_i32_32x32x32_i8 chain over K = 256 gives 32 x 32 exact distances (minus a per-query constant) for
 * 8 x 2 operand registers, against 16 VALU instructions PER PAIR for xor + popcount -- the scan this replaces was a
 * ninth of the front-end's vector instructions.  Layout: A = 32 F2 keypoints (rows), B = 32 queries (columns), so a
 * lane's 16 results all belong to ONE query (column lane & 31) and the per-query key list stays in its registers;
 * a wavefront carries two query blocks (64 queries, their operands expanded once: 64 VGPRs), the workgroup's four
 * wavefronts share the F2 operands, which are expanded from bits to bytes once per 128 keypoints into LDS in
 * fragment order (conflict-free ds_read_b128).  The pairing of bits with operand bytes is free as long as A and B
 * agree (the product sums over k): K-step s = descriptor word s, lane half h and byte j <-> bit 8 (j & 3) + 4 h +
 * (j >> 2), i.e. dword i of a fragment = (W >> (4 h + i)) & 0x01010101 -- two instructions per four operand bytes.
 * Selection: a distance matters only up to Dmax (TH_LOW, or what the ratio test can still see: k_match_knn), so
 * a result is compared with the lane's threshold Dmax - popc(q) -- one v_cmp and one untaken scalar branch per
 * 64 pairs -- and only a hit (a handful per query over the whole frame) runs the candidate tests and the sorted
 * insert.  The two lanes that share a query (rows 4 h ..) merge their lists at the end.  Same keys, same lists as
 * k_match_knn<true>: the kept set is "the MK_TOP smallest of a set of unique keys", whatever the visiting order. */
typedef int mm_v4i __attribute__((ext_vector_type(4)));
typedef int mm_v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ mm_v4i mm_bits01(uint32_t W, int h) {
    const uint32_t w = W >> (4 * h);
    mm_v4i r;
    r.x = (int)(w & 0x01010101u);
    r.y = (int)((w >> 1) & 0x01010101u);
    r.z = (int)((w >> 2) & 0x01010101u);
    r.w = (int)((w >> 3) & 0x01010101u);
    return r;
}
__device__ __forceinline__ int mm_pm1(uint32_t x) { return (int)((((x << 8) - x)) | 0x01010101u); } /* byte 1 -> 0xFF (-1), byte 0 -> +1 */
__device__ __forceinline__ mm_v4i mm_bits_pm(uint32_t W, int h) {
    const mm_v4i b = mm_bits01(W, h);
    mm_v4i r;
    r.x = mm_pm1((uint32_t)b.x);
    r.y = mm_pm1((uint32_t)b.y);
    r.z = mm_pm1((uint32_t)b.z);
    r.w = mm_pm1((uint32_t)b.w);
    return r;
}
/* The 16 results a lane holds of one 32 x 32 tile against the lane's threshold.  Looking at a result must cost next
   to nothing -- there are a million per frame pair and a thousand that matter -- and must not branch: a skipped block
   is a TAKEN branch, and 32 of those per tile cost more than the products (first version of this kernel: 181 us per
   256 pairs against 27 us of MFMA time).  So: one subtract and one v_alignbit per result shift the sign of
   (result - threshold - 1) into a per-lane bit mask, one wave-wide test per block and tile; the lanes with hits then
   walk their mask bits, RECOMPUTE the distance of that (query, keypoint) pair from the raw descriptor words (query:
   registers, keypoint: LDS; a register cannot be indexed by a lane-varying bit number) and run the candidate tests of
   k_match_knn and the sorted insert.  One trip of that loop per tile in the common case. */
__device__ __forceinline__ void mm_scan_tile(const mm_v16i acc, const int thr1, const uint32_t (&qw)[8], const float px, const float py,
                                             uint32_t (&bk)[MK_TOP], const int tile, const int h, const bool brute, const float r,
                                             const float nnratio, const uint32_t *s_o, const float *s_x, const float *s_y,
                                             const uint4 *s_raw) {
    const uint32_t KNONE = 0xFFFFFFFFu;
    uint32_t mask = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) mask = __builtin_amdgcn_alignbit(mask, (uint32_t)(acc[i] - thr1), 31); /* bit 15 - i: result i <= threshold */
    while (__builtin_amdgcn_ballot_w64(mask != 0u) != 0ull) {
        /* straight-line for every lane (a lane without a hit looks at row 0 and drops the key): no divergent blocks */
        const bool has = mask != 0u;
        const int i = has ? 16 - __ffs((int)mask) : 0;
        mask &= mask - 1u;
        const int slot = tile * 32 + (i & 3) + 8 * (i >> 2) + 4 * h; /* C/D row of register i */
        const uint4 t0 = s_raw[2 * slot], t1 = s_raw[2 * slot + 1];
        const uint32_t ord = s_o[slot];
        const float dx = fabsf(s_x[slot] - px), dy = fabsf(s_y[slot] - py);
        const int d = hamming256(qw, t0, t1);
        const int keep = (int)has & (int)(ord != KNONE) & ((int)brute | ((int)(dx < r) & (int)(dy < r))) &
                         ((int)(d <= TH_LOW) | (int)!((float)TH_LOW < (float)d * nnratio)); /* the tests of k_match_knn */
        const uint32_t key = keep ? (((uint32_t)d << (brute ? 16 : 23)) | ord) : KNONE;
#pragma unroll
        for (int k = MK_TOP - 1; k > 0; k--) bk[k] = umed3(bk[k - 1], key, bk[k]);
        bk[0] = key < bk[0] ? key : bk[0];
    }
}
__device__ __forceinline__ void mm_merge_halves(uint32_t (&bk)[MK_TOP]) {
    uint32_t other[MK_TOP];
#pragma unroll
    for (int k = 0; k < MK_TOP; k++) other[k] = (uint32_t)__shfl_xor((int)bk[k], 32);
#pragma unroll
    for (int j = 0; j < MK_TOP; j++) {
        const uint32_t key = other[j];
#pragma unroll
        for (int k = MK_TOP - 1; k > 0; k--) bk[k] = umed3(bk[k - 1], key, bk[k]);
        bk[0] = key < bk[0] ? key : bk[0];
    }
}
/* QB: query blocks of 32 per wavefront (2 = fewer operand reads per pair, 1 = half the registers: 4 waves per SIMD);
   CH: F2 keypoints per LDS stage (8 KB of operand bytes per 32) -- the workgroups of this kernel sit beside the VALU-bound
   kernels of the other lane for a long time, and what they hold in LDS those cannot use */
template <int QB, int CH>
__global__ __launch_bounds__(256) void k_match_knn_mfma(MatchArgs A) {
    __shared__ mm_v4i s_a[(CH / 32) * 8 * 64];
    __shared__ uint4 s_raw[CH * 2]; /* the same keypoints' descriptors as they are: a hit recomputes its distance from these */
    __shared__ float s_x[CH], s_y[CH];
    __shared__ uint32_t s_o[CH];
    const uint32_t KNONE = 0xFFFFFFFFu;
    /* the query blocks of one pair read the same F2 set: they run on ONE XCD (consecutive remapped items share an L2) */
    const int item = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int pair = item / (int)gridDim.x, qblk = item - pair * (int)gridDim.x;
    const int fa = A.pair_a ? A.pair_a[pair] : 2 * pair, fb = A.pair_b ? A.pair_b[pair] : 2 * pair + 1;
    const int n1 = A.counts[fa], n2 = A.counts[fb];
    if (qblk * 128 * QB >= n1) return; /* whole block beyond the query count (uniform) */
    const OrbKeyPoint *k1 = A.kps + (size_t)fa * A.cap, *k2 = A.kps + (size_t)fb * A.cap;
    const uint8_t *D1 = A.desc + (size_t)fa * A.cap * 32, *D2 = A.desc + (size_t)fb * A.cap * 32;
    const bool brute = A.window < 0;
    const float invW = (float)GRID_COLS / (A.max_x - A.min_x), invH = (float)GRID_ROWS / (A.max_y - A.min_y);
    const float r = (float)A.window;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int qbase = qblk * 128 * QB + (threadIdx.x >> 6) * 32 * QB;
    /* the largest distance that can matter: d <= TH_LOW, or TH_LOW >= d * ratio (the exact test is repeated on a hit) */
    const float dq = A.nnratio > 0.0f ? (float)TH_LOW / A.nnratio : 1e9f;
    const int Dmax = dq >= 256.0f ? 256 : max(TH_LOW, (int)dq + 1);
    mm_v4i bq[2][8];
    uint32_t qw0[8], qw1[8]; /* the two queries' descriptor words */
    int thr1[2] = {0, 0};    /* threshold + 1 on the product: d <= Dmax <=> result - (Dmax - popc(q)) - 1 < 0 */
    float px[2] = {0.0f, 0.0f}, py[2] = {0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < 8; s++) {
        qw1[s] = 0u;
        bq[1][s] = mm_v4i{0, 0, 0, 0};
    }
    bool act_any = false;
#pragma unroll
    for (int b = 0; b < QB; b++) {
        const int qi = qbase + 32 * b + c;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool act = false;
        px[b] = py[b] = 0.0f;
        if (qi < n1) {
            const OrbKeyPoint kp1 = k1[qi];
            act = brute || kp1.octave == 0; /* :26-28 */
            px[b] = kp1.x;
            py[b] = kp1.y;
            if (A.prev_xy) {
                px[b] = A.prev_xy[((size_t)pair * A.cap + qi) * 2];
                py[b] = A.prev_xy[((size_t)pair * A.cap + qi) * 2 + 1];
            }
            const uint4 *p = (const uint4 *)(D1 + (size_t)qi * 32);
            const uint4 v0 = p[0], v1 = p[1];
            w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
            w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
        }
        int pc = 0;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            pc += __popc(w[s]);
            bq[b][s] = mm_bits_pm(w[s], h);
            if (b == 0) qw0[s] = w[s]; else qw1[s] = w[s];
        }
        thr1[b] = act ? Dmax - pc + 1 : -100000; /* a result is >= -256: an inactive query never hits */
        act_any = act_any || act;
    }
    uint32_t bk0[MK_TOP], bk1[MK_TOP]; /* per query block: the MK_TOP smallest keys this lane has seen, ascending */
#pragma unroll
    for (int k = 0; k < MK_TOP; k++) bk0[k] = bk1[k] = KNONE;
    const bool any_active = __syncthreads_or(act_any) != 0;
    /* staging roles: thread t < 128 owns the candidate record of F2 keypoint t of the chunk; every thread fetches one half
       descriptor (16 B, coalesced) and expands its four words.  The next chunk's global loads are issued before the
       current one is consumed. */
    const int ej = threadIdx.x >> 1, eh = threadIdx.x & 1; /* keypoint ej of the chunk, words 4 eh .. 4 eh + 3 */
    uint4 Wn = make_uint4(0u, 0u, 0u, 0u);
    uint32_t on = KNONE;
    float xn = 0.0f, yn = 0.0f;
#define MM_FETCH(T0)                                                                                                   \
    {                                                                                                                  \
        Wn = (ej < CH && (T0) + ej < n2) ? ((const uint4 *)(D2 + (size_t)((T0) + ej) * 32))[eh] : make_uint4(0u, 0u, 0u, 0u);        \
        on = KNONE;                                                                                                    \
        if (threadIdx.x < CH && (T0) + (int)threadIdx.x < n2) {                                                  \
            const OrbKeyPoint kp_ = k2[(T0) + threadIdx.x];                                                            \
            const uint32_t ord_ = f2_order_key(kp_, (T0) + threadIdx.x, brute, A.min_x, A.min_y, invW, invH);          \
            if (ord_ != 0xFFFFFFFFu) on = brute ? ord_ : ((ord_ >> 16) << 11 | (ord_ & 0x7FFu)); /* cell << 11 | index */ \
            xn = kp_.x;                                                                                                \
            yn = kp_.y;                                                                                                \
        }                                                                                                              \
    }
    if (any_active && n2 > 0) MM_FETCH(0)
    for (int t0 = 0; any_active && t0 < n2; t0 += CH) {
        const int nt = min(CH, n2 - t0);
        __syncthreads(); /* the previous chunk has been consumed */
        if (threadIdx.x < CH) {
            s_o[threadIdx.x] = on;
            s_x[threadIdx.x] = xn;
            s_y[threadIdx.x] = yn;
        }
        {
            const int base = ((ej >> 5) * 8 + 4 * eh) * 64 + (ej & 31); /* tile ej / 32, K-step 4 eh + k, row ej % 32 */
            if (ej < CH) {
            s_raw[threadIdx.x] = Wn;
            s_a[base] = mm_bits01(Wn.x, 0);
            s_a[base + 32] = mm_bits01(Wn.x, 1);
            s_a[base + 64] = mm_bits01(Wn.y, 0);
            s_a[base + 96] = mm_bits01(Wn.y, 1);
            s_a[base + 128] = mm_bits01(Wn.z, 0);
            s_a[base + 160] = mm_bits01(Wn.z, 1);
            s_a[base + 192] = mm_bits01(Wn.w, 0);
            s_a[base + 224] = mm_bits01(Wn.w, 1);
            }
        }
        __syncthreads();
        if (t0 + CH < n2) MM_FETCH(t0 + CH)
        const int ntile = (nt + 31) >> 5;
        for (int tile = 0; tile < ntile; tile++) {
            mm_v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
#pragma unroll
            for (int s = 0; s < 8; s++) {
                const mm_v4i a = s_a[(tile * 8 + s) * 64 + lane];
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[0][s], acc0, 0, 0, 0);
                if (QB == 2) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[1][s], acc1, 0, 0, 0);
            }
            mm_scan_tile(acc0, thr1[0], qw0, px[0], py[0], bk0, tile, h, brute, r, A.nnratio, s_o, s_x, s_y, s_raw);
            if (QB == 2) mm_scan_tile(acc1, thr1[1], qw1, px[1], py[1], bk1, tile, h, brute, r, A.nnratio, s_o, s_x, s_y, s_raw);
        }
    }
    /* lanes c and c + 32 hold the two halves (rows 4 h ..) of the same queries: each takes in the other's keys */
    mm_merge_halves(bk0);
    if (QB == 2) mm_merge_halves(bk1);
    /* QB = 2: lane (c, h) writes query 32 h + c of the wavefront = block h, column c; QB = 1: the lower half writes */
    const int i1 = QB == 2 ? qbase + lane : qbase + c;
    if (i1 < n1 && (QB == 2 || h == 0)) {
        uint64_t out[MK_TOP];
#pragma unroll
        for (int k = 0; k < MK_TOP; k++) {
            const uint32_t k32 = (QB == 2 && h) ? bk1[k] : bk0[k];
            if (k32 == KNONE)
                out[k] = ~0ull;
            else if (brute)
                out[k] = ((uint64_t)(k32 >> 16) << 32) | (k32 & 0xFFFFu);
            else
                out[k] = ((uint64_t)(k32 >> 23) << 32) | (uint64_t)((((k32 >> 11) & 0xFFFu) << 16) | (k32 & 0x7FFu));
        }
#pragma unroll
        for (int k = 0; k < MK_TOP / 2; k++)
            A.knn[((size_t)pair * A.cap + i1) * (MK_TOP / 2) + k] = make_ulonglong2(out[2 * k], out[2 * k + 1]);
    }
}

/* The plain brute-force 2-NN primitive on the matrix cores (n2 <= 65536; k_hamming_knn2 beyond): the products of
 * k_match_knn_mfma, but EVERY distance matters, so each result becomes a key (d << 16 | j) -- one v_lshl_add and one
 * add of the register's row constant -- and goes through a two-entry sorted insert (v_min + v_med3): 4 VALU
 * instructions per distance-lane where xor / popcount / compare took ~21.  Ties resolve to the lowest index because
 * the index is the low half of the key; the second-best DISTANCE is the distance of the second-smallest key (duplicates
 * count), exactly what the sequential scan keeps.  32 queries per wavefront, 128 per workgroup, F2 in 32-row stages. */
__global__ __launch_bounds__(256) void k_hamming_knn2_mfma(const uint8_t *d1, int n1, const uint8_t *d2, int n2, int32_t *idx,
                                                            uint16_t *best, uint16_t *second) {
    __shared__ mm_v4i s_a[8 * 64];
    const uint32_t KNONE = 0xFFFFFFFFu;
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int qi = blockIdx.x * 128 + (threadIdx.x >> 6) * 32 + c;
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qi < n1) {
        const uint4 *p = (const uint4 *)(d1 + (size_t)qi * 32);
        const uint4 v0 = p[0], v1 = p[1];
        w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w;
        w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
    }
    mm_v4i bq[8];
    int pq = 0;
#pragma unroll
    for (int s = 0; s < 8; s++) {
        pq += __popc(w[s]);
        bq[s] = mm_bits_pm(w[s], h);
    }
    uint32_t b1 = KNONE, b2 = KNONE;
    const int ej = threadIdx.x >> 3, es = threadIdx.x & 7; /* staging: word es of F2 descriptor ej of the stage */
    uint32_t Wn = (n2 > 0 && ej < n2) ? ((const uint32_t *)(d2 + (size_t)ej * 32))[es] : 0u;
    for (int t0 = 0; t0 < n2; t0 += 32) {
        __syncthreads(); /* the previous stage has been consumed */
        s_a[es * 64 + ej] = mm_bits01(Wn, 0);
        s_a[es * 64 + 32 + ej] = mm_bits01(Wn, 1);
        __syncthreads();
        Wn = (t0 + 32 + ej < n2) ? ((const uint32_t *)(d2 + (size_t)(t0 + 32 + ej) * 32))[es] : 0u;
        mm_v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 8; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(s_a[s * 64 + lane], bq[s], acc, 0, 0, 0);
        const uint32_t cb = ((uint32_t)pq << 16) + (uint32_t)(t0 + 4 * h); /* (pq + result) << 16 | F2 index of row 4 h */
        const bool tail = t0 + 32 > n2; /* rows beyond the set (zero operands): their keys are dropped */
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int row = (i & 3) + 8 * (i >> 2); /* C/D row of register i, minus 4 h */
            uint32_t key = ((uint32_t)acc[i] << 16) + cb + (uint32_t)row;
            if (tail && t0 + 4 * h + row >= n2) key = KNONE;
            b2 = umed3(b1, b2, key);
            b1 = key < b1 ? key : b1;
        }
    }
    { /* lanes c and c + 32 saw the two halves of the rows */
        const uint32_t o1 = (uint32_t)__shfl_xor((int)b1, 32), o2 = (uint32_t)__shfl_xor((int)b2, 32);
        b2 = umed3(b1, b2, o1);
        b1 = o1 < b1 ? o1 : b1;
        b2 = umed3(b1, b2, o2);
        b1 = o2 < b1 ? o2 : b1;
    }
    if (qi < n1 && h == 0) {
        idx[qi] = b1 == KNONE ? -1 : (int32_t)(b1 & 0xFFFFu);
        best[qi] = (uint16_t)(b1 == KNONE ? 0xFFFFu : b1 >> 16);
        second[qi] = (uint16_t)(b2 == KNONE ? 0xFFFFu : b2 >> 16);
    }
}

/* PHASE 2 (serial resolution) -- one wavefront per frame pair; dynamic LDS: per F2 keypoint {float x,y; u32 ord; i32 m21; float angle; u16 mdist} */
__global__ __launch_bounds__(64) void k_match_init(MatchArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int pair = blockIdx.x, lane = threadIdx.x;
    const int fa = A.pair_a ? A.pair_a[pair] : 2 * pair, fb = A.pair_b ? A.pair_b[pair] : 2 * pair + 1;
    const int n1 = A.counts[fa], n2 = A.counts[fb];
    const OrbKeyPoint *k1 = A.kps + (size_t)fa * A.cap, *k2 = A.kps + (size_t)fb * A.cap;
    const uint8_t *D1 = A.desc + (size_t)fa * A.cap * 32, *D2 = A.desc + (size_t)fb * A.cap * 32;
    float *s_x = (float *)smem;
    float *s_y = s_x + A.cap;
    uint32_t *s_ord = (uint32_t *)(s_y + A.cap); /* 0xFFFFFFFF = not a candidate for anybody */
    int32_t *s_m21 = (int32_t *)(s_ord + A.cap);
    float *s_ang = (float *)(s_m21 + A.cap);      /* F2 keypoint angles */
    int32_t *s_m12 = (int32_t *)(s_ang + A.cap);   /* vnMatches12, kept in LDS: no global store inside the serial loop */
    uint16_t *s_md = (uint16_t *)(s_m12 + A.cap);  /* 0xFFFF = INT_MAX */
    uint8_t *s_bin = (uint8_t *)(s_md + A.cap);    /* rotation bin of each i1 */
    int32_t *s_stamp = (int32_t *)(smem + (((size_t)A.cap * 27 + 15) & ~(size_t)15)); /* speculation stamps, one per F2 keypoint */
    uint4 *s_desc = (uint4 *)(smem + (((size_t)A.cap * 27 + 15) & ~(size_t)15) + (((size_t)A.cap * 4 + 15) & ~(size_t)15)); /* optional: F2 descriptors */
    __shared__ int s_hist[HISTO_LENGTH];
    int32_t *m12 = s_m12;
    uint8_t *bins = s_bin;
    int32_t *m12_out = A.matches12 + (size_t)pair * A.cap;
    const bool brute = A.window < 0;
    const float invW = (float)GRID_COLS / (A.max_x - A.min_x); /* src/Frame.cpp:59-60 */
    const float invH = (float)GRID_ROWS / (A.max_y - A.min_y);
    for (int j0 = lane; j0 < n2; j0 += 64 * 4) { /* four keypoints per lane in flight before the LDS stores */
        OrbKeyPoint kp[4];
#pragma unroll
        for (int u = 0; u < 4; u++) kp[u] = k2[min(j0 + u * 64, n2 - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int j = j0 + u * 64;
            if (j < n2) {
                s_x[j] = kp[u].x;
                s_y[j] = kp[u].y;
                s_ang[j] = kp[u].angle;
                s_ord[j] = f2_order_key(kp[u], j, brute, A.min_x, A.min_y, invW, invH);
                s_m21[j] = -1;
                s_md[j] = 0xFFFF;
                s_stamp[j] = 0x7FFFFFFF;
            }
        }
    }
    for (int i = lane; i < n1; i += 64) {
        m12[i] = -1;
        bins[i] = 0xFF;
    }
    if (A.desc_in_lds) { /* eight 16-byte loads in flight per lane before the first LDS store */
        for (int i0 = lane; i0 < n2 * 2; i0 += 64 * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = (i0 + u * 64 < n2 * 2) ? ((const uint4 *)D2)[i0 + u * 64] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (i0 + u * 64 < n2 * 2) s_desc[i0 + u * 64] = v[u];
        }
    }
    if (lane < HISTO_LENGTH) s_hist[lane] = 0;
    __syncthreads();
    const float r = (float)A.window;
    const float factor = HISTO_LENGTH / 360.0f;
    const ulonglong2 *knn = A.knn + (size_t)pair * A.cap * (MK_TOP / 2);
    for (int base = 0; base < n1; base += 64) {
      /* The phase-1 result of 64 queries, one per lane.  The greedy loop of the reference is sequential in i1,
       * but query i1 only READS vMatchedDistance of the few candidates it inspects and only WRITES the state of
       * the one F2 keypoint it matches, so the chunk is resolved in speculative rounds: every pending lane
       * decides from the current state; a lane is valid if no earlier pending lane that accepts a match writes
       * into its read set (LDS stamp array, atomicMin of the lane id); the valid prefix commits -- its writes
       * are to distinct F2 keypoints, hence order-free -- and the first invalid lane opens the next round.
       * Results are identical to the sequential walk; a round retires ~15 queries instead of one. */
      ulonglong2 mk[MK_TOP / 2];
#pragma unroll
      for (int k = 0; k < MK_TOP / 2; k++) mk[k] = make_ulonglong2(~0ull, ~0ull);
      float ang1 = 0.f;
      if (base + lane < n1) {
#pragma unroll
          for (int k = 0; k < MK_TOP / 2; k++) mk[k] = knn[(size_t)(base + lane) * (MK_TOP / 2) + k];
          ang1 = k1[base + lane].angle;
      }
      uint64_t c[MK_TOP];
#pragma unroll
      for (int k = 0; k < MK_TOP / 2; k++) {
          c[2 * k] = mk[k].x;
          c[2 * k + 1] = mk[k].y;
      }
      const int nchunk = min(64, n1 - base);
      int done = 0;
      while (done < nchunk) {
        const bool act = lane >= done && lane < nchunk;
        /* ---- speculative decision of every pending query (the fast path of the sequential walk) ---- */
        /* the filtered candidate set (:49-50) is a subset of the unfiltered one, so its two smallest keys are
           the first two ALLOWED entries of the sorted unfiltered list -- if they show up within the top MK_TOP */
        int mdv[MK_TOP];
#pragma unroll
        for (int k = 0; k < MK_TOP; k++) {
            const int jk = (c[k] == ~0ull) ? 0 : (int)(c[k] & 0xFFFFu);
            const uint16_t raw = s_md[jk];
            mdv[k] = raw == 0xFFFF ? 0x7FFFFFFF : (int)raw;
        }
        uint64_t gKey = ~0ull;
        int bestDist2 = 0x7FFFFFFF;
        int found = 0, n_insp = MK_TOP;
        bool complete = false; /* the list is the whole candidate set */
#pragma unroll
        for (int k = 0; k < MK_TOP; k++) {
            if (found < 2 && !complete) {
                if (c[k] == ~0ull) {
                    complete = true;
                    n_insp = k;
                } else {
                    const int dk = (int)(c[k] >> 32);
                    if (!(mdv[k] <= dk)) {
                        if (found == 0)
                            gKey = c[k];
                        else
                            bestDist2 = dk;
                        found++;
                        if (found == 2) n_insp = k + 1;
                    }
                }
            }
        }
        const bool slow = act && !(found == 2 || complete); /* needs the full rescan: resolved alone, below */
        const int bestDist = (int)(gKey >> 32);
        const int W = (int)(gKey & 0xFFFFu);
        const bool accept = act && !slow && gKey != ~0ull && bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * A.nnratio;
        if (accept) atomicMin(&s_stamp[W], lane);
        __syncthreads();
        bool conflict = false;
        if (act && !slow) {
#pragma unroll
            for (int k = 0; k < MK_TOP; k++)
                if (k < n_insp) conflict = conflict || s_stamp[(int)(c[k] & 0xFFFFu)] < lane;
        }
        const uint64_t badm = __ballot(act && (conflict || slow));
        const int first_bad = badm ? (int)__builtin_ctzll(badm) : 64;
        __syncthreads();
        if (accept) s_stamp[W] = 0x7FFFFFFF; /* every stamp of this round is cleared again */
        if (accept && lane < first_bad) {
            const int i1 = base + lane;
            const int old = s_m21[W];
            if (old >= 0) m12[old] = -1;
            m12[i1] = W;
            s_m21[W] = i1;
            s_md[W] = (uint16_t)bestDist;
            if (A.check_ori) {
                float rot = ang1 - s_ang[W];
                if (rot < 0.0f) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                bins[i1] = (uint8_t)bin;
                atomicAdd(&s_hist[bin], 1);
            }
        }
        __syncthreads();
        done = min(first_bad, nchunk);
        if (done < nchunk && ((__ballot(slow) >> done) & 1ull)) {
        /* ---- the rare query whose top MK_TOP list ran out of allowed entries: full rescan, wave-wide ---- */
        const int t = done;
        const int i1 = base + t;
        const float angle1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ang1), t));
        const OrbKeyPoint kp1s = k1[i1];
        float px = kp1s.x, py = kp1s.y;
        if (A.prev_xy) {
            px = A.prev_xy[((size_t)pair * A.cap + i1) * 2];
            py = A.prev_xy[((size_t)pair * A.cap + i1) * 2 + 1];
        }
        uint32_t a[8];
        {
            const uint4 *p = (const uint4 *)(D1 + (size_t)i1 * 32);
            uint4 v0 = p[0], v1 = p[1];
            a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
            a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
        }
        uint64_t bestKey = ~0ull; /* dist << 32 | ord */
        int second = 0x7FFFFFFF;
        for (int j = lane; j < n2; j += 64) {
            const uint32_t ord = s_ord[j];
            bool cand = ord != 0xFFFFFFFFu;
            if (cand && !brute) {
                const float dx = s_x[j] - px, dy = s_y[j] - py;
                cand = fabsf(dx) < r && fabsf(dy) < r;
            }
            if (cand) {
                const uint4 *p = A.desc_in_lds ? (s_desc + 2 * j) : (const uint4 *)(D2 + (size_t)j * 32);
                const int d = hamming256(a, p[0], p[1]);
                const int md = s_md[j] == 0xFFFF ? 0x7FFFFFFF : (int)s_md[j];
                if (!(md <= d)) { /* :49-50 */
                    const uint64_t key = ((uint64_t)(uint32_t)d << 32) | ord;
                    if (key < bestKey) {
                        if (bestKey != ~0ull) second = min(second, (int)(bestKey >> 32));
                        bestKey = key;
                    } else
                        second = min(second, d);
                }
            }
        }
        const uint64_t sKey = wave_min_u64(bestKey);
        const int alt = (bestKey == sKey) ? second : (bestKey == ~0ull ? 0x7FFFFFFF : (int)(bestKey >> 32));
        const int sDist2 = wave_min_i32(alt);
        if (sKey != ~0ull) {
            const int sDist = (int)(sKey >> 32);
            const int sIdx2 = (int)(sKey & 0xFFFFu);
            if (sDist <= TH_LOW && (float)sDist < (float)sDist2 * A.nnratio) {
                if (lane == 0) {
                    const int old = s_m21[sIdx2];
                    if (old >= 0) m12[old] = -1;
                    m12[i1] = sIdx2;
                    s_m21[sIdx2] = i1;
                    s_md[sIdx2] = (uint16_t)sDist;
                    if (A.check_ori) {
                        float rot = angle1 - s_ang[sIdx2];
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        bins[i1] = (uint8_t)bin;
                        s_hist[bin]++;
                    }
                }
            }
        }
        __syncthreads(); /* lane 0's LDS updates before the next round (single-wave workgroup) */
        done++;
        }
      }
    }
    /* nmatches == number of i1 with m12 >= 0 (every accept adds one and every steal removes one) */
    __threadfence_block();
    __syncthreads();
    if (A.check_ori) { /* :95-118 with ComputeThreeMaxima :147-188 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int s = s_hist[i];
            if (s > max1) {
                max3 = max2; max2 = max1; max1 = s;
                ind3 = ind2; ind2 = ind1; ind1 = i;
            } else if (s > max2) {
                max3 = max2; max2 = s;
                ind3 = ind2; ind2 = i;
            } else if (s > max3) {
                max3 = s;
                ind3 = i;
            }
        }
        if (max2 < 0.1f * (float)max1) {
            ind2 = -1;
            ind3 = -1;
        } else if (max3 < 0.1f * (float)max1) {
            ind3 = -1;
        }
        for (int i = lane; i < n1; i += 64) {
            const int b = bins[i];
            if (b != 0xFF && b != ind1 && b != ind2 && b != ind3 && m12[i] >= 0) m12[i] = -1;
        }
        __syncthreads();
    }
    int cnt = 0;
    for (int i = lane; i < n1; i += 64) {
        const int m = m12[i];
        m12_out[i] = m;
        if (m >= 0) {
            cnt++;
            if (A.prev_out) {
                A.prev_out[((size_t)pair * A.cap + i) * 2] = s_x[m];
                A.prev_out[((size_t)pair * A.cap + i) * 2 + 1] = s_y[m];
            }
        } else if (A.prev_out && A.prev_xy) {
            A.prev_out[((size_t)pair * A.cap + i) * 2] = A.prev_xy[((size_t)pair * A.cap + i) * 2];
            A.prev_out[((size_t)pair * A.cap + i) * 2 + 1] = A.prev_xy[((size_t)pair * A.cap + i) * 2 + 1];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) A.nmatches[pair] = cnt;
}

/* Frame::AssignFeaturesToGrid + Frame::GetFeaturesInArea (reference src/Frame.cpp:144-168, 219-271) for a batch
 * of queries against one frame, without materialising the 64 x 48 grid (see the file header: a keypoint is
 * returned iff it was assigned to the grid, passes the level filter and lies strictly inside the square window;
 * the reference's order is (cell x, cell y, insertion order) = ascending (cell << 16 | index)).  One wavefront
 * per query: candidates are collected in index order by ballot compaction, then every candidate's output slot
 * is its rank among the collected keys. */
__global__ __launch_bounds__(64) void k_features_in_area(const OrbKeyPoint *kps, int n, float min_x, float max_x, float min_y,
                                                         float max_y, const float *qxyr, const int32_t *qlevels, int nq,
                                                         int32_t *out_idx, int cap, int32_t *out_count) {
    extern __shared__ uint32_t fa_keys[]; /* n entries */
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= nq) return;
    const float x = qxyr[3 * q], y = qxyr[3 * q + 1], r = qxyr[3 * q + 2];
    const int minLevel = qlevels ? qlevels[2 * q] : -1, maxLevel = qlevels ? qlevels[2 * q + 1] : -1;
    const bool check = (minLevel > 0) || (maxLevel >= 0);
    const float invW = (float)GRID_COLS / (max_x - min_x), invH = (float)GRID_ROWS / (max_y - min_y);
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int nc = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        bool cand = false;
        uint32_t key = 0;
        if (j < n) {
            const OrbKeyPoint kp = kps[j];
            const int ix = (int)round(((double)kp.x - (double)min_x) * (double)invW);
            const int iy = (int)round(((double)kp.y - (double)min_y) * (double)invH);
            cand = ix >= 0 && ix < GRID_COLS && iy >= 0 && iy < GRID_ROWS;
            if (check) {
                if (kp.octave < minLevel) cand = false;
                if (maxLevel >= 0 && kp.octave > maxLevel) cand = false;
            }
            cand = cand && fabsf(kp.x - x) < r && fabsf(kp.y - y) < r;
            key = ((uint32_t)(ix * GRID_ROWS + iy) << 16) | (uint32_t)j;
        }
        const uint64_t m = __ballot(cand);
        if (cand) fa_keys[nc + __popcll(m & lt)] = key;
        nc += __popcll(m);
    }
    __syncthreads();
    for (int i = lane; i < nc; i += 64) {
        const uint32_t key = fa_keys[i];
        int rank = 0;
        for (int k = 0; k < nc; k++) rank += fa_keys[k] < key ? 1 : 0;
        if (rank < cap) out_idx[(size_t)q * cap + rank] = (int32_t)(key & 0xFFFFu);
    }
    if (lane == 0) out_count[q] = nc;
}


/* ------------------------------------------------------------------------------------------------------------------
 * SURVEY N4 -- the candidate scan behind ORBmatcher::SearchByProjection / SearchByBoW (empty bodies in the
 * reference, include/ORBmatcher.h:22,24; upstream ORB-SLAM2 semantics, see oracle/match_oracle.cpp).  One query
 * per lane: a descriptor, a window (x, y, r; r < 0: every keypoint is a candidate, in index order) and a level
 * range; the frame's keypoints stream through LDS in 512-keypoint tiles; kept: the QK_TOP smallest
 * (distance << 32 | grid cell << 16 | index) keys with distance <= keep_max_d -- a farther candidate cannot change
 * what the sequential walk decides (SearchByProjection accepts on best <= TH_HIGH alone; SearchByBoW's ratio test
 * needs a second best only while TH_LOW >= d * ratio), so a list that ends early is complete for that purpose.
 * The walk itself (a feature taken by an earlier query is closed to later ones) runs on the host over these
 * lists. */
#define QK_TOP 8
#define QK_SLICES 8 /* threads per query of k_query_topk */
struct QueryArgs {
    const OrbKeyPoint *kps2;
    const uint8_t *desc2;
    int n2;
    const int32_t *n2_dev; /* NULL, or the frame's feature count in device memory (then n2 above is only a bound) */
    const uint8_t *qdesc; /* [nq][32] */
    const float *qxyr;    /* [nq][3] */
    const int32_t *qlvl;  /* [nq][2] (minLevel, maxLevel) or NULL */
    int nq;
    float min_x, max_x, min_y, max_y;
    int keep_max_d;
    unsigned long long *out; /* [nq][QK_TOP], ~0 = none */
};
__global__ __launch_bounds__(256) void k_query_topk(QueryArgs A) {
    __shared__ uint4 s_d[MK_TILE * 2];
    __shared__ float s_x[MK_TILE], s_y[MK_TILE];
    __shared__ uint32_t s_o[MK_TILE];
    __shared__ int s_oct[MK_TILE];
    __shared__ uint64_t s_m[256 * QK_TOP]; /* the slices' candidate lists of a query, merged by its first thread */
    /* QK_SLICES threads per query, each scanning every QK_SLICES-th feature of a tile: a frame's ~1000 queries are 32
       workgroups instead of 4 (one thread per query left 252 CUs idle for the 0.27 ms the scan took).  The keys
       (distance, order) are unique, so the QK_TOP smallest of the slices' lists are the QK_TOP smallest overall. */
    const int ql = threadIdx.x / QK_SLICES, sl = threadIdx.x - ql * QK_SLICES;
    const int q = blockIdx.x * (256 / QK_SLICES) + ql;
    const int n2 = A.n2_dev ? min(*A.n2_dev, A.n2) : A.n2;
    const float invW = (float)GRID_COLS / (A.max_x - A.min_x), invH = (float)GRID_ROWS / (A.max_y - A.min_y);
    uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float x = 0, y = 0, r = 0;
    int lo = -1, hi = -1;
    if (q < A.nq) {
        const uint4 *p = (const uint4 *)(A.qdesc + (size_t)q * 32);
        const uint4 v0 = p[0], v1 = p[1];
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
        a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
        x = A.qxyr[3 * q];
        y = A.qxyr[3 * q + 1];
        r = A.qxyr[3 * q + 2];
        if (A.qlvl) {
            lo = A.qlvl[2 * q];
            hi = A.qlvl[2 * q + 1];
        }
    }
    const bool brute = r < 0.f;
    const bool check = (lo > 0) || (hi >= 0); /* src/Frame.cpp:245 */
    uint64_t bk[QK_TOP];
#pragma unroll
    for (int k = 0; k < QK_TOP; k++) bk[k] = ~0ull;
    for (int t0 = 0; t0 < n2; t0 += MK_TILE) {
        const int nt = min(MK_TILE, n2 - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < nt; i += 256) {
            const OrbKeyPoint kp = A.kps2[t0 + i];
            const uint4 *src = (const uint4 *)(A.desc2 + (size_t)(t0 + i) * 32);
            s_d[2 * i] = src[0];
            s_d[2 * i + 1] = src[1];
            s_x[i] = kp.x;
            s_y[i] = kp.y;
            s_oct[i] = kp.octave;
            const int ix = (int)round(((double)kp.x - (double)A.min_x) * (double)invW);
            const int iy = (int)round(((double)kp.y - (double)A.min_y) * (double)invH);
            s_o[i] = (ix >= 0 && ix < GRID_COLS && iy >= 0 && iy < GRID_ROWS) ? ((uint32_t)(ix * GRID_ROWS + iy) << 16) | (uint32_t)(t0 + i)
                                                                              : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (q < A.nq) {
            for (int j = sl; j < nt; j += QK_SLICES) {
                bool cand;
                uint32_t ord;
                if (brute) {
                    cand = true;
                    ord = (uint32_t)(t0 + j);
                } else {
                    ord = s_o[j];
                    cand = ord != 0xFFFFFFFFu;
                    if (check) {
                        const int oc = s_oct[j];
                        if (oc < lo) cand = false;
                        if (hi >= 0 && oc > hi) cand = false;
                    }
                    cand = cand && fabsf(s_x[j] - x) < r && fabsf(s_y[j] - y) < r;
                }
                if (!cand) continue;
                const int d = hamming256(a, s_d[2 * j], s_d[2 * j + 1]);
                if (d > A.keep_max_d) continue;
                const uint64_t key = ((uint64_t)(uint32_t)d << 32) | ord;
                if (key < bk[QK_TOP - 1]) {
#pragma unroll
                    for (int k = QK_TOP - 1; k > 0; k--) bk[k] = key < bk[k - 1] ? bk[k - 1] : (key < bk[k] ? key : bk[k]);
                    bk[0] = key < bk[0] ? key : bk[0];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < QK_TOP; k++) s_m[threadIdx.x * QK_TOP + k] = bk[k];
    __syncthreads();
    if (q < A.nq && sl == 0) {
        for (int o = 1; o < QK_SLICES; o++)
            for (int k2 = 0; k2 < QK_TOP; k2++) {
                const uint64_t key = s_m[(threadIdx.x + o) * QK_TOP + k2];
                if (key >= bk[QK_TOP - 1]) break; /* (a slice's list is sorted) */
#pragma unroll
                for (int k = QK_TOP - 1; k > 0; k--) bk[k] = key < bk[k - 1] ? bk[k - 1] : (key < bk[k] ? key : bk[k]);
                bk[0] = key < bk[0] ? key : bk[0];
            }
#pragma unroll
        for (int k = 0; k < QK_TOP; k++) A.out[(size_t)q * QK_TOP + k] = bk[k];
    }
}


/* projection of the last frame's map points with the current pose: query window and level range per feature
 * (upstream SearchByProjection, monocular).  A feature that takes no part gets r = 0: no candidate passes |dx| < 0. */
/* pose (R row-major, t) and per-level scale factors of an N4 projection, passed BY VALUE as a kernel argument: no upload,
   nothing for an asynchronous caller to keep alive */
struct N4Pose {
    double pose12[12];
    float scale_factors[64];
};
__global__ void k_project_queries(const OrbKeyPoint *last_kps, const uint8_t *flags, const double *xw, int n, N4Pose PP,
                                  double fx, double fy, double cx, double cy, float min_x, float max_x, float min_y, float max_y,
                                  int nlevels, float th, float *qxyr, int32_t *qlvl) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *pose12 = PP.pose12;
    const float *scale_factors = PP.scale_factors;
    float u = 0.f, v = 0.f, r = 0.f;
    int lo = 0, hi = 0;
    if (flags[i] & 1) {
        const double *X = xw + (size_t)i * 3, *R = pose12, *t = pose12 + 9;
        const double xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0];
        const double yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1];
        const double zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2];
        const double invzc = 1.0 / zc;
        const int oct = last_kps[i].octave;
        if (!(invzc < 0) && oct >= 0 && oct < nlevels) {
            const float uu = (float)(fx * xc * invzc + cx), vv = (float)(fy * yc * invzc + cy);
            if (!(uu < min_x || uu > max_x) && !(vv < min_y || vv > max_y)) {
                u = uu;
                v = vv;
                r = th * scale_factors[oct];
                lo = oct - 1;
                hi = oct + 1;
            }
        }
    }
    qxyr[3 * i] = u;
    qxyr[3 * i + 1] = v;
    qxyr[3 * i + 2] = r;
    qlvl[2 * i] = lo;
    qlvl[2 * i + 1] = hi;
}

/* ------------------------------------------------------------------------------------------------ */
static OrbErrSlot m_err; /* forwards to orb_last_error() */
#define MCHK(expr)                                                                         \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            m_err = std::string(#expr) + ": " + hipGetErrorString(_e);                     \
            return ORB_E_HIP;                                                              \
        }                                                                                  \
    } while (0)

struct OrbMatcher {
    int device = 0;
    hipStream_t stream = nullptr;
    int max_kp = 0, max_pairs = 0;
    /* host-API staging: two frames */
    OrbKeyPoint *d_kps = nullptr; /* [2][max_kp] */
    uint8_t *d_desc = nullptr;    /* [2][max_kp][32] */
    int32_t *d_counts = nullptr;  /* [2] */
    float *d_prev = nullptr, *d_prev_out = nullptr; /* [max_kp][2] */
    uint8_t *d_in = nullptr;  /* one device block behind d_counts / d_kps / d_desc / d_prev: the host-pointer path uploads it in one copy */
    uint8_t *h_up = nullptr;  /* pinned mirror of d_in */
    uint8_t *h_down = nullptr; /* pinned results of the host-pointer path: [nmatches 64 B | matches12 K*4 | prev_out K*8] */
    size_t in_bytes = 0, off_kps = 0, off_desc = 0, off_prev = 0;
    int32_t *d_m12 = nullptr;     /* [max_pairs][max_kp] */
    int32_t *d_nm = nullptr;      /* [max_pairs] */
    uint8_t *d_bins = nullptr;    /* [max_pairs][max_kp] */
    ulonglong2 *d_knn = nullptr;  /* [max_pairs][max_kp] */
    int32_t *d_idx = nullptr;
    uint16_t *d_best = nullptr, *d_second = nullptr;
    uint8_t *d_scratch = nullptr; /* N4 searches: query parameters and result lists, grown on demand */
    size_t cap_scratch = 0;
    uint8_t *h_n4 = nullptr;      /* pinned mirror of d_scratch (one upload, one download per N4 search) */
    size_t cap_h_n4 = 0;
    bool profile = false;
    static const int kProfRing = 64;
    hipEvent_t ev[64][3] = {}; /* start, end, between phase 1 (knn) and phase 2 (resolve) */
    long prof_calls = 0;
};

extern "C" int orb_matcher_create(int max_keypoints, int max_pairs, int device_id, OrbMatcher **out) {
    if (!out || max_keypoints <= 0 || max_pairs <= 0 || max_keypoints > 65535) return ORB_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return ORB_E_NO_DEVICE;
    MCHK(hipSetDevice(device_id));
    OrbMatcher *m = new OrbMatcher();
    m->device = device_id;
    m->max_kp = max_keypoints;
    m->max_pairs = max_pairs;
    const size_t K = (size_t)max_keypoints;
    /* input block: [counts 64 B | kps 2K x 28 | desc 2K x 32 | prev K x 8], every part 64-B aligned */
    auto al64 = [](size_t v) { return (v + 63) & ~(size_t)63; };
    m->off_kps = 64;
    m->off_desc = al64(m->off_kps + 2 * K * sizeof(OrbKeyPoint));
    m->off_prev = al64(m->off_desc + 2 * K * 32);
    m->in_bytes = al64(m->off_prev + K * 8);
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&m->d_in, m->in_bytes) != hipSuccess || hipHostMalloc((void **)&m->h_up, m->in_bytes) != hipSuccess ||
        hipHostMalloc((void **)&m->h_down, 64 + K * 12) != hipSuccess || hipMalloc((void **)&m->d_prev_out, K * 8) != hipSuccess ||
        hipMalloc((void **)&m->d_m12, (size_t)max_pairs * K * 4) != hipSuccess ||
        hipMalloc((void **)&m->d_nm, (size_t)max_pairs * 4) != hipSuccess ||
        hipMalloc((void **)&m->d_bins, (size_t)max_pairs * K) != hipSuccess ||
        hipMalloc((void **)&m->d_knn, (size_t)max_pairs * K * (MK_TOP / 2) * sizeof(ulonglong2)) != hipSuccess ||
        hipMalloc((void **)&m->d_idx, K * 4) != hipSuccess || hipMalloc((void **)&m->d_best, K * 2) != hipSuccess ||
        hipMalloc((void **)&m->d_second, K * 2) != hipSuccess) {
        orb_matcher_destroy(m);
        return ORB_E_HIP;
    }
    m->d_counts = (int32_t *)m->d_in;
    m->d_kps = (OrbKeyPoint *)(m->d_in + m->off_kps);
    m->d_desc = m->d_in + m->off_desc;
    m->d_prev = (float *)(m->d_in + m->off_prev);
    *out = m;
    return ORB_OK;
}

extern "C" void orb_matcher_destroy(OrbMatcher *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    (void)hipFree(m->d_in); /* d_counts / d_kps / d_desc / d_prev live inside it */
    if (m->h_up) (void)hipHostFree(m->h_up);
    if (m->h_down) (void)hipHostFree(m->h_down);
    (void)hipFree(m->d_prev_out);
    (void)hipFree(m->d_m12);
    (void)hipFree(m->d_nm);
    (void)hipFree(m->d_bins);
    (void)hipFree(m->d_knn);
    (void)hipFree(m->d_idx);
    (void)hipFree(m->d_best);
    (void)hipFree(m->d_second);
    (void)hipFree(m->d_scratch);
    if (m->h_n4) (void)hipHostFree(m->h_n4);
    for (int r = 0; r < OrbMatcher::kProfRing; r++)
        for (int i = 0; i < 3; i++)
            if (m->ev[r][i]) (void)hipEventDestroy(m->ev[r][i]);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" void *orb_matcher_stream(OrbMatcher *m) { return m ? (void *)m->stream : nullptr; }
extern "C" int orb_matcher_set_profiling(OrbMatcher *m, int enable) {
    if (!m) return ORB_E_INVALID;
    MCHK(hipSetDevice(m->device));
    if (enable && !m->ev[0][0])
        for (int r = 0; r < OrbMatcher::kProfRing; r++)
            for (int i = 0; i < 3; i++) MCHK(hipEventCreate(&m->ev[r][i]));
    m->profile = enable != 0;
    m->prof_calls = 0;
    return ORB_OK;
}
/* average kernel duration over the (at most 64 most recent) batch calls since profiling was enabled */
extern "C" int orb_matcher_get_ms(OrbMatcher *m, float *ms) {
    if (!m || !ms || !m->ev[0][0] || m->prof_calls == 0) return ORB_E_INVALID;
    MCHK(hipSetDevice(m->device));
    const int n = (int)(m->prof_calls < OrbMatcher::kProfRing ? m->prof_calls : OrbMatcher::kProfRing);
    *ms = 0.f;
    for (int r = 0; r < n; r++) {
        float t = 0.f;
        MCHK(hipEventSynchronize(m->ev[r][1]));
        MCHK(hipEventElapsedTime(&t, m->ev[r][0], m->ev[r][1]));
        *ms += t / n;
    }
    m->prof_calls = 0;
    return ORB_OK;
}
/* the same average split into the two kernels: ms2[0] = phase 1 (k_match_knn), ms2[1] = phase 2 (k_match_init) */
extern "C" int orb_matcher_get_stage_ms(OrbMatcher *m, float *ms2) {
    if (!m || !ms2 || !m->ev[0][0] || m->prof_calls == 0) return ORB_E_INVALID;
    MCHK(hipSetDevice(m->device));
    const int n = (int)(m->prof_calls < OrbMatcher::kProfRing ? m->prof_calls : OrbMatcher::kProfRing);
    ms2[0] = ms2[1] = 0.f;
    for (int r = 0; r < n; r++) {
        float t = 0.f;
        MCHK(hipEventSynchronize(m->ev[r][1]));
        MCHK(hipEventElapsedTime(&t, m->ev[r][0], m->ev[r][2]));
        ms2[0] += t / n;
        MCHK(hipEventElapsedTime(&t, m->ev[r][2], m->ev[r][1]));
        ms2[1] += t / n;
    }
    m->prof_calls = 0;
    return ORB_OK;
}
extern "C" int orb_matcher_synchronize(OrbMatcher *m) {
    if (!m) return ORB_E_INVALID;
    MCHK(hipSetDevice(m->device));
    MCHK(hipStreamSynchronize(m->stream));
    return ORB_OK;
}

/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cpp:128-144 (host-side scalar; same SWAR popcount) */
extern "C" int orb_descriptor_distance(const uint8_t *a, const uint8_t *b) {
    uint32_t pa[8], pb[8];
    memcpy(pa, a, 32);
    memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cpp:147-188 */
extern "C" void orb_compute_three_maxima(const int32_t *histo, int L, int32_t *i1, int32_t *i2, int32_t *i3) {
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
    }
    *i1 = ind1;
    *i2 = ind2;
    *i3 = ind3;
}

extern "C" int orb_hamming_knn2(OrbMatcher *m, const uint8_t *d1, int n1, const uint8_t *d2, int n2, int32_t *idx,
                                uint16_t *best, uint16_t *second) {
    if (!m || n1 < 0 || n2 < 0 || (n1 > 0 && (!d1 || !idx || !best || !second)) || (n2 > 0 && !d2)) return ORB_E_INVALID;
    if (n1 > m->max_kp || n2 > m->max_kp) return ORB_E_CAPACITY;
    if (n1 == 0) return ORB_OK;
    MCHK(hipSetDevice(m->device));
    MCHK(hipMemcpyAsync(m->d_desc, d1, (size_t)n1 * 32, hipMemcpyHostToDevice, m->stream));
    if (n2 > 0)
        MCHK(hipMemcpyAsync(m->d_desc + (size_t)m->max_kp * 32, d2, (size_t)n2 * 32, hipMemcpyHostToDevice, m->stream));
    static const int knn2_no_mfma = getenv("ORBSLAM_AMD_MATCH_NO_MFMA") ? 1 : 0;
    if (n2 <= 65536 && !knn2_no_mfma)
        hipLaunchKernelGGL(k_hamming_knn2_mfma, dim3((n1 + 127) / 128), dim3(256), 0, m->stream, (const uint8_t *)m->d_desc, n1,
                           (const uint8_t *)(m->d_desc + (size_t)m->max_kp * 32), n2, m->d_idx, m->d_best, m->d_second);
    else
        hipLaunchKernelGGL(k_hamming_knn2, dim3((n1 + 255) / 256), dim3(256), 0, m->stream, (const uint8_t *)m->d_desc, n1,
                           (const uint8_t *)(m->d_desc + (size_t)m->max_kp * 32), n2, m->d_idx, m->d_best, m->d_second);
    MCHK(hipGetLastError());
    MCHK(hipMemcpyAsync(idx, m->d_idx, (size_t)n1 * 4, hipMemcpyDeviceToHost, m->stream));
    MCHK(hipMemcpyAsync(best, m->d_best, (size_t)n1 * 2, hipMemcpyDeviceToHost, m->stream));
    MCHK(hipMemcpyAsync(second, m->d_second, (size_t)n1 * 2, hipMemcpyDeviceToHost, m->stream));
    MCHK(hipStreamSynchronize(m->stream));
    return ORB_OK;
}

static size_t match_lds_bytes(int cap) { return (((size_t)cap * 27 + 15) & ~(size_t)15) + (((size_t)cap * 4 + 15) & ~(size_t)15) + 64; }

static int launch_match(const MatchArgs &A_in, int n_pairs, hipStream_t st, hipEvent_t mid = nullptr) {
    MatchArgs A = A_in;
    size_t lds = match_lds_bytes(A.cap);
    /* F2 descriptors LDS-resident for the (rare) rescans: worth it for a call of a few pairs (latency); in a batch the 256
       single-wavefront workgroups would each hold another 32 KB of their CU's LDS for the length of the serial walk, which
       the VALU-bound kernels of the other lane then cannot use (160.5 k -> 163 k frames/s without) */
    static const int desc_lds_max = getenv("ORBSLAM_AMD_MATCH_DESC_LDS") ? atoi(getenv("ORBSLAM_AMD_MATCH_DESC_LDS")) : 8;
    A.desc_in_lds = (n_pairs <= desc_lds_max && lds + (size_t)A.cap * 32 <= 96 * 1024) ? 1 : 0;
    if (A.desc_in_lds) lds += (size_t)A.cap * 32;
    if (lds > 160 * 1024 - 1024) {
        m_err = "matcher: keypoint capacity too large for the LDS-resident frame state";
        return ORB_E_CAPACITY;
    }
    static PerDeviceLdsLimit lim;
    MCHK(lim.ensure((const void *)k_match_init, lds));
    static const int no_mfma = getenv("ORBSLAM_AMD_MATCH_NO_MFMA") ? 1 : 0; /* diagnostics: the xor / popcount scan */
    static const int mfma_qb = getenv("ORBSLAM_AMD_MATCH_QB") ? atoi(getenv("ORBSLAM_AMD_MATCH_QB")) : 1;
    static const int mfma_ch = getenv("ORBSLAM_AMD_MATCH_CHUNK") ? atoi(getenv("ORBSLAM_AMD_MATCH_CHUNK")) : 32;
    const bool key32 = A.cap <= (A.window < 0 ? 65536 : 2048);
    if (key32 && !no_mfma && mfma_qb == 1 && mfma_ch <= 32)
        hipLaunchKernelGGL((k_match_knn_mfma<1, 32>), dim3((A.cap + 127) / 128, n_pairs), dim3(256), 0, st, A);
    else if (key32 && !no_mfma && mfma_qb == 1 && mfma_ch <= 64)
        hipLaunchKernelGGL((k_match_knn_mfma<1, 64>), dim3((A.cap + 127) / 128, n_pairs), dim3(256), 0, st, A);
    else if (key32 && !no_mfma && mfma_qb == 1)
        hipLaunchKernelGGL((k_match_knn_mfma<1, 128>), dim3((A.cap + 127) / 128, n_pairs), dim3(256), 0, st, A);
    else if (key32 && !no_mfma)
        hipLaunchKernelGGL((k_match_knn_mfma<2, 128>), dim3((A.cap + 255) / 256, n_pairs), dim3(256), 0, st, A);
    else if (A.cap <= (A.window < 0 ? 65536 : 2048))
        hipLaunchKernelGGL(k_match_knn<true>, dim3((A.cap + 255) / 256, n_pairs), dim3(256), 0, st, A);
    else
        hipLaunchKernelGGL(k_match_knn<false>, dim3((A.cap + 255) / 256, n_pairs), dim3(256), 0, st, A);
    MCHK(hipGetLastError());
    if (mid) MCHK(hipEventRecord(mid, st));
    hipLaunchKernelGGL(k_match_init, dim3(n_pairs), dim3(64), lds, st, A);
    MCHK(hipGetLastError());
    return ORB_OK;
}

extern "C" int orb_match_init(OrbMatcher *m, const OrbFrameView *f1, const OrbFrameView *f2, float *prev_matched_xy,
                              int32_t *matches12, int window_size, float nnratio, int check_orientation,
                              int32_t *n_matches) {
    if (!m || !f1 || !f2 || !n_matches || f1->n < 0 || f2->n < 0 || (f1->n > 0 && (!matches12 || !f1->kps || !f1->desc)) ||
        (f2->n > 0 && (!f2->kps || !f2->desc)) || (window_size >= 0 && f1->n > 0 && !prev_matched_xy))
        return ORB_E_INVALID;
    if (f1->n > m->max_kp || f2->n > m->max_kp) return ORB_E_CAPACITY;
    *n_matches = 0;
    if (f1->n == 0) return ORB_OK;
    MCHK(hipSetDevice(m->device));
    const size_t K = (size_t)m->max_kp;
    int32_t counts[2] = {f1->n, f2->n};
    /* stage the two frames in the pinned mirror of the device input block; a small block goes up in ONE copy
       (pageable sources would cost a blocking bounce per copy), a large one part by part */
    memcpy(m->h_up, counts, 8);
    memcpy(m->h_up + m->off_kps, f1->kps, (size_t)f1->n * sizeof(OrbKeyPoint));
    memcpy(m->h_up + m->off_desc, f1->desc, (size_t)f1->n * 32);
    if (f2->n > 0) {
        memcpy(m->h_up + m->off_kps + K * sizeof(OrbKeyPoint), f2->kps, (size_t)f2->n * sizeof(OrbKeyPoint));
        memcpy(m->h_up + m->off_desc + K * 32, f2->desc, (size_t)f2->n * 32);
    }
    if (prev_matched_xy) memcpy(m->h_up + m->off_prev, prev_matched_xy, (size_t)f1->n * 8);
    if (m->in_bytes <= (size_t)1 << 20) {
        MCHK(hipMemcpyAsync(m->d_in, m->h_up, m->in_bytes, hipMemcpyHostToDevice, m->stream));
    } else {
        MCHK(hipMemcpyAsync(m->d_in, m->h_up, 64, hipMemcpyHostToDevice, m->stream));
        MCHK(hipMemcpyAsync(m->d_kps, m->h_up + m->off_kps, (size_t)f1->n * sizeof(OrbKeyPoint), hipMemcpyHostToDevice, m->stream));
        MCHK(hipMemcpyAsync(m->d_desc, m->h_up + m->off_desc, (size_t)f1->n * 32, hipMemcpyHostToDevice, m->stream));
        if (f2->n > 0) {
            MCHK(hipMemcpyAsync(m->d_kps + K, m->h_up + m->off_kps + K * sizeof(OrbKeyPoint), (size_t)f2->n * sizeof(OrbKeyPoint),
                                hipMemcpyHostToDevice, m->stream));
            MCHK(hipMemcpyAsync(m->d_desc + K * 32, m->h_up + m->off_desc + K * 32, (size_t)f2->n * 32, hipMemcpyHostToDevice, m->stream));
        }
        if (prev_matched_xy)
            MCHK(hipMemcpyAsync(m->d_prev, m->h_up + m->off_prev, (size_t)f1->n * 8, hipMemcpyHostToDevice, m->stream));
    }
    MatchArgs A{};
    A.kps = m->d_kps;
    A.desc = m->d_desc;
    A.counts = m->d_counts;
    A.cap = m->max_kp;
    A.pair_a = nullptr;
    A.pair_b = nullptr;
    A.prev_xy = prev_matched_xy ? m->d_prev : nullptr;
    A.prev_out = prev_matched_xy ? m->d_prev_out : nullptr;
    A.min_x = f2->min_x;
    A.max_x = f2->max_x;
    A.min_y = f2->min_y;
    A.max_y = f2->max_y;
    A.window = window_size;
    A.nnratio = nnratio;
    A.check_ori = check_orientation;
    A.matches12 = m->d_m12;
    A.nmatches = m->d_nm;
    A.bin_scratch = m->d_bins;
    A.knn = m->d_knn;
    int rc = launch_match(A, 1, m->stream);
    if (rc) return rc;
    MCHK(hipMemcpyAsync(m->h_down + 64, m->d_m12, (size_t)f1->n * 4, hipMemcpyDeviceToHost, m->stream));
    MCHK(hipMemcpyAsync(m->h_down, m->d_nm, 4, hipMemcpyDeviceToHost, m->stream));
    if (prev_matched_xy)
        MCHK(hipMemcpyAsync(m->h_down + 64 + K * 4, m->d_prev_out, (size_t)f1->n * 8, hipMemcpyDeviceToHost, m->stream));
    MCHK(hipStreamSynchronize(m->stream));
    memcpy(matches12, m->h_down + 64, (size_t)f1->n * 4);
    memcpy(n_matches, m->h_down, 4);
    if (prev_matched_xy) memcpy(prev_matched_xy, m->h_down + 64 + K * 4, (size_t)f1->n * 8);
    return ORB_OK;
}

extern "C" int orb_features_in_area(OrbMatcher *m, const OrbFrameView *f, const float *queries_xyr, const int32_t *query_levels,
                                    int n_queries, int32_t *out_idx, int cap, int32_t *out_count) {
    if (!m || !f || f->n < 0 || n_queries < 0 || cap < 0 || (n_queries > 0 && (!queries_xyr || !out_count)) ||
        (n_queries > 0 && cap > 0 && !out_idx) || (f->n > 0 && !f->kps))
        return ORB_E_INVALID;
    if (f->n > m->max_kp || f->n > 16384) { /* the candidate keys of one query live in (static-limit) LDS: 4 B per keypoint */
        m_err = "orb_features_in_area: more than 16384 keypoints (or than the matcher's capacity) in the frame";
        return ORB_E_CAPACITY;
    }
    if (n_queries == 0) return ORB_OK;
    MCHK(hipSetDevice(m->device));
    float *d_q = nullptr;
    int32_t *d_lv = nullptr, *d_out = nullptr, *d_cnt = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_q);
        (void)hipFree(d_lv);
        (void)hipFree(d_out);
        (void)hipFree(d_cnt);
    };
#define FCHK(e)                                  \
    do {                                         \
        hipError_t _e = (e);                     \
        if (_e != hipSuccess) {                  \
            m_err = hipGetErrorString(_e);       \
            cleanup();                           \
            return ORB_E_HIP;                    \
        }                                        \
    } while (0)
    FCHK(hipMalloc(&d_q, (size_t)n_queries * 12));
    FCHK(hipMalloc(&d_cnt, (size_t)n_queries * 4));
    FCHK(hipMalloc(&d_out, std::max<size_t>((size_t)n_queries * cap * 4, 4)));
    if (query_levels) {
        FCHK(hipMalloc(&d_lv, (size_t)n_queries * 8));
        FCHK(hipMemcpyAsync(d_lv, query_levels, (size_t)n_queries * 8, hipMemcpyHostToDevice, m->stream));
    }
    FCHK(hipMemcpyAsync(d_q, queries_xyr, (size_t)n_queries * 12, hipMemcpyHostToDevice, m->stream));
    if (f->n > 0) FCHK(hipMemcpyAsync(m->d_kps, f->kps, (size_t)f->n * sizeof(OrbKeyPoint), hipMemcpyHostToDevice, m->stream));
    FCHK(hipMemsetAsync(d_out, 0xFF, std::max<size_t>((size_t)n_queries * cap * 4, 4), m->stream));
    hipLaunchKernelGGL(k_features_in_area, dim3(n_queries), dim3(64), (size_t)std::max(f->n, 1) * 4, m->stream, m->d_kps, f->n,
                       f->min_x, f->max_x, f->min_y, f->max_y, d_q, d_lv, n_queries, d_out, cap, d_cnt);
    FCHK(hipGetLastError());
    if (cap > 0) FCHK(hipMemcpyAsync(out_idx, d_out, (size_t)n_queries * cap * 4, hipMemcpyDeviceToHost, m->stream));
    FCHK(hipMemcpyAsync(out_count, d_cnt, (size_t)n_queries * 4, hipMemcpyDeviceToHost, m->stream));
    FCHK(hipStreamSynchronize(m->stream));
#undef FCHK
    cleanup();
    return ORB_OK;
}


/* ---- SURVEY N4 on the device: the sequential walk and the rotation histogram of SearchByProjection ---------------------
 * ("as k_match_init does for a11".)  Upstream's loop is sequential in the last frame's features -- feature i
 * takes the closest candidate that is still free, and a feature taken by an earlier one is closed to later ones -- but,
 * as in SearchForInitialization, a query only READS the taken flags of the candidates it inspects and only WRITES the
 * state of the one feature it matches, so 64 queries are resolved in speculative rounds by ONE wavefront: every pending
 * lane decides from the current LDS state; a lane that TAKES its feature stamps it (atomicMin of the lane id); a lane is
 * valid iff no stamp in its read set is smaller than its own id; the valid prefix commits (cur_match by atomicMax of the
 * query index: "the last writer wins" of the sequential loop; taken flags of distinct features), the first invalid lane
 * opens the next round.  A query whose QK_TOP closest candidates are all taken is resolved alone by a wave-wide rescan of
 * the whole window (the same keys as the host walk: distance << 32 | grid cell << 16 | index).  The rotation histogram
 * (reference src/ORBmatcher.cpp:79-118 semantics, per ACCEPTANCE) closes the kernel: an acceptance in a bin outside the
 * three maxima clears its feature (match -1, taken 0) and takes one off the count.  Identical results to the host walk
 * (tests/test_gpu_n4.py runs both).  One wavefront per launch: for the host-pointer API the host walk is FASTER (10-20 us
 * against ~70 rounds here); this kernel is for callers whose frames stay on the device (orb_search_by_projection_device). */
struct N4WalkArgs {
    const unsigned long long *lists; /* [nq][QK_TOP] from k_query_topk */
    const uint8_t *flags;            /* [nq] bit0: holds a map point, bit1: the point has observations (takes the feature) */
    const float *qxyr;               /* [nq][3] from k_project_queries (r > 0: a valid query) */
    const int32_t *qlvl;             /* [nq][2] */
    const OrbKeyPoint *kps2;         /* current frame: candidates */
    const uint8_t *desc2;
    int n2;
    const int32_t *n2_dev;           /* NULL, or the current frame's count in device memory (n2 is then the bound the LDS layout uses) */
    const OrbKeyPoint *kps1;         /* last frame: queries */
    const uint8_t *desc1;
    int nq;
    float min_x, max_x, min_y, max_y;
    int th_high, check_ori;
    uint8_t *taken;     /* [n2] in / out */
    int32_t *match;     /* [n2] out */
    int32_t *n_matches; /* out */
};
__global__ __launch_bounds__(64) void k_n4_walk(N4WalkArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char n4_smem[];
    const int lane = threadIdx.x, nq = A.nq;
    const int n2 = A.n2_dev ? min(*A.n2_dev, A.n2) : A.n2;
    int32_t *s_stamp = (int32_t *)n4_smem;     /* [n2] (laid out for the bound A.n2) */
    int32_t *s_match = s_stamp + A.n2;         /* [n2] */
    int32_t *s_feat = s_match + A.n2;          /* [nq] accepted feature of query i, or -1 */
    uint8_t *s_taken = (uint8_t *)(s_feat + nq); /* [n2] */
    uint8_t *s_bin = s_taken + A.n2;           /* [nq] */
    __shared__ int s_hist[HISTO_LENGTH];
    __shared__ int s_acc;
    for (int j = lane; j < n2; j += 64) {
        s_stamp[j] = 0x7FFFFFFF;
        s_match[j] = -1;
        s_taken[j] = A.taken[j];
    }
    for (int i = lane; i < nq; i += 64) {
        s_feat[i] = -1;
        s_bin[i] = 0xFF;
    }
    if (lane < HISTO_LENGTH) s_hist[lane] = 0;
    if (lane == 0) s_acc = 0;
    __syncthreads();
    const float invW = (float)GRID_COLS / (A.max_x - A.min_x), invH = (float)GRID_ROWS / (A.max_y - A.min_y);
    const float factor = HISTO_LENGTH / 360.0f;
    for (int base = 0; base < nq; base += 64) {
        const int i = base + lane;
        uint64_t c[QK_TOP];
#pragma unroll
        for (int k = 0; k < QK_TOP; k++) c[k] = ~0ull;
        int fl = 0;
        float ang1 = 0.f;
        bool valid = false;
        if (i < nq) {
            fl = A.flags[i];
            valid = (fl & 1) && A.qxyr[3 * i + 2] > 0.f;
            if (valid) {
#pragma unroll
                for (int k = 0; k < QK_TOP; k++) c[k] = A.lists[(size_t)i * QK_TOP + k];
                ang1 = A.kps1[i].angle;
            }
        }
        const int nchunk = min(64, nq - base);
        int done = 0;
        while (done < nchunk) {
            const bool act = valid && lane >= done && lane < nchunk;
            int best = -1, bestDist = 256, real = 0, n_insp = 0;
            bool stop = false;
#pragma unroll
            for (int k = 0; k < QK_TOP; k++) {
                if (!stop) {
                    if (c[k] == ~0ull) {
                        stop = true;
                    } else {
                        real++;
                        n_insp = k + 1;
                        const int j = (int)(c[k] & 0xFFFFu);
                        if (!s_taken[j]) {
                            best = j;
                            bestDist = (int)(c[k] >> 32);
                            stop = true;
                        }
                    }
                }
            }
            const bool slow = act && best < 0 && real == QK_TOP; /* every listed candidate taken: whole-window rescan, alone */
            const bool accept = act && !slow && best >= 0 && bestDist <= A.th_high;
            const bool takes = accept && (fl & 2);
            if (takes) atomicMin(&s_stamp[best], lane);
            __syncthreads();
            bool conflict = false;
            if (act && !slow) {
#pragma unroll
                for (int k = 0; k < QK_TOP; k++)
                    if (k < n_insp) conflict = conflict || s_stamp[(int)(c[k] & 0xFFFFu)] < lane;
            }
            const uint64_t badm = __ballot(act && (conflict || slow));
            const int first_bad = badm ? (int)__builtin_ctzll(badm) : 64;
            __syncthreads();
            if (takes) s_stamp[best] = 0x7FFFFFFF; /* every stamp of this round is cleared again */
            if (accept && lane < first_bad) {
                atomicMax(&s_match[best], i); /* cur_match[best] = i: within a round the largest query index is the last writer */
                if (takes) s_taken[best] = 1;
                s_feat[i] = best;
                atomicAdd(&s_acc, 1);
                if (A.check_ori) {
                    float rot = ang1 - A.kps2[best].angle;
                    if (rot < 0.0f) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    s_bin[i] = (uint8_t)bin;
                    atomicAdd(&s_hist[bin], 1);
                }
            }
            __syncthreads();
            done = min(first_bad, nchunk);
            if (done < nchunk && ((__ballot(slow) >> done) & 1ull)) {
                const int t = done, it = base + t;
                const float x = A.qxyr[3 * it], y = A.qxyr[3 * it + 1], r = A.qxyr[3 * it + 2];
                const int lo = A.qlvl[2 * it], hi = A.qlvl[2 * it + 1];
                uint32_t a[8];
                {
                    const uint4 *p = (const uint4 *)(A.desc1 + (size_t)it * 32);
                    const uint4 v0 = p[0], v1 = p[1];
                    a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
                    a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
                }
                uint64_t bestKey = ~0ull;
                for (int j = lane; j < n2; j += 64) {
                    if (s_taken[j]) continue;
                    const OrbKeyPoint kp = A.kps2[j];
                    const int ix = (int)round(((double)kp.x - (double)A.min_x) * (double)invW);
                    const int iy = (int)round(((double)kp.y - (double)A.min_y) * (double)invH);
                    if (ix < 0 || ix >= GRID_COLS || iy < 0 || iy >= GRID_ROWS) continue;
                    if (kp.octave < lo || (hi >= 0 && kp.octave > hi)) continue;
                    if (!(fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)) continue;
                    const uint4 *p = (const uint4 *)(A.desc2 + (size_t)j * 32);
                    const int d = hamming256(a, p[0], p[1]);
                    const uint64_t key = ((uint64_t)(uint32_t)d << 32) | ((uint64_t)(uint32_t)(ix * GRID_ROWS + iy) << 16) | (uint32_t)j;
                    if (key < bestKey) bestKey = key;
                }
                const uint64_t sKey = wave_min_u64(bestKey);
                if (sKey != ~0ull && (int)(sKey >> 32) <= A.th_high && lane == 0) {
                    const int b = (int)(sKey & 0xFFFFu);
                    s_match[b] = it; /* (later than every committed query) */
                    if (A.flags[it] & 2) s_taken[b] = 1;
                    s_feat[it] = b;
                    s_acc++;
                    if (A.check_ori) {
                        float rot = A.kps1[it].angle - A.kps2[b].angle;
                        if (rot < 0.0f) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        s_bin[it] = (uint8_t)bin;
                        s_hist[bin]++;
                    }
                }
                __syncthreads();
                done++;
            }
        }
    }
    __syncthreads();
    int removed = 0;
    if (A.check_ori) { /* ComputeThreeMaxima (reference src/ORBmatcher.cpp:147-188), then the per-acceptance filter (:95-118) */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int max1 = 0, max2 = 0, max3 = 0;
        for (int b = 0; b < HISTO_LENGTH; b++) {
            const int sz = s_hist[b];
            if (sz > max1) {
                max3 = max2; max2 = max1; max1 = sz;
                ind3 = ind2; ind2 = ind1; ind1 = b;
            } else if (sz > max2) {
                max3 = max2; max2 = sz;
                ind3 = ind2; ind2 = b;
            } else if (sz > max3) {
                max3 = sz;
                ind3 = b;
            }
        }
        if (max2 < 0.1f * (float)max1) {
            ind2 = -1;
            ind3 = -1;
        } else if (max3 < 0.1f * (float)max1) {
            ind3 = -1;
        }
        for (int i0 = 0; i0 < nq; i0 += 64) {
            const int i = i0 + lane;
            bool bad = false;
            if (i < nq) {
                const int f = s_feat[i], b = s_bin[i];
                bad = f >= 0 && b != ind1 && b != ind2 && b != ind3;
                if (bad) {
                    s_match[f] = -1;
                    s_taken[f] = 0;
                }
            }
            removed += __popcll(__ballot(bad));
        }
        __syncthreads();
    }
    for (int j = lane; j < n2; j += 64) {
        A.match[j] = s_match[j];
        A.taken[j] = s_taken[j];
    }
    if (lane == 0) *A.n_matches = s_acc - removed;
}
static size_t n4_walk_lds(int n2, int nq) { return (size_t)n2 * 9 + (size_t)nq * 5 + 64; }

/* ---- SURVEY N4 host side ---------------------------------------------------------------------------------------- */
static int n4_scratch(OrbMatcher *m, size_t bytes) {
    if (bytes <= m->cap_scratch) return ORB_OK;
    if (m->d_scratch) MCHK(hipFree(m->d_scratch));
    m->d_scratch = nullptr;
    m->cap_scratch = 0;
    MCHK(hipMalloc((void **)&m->d_scratch, bytes));
    m->cap_scratch = bytes;
    return ORB_OK;
}
static int n4_pinned(OrbMatcher *m, size_t bytes) {
    if (bytes <= m->cap_h_n4) return ORB_OK;
    if (m->h_n4) MCHK(hipHostFree(m->h_n4));
    m->h_n4 = nullptr;
    m->cap_h_n4 = 0;
    MCHK(hipHostMalloc((void **)&m->h_n4, bytes + bytes / 4));
    m->cap_h_n4 = bytes + bytes / 4;
    return ORB_OK;
}
static inline int host_hamming(const uint8_t *a, const uint8_t *b) {
    int d = 0;
    for (int k = 0; k < 4; k++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * k, 8);
        memcpy(&y, b + 8 * k, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}
/* rotation-consistency filter of the reference (src/ORBmatcher.cpp:79-118) over the list of acceptances (bin, feature)
 * in the order they were made: an entry of a bin outside the three maxima clears its feature and takes one off the
 * count -- per ENTRY, as upstream's rotHist does: a feature that was given a map point twice (the first point had no
 * observations) has two entries */
static int n4_histogram_filter(const std::vector<std::pair<int, int>> &acc, int32_t *match, uint8_t *taken_or_null) {
    int32_t sizes[HISTO_LENGTH] = {0};
    for (const auto &e : acc) sizes[e.first]++;
    int32_t i1 = -1, i2 = -1, i3 = -1;
    orb_compute_three_maxima(sizes, HISTO_LENGTH, &i1, &i2, &i3);
    int removed = 0;
    for (const auto &e : acc)
        if (e.first != i1 && e.first != i2 && e.first != i3) {
            match[e.second] = -1;
            if (taken_or_null) taken_or_null[e.second] = 0;
            removed++;
        }
    return removed;
}
static inline int n4_bin(float a1, float a2) {
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)roundf(rot * (HISTO_LENGTH / 360.0f));
    return bin == HISTO_LENGTH ? 0 : bin;
}

extern "C" int orb_search_by_projection(OrbMatcher *m, const OrbFrameView *cur, const OrbFrameView *last, const uint8_t *last_flags,
                                        const double *last_xw, const double *Rcw9, const double *tcw3, double fx, double fy,
                                        double cx, double cy, const float *scale_factors, int nlevels, uint8_t *cur_taken,
                                        float th, int check_orientation, int32_t *cur_match, int32_t *n_matches) {
    if (!m || !cur || !last || !n_matches || cur->n < 0 || last->n < 0 || nlevels <= 0 || nlevels > 64 || !scale_factors || !Rcw9 ||
        !tcw3 || (last->n > 0 && (!last_flags || !last_xw || !last->kps || !last->desc)) ||
        (cur->n > 0 && (!cur_taken || !cur_match || !cur->kps || !cur->desc)))
        return ORB_E_INVALID;
    if (cur->n > m->max_kp || last->n > m->max_kp) return ORB_E_CAPACITY;
    *n_matches = 0;
    for (int j = 0; j < cur->n; j++) cur_match[j] = -1;
    if (cur->n == 0 || last->n == 0) return ORB_OK;
    MCHK(hipSetDevice(m->device));
    const int nq = last->n, n2 = cur->n, TH_HIGH = 100;
    /* device scratch: [ inputs: flags | xw | pose | scale factors | cur kps | cur desc | last kps | last desc ]
                       [ outputs: qxyr | qlvl | candidate lists ]
       The inputs are packed into ONE pinned block and go up as one copy, the outputs come down as one copy into pinned
       memory: eight uploads from pageable memory and three downloads into std::vectors (each a blocking bounce through the
       runtime's staging buffer) were most of the call -- Tracking makes it once or twice per frame (src/Tracking.cpp:344,348) */
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_fl = 0, o_xw = al(o_fl + nq), o_pose = al(o_xw + (size_t)nq * 24), o_sf = al(o_pose + 96),
                 o_k2 = al(o_sf + (size_t)nlevels * 4), o_d2 = al(o_k2 + (size_t)n2 * sizeof(OrbKeyPoint)),
                 o_k1 = al(o_d2 + (size_t)n2 * 32), o_d1 = al(o_k1 + (size_t)nq * sizeof(OrbKeyPoint)), in_bytes = al(o_d1 + (size_t)nq * 32),
                 o_q = in_bytes, o_lv = al(o_q + (size_t)nq * 12), o_out = al(o_lv + (size_t)nq * 8),
                 total = al(o_out + (size_t)nq * QK_TOP * 8), out_bytes = total - o_q,
                 o_tk = total, o_mt = al(o_tk + n2), o_nm = al(o_mt + (size_t)n2 * 4), total2 = al(o_nm + 4); /* (device walk: taken | match | count) */
    int rc;
    if ((rc = n4_scratch(m, total2))) return rc;
    if ((rc = n4_pinned(m, total2))) return rc;
    uint8_t *S = m->d_scratch, *P = m->h_n4;
    memcpy(P + o_fl, last_flags, nq);
    memcpy(P + o_xw, last_xw, (size_t)nq * 24);
    memcpy(P + o_k2, cur->kps, (size_t)n2 * sizeof(OrbKeyPoint));
    memcpy(P + o_d2, cur->desc, (size_t)n2 * 32);
    memcpy(P + o_k1, last->kps, (size_t)nq * sizeof(OrbKeyPoint));
    memcpy(P + o_d1, last->desc, (size_t)nq * 32);
    MCHK(hipMemcpyAsync(S, P, in_bytes, hipMemcpyHostToDevice, m->stream));
    N4Pose PP;
    memcpy(PP.pose12, Rcw9, 72);
    memcpy(PP.pose12 + 9, tcw3, 24);
    for (int l = 0; l < 64; l++) PP.scale_factors[l] = l < nlevels ? scale_factors[l] : 0.f;
    hipLaunchKernelGGL(k_project_queries, dim3((nq + 255) / 256), dim3(256), 0, m->stream, (const OrbKeyPoint *)(S + o_k1), S + o_fl,
                       (const double *)(S + o_xw), nq, PP, fx, fy, cx, cy, cur->min_x, cur->max_x, cur->min_y, cur->max_y, nlevels, th,
                       (float *)(S + o_q), (int32_t *)(S + o_lv));
    QueryArgs A{};
    A.kps2 = (const OrbKeyPoint *)(S + o_k2);
    A.desc2 = S + o_d2;
    A.n2 = n2;
    A.qdesc = S + o_d1;
    A.qxyr = (const float *)(S + o_q);
    A.qlvl = (const int32_t *)(S + o_lv);
    A.nq = nq;
    A.min_x = cur->min_x; A.max_x = cur->max_x; A.min_y = cur->min_y; A.max_y = cur->max_y;
    A.keep_max_d = TH_HIGH;
    A.out = (unsigned long long *)(S + o_out);
    hipLaunchKernelGGL(k_query_topk, dim3((nq + 256 / QK_SLICES - 1) / (256 / QK_SLICES)), dim3(256), 0, m->stream, A);
    MCHK(hipGetLastError());
    if (getenv("ORBSLAM_AMD_N4_DEVICE_WALK") && n4_walk_lds(n2, nq) <= 120 * 1024) {
        /* the walk and the rotation histogram on the device too (k_n4_walk) -- what orb_search_by_projection_device runs;
           switchable here so that the tests hold it against the host walk below and the oracle */
        memcpy(P + o_tk, cur_taken, n2);
        MCHK(hipMemcpyAsync(S + o_tk, P + o_tk, n2, hipMemcpyHostToDevice, m->stream));
        N4WalkArgs WA{};
        WA.lists = (const unsigned long long *)(S + o_out);
        WA.flags = S + o_fl;
        WA.qxyr = (const float *)(S + o_q);
        WA.qlvl = (const int32_t *)(S + o_lv);
        WA.kps2 = (const OrbKeyPoint *)(S + o_k2);
        WA.desc2 = S + o_d2;
        WA.n2 = n2;
        WA.kps1 = (const OrbKeyPoint *)(S + o_k1);
        WA.desc1 = S + o_d1;
        WA.nq = nq;
        WA.min_x = cur->min_x; WA.max_x = cur->max_x; WA.min_y = cur->min_y; WA.max_y = cur->max_y;
        WA.th_high = TH_HIGH;
        WA.check_ori = check_orientation;
        WA.taken = S + o_tk;
        WA.match = (int32_t *)(S + o_mt);
        WA.n_matches = (int32_t *)(S + o_nm);
        static PerDeviceLdsLimit walk_lim;
        MCHK(walk_lim.ensure((const void *)k_n4_walk, n4_walk_lds(n2, nq)));
        hipLaunchKernelGGL(k_n4_walk, dim3(1), dim3(64), n4_walk_lds(n2, nq), m->stream, WA);
        MCHK(hipGetLastError());
        MCHK(hipMemcpyAsync(P + o_tk, S + o_tk, total2 - o_tk, hipMemcpyDeviceToHost, m->stream));
        MCHK(hipStreamSynchronize(m->stream));
        memcpy(cur_taken, P + o_tk, n2);
        memcpy(cur_match, P + o_mt, (size_t)n2 * 4);
        memcpy(n_matches, P + o_nm, 4);
        return ORB_OK;
    }
    MCHK(hipMemcpyAsync(P + o_q, S + o_q, out_bytes, hipMemcpyDeviceToHost, m->stream));
    MCHK(hipStreamSynchronize(m->stream));
    const unsigned long long *lists = (const unsigned long long *)(P + o_out);
    const float *qxyr = (const float *)(P + o_q);
    const int32_t *qlvl = (const int32_t *)(P + o_lv);
    /* the sequential walk: feature i takes the closest candidate that is still free */
    const float invW = (float)GRID_COLS / (cur->max_x - cur->min_x), invH = (float)GRID_ROWS / (cur->max_y - cur->min_y);
    std::vector<std::pair<int, int>> acc; /* (rotation bin, feature) of every acceptance */
    int nm = 0;
    for (int i = 0; i < nq; i++) {
        if (!(last_flags[i] & 1) || !(qxyr[3 * i + 2] > 0.f)) continue;
        const unsigned long long *L = &lists[(size_t)i * QK_TOP];
        int best = -1, bestDist = 256, real = 0;
        for (int k = 0; k < QK_TOP && best < 0; k++) {
            if (L[k] == ~0ull) break;
            real++;
            const int j = (int)(L[k] & 0xFFFFu);
            if (!cur_taken[j]) {
                best = j;
                bestDist = (int)(L[k] >> 32);
            }
        }
        if (best < 0 && real == QK_TOP) {
            /* all QK_TOP closest candidates were taken by earlier features: look at the whole window (rare) */
            const float x = qxyr[3 * i], y = qxyr[3 * i + 1], r = qxyr[3 * i + 2];
            const int lo = qlvl[2 * i], hi = qlvl[2 * i + 1];
            unsigned long long bestKey = ~0ull;
            for (int j = 0; j < n2; j++) {
                if (cur_taken[j]) continue;
                const OrbKeyPoint &kp = cur->kps[j];
                const int ix = (int)round(((double)kp.x - (double)cur->min_x) * (double)invW);
                const int iy = (int)round(((double)kp.y - (double)cur->min_y) * (double)invH);
                if (ix < 0 || ix >= GRID_COLS || iy < 0 || iy >= GRID_ROWS) continue;
                if (kp.octave < lo || (hi >= 0 && kp.octave > hi)) continue;
                if (!(fabsf(kp.x - x) < r && fabsf(kp.y - y) < r)) continue;
                const int d = host_hamming(last->desc + (size_t)i * 32, cur->desc + (size_t)j * 32);
                const unsigned long long key = ((unsigned long long)d << 32) | ((unsigned long long)(ix * GRID_ROWS + iy) << 16) | (unsigned)j;
                if (key < bestKey) bestKey = key;
            }
            if (bestKey != ~0ull) {
                best = (int)(bestKey & 0xFFFFu);
                bestDist = (int)(bestKey >> 32);
            }
        }
        if (best >= 0 && bestDist <= TH_HIGH) {
            cur_match[best] = i;
            if (last_flags[i] & 2) cur_taken[best] = 1;
            nm++;
            if (check_orientation) acc.emplace_back(n4_bin(last->kps[i].angle, cur->kps[best].angle), best);
        }
    }
    if (check_orientation) nm -= n4_histogram_filter(acc, cur_match, cur_taken);
    *n_matches = nm;
    return ORB_OK;
}

/* SearchByProjection on DEVICE-RESIDENT frames (SURVEY N4 / VERDICT r2 "no host round trip"): the current and the last
 * frame's feature slabs as an extractor left them in HBM (orb_extract_batch_device), the last frame's map-point flags and
 * world positions in device memory, results (cur_match, cur_taken, the count) in device memory; projection, candidate
 * scan, sequential walk and rotation histogram are three kernels on `stream`, nothing is copied and nothing waits.  The
 * scratch of the matcher handle holds the query parameters and candidate lists: one search in flight per handle. */
extern "C" int orb_search_by_projection_device(OrbMatcher *m, const OrbKeyPoint *d_cur_kps, const uint8_t *d_cur_desc, int n_cur,
                                               const int32_t *d_n_cur, const OrbKeyPoint *d_last_kps, const uint8_t *d_last_desc, int n_last,
                                               const uint8_t *d_last_flags, const double *d_last_xw, const double *Rcw9,
                                               const double *tcw3, double fx, double fy, double cx, double cy,
                                               const float *scale_factors, int nlevels, float min_x, float max_x, float min_y,
                                               float max_y, uint8_t *d_cur_taken, float th, int check_orientation,
                                               int32_t *d_cur_match, int32_t *d_n_matches, void *stream) {
    if (!m || n_cur <= 0 || n_last <= 0 || nlevels <= 0 || nlevels > 64 || !scale_factors || !Rcw9 || !tcw3 || !d_cur_kps ||
        !d_cur_desc || !d_last_kps || !d_last_desc || !d_last_flags || !d_last_xw || !d_cur_taken || !d_cur_match || !d_n_matches)
        return ORB_E_INVALID;
    if (n_cur > m->max_kp || n_last > m->max_kp || n4_walk_lds(n_cur, n_last) > 120 * 1024) return ORB_E_CAPACITY;
    MCHK(hipSetDevice(m->device));
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    const int nq = n_last, n2 = n_cur;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_q = 0, o_lv = al(o_q + (size_t)nq * 12), o_out = al(o_lv + (size_t)nq * 8), total = al(o_out + (size_t)nq * QK_TOP * 8);
    int rc;
    if (total > m->cap_scratch) { /* (growing the scratch frees the old block: not under a search that may still be running) */
        MCHK(hipStreamSynchronize(st));
        MCHK(hipStreamSynchronize(m->stream));
        if ((rc = n4_scratch(m, total))) return rc;
    }
    uint8_t *S = m->d_scratch;
    N4Pose PP;
    memcpy(PP.pose12, Rcw9, 72);
    memcpy(PP.pose12 + 9, tcw3, 24);
    for (int l = 0; l < 64; l++) PP.scale_factors[l] = l < nlevels ? scale_factors[l] : 0.f;
    hipLaunchKernelGGL(k_project_queries, dim3((nq + 255) / 256), dim3(256), 0, st, d_last_kps, d_last_flags, d_last_xw, nq, PP, fx, fy,
                       cx, cy, min_x, max_x, min_y, max_y, nlevels, th, (float *)(S + o_q), (int32_t *)(S + o_lv));
    QueryArgs A{};
    A.kps2 = d_cur_kps;
    A.desc2 = d_cur_desc;
    A.n2 = n2;
    A.n2_dev = d_n_cur;
    A.qdesc = d_last_desc;
    A.qxyr = (const float *)(S + o_q);
    A.qlvl = (const int32_t *)(S + o_lv);
    A.nq = nq;
    A.min_x = min_x; A.max_x = max_x; A.min_y = min_y; A.max_y = max_y;
    A.keep_max_d = 100; /* TH_HIGH */
    A.out = (unsigned long long *)(S + o_out);
    hipLaunchKernelGGL(k_query_topk, dim3((nq + 256 / QK_SLICES - 1) / (256 / QK_SLICES)), dim3(256), 0, st, A);
    N4WalkArgs WA{};
    WA.lists = (const unsigned long long *)(S + o_out);
    WA.flags = d_last_flags;
    WA.qxyr = (const float *)(S + o_q);
    WA.qlvl = (const int32_t *)(S + o_lv);
    WA.kps2 = d_cur_kps;
    WA.desc2 = d_cur_desc;
    WA.n2 = n2;
    WA.n2_dev = d_n_cur;
    WA.kps1 = d_last_kps;
    WA.desc1 = d_last_desc;
    WA.nq = nq;
    WA.min_x = min_x; WA.max_x = max_x; WA.min_y = min_y; WA.max_y = max_y;
    WA.th_high = 100;
    WA.check_ori = check_orientation;
    WA.taken = d_cur_taken;
    WA.match = d_cur_match;
    WA.n_matches = d_n_matches;
    static PerDeviceLdsLimit walk_lim;
    MCHK(walk_lim.ensure((const void *)k_n4_walk, n4_walk_lds(n2, nq)));
    hipLaunchKernelGGL(k_n4_walk, dim3(1), dim3(64), n4_walk_lds(n2, nq), st, WA);
    MCHK(hipGetLastError());
    return ORB_OK;
}

extern "C" int orb_search_by_bow(OrbMatcher *m, const OrbFrameView *kf, const uint8_t *kf_valid, const OrbFrameView *f, float nnratio,
                                 int check_orientation, int32_t *f_match, int32_t *n_matches) {
    if (!m || !kf || !f || !n_matches || kf->n < 0 || f->n < 0 || (kf->n > 0 && (!kf_valid || !kf->kps || !kf->desc)) ||
        (f->n > 0 && (!f_match || !f->kps || !f->desc)))
        return ORB_E_INVALID;
    if (kf->n > m->max_kp || f->n > m->max_kp) return ORB_E_CAPACITY;
    *n_matches = 0;
    for (int j = 0; j < f->n; j++) f_match[j] = -1;
    if (kf->n == 0 || f->n == 0) return ORB_OK;
    MCHK(hipSetDevice(m->device));
    const int nq = kf->n, n2 = f->n;
    /* a candidate beyond TH_LOW matters only as second best, and only while TH_LOW >= d * ratio (cf. k_match_knn) */
    int keep = TH_LOW;
    while (keep < 256 && !((float)TH_LOW < (float)(keep + 1) * nnratio)) keep++;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_q = 0, o_out = al((size_t)nq * 12), total = al(o_out + (size_t)nq * QK_TOP * 8);
    int rc;
    if ((rc = n4_scratch(m, total))) return rc;
    uint8_t *S = m->d_scratch;
    std::vector<float> qxyr((size_t)nq * 3);
    for (int i = 0; i < nq; i++) {
        qxyr[3 * i] = qxyr[3 * i + 1] = 0.f;
        qxyr[3 * i + 2] = kf_valid[i] ? -1.f : 0.f; /* r < 0: every frame feature is a candidate; r = 0: none */
    }
    MCHK(hipMemcpyAsync(S + o_q, qxyr.data(), qxyr.size() * 4, hipMemcpyHostToDevice, m->stream));
    MCHK(hipMemcpyAsync(m->d_kps, f->kps, (size_t)n2 * sizeof(OrbKeyPoint), hipMemcpyHostToDevice, m->stream));
    MCHK(hipMemcpyAsync(m->d_desc, f->desc, (size_t)n2 * 32, hipMemcpyHostToDevice, m->stream));
    MCHK(hipMemcpyAsync(m->d_desc + (size_t)m->max_kp * 32, kf->desc, (size_t)nq * 32, hipMemcpyHostToDevice, m->stream));
    QueryArgs A{};
    A.kps2 = m->d_kps;
    A.desc2 = m->d_desc;
    A.n2 = n2;
    A.qdesc = m->d_desc + (size_t)m->max_kp * 32;
    A.qxyr = (const float *)(S + o_q);
    A.qlvl = nullptr;
    A.nq = nq;
    A.min_x = f->min_x; A.max_x = f->max_x; A.min_y = f->min_y; A.max_y = f->max_y;
    A.keep_max_d = keep;
    A.out = (unsigned long long *)(S + o_out);
    hipLaunchKernelGGL(k_query_topk, dim3((nq + 256 / QK_SLICES - 1) / (256 / QK_SLICES)), dim3(256), 0, m->stream, A);
    MCHK(hipGetLastError());
    std::vector<unsigned long long> lists((size_t)nq * QK_TOP);
    MCHK(hipMemcpyAsync(lists.data(), S + o_out, lists.size() * 8, hipMemcpyDeviceToHost, m->stream));
    MCHK(hipStreamSynchronize(m->stream));
    std::vector<std::pair<int, int>> acc; /* (rotation bin, feature) of every acceptance */
    int nm = 0;
    for (int i = 0; i < nq; i++) {
        if (!kf_valid[i]) continue;
        const unsigned long long *L = &lists[(size_t)i * QK_TOP];
        int best = -1, d1 = 256, d2 = 256, real = 0, found = 0;
        for (int k = 0; k < QK_TOP && found < 2; k++) {
            if (L[k] == ~0ull) break;
            real++;
            const int j = (int)(L[k] & 0xFFFFFFFFu);
            if (f_match[j] >= 0) continue;
            if (found == 0) {
                best = j;
                d1 = (int)(L[k] >> 32);
            } else
                d2 = (int)(L[k] >> 32);
            found++;
        }
        if (found < 2 && real == QK_TOP) { /* the list ran out before two free features turned up: scan them all (rare) */
            best = -1;
            d1 = d2 = 256;
            for (int j = 0; j < n2; j++) {
                if (f_match[j] >= 0) continue;
                const int d = host_hamming(kf->desc + (size_t)i * 32, f->desc + (size_t)j * 32);
                if (d < d1) {
                    d2 = d1;
                    d1 = d;
                    best = j;
                } else if (d < d2)
                    d2 = d;
            }
        }
        if (best >= 0 && d1 <= TH_LOW && (float)d1 < nnratio * (float)d2) {
            f_match[best] = i;
            nm++;
            if (check_orientation) acc.emplace_back(n4_bin(kf->kps[i].angle, f->kps[best].angle), best);
        }
    }
    if (check_orientation) nm -= n4_histogram_filter(acc, f_match, nullptr);
    *n_matches = nm;
    return ORB_OK;
}

extern "C" int orb_match_batch_device(OrbMatcher *m, const OrbKeyPoint *d_kps, const uint8_t *d_desc,
                                      const int32_t *d_counts, int cap, const int32_t *d_pair_a, const int32_t *d_pair_b,
                                      int n_pairs, float min_x, float max_x, float min_y, float max_y, int window_size,
                                      float nnratio, int check_orientation, int32_t *d_matches12, int32_t *d_nmatches,
                                      void *stream) {
    if (!m || !d_kps || !d_desc || !d_counts || !d_pair_a || !d_pair_b || !d_matches12 || !d_nmatches || n_pairs <= 0 ||
        cap <= 0)
        return ORB_E_INVALID;
    if (n_pairs > m->max_pairs || cap > m->max_kp) return ORB_E_CAPACITY;
    MCHK(hipSetDevice(m->device));
    MatchArgs A{};
    A.kps = d_kps;
    A.desc = d_desc;
    A.counts = d_counts;
    A.cap = cap;
    A.pair_a = d_pair_a;
    A.pair_b = d_pair_b;
    A.prev_xy = nullptr;
    A.prev_out = nullptr;
    A.min_x = min_x;
    A.max_x = max_x;
    A.min_y = min_y;
    A.max_y = max_y;
    A.window = window_size;
    A.nnratio = nnratio;
    A.check_ori = check_orientation;
    A.matches12 = d_matches12;
    A.nmatches = d_nmatches;
    A.bin_scratch = m->d_bins;
    A.knn = m->d_knn;
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    hipEvent_t *ev = m->ev[m->prof_calls % OrbMatcher::kProfRing];
    if (m->profile) MCHK(hipEventRecord(ev[0], st));
    int rc = launch_match(A, n_pairs, st, m->profile ? ev[2] : nullptr);
    if (m->profile) {
        MCHK(hipEventRecord(ev[1], st));
        m->prof_calls++;
    }
    return rc;
}
