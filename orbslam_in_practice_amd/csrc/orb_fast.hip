/*
 * orb_fast.hip -- per-cell FAST-9/16 corner scoring, 3x3 non-max suppression and the two-threshold
 * fallback of ORBextractor::ComputeKeyPointsOctTree (reference src/ORBextractor.cpp:745-786), all
 * cells of all levels of all frames in ONE launch: one WAVEFRONT per (cell, frame)
 * (k_fast_cells_wave; k_fast_cells, one 256-thread workgroup per cell, remains for geometries whose
 * cells exceed the wave kernel's tile bookkeeping).
 *
 *   - the cell ROI (<= 65x65 u8, incl. the 3-px rim cv::FAST ignores) is staged in LDS with
 *     row-coalesced dword loads;
 *   - "corner at threshold t"  <=>  cornerScore >= t (the score is the largest threshold at which the
 *     pixel is still a 9/16 corner), so the detection zone is first filtered at minThFAST -- a compass
 *     pre-test, then the full sign-bit arc test -- and only the survivors (queued in row-major order in
 *     LDS) get their exact cornerScore; ONE scoring pass serves both iniThFAST and minThFAST;
 *   - NMS keeps a pixel iff its score is strictly greater than its 8 neighbours INSIDE the cell's
 *     detection zone (scores outside, and of non-corners, are 0, exactly like cv::FAST on the cell ROI);
 *   - if no pixel survives at iniThFAST the cell falls back to minThFAST (per-cell decision);
 *   - survivors are emitted in the reference's order (row-major inside the cell) into the cell's slab
 *     slot with a wavefront ballot/popcount compaction -- no atomics, deterministic order.
 *
 * NMS on scores >= minTh followed by filtering at iniTh equals cv::FAST(iniTh, nms=true): a neighbour
 * whose score is below iniTh can never beat a pixel whose score is >= iniTh.
 */
#include "orb_common.h"
#include <cstdlib>

#define FT_LD 68 /* LDS row pitch */

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned short v2u __attribute__((ext_vector_type(2)));

/* cornerScore of one pixel.  Both ring polarities ride in one packed register: element k holds
 * (d[k], -d[k]) with d = v - ring[k]; the max over the 16 nine-arcs of the arc minimum is then the "dark"
 * score in the low half and the "bright" score in the high half, so ONE log-step sliding-minimum network of
 * v_pk_min_i16 / v_pk_max_i16 serves both (windows of 2, 4, 8 and then 9 ring pixels). */
__device__ __forceinline__ int fast_score16(const uint8_t *c, int v) {
    /* ring in OpenCV's order: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)
       (-3,0)(-3,1)(-2,2)(-1,3) */
    int d[16];
    d[0] = v - c[3 * FT_LD + 0];
    d[1] = v - c[3 * FT_LD + 1];
    d[2] = v - c[2 * FT_LD + 2];
    d[3] = v - c[1 * FT_LD + 3];
    d[4] = v - c[3];
    d[5] = v - c[-1 * FT_LD + 3];
    d[6] = v - c[-2 * FT_LD + 2];
    d[7] = v - c[-3 * FT_LD + 1];
    d[8] = v - c[-3 * FT_LD + 0];
    d[9] = v - c[-3 * FT_LD - 1];
    d[10] = v - c[-2 * FT_LD - 2];
    d[11] = v - c[-1 * FT_LD - 3];
    d[12] = v - c[-3];
    d[13] = v - c[1 * FT_LD - 3];
    d[14] = v - c[2 * FT_LD - 2];
    d[15] = v - c[3 * FT_LD - 1];
    v2s p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        p[k].x = (short)d[k];
        p[k].y = (short)(-d[k]);
    }
    v2s w2[16], w4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w2[k] = __builtin_elementwise_min(p[k], p[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) w4[k] = __builtin_elementwise_min(w2[k], w2[(k + 2) & 15]);
    v2s best = {-1000, -1000};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        v2s w9 = __builtin_elementwise_min(__builtin_elementwise_min(w4[k], w4[(k + 4) & 15]), p[(k + 8) & 15]);
        best = __builtin_elementwise_max(best, w9);
    }
    return max((int)best.x, (int)best.y) - 1;
}

__global__ __launch_bounds__(256, 8) void k_fast_cells(OrbGeom g, const uint8_t *pyr, const CellDesc *cells,
                                                     uint32_t *slab, int32_t *cell_cnt) {
    __shared__ __attribute__((aligned(16))) uint8_t s_px[ORB_CELL_MAX][FT_LD];
    __shared__ __attribute__((aligned(16))) uint8_t s_sc[ORB_CELL_MAX][FT_LD];
    __shared__ int s_wave[4];
    __shared__ int s_any;
    const int item = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int frame = item / g.total_cells, cell = item - frame * g.total_cells;
    const CellDesc cd = cells[cell];
    const LevelGeom &L = g.lv[cd.level];
    const int rw = cd.rw, rh = cd.rh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *src = level_ptr(pyr, g, frame, cd.level) + (int64_t)cd.iniY * L.pitch + cd.iniX;
    /* stage ROI: one wave per row, lanes along x (row-coalesced) ; zero the score tile */
    for (int r = wave; r < rh; r += 4) {
        const uint8_t *row = src + (int64_t)r * L.pitch;
        for (int c = lane; c < rw; c += 64) s_px[r][c] = row[c];
    }
    for (int i = tid; i < ORB_CELL_MAX * FT_LD / 4; i += 256) ((uint32_t *)&s_sc[0][0])[i] = 0u;
    if (tid == 0) s_any = 0;
    __syncthreads();
    const int iw = rw - 6, ih = rh - 6; /* detection zone */
    const int n_in = (iw > 0 && ih > 0) ? iw * ih : 0;
    /* i / iw without an integer divide: exact for i < 2^12, iw <= 59 (|error| of (i+0.5)/iw is far below 0.5/iw) */
    const float inv_iw = 1.0f / (float)(iw > 0 ? iw : 1);
#define DIV_IW(i) ((int)(((float)(i) + 0.5f) * inv_iw))
    /* scores (0 when not a corner at minThFAST) */
    for (int i = tid; i < n_in; i += 256) {
        int y = DIV_IW(i), x = i - y * iw;
        const uint8_t *c = &s_px[y + 3][x + 3];
        int s = fast_score16(c, (int)c[0]);
        s_sc[y + 3][x + 3] = (uint8_t)(s >= g.min_th ? s : 0);
    }
    __syncthreads();
    /* NMS: strictly greater than the 8 neighbours.  Each thread owns a contiguous run of the row-major
       order so that the compaction below needs a single block scan. */
    const int per = (n_in + 255) / 256;
    const int beg = min(tid * per, n_in), end = min(beg + per, n_in);
    int any_ini = 0;
    uint32_t keep_lo = 0; /* per <= 14 (59*59/256) : bit k = element beg+k kept */
    const int y_beg = DIV_IW(beg), x_beg = beg - y_beg * iw;
    for (int i = beg, y = y_beg + 3, x = x_beg + 3; i < end; i++, x++) {
        if (x == iw + 3) {
            x = 3;
            y++;
        }
        int s = s_sc[y][x];
        bool k = s > 0 && s > s_sc[y - 1][x - 1] && s > s_sc[y - 1][x] && s > s_sc[y - 1][x + 1] &&
                 s > s_sc[y][x - 1] && s > s_sc[y][x + 1] && s > s_sc[y + 1][x - 1] && s > s_sc[y + 1][x] &&
                 s > s_sc[y + 1][x + 1];
        if (k) {
            keep_lo |= 1u << (i - beg);
            any_ini |= (s >= g.ini_th);
        }
    }
    if (any_ini) s_any = 1; /* benign race: all writers store 1 */
    __syncthreads();
    const int th = s_any ? g.ini_th : g.min_th;
    int mine = 0;
    uint32_t sel = 0;
    for (int i = beg, y = y_beg + 3, x = x_beg + 3; i < end; i++, x++) {
        if (x == iw + 3) {
            x = 3;
            y++;
        }
        if (keep_lo & (1u << (i - beg))) {
            if (s_sc[y][x] >= th) {
                sel |= 1u << (i - beg);
                mine++;
            }
        }
    }
    /* block exclusive scan of `mine` in thread order: wave scan by shuffles + 4 wave totals */
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += s_wave[w];
    int pos = base + incl - mine;
    uint32_t *out = slab + (int64_t)frame * g.slab_frame_count + cd.slab_off;
    for (int i = beg, y = y_beg + 3, x = x_beg + 3; i < end; i++, x++) {
        if (x == iw + 3) {
            x = 3;
            y++;
        }
        if (sel & (1u << (i - beg))) {
            out[pos++] = KEY_PACK(x + cd.offX, y + cd.offY, s_sc[y][x]);
        }
    }
    if (tid == 255) cell_cnt[(int64_t)frame * g.total_cells + cell] = base + incl;
}


/* ---- one wavefront per cell -------------------------------------------------------------------------- */
/* cornerScore with a run-time LDS row pitch.  The network runs on the ring VALUES, not on differences: with
 * p[k] = (r[k], -r[k]) (low / high 16 bits; r * -65535 = r - (r << 16), one full-rate v_mul_i32_i24) the max
 * over the 16 nine-arcs of the arc minimum is (max-min r, -(min-max r)), and
 *   bright score = max-min r - v,   dark score = v - min-max r
 * so ONE log-step sliding-minimum network of v_pk_min_i16 / v_pk_max_i16 serves both polarities; it is ordered
 * level by level (dependent packed ops cost a wait state each on gfx950). */
__device__ __forceinline__ int fast_score16_ld(const uint8_t *c, int v, int ld) {
    const uint8_t *rm3 = c - 3 * ld, *rm2 = c - 2 * ld, *rm1 = c - ld, *rp1 = c + ld, *rp2 = c + 2 * ld, *rp3 = c + 3 * ld;
    /* ring in OpenCV's order: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)
       (-3,0)(-3,1)(-2,2)(-1,3) */
    int r[16];
    r[0] = rp3[0];
    r[1] = rp3[1];
    r[2] = rp2[2];
    r[3] = rp1[3];
    r[4] = c[3];
    r[5] = rm1[3];
    r[6] = rm2[2];
    r[7] = rm3[1];
    r[8] = rm3[0];
    r[9] = rm3[-1];
    r[10] = rm2[-2];
    r[11] = rm1[-3];
    r[12] = c[-3];
    r[13] = rp1[-3];
    r[14] = rp2[-2];
    r[15] = rp3[-1];
    v2s p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int packed = __mul24(r[k], -65535);
        p[k] = __builtin_bit_cast(v2s, packed);
    }
    v2s w2[16], w4[16], w8[16], w9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w2[k] = __builtin_elementwise_min(p[k], p[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) w4[k] = __builtin_elementwise_min(w2[k], w2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) w8[k] = __builtin_elementwise_min(w4[k], w4[(k + 4) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) w9[k] = __builtin_elementwise_min(w8[k], p[(k + 8) & 15]);
#pragma unroll
    for (int k = 0; k < 8; k++) w9[k] = __builtin_elementwise_max(w9[k], w9[k + 8]);
#pragma unroll
    for (int k = 0; k < 4; k++) w9[k] = __builtin_elementwise_max(w9[k], w9[k + 4]);
#pragma unroll
    for (int k = 0; k < 2; k++) w9[k] = __builtin_elementwise_max(w9[k], w9[k + 2]);
    const v2s best = __builtin_elementwise_max(w9[0], w9[1]);
    return max((int)best.x - v, (int)best.y + v) - 1;
}

/* Is the pixel a FAST-9/16 corner at threshold t (<=> cornerScore >= t)?
 * With V = (t - v) << 16 | (t + v), the word X[k] = V + 65535 * r[k] (one v_mad_u32_u24) holds t + v - r[k] in
 * its low half and t - v + r[k] in its high half: sign bit 15 set <=> ring pixel brighter than v + t, sign bit
 * 31 set <=> darker than v - t.  (A negative low half borrows 1 from the high half, but then r > v + t and the
 * high half, r - v + t - 1 >= 2t, stays non-negative: no false dark bit.)  "Nine contiguous ring pixels all
 * brighter, or all darker" is then a bitwise AND over nine consecutive X words read off the two sign bits:
 * 16 three-input ANDs give the runs of 3, 16 more the runs of 9 (gfx950 v_bitop3_b32), an OR tree merges the 16
 * arcs.  Under half the instructions of the exact score, which only the few percent of pixels that pass need. */
__device__ __forceinline__ int fast_is_corner_ld(const uint8_t *c, int ld, uint32_t tt) {
    /* row pointers from the top-left corner of the 7x7 window: every LDS offset is a non-negative immediate */
    const uint8_t *q0 = c - 3 * ld - 3, *q1 = q0 + ld, *q2 = q0 + 2 * ld, *q3 = q0 + 3 * ld, *q4 = q0 + 4 * ld, *q5 = q0 + 5 * ld,
                  *q6 = q0 + 6 * ld;
    uint32_t r[16];
    r[0] = q6[3];
    r[1] = q6[4];
    r[2] = q5[5];
    r[3] = q4[6];
    r[4] = q3[6];
    r[5] = q2[6];
    r[6] = q1[5];
    r[7] = q0[4];
    r[8] = q0[3];
    r[9] = q0[2];
    r[10] = q1[1];
    r[11] = q2[0];
    r[12] = q3[0];
    r[13] = q4[0];
    r[14] = q5[1];
    r[15] = q6[2];
    uint32_t V = tt + (uint32_t)__mul24((int)q3[3], -65535);
    asm volatile("" : "+v"(V)); /* opaque: keeps X[k] a single mad (the compiler would re-associate into (r - v) * 65535 + tt) */
    uint32_t sg[16], t3[16], t9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) sg[k] = __umul24(r[k], 65535u) + V;
#pragma unroll
    for (int k = 0; k < 16; k++) t3[k] = sg[k] & sg[(k + 1) & 15] & sg[(k + 2) & 15];
#pragma unroll
    for (int k = 0; k < 16; k++) t9[k] = t3[k] & t3[(k + 3) & 15] & t3[(k + 6) & 15];
    const uint32_t o0 = t9[0] | t9[1] | t9[2], o1 = t9[3] | t9[4] | t9[5], o2 = t9[6] | t9[7] | t9[8], o3 = t9[9] | t9[10] | t9[11],
                   o4 = t9[12] | t9[13] | t9[14];
    /* a bright and a dark 9-arc cannot coexist on a 16-ring (they would share two pixels): at most one bit is set */
    const uint32_t o = ((o0 | o1 | o2) | (o3 | o4 | t9[15])) & 0x80008000u;
    return o == 0 ? 0 : ((o & 0x8000u) ? 1 : 2); /* 1: brighter arc, 2: darker arc */
}

/* Exact cornerScore of TWO corners of known polarity in one pass.  For polarity s (+1 bright, -1 dark) the score is
 * max over the 16 nine-arcs of min_k s*(r[k] - v), minus 1 -- the other polarity cannot win (see above).  The two
 * pixels ride in the halves of one register as biased non-negative values d = s*(r - v) + 256 (0..511), built
 * with two v_mad_i32_i24 per ring position (the second shifts its product into the high half), and share ONE
 * sliding-minimum network of v_pk_min_i16 / v_pk_max_i16. */
__device__ __forceinline__ uint32_t fast_score_pair_ld(const uint8_t *ca, int sa, const uint8_t *cb, int sb, int ld) {
    const uint8_t *a0 = ca - 3 * ld - 3, *a1 = a0 + ld, *a2 = a0 + 2 * ld, *a3 = a0 + 3 * ld, *a4 = a0 + 4 * ld, *a5 = a0 + 5 * ld,
                  *a6 = a0 + 6 * ld;
    const uint8_t *b0 = cb - 3 * ld - 3, *b1 = b0 + ld, *b2 = b0 + 2 * ld, *b3 = b0 + 3 * ld, *b4 = b0 + 4 * ld, *b5 = b0 + 5 * ld,
                  *b6 = b0 + 6 * ld;
    int ra[16], rb[16];
    ra[0] = a6[3];  rb[0] = b6[3];
    ra[1] = a6[4];  rb[1] = b6[4];
    ra[2] = a5[5];  rb[2] = b5[5];
    ra[3] = a4[6];  rb[3] = b4[6];
    ra[4] = a3[6];  rb[4] = b3[6];
    ra[5] = a2[6];  rb[5] = b2[6];
    ra[6] = a1[5];  rb[6] = b1[5];
    ra[7] = a0[4];  rb[7] = b0[4];
    ra[8] = a0[3];  rb[8] = b0[3];
    ra[9] = a0[2];  rb[9] = b0[2];
    ra[10] = a1[1]; rb[10] = b1[1];
    ra[11] = a2[0]; rb[11] = b2[0];
    ra[12] = a3[0]; rb[12] = b3[0];
    ra[13] = a4[0]; rb[13] = b4[0];
    ra[14] = a5[1]; rb[14] = b5[1];
    ra[15] = a6[2]; rb[15] = b6[2];
    const int va = a3[3], vb = b3[3];
    /* K = (256 - sa*va) + ((256 - sb*vb) << 16); word[k] = K + sa*ra[k] + (sb*rb[k] << 16): both halves in 0..511 */
    int K = (256 - sa * va) + ((256 - sb * vb) << 16);
    asm volatile("" : "+v"(K)); /* opaque: keeps each word two plain mads */
    const int sb16 = sb * 65536;
    v2s p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int w = __mul24(rb[k], sb16) + (__mul24(ra[k], sa) + K);
        p[k] = __builtin_bit_cast(v2s, w);
    }
    v2s w2[16], w4[16], w8[16], w9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w2[k] = __builtin_elementwise_min(p[k], p[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) w4[k] = __builtin_elementwise_min(w2[k], w2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) w8[k] = __builtin_elementwise_min(w4[k], w4[(k + 4) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) w9[k] = __builtin_elementwise_min(w8[k], p[(k + 8) & 15]);
#pragma unroll
    for (int k = 0; k < 8; k++) w9[k] = __builtin_elementwise_max(w9[k], w9[k + 8]);
#pragma unroll
    for (int k = 0; k < 4; k++) w9[k] = __builtin_elementwise_max(w9[k], w9[k + 4]);
#pragma unroll
    for (int k = 0; k < 2; k++) w9[k] = __builtin_elementwise_max(w9[k], w9[k + 2]);
    const v2s best = __builtin_elementwise_max(w9[0], w9[1]);
    const uint32_t sc_a = (uint32_t)((int)best.x - 257), sc_b = (uint32_t)((int)best.y - 257);
    return sc_a | (sc_b << 16);
}

/* ONE WAVEFRONT per (cell, frame): a 64-thread workgroup, LDS tiles sized to the largest cell of the current
 * geometry (about 4 KB at 640x480), ROI staged as aligned dwords.  No cross-wave barrier, no workgroup scan.
 *   pass A0 compass pre-test (4 ring pixels) of the detection zone in dense 64-pixel row-major chunks; survivors
 *           (about a quarter) are appended, in order, to an LDS queue by ballot/popcount prefix;
 *   pass A  full corner test at minThFAST of the queued pixels, queue compacted in place (about a tenth remain);
 *   pass S  exact cornerScore of the queued pixels only, two per lane (one shared network), 128 per step;
 *   pass B  3x3 non-maximum suppression of the queued pixels, flags kept / kept-and->=iniThFAST written back
 *           into the queue entry;
 *   pass C  ordered emission of the queue entries selected by the per-cell threshold fallback. */
__global__ __launch_bounds__(64) void k_fast_cells_wave(OrbGeom g, const uint8_t *pyr, const CellDesc *cells, uint32_t *slab,
                                                        int32_t *cell_cnt, int ld, int lh, int stop_after, int fast_a0_bytes) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fw_smem[];
    uint8_t *s_px = fw_smem;                             /* [lh][ld], column 0 <-> x = iniX & ~3 */
    uint8_t *s_sc = fw_smem + ld * lh;                   /* scores, same layout */
    uint16_t *s_q = (uint16_t *)(fw_smem + 2 * ld * lh); /* queue: y << 7 | x, bit 14 kept, bit 15 kept at iniTh */
    const int item = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int frame = item / g.total_cells, cell = item - frame * g.total_cells;
    const CellDesc cd = cells[cell];
    const LevelGeom &L = g.lv[cd.level];
    const int rw = cd.rw, rh = cd.rh, lane = threadIdx.x;
    const int ox = cd.iniX & 3;
    const uint8_t *src = level_ptr(pyr, g, frame, cd.level) + (int64_t)cd.iniY * L.pitch + (cd.iniX & ~3);
    const int ndw = (ox + rw + 3) >> 2; /* aligned dwords per ROI row (the pitch is a multiple of 64) */
    const float inv_ndw = 1.0f / (float)ndw; /* i / ndw without the integer divide: exact for i < 2^12 */
    if (ndw <= 16 && !fast_a0_bytes) {
        /* lane -> (row lane / 16, dword column lane % 16): no division per element, the row step is a pointer increment;
           the lanes of columns >= ndw idle (ROI rows are 9 .. 11 dwords at 640 x 480) */
        const int c4 = lane & 15;
        if (c4 < ndw) {
            const uint8_t *g = src + __mul24(lane >> 4, L.pitch) + c4 * 4;
            uint8_t *d = s_px + __mul24(lane >> 4, ld) + c4 * 4;
            const int gstep = 4 * L.pitch, dstep = 4 * ld;
#pragma unroll 4
            for (int r = lane >> 4; r < rh; r += 4, g += gstep, d += dstep) *(uint32_t *)d = *(const uint32_t *)g;
        }
    } else
    for (int i = lane; i < rh * ndw; i += 64) {
        const int r = (int)(((float)i + 0.5f) * inv_ndw), c4 = i - __mul24(r, ndw);
        *(uint32_t *)(s_px + __mul24(r, ld) + c4 * 4) = *(const uint32_t *)(src + __mul24(r, L.pitch) + c4 * 4);
    }
    for (int i = lane; i < (ld * lh) >> 2; i += 64) ((uint32_t *)s_sc)[i] = 0u;
    __syncthreads();
    if (stop_after == 9) return; /* staging only */
    const int iw = rw - 6, ih = rh - 6;
    const int n_in = (iw > 0 && ih > 0) ? iw * ih : 0;
    const float inv_iw = 1.0f / (float)(iw > 0 ? iw : 1);
    const int nch = (n_in + 63) >> 6;
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    /* The reference runs cv::FAST at iniThFAST and only a cell WITHOUT a single keypoint is redone at minThFAST
       (src/ORBextractor.cpp:765-773).  Same here: the passes run at iniThFAST first -- non-maximum suppression among
       the corners of that threshold is exactly cv::FAST(iniThFAST, true), scores below the threshold count as 0 there
       too -- and the (rare, flat) cell that keeps nothing repeats them at minThFAST.  At minThFAST = 7 more than a
       third of the pixels of a textured frame are "corners" (every pixel beside an edge has nine contiguous brighter
       ring pixels); at iniThFAST = 20 a few percent are, so the common case tests and scores a fraction of the
       pixels. */
    int nq = 0;
    bool any_kept = false;
#pragma unroll 1
    for (int attempt = 0; attempt < 2 && !any_kept; attempt++) {
    const uint32_t tt = (uint32_t)(attempt == 0 ? g.ini_th : g.min_th) * 0x00010001u;
    /* pass A0: compass pre-test of every pixel.  A run of 9 on the 16-ring always contains two ADJACENT compass
       points (ring positions 0, 4, 8, 12), and "two adjacent of a 4-cycle" is (N|S)&(E|W): with the sign-bit words
       X[k] of fast_is_corner_ld this is two logic ops for both polarities at once.  About a quarter of the pixels
       of a textured image survive; they are queued in row-major order. */
    int nq0 = 0;
    if (!fast_a0_bytes) {
        /* FOUR pixels per lane: the aligned dword under the pixels, its neighbours left and right and the dwords three
           rows up and down -- 5 LDS dword reads instead of 20 byte reads per 4 pixels, one index computation per 4 pixels
           -- and the test on packed 16-bit halves (even / odd bytes of every dword):
               brighter on two adjacent compass points  <=>  min(max(N, S), max(E, W)) > v + t
               darker   on two adjacent compass points  <=>  max(min(N, S), min(E, W)) < v - t
           six v_pk_min/max_u16 and four v_pk_add/sub per pixel pair.  Same predicate, same queue order (groups walk the
           zone row-major, a lane appends its pixels left to right behind the pixels of the lanes below it). */
        const int xb = (3 + ox) & ~3, xe = 3 + ox + iw; /* first aligned x that covers the zone; zone end */
        const int ng = (iw > 0 && ih > 0) ? (xe - xb + 3) >> 2 : 0;
        const int ngroups = ng * (ih > 0 ? ih : 0);
        const float inv_ng = 1.0f / (float)(ng > 0 ? ng : 1);
        const v2u T = __builtin_bit_cast(v2u, tt);
        for (int g0 = 0; g0 < ngroups; g0 += 64) {
            const int g = g0 + lane;
            uint32_t hits = 0;
            int x0 = 0, y = 0;
            if (g < ngroups) {
                const int yy = (int)(((float)g + 0.5f) * inv_ng);
                x0 = xb + 4 * (g - __mul24(yy, ng));
                y = yy + 3;
                const uint8_t *row = s_px + __mul24(y, ld) + x0;
                const uint32_t C = *(const uint32_t *)row, L = *(const uint32_t *)(row - 4), R = *(const uint32_t *)(row + 4);
                const uint32_t N = *(const uint32_t *)(row - 3 * ld), S = *(const uint32_t *)(row + 3 * ld);
                const uint32_t W = __builtin_amdgcn_alignbyte(C, L, 1), E = __builtin_amdgcn_alignbyte(R, C, 3);
                uint32_t bits[2];
#pragma unroll
                for (int par = 0; par < 2; par++) {
                    const int sh = par * 8;
                    const v2u c = __builtin_bit_cast(v2u, (C >> sh) & 0x00FF00FFu), n = __builtin_bit_cast(v2u, (N >> sh) & 0x00FF00FFu),
                              so = __builtin_bit_cast(v2u, (S >> sh) & 0x00FF00FFu), e = __builtin_bit_cast(v2u, (E >> sh) & 0x00FF00FFu),
                              we = __builtin_bit_cast(v2u, (W >> sh) & 0x00FF00FFu);
                    const v2u hi = __builtin_elementwise_min(__builtin_elementwise_max(n, so), __builtin_elementwise_max(e, we));
                    const v2u lo = __builtin_elementwise_max(__builtin_elementwise_min(n, so), __builtin_elementwise_min(e, we));
                    const v2s br = __builtin_bit_cast(v2s, (v2u)(c + T)) - __builtin_bit_cast(v2s, hi);  /* < 0: brighter pair */
                    const v2s dk = __builtin_bit_cast(v2s, lo) - (__builtin_bit_cast(v2s, c) - __builtin_bit_cast(v2s, T)); /* < 0: darker pair */
                    bits[par] = (__builtin_bit_cast(uint32_t, br) | __builtin_bit_cast(uint32_t, dk)) & 0x80008000u;
                }
                /* pixel k of the dword: even halves carry k = 0 (bit 15) and k = 2 (bit 31), odd halves k = 1 and k = 3 */
                hits = ((bits[0] >> 15) & 1u) | ((bits[1] >> 14) & 2u) | ((bits[0] >> 29) & 4u) | ((bits[1] >> 28) & 8u);
                /* pixels of the dword outside the detection zone [3 + ox, xe) */
                const int lo_cut = 3 + ox - x0, hi_cut = xe - x0; /* valid k: lo_cut <= k < hi_cut */
                uint32_t valid = 0xFu;
                if (lo_cut > 0) valid &= 0xFu << lo_cut;
                if (hi_cut < 4) valid &= (1u << hi_cut) - 1u;
                hits &= valid;
            }
            const uint64_t m0 = __ballot(hits & 1u), m1 = __ballot(hits & 2u), m2 = __ballot(hits & 4u), m3 = __ballot(hits & 8u);
            int pos = nq0 + __popcll(m0 & lt) + __popcll(m1 & lt) + __popcll(m2 & lt) + __popcll(m3 & lt);
            const int e0 = y << 7 | x0;
            if (hits & 1u) s_q[pos++] = (uint16_t)e0;
            if (hits & 2u) s_q[pos++] = (uint16_t)(e0 + 1);
            if (hits & 4u) s_q[pos++] = (uint16_t)(e0 + 2);
            if (hits & 8u) s_q[pos++] = (uint16_t)(e0 + 3);
            nq0 += __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
        }
    } else
    for (int ch = 0; ch < nch; ch++) {
        const int i = ch * 64 + lane;
        bool hit = false;
        int x = 0, y = 0;
        if (i < n_in) {
            const int yy = (int)(((float)i + 0.5f) * inv_iw);
            x = i - __mul24(yy, iw) + 3 + ox;
            y = yy + 3;
            const uint8_t *c = s_px + __mul24(y, ld) + x - 3; /* centre at c[3] */
            const uint8_t *up = c - 3 * ld, *dn = c + 3 * ld;
            uint32_t V = tt + (uint32_t)__mul24((int)c[3], -65535);
            asm volatile("" : "+v"(V));
            const uint32_t s0 = __umul24((uint32_t)dn[3], 65535u) + V, s4 = __umul24((uint32_t)c[6], 65535u) + V,
                           s8 = __umul24((uint32_t)up[3], 65535u) + V, s12 = __umul24((uint32_t)c[0], 65535u) + V;
            hit = (((s0 | s8) & (s4 | s12)) & 0x80008000u) != 0;
        }
        const uint64_t m = __ballot(hit);
        if (hit) s_q[nq0 + __popcll(m & lt)] = (uint16_t)(y << 7 | x);
        nq0 += __popcll(m);
    }
    __syncthreads();
    if (stop_after == 1) return; /* diagnostics (ORBSLAM_AMD_FAST_STOP): time split of the passes, results invalid */
    /* pass A: the full corner test on the queued pixels, the queue compacted in place (a wave reads its 64
       entries before it writes at most as many at or below them) */
    nq = 0;
#pragma unroll 1
    for (int j0 = 0; j0 < nq0; j0 += 64) {
        const int j = j0 + lane;
        int pol = 0, e = 0;
        if (j < nq0) {
            e = s_q[j];
            pol = fast_is_corner_ld(s_px + __mul24(e >> 7, ld) + (e & 127), ld, tt);
        }
        __syncthreads();
        const bool hit = pol != 0;
        const uint64_t m = __ballot(hit);
        if (hit) s_q[nq + __popcll(m & lt)] = (uint16_t)(e | (pol == 2 ? 1 << 14 : 0)); /* bit 14: darker arc */
        nq += __popcll(m);
    }
    __syncthreads();
    if (stop_after == 2) return;
    /* pass S: two queued corners per lane (entries j and j + 64 of each 128-entry step) */
#pragma unroll 1
    for (int j0 = 0; j0 < nq; j0 += 128) {
        const int ja = j0 + lane, jb = ja + 64;
        if (ja < nq) {
            const int ea = s_q[ja], eb = jb < nq ? (int)s_q[jb] : ea;
            const int pa = __mul24((ea >> 7) & 127, ld) + (ea & 127), pb = __mul24((eb >> 7) & 127, ld) + (eb & 127);
            const uint32_t sc = fast_score_pair_ld(s_px + pa, (ea & (1 << 14)) ? -1 : 1, s_px + pb, (eb & (1 << 14)) ? -1 : 1, ld);
            s_sc[pa] = (uint8_t)(sc & 0xFFFFu);
            if (jb < nq) s_sc[pb] = (uint8_t)(sc >> 16);
            s_q[ja] = (uint16_t)(ea & 0x3FFF); /* the polarity bit has served: bits 14/15 become the NMS flags */
            if (jb < nq) s_q[jb] = (uint16_t)(eb & 0x3FFF);
        }
    }
    __syncthreads();
    if (stop_after == 3) return;
    /* pass B: 3x3 non-maximum suppression among the corners of this attempt's threshold (bit 14 = kept) */
#pragma unroll 1
    for (int j0 = 0; j0 < nq; j0 += 64) {
        const int j = j0 + lane;
        bool k = false;
        if (j < nq) {
            const int e = s_q[j];
            const uint8_t *q = s_sc + __mul24((e >> 7) & 127, ld) + (e & 127);
            const int sv = q[0];
            k = sv > q[-ld - 1] && sv > q[-ld] && sv > q[-ld + 1] && sv > q[-1] && sv > q[1] && sv > q[ld - 1] &&
                sv > q[ld] && sv > q[ld + 1];
            s_q[j] = (uint16_t)(e | (k ? 1 << 14 : 0));
        }
        any_kept = any_kept || (__ballot(k) != 0);
    }
    __syncthreads();
    } /* attempt: iniThFAST, then minThFAST for a cell that kept nothing (:765-773) */
    /* pass C */
    uint32_t *out = slab + (int64_t)frame * g.slab_frame_count + cd.slab_off;
    const int want = 1 << 14;
    int base = 0;
#pragma unroll 1
    for (int j0 = 0; j0 < nq; j0 += 64) {
        const int j = j0 + lane;
        const int e = j < nq ? (int)s_q[j] : 0;
        const bool sel = (e & want) != 0;
        const uint64_t m = __ballot(sel);
        if (sel) {
            const int y = (e >> 7) & 127, x = e & 127;
            out[base + __popcll(m & lt)] = KEY_PACK(x - ox + cd.offX, y + cd.offY, s_sc[__mul24(y, ld) + x]);
        }
        base += __popcll(m);
    }
    if (lane == 0) cell_cnt[(int64_t)frame * g.total_cells + cell] = base;
}

hipError_t launch_fast(const OrbGeom &g, const OrbBuffers &b, int n_frames, hipStream_t st) {
    if (g.total_cells == 0) return hipSuccess;
    dim3 grid(g.total_cells, n_frames);
    if (g.cell_max_rw > 0) { /* one wavefront per cell */
        const int ld = (g.cell_max_rw + 3 + 3) & ~3, lh = g.cell_max_rh;
        static const int stop_after = getenv("ORBSLAM_AMD_FAST_STOP") ? atoi(getenv("ORBSLAM_AMD_FAST_STOP")) : 0;
        static const int a0_bytes = getenv("ORBSLAM_AMD_FAST_A0_BYTES") ? 1 : 0; /* diagnostics: the one-pixel-per-lane pre-test */
        hipLaunchKernelGGL(k_fast_cells_wave, grid, dim3(64), (size_t)2 * ld * lh + 2 * (size_t)(g.cell_max_rw * g.cell_max_rh), st, g, (const uint8_t *)b.pyr, b.cells, b.slab,
                           b.cell_cnt, ld, lh, stop_after, a0_bytes);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, st, g, (const uint8_t *)b.pyr, b.cells, b.slab, b.cell_cnt);
    return hipGetLastError();
}
