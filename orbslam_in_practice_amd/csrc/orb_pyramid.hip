/*
 * orb_pyramid.hip -- scale pyramid and Gaussian blur for gfx950.
 *
 * Replaces ORBextractor::ComputePyramid (reference src/ORBextractor.cpp:1071-1096: cv::resize
 * INTER_LINEAR level l-1 -> l) and the per-level cv::GaussianBlur(7x7, sigma 2, REFLECT_101) of
 * operator() (:1045-1046).  Integer fixed-point arithmetic identical to the OpenCV 2.4/3.x CV_8U
 * paths (11-bit bilinear coefficients; 8-bit Gaussian taps {18,34,49,55,49,34,18}), so results are
 * bit-exact with the CPU restatement in oracle/orb_oracle.cpp.
 *
 * HBM-bound byte work: rows are read/written as coalesced 64..256-byte segments, the blur stages its
 * (32+6)x(64+6) input tile and the horizontal pass in LDS.  The 19-px apron of mvImagePyramid is not
 * stored on the device (no kernel reads it); orb_download_level() rebuilds it on request.
 */
#include "orb_common.h"

struct TileMap {
    int prefix[ORB_MAX_LEVELS + 1];
    int tiles_x[ORB_MAX_LEVELS];
};

/* ---- level 0: copy the input frame into the pitch-aligned pyramid block ------------------------ */
__global__ __launch_bounds__(256) void k_copy_level0(OrbGeom g, uint8_t *pyr, const uint8_t *imgs, int stride,
                                                      int64_t frame_stride, int vec16) {
    const int frame = blockIdx.z;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= g.h) return;
    const uint8_t *src = imgs + (int64_t)frame * frame_stride + (int64_t)y * stride;
    uint8_t *dst = level_ptr(pyr, g, frame, 0) + (int64_t)y * g.lv[0].pitch;
    const int lane = threadIdx.x & 63;
    if (vec16) {
        for (int x = (blockIdx.x * 64 + lane) * 16; x < g.w; x += gridDim.x * 64 * 16) {
            if (x + 16 <= g.w) {
                *(uint4 *)(dst + x) = *(const uint4 *)(src + x);
            } else {
                for (int i = 0; x + i < g.w; i++) dst[x + i] = src[x + i];
            }
        }
    } else {
        for (int x = (blockIdx.x * 64 + lane) * 4; x < g.w; x += gridDim.x * 64 * 4) {
            uint32_t v = 0;
            for (int i = 0; i < 4; i++)
                if (x + i < g.w) v |= (uint32_t)src[x + i] << (8 * i);
            *(uint32_t *)(dst + x) = v; /* pitch is a multiple of 64: the tail store stays inside the row */
        }
    }
}

/* ---- level l from level l-1, cv::resize INTER_LINEAR fixed point -------------------------------- */
/* One workgroup = a 128 x 32 output tile.  The source window it touches is staged in LDS as aligned dwords
 * (row-coalesced; window size per level computed on the host from the same tables, LevelGeom::rs_lw/rs_lh).
 * A thread owns 4 adjacent output columns -- their {sx, a0, a1} table entries are loaded once and stay in
 * registers -- and walks 4 output rows (y, y+8, y+16, y+24), one packed u32 store per row.  All index
 * arithmetic is 24-bit multiplies / float reciprocals: no v_mul_lo_u32, no integer division in the loops. */
#define RS_TW 128
#define RS_TH 32
__global__ __launch_bounds__(256) void k_resize(OrbGeom g, uint8_t *pyr, const int32_t *tab_i, const int16_t *tab_c,
                                                 int level) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_src[];
    const LevelGeom &L = g.lv[level];
    const LevelGeom &S = g.lv[level - 1];
    const int nbx = gridDim.x, nby = gridDim.y;
    const int item = xcd_remap((blockIdx.z * nby + blockIdx.y) * nbx + blockIdx.x, nbx * nby * (int)gridDim.z);
    const int frame = item / (nbx * nby), rem = item - frame * (nbx * nby);
    const int by = rem / nbx, bx = rem - by * nbx;
    const int tx0 = bx * RS_TW, ty0 = by * RS_TH;
    const int tid = threadIdx.x;
    const uint8_t *src = level_ptr(pyr, g, frame, level - 1);
    uint8_t *dst = level_ptr(pyr, g, frame, level);
    const int32_t *xofs = tab_i + L.xofs_off;
    const int32_t *yofs = tab_i + L.yofs_off;
    const uint32_t *ialpha = (const uint32_t *)(tab_c + 2 * L.xofs_off); /* (a0, a1) pairs, 4-byte aligned */
    const uint32_t *ibeta = (const uint32_t *)(tab_c + 2 * L.yofs_off);
    const int lw = L.rs_lw;
    if (lw > 0) {
        /* Every table read of this thread is issued up front so that they share ONE memory round trip: the four
           window bounds of the tile (lanes 0..3 of each wave fetch one each, broadcast by v_readlane), the
           {sx, a0a1} entries of its 4 columns and the {sy, b0b1} entries of its 4 rows. */
        const int x_last = min(tx0 + RS_TW, L.w) - 1, y_last = min(ty0 + RS_TH, L.h) - 1;
        const int sel = tid & 3;
        const int32_t *bp = sel == 0 ? xofs + tx0 : sel == 1 ? xofs + x_last : sel == 2 ? yofs + ty0 : yofs + y_last;
        const int bound = *bp;
        const int x0 = tx0 + (tid & 31) * 4;
        int sxv[4], syv[RS_TH / 8];
        uint32_t av[4], bv[RS_TH / 8];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = min(x0 + i, L.w - 1);
            sxv[i] = xofs[x];
            av[i] = ialpha[x];
        }
#pragma unroll
        for (int rr = 0; rr < RS_TH / 8; rr++) {
            const int y = min(ty0 + (tid >> 5) + 8 * rr, L.h - 1);
            syv[rr] = yofs[y];
            bv[rr] = ibeta[y];
        }
        /* source window of this tile */
        const int sx_lo = __builtin_amdgcn_readlane(bound, 0) & ~3; /* xofs is non-decreasing and >= 0 */
        const int sx_hi = min(__builtin_amdgcn_readlane(bound, 1) + 1, S.w - 1);
        const int sy_lo = min(max(__builtin_amdgcn_readlane(bound, 2), 0), S.h - 1);
        const int sy_hi = min(max(__builtin_amdgcn_readlane(bound, 3) + 1, 0), S.h - 1);
        const int ndw = (sx_hi - sx_lo) / 4 + 1, nrow = sy_hi - sy_lo + 1;
        const float inv_ndw = 1.0f / (float)ndw;
        const uint8_t *win = src + (int64_t)sy_lo * S.pitch + sx_lo;
        for (int i = tid; i < nrow * ndw; i += 256) {
            const int r = (int)(((float)i + 0.5f) * inv_ndw), c4 = i - __mul24(r, ndw);
            /* the pitch is a multiple of 64 and planes are 256-B aligned: the dword stays inside the plane */
            *(uint32_t *)(s_src + __mul24(r, lw) + c4 * 4) = *(const uint32_t *)(win + __mul24(r, S.pitch) + c4 * 4);
        }
        __syncthreads();
        if (x0 >= L.w) return;
        /* bilinear coefficients are non-negative (0..2048): unsigned arithmetic with visible bit widths lets the
           compiler pick the full-rate 24-bit multiplies */
        int c0[4], c1[4];
        uint32_t a0[4], a1[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            c0[i] = sxv[i] - sx_lo;
            c1[i] = min(sxv[i] + 1, S.w - 1) - sx_lo;
            a0[i] = av[i] & 0xFFFFu;
            a1[i] = av[i] >> 16;
        }
#pragma unroll
        for (int rr = 0; rr < RS_TH / 8; rr++) {
            const int y = ty0 + (tid >> 5) + 8 * rr;
            if (y >= L.h) break;
            const int sy = syv[rr];
            const uint32_t b0 = bv[rr] & 0xFFFFu, b1 = bv[rr] >> 16;
            const uint8_t *p0 = s_src + __mul24(min(max(sy, 0), S.h - 1) - sy_lo, lw);
            const uint8_t *p1 = s_src + __mul24(min(max(sy + 1, 0), S.h - 1) - sy_lo, lw);
            uint32_t packed = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t h0 = (uint32_t)p0[c0[i]] * a0[i] + (uint32_t)p0[c1[i]] * a1[i];
                const uint32_t h1 = (uint32_t)p1[c0[i]] * a0[i] + (uint32_t)p1[c1[i]] * a1[i];
                const uint32_t v = (((b0 * ((h0 >> 4) & 0xFFFFu)) >> 16) + ((b1 * ((h1 >> 4) & 0xFFFFu)) >> 16) + 2) >> 2;
                packed |= (v & 0xFF) << (8 * i);
            }
            /* columns past the level width are padding inside the 64-byte-aligned pitch */
            *(uint32_t *)(dst + __mul24(y, L.pitch) + x0) = packed;
        }
        return;
    }
    /* generic path (a scale step so large that the window does not fit the LDS budget): direct gathers */
    for (int rr = 0; rr < RS_TH / 8; rr++) {
        const int y = ty0 + (tid >> 5) + 8 * rr, x0 = tx0 + (tid & 31) * 4;
        if (y >= L.h || x0 >= L.w) return;
        const int sy = yofs[y];
        const uint8_t *r0 = src + (int64_t)min(max(sy, 0), S.h - 1) * S.pitch;
        const uint8_t *r1 = src + (int64_t)min(max(sy + 1, 0), S.h - 1) * S.pitch;
        const uint32_t b = ibeta[y];
        const int b0 = (int)(int16_t)(b & 0xFFFF), b1 = (int)(int16_t)(b >> 16);
        uint32_t packed = 0;
        for (int i = 0; i < 4; i++) {
            const int x = x0 + i;
            if (x < L.w) {
                const int sx = xofs[x], sx1 = min(sx + 1, S.w - 1);
                const uint32_t a = ialpha[x];
                const int a0 = (int)(int16_t)(a & 0xFFFF), a1 = (int)(int16_t)(a >> 16);
                const int h0 = r0[sx] * a0 + r0[sx1] * a1, h1 = r1[sx] * a0 + r1[sx1] * a1;
                const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                packed |= (uint32_t)(v & 0xFF) << (8 * i);
            }
        }
        *(uint32_t *)(dst + (int64_t)y * L.pitch + x0) = packed;
    }
}

/* ---- 7x7 Gaussian blur, all levels in one launch -------------------------------------------------- */
#define BL_TW 64
#define BL_TH 32
__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

/* 64 x 32 output tile per workgroup.  Staging: rows [ty0-3, ty0+35) x bytes [tx0-4, tx0+68) as aligned
 * dwords (18 per row, coalesced); tiles that touch the left/right image edge take the byte path with
 * REFLECT_101 indexing.
 * Horizontal pass: 4 adjacent pixels per thread from 3 LDS dwords; the 7 taps of one pixel are two
 * v_dot4_u32_u8 over byte windows cut out with v_alignbyte_b32 (taps {18,34,49,55} and {49,34,18,0}).
 * The 16-bit row sums (<= 255 * 257) are stored as VERTICAL pairs -- dword (m, x) = rows (2m, 2m+1) of column
 * x -- so that the vertical pass is four v_dot2_u32_u16 per pixel: output rows 2t and 2t+1 both read the pairs
 * t..t+3, with taps {18,34}{49,55}{49,34}{18,0} resp. {0,18}{34,49}{55,49}{34,18}.  One thread = 4 columns x
 * 2 rows; the rounding constant rides in the accumulator.  Exact integer arithmetic in any order, so the
 * result equals the separable row-then-column filter of the reference bit for bit. */
typedef unsigned short blur_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t blur_dot2(uint32_t pair, uint32_t taps, uint32_t acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(blur_us2, pair), __builtin_bit_cast(blur_us2, taps), acc, false);
}
__global__ __launch_bounds__(256) void k_blur(OrbGeom g, const uint8_t *pyr, uint8_t *blur, TileMap tm) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[BL_TH + 6][BL_TW + 8];   /* column c <-> x = tx0 - 4 + c */
    __shared__ __attribute__((aligned(16))) uint32_t s_hp[(BL_TH + 6) / 2][BL_TW]; /* row pairs of the horizontal pass */
    const int item = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int frame = item / (int)gridDim.x, bx = item - frame * (int)gridDim.x;
    int level = 0;
    while (level + 1 < g.nlevels && bx >= tm.prefix[level + 1]) level++;
    const LevelGeom &L = g.lv[level];
    const int t = bx - tm.prefix[level];
    const int tx0 = (t % tm.tiles_x[level]) * BL_TW, ty0 = (t / tm.tiles_x[level]) * BL_TH;
    const uint8_t *src = level_ptr(pyr, g, frame, level);
    uint8_t *dst = level_ptr(blur, g, frame, level);
    const int tid = threadIdx.x;
    const bool interior = tx0 >= 4 && tx0 + BL_TW + 4 <= L.w; /* every staged byte is a real pixel of its row */
    /* staging: every load of a thread is issued before the first LDS store (one memory round trip per tile instead
       of one per loop step) */
    if (interior) {
        constexpr int NI = ((BL_TH + 6) * 18 + 255) / 256;
        uint32_t v[NI];
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = tid + k * 256;
            const int r = i / 18, c4 = i - r * 18;
            const int y = reflect101(ty0 + min(r, BL_TH + 5) - 3, L.h);
            v[k] = (i < (BL_TH + 6) * 18) ? *(const uint32_t *)(src + __mul24(y, L.pitch) + (tx0 - 4) + c4 * 4) : 0u;
        }
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = tid + k * 256;
            const int r = i / 18, c4 = i - r * 18;
            if (i < (BL_TH + 6) * 18) *(uint32_t *)&s_in[r][c4 * 4] = v[k];
        }
    } else if (L.w >= 8) {
        /* a tile on the left / right edge of its level: the same dword staging with the column clamped into the row
           (bytes outside [0, w) come out as whatever lies there), then the SIX columns per row the filter can see
           beyond the edges -- x = -3 .. -1 and w .. w + 2, REFLECT_101 -- are patched by one thread each.  The byte
           path below did this with two reflections and a byte load for every one of the 38 x 72 staged bytes: three
           times the instructions of an interior tile, on 29 % of the tiles of a 640 x 480 pyramid. */
        constexpr int NI = ((BL_TH + 6) * 18 + 255) / 256;
        uint32_t v[NI];
        const int xmax4 = L.pitch - 4;
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = tid + k * 256;
            const int r = i / 18, c4 = i - r * 18;
            const int y = reflect101(ty0 + min(r, BL_TH + 5) - 3, L.h);
            const int xc = min(max(tx0 - 4 + c4 * 4, 0), xmax4);
            v[k] = (i < (BL_TH + 6) * 18) ? *(const uint32_t *)(src + __mul24(y, L.pitch) + xc) : 0u;
        }
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int i = tid + k * 256;
            const int r = i / 18, c4 = i - r * 18;
            if (i < (BL_TH + 6) * 18) *(uint32_t *)&s_in[r][c4 * 4] = v[k];
        }
        __syncthreads();
        if (tid < (BL_TH + 6) * 6) {
            const int r = tid / 6, k = tid - r * 6;
            const int x = k < 3 ? -1 - k : L.w + (k - 3);
            const int c = x - (tx0 - 4);
            if (c >= 0 && c < BL_TW + 8) s_in[r][c] = src[__mul24(reflect101(ty0 + r - 3, L.h), L.pitch) + reflect101(x, L.w)];
        }
    } else {
        constexpr int NB = ((BL_TH + 6) * (BL_TW + 8) + 255) / 256;
        uint8_t v[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = tid + k * 256;
            const int r = min(i / (BL_TW + 8), BL_TH + 5), c = i - (i / (BL_TW + 8)) * (BL_TW + 8);
            const int y = reflect101(ty0 + r - 3, L.h), x = reflect101(tx0 - 4 + c, L.w);
            v[k] = src[__mul24(y, L.pitch) + x];
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int i = tid + k * 256;
            if (i < (BL_TH + 6) * (BL_TW + 8)) s_in[i / (BL_TW + 8)][i - (i / (BL_TW + 8)) * (BL_TW + 8)] = v[k];
        }
    }
    __syncthreads();
    /* horizontal: outputs x = tx0 + 4q .. +3 need bytes (4q+1 .. 4q+10) of the staged row */
    const uint32_t T1 = 18u | 34u << 8 | 49u << 16 | 55u << 24, T2 = 49u | 34u << 8 | 18u << 16;
    for (int i = tid; i < (BL_TH + 6) * 16; i += 256) {
        const int r = i >> 4, q = i & 15;
        const uint32_t *p = (const uint32_t *)&s_in[r][q * 4];
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        uint32_t o[4];
        o[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), T1,
                                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), T2, 0u, false), false);
        o[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), T1,
                                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), T2, 0u, false), false);
        o[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), T1,
                                      __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), T2, 0u, false), false);
        o[3] = __builtin_amdgcn_udot4(w1, T1, __builtin_amdgcn_udot4(w2, T2, 0u, false), false);
        uint16_t *hp = (uint16_t *)&s_hp[r >> 1][q * 4] + (r & 1);
#pragma unroll
        for (int k = 0; k < 4; k++) hp[2 * k] = (uint16_t)o[k]; /* each <= 65535 */
    }
    __syncthreads();
    /* vertical: thread -> (column group q of 4, output rows 2*rg and 2*rg+1) */
    {
        const int q = tid & 15, rg = tid >> 4;
        const int x = tx0 + q * 4, y0 = ty0 + rg * 2;
        if (x < L.w && y0 < L.h) {
            uint4 pr[4];
#pragma unroll
            for (int k = 0; k < 4; k++) pr[k] = *(const uint4 *)&s_hp[rg + k][q * 4];
            const uint32_t E0 = 18u | 34u << 16, E1 = 49u | 55u << 16, E2 = 49u | 34u << 16, E3 = 18u;
            const uint32_t O0 = 18u << 16, O1 = 34u | 49u << 16, O2 = 55u | 49u << 16, O3 = 34u | 18u << 16;
            uint32_t pe = 0, po = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t c0 = k == 0 ? pr[0].x : k == 1 ? pr[0].y : k == 2 ? pr[0].z : pr[0].w;
                const uint32_t c1 = k == 0 ? pr[1].x : k == 1 ? pr[1].y : k == 2 ? pr[1].z : pr[1].w;
                const uint32_t c2 = k == 0 ? pr[2].x : k == 1 ? pr[2].y : k == 2 ? pr[2].z : pr[2].w;
                const uint32_t c3 = k == 0 ? pr[3].x : k == 1 ? pr[3].y : k == 2 ? pr[3].z : pr[3].w;
                uint32_t e = blur_dot2(c3, E3, blur_dot2(c2, E2, blur_dot2(c1, E1, blur_dot2(c0, E0, 1u << 15))));
                uint32_t o = blur_dot2(c3, O3, blur_dot2(c2, O2, blur_dot2(c1, O1, blur_dot2(c0, O0, 1u << 15))));
                e = min(e >> 16, 255u);
                o = min(o >> 16, 255u);
                pe |= e << (8 * k);
                po |= o << (8 * k);
            }
#pragma unroll
            for (int rr = 0; rr < 2; rr++) {
                const int y = y0 + rr;
                if (y < L.h) {
                    const uint32_t packed = rr ? po : pe;
                    uint8_t *d = dst + __mul24(y, L.pitch) + x;
                    if (x + 4 <= L.w)
                        *(uint32_t *)d = packed;
                    else
                        for (int k = 0; x + k < L.w; k++) d[k] = (uint8_t)(packed >> (8 * k));
                }
            }
        }
    }
}

hipError_t launch_pyramid(const OrbGeom &g, const OrbBuffers &b, const uint8_t *d_imgs, int stride,
                          int64_t frame_stride, int n_frames, hipStream_t st) {
    int vec16 = (((uintptr_t)d_imgs | (uintptr_t)stride | (uintptr_t)frame_stride) & 15) == 0;
    dim3 grid0((g.w + 1023) / 1024, (g.h + 3) / 4, n_frames);
    hipLaunchKernelGGL(k_copy_level0, grid0, dim3(256), 0, st, g, b.pyr, d_imgs, stride, frame_stride, vec16);
    for (int l = 1; l < g.nlevels; l++) {
        dim3 grid((g.lv[l].w + RS_TW - 1) / RS_TW, (g.lv[l].h + RS_TH - 1) / RS_TH, n_frames);
        hipLaunchKernelGGL(k_resize, grid, dim3(256), (size_t)g.lv[l].rs_lw * g.lv[l].rs_lh, st, g, b.pyr, b.tab_i, b.tab_c, l);
    }
    return hipGetLastError();
}

hipError_t launch_blur(const OrbGeom &g, const OrbBuffers &b, int n_frames, hipStream_t st) {
    TileMap tm;
    tm.prefix[0] = 0;
    for (int l = 0; l < g.nlevels; l++) {
        tm.tiles_x[l] = (g.lv[l].w + BL_TW - 1) / BL_TW;
        int ty = (g.lv[l].h + BL_TH - 1) / BL_TH;
        tm.prefix[l + 1] = tm.prefix[l] + tm.tiles_x[l] * ty;
    }
    for (int l = g.nlevels; l < ORB_MAX_LEVELS; l++) tm.prefix[l + 1] = tm.prefix[g.nlevels];
    dim3 grid(tm.prefix[g.nlevels], n_frames);
    hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, st, g, (const uint8_t *)b.pyr, b.blur, tm);
    return hipGetLastError();
}
