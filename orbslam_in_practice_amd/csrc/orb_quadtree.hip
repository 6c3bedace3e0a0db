/*
 * orb_quadtree.hip -- ORBextractor::DistributeOctTree (reference src/ORBextractor.cpp:489-718,
 * ExtractorNode::DivideNode :431-487) as ONE 512-thread workgroup per (pyramid level, frame).
 *
 * The reference walks a std::list and pushes children to its front; every pass of that walk is a
 * deterministic permutation, so it is restated here as data-parallel passes over an array that is
 * kept in list order:
 *
 *   OUTER pass  (:558-617)  every node with >1 keys is divided, in list order.
 *   SORTED pass (:626-686)  nodes with >1 keys are divided largest-first (ties: most recently
 *                           created first -- the deterministic stand-in for the reference's
 *                           pointer-value tie-break, see oracle/orb_oracle.cpp), stopping as soon
 *                           as the list holds >= N nodes.
 *
 *   new list = children of the LAST divided node (n4,n3,n2,n1, non-empty only) ... children of the
 *              FIRST divided node, then every undivided node in its old order.
 *
 * Keys (packed x|y|score, 4 B) live in two global ping-pong arrays; a node owns a contiguous segment
 * whose order is the reference's vKeys order.  Dividing a node is a stable 4-way partition done by one
 * wavefront with ballot/popcount ranks (count sweep, then placement sweep); list positions come from
 * workgroup prefix scans in LDS.  No atomics, no host round trip, bit-exact list order.
 */
#include "orb_common.h"
#include <algorithm>
#include <cstdlib>

/* The node-list capacity NN is a template parameter (256 / 512 / 1024, picked from the largest per-level
 * quota); the workgroup has NN/2 threads so that the list-order scans handle two entries per thread. */
#define QT_T (NN / 2)
#define QT_W (NN / 128)
#define MAXN NN

template <int NN> struct QtLists {
    short x0[2][MAXN], y0[2][MAXN], x1[2][MAXN], y1[2][MAXN];
    uint32_t start[2][MAXN];
    uint32_t cnt[2][MAXN]; /* bit 31: which key buffer holds the segment */
    uint16_t born[2][MAXN];
};

template <int NN> struct QtShared {
    QtLists<NN> L;
    uint32_t ccnt[MAXN][4]; /* child key counts of candidate nodes           */
    uint16_t rnode[MAXN];   /* processing rank -> node index                 */
    uint16_t rank[MAXN];    /* node index -> processing rank (0xFFFF = none) */
    uint32_t scanR[MAXN];   /* scans in rank order                           */
    uint32_t scanL[MAXN];   /* scans in list order                           */
    uint32_t wtot[QT_W];
    uint32_t bcast[4];
};

#define CNT(v) ((v) & 0x7FFFFFFFu)
#define BUF(v) ((v) >> 31)

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

/* in-place inclusive scan of v[0..MAXN) (entries >= n must be 0); returns the total; all threads call */
template <int NN> __device__ __forceinline__ uint32_t block_scan_incl(uint32_t *v, QtShared<NN> &s) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); /* uniform: scalar unit */
    uint32_t a = v[2 * tid], b = v[2 * tid + 1];
    uint32_t incl = wave_incl_scan(a + b, lane);
    if (lane == 63) s.wtot[wave] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (int w = 0; w < QT_W; w++) {
        if (w < wave) base += s.wtot[w];
        total += s.wtot[w];
    }
    v[2 * tid] = base + incl - b;
    v[2 * tid + 1] = base + incl;
    __syncthreads();
    return total;
}

__device__ __forceinline__ int quadrant(uint32_t key, int mx, int my) {
    int x = KEY_X(key), y = KEY_Y(key);
    return (x < mx) ? ((y < my) ? 0 : 2) : ((y < my) ? 1 : 3);
}

/* one division pass.  sorted=0: OUTER, sorted=1: SORTED.  Returns the new list length, and through
 * *n_to_expand the number of new children with >1 keys.  All threads call (uniform control flow). */
template <int NN> __device__ __forceinline__ int qt_pass(QtShared<NN> &s, int cur, int nn, int N, int sorted, uint32_t *keys0, uint32_t *keys1,
                       int *n_to_expand) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); /* uniform: scalar unit */
    const int nxt = cur ^ 1;
    /* ---- A: candidates and their processing rank ---- */
    for (int e = 2 * tid; e < 2 * tid + 2; e++) {
        s.scanL[e] = (e < nn && CNT(s.L.cnt[cur][e]) > 1) ? 1u : 0u;
        s.rank[e] = 0xFFFF;
    }
    __syncthreads();
    int K;
    if (!sorted) {
        K = (int)block_scan_incl<NN>(s.scanL, s); /* inclusive: rank = scan-1 for candidates */
        for (int e = 2 * tid; e < 2 * tid + 2; e++)
            if (e < nn && CNT(s.L.cnt[cur][e]) > 1) {
                int r = (int)s.scanL[e] - 1;
                s.rank[e] = (uint16_t)r;
                s.rnode[r] = (uint16_t)e;
            }
    } else {
        uint32_t mine = 0;
        for (int e = 2 * tid; e < 2 * tid + 2; e++) {
            if (e < nn && CNT(s.L.cnt[cur][e]) > 1) {
                mine++;
                uint64_t ke = ((uint64_t)CNT(s.L.cnt[cur][e]) << 16) | s.L.born[cur][e];
                int r = 0;
                for (int j = 0; j < nn; j++) {
                    uint32_t cj = CNT(s.L.cnt[cur][j]);
                    uint64_t kj = ((uint64_t)cj << 16) | s.L.born[cur][j];
                    r += (cj > 1 && kj > ke) ? 1 : 0;
                }
                s.rank[e] = (uint16_t)r;
                s.rnode[r] = (uint16_t)e;
            }
        }
        uint32_t incl = wave_incl_scan(mine, lane);
        if (lane == 63) s.wtot[wave] = incl;
        __syncthreads();
        K = 0;
        for (int w = 0; w < QT_W; w++) K += s.wtot[w];
    }
    __syncthreads();
    /* ---- B: count the four children of every candidate (one wavefront per candidate) ---- */
    for (int r = wave; r < K; r += QT_W) {
        const int i = s.rnode[r];
        const uint32_t cv = s.L.cnt[cur][i];
        const uint32_t *src = (BUF(cv) ? keys1 : keys0) + s.L.start[cur][i];
        const int n = (int)CNT(cv);
        const int mx = s.L.x0[cur][i] + ((s.L.x1[cur][i] - s.L.x0[cur][i] + 1) >> 1);
        const int my = s.L.y0[cur][i] + ((s.L.y1[cur][i] - s.L.y0[cur][i] + 1) >> 1);
        /* private per-lane counts (packed 4 x 8 bit per batch), 8 independent loads in flight per lane */
        uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int j = lane; j < n; j += 64 * 8) {
            uint32_t kk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) kk[u] = (j + u * 64 < n) ? src[j + u * 64] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (j + u * 64 < n) {
                    const int q = quadrant(kk[u], mx, my);
                    c0 += q == 0;
                    c1 += q == 1;
                    c2 += q == 2;
                    c3 += q == 3;
                }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c0 += __shfl_xor(c0, o);
            c1 += __shfl_xor(c1, o);
            c2 += __shfl_xor(c2, o);
            c3 += __shfl_xor(c3, o);
        }
        if (lane == 0) {
            s.ccnt[i][0] = c0;
            s.ccnt[i][1] = c1;
            s.ccnt[i][2] = c2;
            s.ccnt[i][3] = c3;
        }
    }
    __syncthreads();
    /* ---- C: how many candidates are divided (m), and where everything goes ---- */
    for (int r = 2 * tid; r < 2 * tid + 2; r++) {
        uint32_t nch = 0;
        if (r < K) {
            const uint32_t *c = s.ccnt[s.rnode[r]];
            nch = (c[0] > 0) + (c[1] > 0) + (c[2] > 0) + (c[3] > 0);
        }
        s.scanR[r] = nch; /* 0 beyond K */
    }
    __syncthreads();
    block_scan_incl<NN>(s.scanR, s); /* H[r] = children of ranks 0..r */
    int m = K;
    if (sorted) {
        /* rank r is divided iff the list still has < N nodes before it: nn + (H[r-1] - r) < N */
        uint32_t mine = 0;
        for (int r = 2 * tid; r < 2 * tid + 2; r++)
            if (r < K) {
                int before = nn + (r > 0 ? (int)s.scanR[r - 1] - r : 0);
                mine += (before < N) ? 1 : 0;
            }
        uint32_t incl = wave_incl_scan(mine, lane);
        if (lane == 63) s.wtot[wave] = incl;
        __syncthreads();
        m = 0;
        for (int w = 0; w < QT_W; w++) m += s.wtot[w];
        __syncthreads();
    }
    const uint32_t Htot = m > 0 ? s.scanR[m - 1] : 0;
    /* undivided nodes keep their relative order behind all new children */
    for (int e = 2 * tid; e < 2 * tid + 2; e++) s.scanL[e] = (e < nn && s.rank[e] >= m) ? 1u : 0u;
    __syncthreads();
    block_scan_incl<NN>(s.scanL, s);
    for (int e = 2 * tid; e < 2 * tid + 2; e++)
        if (e < nn && s.rank[e] >= m) {
            int p = (int)Htot + (int)s.scanL[e] - 1;
            s.L.x0[nxt][p] = s.L.x0[cur][e];
            s.L.y0[nxt][p] = s.L.y0[cur][e];
            s.L.x1[nxt][p] = s.L.x1[cur][e];
            s.L.y1[nxt][p] = s.L.y1[cur][e];
            s.L.start[nxt][p] = s.L.start[cur][e];
            s.L.cnt[nxt][p] = s.L.cnt[cur][e];
            s.L.born[nxt][p] = s.L.born[cur][e];
        }
    /* number of new children with more than one key (nToExpand, :573..:606) */
    {
        uint32_t mine = 0;
        for (int r = 2 * tid; r < 2 * tid + 2; r++)
            if (r < m) {
                const uint32_t *c = s.ccnt[s.rnode[r]];
                mine += (c[0] > 1) + (c[1] > 1) + (c[2] > 1) + (c[3] > 1);
            }
        uint32_t incl = wave_incl_scan(mine, lane);
        __syncthreads(); /* wtot reuse */
        if (lane == 63) s.wtot[wave] = incl;
        __syncthreads();
        uint32_t t = 0;
        for (int w = 0; w < QT_W; w++) t += s.wtot[w];
        *n_to_expand = (int)t;
    }
    /* ---- D: stable 4-way partition of every divided node + creation of its children ---- */
    for (int r = wave; r < m; r += QT_W) {
        const int i = s.rnode[r];
        const uint32_t cv = s.L.cnt[cur][i];
        const uint32_t b = BUF(cv);
        const uint32_t st = s.L.start[cur][i];
        const uint32_t *src = (b ? keys1 : keys0) + st;
        uint32_t *dst = (b ? keys0 : keys1) + st;
        const int n = (int)CNT(cv);
        const int X0 = s.L.x0[cur][i], Y0 = s.L.y0[cur][i], X1 = s.L.x1[cur][i], Y1 = s.L.y1[cur][i];
        const int mx = X0 + ((X1 - X0 + 1) >> 1), my = Y0 + ((Y1 - Y0 + 1) >> 1);
        const uint32_t c0 = s.ccnt[i][0], c1 = s.ccnt[i][1], c2 = s.ccnt[i][2], c3 = s.ccnt[i][3];
        uint32_t o0 = 0, o1 = c0, o2 = c0 + c1, o3 = c0 + c1 + c2;
        const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        for (int j0 = 0; j0 < n; j0 += 64 * 8) {
            uint32_t kk[8];
#pragma unroll
            for (int u = 0; u < 8; u++) kk[u] = (j0 + u * 64 + lane < n) ? src[j0 + u * 64 + lane] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (j0 + u * 64 >= n) break; /* wave-uniform */
                const uint32_t key = kk[u];
                const int q = (j0 + u * 64 + lane < n) ? quadrant(key, mx, my) : -1;
                const uint64_t b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
                if (q == 0) dst[o0 + __popcll(b0 & lt)] = key;
                if (q == 1) dst[o1 + __popcll(b1 & lt)] = key;
                if (q == 2) dst[o2 + __popcll(b2 & lt)] = key;
                if (q == 3) dst[o3 + __popcll(b3 & lt)] = key;
                o0 += __popcll(b0);
                o1 += __popcll(b1);
                o2 += __popcll(b2);
                o3 += __popcll(b3);
            }
        }
        if (lane < 4) {
            const int q = lane;
            const uint32_t cq = (q == 0) ? c0 : (q == 1) ? c1 : (q == 2) ? c2 : c3;
            if (cq > 0) {
                /* children are pushed to the FRONT in the order n1..n4, so n4 ends up first */
                int after = ((q < 1 && c1 > 0) ? 1 : 0) + ((q < 2 && c2 > 0) ? 1 : 0) + ((q < 3 && c3 > 0) ? 1 : 0);
                int p = (int)(Htot - s.scanR[r]) + after;
                s.L.x0[nxt][p] = (short)((q & 1) ? mx : X0);
                s.L.x1[nxt][p] = (short)((q & 1) ? X1 : mx);
                s.L.y0[nxt][p] = (short)((q & 2) ? my : Y0);
                s.L.y1[nxt][p] = (short)((q & 2) ? Y1 : my);
                uint32_t off = (q == 0) ? 0 : (q == 1) ? c0 : (q == 2) ? c0 + c1 : c0 + c1 + c2;
                s.L.start[nxt][p] = st + off;
                s.L.cnt[nxt][p] = cq | ((b ^ 1u) << 31);
                s.L.born[nxt][p] = (uint16_t)(r * 4 + q);
            }
        }
    }
    __syncthreads();
    return (int)Htot + (nn - m);
}

template <int NN> __global__ __launch_bounds__(NN / 2) void k_quadtree(OrbGeom g, const CellDesc *cells, const uint32_t *slab,
                                                    const int32_t *cell_cnt, uint32_t *keysA, uint32_t *keysB,
                                                    int32_t *cand_cnt, uint32_t *lvl_kp, int32_t *lvl_cnt, int lds_keys) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qt_smem[];
    QtShared<NN> &s = *reinterpret_cast<QtShared<NN> *>(qt_smem);
    const int level = blockIdx.x, frame = blockIdx.y;
    const LevelGeom &L = g.lv[level];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); /* uniform: scalar unit */
    const int N = L.quota;
    uint32_t *k0 = keysA + (int64_t)frame * g.key_frame_count + L.key_off;
    uint32_t *k1 = keysB + (int64_t)frame * g.key_frame_count + L.key_off;
    const uint32_t *fslab = slab + (int64_t)frame * g.slab_frame_count;
    const int32_t *ccnt = cell_cnt + (int64_t)frame * g.total_cells + L.cell_base;
    const int ncell = L.n_cells;

    /* ---- where the keys live: LDS when the launch reserved room for them (few frames per call = latency mode:
       every pass then costs LDS round trips instead of global ones) and this level's candidates fit, else the
       global ping-pong arrays.  The pointers are generic, the passes do not care. ---- */
    {
        uint32_t mine = 0;
        for (int c = tid; c < ncell; c += QT_T) mine += (uint32_t)ccnt[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        if (lane == 0) s.wtot[wave] = mine;
        __syncthreads();
        uint32_t total = 0;
        for (int w = 0; w < QT_W; w++) total += s.wtot[w];
        __syncthreads();
        if (lds_keys > 0 && total <= (uint32_t)lds_keys) {
            k0 = reinterpret_cast<uint32_t *>(qt_smem + ((sizeof(QtShared<NN>) + 15) & ~(size_t)15));
            k1 = k0 + lds_keys;
        }
    }
    /* ---- gather the per-cell FAST output into one array, cell-row-major (:777-783): per chunk of QT_T cells an
       exclusive scan of the cell counts, then a flat copy -- key j of the chunk finds its cell by binary search
       over the scan, four independent loads in flight per thread ---- */
    uint32_t run = 0;
    for (int c0 = 0; c0 < ncell; c0 += QT_T) {
        const int c = c0 + tid;
        const uint32_t v = (c < ncell) ? (uint32_t)ccnt[c] : 0u;
        const uint32_t soff = (c < ncell) ? (uint32_t)cells[L.cell_base + c].slab_off : 0u;
        uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) s.wtot[wave] = incl;
        __syncthreads();
        uint32_t base = 0, tot = 0;
        for (int w = 0; w < QT_W; w++) {
            if (w < wave) base += s.wtot[w];
            tot += s.wtot[w];
        }
        s.scanL[tid] = base + incl - v; /* exclusive prefix inside this chunk */
        s.scanR[tid] = soff;
        __syncthreads();
        {
            const int nc = min(QT_T, ncell - c0);
            for (uint32_t j0 = tid; j0 < tot; j0 += QT_T * 4) {
                uint32_t val[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t j = j0 + u * QT_T;
                    val[u] = 0;
                    if (j < tot) {
                        int lo = 0, hi = nc - 1; /* largest cell whose exclusive prefix is <= j */
                        while (lo < hi) {
                            const int mid = (lo + hi + 1) >> 1;
                            if (s.scanL[mid] <= j)
                                lo = mid;
                            else
                                hi = mid - 1;
                        }
                        val[u] = fslab[s.scanR[lo] + (j - s.scanL[lo])];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t j = j0 + u * QT_T;
                    if (j < tot) k0[run + j] = val[u];
                }
            }
        }
        run += tot;
        __syncthreads();
    }
    const int n = (int)run;
    if (tid == 0) cand_cnt[(int64_t)frame * g.nlevels + level] = n;
    __syncthreads(); /* k0 writes visible to the whole workgroup */

    /* ---- root nodes (:493-534) ---- */
    const int nIni = L.nIni;
    int nn = 0;
    int cur = 0;
    if (n > 0 && nIni >= 1) {
        if (nIni == 1) {
            if (tid == 0) {
                s.L.x0[0][0] = 0;
                s.L.y0[0][0] = 0;
                s.L.x1[0][0] = (short)(int)(L.hX * 1.0f);
                s.L.y1[0][0] = (short)(L.maxBY - ORB_MIN_BORDER);
                s.L.start[0][0] = 0;
                s.L.cnt[0][0] = (uint32_t)n; /* buffer 0 */
                s.L.born[0][0] = 0;
            }
            nn = 1;
        } else {
            /* stable partition of the keys by root index (int)(x / hX) into buffer 1 */
            uint32_t outBase = 0;
            for (int r = 0; r < nIni; r++) {
                uint32_t cntR = 0;
                for (int j0 = 0; j0 < n; j0 += QT_T) {
                    const int j = j0 + tid;
                    uint32_t key = 0;
                    bool in = false;
                    if (j < n) {
                        key = k0[j];
                        in = ((int)((float)KEY_X(key) / L.hX)) == r;
                    }
                    const uint64_t bal = __ballot(in);
                    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                    if (lane == 0) s.wtot[wave] = (uint32_t)__popcll(bal);
                    __syncthreads();
                    uint32_t base = 0, tot = 0;
                    for (int w = 0; w < QT_W; w++) {
                        if (w < wave) base += s.wtot[w];
                        tot += s.wtot[w];
                    }
                    if (in) k1[outBase + cntR + base + __popcll(bal & lt)] = key;
                    cntR += tot;
                    __syncthreads();
                }
                if (cntR > 0) { /* empty roots are erased (:529-530) */
                    if (tid == 0) {
                        s.L.x0[0][nn] = (short)(int)(L.hX * (float)r);
                        s.L.x1[0][nn] = (short)(int)(L.hX * (float)(r + 1));
                        s.L.y0[0][nn] = 0;
                        s.L.y1[0][nn] = (short)(L.maxBY - ORB_MIN_BORDER);
                        s.L.start[0][nn] = outBase;
                        s.L.cnt[0][nn] = cntR | (1u << 31);
                        s.L.born[0][nn] = (uint16_t)r;
                    }
                    nn++;
                }
                outBase += cntR;
            }
        }
        __syncthreads();
        /* ---- division passes (:540-690) ---- */
        bool finish = false;
        int guard = 0;
        while (!finish && guard++ < 64) {
            int prevSize = nn, nToExpand = 0;
            nn = qt_pass<NN>(s, cur, nn, N, 0, k0, k1, &nToExpand);
            cur ^= 1;
            if (nn >= N || nn == prevSize) {
                finish = true;
            } else if (nn + nToExpand * 3 > N) {
                while (!finish && guard++ < 64) {
                    prevSize = nn;
                    int dummy;
                    nn = qt_pass<NN>(s, cur, nn, N, 1, k0, k1, &dummy);
                    cur ^= 1;
                    if (nn >= N || nn == prevSize) finish = true;
                }
            }
        }
    }
    /* ---- keep the best key of every node, first one on ties (:697-715) ---- */
    uint32_t *out = lvl_kp + (int64_t)frame * g.total_out_cap + L.out_base;
    for (int i = wave; i < nn; i += QT_W) {
        const uint32_t cv = s.L.cnt[cur][i];
        const uint32_t *src = (BUF(cv) ? k1 : k0) + s.L.start[cur][i];
        const int cn = (int)CNT(cv);
        uint32_t best = 0; /* (response << 22) | (0x3FFFFF - position) */
        for (int j = lane; j < cn; j += 64) {
            uint32_t v = ((uint32_t)KEY_R(src[j]) << 22) | (0x3FFFFFu - (uint32_t)j);
            best = v > best ? v : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            uint32_t t = __shfl_xor(best, o);
            best = t > best ? t : best;
        }
        if (lane == 0) out[i] = src[0x3FFFFFu - (best & 0x3FFFFFu)];
    }
    if (tid == 0) lvl_cnt[(int64_t)frame * g.nlevels + level] = nn;
}

template <int NN> static hipError_t launch_qt(const OrbGeom &g, const OrbBuffers &b, int n_frames, hipStream_t st) {
    dim3 grid(g.nlevels, n_frames);
    /* few frames per call = latency mode: reserve LDS for the key ping-pong of a level (a big batch keeps the keys
       in global memory -- 2048 workgroups holding 100 KB of LDS each would starve the kernels running beside them) */
    const size_t fixed = (sizeof(QtShared<NN>) + 15) & ~(size_t)15;
    const size_t room = fixed < 128 * 1024 ? 128 * 1024 - fixed : 0; /* keep the workgroup under 128 KB of LDS */
    static const int batch_keys = getenv("ORBSLAM_AMD_QT_LDS_KEYS") ? atoi(getenv("ORBSLAM_AMD_QT_LDS_KEYS")) : 0;
    const int lds_keys = n_frames <= 8 ? (int)std::min<size_t>(12288, room / 8) : (int)std::min<size_t>(batch_keys, room / 8);
    const size_t lds = ((sizeof(QtShared<NN>) + 15) & ~(size_t)15) + (size_t)lds_keys * 8;
    static PerDeviceLdsLimit lim2;
    {
        hipError_t e = lim2.ensure((const void *)k_quadtree<NN>, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_quadtree<NN>, grid, dim3(NN / 2), lds, st, g, b.cells, (const uint32_t *)b.slab,
                       (const int32_t *)b.cell_cnt, b.keysA, b.keysB, b.cand_cnt, b.lvl_kp, b.lvl_cnt, lds_keys);
    return hipGetLastError();
}

hipError_t launch_quadtree(const OrbGeom &g, const OrbBuffers &b, int n_frames, hipStream_t st) {
    int need = 0;
    for (int l = 0; l < g.nlevels; l++) need = need > g.lv[l].out_cap ? need : g.lv[l].out_cap;
    /* few frames per call (latency mode): the widest workgroup -- 8 wavefronts walk the nodes of a pass side by
       side, 0.22 -> 0.11 ms for one 640x480 frame; a big batch has thousands of workgroups in flight and does
       better with the narrow ones */
    if (need > 1024) return launch_qt<2048>(g, b, n_frames, st); /* > 1020 features on one level: the 1024-thread variant */
    /* up to 32 frames per launch (8 levels x 32 = 256 workgroups = one per CU) the wide variant wins -- 64 frames of
       1280x720 per rank, BASELINE configs[3]: 43.0 -> 57.0 k frames/s; at 128 frames per launch it loses (61.0 -> 57.3) */
    static const int wide_max = getenv("ORBSLAM_AMD_QT_WIDE_MAX") ? atoi(getenv("ORBSLAM_AMD_QT_WIDE_MAX")) : 32;
    if (n_frames <= wide_max) return launch_qt<1024>(g, b, n_frames, st);
    if (need <= 256) return launch_qt<256>(g, b, n_frames, st);
    if (need <= 512) return launch_qt<512>(g, b, n_frames, st);
    return launch_qt<1024>(g, b, n_frames, st);
}
