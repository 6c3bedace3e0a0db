/*
 * ba_api.hip -- bundle adjustment on gfx950: Optimizer::BundleAdjustment / LocalBundleAdjustment /
 * PoseOptimization numerical core behind ba_solve*() (include/orbslam_amd.h).
 *
 * What the reference does (src/Optimizer.cpp:23-129): build a g2o graph (SE3 pose vertices, XYZ point
 * vertices marginalised, EdgeSE3ProjectXYZ with Huber kernel) and run g2o's Levenberg-Marquardt with a
 * Schur-complement block solver.  Here the same algorithm runs on the device, all in f64:
 *
 *   ONE 512-thread workgroup per BA window, one kernel launch per LM iteration for the whole batch of
 *   windows (the host only checks pbStopFlag between launches and never touches per-iteration data).
 *   Inside the launch, per window:
 *     residuals + robust chi2 -> per-edge 2x3 / 2x6 Jacobian blocks -> Hll (3x3, one thread per point,
 *     its edges are contiguous), W = Hpl blocks (6x3 per edge), Hpp (6x6, one wavefront per camera with
 *     shuffle reductions) -> LM trials: Dinv = (Hll + lambda I)^-1, Y = W Dinv, block-sparse reduced
 *     camera system S = Hpp + lambda I - sum Y W^T (one wavefront per non-zero 6x6 block, fixed
 *     contribution order: no atomics, bit-reproducible), g = bp - sum Y bl -> dense blocked Cholesky of S
 *     inside its envelope (ba_chol_solve: 16 x 16 tiles, panel solve and trailing update on
 *     v_mfma_f64_16x16x4_f64) -- or PCG with block-Jacobi preconditioner (p, r in LDS, row-per-thread
 *     block-CSR product) when the LDS panel does not fit or ORBSLAM_AMD_BA_SOLVER=pcg asks for it ->
 *     back-substitution -> exp-map update -> trial chi2 -> gain ratio / lambda update exactly as
 *     OptimizationAlgorithmLevenberg.
 *
 * MFMA note: on gfx950 the f64 matrix rate equals the f64 vector rate (78.6 TFLOP/s both); the matrix cores
 * are used where the work is GEMM-shaped (the Cholesky panel/trailing products: 64 FMAs per issued
 * instruction), not for the matrix-vector products of PCG, which stream S from L2.
 *
 * No CPU fallback: the host code below only sorts indices and moves buffers.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/orbslam_amd.h"
#include "orb_common.h" /* PerDeviceLdsLimit */
void orb_trace_exit_late(); /* orb_api.hip: exit-handler marker for tools/exit_hang_probe.sh */

#define BA_T 512
#define BA_W (BA_T / 64)
#define BA_MAX_N 1536 /* reduced system size (6 * free cameras) whose PCG vectors fit the static LDS arrays */
#define BA_GMAX 64    /* most workgroups that may team up on one window */
#define BA_CHUNK_E 368 /* Schur staging: edges (W records, 144 B) of one point chunk held in LDS; sized so that TWO
                          workgroups' tiles (BA_CHUNK_DOUBLES: 77.7 KB) fit the 160 KB of a CU (k_ba_ph_schur2) */
#define BA_CHUNK_P 96  /* ... and its points (Dinv 72 B, tv 24 B) */
#define BA_CHUNK_C 2560 /* ... and its contribution list (packed local edges + point, ints) */
#define BA_CHUNK_B 256  /* ... and the blocks it touches (id, list start) */
#define BA_CH_DESC 16   /* ints per chunk descriptor */
/* an entry of a chunk's block list: block id, plus two flags for the two-part Schur pass of the phase-kernel path --
   FIRST: no earlier chunk of the same part touches this block, so the part does not read the block but starts it (at
   Hpp + lambda I for a diagonal block of part 0, at zero otherwise) -- which is what spares that path the
   initialisation pass over S and the zero fill of S2 (360 MB of 600 per trial and 256 windows); DIAG: a diagonal block */
#define BA_BLK_FIRST (1 << 30)
#define BA_BLK_DIAG (1 << 29)
#define BA_BLK_MASK ((1 << 29) - 1)
#define BA_SCHUR_PARTS 2
#define BA_CHUNK_DOUBLES (BA_CHUNK_E * 18 + BA_CHUNK_P * 12 + BA_CHUNK_E / 2 + BA_CHUNK_C / 2 + BA_CHUNK_B + BA_CHUNK_E / 2 + 32)

static OrbErrSlot b_err; /* forwards to orb_last_error() */
#define BCHK(expr)                                                          \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) {                                             \
            b_err = std::string(#expr) + ": " + hipGetErrorString(_e);      \
            return ORB_E_HIP;                                               \
        }                                                                   \
    } while (0)

/* per-window descriptor: element offsets into the batch buffers */
struct BAWin {
    int Nc, Np, E, Ncf, nnzb, nub; /* nub: blocks on or above the diagonal */
    int ch_pm;                     /* rows of the Cholesky LDS panel: 16 x the widest envelope column (tile rows below a diagonal tile) */
    double fx, fy, cx, cy, delta, chi2_thr;
    /* doubles */
    int64_t poses, poses_t, pts, pts_t, obs, info, err, err_t, W, tv, Hpp, bp, Hll, bl, Dinv, S, g, Minv, x, xl, st;
    int64_t A, Linv; /* dense reduced system (np x np) and inverse diagonal blocks of its Cholesky factor */
    int64_t fplan;   /* doubles offset (scratch), read as ints: fill plan of the band Cholesky (k_ba_ph_fillplan); -1: none */
    int64_t Uf, tu;  /* per point: the upper-triangular U with Dinv = U U^T (6 doubles) and u = U^T bl (phase-kernel Schur pass) */
    int64_t Rc;      /* rotation matrices of the cameras (9 doubles each), written by the camera pass of the phase-kernel path */
    int64_t S2, g2;  /* second accumulator of the Schur products and of g (phase-kernel path: the window's chunks are split
                        over two workgroups; the Cholesky assembly adds S and S2, g and g2); chunked windows only */
    /* ints */
    int64_t cam, pt, pt_start, cam_start, cam_edges, slot, blk_row, blk_col, blk_diag, blk_mirror, blk_upper, blk_rowid, con_start, con_e1, con_e2;
    int64_t ch_first, ch_last; /* Cholesky envelope per 16-row tile: first tile column of tile row i (non-decreasing), last tile row that reaches tile column k */
    /* batch path of the Schur complement: point chunks (BA_CH_DESC ints each: edge range, point range, block range, camera
       range, start and length of the chunk's contribution list and of its camera edge list), per chunk the blocks it touches with their contribution lists (local edges a, b and local point p packed a | b << 9 | p << 18) and
       the free cameras it touches with their local edge lists; nchunks == 0: list path only */
    int nchunks, pad_;
    int64_t ch_desc, cb_blk, cb_start, cb_con, cg_slot, cg_start, cg_edge;
    int64_t blk_touch;         /* ints, per block: bit p set = part p of the two-part Schur pass touches (and starts) the block */
    int64_t gbar;              /* ints: two arrival counters of the window's workgroup team (alternating per launch) */
    int64_t gred;              /* doubles (scratch): 2 slots x 2 values x BA_GMAX partial sums of the team */
    int64_t pcgv;              /* doubles (scratch): x, r, z, p, Ap of a PCG whose vectors exceed the LDS arrays (n > BA_MAX_N) */
    /* bytes */
    int64_t flags; /* per edge: bit0 active, bit1 robust */
};
/* LM state (doubles at D[st + i]) */
enum { ST_LAMBDA = 0, ST_NI, ST_CHI, ST_CHI0, ST_ITERS, ST_SOLVES, ST_TERM, ST_ROUND_IT, ST_HAVE_CHI0, ST_PCG_ITERS,
       ST_ABORT, /* a team barrier gave up waiting (never expected; the host turns it into an error) */
       /* state machine of the phase-kernel batch path (k_ba_ph_*): where the window stands in its optimize() round */
       ST_PH,      /* 0: an LM iteration starts (build), 1: same iteration, next trial (solve again), 2: round over */
       ST_QMAX, ST_HAVECHI, ST_OK2, ST_BUDGET, ST_FIRST, ST_ACCEPT,
       ST_T0, /* 10 phase timers + 6 sub-timers of the chunked Schur pass (profile mode), 100 MHz ticks */
       ST_COUNT = ST_T0 + 24 };

struct BABufs {
    double *D;
    int32_t *I;
    uint8_t *B;
    const BAWin *wins;
};

/* ---------------------------------------------------------------------------------------------- */
__device__ __forceinline__ void quat_to_R(const double *q, double R[9]) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ __forceinline__ void quat_normalize(double q[4]) {
    if (q[0] < 0) {
        q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3];
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
__device__ __forceinline__ void R_to_quat(const double R[9], double q[4]) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (R[7] - R[5]) * t;
        q[2] = (R[2] - R[6]) * t;
        q[3] = (R[3] - R[1]) * t;
    } else {
        /* Eigen's branch on the largest diagonal entry, written out for the three cases so that every index is a
           constant and R[] stays in registers (run-time indices put it in scratch memory) */
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > (i == 0 ? R[0] : R[4])) i = 2;
#define R2Q_CASE(I, J, K)                                        \
    {                                                            \
        t = sqrt(R[I * 4] - R[J * 4] - R[K * 4] + 1.0);          \
        const double vi = 0.5 * t;                               \
        t = 0.5 / t;                                             \
        q[0] = (R[K * 3 + J] - R[J * 3 + K]) * t;                \
        q[1 + I] = vi;                                           \
        q[1 + J] = (R[J * 3 + I] + R[I * 3 + J]) * t;            \
        q[1 + K] = (R[K * 3 + I] + R[I * 3 + K]) * t;            \
    }
        if (i == 0)
            R2Q_CASE(0, 1, 2)
        else if (i == 1)
            R2Q_CASE(1, 2, 0)
        else
            R2Q_CASE(2, 0, 1)
#undef R2Q_CASE
    }
}
/* pose_out = SE3Quat::exp(u) * pose_in   (VertexSE3Expmap::oplusImpl) */
__device__ __forceinline__ void se3_exp_mul(const double u[6], const double *pin, double *pout) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i];
            V[i] = R[i];
        }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta),
                     c = (theta - sin(theta)) / (theta * theta * theta);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i];
        }
    }
    double qe[4], te[3];
    R_to_quat(R, qe);
    quat_normalize(qe);
#pragma unroll
    for (int i = 0; i < 3; i++) te[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    double RE[9];
    quat_to_R(qe, RE);
    const double *qi = pin, *ti = pin + 4;
    double qo[4];
    qo[0] = qe[0] * qi[0] - qe[1] * qi[1] - qe[2] * qi[2] - qe[3] * qi[3];
    qo[1] = qe[0] * qi[1] + qe[1] * qi[0] + qe[2] * qi[3] - qe[3] * qi[2];
    qo[2] = qe[0] * qi[2] - qe[1] * qi[3] + qe[2] * qi[0] + qe[3] * qi[1];
    qo[3] = qe[0] * qi[3] + qe[1] * qi[2] - qe[2] * qi[1] + qe[3] * qi[0];
    quat_normalize(qo);
#pragma unroll
    for (int i = 0; i < 4; i++) pout[i] = qo[i];
#pragma unroll
    for (int i = 0; i < 3; i++) pout[4 + i] = te[i] + RE[i * 3] * ti[0] + RE[i * 3 + 1] * ti[1] + RE[i * 3 + 2] * ti[2];
}
/* The same update for the pose-only optimisation, whose 40 LM iterations each wait for it (k_pose_opt: solve + exp map were
 * half of an iteration).  Same mathematics, shorter dependent chain: no f64 square root or division (v_rsq_f64 / v_rcp-free
 * reciprocals by one cubically convergent step), ONE sincos -- of the half angle -- from which the quaternion of exp(omega)
 * follows directly (cos(t/2), sin(t/2) omega / t) instead of through a rotation matrix, Eigen's matrix-to-quaternion branch,
 * a normalisation and back to a matrix; sin t = 2 s c and 1 - cos t = 2 s^2 feed R and V.  Differs from se3_exp_mul (which
 * follows g2o's operation sequence, and stays in the BA kernels) by rounding only. */
__device__ __forceinline__ double po_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-(x * r), r, 1.0);
    return __builtin_fma(r, __builtin_fma(0.375, e, 0.5) * e, r);
}
__device__ __forceinline__ void se3_exp_mul_fast(const double u[6], const double *pin, double *pout) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta2 = wx * wx + wy * wy + wz * wz;
    if (theta2 < 1e-10) { /* the small-angle branch of SE3Quat::exp (theta < 1e-5): rare, the general code */
        se3_exp_mul(u, pin, pout);
        return;
    }
    const double it = po_rsqrt(theta2), theta = theta2 * it, it2 = it * it;
    double sh, ch;
    sincos(0.5 * theta, &sh, &ch);
    const double st = 2.0 * sh * ch, omc = 2.0 * sh * sh; /* sin t, 1 - cos t */
    const double a = st * it, b = omc * it2, c = (theta - st) * it2 * it;
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double RE[9], V[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        RE[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i];
        V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i];
    }
    const double k = sh * it; /* sin(t/2) / t; cos(t/2) > 0 for every step an LM iteration takes (t < pi) */
    double qe[4] = {ch, k * wx, k * wy, k * wz};
    if (qe[0] < 0) {
        qe[0] = -qe[0]; qe[1] = -qe[1]; qe[2] = -qe[2]; qe[3] = -qe[3];
    }
    double te[3];
#pragma unroll
    for (int i = 0; i < 3; i++) te[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    const double *qi = pin, *ti = pin + 4;
    double qo[4];
    qo[0] = qe[0] * qi[0] - qe[1] * qi[1] - qe[2] * qi[2] - qe[3] * qi[3];
    qo[1] = qe[0] * qi[1] + qe[1] * qi[0] + qe[2] * qi[3] - qe[3] * qi[2];
    qo[2] = qe[0] * qi[2] - qe[1] * qi[3] + qe[2] * qi[0] + qe[3] * qi[1];
    qo[3] = qe[0] * qi[3] + qe[1] * qi[2] - qe[2] * qi[1] + qe[3] * qi[0];
    if (qo[0] < 0) {
        qo[0] = -qo[0]; qo[1] = -qo[1]; qo[2] = -qo[2]; qo[3] = -qo[3];
    }
    const double rn = po_rsqrt(qo[0] * qo[0] + qo[1] * qo[1] + qo[2] * qo[2] + qo[3] * qo[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) pout[i] = qo[i] * rn;
#pragma unroll
    for (int i = 0; i < 3; i++) pout[4 + i] = te[i] + RE[i * 3] * ti[0] + RE[i * 3 + 1] * ti[1] + RE[i * 3 + 2] * ti[2];
}
__device__ __forceinline__ void project_err(const double *pose, const double *X, const double *ob, double fx, double fy,
                                            double cx, double cy, double e[2], double pc[3]) {
    double R[9];
    quat_to_R(pose, R);
    pc[0] = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + pose[4];
    pc[1] = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + pose[5];
    pc[2] = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + pose[6];
    e[0] = ob[0] - (pc[0] / pc[2] * fx + cx);
    e[1] = ob[1] - (pc[1] / pc[2] * fy + cy);
}
__device__ __forceinline__ double huber_rho0(double e, double delta) {
    const double d2 = delta * delta;
    return e <= d2 ? e : 2 * sqrt(e) * delta - d2;
}
__device__ __forceinline__ double huber_w(double e, double delta) {
    const double d2 = delta * delta;
    return e <= d2 ? 1.0 : delta / sqrt(e);
}
/* EdgeSE3ProjectXYZ::linearizeOplus: Ji = d e/d point (2x3), Jj = d e/d pose (2x6) */
__device__ __forceinline__ void jacobians(const double *pose, const double pc[3], double fx, double fy, double Ji[6],
                                          double Jj[12]) {
    double R[9];
    quat_to_R(pose, R);
    const double x = pc[0], y = pc[1], z = pc[2], z_2 = z * z;
    const double t0 = fx, t2 = -x / z * fx, t4 = fy, t5 = -y / z * fy;
    for (int c = 0; c < 3; c++) {
        Ji[c] = -1. / z * (t0 * R[c] + t2 * R[6 + c]);
        Ji[3 + c] = -1. / z * (t4 * R[3 + c] + t5 * R[6 + c]);
    }
    Jj[0] = x * y / z_2 * fx;
    Jj[1] = -(1 + (x * x / z_2)) * fx;
    Jj[2] = y / z * fx;
    Jj[3] = -1. / z * fx;
    Jj[4] = 0;
    Jj[5] = x / z_2 * fx;
    Jj[6] = (1 + y * y / z_2) * fy;
    Jj[7] = -x * y / z_2 * fy;
    Jj[8] = -x / z * fy;
    Jj[9] = 0;
    Jj[10] = -1. / z * fy;
    Jj[11] = y / z_2 * fy;
}

/* The phase-kernel batch path does not store W = Hpl (18 doubles per edge, written once and read twice per LM trial:
 * with 256 windows in flight that is three quarters of the solver's HBM traffic).  It stores what W is a function of
 * besides the camera's rotation: x/z, y/z, 1/z of the point in the camera frame and the edge's (robustified) weight --
 * 32 bytes -- and every consumer rebuilds the 6 x 3 block from the record and R (no division, ~110 flops). */
__device__ __forceinline__ void ba_w_from_record(const double rec[4], const double *R, double fx, double fy, double W[18]) {
    const double xz = rec[0], yz = rec[1], iz = rec[2], wgt = rec[3];
    double Ji[6], Jj[12];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        Ji[c] = -iz * (fx * R[c] - xz * fx * R[6 + c]);
        Ji[3 + c] = -iz * (fy * R[3 + c] - yz * fy * R[6 + c]);
    }
    Jj[0] = xz * yz * fx;
    Jj[1] = -(1 + xz * xz) * fx;
    Jj[2] = yz * fx;
    Jj[3] = -iz * fx;
    Jj[4] = 0;
    Jj[5] = xz * iz * fx;
    Jj[6] = (1 + yz * yz) * fy;
    Jj[7] = -xz * yz * fy;
    Jj[8] = -xz * fy;
    Jj[9] = 0;
    Jj[10] = -iz * fy;
    Jj[11] = yz * iz * fy;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) W[a * 3 + b] = wgt * (Jj[a] * Ji[b] + Jj[6 + a] * Ji[3 + b]);
}

struct BAShared {
    double vp[BA_MAX_N]; /* PCG search direction p */
    double vr[BA_MAX_N]; /* residual r             */
    double red[BA_W];
    double bc[8];
};

template <class SH> __device__ __forceinline__ double block_sum(double v, SH &s) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* uniform: scalar unit */
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) s.red[wave] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < BA_W; w++) t += s.red[w];
    return t;
}
template <class SH> __device__ __forceinline__ double block_max(double v, SH &s) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* uniform: scalar unit */
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    __syncthreads();
    if (lane == 0) s.red[wave] = v;
    __syncthreads();
    double t = s.red[0];
    for (int w = 1; w < BA_W; w++) t = fmax(t, s.red[w]);
    return t;
}

/* ------------------------------------------------------------------------------------------------------------
 * A TEAM of G workgroups per window (G == 1: the batch path, one workgroup per window, nothing below costs
 * anything).  LocalMapping hands over ONE window per keyframe; with one workgroup that window runs on one of 256
 * CUs.  With a team, every streaming phase (chi2, Hll/W, Hpp, Dinv/Y, Schur blocks, g, back-substitution) is strided
 * over G x 512 threads and the phases are separated by a team barrier: an arrival counter in global memory
 * (agent-scope atomics; release/acquire fences write back / invalidate the XCD-private L2 so that the parts, which
 * land on different XCDs, see each other's data).  The team is launched cooperatively (all G x windows workgroups
 * co-resident), so the barrier cannot deadlock.  Sums over the team are partial sums per workgroup combined in part
 * order: deterministic, and for G == 1 bit-identical to the single-workgroup code. */
struct BATeam {
    int G, part;
    unsigned *bar;   /* this launch's arrival counter */
    unsigned *abort; /* sticky: some barrier of this window timed out */
    unsigned epoch;  /* barriers passed so far in this launch (uniform) */
    double *red;     /* [2 slots][2 values][BA_GMAX] */
    int slot;
    bool dead;       /* uniform: this workgroup has seen the abort word -- barriers degrade to __syncthreads() */
};
#define BA_SPIN_LIMIT (1u << 23) /* x ~0.1 us: about a second -- a healthy barrier completes in microseconds */
/* EXIT CONDITION: the spin is bounded.  If a team member never arrives (it cannot with a cooperative launch, but a
 * kernel must drain whatever happens), the waiting lane raises the window's abort word; every workgroup of the
 * team that sees it stops waiting at this and all later barriers, runs its (bounded) loops to the end on whatever
 * data it has and exits; the host reports ORB_E_HIP for the batch. */
__device__ __forceinline__ void team_sync(BATeam &t, BAShared &s) {
    if (t.G == 1 || t.dead) {
        __syncthreads();
        return;
    }
    __threadfence(); /* release: this thread's stores reach device scope */
    __syncthreads();
    t.epoch++;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(t.bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = t.epoch * (unsigned)t.G;
        unsigned spins = 0;
        bool dead = false;
        while (__hip_atomic_load(t.bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0 &&
                (spins >= BA_SPIN_LIMIT || __hip_atomic_load(t.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(t.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dead = true;
                break;
            }
        }
        s.bc[7] = dead ? 1.0 : 0.0;
    }
    __syncthreads();
    t.dead = s.bc[7] != 0.0;
    __threadfence(); /* acquire: later loads of every wave see the other parts' stores */
}
/* sums of a and b over all threads of the team (b may be unused) */
__device__ __forceinline__ void team_sum2(double &a, double &b, BAShared &s, BATeam &t) {
    a = block_sum(a, s);
    b = block_sum(b, s);
    if (t.G == 1) return;
    double *r = t.red + t.slot * 2 * BA_GMAX;
    t.slot ^= 1;
    if (threadIdx.x == 0) {
        r[t.part] = a;
        r[BA_GMAX + t.part] = b;
    }
    team_sync(t, s);
    double ta = 0, tb = 0;
    for (int g = 0; g < t.G; g++) {
        ta += __hip_atomic_load(r + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tb += __hip_atomic_load(r + BA_GMAX + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    a = ta;
    b = tb;
}
__device__ __forceinline__ double team_sum(double v, BAShared &s, BATeam &t) {
    if (t.G == 1) return block_sum(v, s);
    double z = 0;
    v = block_sum(v, s);
    double *r = t.red + t.slot * 2 * BA_GMAX;
    t.slot ^= 1;
    if (threadIdx.x == 0) r[t.part] = v;
    team_sync(t, s);
    for (int g = 0; g < t.G; g++) z += __hip_atomic_load(r + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return z;
}
__device__ __forceinline__ double team_max(double v, BAShared &s, BATeam &t) {
    v = block_max(v, s);
    if (t.G == 1) return v;
    double *r = t.red + t.slot * 2 * BA_GMAX;
    t.slot ^= 1;
    if (threadIdx.x == 0) r[t.part] = v;
    team_sync(t, s);
    double z = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int g = 1; g < t.G; g++) z = fmax(z, __hip_atomic_load(r + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return z;
}

/* errors of all active edges at (poses, pts) -> err ; returns robust chi2 (activeRobustChi2) */
__device__ double eval_chi2_partial(const BAWin &w, double *D, const int32_t *I, const uint8_t *B, int64_t poses, int64_t pts,
                                    int64_t err, const BATeam &t) {
    constexpr bool STORE = true; /* (the fused / team kernel reads the err array back; the phase kernels' twin, ba_ph_chi2_part, may not store) */
    /* two edges per step, every index / flag / observation load issued before the first dependent gather and none of
       them behind the "inactive edge" branch (inactive edges are computed and discarded): a quarter of the dependent
       memory round trips of the plain loop.  Same per-thread summation order (e, e + T, e + 2T, ...). */
    double acc = 0;
    const int TT = t.G * BA_T; /* threads of the team */
    for (int e0 = t.part * BA_T + threadIdx.x; e0 < w.E; e0 += 2 * TT) {
        const int eA = e0, eB = (e0 + TT < w.E) ? e0 + TT : e0;
        const bool vB = e0 + TT < w.E;
        const uint8_t fA = B[w.flags + eA], fB = B[w.flags + eB];
        const int cA = I[w.cam + eA], cB = I[w.cam + eB], pA = I[w.pt + eA], pB = I[w.pt + eB];
        const double iA = D[w.info + eA], iB = D[w.info + eB];
        double erA[2], erB[2], pc[3];
        project_err(D + poses + (int64_t)cA * 7, D + pts + (int64_t)pA * 3, D + w.obs + (int64_t)eA * 2, w.fx, w.fy, w.cx, w.cy, erA,
                    pc);
        project_err(D + poses + (int64_t)cB * 7, D + pts + (int64_t)pB * 3, D + w.obs + (int64_t)eB * 2, w.fx, w.fy, w.cx, w.cy, erB,
                    pc);
        if (fA & 1) {
            if (STORE) {
                D[err + (int64_t)eA * 2] = erA[0];
                D[err + (int64_t)eA * 2 + 1] = erA[1];
            }
            double c = (erA[0] * erA[0] + erA[1] * erA[1]) * iA;
            if (fA & 2) c = huber_rho0(c, w.delta);
            acc += c;
        }
        if (vB && (fB & 1)) {
            if (STORE) {
                D[err + (int64_t)eB * 2] = erB[0];
                D[err + (int64_t)eB * 2 + 1] = erB[1];
            }
            double c = (erB[0] * erB[0] + erB[1] * erB[1]) * iB;
            if (fB & 2) c = huber_rho0(c, w.delta);
            acc += c;
        }
    }
    return acc;
}


/* ------------------------------------------------------------------------------------------------------------
 * Dense blocked Cholesky solve of the reduced camera system S x = g, in-kernel, one workgroup per window
 * (what g2o's LinearSolverEigen computes with a sparse Cholesky).  Block size 16:
 *   diagonal 16x16 block  : factorised by one wavefront in LDS (in-order LDS ops of a single wave), plus its
 *                           inverse, so that the panel solve becomes a matrix product;
 *   panel  L21 = A21 L11^-T and trailing update A22 -= L21 L21^T : v_mfma_f64_16x16x4_f64 (the GEMM-shaped part
 *                           of the reduced camera system -- the one place of this code base where matrix cores
 *                           fit), operands from an LDS-staged panel with a 17-double row pitch (bank-conflict
 *                           free for ds_read_b64), accumulator tiles read-modify-written in global memory;
 *   forward / backward substitution : blocked, 16 FMAs per vector element and step.
 * f64 MFMA fragment layout (cdna_hip_programming.md section 3): A lane l holds A[l&15][k=l>>4], B lane l holds
 * B[k=l>>4][l&15], C/D lane l register r holds C[(l>>4)+4r][l&15].
 * Touches S about 5x less than a 320-iteration PCG does, which is what bounds the batched throughput. */
typedef double v4d __attribute__((ext_vector_type(4)));
#define CH_LD 17
/* lanes of ONE wavefront exchanging data through LDS: the hardware executes a wave's LDS operations in order,
 * the fences keep the compiler from moving this thread's loads across another lane's stores */
#define CH_WAVE_SYNC()                                        \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

__device__ int g_chol_scalar = 0; /* debug: 1 = plain loops instead of MFMA */

/* the dense np x np matrix of the reduced camera system, from the block-sparse S: zero the envelope, then scatter the
 * blocks on or above the diagonal (the only ones the Schur phase fills) TRANSPOSED -- the factorisation works on the
 * lower triangle -- and pad the diagonal with ones.  Strided over the whole team (gtid of gthreads; rows by gwave of
 * gwaves): disjoint writers, except that a block overwrites zeros of the first loop, hence the barrier in between
 * (the caller supplies it as a callable: __syncthreads() or the team barrier). */
template <bool TWO = false, class Sync>
__device__ __forceinline__ void ba_chol_assemble(const BAWin &w, double *D, const int32_t *I, int n, int gtid, int gthreads, int gwave,
                                                 int gwaves, Sync &&sync) {
    const int np = (n + 15) & ~15, lane = threadIdx.x & 63;
    double *A = D + w.A;
    const int32_t *efirst = I + w.ch_first;
    for (int r = gwave; r < np; r += gwaves) { /* only the envelope [first(tile row) * 16, diagonal tile] is ever read */
        const int c0 = efirst[r >> 4] * 16, c1 = ((r >> 4) + 1) * 16;
        for (int c = c0 + lane; c < c1; c += 64) A[(int64_t)r * np + c] = 0.0;
    }
    sync();
    for (int i = gtid; i < w.nnzb * 36; i += gthreads) {
        const int bk = i / 36, e = i - bk * 36;
        const int br = I[w.blk_rowid + bk], bc = I[w.blk_col + bk];
        if (bc >= br) A[(int64_t)(bc * 6 + e % 6) * np + br * 6 + e / 6] = (TWO && (I[w.blk_touch + bk] & 2)) ? D[w.S + i] + D[w.S2 + i] : D[w.S + i];
    }
    for (int i = n + gtid; i < np; i += gthreads) A[(int64_t)i * np + i] = 1.0; /* identity padding */
}

/* value of `v` in lane `src` (a compile-time constant), through two v_readlane_b32: no LDS round trip */
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
template <bool TWO = false, bool AHEAD = false>
__device__ __noinline__ bool ba_chol_solve(const BAWin &w, double *D, const int32_t *I, int n, double *dyn, double *bad_flag, int debug) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); /* uniform: scalar unit */
    unsigned long long tp = debug ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define CH_STAMP(k)                                                         \
    if (debug) {                                                            \
        __syncthreads();                                                    \
        if (tid == 0) {                                                     \
            const unsigned long long tn = __builtin_amdgcn_s_memrealtime(); \
            D[w.st + ST_T0 + 16 + (k)] += (double)(tn - tp);                \
            tp = tn;                                                        \
        }                                                                   \
    }
    const int np = (n + 15) & ~15, nb = np >> 4;
    double *A = D + w.A;
    double *Li = D + w.Linv;
    double *P = dyn;                        /* panel: ch_pm x CH_LD (only the tile rows that reach a column) */
    double *sd = P + (size_t)w.ch_pm * CH_LD; /* diagonal block, 16 x CH_LD   */
    double *sl = sd + 16 * CH_LD;           /* its inverse, 16 x CH_LD      */
    double *z = sl + 16 * CH_LD;            /* right-hand side / solution   */
    const int32_t *efirst = I + w.ch_first, *elast = I + w.ch_last;
    if (tid == 0) bad_flag[0] = 0.0;
    __syncthreads();
    CH_STAMP(0)
    /* ---- factorisation ---- */
    /* diagonal 16 x 16 tile: Cholesky factor and its inverse by ONE wavefront in registers -- lane r < 16 holds row r,
       a pivot or a column entry travels by v_readlane (lane_bcast), every loop is unrolled so that all register
       indices are static.  The textbook right-looking loop (a[i][k] -= l[i][j] * l[k][j] for j < k <= i), with one
       reciprocal square root per column instead of a division per entry; a third of the time of the LDS version it
       replaces (16 columns x 3 LDS round trips per column there).  The tile comes from memory, or from `sd` when this
       wavefront has just updated it (look-ahead, below); only its inverse is needed outside (the substitutions and the
       panel use L11^-1, never L11 itself). */
    auto factor_diag = [&](int kb, bool from_lds) {
        const int rl = lane & 15;
        double a[16];
        unsigned long long tq0 = 0, tq1 = 0, tq2 = 0, tq3 = 0; /* (profile mode: time split of the tile) */
        if (debug) tq0 = __builtin_amdgcn_s_memrealtime();
        if (from_lds) {
#pragma unroll
            for (int k = 0; k < 16; k++) a[k] = sd[rl * CH_LD + k];
        } else {
            const double *Ar = A + (int64_t)(kb * 16 + rl) * np + kb * 16;
#pragma unroll
            for (int k = 0; k < 16; k++) a[k] = Ar[k];
        }
        if (debug) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tq1 = __builtin_amdgcn_s_memrealtime(); }
        bool bad = false;
        double inv[16]; /* 1 / l[j][j]: the column scaling and the inverse below multiply by it */
#pragma unroll
        for (int j = 0; j < 16; j++) {
            double ajj = lane_bcast(a[j], j);
            if (!(ajj > 0.0) || !isfinite(ajj)) {
                bad = true;
                ajj = 1.0;
            }
            /* 1 / sqrt(ajj) from v_rsq_f64 (about 2^-23) and two Newton steps, the pivot as ajj * that: 7 dependent
               f64 operations where sqrt() followed by a division is about 30 -- and the 16 columns are sequential */
            double r = __builtin_amdgcn_rsq(ajj);
#pragma unroll
            for (int nr = 0; nr < 2; nr++) {
                const double e = __builtin_fma(-(ajj * r), r, 1.0);
                r = __builtin_fma(0.5 * r, e, r);
            }
            inv[j] = r;
            const double pv = ajj * r;
            a[j] = (rl == j) ? pv : a[j] * r; /* rows above j hold don't-care values in column j */
#pragma unroll
            for (int k = j + 1; k < 16; k++) {
                const double lkj = lane_bcast(a[j], k);
                a[k] -= a[j] * lkj;
            }
        }
        if (debug) { asm volatile("" : "+v"(a[15])); tq2 = __builtin_amdgcn_s_memrealtime(); }
        /* column rl of L^-1 by forward substitution: x[i] = (delta - sum_{k<i} l[i][k] x[k]) / l[i][i] */
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double v = (i == rl) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < i) v -= lane_bcast(a[k], i) * x[k];
            x[i] = (i >= rl) ? v * inv[i] : 0.0;
        }
        if (debug) { asm volatile("" : "+v"(x[15])); tq3 = __builtin_amdgcn_s_memrealtime(); }
        if (debug && lane == 0) {
            D[w.st + ST_T0 + 17] += (double)(tq1 - tq0);
            D[w.st + ST_T0 + 14] += (double)(tq2 - tq1);
            D[w.st + ST_T0 + 15] += (double)(tq3 - tq2);
        }
        if (lane < 16) {
            if (!AHEAD) { /* L11 over the tile in memory.  NOTHING reads it (the substitutions and the panel use L11^-1), yet with
                             256 windows in flight the kernel takes 8.7 ms per batch with this store and 10.3 without -- measured
                             both ways on one device, not understood; kept where it pays */
                double *Ar = A + (int64_t)(kb * 16 + rl) * np + kb * 16;
#pragma unroll
                for (int c = 0; c < 16; c++) Ar[c] = (c <= rl) ? a[c] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                sl[i * CH_LD + rl] = x[i];           /* (x[i] is 0 above the diagonal) */
                Li[kb * 256 + i * 16 + rl] = x[i];
            }
        }
        if (bad && lane == 0) bad_flag[0] = 1.0;
    };
    /* panel of column kb: L_i = A_i * L11^-T for the m tile rows below the diagonal that reach the column */
    auto panel = [&](int kb, int m) {
        if (g_chol_scalar) {
            for (int e = tid; e < m * 256; e += BA_T) {
                const int bi = e >> 8, row = (e >> 4) & 15, col = e & 15;
                const int gi = kb + 1 + bi;
                double v = 0;
                for (int k = 0; k < 16; k++) v += A[(int64_t)(gi * 16 + row) * np + kb * 16 + k] * sl[col * CH_LD + k];
                P[(bi * 16 + row) * CH_LD + col] = v;
            }
            __syncthreads();
            for (int e = tid; e < m * 256; e += BA_T) {
                const int bi = e >> 8, row = (e >> 4) & 15, col = e & 15;
                A[(int64_t)((kb + 1 + bi) * 16 + row) * np + kb * 16 + col] = P[(bi * 16 + row) * CH_LD + col];
            }
            return;
        }
        for (int bi = wave; bi < m; bi += BA_W) {
            const int gi = kb + 1 + bi;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const double a = A[(int64_t)(gi * 16 + (lane & 15)) * np + kb * 16 + 4 * s4 + (lane >> 4)];
                const double b = (4 * s4 + (lane >> 4) <= (lane & 15)) ? sl[(lane & 15) * CH_LD + 4 * s4 + (lane >> 4)] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = (lane >> 4) + 4 * r, col = lane & 15;
                A[(int64_t)(gi * 16 + row) * np + kb * 16 + col] = acc[r];
                P[(bi * 16 + row) * CH_LD + col] = acc[r];
            }
        }
    };
    /* trailing update: A[i][j] -= L_i L_j^T for kb < j <= i, tile t of the m (m + 1) / 2; `keep`: the tile (the NEXT
       step's diagonal tile) goes to `sd` instead of memory */
    auto update_tile = [&](int kb, int t, bool keep) {
        int ii = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
        while (ii * (ii + 1) / 2 > t) ii--;
        const int jj = t - ii * (ii + 1) / 2;
        const int gi = kb + 1 + ii, gj = kb + 1 + jj;
        double *C = A + (int64_t)(gi * 16 + (lane >> 4)) * np + gj * 16 + (lane & 15);
        v4d c4;
#pragma unroll
        for (int r = 0; r < 4; r++) c4[r] = C[(int64_t)4 * r * np];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const double a = -P[(ii * 16 + (lane & 15)) * CH_LD + 4 * s4 + (lane >> 4)];
            const double b = P[(jj * 16 + (lane & 15)) * CH_LD + 4 * s4 + (lane >> 4)];
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
        }
        if (keep) {
#pragma unroll
            for (int r = 0; r < 4; r++) sd[((lane >> 4) + 4 * r) * CH_LD + (lane & 15)] = c4[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) C[(int64_t)4 * r * np] = c4[r];
        }
    };
    auto update_scalar = [&](int kb, int m) { /* (debug) */
        for (int e = tid; e < m * 16 * m * 16; e += BA_T) {
            const int r = e / (m * 16), c = e % (m * 16);
            if (c > r) continue;
            double v = 0;
            for (int k = 0; k < 16; k++) v += P[r * CH_LD + k] * P[c * CH_LD + k];
            A[(int64_t)((kb + 1) * 16 + r) * np + (kb + 1) * 16 + c] -= v;
        }
    };
    if (AHEAD) {
        /* LOOK-AHEAD: step s runs the panel and the trailing update of column s and, beside the update, the factorisation
           of diagonal tile s + 1 -- the first tile of the update IS that tile: the wavefront that updates it keeps it
           (through `sd`: the accumulator layout is not the row-per-lane layout of the factorisation) and factorises it at
           once, while the other seven wavefronts finish the update.  The sequential chain of the diagonal tile, half of a
           solve, runs beside the only other phase that needs neither its result nor `sl`: Cholesky 5.3 -> 4.6 ms of a
           100 x 5000 window's solve with ONE window in flight (its matrix resident in L2).  Step -1 is the factorisation
           of tile 0 alone, so that the (several thousand instructions long) tile factorisation is inlined once.
           With 256 windows in flight the look-ahead LOSES (9.95 against 8.75 ms per batch, also in this single-site
           form, so it is not the instruction cache): the kept tile's read-modify-write comes from HBM and lands on
           the chain; the batch instantiation takes the plain loop below. */
        for (int s = -1; s < nb - 1; s++) {
            const int m = s >= 0 ? elast[s] - s : 0; /* tile rows below the diagonal that reach column s */
            if (m > 0) {
                panel(s, m);
                __syncthreads();
            }
            CH_STAMP(3)
            const int ntile = m * (m + 1) / 2;
            if (g_chol_scalar && m > 0) {
                update_scalar(s, m);
                __syncthreads();
                if (wave == 0) factor_diag(s + 1, false);
            } else if (wave == 0) {
                if (m > 0) {
                    update_tile(s, 0, true);
                    CH_WAVE_SYNC();
                }
                factor_diag(s + 1, m > 0);
            } else {
                for (int t = wave; t < ntile; t += BA_W - 1) update_tile(s, t, false);
            }
            __syncthreads();
            CH_STAMP(4)
        }
    } else {
        for (int kb = 0; kb < nb; kb++) {
            const int m = elast[kb] - kb; /* tile rows below the diagonal that reach column kb */
            if (wave == 0) factor_diag(kb, false);
            __syncthreads();
            CH_STAMP(2)
            panel(kb, m);
            __syncthreads();
            CH_STAMP(3)
            if (g_chol_scalar) {
                update_scalar(kb, m);
            } else {
                const int ntile = m * (m + 1) / 2;
                for (int t = wave; t < ntile; t += BA_W) update_tile(kb, t, false);
            }
            __syncthreads();
            CH_STAMP(4)
        }
    }
    const bool ok = bad_flag[0] == 0.0;
    /* ---- forward substitution L y = g ---- */
    for (int i = tid; i < np; i += BA_T) z[i] = (i < n) ? (TWO ? D[w.g + i] + D[w.g2 + i] : D[w.g + i]) : 0.0;
    __syncthreads();
    for (int kb = 0; kb < nb; kb++) {
        double v = 0.0;
        if (tid < 16)
            for (int k = 0; k < 16; k++) v += Li[kb * 256 + tid * 16 + k] * z[kb * 16 + k];
        __syncthreads();
        if (tid < 16) z[kb * 16 + tid] = v;
        __syncthreads();
        for (int i = (kb + 1) * 16 + tid; i < (elast[kb] + 1) * 16; i += BA_T) {
            const double *Ar = A + (int64_t)i * np + kb * 16;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++) acc += Ar[k] * z[kb * 16 + k];
            z[i] -= acc;
        }
        __syncthreads();
    }
    CH_STAMP(5)
    /* ---- backward substitution L^T x = y ---- */
    for (int kb = nb - 1; kb >= 0; kb--) {
        double v = 0.0;
        if (tid < 16)
            for (int k = 0; k < 16; k++) v += Li[kb * 256 + k * 16 + tid] * z[kb * 16 + k];
        __syncthreads();
        if (tid < 16) z[kb * 16 + tid] = v;
        __syncthreads();
        for (int j = efirst[kb] * 16 + tid; j < kb * 16; j += BA_T) {
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 16; r++) acc += A[(int64_t)(kb * 16 + r) * np + j] * z[kb * 16 + r];
            z[j] -= acc;
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += BA_T) D[w.x + i] = ok ? z[i] : 0.0;
    __syncthreads();
    CH_STAMP(6)
    return ok;
}

/* ------------------------------------------------------------------------------------------------------------
 * BAND Cholesky with an LDS-resident sliding window (round 3): the same factorisation as ba_chol_solve -- 16 x 16 tiles,
 * the diagonal tile and its inverse by one wavefront in registers, panel and trailing update on v_mfma_f64_16x16x4 --
 * with NO global-memory round trip on its sequential chain.
 *
 * Local-BA windows are chain-like: the envelope of S is a band (100 keyframes x 5000 points: 38 tile rows, at most
 * M = 4 tile rows below a diagonal tile reach its column), and a step of the factorisation is tiny (<= M panel tiles,
 * <= M (M + 1) / 2 update tiles).  What a step cost in ba_chol_solve (12 us, 38 of them per solve) was latency: the
 * diagonal tile, the panel and the trailing tiles each came from and went back to global memory -- HBM, with a few
 * hundred windows in flight -- and so did the rows of the backward substitution.  Here:
 *
 *   - the tiles that are still being updated live in LDS: tile (i, j), d = i - j, sits in ring d at position
 *     j mod (M + 2 - d).  At step s the live tiles of diagonal d are the columns s .. elast[s + 1] - d, a contiguous
 *     range no longer than the ring, so the modular index never collides; all rings together hold (M + 1)(M + 4) / 2
 *     tiles (M = 4: 20 tiles = 43 KB);
 *   - a tile row ENTERS the window one step before its first column is eliminated, straight from the block-sparse S
 *     (+ S2 of the two-part Schur pass): the dense np x np matrix is never assembled in memory (zero fill + scatter +
 *     first read of the envelope were a sixth of the batch's HBM traffic).  Its loads are a chain of two round trips
 *     (fill plan entry -> S value) and are taken OFF the factorisation's chain by software pipelining across the steps:
 *     step s issues the plan loads of the rows that enter at step s + 2 and the S loads of those that enter at s + 1,
 *     and writes the values that arrived for the rows entering now -- the step barriers inside the loop are raw
 *     s_barrier with an LDS-only wait (chb_barrier), so loads stay in flight across them.  The FILL PLAN (per
 *     lower-triangle entry of S inside the envelope: source index, needs-S2 flag, LDS offset of its destination) is a
 *     function of the window's pattern alone: built once per batch by k_ba_ph_fillplan, read by every trial;
 *   - LOOK-AHEAD: the wavefront that updates tile (s + 1, s + 1) factorises it at once, beside the rest of the
 *     trailing update -- with the tiles in LDS the kept tile no longer drags an HBM read-modify-write onto the chain
 *     (why the look-ahead lost on the batch path in round 2);
 *   - the forward substitution rides along (y_s = L_ss^-1 z_s during the panel, z_i -= L_is y_s during the update);
 *   - finished panel tiles and inverse diagonal tiles go to memory once, for the backward substitution (x_i = L_ii^-T
 *     z_i, then z_j -= L_ij^T x_i over the columns j of row i's envelope: contiguous in memory, one lane per column):
 *     row i belongs to wavefront i mod NW, which holds it in registers and reloads them with its next row right after
 *     use -- every row is requested NW steps before it is needed.
 *
 * NT threads per window (256: two windows share a CU -- one's chain overlaps the other's products; 512 for batches
 * that cannot fill the device twice).  Same arithmetic per entry as ba_chol_solve up to the order of the MFMA
 * accumulations (none changed), so results agree to rounding; the oracle tests hold both. */
#define CHB_TILE (16 * CH_LD)
#define CHB_MAX_M 8     /* widest band (tile rows below a diagonal tile) the band kernel takes */
#define CHB_MAX_NB 1024 /* most tile rows (k_ba_ph_fillplan counts per tile row in LDS) */
__host__ __device__ __forceinline__ int chb_ring(int M, int d) { return M + 2 - d; }
__host__ __device__ __forceinline__ int chb_base(int M, int d) { return d * (M + 2) - d * (d - 1) / 2; }
__host__ __device__ __forceinline__ int chb_slots(int M) { return chb_base(M, M + 1); }
__host__ __device__ __forceinline__ int chb_ent0(int nb) { return (nb + 2) & ~1; } /* first entry of a fill plan, in ints (8-byte aligned) */
/* doubles of dynamic LDS: tile rings, inverse of the current diagonal tile, z, and (as ints) efirst / elast / plan starts */
__host__ __device__ __forceinline__ size_t chb_lds_doubles(int M, int np) {
    return (size_t)chb_slots(M) * CHB_TILE + 2 * CHB_TILE + np + 16 + ((3 + (size_t)M + 1) * (size_t)(np / 16 + 4) + 1) / 2;
}
/* workgroup barrier that waits for this wave's LDS operations only: global loads issued before it stay in flight */
__device__ __forceinline__ void chb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

/* the fill plan of a window (once per batch): plan ints = [ starts: nb + 1 | entries: 2 per lower-triangle scalar of S ]
   entry = { source index into S (bit 31: add S2), LDS double offset of the destination inside the tile rings } */
__global__ __launch_bounds__(256) void k_ba_ph_fillplan(BABufs bufs) {
    __shared__ int cnt[CHB_MAX_NB + 1];
    const BAWin w = bufs.wins[blockIdx.x];
    const int32_t *I = bufs.I;
    if (w.fplan < 0) return;
    int32_t *plan = (int32_t *)(bufs.D + w.fplan);
    const int n = w.Ncf * 6, np = (n + 15) & ~15, nb = np >> 4, M = w.ch_pm >> 4;
    for (int i = threadIdx.x; i <= nb; i += 256) cnt[i] = 0;
    __syncthreads();
    /* one thread per BLOCK of the lower triangle (its four index loads form one round trip, shared by its 36 entries);
       pass 0 counts the entries per tile row, pass 1 writes them behind the scanned starts */
    for (int pass = 0; pass < 2; pass++) {
        for (int bk = threadIdx.x; bk < w.nnzb; bk += 256) {
            const int c = I[w.blk_rowid + bk], j = I[w.blk_col + bk];
            if (j > c) continue;
            const int mi = I[w.blk_mirror + bk];
            const int two = (I[w.blk_touch + mi] & 2) ? (int)0x80000000 : 0;
#pragma unroll 6
            for (int e = 0; e < 36; e++) {
                const int rr = e / 6, cc = e - rr * 6;
                const int row = 6 * c + cc, col = 6 * j + rr; /* lower-triangle entry = upper block (j, c) entry [rr][cc] */
                if (col > row) continue;
                const int ti = row >> 4, tj = col >> 4;
                if (pass == 0) {
                    atomicAdd(&cnt[ti], 1);
                } else {
                    const int pos = atomicAdd(&cnt[ti], 1);
                    const int d = ti - tj, slot = chb_base(M, d) + tj % chb_ring(M, d);
                    plan[chb_ent0(nb) + 2 * pos] = (mi * 36 + e) | two;
                    plan[chb_ent0(nb) + 2 * pos + 1] = slot * CHB_TILE + (row & 15) * CH_LD + (col & 15);
                }
            }
        }
        __syncthreads();
        if (pass == 0) {
            if (threadIdx.x == 0) { /* exclusive scan; cnt[] becomes the write cursor of pass 1 */
                int acc = 0;
                for (int i = 0; i < nb; i++) {
                    const int c = cnt[i];
                    plan[i] = acc;
                    cnt[i] = acc;
                    acc += c;
                }
                plan[nb] = acc;
            }
            __syncthreads();
        }
    }
}

template <int NT, bool TWO, int MAXM>
__device__ __forceinline__ bool ba_chol_band(const BAWin &w, double *D, const int32_t *I, int n, double *dyn, double *bad_flag, int skip = 0) {
    constexpr int NW = NT / 64;
    constexpr int NTF = NT - 64;                 /* threads of the wavefronts beside the chain: they carry the fill */
    constexpr int IPT = NT == 256 ? 8 : 4;       /* plan entries per such thread and step (more: a synchronous remainder loop) */
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int np = (n + 15) & ~15, nb = np >> 4, M = w.ch_pm >> 4;
    double *A = D + w.A;     /* finished L tiles (dense row-major layout, only the band is touched) */
    double *Li = D + w.Linv; /* inverse diagonal tiles */
    const double *Sv = D + w.S, *S2v = D + w.S2;
    const int32_t *plan = (const int32_t *)(D + w.fplan);
    double *tiles = dyn;
    double *sl2 = tiles + (size_t)chb_slots(M) * CHB_TILE; /* inverses of the diagonal tiles of the current and the next step (s & 1) */
    double *z = sl2 + 2 * CHB_TILE;
    int32_t *efirst = (int32_t *)(z + np + 16), *elast = efirst + nb + 4, *pst = elast + nb + 4; /* LDS copies: nothing of the step bookkeeping waits for memory */
    int32_t *rpos = pst + nb + 4; /* rpos[d * (M + 2) + (j mod (M + 2))]... see T(): ring position of column j on diagonal d, as a table */
    /* tile (i, j): ring d = i - j, position j mod (M + 2 - d).  The modulus is a run-time value (an integer division per
       tile address otherwise): columns advance by one per step, so the position of column j is looked up in a table indexed
       by d and j mod LCMLEN -- with LCMLEN = nb rounded up the table is simply indexed by j */
    auto T = [&](int i, int j) -> double * {
        const int d = i - j;
        return tiles + rpos[d * nb + j];
    };
    auto last_of = [&](int s) { return s < nb ? elast[s] : nb - 1; };
    /* tile rows (r0, r1] enter: zero their tiles (columns efirst .. diagonal), identity on the padding rows */
    auto zero_rows = [&](int r0, int r1) {
        if (skip & 8) return;
        for (int i = r0 + 1; i <= r1; i++) {
            const int c0 = efirst[i];
            for (int e = tid; e < (i - c0 + 1) * 256; e += NT) {
                const int j = c0 + (e >> 8), r = (e >> 4) & 15, c = e & 15;
                tiles[rpos[(i - j) * nb + j] + r * CH_LD + c] = (j == i && r == c && i * 16 + r >= n) ? 1.0 : 0.0;
            }
        }
    };
    /* diagonal tile kb (in LDS): Cholesky factor and inverse by one wavefront in registers, as in ba_chol_solve */
    auto factor_diag = [&](int kb) {
        if (skip & 2) return;
        double *sl = sl2 + (kb & 1) * CHB_TILE;
        const int rl = lane & 15;
        const double *Td = T(kb, kb);
        double a[16];
#pragma unroll
        for (int k = 0; k < 16; k++) a[k] = Td[rl * CH_LD + k];
        bool bad = false;
        double inv[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            double ajj = lane_bcast(a[j], j);
            if (!(ajj > 0.0) || !isfinite(ajj)) {
                bad = true;
                ajj = 1.0;
            }
            /* 1 / sqrt(ajj): v_rsq_f64 (about 2^-23) and ONE step of the cubically convergent iteration
               r (1 + e / 2 + 3 e^2 / 8), e = 1 - ajj r^2 (error ~ e^3: below 2^-60) -- five dependent f64 operations where
               two Newton steps are eight; f64 issues at half rate, and this wavefront IS the solver's chain */
            double r = __builtin_amdgcn_rsq(ajj);
            {
                const double e = __builtin_fma(-(ajj * r), r, 1.0);
                r = __builtin_fma(r, __builtin_fma(0.375, e, 0.5) * e, r);
            }
            inv[j] = r;
            const double pv = ajj * r;
            a[j] = (rl == j) ? pv : a[j] * r;
#pragma unroll
            for (int k = j + 1; k < 16; k++) {
                const double lkj = lane_bcast(a[j], k);
                a[k] = __builtin_fma(-a[j], lkj, a[k]); /* (one instruction: the translation unit is built with -ffp-contract=off) */
            }
        }
        /* the inverse: column rl of L^-1 by forward substitution, x[i] = (delta - sum_{k<i} l[i][k] x[k]) / l[i][i].  Its 120
           entries of L travel through LDS -- written once, read as broadcasts (every lane the same address) -- instead of
           through 240 v_readlane on the wavefront that carries the factorisation's chain */
        double *Ld = T(kb, kb); /* the tile's own slot: nobody else reads it before the step barrier */
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 16; k++) Ld[rl * CH_LD + k] = a[k];
        }
        CH_WAVE_SYNC();
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double v = (i == rl) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < i) v = __builtin_fma(-Ld[i * CH_LD + k], x[k], v);
            x[i] = (i >= rl) ? v * inv[i] : 0.0;
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                sl[i * CH_LD + rl] = x[i];
                Li[kb * 256 + i * 16 + rl] = x[i];
            }
        }
        if (bad && lane == 0) bad_flag[0] = 1.0;
    };
    /* panel tile (gi, s): L = A L_ss^-T in place, and to memory for the backward substitution */
    auto panel_tile = [&](int gi, int s) {
        const double *sl = sl2 + (s & 1) * CHB_TILE;
        double *Tp = T(gi, s);
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const double a = Tp[(lane & 15) * CH_LD + 4 * s4 + (lane >> 4)];
            const double b = (4 * s4 + (lane >> 4) <= (lane & 15)) ? sl[(lane & 15) * CH_LD + 4 * s4 + (lane >> 4)] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = (lane >> 4) + 4 * r, col = lane & 15;
            Tp[row * CH_LD + col] = acc[r];
            A[(int64_t)(gi * 16 + row) * np + s * 16 + col] = acc[r];
        }
    };
    /* trailing tile (gi, gj) -= L_(gi,s) L_(gj,s)^T, all three in LDS */
    auto update_tile = [&](int gi, int gj, int s) {
        double *C = T(gi, gj) + (lane >> 4) * CH_LD + (lane & 15);
        const double *Pa = T(gi, s), *Pb = T(gj, s);
        v4d c4;
#pragma unroll
        for (int r = 0; r < 4; r++) c4[r] = C[4 * r * CH_LD];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const double a = -Pa[(lane & 15) * CH_LD + 4 * s4 + (lane >> 4)];
            const double b = Pb[(lane & 15) * CH_LD + 4 * s4 + (lane >> 4)];
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c4, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) C[4 * r * CH_LD] = c4[r];
    };
    auto s_value = [&](int src) -> double { /* entry of S (+ S2 where the second Schur part touched the block) */
        const int si = src & 0x7fffffff;
        double v = Sv[si];
        if (TWO && src < 0) v += S2v[si];
        return v;
    };
    /* plain (waiting) fill of plan entries [p0, p1) by `nthr` threads, this one being number t0 */
    auto fill_now = [&](int p0, int p1, int t0, int nthr) {
        if (skip & 1) return;
        for (int q = p0 + t0; q < p1; q += nthr) {
            const int2 en = ((const int2 *)(plan + chb_ent0(nb)))[q];
            tiles[en.y] = s_value(en.x);
        }
    };

    if (tid == 0) bad_flag[0] = 0.0;
    for (int i = tid; i < nb; i += NT) {
        efirst[i] = I[w.ch_first + i];
        elast[i] = I[w.ch_last + i];
    }
    for (int i = tid; i <= nb; i += NT) pst[i] = plan[i];
    for (int i = tid; i < (M + 1) * nb; i += NT) {
        const int d = i / nb, j = i - d * nb;
        rpos[i] = (chb_base(M, d) + j % chb_ring(M, d)) * CHB_TILE;
    }
    for (int i = tid; i < np; i += NT) z[i] = (i < n) ? (TWO ? D[w.g + i] + D[w.g2 + i] : D[w.g + i]) : 0.0;
    __syncthreads();
    /* rows 0 .. elast[0] are in the window before step 0 */
    zero_rows(-1, elast[0]);
    __syncthreads();
    fill_now(pst[0], pst[elast[0] + 1], tid, NT);
    /* software pipeline of the fill, on the threads beside the chain: entA = plan entries of the rows that enter two
       steps from now (stage A), val / offB = the S values / destinations of those that enter next step (stage B) */
    int2 entA[IPT];
    double val[IPT];
    int offB[IPT];
    int baseA = 0, endA = 0, baseB = 0, endB = 0; /* plan ranges of the stages */
    const int tf = tid - 64;
    auto stage_A = [&](int r0, int r1) { /* rows (r0, r1]: issue the plan loads */
        baseA = pst[r0 + 1];
        endA = pst[r1 + 1];
        if (skip & 1) endA = baseA;
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const int q = baseA + tf + k * NTF;
            entA[k] = ((const int2 *)(plan + chb_ent0(nb)))[q < endA ? q : (baseA < endA ? baseA : 0)];
        }
    };
    auto stage_B = [&]() { /* the entries of stage A have arrived: issue their S loads */
        baseB = baseA;
        endB = endA;
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            val[k] = (baseB + tf + k * NTF < endB) ? s_value(entA[k].x) : 0.0;
            offB[k] = entA[k].y;
        }
    };
    auto stage_C = [&]() { /* the values of stage B have arrived: into the (zeroed) tiles; what did not fit the registers, plainly */
#pragma unroll
        for (int k = 0; k < IPT; k++)
            if (baseB + tf + k * NTF < endB) tiles[offB[k]] = val[k];
        fill_now(baseB + IPT * NTF, endB, tf, NTF);
    };
    __syncthreads();
    if (wave != 0) {
        stage_A(elast[0], last_of(1));
        stage_B();
        stage_A(last_of(1), last_of(2));
    } else {
        factor_diag(0);
    }
    __syncthreads();
    /* (skip & 32: wavefront 0 times its own sections -- s_memrealtime ticks of 10 ns -- and prints them; diagnostics) */
    unsigned long long tA = 0, tbA = 0, tU = 0, tF = 0, tbB = 0, t0 = 0, t1;
#define CHB_T(acc)                                      \
    if (skip & 32) {                                    \
        t1 = __builtin_amdgcn_s_memrealtime();          \
        acc += t1 - t0;                                 \
        t0 = t1;                                        \
    }
    if (skip & 32) t0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < nb; s++) {
        const int m = (skip & 16) ? 0 : elast[s] - s;
        /* phase A: panel of column s; y_s; zero the rows that enter at this step */
        for (int bi = wave; bi < m; bi += NW) panel_tile(s + 1 + bi, s);
        zero_rows(elast[s], last_of(s + 1));
        CHB_T(tA)
        chb_barrier();
        CHB_T(tbA)
        /* phase B: wavefront 0 carries the chain (next diagonal tile); the others the rest of the update, the forward
           substitution and the fill pipeline */
        if (wave == 0) {
            if (m > 0) {
                update_tile(s + 1, s + 1, s);
                CH_WAVE_SYNC();
            }
            CHB_T(tU)
            if (s + 1 < nb) factor_diag(s + 1);
            CHB_T(tF)
        } else {
            const int ntile = m * (m + 1) / 2;
            for (int t = wave; t < ntile; t += NW - 1) { /* tile t = (ii, jj), jj <= ii, row-major over the triangle; t = 0 is wavefront 0's */
                int ii = 0;
                while ((ii + 1) * (ii + 2) / 2 <= t) ii++;
                const int jj = t - ii * (ii + 1) / 2;
                update_tile(s + 1 + ii, s + 1 + jj, s);
            }
            if (wave == NW - 1) { /* y_s = L_ss^-1 z_s, then z_i -= L_(i,s) y_s: one lane per row of the panel (m <= 8: two rows per lane at most) */
                const double *sl = sl2 + (s & 1) * CHB_TILE; /* (the chain wavefront is writing the OTHER inverse meanwhile) */
                double ys = 0.0;
                if (lane < 16) {
#pragma unroll
                    for (int k = 0; k < 16; k++) ys += sl[lane * CH_LD + k] * z[s * 16 + k]; /* (sl is zero above its diagonal) */
                }
                CH_WAVE_SYNC();
                if (lane < 16) z[s * 16 + lane] = ys;
                CH_WAVE_SYNC();
                for (int r = lane; r < m * 16; r += 64) {
                    const double *Lr = T(s + 1 + (r >> 4), s) + (r & 15) * CH_LD;
                    double acc = 0.0;
#pragma unroll
                    for (int k = 0; k < 16; k++) acc += Lr[k] * z[s * 16 + k];
                    z[(s + 1) * 16 + r] -= acc;
                }
            }
            stage_C();                                 /* rows (elast[s], elast[s + 1]]: first used at step s + 1 */
            stage_B();                                 /* rows (elast[s + 1], elast[s + 2]] */
            stage_A(last_of(s + 2), last_of(s + 3));
        }
        chb_barrier();
        CHB_T(tbB)
    }
    if ((skip & 32) && tid == 0 && blockIdx.x == 0)
        printf("[chol band] %d steps, us per step on the chain wavefront: phase A %.2f | barrier A %.2f | update (s+1,s+1) %.2f | factor+inverse %.2f | barrier B %.2f\n",
               nb, tA * 0.01 / nb, tbA * 0.01 / nb, tU * 0.01 / nb, tF * 0.01 / nb, tbB * 0.01 / nb);
    __syncthreads(); /* (a full one: the L tiles and inverse diagonal tiles written above are read below) */
    const bool ok = bad_flag[0] == 0.0;
    /* backward substitution L^T x = y, row by row from the bottom; row i belongs to wavefront i mod NW */
    if (!(skip & 4)) {
        constexpr int CPL = (MAXM * 16 + 63) / 64; /* envelope columns of a row per lane (MAXM: widest band of the instantiation) */
        double rowv[CPL][16], liv[16];
        auto load_row = [&](int i) {
            const int c0 = efirst[i] * 16, ncol = i * 16 - c0;
#pragma unroll
            for (int q = 0; q < CPL; q++) {
                const int j = lane + 64 * q;
#pragma unroll
                for (int r = 0; r < 16; r++) rowv[q][r] = j < ncol ? A[(int64_t)(i * 16 + r) * np + c0 + j] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) liv[k] = Li[i * 256 + k * 16 + (lane & 15)];
        };
        int mine = nb - 1 - ((nb - 1 - wave) % NW + NW) % NW; /* the last row i with i mod NW == wave */
        if (mine >= 0) load_row(mine);
        for (int i = nb - 1; i >= 0; i--) {
            if (i == mine) {
                double x = 0.0;
#pragma unroll
                for (int k = 0; k < 16; k++) x += liv[k] * z[i * 16 + k]; /* x_i[c] = sum_k Linv[k][c] z_i[k] */
                CH_WAVE_SYNC();
                if (lane < 16) z[i * 16 + lane] = x;
                CH_WAVE_SYNC();
                const int c0 = efirst[i] * 16, ncol = i * 16 - c0;
#pragma unroll
                for (int q = 0; q < CPL; q++) {
                    const int j = lane + 64 * q;
                    double acc = 0.0;
#pragma unroll
                    for (int r = 0; r < 16; r++) acc += rowv[q][r] * z[i * 16 + r];
                    if (j < ncol) z[c0 + j] -= acc;
                }
                mine -= NW;
                if (mine >= 0) load_row(mine);
            }
            chb_barrier();
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) D[w.x + i] = ok ? z[i] : 0.0;
    return ok;
}

/* out-of-line instances for the fused kernel (k_ba_iteration: 512 threads; S only -- no second Schur accumulator there) */
__device__ __noinline__ bool ba_chol_band_fused4(const BAWin &w, double *D, const int32_t *I, int n, double *dyn, double *bad_flag) {
    return ba_chol_band<BA_T, false, 4>(w, D, I, n, dyn, bad_flag);
}
__device__ __noinline__ bool ba_chol_band_fused8(const BAWin &w, double *D, const int32_t *I, int n, double *dyn, double *bad_flag) {
    return ba_chol_band<BA_T, false, 8>(w, D, I, n, dyn, bad_flag);
}
/* ... and for k_ba_ph_finish (the two-part Schur accumulators S + S2, g + g2) */
__device__ __noinline__ bool ba_chol_band_two4(const BAWin &w, double *D, const int32_t *I, int n, double *dyn, double *bad_flag) {
    return ba_chol_band<BA_T, true, 4>(w, D, I, n, dyn, bad_flag);
}
__device__ __noinline__ bool ba_chol_band_two8(const BAWin &w, double *D, const int32_t *I, int n, double *dyn, double *bad_flag) {
    return ba_chol_band<BA_T, true, 8>(w, D, I, n, dyn, bad_flag);
}

/* Schur complement, batch path (one workgroup per window).
 * The points are laid out sorted by their first free camera, so a chunk of consecutive points (<= BA_CHUNK_E edges,
 * <= BA_CHUNK_P points) touches a compact band of S.  Per chunk: the W records of its edges, the Dinv / tv of its
 * points and the edge -> point map are streamed ONCE, coalesced, into the dynamic LDS region (free until the Cholesky
 * needs it) -- fetched into registers while the previous chunk is still being worked on, so the memory latency hides
 * behind the block products; every (block, chunk) pair has one owner, a 4-lane group, that adds up the chunk's
 * contributions to that block from LDS in list order (Y = W Dinv formed in registers) and subtracts the sum from the
 * block in memory (the block's current value is requested before the products start); chunks follow each other behind
 * a barrier.  Fixed order, no atomics.  g = bp - sum W tv rides along the same way with one owner per
 * (camera, chunk).  HBM sees each W record once per solve instead of once per block it takes part in (the
 * list-based version moved 5.7x the algorithmic bytes with 256 windows in flight).
 * A separate function with its own register allocation (inlined, the live doubles of the block products pushed the
 * whole iteration kernel from 39 to 83 spilled VGPRs). */
#define BA_CH_WPT ((BA_CHUNK_E * 9 + BA_T - 1) / BA_T)  /* double2 of W per thread */
#define BA_CH_DPT ((BA_CHUNK_P * 9 + BA_T - 1) / BA_T)  /* doubles of Dinv per thread */
#define BA_CH_CPT (BA_CHUNK_C / BA_T)                   /* contribution-list ints per thread */
/* sum of 36 per-lane partials over the 16 lanes of a group by recursive halving: at offset 8, 4, 2, 1 a lane keeps
 * the half of its values that its lane bit selects and adds the partner's copy of that half -- 24 + 12 + 6 + 3 = 45
 * exchanges instead of the 144 of four full butterflies (each f64 exchange is two LDS-crossbar permutes).  Lane q ends
 * up with the totals of entries 3q, 3q+1, 3q+2 (lanes 12..15: padding). */
/* value of `v` in lane (lane ^ M), M = 1, 2, 4, 8: data-parallel-primitive moves inside a row of 16 lanes -- VALU
 * instructions on each of the four SIMDs, where a ds_bpermute pair per f64 goes through the CU's one LDS pipe (the 90
 * bpermutes that ended every (block, chunk) pair of the Schur pass were that pipe's largest customer).
 * xor 1 / xor 2: quad_perm; xor 8: row_ror:8; xor 4 = row_half_mirror (i -> i ^ 7) then quad_perm [3,2,1,0] (i -> i ^ 3). */
template <int M> __device__ __forceinline__ int xchg16_b32(int v) {
    if (M == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);
    if (M == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);
    if (M == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);
    const int h = __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false);
    return __builtin_amdgcn_update_dpp(h, h, 0x1B, 0xF, 0xF, false);
}
template <int M> __device__ __forceinline__ double xchg16(double v) {
    return __hiloint2double(xchg16_b32<M>(__double2hiint(v)), xchg16_b32<M>(__double2loint(v)));
}
__device__ __forceinline__ void group16_reduce36(const double *acc, int l16, double &o0, double &o1, double &o2) {
    double a24[24], a12[12], a6[6];
    const bool h8 = l16 & 8, h4 = l16 & 4, h2 = l16 & 2, h1 = l16 & 1;
#pragma unroll
    for (int i = 0; i < 24; i++) {
        const double lo = acc[i], hi = (24 + i < 36) ? acc[24 + i] : 0.0;
        a24[i] = (h8 ? hi : lo) + xchg16<8>(h8 ? lo : hi);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) a12[i] = (h4 ? a24[12 + i] : a24[i]) + xchg16<4>(h4 ? a24[i] : a24[12 + i]);
#pragma unroll
    for (int i = 0; i < 6; i++) a6[i] = (h2 ? a12[6 + i] : a12[i]) + xchg16<2>(h2 ? a12[i] : a12[6 + i]);
    o0 = (h1 ? a6[3] : a6[0]) + xchg16<1>(h1 ? a6[0] : a6[3]);
    o1 = (h1 ? a6[4] : a6[1]) + xchg16<1>(h1 ? a6[1] : a6[4]);
    o2 = (h1 ? a6[5] : a6[2]) + xchg16<1>(h1 ? a6[2] : a6[5]);
}
/* the same over a group of 8 or 4 lanes (offsets 4, 2, 1 / 2, 1): 8 lanes -- 36 padded to 48, lane q ends up with the
 * totals of entries 6q .. 6q+5 (lanes 6, 7: padding), 24 + 12 + 6 = 42 exchanges; 4 lanes -- lane q holds entries
 * 9q .. 9q+8, 18 + 9 = 27 exchanges.  A (block, chunk) pair's list is 10 contributions long on average, so the narrower
 * group idles less; BA_SG picks the width of the phase-kernel Schur pass. */
#define BA_SG 4
#define BA_SG_OUT (BA_SG == 8 ? 6 : 9)
__device__ __forceinline__ void group_reduce36(const double *acc, int lg, double *o) {
#if BA_SG == 8
    double a24[24], a12[12];
    const bool h4 = lg & 4, h2 = lg & 2, h1 = lg & 1;
#pragma unroll
    for (int i = 0; i < 24; i++) {
        const double lo = acc[i], hi = (24 + i < 36) ? acc[24 + i] : 0.0;
        a24[i] = (h4 ? hi : lo) + xchg16<4>(h4 ? lo : hi);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) a12[i] = (h2 ? a24[12 + i] : a24[i]) + xchg16<2>(h2 ? a24[i] : a24[12 + i]);
#pragma unroll
    for (int i = 0; i < 6; i++) o[i] = (h1 ? a12[6 + i] : a12[i]) + xchg16<1>(h1 ? a12[i] : a12[6 + i]);
#else
    double a18[18];
    const bool h2 = lg & 2, h1 = lg & 1;
#pragma unroll
    for (int i = 0; i < 18; i++) a18[i] = (h2 ? acc[18 + i] : acc[i]) + xchg16<2>(h2 ? acc[i] : acc[18 + i]);
#pragma unroll
    for (int i = 0; i < 9; i++) o[i] = (h1 ? a18[9 + i] : a18[i]) + xchg16<1>(h1 ? a18[i] : a18[9 + i]);
#endif
}

__device__ __noinline__ void ba_schur_chunked(const BAWin &w, double *D, const int32_t *I, double *ba_dyn, int debug) {
#pragma clang fp contract(fast) /* the block products as FMAs: half the f64 instructions; estimates are compared at 1e-5 */
    const int tid = threadIdx.x;
    unsigned long long tp = (debug & 3) ? __builtin_amdgcn_s_memrealtime() : 0ull; /* bits 4..: timing experiments that skip parts */
#define SCH_STAMP(k)                                                        \
    if (debug & 3) {                                                          \
        __syncthreads();                                                    \
        if (tid == 0) {                                                     \
            const unsigned long long tn = __builtin_amdgcn_s_memrealtime(); \
            D[w.st + ST_T0 + 10 + (k)] += (double)(tn - tp);                \
            tp = tn;                                                        \
        }                                                                   \
    }
    /* LDS tiles of one chunk */
    double *sW = ba_dyn, *sD = sW + BA_CHUNK_E * 18, *sT = sD + BA_CHUNK_P * 9;
    int32_t *sP = (int32_t *)(sT + BA_CHUNK_P * 3); /* edge -> local point                        */
    int32_t *sC = sP + BA_CHUNK_E;                  /* contribution lists (a | b << 9 | p << 18)   */
    int32_t *sB = sC + BA_CHUNK_C;                  /* blocks touched                              */
    int32_t *sBS = sB + BA_CHUNK_B;                 /* their list starts, relative to the chunk's  */
    int32_t *sG = sBS + BA_CHUNK_B + 8;             /* local edges by free camera                  */
    const int grp = tid >> 4, l16 = tid & 15;
    double2 rW[BA_CH_WPT];
    double rD[BA_CH_DPT], rT;
    int32_t rP, rC[BA_CH_CPT], rB, rBS, rG;
    static_assert(BA_CHUNK_E <= 512 && BA_CHUNK_P <= 8192, "a | b << 9 | p << 18 in 32 bits");
    static_assert(BA_CHUNK_E <= BA_T && BA_CHUNK_P * 3 <= BA_T && BA_CHUNK_B + 1 <= BA_T && BA_CHUNK_C % BA_T == 0,
                  "one element per thread for the small tiles");
    auto fetch = [&](int ch) { /* chunk ch -> registers, every load issued back to back */
        const int32_t *cd = I + w.ch_desc + ch * BA_CH_DESC;
        const int e0 = cd[0], ne = cd[1] - cd[0], p0 = cd[2], npnt = cd[3] - cd[2], cb0 = cd[4], ncb = cd[5] - cd[4];
        const int c0 = cd[8], nc = cd[9], g0 = cd[10], ng = cd[11];
        const double2 *src = (const double2 *)(D + w.W + (int64_t)e0 * 18); /* 144-B records: 16-B aligned */
#pragma unroll
        for (int k = 0; k < BA_CH_WPT; k++) {
            const int i = tid + k * BA_T;
            rW[k] = i < ne * 9 ? src[i] : double2{0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < BA_CH_DPT; k++) {
            const int i = tid + k * BA_T;
            rD[k] = i < npnt * 9 ? D[w.Dinv + (int64_t)p0 * 9 + i] : 0.0;
        }
        rT = tid < npnt * 3 ? D[w.tv + (int64_t)p0 * 3 + tid] : 0.0;
        rP = tid < ne ? I[w.pt + e0 + tid] - p0 : 0;
#pragma unroll
        for (int k = 0; k < BA_CH_CPT; k++) {
            const int i = tid + k * BA_T;
            rC[k] = i < nc ? I[w.cb_con + c0 + i] : 0;
        }
        rB = tid < ncb ? I[w.cb_blk + cb0 + tid] : 0;
        rBS = tid <= ncb ? I[w.cb_start + cb0 + tid] - c0 : 0;
        rG = tid < ng ? I[w.cg_edge + g0 + tid] : 0;
    };
    auto commit = [&]() { /* registers -> LDS tiles */
#pragma unroll
        for (int k = 0; k < BA_CH_WPT; k++) {
            const int i = tid + k * BA_T;
            if (i < BA_CHUNK_E * 9) ((double2 *)sW)[i] = rW[k];
        }
#pragma unroll
        for (int k = 0; k < BA_CH_DPT; k++) {
            const int i = tid + k * BA_T;
            if (i < BA_CHUNK_P * 9) sD[i] = rD[k];
        }
        if (tid < BA_CHUNK_P * 3) sT[tid] = rT;
        if (tid < BA_CHUNK_E) sP[tid] = rP;
#pragma unroll
        for (int k = 0; k < BA_CH_CPT; k++) sC[tid + k * BA_T] = rC[k];
        if (tid < BA_CHUNK_B) sB[tid] = rB;
        if (tid <= BA_CHUNK_B) sBS[tid] = rBS;
        if (tid < BA_CHUNK_E) sG[tid] = rG;
    };
    fetch(0);
    for (int ch = 0; ch < w.nchunks; ch++) {
        const int32_t *cd = I + w.ch_desc + ch * BA_CH_DESC;
        const int ncb = cd[5] - cd[4], cg0 = cd[6], ncg = cd[7] - cd[6], g0 = cd[10];
        commit();
        __syncthreads();
        SCH_STAMP(0)
        if (ch + 1 < w.nchunks) fetch(ch + 1); /* in flight during the products below */
        SCH_STAMP(1)
        /* blocks: group g takes blocks g, g + 32, ... of the chunk's list (sorted heaviest first on the host) */
        for (int u = grp; u < ncb; u += BA_T / 16) {
            const int bk = sB[u] & BA_BLK_MASK, k0 = sBS[u], k1 = sBS[u + 1]; /* (the flags serve the two-part pass) */
            /* lanes 0..11 of the group own 3 entries of the 6 x 6 block each for the update; requested now, used last */
            double *Sb = D + w.S + (int64_t)bk * 36 + (l16 < 12 ? l16 : 0) * 3;
            const double sb0 = Sb[0], sb1 = Sb[1], sb2 = Sb[2];
            double acc[36];
#pragma unroll
            for (int i = 0; i < 36; i++) acc[i] = 0.0;
            for (int k = k0 + l16; k < k1; k += 16) {
                const uint32_t ab = (uint32_t)sC[k];
                const int a = ab & 511, b = (ab >> 9) & 511;
                const double *Wa = sW + a * 18, *Wb = sW + b * 18, *Iv = sD + (ab >> 18) * 9;
                double y[18], ww[18];
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const double w0 = Wa[r * 3], w1 = Wa[r * 3 + 1], w2 = Wa[r * 3 + 2];
                    y[r * 3] = w0 * Iv[0] + w1 * Iv[3] + w2 * Iv[6];
                    y[r * 3 + 1] = w0 * Iv[1] + w1 * Iv[4] + w2 * Iv[7];
                    y[r * 3 + 2] = w0 * Iv[2] + w1 * Iv[5] + w2 * Iv[8];
                }
#pragma unroll
                for (int i = 0; i < 18; i++) ww[i] = Wb[i];
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c < 6; c++)
                        acc[r * 6 + c] += y[r * 3] * ww[c * 3] + y[r * 3 + 1] * ww[c * 3 + 1] + y[r * 3 + 2] * ww[c * 3 + 2];
            }
            double d0, d1, d2;
            group16_reduce36(acc, l16, d0, d1, d2); /* lane q now holds the totals of entries 3q .. 3q+2 */
            if (l16 < 12) {
                Sb[0] = sb0 - d0;
                Sb[1] = sb1 - d1;
                Sb[2] = sb2 - d2;
            }
        }
        SCH_STAMP(2)
        /* g: group g takes cameras g, g + 32, ... of the chunk */
        for (int u = (debug & 64) ? ncg : grp; u < ncg; u += BA_T / 16) {
            const int sl = I[w.cg_slot + cg0 + u];
            const int k0 = I[w.cg_start + cg0 + u] - g0, k1 = I[w.cg_start + cg0 + u + 1] - g0;
            double acc[6] = {0, 0, 0, 0, 0, 0};
            for (int k = k0 + l16; k < k1; k += 16) {
                const int a = sG[k];
                const double *Wa = sW + a * 18, *t = sT + sP[a] * 3;
                const double t0 = t[0], t1 = t[1], t2 = t[2];
#pragma unroll
                for (int r = 0; r < 6; r++) acc[r] += Wa[r * 3] * t0 + Wa[r * 3 + 1] * t1 + Wa[r * 3 + 2] * t2;
            }
#pragma unroll
            for (int r = 0; r < 6; r++) {
                double v = acc[r];
                v += xchg16<8>(v);
                v += xchg16<4>(v);
                v += xchg16<2>(v);
                v += xchg16<1>(v);
                acc[r] = v;
            }
            if (l16 == 0) {
#pragma unroll
                for (int r = 0; r < 6; r++) D[w.g + sl * 6 + r] -= acc[r];
            }
        }
        __syncthreads(); /* the next chunk overwrites the LDS tiles and may touch the same blocks */
        SCH_STAMP(3)
    }
}

__device__ __noinline__ void ba_schur_lists(const BAWin &w, double *D, const int32_t *I, int gwave, int gwaves) {
    const int lane = threadIdx.x & 63;
    /* ---- TEAM PATH (one window over G workgroups) and windows with a point too large for a chunk: every
       block on or above the diagonal has one owner for the whole solve, a 16-lane group, that walks the block's
       contribution list (edge pairs of the points both cameras see) with W and Dinv gathered from memory --
       a single window's 8 MB stay in L2 / Infinity Cache.  Four blocks per wavefront, the edge ids of the NEXT
       contribution fetched ahead of the operand loads of the current one. ---- */
    /* The index chain of a block -- blk_upper -> con_start -> con_e1 / con_e2 -> pt -- is four dependent memory round
       trips before the first operand load, and one window in flight has nothing else to hide them behind: they are
       walked for SL_R rounds at once (independent loads side by side), so a round pays one round trip for its operands
       and a quarter of the chain.  The update of S is spread over the group's 16 lanes (2-3 entries each, loads in
       parallel) instead of 36 dependent read-modify-writes on one lane. */
    constexpr int SL_R = 4;
    const int sub = lane >> 4, l16 = lane & 15;
    for (int ub = gwave * 4; ub < w.nub; ub += gwaves * 4 * SL_R) {
        int bkr[SL_R], k1r[SL_R], kqr[SL_R], e1r[SL_R], e2r[SL_R], ppr[SL_R];
#pragma unroll
        for (int r = 0; r < SL_R; r++) {
            const int u = ub + r * gwaves * 4 + sub;
            bkr[r] = u < w.nub ? I[w.blk_upper + u] : w.nnzb; /* upper blocks only */
        }
#pragma unroll
        for (int r = 0; r < SL_R; r++) {
            const bool on = bkr[r] < w.nnzb;
            const int k0 = on ? I[w.con_start + bkr[r]] : 0;
            k1r[r] = on ? I[w.con_start + bkr[r] + 1] : 0;
            kqr[r] = k0 + l16;
        }
#pragma unroll
        for (int r = 0; r < SL_R; r++) {
            const bool on = kqr[r] < k1r[r];
            e1r[r] = on ? I[w.con_e1 + kqr[r]] : 0;
            e2r[r] = on ? I[w.con_e2 + kqr[r]] : 0;
        }
#pragma unroll
        for (int r = 0; r < SL_R; r++) ppr[r] = kqr[r] < k1r[r] ? I[w.pt + e1r[r]] : 0;
#pragma unroll
        for (int r = 0; r < SL_R; r++) {
        if (ub + r * gwaves * 4 >= w.nub) break; /* uniform */
        const int bk = bkr[r];
        double acc[36];
#pragma unroll
        for (int i = 0; i < 36; i++) acc[i] = 0.0;
        if (bk < w.nnzb) {
            const int k1 = k1r[r];
            const int kq = kqr[r];
            int e1n = e1r[r], e2n = e2r[r];
            int pn = ppr[r];
            for (int k = kq; k < k1; k += 16) {
                const int e1 = e1n, e2 = e2n, pp = pn;
                if (k + 16 < k1) {
                    e1n = I[w.con_e1 + k + 16];
                    e2n = I[w.con_e2 + k + 16];
                    pn = I[w.pt + e1n];
                }
                const double2 *W1 = (const double2 *)(D + w.W + (int64_t)e1 * 18); /* 144-B records */
                const double2 *W2 = (const double2 *)(D + w.W + (int64_t)e2 * 18);
                const double *Iv = D + w.Dinv + (int64_t)pp * 9;
                double w1[18], ww[18], iv[9];
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const double2 a2 = W1[i], b2 = W2[i];
                    w1[2 * i] = a2.x;
                    w1[2 * i + 1] = a2.y;
                    ww[2 * i] = b2.x;
                    ww[2 * i + 1] = b2.y;
                    iv[i] = Iv[i];
                }
                double y[18];
#pragma unroll
                for (int rr = 0; rr < 6; rr++) {
                    y[rr * 3] = w1[rr * 3] * iv[0] + w1[rr * 3 + 1] * iv[3] + w1[rr * 3 + 2] * iv[6];
                    y[rr * 3 + 1] = w1[rr * 3] * iv[1] + w1[rr * 3 + 1] * iv[4] + w1[rr * 3 + 2] * iv[7];
                    y[rr * 3 + 2] = w1[rr * 3] * iv[2] + w1[rr * 3 + 1] * iv[5] + w1[rr * 3 + 2] * iv[8];
                }
#pragma unroll
                for (int rr = 0; rr < 6; rr++)
#pragma unroll
                    for (int c = 0; c < 6; c++)
                        acc[rr * 6 + c] += y[rr * 3] * ww[c * 3] + y[rr * 3 + 1] * ww[c * 3 + 1] + y[rr * 3 + 2] * ww[c * 3 + 2];
            }
        }
#pragma unroll
        for (int i = 0; i < 36; i++) {
            double v = acc[i];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
            acc[i] = v;
        }
        if (bk < w.nnzb) { /* sole owner group: no race; every lane of the group holds all 36 sums */
            double *Sb = D + w.S + (int64_t)bk * 36;
            const double a0 = l16 == 0 ? acc[0] : l16 == 1 ? acc[1] : l16 == 2 ? acc[2] : l16 == 3 ? acc[3] : l16 == 4 ? acc[4] : l16 == 5 ? acc[5] : l16 == 6 ? acc[6] : l16 == 7 ? acc[7] : l16 == 8 ? acc[8] : l16 == 9 ? acc[9] : l16 == 10 ? acc[10] : l16 == 11 ? acc[11] : l16 == 12 ? acc[12] : l16 == 13 ? acc[13] : l16 == 14 ? acc[14] : acc[15];
            const double a1 = l16 == 0 ? acc[16] : l16 == 1 ? acc[17] : l16 == 2 ? acc[18] : l16 == 3 ? acc[19] : l16 == 4 ? acc[20] : l16 == 5 ? acc[21] : l16 == 6 ? acc[22] : l16 == 7 ? acc[23] : l16 == 8 ? acc[24] : l16 == 9 ? acc[25] : l16 == 10 ? acc[26] : l16 == 11 ? acc[27] : l16 == 12 ? acc[28] : l16 == 13 ? acc[29] : l16 == 14 ? acc[30] : acc[31];
            const double a2 = l16 == 0 ? acc[32] : l16 == 1 ? acc[33] : l16 == 2 ? acc[34] : acc[35];
            const double s0 = Sb[l16], s1 = Sb[16 + l16], s2 = l16 < 4 ? Sb[32 + l16] : 0.0;
            Sb[l16] = s0 - a0;
            Sb[16 + l16] = s1 - a1;
            if (l16 < 4) Sb[32 + l16] = s2 - a2;
        }
        }
    }
}

__device__ __noinline__ void ba_g_lists(const BAWin &w, double *D, const int32_t *I, int gwave, int gwaves) {
    const int lane = threadIdx.x & 63;
    /* g = bp - sum_e W_e tv : one wavefront per free camera, lanes over its edges (independent gathers) */
    for (int sl = gwave; sl < w.Ncf; sl += gwaves) {
        const int cam = I[w.blk_row + w.Ncf + 1 + sl]; /* slot -> camera (stored behind the row starts) */
        const int k0 = I[w.cam_start + cam], k1 = I[w.cam_start + cam + 1];
        double acc[6] = {0, 0, 0, 0, 0, 0};
        /* edge id -> point id -> operands are three dependent round trips per step: the ids of GL_U steps are fetched side
           by side first, so that a step waits for its operands only */
        constexpr int GL_U = 4;
        for (int kb = k0 + lane; kb < k1; kb += 64 * GL_U) {
            int eu[GL_U], pu[GL_U];
#pragma unroll
            for (int u = 0; u < GL_U; u++) eu[u] = kb + 64 * u < k1 ? I[w.cam_edges + kb + 64 * u] : -1;
#pragma unroll
            for (int u = 0; u < GL_U; u++) pu[u] = eu[u] >= 0 ? I[w.pt + eu[u]] : 0;
#pragma unroll
            for (int u = 0; u < GL_U; u++) {
                if (eu[u] < 0) continue;
                const double2 *We = (const double2 *)(D + w.W + (int64_t)eu[u] * 18);
                const double *tp = D + w.tv + (int64_t)pu[u] * 3;
                const double b0 = tp[0], b1 = tp[1], b2 = tp[2];
                double y[18];
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const double2 t = We[i];
                    y[2 * i] = t.x;
                    y[2 * i + 1] = t.y;
                }
#pragma unroll
                for (int a = 0; a < 6; a++) acc[a] += y[a * 3] * b0 + y[a * 3 + 1] * b1 + y[a * 3 + 2] * b2;
            }
        }
#pragma unroll
        for (int a = 0; a < 6; a++) {
            double v = acc[a];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            acc[a] = v;
        }
        if (lane == 0)
            for (int a = 0; a < 6; a++) D[w.g + sl * 6 + a] -= acc[a];
    }
}

/* PCG on S x = g for a reduced system whose vectors do not fit the LDS arrays (more than 256 free cameras AND an
 * envelope too wide for the Cholesky panel: a large, nearly dense global BA).  Same algorithm, preconditioner,
 * tolerance and iteration cap as the LDS version below; x, r, z, p, Ap live in global scratch.  One workgroup. */
__device__ __noinline__ bool ba_pcg_global(const BAWin &w, double *D, const int32_t *I, int n, double pcg_tol2, int pcg_maxit,
                                           bool minv_ok, int *iters, BAShared &s) {
    const int tid = threadIdx.x;
    double *x = D + w.pcgv, *r = x + n, *z = r + n, *pv = z + n, *Ap = pv + n;
    auto precond = [&](double &rz_out) { /* z = M^-1 r, returns r.z */
        double loc = 0;
        for (int i = tid; i < n; i += BA_T) {
            const int sl = i / 6, a = i % 6;
            const double *Mi = D + w.Minv + (int64_t)sl * 36 + a * 6;
            double v = 0;
            for (int c = 0; c < 6; c++) v += Mi[c] * r[sl * 6 + c];
            z[i] = v;
            loc += r[i] * v;
        }
        rz_out = block_sum(loc, s);
    };
    for (int i = tid; i < n; i += BA_T) {
        x[i] = 0.0;
        r[i] = D[w.g + i];
    }
    __syncthreads();
    double rz;
    precond(rz);
    for (int i = tid; i < n; i += BA_T) pv[i] = z[i];
    __syncthreads();
    const double rz0 = rz;
    bool ok = block_sum(minv_ok ? 0.0 : 1.0, s) == 0.0;
    int it = 0;
    if (ok && rz0 > 0) {
        for (it = 0; it < pcg_maxit; it++) {
            double loc = 0;
            for (int i = tid; i < n; i += BA_T) {
                const int sl = i / 6, a = i % 6;
                double v = 0;
                for (int bk = I[w.blk_row + sl]; bk < I[w.blk_row + sl + 1]; bk++) {
                    const double *Sr = D + w.S + (int64_t)bk * 36 + a * 6;
                    const double *pc = pv + I[w.blk_col + bk] * 6;
                    v += Sr[0] * pc[0] + Sr[1] * pc[1] + Sr[2] * pc[2] + Sr[3] * pc[3] + Sr[4] * pc[4] + Sr[5] * pc[5];
                }
                Ap[i] = v;
                loc += pv[i] * v;
            }
            const double pAp = block_sum(loc, s);
            if (!(pAp > 0)) {
                ok = false;
                break;
            }
            const double alpha = rz / pAp;
            for (int i = tid; i < n; i += BA_T) {
                x[i] += alpha * pv[i];
                r[i] -= alpha * Ap[i];
            }
            __syncthreads();
            double rz_new;
            precond(rz_new);
            if (rz_new <= pcg_tol2 * rz0) {
                it++;
                break;
            }
            const double beta = rz_new / rz;
            rz = rz_new;
            for (int i = tid; i < n; i += BA_T) pv[i] = z[i] + beta * pv[i];
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += BA_T) D[w.x + i] = ok ? x[i] : 0.0;
    *iters = it;
    return ok;
}

/* ---- phases of one LM iteration, shared by the fused kernel (k_ba_iteration: a window's whole optimize() round in one
 * launch, teams) and the phase kernels of the batch path (k_ba_ph_*).  Each is strided over gthreads threads (gwaves
 * wavefronts) of which this thread is gtid (this wavefront gwave). ---- */
/* buildSystem: Hll, bl (thread per point; W is written by the camera pass, which has the same Jacobians in hand and
   stores whole 144-byte records instead of 18 doubles strided over the lanes) */
/* FROM_OBS (the phase-kernel path): the edge's error is recomputed from its observation and the projection this pass has in
   hand anyway -- same expression as project_err, hence the same bits -- instead of being read from the err array; nothing
   reads that array then, so the trial pass does not write it and an accepted trial does not copy it (345 MB per trial and
   256 windows: a seventh of the batch's HBM traffic). */
template <bool FROM_OBS = false>
__device__ __forceinline__ void ba_phase_hll(const BAWin &w, double *D, const int32_t *I, const uint8_t *B, int gtid, int gthreads) {
    const int64_t esrc = FROM_OBS ? w.obs : w.err;
    for (int p = gtid; p < w.Np; p += gthreads) {
        double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
        const int e0 = I[w.pt_start + p], e1 = I[w.pt_start + p + 1];
        /* software pipeline over the point's edges: flag, camera id, residual and information of the NEXT edge are
           fetched while the current one waits for its pose (one dependent round trip per edge instead of three) */
        uint8_t fn = 0;
        int cn = 0;
        double er0n = 0, er1n = 0, infon = 0;
        if (e0 < e1) {
            fn = B[w.flags + e0];
            cn = I[w.cam + e0];
            er0n = D[esrc + (int64_t)e0 * 2];
            er1n = D[esrc + (int64_t)e0 * 2 + 1];
            infon = D[w.info + e0];
        }
        for (int e = e0; e < e1; e++) {
            const uint8_t f = fn;
            const int c = cn;
            double er0 = er0n, er1 = er1n;
            const double info = infon;
            if (e + 1 < e1) {
                fn = B[w.flags + e + 1];
                cn = I[w.cam + e + 1];
                er0n = D[esrc + (int64_t)(e + 1) * 2];
                er1n = D[esrc + (int64_t)(e + 1) * 2 + 1];
                infon = D[w.info + e + 1];
            }
            if (!(f & 1)) continue;
            const double *pose = D + w.poses + (int64_t)c * 7;
            const double *X = D + w.pts + (int64_t)p * 3;
            double R[9], pc[3];
            quat_to_R(pose, R);
            pc[0] = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + pose[4];
            pc[1] = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + pose[5];
            pc[2] = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + pose[6];
            if (FROM_OBS) { /* (er0, er1) held the observation */
                er0 = er0 - (pc[0] / pc[2] * w.fx + w.cx);
                er1 = er1 - (pc[1] / pc[2] * w.fy + w.cy);
            }
            double Ji[6], Jj[12];
            jacobians(pose, pc, w.fx, w.fy, Ji, Jj);
            double wgt = info, r0 = -info * er0, r1 = -info * er1;
            if (f & 2) {
                const double rw = huber_w((er0 * er0 + er1 * er1) * info, w.delta);
                wgt *= rw;
                r0 *= rw;
                r1 *= rw;
            }
            for (int a = 0; a < 3; a++) {
                b[a] += Ji[a] * r0 + Ji[3 + a] * r1;
                for (int bb = 0; bb < 3; bb++) H[a * 3 + bb] += wgt * (Ji[a] * Ji[bb] + Ji[3 + a] * Ji[3 + bb]);
            }
        }
        for (int i = 0; i < 9; i++) D[w.Hll + (int64_t)p * 9 + i] = H[i];
        for (int i = 0; i < 3; i++) D[w.bl + (int64_t)p * 3 + i] = b[i];
    }
}
/* Hpp, bp, W = Hpl: one wavefront per free camera, lanes over its edges, shuffle reduction.  (The W record of an edge of a
   FIXED camera is never read: the Schur lists, g and the back-substitution only follow free cameras.) */
template <bool COMPACT = false, bool FROM_OBS = false>
__device__ __forceinline__ void ba_phase_hpp(const BAWin &w, double *D, const int32_t *I, const uint8_t *B, int gwave, int gwaves, int lane) {
    for (int c = gwave; c < w.Nc; c += gwaves) {
        const int sl = I[w.slot + c];
        if (sl < 0) continue;
        double acc[27];
        for (int i = 0; i < 27; i++) acc[i] = 0.0;
        const int k0 = I[w.cam_start + c], k1 = I[w.cam_start + c + 1];
        const double *pose = D + w.poses + (int64_t)c * 7;
        if (COMPACT && lane < 9) { /* the camera's rotation, for the consumers of the W records */
            double Rm[9];
            quat_to_R(pose, Rm);
            double v = Rm[0];
#pragma unroll
            for (int i = 1; i < 9; i++) v = lane == i ? Rm[i] : v;
            D[w.Rc + (int64_t)c * 9 + lane] = v;
        }
        for (int k = k0 + lane; k < k1; k += 64) {
            const int e = I[w.cam_edges + k];
            const uint8_t f = B[w.flags + e];
            double2 *We = (double2 *)(D + w.W + (int64_t)e * (COMPACT ? 4 : 18));
            if (!(f & 1)) { /* switched off: contributes nothing, but its record is read by the block products */
#pragma unroll
                for (int i = 0; i < (COMPACT ? 2 : 9); i++) We[i] = double2{0.0, 0.0};
                continue;
            }
            const double *X = D + w.pts + (int64_t)I[w.pt + e] * 3;
            double R[9], pc[3];
            quat_to_R(pose, R);
            pc[0] = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + pose[4];
            pc[1] = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + pose[5];
            pc[2] = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + pose[6];
            double Ji[6], Jj[12];
            jacobians(pose, pc, w.fx, w.fy, Ji, Jj);
            double er0 = D[(FROM_OBS ? w.obs : w.err) + (int64_t)e * 2], er1 = D[(FROM_OBS ? w.obs : w.err) + (int64_t)e * 2 + 1];
            if (FROM_OBS) { /* the observation: the error from the projection in hand (ba_phase_hll has the note) */
                er0 = er0 - (pc[0] / pc[2] * w.fx + w.cx);
                er1 = er1 - (pc[1] / pc[2] * w.fy + w.cy);
            }
            const double info = D[w.info + e];
            double wgt = info, r0 = -info * er0, r1 = -info * er1;
            if (f & 2) {
                const double rw = huber_w((er0 * er0 + er1 * er1) * info, w.delta);
                wgt *= rw;
                r0 *= rw;
                r1 *= rw;
            }
            int q = 0;
            for (int a = 0; a < 6; a++)
                for (int bb = a; bb < 6; bb++) acc[q++] += wgt * (Jj[a] * Jj[bb] + Jj[6 + a] * Jj[6 + bb]);
            for (int a = 0; a < 6; a++) acc[21 + a] += Jj[a] * r0 + Jj[6 + a] * r1;
            if (COMPACT) {
                const double iz = 1.0 / pc[2];
                We[0] = double2{pc[0] * iz, pc[1] * iz};
                We[1] = double2{iz, wgt};
            } else {
                double wv[18];
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int bb = 0; bb < 3; bb++) wv[a * 3 + bb] = wgt * (Jj[a] * Ji[bb] + Jj[6 + a] * Ji[3 + bb]);
#pragma unroll
                for (int i = 0; i < 9; i++) We[i] = double2{wv[2 * i], wv[2 * i + 1]};
            }
        }
        /* sum of the 27 partials over the 64 lanes: recursive halving across the four rows of 16 (28 -> 14 -> 7 values,
           21 f64 exchanges through the LDS crossbar instead of 54), then row-local DPP moves.  Lane 0 ends up with
           entries 0 .. 6, lanes 16, 32, 48 with the other three groups of 7 (see the store below). */
        double a14[14], a7[7];
        const bool h32 = lane & 32, h16 = lane & 16;
#pragma unroll
        for (int i = 0; i < 14; i++) {
            const double lo = acc[i], hi = (14 + i < 27) ? acc[14 + i] : 0.0;
            a14[i] = (h32 ? hi : lo) + __shfl_xor(h32 ? lo : hi, 32);
        }
#pragma unroll
        for (int i = 0; i < 7; i++) a7[i] = (h16 ? a14[7 + i] : a14[i]) + __shfl_xor(h16 ? a14[i] : a14[7 + i], 16);
#pragma unroll
        for (int i = 0; i < 7; i++) {
            double v = a7[i];
            v += xchg16<8>(v);
            v += xchg16<4>(v);
            v += xchg16<2>(v);
            v += xchg16<1>(v);
            a7[i] = v;
        }
        if ((lane & 15) == 0) {
            const int base = (h32 ? 14 : 0) + (h16 ? 7 : 0);
#pragma unroll
            for (int i = 0; i < 7; i++) {
                const int q = base + i;
                if (q >= 27) continue;
                if (q >= 21) {
                    D[w.bp + (int64_t)sl * 6 + (q - 21)] = a7[i];
                } else {
                    /* q-th pair (a, bb) of the upper triangle, row-major */
                    int a = 0, rem = q;
                    while (rem >= 6 - a) {
                        rem -= 6 - a;
                        a++;
                    }
                    const int bb = a + rem;
                    D[w.Hpp + (int64_t)sl * 36 + a * 6 + bb] = a7[i];
                    D[w.Hpp + (int64_t)sl * 36 + bb * 6 + a] = a7[i];
                }
            }
        }
    }
}
/* computeLambdaInit: this thread's share of the largest diagonal entry of the Hessian */
__device__ __forceinline__ double ba_phase_lambda0(const BAWin &w, const double *D, int gtid, int gthreads) {
    double m = 0;
    for (int i = gtid; i < w.Ncf * 6; i += gthreads) m = fmax(m, fabs(D[w.Hpp + (int64_t)(i / 6) * 36 + (i % 6) * 7]));
    for (int i = gtid; i < w.Np * 3; i += gthreads) m = fmax(m, fabs(D[w.Hll + (int64_t)(i / 3) * 9 + (i % 3) * 4]));
    return m;
}
/* Dinv = (Hll + lambda I)^-1 and tv = Dinv bl (thread per point); S := Hpp + lambda I on its diagonal blocks, g := bp.
   Y = W Dinv is never stored: the Schur products form it in registers from W and Dinv (54 FMAs per contribution against
   a 144-byte record written, read back ~6 times and mostly missing every cache when 256 windows run at once). */
/* PARTS: the two-part Schur pass starts the blocks it touches itself (BA_BLK_FIRST); only the others are written here */
template <bool PARTS = false>
__device__ __forceinline__ void ba_phase_dinv(const BAWin &w, double *D, const int32_t *I, double lambda, int gtid, int gthreads) {
    /* PARTS (a chunked window on the phase-kernel path): Dinv and tv are not formed at all -- the Schur pass works with the
       factor U of ba_phase_ufactor (Dinv = U U^T, u = U^T bl) and the back-substitution applies U (U^T c) -- 96 bytes per
       point and trial that were written and read for nothing */
    for (int p = gtid; p < (PARTS ? 0 : w.Np); p += gthreads) {
        double Dm[9];
        for (int i = 0; i < 9; i++) Dm[i] = D[w.Hll + (int64_t)p * 9 + i];
        Dm[0] += lambda;
        Dm[4] += lambda;
        Dm[8] += lambda;
        const double det = Dm[0] * (Dm[4] * Dm[8] - Dm[5] * Dm[7]) - Dm[1] * (Dm[3] * Dm[8] - Dm[5] * Dm[6]) +
                           Dm[2] * (Dm[3] * Dm[7] - Dm[4] * Dm[6]);
        const double id = 1.0 / det;
        double Iv[9];
        Iv[0] = (Dm[4] * Dm[8] - Dm[5] * Dm[7]) * id;
        Iv[1] = (Dm[2] * Dm[7] - Dm[1] * Dm[8]) * id;
        Iv[2] = (Dm[1] * Dm[5] - Dm[2] * Dm[4]) * id;
        Iv[3] = (Dm[5] * Dm[6] - Dm[3] * Dm[8]) * id;
        Iv[4] = (Dm[0] * Dm[8] - Dm[2] * Dm[6]) * id;
        Iv[5] = (Dm[2] * Dm[3] - Dm[0] * Dm[5]) * id;
        Iv[6] = (Dm[3] * Dm[7] - Dm[4] * Dm[6]) * id;
        Iv[7] = (Dm[1] * Dm[6] - Dm[0] * Dm[7]) * id;
        Iv[8] = (Dm[0] * Dm[4] - Dm[1] * Dm[3]) * id;
        for (int i = 0; i < 9; i++) D[w.Dinv + (int64_t)p * 9 + i] = Iv[i];
        const double b0 = D[w.bl + (int64_t)p * 3], b1 = D[w.bl + (int64_t)p * 3 + 1], b2 = D[w.bl + (int64_t)p * 3 + 2];
        for (int a = 0; a < 3; a++) D[w.tv + (int64_t)p * 3 + a] = Iv[a * 3] * b0 + Iv[a * 3 + 1] * b1 + Iv[a * 3 + 2] * b2;
    }
    /* S starts as Hpp + lambda I on its diagonal blocks and zero elsewhere, g as bp: the Schur products are
       subtracted from them below (disjoint writers, no barrier needed between the two loops) */
    for (int i = gtid; i < w.nnzb * 36; i += gthreads) {
        const int bk = i / 36, e = i - bk * 36;
        if (PARTS && (I[w.blk_touch + bk] & 1)) continue;
        const int sl = I[w.blk_rowid + bk];
        if (PARTS && I[w.blk_col + bk] < sl) continue; /* (the Cholesky assembly reads the blocks on or above the diagonal only) */
        double v = 0.0;
        if (I[w.blk_col + bk] == sl) {
            v = D[w.Hpp + (int64_t)sl * 36 + e];
            if (e % 7 == 0) v += lambda;
        }
        D[w.S + i] = v;
    }
    for (int i = gtid; i < w.Ncf * 6; i += gthreads) D[w.g + i] = D[w.bp + i];
}
/* back-substitution xl = Dinv (bl - sum_e W_e^T x_c) (thread per point) and the trial estimate; returns this thread's
   share of the gain-ratio denominator */
template <bool COMPACT = false>
__device__ __forceinline__ double ba_phase_backsub(const BAWin &w, double *D, const int32_t *I, bool ok2, double lambda, int gtid, int gthreads) {
    double scale_loc = 0;
    for (int p = gtid; p < w.Np; p += gthreads) {
        double cl[3] = {D[w.bl + (int64_t)p * 3], D[w.bl + (int64_t)p * 3 + 1], D[w.bl + (int64_t)p * 3 + 2]};
        const int e0 = I[w.pt_start + p], e1 = I[w.pt_start + p + 1];
        /* camera id -> slot -> operands: three dependent round trips per edge; the two index levels of BS_U edges go out
           side by side (a point has ~6 edges: two groups instead of a dozen dependent index loads) */
        constexpr int BS_U = 4;
        for (int eb = e0; eb < e1; eb += BS_U) {
            int cu[BS_U], su[BS_U];
#pragma unroll
            for (int u = 0; u < BS_U; u++) cu[u] = eb + u < e1 ? I[w.cam + eb + u] : -1;
#pragma unroll
            for (int u = 0; u < BS_U; u++) su[u] = cu[u] >= 0 ? I[w.slot + cu[u]] : -1;
#pragma unroll
            for (int u = 0; u < BS_U; u++) {
            const int e = eb + u, c = cu[u], sl = su[u];
            if (sl < 0) continue;
            const double *xc = D + w.x + sl * 6;
            if (COMPACT) {
                const double2 *rc = (const double2 *)(D + w.W + (int64_t)e * 4);
                const double2 r0 = rc[0], r1 = rc[1];
                const double rec[4] = {r0.x, r0.y, r1.x, r1.y};
                double Wr[18];
                ba_w_from_record(rec, D + w.Rc + (int64_t)c * 9, w.fx, w.fy, Wr);
                for (int bb = 0; bb < 3; bb++)
                    for (int a = 0; a < 6; a++) cl[bb] -= Wr[a * 3 + bb] * xc[a];
            } else {
                const double *We = D + w.W + (int64_t)e * 18;
                for (int bb = 0; bb < 3; bb++)
                    for (int a = 0; a < 6; a++) cl[bb] -= We[a * 3 + bb] * xc[a];
            }
            }
        }
        double vv[3];
        if (COMPACT) { /* Dinv c = U (U^T c), U upper triangular (ba_phase_ufactor): the window keeps no Dinv */
            const double *U = D + w.Uf + (int64_t)p * 6;
            const double u0 = U[0], u1 = U[1], u2 = U[2], u3 = U[3], u4 = U[4], u5 = U[5];
            const double t0 = u0 * cl[0], t1 = u1 * cl[0] + u3 * cl[1], t2 = u2 * cl[0] + u4 * cl[1] + u5 * cl[2];
            vv[0] = u0 * t0 + u1 * t1 + u2 * t2;
            vv[1] = u3 * t1 + u4 * t2;
            vv[2] = u5 * t2;
        } else {
            const double *Iv = D + w.Dinv + (int64_t)p * 9;
            for (int a = 0; a < 3; a++) vv[a] = Iv[a * 3] * cl[0] + Iv[a * 3 + 1] * cl[1] + Iv[a * 3 + 2] * cl[2];
        }
        for (int a = 0; a < 3; a++) {
            const double v = ok2 ? vv[a] : 0.0;
            D[w.pts_t + (int64_t)p * 3 + a] = D[w.pts + (int64_t)p * 3 + a] + v;
            scale_loc += v * (lambda * v + D[w.bl + (int64_t)p * 3 + a]);
        }
    }
    for (int c = gtid; c < w.Nc; c += gthreads) {
        const int sl = I[w.slot + c];
        if (sl >= 0 && ok2) {
            double u[6];
            for (int a = 0; a < 6; a++) {
                u[a] = D[w.x + sl * 6 + a];
                scale_loc += u[a] * (lambda * u[a] + D[w.bp + (int64_t)sl * 6 + a]);
            }
            se3_exp_mul(u, D + w.poses + (int64_t)c * 7, D + w.poses_t + (int64_t)c * 7);
        } else {
            for (int a = 0; a < 7; a++) D[w.poses_t + (int64_t)c * 7 + a] = D[w.poses + (int64_t)c * 7 + a];
        }
    }
    return scale_loc;
}
/* Dinv = (Hll + lambda I)^-1 factored as U U^T (U upper triangular: the inverse transpose of the Cholesky factor of
   Hll + lambda I) and u = U^T bl.  With Z_e = W_e U the Schur products become S -= sum Z_a Z_b^T and g -= sum Z_e u:
   W Dinv is formed once per EDGE (36 FMAs at staging) instead of once per contribution (54 in every block product),
   and a contribution reads 36 doubles from LDS instead of 45.  A point whose matrix is not positive definite in
   floating point gets NaNs, which fail the Cholesky of S and send the LM trial to a larger lambda -- what an
   indefinite Dinv does on the other path. */
__device__ __forceinline__ void ba_phase_ufactor(const BAWin &w, double *D, double lambda, int gtid, int gthreads) {
    for (int p = gtid; p < w.Np; p += gthreads) {
        const double *H = D + w.Hll + (int64_t)p * 9;
        const double a00 = H[0] + lambda, a10 = H[3], a20 = H[6], a11 = H[4] + lambda, a21 = H[7], a22 = H[8] + lambda;
        const double nan_ = __builtin_nan("");
        double U[6] = {nan_, nan_, nan_, nan_, nan_, nan_};
        const double c00 = sqrt(a00), c10 = a10 / c00, c20 = a20 / c00;
        const double d11 = a11 - c10 * c10;
        const double c11 = sqrt(d11), c21 = (a21 - c20 * c10) / c11;
        const double d22 = a22 - c20 * c20 - c21 * c21;
        const double c22 = sqrt(d22);
        if (a00 > 0.0 && d11 > 0.0 && d22 > 0.0 && isfinite(c22)) {
            const double i00 = 1.0 / c00, i11 = 1.0 / c11, i22 = 1.0 / c22;
            const double i10 = -c10 * i00 * i11, i21 = -c21 * i11 * i22, i20 = -(c20 * i00 + c21 * i10) * i22;
            U[0] = i00;
            U[1] = i10;
            U[2] = i20;
            U[3] = i11;
            U[4] = i21;
            U[5] = i22;
        }
        double *Ud = D + w.Uf + (int64_t)p * 6;
#pragma unroll
        for (int i = 0; i < 6; i++) Ud[i] = U[i];
        const double b0 = D[w.bl + (int64_t)p * 3], b1 = D[w.bl + (int64_t)p * 3 + 1], b2 = D[w.bl + (int64_t)p * 3 + 2];
        D[w.tu + (int64_t)p * 3] = U[0] * b0;
        D[w.tu + (int64_t)p * 3 + 1] = U[1] * b0 + U[3] * b1;
        D[w.tu + (int64_t)p * 3 + 2] = U[2] * b0 + U[4] * b1 + U[5] * b2;
    }
}
/* the second accumulator of g starts at zero (S2 needs no fill: part 1 starts the blocks it touches, the assembly
   reads no others) */
__device__ __forceinline__ void ba_phase_zero2(const BAWin &w, double *D, int gtid, int gthreads) {
    for (int i = gtid; i < w.Ncf * 6; i += gthreads) D[w.g2 + i] = 0.0;
}
/* accepted trial: it becomes the estimate */
template <bool COPY_ERR = true>
__device__ __forceinline__ void ba_phase_accept(const BAWin &w, double *D, int gtid, int gthreads) {
    for (int i = gtid; i < w.Nc * 7; i += gthreads) D[w.poses + i] = D[w.poses_t + i];
    for (int i = gtid; i < w.Np * 3; i += gthreads) D[w.pts + i] = D[w.pts_t + i];
    if (COPY_ERR)
        for (int i = gtid; i < w.E * 2; i += gthreads) D[w.err + i] = D[w.err_t + i];
}

/* n_iters LM iterations (OptimizationAlgorithmLevenberg::solve) for every window of the batch.  A whole optimize()
 * round runs in ONE launch unless a stop flag has to be polled by the host between iterations: the windows of a batch
 * then advance independently (a window that needs a second LM trial in some iteration does not hold up the other 255,
 * which a launch per iteration did -- the launch lasted as long as its slowest window). */
__global__ __launch_bounds__(BA_T) void k_ba_iteration(BABufs bufs, int first_of_round, double pcg_tol2, int pcg_maxit,
                                                       int debug, int solver, int dyn_doubles, int G, int launch_seq,
                                                       int n_iters, int band) {
    __shared__ BAShared s;
    extern __shared__ __attribute__((aligned(16))) double ba_dyn[];
    /* workgroups [win * G, win * G + G) are the team of window `win` */
    const int win = __builtin_amdgcn_readfirstlane((int)blockIdx.x / G);
    const BAWin &w = bufs.wins[win]; /* by reference: the ~110-dword descriptor stays in memory (scalar cache), not in registers */
    double *D = bufs.D;
    const int32_t *I = bufs.I;
    const uint8_t *B = bufs.B;
    double *st = D + w.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6); /* uniform: scalar unit */
    BATeam team;
    team.G = G;
    team.part = __builtin_amdgcn_readfirstlane((int)blockIdx.x - win * G);
    team.bar = (unsigned *)(bufs.I + w.gbar) + (launch_seq & 1);
    team.abort = (unsigned *)(bufs.I + w.gbar) + 2;
    team.epoch = 0;
    team.red = D + w.gred;
    team.slot = 0;
    team.dead = false;
    /* the OTHER arrival counter serves the next launch and nobody is using it now; every launch clears it, also one
       that returns on the next line, so that a launch always finds its own counter at zero */
    if (G > 1 && team.part == 0 && tid == 0)
        __hip_atomic_store((unsigned *)(bufs.I + w.gbar) + ((launch_seq + 1) & 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st[ST_TERM] != 0.0) return; /* uniform over the team: this window's round already terminated */
    const int gtid = team.part * BA_T + tid, gthreads = G * BA_T; /* thread / wavefront index within the team */
    const int gwave = team.part * BA_W + wave, gwaves = G * BA_W;
    const bool lead = team.part == 0;
    const int n = w.Ncf * 6;
    /* profile mode: thread 0 accumulates wall time per phase (diagnostic build path only, never in timed runs) */
    unsigned long long t_prev = debug ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define BA_STAMP(slot)                                                        \
    if (debug) {                                                              \
        __syncthreads();                                                      \
        if (tid == 0 && blockIdx.x % G == 0) {                                \
            const unsigned long long t_now = __builtin_amdgcn_s_memrealtime(); \
            st[ST_T0 + (slot)] += (double)(t_now - t_prev);                   \
            t_prev = t_now;                                                   \
        }                                                                     \
    }

    double lambda = st[ST_LAMBDA], ni = st[ST_NI]; /* uniform over the team from here on (every sum is a team sum) */
    double currentChi = 0.0;
    bool have_chi = false; /* the previous iteration of this launch was accepted: its trial errors and chi2 ARE the current
                              ones (same edges, same estimate, same summation order), no need to evaluate them again */
    for (int iter = 0; iter < n_iters; iter++) {
    /* ---- computeActiveErrors + activeRobustChi2 ---- */
    if (!have_chi) {
        currentChi = team_sum(eval_chi2_partial(w, D, I, B, w.poses, w.pts, w.err, team), s, team);
        if (lead && tid == 0 && st[ST_HAVE_CHI0] == 0.0) {
            st[ST_CHI0] = currentChi;
            st[ST_HAVE_CHI0] = 1.0;
        }
        if (G == 1) __syncthreads(); /* (the team's sum already ended in a barrier) */
    }

    BA_STAMP(0)
    /* ---- buildSystem: Hll, bl (thread per point; W is written by the camera pass below, which has the same Jacobians
       in hand and stores whole 144-byte records instead of 18 doubles strided over the lanes) ---- */
    ba_phase_hll(w, D, I, B, gtid, gthreads);
    BA_STAMP(1)
    /* ---- Hpp, bp, W = Hpl: one wavefront per free camera, lanes over its edges, shuffle reduction.  (The W record of an
       edge of a FIXED camera is never read: the Schur lists, g and the back-substitution only follow free cameras.) ---- */
    ba_phase_hpp(w, D, I, B, gwave, gwaves, lane);
    team_sync(team, s);

    BA_STAMP(2)
    /* ---- computeLambdaInit on the first iteration of an optimize() call ---- */
    if (first_of_round && iter == 0) {
        double m = ba_phase_lambda0(w, D, gtid, gthreads);
        m = team_max(m, s, team);
        lambda = 1e-5 * m;
        ni = 2;
    }

    double rho = 0;
    int qmax = 0;
    int solves = 0, pcg_total = 0;
    do {
        /* ---- Dinv = (Hll + lambda I)^-1 and tv = Dinv bl (thread per point).  Y = W Dinv is never stored: the Schur
           products form it in registers from W and Dinv (54 FMAs per contribution against a 144-byte record written,
           read back ~6 times and mostly missing every cache when 256 windows run at once) ---- */
        ba_phase_dinv(w, D, I, lambda, gtid, gthreads);
        team_sync(team, s);
        BA_STAMP(3)
        if (w.nchunks > 0 && G == 1) {
            ba_schur_chunked(w, D, I, ba_dyn, debug);
            BA_STAMP(4)
        } else {
            ba_schur_lists(w, D, I, gwave, gwaves);
            BA_STAMP(4)
            ba_g_lists(w, D, I, gwave, gwaves);
        }
        team_sync(team, s);
        BA_STAMP(5)
        const int np_ = (n + 15) & ~15;
        const bool use_chol = solver == 1 && n > 0 && w.ch_pm > 0 && (w.ch_pm * CH_LD + 2 * 16 * CH_LD + np_) <= dyn_doubles;
        /* band Cholesky with the LDS-resident window (ba_chol_band): no dense matrix to assemble, the lead workgroup fills its
           tiles straight from S.  `band` (a launch argument) says that the fill plans exist (k_ba_ph_fillplan has run). */
        const bool use_band = use_chol && band && debug <= 1 && w.fplan >= 0 && (w.ch_pm >> 4) <= CHB_MAX_M &&
                              chb_lds_doubles(w.ch_pm >> 4, np_) <= (size_t)dyn_doubles;
        if (!use_chol || debug > 1) {
            /* the Cholesky reads the upper blocks only; PCG (and the debug residual) need the full symmetric S */
            for (int i = gtid; i < w.nnzb * 36; i += gthreads) {
                const int bk = i / 36, e = i - bk * 36;
                const int mk = I[w.blk_mirror + bk];
                if (mk < bk) D[w.S + i] = D[w.S + (int64_t)mk * 36 + (e % 6) * 6 + e / 6];
            }
            team_sync(team, s);
        }
        if (use_chol && !use_band) { /* dense matrix of the reduced system: assembled by the whole team */
            ba_chol_assemble(w, D, I, n, gtid, gthreads, gwave, gwaves, [&] { team_sync(team, s); });
            team_sync(team, s);
        }
        bool ok2 = true;
        int it = 0;
        if (lead) { /* the linear solve runs on the team's first workgroup; the others wait at the barrier below */
        /* linear solver of the reduced camera system: dense blocked Cholesky (MFMA f64) when its LDS panel fits,
           PCG otherwise / on request */
        if (use_chol) {
            BA_STAMP(5)
            if (use_band)
                ok2 = (w.ch_pm >> 4) <= 4 ? ba_chol_band_fused4(w, D, I, n, ba_dyn, &s.bc[0]) : ba_chol_band_fused8(w, D, I, n, ba_dyn, &s.bc[0]);
            else
                ok2 = G > 1 ? ba_chol_solve<false, true>(w, D, I, n, ba_dyn, &s.bc[0], debug) : ba_chol_solve<false, false>(w, D, I, n, ba_dyn, &s.bc[0], debug);
            solves++;
            if (debug > 1) { /* residual of the direct solve against the block-sparse S */
                double r2 = 0, g2 = 0;
                for (int i = tid; i < n; i += BA_T) {
                    const int sl_ = i / 6, a_ = i % 6;
                    double v = 0;
                    for (int bk = I[w.blk_row + sl_]; bk < I[w.blk_row + sl_ + 1]; bk++)
                        for (int c = 0; c < 6; c++) v += D[w.S + (int64_t)bk * 36 + a_ * 6 + c] * D[w.x + I[w.blk_col + bk] * 6 + c];
                    const double gi_ = D[w.g + i];
                    r2 += (v - gi_) * (v - gi_);
                    g2 += gi_ * gi_;
                }
                r2 = block_sum(r2, s);
                g2 = block_sum(g2, s);
                if (tid == 0) printf("[dev] chol residual |Sx-g|/|g| = %.3e (ok %d, n %d)\n", sqrt(r2 / g2), (int)ok2, n);
            }
            BA_STAMP(6)
        } else {
        /* ---- block-Jacobi preconditioner: inverse of the 6x6 diagonal blocks (thread per camera) ---- */
        bool ok_local = true;
        for (int sl = tid; sl < w.Ncf; sl += BA_T) {
            double A[36], Linv[36];
            const double *Sd = D + w.S + (int64_t)I[w.blk_diag + sl] * 36;
            for (int i = 0; i < 36; i++) A[i] = Sd[i];
            /* Cholesky A = L L^T */
            for (int j = 0; j < 6; j++) {
                double d = A[j * 6 + j];
                for (int k = 0; k < j; k++) d -= A[j * 6 + k] * A[j * 6 + k];
                if (!(d > 0)) {
                    ok_local = false;
                    d = 1.0;
                }
                d = sqrt(d);
                A[j * 6 + j] = d;
                for (int i = j + 1; i < 6; i++) {
                    double v = A[i * 6 + j];
                    for (int k = 0; k < j; k++) v -= A[i * 6 + k] * A[j * 6 + k];
                    A[i * 6 + j] = v / d;
                }
            }
            /* Linv */
            for (int i = 0; i < 36; i++) Linv[i] = 0;
            for (int j = 0; j < 6; j++) {
                Linv[j * 6 + j] = 1.0 / A[j * 6 + j];
                for (int i = j + 1; i < 6; i++) {
                    double v = 0;
                    for (int k = j; k < i; k++) v -= A[i * 6 + k] * Linv[k * 6 + j];
                    Linv[i * 6 + j] = v / A[i * 6 + i];
                }
            }
            double *Mi = D + w.Minv + (int64_t)sl * 36;
            for (int i = 0; i < 6; i++)
                for (int j = 0; j < 6; j++) {
                    double v = 0;
                    for (int k = (i > j ? i : j); k < 6; k++) v += Linv[k * 6 + i] * Linv[k * 6 + j];
                    Mi[i * 6 + j] = v;
                }
        }
        BA_STAMP(6)
        if (n > BA_MAX_N) {
            __syncthreads();
            ok2 = ba_pcg_global(w, D, I, n, pcg_tol2, pcg_maxit, ok_local, &it, s);
            pcg_total += it;
            solves++;
        } else {
        /* ---- PCG on S x = g ---- */
        double xi[3] = {0, 0, 0}, ri[3], zi[3], pi[3]; /* each thread owns rows tid, tid+BA_T, tid+2*BA_T */
        double rz_loc = 0;
        __syncthreads();
        for (int q = 0; q < 3; q++) {
            const int i = tid + q * BA_T;
            ri[q] = (i < n) ? D[w.g + i] : 0.0;
            if (i < n) s.vr[i] = ri[q];
        }
        __syncthreads();
        for (int q = 0; q < 3; q++) {
            const int i = tid + q * BA_T;
            zi[q] = 0;
            if (i < n) {
                const int sl = i / 6, a = i % 6;
                const double *Mi = D + w.Minv + (int64_t)sl * 36 + a * 6;
                double v = 0;
                for (int c = 0; c < 6; c++) v += Mi[c] * s.vr[sl * 6 + c];
                zi[q] = v;
                s.vp[i] = v;
            }
            pi[q] = zi[q];
            rz_loc += ri[q] * zi[q];
        }
        double rz = block_sum(rz_loc, s);
        const double rz0 = rz;
        ok2 = (block_sum(ok_local ? 0.0 : 1.0, s) == 0.0);
        it = 0;
        if (n > 0 && ok2 && rz0 > 0) {
            for (it = 0; it < pcg_maxit; it++) {
                /* Ap = S p : block-CSR row product, p read from LDS */
                double Ap[3], pAp_loc = 0;
                for (int q = 0; q < 3; q++) {
                    const int i = tid + q * BA_T;
                    Ap[q] = 0;
                    if (i < n) {
                        const int sl = i / 6, a = i % 6;
                        const int b0 = I[w.blk_row + sl], b1 = I[w.blk_row + sl + 1];
                        double v = 0;
#pragma unroll 4
                        for (int bk = b0; bk < b1; bk++) {
                            const double2 *Sr = (const double2 *)(D + w.S + (int64_t)bk * 36 + a * 6); /* 16-B aligned */
                            const double *pv = s.vp + I[w.blk_col + bk] * 6;
                            const double2 s0 = Sr[0], s1 = Sr[1], s2 = Sr[2];
                            v += s0.x * pv[0] + s0.y * pv[1] + s1.x * pv[2] + s1.y * pv[3] + s2.x * pv[4] + s2.y * pv[5];
                        }
                        Ap[q] = v;
                        pAp_loc += pi[q] * v;
                    }
                }
                const double pAp = block_sum(pAp_loc, s);
                if (!(pAp > 0)) { /* not positive definite: same outcome as a failed Cholesky */
                    ok2 = false;
                    break;
                }
                const double alpha = rz / pAp;
                for (int q = 0; q < 3; q++) {
                    const int i = tid + q * BA_T;
                    xi[q] += alpha * pi[q];
                    ri[q] -= alpha * Ap[q];
                    if (i < n) s.vr[i] = ri[q];
                }
                __syncthreads();
                double rz_new_loc = 0;
                for (int q = 0; q < 3; q++) {
                    const int i = tid + q * BA_T;
                    if (i < n) {
                        const int sl = i / 6, a = i % 6;
                        const double *Mi = D + w.Minv + (int64_t)sl * 36 + a * 6;
                        double v = 0;
                        for (int c = 0; c < 6; c++) v += Mi[c] * s.vr[sl * 6 + c];
                        zi[q] = v;
                        rz_new_loc += ri[q] * v;
                    }
                }
                const double rz_new = block_sum(rz_new_loc, s);
                if (rz_new <= pcg_tol2 * rz0) {
                    it++;
                    break;
                }
                const double beta = rz_new / rz;
                rz = rz_new;
                for (int q = 0; q < 3; q++) {
                    const int i = tid + q * BA_T;
                    pi[q] = zi[q] + beta * pi[q];
                    if (i < n) s.vp[i] = pi[q];
                }
                __syncthreads();
            }
        }
        pcg_total += it;
        solves++;
        for (int q = 0; q < 3; q++) {
            const int i = tid + q * BA_T;
            if (i < n) D[w.x + i] = ok2 ? xi[q] : 0.0;
        }
        } /* LDS-resident PCG */
        }
        if (G > 1 && tid == 0) D[w.xl] = ok2 ? 1.0 : 0.0;
        } /* lead */
        team_sync(team, s);
        if (G > 1) ok2 = __hip_atomic_load(D + w.xl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0;
        BA_STAMP(7)
        /* ---- back-substitution: xl = Dinv (bl - sum_e W_e^T x_c) (thread per point) + trial update ---- */
        const double scale_loc = ba_phase_backsub(w, D, I, ok2, lambda, gtid, gthreads);
        team_sync(team, s);
        BA_STAMP(8)
        double tempChi = eval_chi2_partial(w, D, I, B, w.poses_t, w.pts_t, w.err_t, team);
        double scale = scale_loc;
        team_sum2(tempChi, scale, s, team);
        if (!ok2) tempChi = 1.7976931348623157e308;
        if (!ok2) scale = 0;
        scale += 1e-3;
        rho = (currentChi - tempChi) / scale;
        if (rho > 0 && isfinite(tempChi)) { /* accept: keep the trial state */
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = fmin(alpha, 2. / 3.);
            lambda *= fmax(1. / 3., alpha);
            ni = 2;
            currentChi = tempChi;
            have_chi = true;
            ba_phase_accept(w, D, gtid, gthreads);
        } else {
            lambda *= ni;
            ni *= 2;
        }
        qmax++;
        BA_STAMP(9)
        if (debug > 1 && tid == 0)
            printf("[dev] win %d trial %d ok %d pcg %d cur %.10e temp %.10e rho %.6e lambda %.6e\n", win, qmax,
                   (int)ok2, it, currentChi, tempChi, rho, lambda);
        __syncthreads();
    } while (rho < 0 && qmax < 10);

    if (lead && tid == 0) {
        st[ST_LAMBDA] = lambda;
        st[ST_NI] = ni;
        st[ST_CHI] = currentChi;
        st[ST_ITERS] += 1;
        st[ST_SOLVES] += solves;
        st[ST_PCG_ITERS] += pcg_total;
        if (qmax == 10 || rho == 0) st[ST_TERM] = 1.0; /* OptimizationAlgorithmLevenberg returns Terminate */
    }
    if (team.dead) {
        if (tid == 0) st[ST_ABORT] = 1.0;
        break;
    }
    if (qmax == 10 || rho == 0) break; /* uniform: terminated */
    if (iter + 1 < n_iters) team_sync(team, s); /* the accepted estimate (written team-strided) is read edge-strided next */
    } /* iterations of this launch */
}

/* ------------------------------------------------------------------------------------------------------------
 * The batch path as PHASE KERNELS.  k_ba_iteration keeps a window's whole optimize() round in one launch -- right
 * for a team working on one window, where a launch per phase would cost more than the phases -- but as ONE kernel
 * it is compiled for the union of its phases: 256 VGPRs plus 120 spilled, two wavefronts per SIMD, and every phase
 * runs at that occupancy.  With a few hundred windows in flight the launches are free, so the batch path runs an LM
 * trial as seven kernels over all windows, each with the registers, LDS and occupancy of its own phase:
 *     chi2 | Hll | Hpp + W | lambda0, Dinv, S := Hpp + lambda I | Schur | Cholesky | back-substitution, trial, decision
 * A window's position in g2o's control flow (which iteration, which trial of it, accepted or not) lives in its state
 * record (ST_PH ...); a workgroup whose window has nothing to do in a phase returns at once.  The host enqueues
 * n_iters + BA_PH_SPARE trials per round; a window that needed more rejected trials than that is finished by
 * k_ba_ph_finish, which loops the same phases inside one workgroup until the round is over (so the result never
 * depends on the number of trials enqueued).  Same arithmetic, same order as the fused kernel with a team of one. */
struct BARed {
    double red[BA_W];
    double bc[8];
};
#define BA_PH_SPARE 3 /* trials enqueued per round beyond its iterations (2 -> 3: 16.56 -> 16.16 ms per round of 256 windows; 4: 16.18) */
/* phase kernels from this many windows on: 6, or 4 when no window has BA_SPLIT_BIG_EDGES edges (ba_stage has the
   measurements) */
#define BA_SPLIT_MIN_WINDOWS 6
#define BA_SPLIT_MIN_SMALL 4
#define BA_SPLIT_BIG_EDGES 16000

__global__ void k_ba_ph_round(BABufs bufs, int nwin, int n_iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwin) return;
    double *st = bufs.D + bufs.wins[i].st;
    st[ST_PH] = n_iters > 0 ? 0.0 : 2.0;
    st[ST_BUDGET] = (double)n_iters;
    st[ST_FIRST] = 1.0;
    st[ST_HAVECHI] = 0.0;
    st[ST_QMAX] = 0.0;
    st[ST_ACCEPT] = 0.0;
}

/* computeActiveErrors + activeRobustChi2 at the current estimate (skipped when the last trial was accepted: its
   errors and chi2 are the current ones) */
template <class SH> __device__ __forceinline__ void ba_ph_chi2(const BAWin &w, double *D, const int32_t *I, const uint8_t *B, SH &s) {
    double *st = D + w.st;
    BATeam one;
    one.G = 1;
    one.part = 0;
    const double chi = block_sum(eval_chi2_partial(w, D, I, B, w.poses, w.pts, w.err, one), s);
    if (threadIdx.x == 0) {
        st[ST_CHI] = chi;
        if (st[ST_HAVE_CHI0] == 0.0) {
            st[ST_CHI0] = chi;
            st[ST_HAVE_CHI0] = 1.0;
        }
    }
}
/* computeLambdaInit on the first iteration of a round, then Dinv / tv and the start values of S and g */
template <class SH> __device__ __forceinline__ void ba_ph_dinv(const BAWin &w, double *D, const int32_t *I, SH &s) {
    double *st = D + w.st;
    const bool first = st[ST_FIRST] != 0.0, build = st[ST_PH] == 0.0;
    double lambda = st[ST_LAMBDA];
    if (first) lambda = 1e-5 * block_max(ba_phase_lambda0(w, D, threadIdx.x, BA_T), s); /* (its barriers order the reads above before the writes below) */
    if (threadIdx.x == 0) {
        if (first) {
            st[ST_LAMBDA] = lambda;
            st[ST_NI] = 2.0;
            st[ST_FIRST] = 0.0;
        }
        if (build) st[ST_QMAX] = 0.0;
    }
    if (w.nchunks > 0) {
        ba_phase_dinv<true>(w, D, I, lambda, threadIdx.x, BA_T);
        ba_phase_ufactor(w, D, lambda, threadIdx.x, BA_T);
    } else {
        ba_phase_dinv<false>(w, D, I, lambda, threadIdx.x, BA_T);
    }
}
__device__ __forceinline__ void ba_ph_chol(const BAWin &w, double *D, const int32_t *I, double *dyn, double *bad_flag, bool band = false) {
    double *st = D + w.st;
    const int n = w.Ncf * 6;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    bool ok2;
    if (band) { /* the batch runs the band Cholesky: the same solver (and the same arithmetic) for the trials finished here */
        ok2 = (w.ch_pm >> 4) <= 4 ? ba_chol_band_two4(w, D, I, n, dyn, bad_flag) : ba_chol_band_two8(w, D, I, n, dyn, bad_flag);
    } else {
        ba_chol_assemble<true>(w, D, I, n, threadIdx.x, BA_T, wave, BA_W, [&] { __syncthreads(); });
        __syncthreads();
        ok2 = ba_chol_solve<true>(w, D, I, n, dyn, bad_flag, 0);
    }
    if (threadIdx.x == 0) {
        st[ST_OK2] = ok2 ? 1.0 : 0.0;
        st[ST_SOLVES] += 1.0;
    }
}
/* back-substitution and trial estimate, its chi2, the gain ratio and g2o's accept / reject bookkeeping */
template <class SH> __device__ __forceinline__ void ba_ph_update(const BAWin &w, double *D, const int32_t *I, const uint8_t *B, SH &s) {
    double *st = D + w.st;
    double lambda = st[ST_LAMBDA], ni = st[ST_NI];
    const double currentChi = st[ST_CHI];
    const bool ok2 = st[ST_OK2] != 0.0;
    int qmax = (int)st[ST_QMAX], budget = (int)st[ST_BUDGET];
    double scale = w.nchunks > 0 ? ba_phase_backsub<true>(w, D, I, ok2, lambda, threadIdx.x, BA_T) : ba_phase_backsub<false>(w, D, I, ok2, lambda, threadIdx.x, BA_T);
    __syncthreads();
    BATeam one;
    one.G = 1;
    one.part = 0;
    double tempChi = block_sum(eval_chi2_partial(w, D, I, B, w.poses_t, w.pts_t, w.err_t, one), s);
    scale = block_sum(scale, s);
    if (!ok2) tempChi = 1.7976931348623157e308;
    if (!ok2) scale = 0;
    scale += 1e-3;
    const double rho = (currentChi - tempChi) / scale;
    const bool accept = rho > 0 && isfinite(tempChi);
    if (accept) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2;
        ba_phase_accept<false>(w, D, threadIdx.x, BA_T);
    } else {
        lambda *= ni;
        ni *= 2;
    }
    qmax++;
    if (threadIdx.x == 0) {
        st[ST_LAMBDA] = lambda;
        st[ST_NI] = ni;
        st[ST_QMAX] = (double)qmax;
        if (accept) {
            st[ST_CHI] = tempChi;
            st[ST_HAVECHI] = 1.0;
        }
        if (rho < 0 && qmax < 10) {
            st[ST_PH] = 1.0; /* same iteration, larger lambda */
        } else {
            st[ST_ITERS] += 1;
            if (qmax == 10 || rho == 0) { /* OptimizationAlgorithmLevenberg returns Terminate */
                st[ST_TERM] = 1.0;
                st[ST_PH] = 2.0;
            } else {
                budget--;
                st[ST_BUDGET] = (double)budget;
                st[ST_PH] = budget > 0 ? 0.0 : 2.0;
            }
        }
    }
}

/* Launch geometry of the phase kernels: grid (window, part).  The streaming phases are strided over the threads of
 * ALL parts of a window (the phase functions take the thread's index and the stride), so a phase fills the machine
 * with a few hundred windows however few threads one window can feed; sums over a window are per-part partial sums
 * (gred) added up in part order by the single-workgroup kernel that needs them: deterministic. */
#define BA_PH_WINDOW                        \
    const BAWin &w = bufs.wins[blockIdx.x]; \
    double *D = bufs.D;                     \
    const int32_t *I = bufs.I;              \
    const uint8_t *B = bufs.B;              \
    const double ph = D[w.st + ST_PH];      \
    (void)B;                                \
    (void)I;
/* the same with the window's descriptor COPIED: its fields are then read once, at kernel entry and through the scalar
   cache, and live in scalar registers.  Through the reference every use of a field after a store or a barrier is a fresh
   (vector) load from memory -- the compiler cannot know that no store reaches the descriptor -- and the address of
   nearly every load of a phase kernel hangs on one.  Only for kernels whose callees are all inlined (a non-inlined
   callee would take the copy through scratch memory). */
#define BA_PH_WINDOW_COPY                   \
    const BAWin w = bufs.wins[blockIdx.x];  \
    double *D = bufs.D;                     \
    const int32_t *I = bufs.I;              \
    const uint8_t *B = bufs.B;              \
    const double ph = D[w.st + ST_PH];      \
    (void)B;                                \
    (void)I;
#define BA_PH_T 256 /* threads per workgroup of the streaming phase kernels */
enum { PH_RED_CHI = 0, PH_RED_SCALE = BA_GMAX, PH_RED_TRIAL = 2 * BA_GMAX }; /* partial-sum rows in gred */
/* parts that SUM over a window's edges / points: a function of the window alone, so that its arithmetic (the order of
   its partial sums) does not depend on which other windows share the batch */
__host__ __device__ __forceinline__ int ba_ph_parts_e(int E) {
    const int n = (E + 4 * BA_PH_T - 1) / (4 * BA_PH_T);
    return n < 1 ? 1 : (n > BA_GMAX ? BA_GMAX : n);
}
__host__ __device__ __forceinline__ int ba_ph_parts_p(int Np) {
    const int n = (Np + BA_PH_T - 1) / BA_PH_T;
    return n < 1 ? 1 : (n > BA_GMAX ? BA_GMAX : n);
}

template <class SH> __device__ __forceinline__ double block_sum_n(double v, SH &s, int nwaves) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) s.red[wave] = v;
    __syncthreads();
    double t = 0;
    for (int k = 0; k < nwaves; k++) t += s.red[k];
    return t;
}
/* errors + robust chi2 of this part's edges at (poses, pts): the same two-edges-per-step loop as eval_chi2_partial,
   for a workgroup of BA_PH_T threads */
template <bool STORE = true>
__device__ __forceinline__ double ba_ph_chi2_part(const BAWin &w, double *D, const int32_t *I, const uint8_t *B, int64_t poses, int64_t pts,
                                                  int64_t err, int gtid, int gthreads) {
    double acc = 0;
    for (int e0 = gtid; e0 < w.E; e0 += 2 * gthreads) {
        const int eA = e0, eB = (e0 + gthreads < w.E) ? e0 + gthreads : e0;
        const bool vB = e0 + gthreads < w.E;
        const uint8_t fA = B[w.flags + eA], fB = B[w.flags + eB];
        const int cA = I[w.cam + eA], cB = I[w.cam + eB], pA = I[w.pt + eA], pB = I[w.pt + eB];
        const double iA = D[w.info + eA], iB = D[w.info + eB];
        double erA[2], erB[2], pc[3];
        project_err(D + poses + (int64_t)cA * 7, D + pts + (int64_t)pA * 3, D + w.obs + (int64_t)eA * 2, w.fx, w.fy, w.cx, w.cy, erA, pc);
        project_err(D + poses + (int64_t)cB * 7, D + pts + (int64_t)pB * 3, D + w.obs + (int64_t)eB * 2, w.fx, w.fy, w.cx, w.cy, erB, pc);
        if (fA & 1) {
            if (STORE) {
                D[err + (int64_t)eA * 2] = erA[0];
                D[err + (int64_t)eA * 2 + 1] = erA[1];
            }
            double c = (erA[0] * erA[0] + erA[1] * erA[1]) * iA;
            if (fA & 2) c = huber_rho0(c, w.delta);
            acc += c;
        }
        if (vB && (fB & 1)) {
            if (STORE) {
                D[err + (int64_t)eB * 2] = erB[0];
                D[err + (int64_t)eB * 2 + 1] = erB[1];
            }
            double c = (erB[0] * erB[0] + erB[1] * erB[1]) * iB;
            if (fB & 2) c = huber_rho0(c, w.delta);
            acc += c;
        }
    }
    return acc;
}

__global__ __launch_bounds__(BA_PH_T) void k_ba_ph_chi2(BABufs bufs) {
    __shared__ BARed s;
    BA_PH_WINDOW_COPY
    const int parts = ba_ph_parts_e(w.E);
    if (ph != 0.0 || D[w.st + ST_HAVECHI] != 0.0 || (int)blockIdx.y >= parts) return;
    const double v = block_sum_n(ba_ph_chi2_part<false>(w, D, I, B, w.poses, w.pts, w.err, blockIdx.y * BA_PH_T + threadIdx.x, parts * BA_PH_T), s, BA_PH_T / 64);
    if (threadIdx.x == 0) D[w.gred + PH_RED_CHI + blockIdx.y] = v;
}
__global__ __launch_bounds__(BA_PH_T) void k_ba_ph_hll(BABufs bufs) {
    BA_PH_WINDOW_COPY
    if (ph != 0.0) return;
    ba_phase_hll<true>(w, D, I, B, blockIdx.y * BA_PH_T + threadIdx.x, gridDim.y * BA_PH_T);
}
__global__ __launch_bounds__(BA_PH_T) void k_ba_ph_hpp(BABufs bufs) { /* one wavefront per camera */
    BA_PH_WINDOW_COPY
    if (ph != 0.0) return;
    const int gw = blockIdx.y * (BA_PH_T / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), gws = gridDim.y * (BA_PH_T / 64);
    if (w.nchunks > 0)
        ba_phase_hpp<true, true>(w, D, I, B, gw, gws, threadIdx.x & 63);  /* point chunks: 32-byte edge records */
    else
        ba_phase_hpp<false, true>(w, D, I, B, gw, gws, threadIdx.x & 63); /* per-block lists (long tracks): W itself */
}
/* single workgroup per window: the current chi2 from its partial sums, computeLambdaInit on the first iteration of a
   round, trial counter */
__global__ __launch_bounds__(BA_T) void k_ba_ph_prep(BABufs bufs) {
    __shared__ BARed s;
    BA_PH_WINDOW
    if (ph == 2.0) return;
    double *st = D + w.st;
    const bool first = st[ST_FIRST] != 0.0, build = ph == 0.0, need_chi = build && st[ST_HAVECHI] == 0.0;
    double lambda = 0;
    if (first) lambda = 1e-5 * block_max(ba_phase_lambda0(w, D, threadIdx.x, BA_T), s);
    if (threadIdx.x == 0) {
        if (need_chi) {
            double chi = 0;
            for (int g = 0; g < ba_ph_parts_e(w.E); g++) chi += D[w.gred + PH_RED_CHI + g];
            st[ST_CHI] = chi;
            if (st[ST_HAVE_CHI0] == 0.0) {
                st[ST_CHI0] = chi;
                st[ST_HAVE_CHI0] = 1.0;
            }
        }
        if (first) {
            st[ST_LAMBDA] = lambda;
            st[ST_NI] = 2.0;
            st[ST_FIRST] = 0.0;
        }
        if (build) st[ST_QMAX] = 0.0;
    }
}
__global__ __launch_bounds__(BA_PH_T) void k_ba_ph_dinv(BABufs bufs) {
    BA_PH_WINDOW_COPY
    if (ph == 2.0) return;
    if (w.nchunks > 0) {
        ba_phase_dinv<true>(w, D, I, D[w.st + ST_LAMBDA], blockIdx.y * BA_PH_T + threadIdx.x, gridDim.y * BA_PH_T);
        ba_phase_ufactor(w, D, D[w.st + ST_LAMBDA], blockIdx.y * BA_PH_T + threadIdx.x, gridDim.y * BA_PH_T);
    } else {
        ba_phase_dinv<false>(w, D, I, D[w.st + ST_LAMBDA], blockIdx.y * BA_PH_T + threadIdx.x, gridDim.y * BA_PH_T);
    }
    ba_phase_zero2(w, D, blockIdx.y * BA_PH_T + threadIdx.x, gridDim.y * BA_PH_T);
}
/* The Schur pass of the phase-kernel path: the window's point chunks are split over two groups of 256 threads,
 * part 0 subtracting its block products from S and g, part 1 from S2 and g2 (zeroed by the Dinv phase; the Cholesky
 * assembly adds the copies).  k_ba_ph_schur2 runs the parts as workgroups: two of them fit a CU (LDS 78 KB, 4
 * wavefronts each), so one streams its next chunk into LDS -- every load of the chunk in flight at once, written
 * through as it lands: no registers held across the products, which is what spilled the single-workgroup version's
 * prefetch -- while the other's lanes are in the block products.  The W tile is REBUILT from the 32-byte edge
 * records (ba_w_from_record): the pass reads 0.3 GB per 256 windows instead of 1.3.
 * The products themselves are those of ba_schur_chunked (a full 6 x 6 block per lane, 16 lanes per (block, chunk)
 * pair, recursive-halving reduction): a row-per-lane mapping with no reduction was built and LOST -- broadcast reads
 * do not shorten an LDS instruction, so 6 lanes reading 30 doubles each per contribution cost 5x the LDS cycles of
 * one lane reading 45.
 * `tid` is the thread's index inside its part, `nloop` the (uniform) number of chunk steps every part of the
 * workgroup goes through -- k_ba_ph_finish runs both parts inside one 512-thread workgroup and they share its barriers. */
#define BA_SH_T 256
__device__ __forceinline__ void ba_schur_part(const BAWin &w, double *D, const int32_t *I, double *tile, int tid, int part, int nparts,
                                              int nloop, int skip) {
#pragma clang fp contract(fast)
    const int ch0 = (int)((int64_t)w.nchunks * part / nparts), ch1 = (int)((int64_t)w.nchunks * (part + 1) / nparts);
    const int64_t So = part == 0 ? w.S : w.S2, go = part == 0 ? w.g : w.g2;
    const double lambda = D[w.st + ST_LAMBDA];
    double *sW = tile, *sD = sW + BA_CHUNK_E * 18, *sT = sD + BA_CHUNK_P * 9;
    int32_t *sP = (int32_t *)(sT + BA_CHUNK_P * 3);
    int32_t *sC = sP + BA_CHUNK_E;
    int32_t *sB = sC + BA_CHUNK_C;
    int32_t *sBS = sB + BA_CHUNK_B;
    int32_t *sG = sBS + BA_CHUNK_B + 8;
    const int grp = tid >> 4, l16 = tid & 15; /* g: 16 lanes per camera */
    const int grpS = tid / BA_SG, lS = tid % BA_SG; /* S: BA_SG lanes per block */
    /* A chunk's loads form a chain of three round trips -- its descriptor, the camera / point index of each edge, the
       camera rotation and the point's factor gathered through those -- and the chain, not the bytes, is what the load
       phase costs (5.8 of the 15.7 us a chunk takes).  The descriptor is therefore fetched TWO chunks ahead (uniform:
       scalar registers) and the edge indices ONE chunk ahead (six registers per thread), both landing during the block
       products, so that everything a chunk needs goes out in one round trip. */
    constexpr int EPT = (BA_CHUNK_E + BA_SH_T - 1) / BA_SH_T;
    struct ChDesc { int e0, ne, p0, npnt, cb0, ncb, cg0, ncg, c0, nc, g0, ng; };
    auto load_desc = [&](int it) {
        const int ch = ch0 + it;
        const bool live = ch < ch1;
        const int32_t *cd = I + w.ch_desc + (live ? ch : ch0) * BA_CH_DESC;
        ChDesc d;
        d.e0 = cd[0], d.ne = live ? cd[1] - cd[0] : 0, d.p0 = cd[2], d.npnt = live ? cd[3] - cd[2] : 0, d.cb0 = cd[4], d.ncb = live ? cd[5] - cd[4] : 0;
        d.cg0 = cd[6], d.ncg = live ? cd[7] - cd[6] : 0, d.c0 = cd[8], d.nc = live ? cd[9] : 0, d.g0 = cd[10], d.ng = live ? cd[11] : 0;
        return d;
    };
    auto load_idx = [&](const ChDesc &d, int32_t *cam, int32_t *pt, int32_t *cg) {
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const int el = tid + k * BA_SH_T;
            cam[k] = I[w.cam + d.e0 + (el < d.ne ? el : 0)];
            pt[k] = el < d.ne ? I[w.pt + d.e0 + el] : d.p0; /* the point's index in the WINDOW: arithmetic on a guarded load
                                                               makes the compiler wait for it inside the guard -- a
                                                               round trip in the middle of the load sequence */
            cg[k] = el < d.ng ? I[w.cg_edge + d.g0 + el] : 0;
        }
    };
    ChDesc dcur = load_desc(0), dnext = load_desc(1);
    int32_t rCam[EPT], rP[EPT], rG[EPT];
    load_idx(dcur, rCam, rP, rG);
    for (int it = 0; it < nloop; it++) {
        const int e0 = dcur.e0, ne = dcur.ne, p0 = dcur.p0, npnt = dcur.npnt, cb0 = dcur.cb0, ncb = dcur.ncb;
        const int cg0 = dcur.cg0, ncg = dcur.ncg, c0 = dcur.c0, nc = dcur.nc, g0 = dcur.g0;
        const ChDesc dafter = load_desc(it + 2);
        int32_t nCam[EPT], nP[EPT], nG[EPT];
        { /* every load of the chunk issued before the first LDS write */
            constexpr int TPT = (BA_CHUNK_P * 3 + BA_SH_T - 1) / BA_SH_T, CPT = (BA_CHUNK_C + BA_SH_T - 1) / BA_SH_T,
                          BPT = (BA_CHUNK_B + 1 + BA_SH_T - 1) / BA_SH_T;
            double2 rR[EPT][2];
            double rT[TPT];
            int32_t rC[CPT], rB[BPT], rBS[BPT];
#pragma unroll
            for (int k = 0; k < EPT; k++) {
                const int el = tid + k * BA_SH_T;
                const double2 *rc = (const double2 *)(D + w.W + (int64_t)(e0 + (el < ne ? el : 0)) * 4);
                rR[k][0] = rc[0];
                rR[k][1] = rc[1];
            }
#pragma unroll
            for (int k = 0; k < TPT; k++) rT[k] = tid + k * BA_SH_T < npnt * 3 ? D[w.tu + (int64_t)p0 * 3 + tid + k * BA_SH_T] : 0.0;
#pragma unroll
            for (int k = 0; k < CPT; k++) rC[k] = tid + k * BA_SH_T < nc ? I[w.cb_con + c0 + tid + k * BA_SH_T] : 0;
#pragma unroll
            for (int k = 0; k < BPT; k++) {
                rB[k] = tid + k * BA_SH_T < ncb ? I[w.cb_blk + cb0 + tid + k * BA_SH_T] : 0;
                rBS[k] = tid + k * BA_SH_T <= ncb ? I[w.cb_start + cb0 + tid + k * BA_SH_T] : c0; /* (- c0 at the LDS write) */
            }
            double rRot[EPT][9], rU[EPT][6];
#pragma unroll
            for (int k = 0; k < EPT; k++) {
#pragma unroll
                for (int i = 0; i < 9; i++) rRot[k][i] = D[w.Rc + (int64_t)rCam[k] * 9 + i];
#pragma unroll
                for (int i = 0; i < 6; i++) rU[k][i] = D[w.Uf + (int64_t)rP[k] * 6 + i];
            }
            load_idx(dnext, nCam, nP, nG); /* the next chunk's indices, behind this chunk's loads */
#pragma unroll
            for (int k = 0; k < EPT; k++) { /* the W tile: one edge per thread, rebuilt from its record */
                const int el = tid + k * BA_SH_T;
                if (el < BA_CHUNK_E) {
                    const double rec[4] = {rR[k][0].x, rR[k][0].y, rR[k][1].x, el < ne ? rR[k][1].y : 0.0};
                    double Wr[18];
                    ba_w_from_record(rec, rRot[k], w.fx, w.fy, Wr);
                    double Z[18]; /* Z = W U */
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        const double w0 = Wr[r * 3], w1 = Wr[r * 3 + 1], w2 = Wr[r * 3 + 2];
                        Z[r * 3] = w0 * rU[k][0];
                        Z[r * 3 + 1] = w0 * rU[k][1] + w1 * rU[k][3];
                        Z[r * 3 + 2] = w0 * rU[k][2] + w1 * rU[k][4] + w2 * rU[k][5];
                    }
#pragma unroll
                    for (int i = 0; i < 9; i++) ((double2 *)(sW + el * 18))[i] = double2{Z[2 * i], Z[2 * i + 1]};
                    sP[el] = rP[k] - p0;
                    sG[el] = rG[k];
                }
            }
#pragma unroll
            for (int k = 0; k < TPT; k++)
                if (tid + k * BA_SH_T < BA_CHUNK_P * 3) sT[tid + k * BA_SH_T] = rT[k];
#pragma unroll
            for (int k = 0; k < CPT; k++)
                if (tid + k * BA_SH_T < BA_CHUNK_C) sC[tid + k * BA_SH_T] = rC[k];
#pragma unroll
            for (int k = 0; k < BPT; k++) {
                if (tid + k * BA_SH_T < BA_CHUNK_B) sB[tid + k * BA_SH_T] = rB[k];
                if (tid + k * BA_SH_T <= BA_CHUNK_B) sBS[tid + k * BA_SH_T] = rBS[k] - c0;
            }
        }
        __syncthreads();
        /* blocks: group g (BA_SG lanes) takes blocks g, g + 256 / BA_SG, ... of the chunk's list (sorted heaviest first on
           the host); the block's current value is fetched one pair ahead */
        /* current value of this lane's BA_SG_OUT entries of a block: from memory, or -- FIRST touch by this part -- what
           the block starts at: Hpp + lambda I for a diagonal block of part 0, zero otherwise */
        const int e0S = (lS * BA_SG_OUT < 36 ? lS : 0) * BA_SG_OUT;
        auto block_value = [&](int raw, double *&ptr, double *v) {
            const int bk = raw & BA_BLK_MASK;
            ptr = D + So + (int64_t)bk * 36 + e0S;
            if (!(raw & BA_BLK_FIRST)) {
#pragma unroll
                for (int i = 0; i < BA_SG_OUT; i++) v[i] = ptr[i];
            } else if (part == 0 && (raw & BA_BLK_DIAG)) {
                const double *Hd = D + w.Hpp + (int64_t)I[w.blk_rowid + bk] * 36 + e0S;
#pragma unroll
                for (int i = 0; i < BA_SG_OUT; i++) v[i] = Hd[i] + ((e0S + i) % 7 == 0 ? lambda : 0.0);
            } else {
#pragma unroll
                for (int i = 0; i < BA_SG_OUT; i++) v[i] = 0.0;
            }
        };
        int u = (skip & 16) ? ncb : grpS;
        double *Sb = nullptr;
        double sb[BA_SG_OUT];
#pragma unroll
        for (int i = 0; i < BA_SG_OUT; i++) sb[i] = 0.0;
        if (u < ncb) block_value(sB[u], Sb, sb);
        for (; u < ncb; u += BA_SH_T / BA_SG) {
            const int k0 = sBS[u], k1 = sBS[u + 1];
            const int un = u + BA_SH_T / BA_SG;
            double *Sn = Sb;
            double sn[BA_SG_OUT];
#pragma unroll
            for (int i = 0; i < BA_SG_OUT; i++) sn[i] = 0.0;
            if (un < ncb) block_value(sB[un], Sn, sn);
            double acc[36];
#pragma unroll
            for (int i = 0; i < 36; i++) acc[i] = 0.0;
            for (int k = (skip & 256) ? k1 : k0 + lS; k < k1; k += BA_SG) {
                const uint32_t ab = (uint32_t)sC[k];
                const double *Za = sW + (ab & 511) * 18, *Zb = sW + ((ab >> 9) & 511) * 18;
                double y[18], ww[18];
#pragma unroll
                for (int i = 0; i < 18; i++) y[i] = Za[i];
#pragma unroll
                for (int i = 0; i < 18; i++) ww[i] = Zb[i];
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c < 6; c++)
                        acc[r * 6 + c] += y[r * 3] * ww[c * 3] + y[r * 3 + 1] * ww[c * 3 + 1] + y[r * 3 + 2] * ww[c * 3 + 2];
            }
            double d[BA_SG_OUT];
            group_reduce36(acc, lS, d); /* lane q now holds the totals of entries q * BA_SG_OUT .. */
            if (lS * BA_SG_OUT < 36) {
#pragma unroll
                for (int i = 0; i < BA_SG_OUT; i++) Sb[i] = sb[i] - d[i];
            }
            Sb = Sn;
#pragma unroll
            for (int i = 0; i < BA_SG_OUT; i++) sb[i] = sn[i];
        }
        /* g: group g takes cameras g, g + 16, ... of the chunk (16 lanes per camera; 4-lane groups -- fewer sequential
           read-modify-writes of g per group -- measured slower: 13.4 against 11.7 ms) */
        for (int u2 = (skip & 64) ? ncg : grp; u2 < ncg; u2 += BA_SH_T / 16) {
            const int sl = I[w.cg_slot + cg0 + u2];
            const int k0 = I[w.cg_start + cg0 + u2] - g0, k1 = I[w.cg_start + cg0 + u2 + 1] - g0;
            double acc[6] = {0, 0, 0, 0, 0, 0};
            for (int k = k0 + l16; k < k1; k += 16) {
                const int a = sG[k];
                const double *Wa = sW + a * 18, *t = sT + sP[a] * 3;
                const double t0 = t[0], t1 = t[1], t2 = t[2];
#pragma unroll
                for (int r = 0; r < 6; r++) acc[r] += Wa[r * 3] * t0 + Wa[r * 3 + 1] * t1 + Wa[r * 3 + 2] * t2;
            }
#pragma unroll
            for (int r = 0; r < 6; r++) {
                double v = acc[r];
                v += xchg16<8>(v);
                v += xchg16<4>(v);
                v += xchg16<2>(v);
                v += xchg16<1>(v);
                acc[r] = v;
            }
            if (l16 == 0) {
#pragma unroll
                for (int r = 0; r < 6; r++) D[go + sl * 6 + r] -= acc[r];
            }
        }
        __syncthreads(); /* the next chunk overwrites the LDS tiles and may touch the same blocks */
        dcur = dnext;
        dnext = dafter;
#pragma unroll
        for (int k = 0; k < EPT; k++) rCam[k] = nCam[k], rP[k] = nP[k], rG[k] = nG[k];
    }
}
__global__ __launch_bounds__(BA_SH_T) __attribute__((amdgpu_waves_per_eu(2))) void k_ba_ph_schur2(BABufs bufs, int skip) {
    extern __shared__ __attribute__((aligned(16))) double ba_dyn[];
    BA_PH_WINDOW_COPY
    if (ph == 2.0 || w.nchunks <= 0) return;
    const int part = blockIdx.y, nparts = gridDim.y;
    const int nloop = (int)((int64_t)w.nchunks * (part + 1) / nparts) - (int)((int64_t)w.nchunks * part / nparts);
    ba_schur_part(w, D, I, ba_dyn, threadIdx.x, part, nparts, nloop, skip);
}
/* the Schur pass of a window whose points have LONG tracks (ba_window_chunkable): one owner per block walking the block's
   contribution list, and per camera for g -- the team path's functions, strided over the workgroups of the window */
__global__ __launch_bounds__(BA_T) void k_ba_ph_schur_lists(BABufs bufs) {
    BA_PH_WINDOW
    if (ph == 2.0 || w.nchunks > 0) return;
    const int gw = blockIdx.y * BA_W + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), gws = gridDim.y * BA_W;
    ba_schur_lists(w, D, I, gw, gws);
    ba_g_lists(w, D, I, gw, gws);
}
__global__ __launch_bounds__(BA_T) void k_ba_ph_chol(BABufs bufs) {
    extern __shared__ __attribute__((aligned(16))) double ba_dyn[];
    __shared__ double bad_flag[1];
    BA_PH_WINDOW
    if (ph == 2.0) return;
    ba_ph_chol(w, D, I, ba_dyn, bad_flag);
}
/* the band Cholesky (ba_chol_band) as a phase kernel: NT threads per window, bands up to MAXM tile rows */
template <int NT, int MAXM>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void k_ba_ph_chol_band(BABufs bufs, int skip) {
    extern __shared__ __attribute__((aligned(16))) double ba_dyn[];
    __shared__ double bad_flag[1];
    BA_PH_WINDOW_COPY
    if (ph == 2.0) return;
    const bool ok2 = ba_chol_band<NT, true, MAXM>(w, D, I, w.Ncf * 6, ba_dyn, bad_flag, skip);
    if (threadIdx.x == 0) {
        D[w.st + ST_OK2] = ok2 ? 1.0 : 0.0;
        D[w.st + ST_SOLVES] += 1.0;
    }
}
__global__ __launch_bounds__(BA_PH_T) void k_ba_ph_backsub(BABufs bufs) {
    __shared__ BARed s;
    BA_PH_WINDOW_COPY
    const int parts = ba_ph_parts_p(w.Np);
    if (ph == 2.0 || (int)blockIdx.y >= parts) return;
    const bool ok2 = D[w.st + ST_OK2] != 0.0;
    const double lam = D[w.st + ST_LAMBDA];
    const double part_sum = w.nchunks > 0 ? ba_phase_backsub<true>(w, D, I, ok2, lam, blockIdx.y * BA_PH_T + threadIdx.x, parts * BA_PH_T)
                                          : ba_phase_backsub<false>(w, D, I, ok2, lam, blockIdx.y * BA_PH_T + threadIdx.x, parts * BA_PH_T);
    const double v = block_sum_n(part_sum, s, BA_PH_T / 64);
    if (threadIdx.x == 0) D[w.gred + PH_RED_SCALE + blockIdx.y] = v;
}
__global__ __launch_bounds__(BA_PH_T) void k_ba_ph_trial(BABufs bufs) {
    __shared__ BARed s;
    BA_PH_WINDOW_COPY
    const int parts = ba_ph_parts_e(w.E);
    if (ph == 2.0 || (int)blockIdx.y >= parts) return;
    const double v = block_sum_n(ba_ph_chi2_part<false>(w, D, I, B, w.poses_t, w.pts_t, w.err_t, blockIdx.y * BA_PH_T + threadIdx.x, parts * BA_PH_T), s, BA_PH_T / 64);
    if (threadIdx.x == 0) D[w.gred + PH_RED_TRIAL + blockIdx.y] = v;
}
/* gain ratio and g2o's accept / reject bookkeeping: one thread per window */
__global__ void k_ba_ph_decide(BABufs bufs, int nwin, int trial_seq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwin) return;
    const BAWin &w = bufs.wins[i];
    double *D = bufs.D, *st = D + w.st;
    if (st[ST_PH] == 2.0) return;
    double lambda = st[ST_LAMBDA], ni = st[ST_NI];
    const double currentChi = st[ST_CHI];
    const bool ok2 = st[ST_OK2] != 0.0;
    int qmax = (int)st[ST_QMAX], budget = (int)st[ST_BUDGET];
    double tempChi = 0, scale = 0;
    for (int g = 0; g < ba_ph_parts_e(w.E); g++) tempChi += D[w.gred + PH_RED_TRIAL + g];
    for (int g = 0; g < ba_ph_parts_p(w.Np); g++) scale += D[w.gred + PH_RED_SCALE + g];
    if (!ok2) tempChi = 1.7976931348623157e308;
    if (!ok2) scale = 0;
    scale += 1e-3;
    const double rho = (currentChi - tempChi) / scale;
    const bool accept = rho > 0 && isfinite(tempChi);
    if (accept) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, 2. / 3.);
        lambda *= fmax(1. / 3., alpha);
        ni = 2;
        st[ST_CHI] = tempChi;
        st[ST_HAVECHI] = 1.0;
        st[ST_ACCEPT] = (double)(trial_seq + 1); /* k_ba_ph_accept of this trial copies the trial state */
    } else {
        lambda *= ni;
        ni *= 2;
    }
    qmax++;
    st[ST_LAMBDA] = lambda;
    st[ST_NI] = ni;
    st[ST_QMAX] = (double)qmax;
    if (rho < 0 && qmax < 10) {
        st[ST_PH] = 1.0; /* same iteration, larger lambda */
    } else {
        st[ST_ITERS] += 1;
        if (qmax == 10 || rho == 0) { /* OptimizationAlgorithmLevenberg returns Terminate */
            st[ST_TERM] = 1.0;
            st[ST_PH] = 2.0;
        } else {
            budget--;
            st[ST_BUDGET] = (double)budget;
            st[ST_PH] = budget > 0 ? 0.0 : 2.0;
        }
    }
}
__global__ __launch_bounds__(BA_PH_T) void k_ba_ph_accept(BABufs bufs, int trial_seq) {
    const BAWin &w = bufs.wins[blockIdx.x];
    double *D = bufs.D;
    if (D[w.st + ST_ACCEPT] != (double)(trial_seq + 1)) return;
    ba_phase_accept<false>(w, D, blockIdx.y * BA_PH_T + threadIdx.x, gridDim.y * BA_PH_T);
}
/* whatever is left of the round, in one workgroup per window (normally nothing: every window is at ST_PH == 2) */
__global__ __launch_bounds__(BA_T) void k_ba_ph_finish(BABufs bufs, int band) {
    extern __shared__ __attribute__((aligned(16))) double ba_dyn[];
    __shared__ BARed s;
    const BAWin &w = bufs.wins[blockIdx.x];
    double *D = bufs.D;
    const int32_t *I = bufs.I;
    const uint8_t *B = bufs.B;
    /* EXIT CONDITION: every pass either spends one of the <= 10 trials of an iteration or one iteration of the budget;
       the pass counter bounds the loop whatever the state record holds */
    for (int pass = 0; pass < 4096; pass++) {
        const double ph = D[w.st + ST_PH];
        if (!(ph == 0.0 || ph == 1.0)) break;
        if (ph == 0.0) {
            if (D[w.st + ST_HAVECHI] == 0.0) ba_ph_chi2(w, D, I, B, s);
            __syncthreads();
            ba_phase_hll<true>(w, D, I, B, threadIdx.x, BA_T);
            if (w.nchunks > 0)
                ba_phase_hpp<true, true>(w, D, I, B, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), BA_W, threadIdx.x & 63);
            else
                ba_phase_hpp<false, true>(w, D, I, B, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), BA_W, threadIdx.x & 63);
            __syncthreads();
        }
        ba_ph_dinv(w, D, I, s);
        ba_phase_zero2(w, D, threadIdx.x, BA_T);
        __syncthreads();
        if (w.nchunks > 0) { /* both parts of the Schur pass inside this workgroup: threads 0..255 part 0, 256..511 part 1, own LDS tiles */
            const int half = threadIdx.x / BA_SH_T;
            ba_schur_part(w, D, I, ba_dyn + half * BA_CHUNK_DOUBLES, threadIdx.x - half * BA_SH_T, half, 2, (w.nchunks + 1) / 2, 0);
        } else { /* long tracks: per-block lists */
            const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            ba_schur_lists(w, D, I, wv, BA_W);
            ba_g_lists(w, D, I, wv, BA_W);
        }
        __syncthreads();
        ba_ph_chol(w, D, I, ba_dyn, &s.bc[0], band != 0);
        __syncthreads();
        ba_ph_update(w, D, I, B, s);
        __syncthreads();
    }
}

/* between rounds: re-evaluate every edge at the accepted estimate, switch off edges with chi2 above
 * the threshold or non-positive depth, optionally drop the robust kernel (LocalBundleAdjustment) */
__global__ __launch_bounds__(BA_T) void k_ba_classify(BABufs bufs, int drop_robust, int final_only) {
    __shared__ BAShared s;
    const BAWin w = bufs.wins[blockIdx.x];
    double *D = bufs.D;
    const int32_t *I = bufs.I;
    uint8_t *B = bufs.B;
    double acc = 0;
    for (int e = threadIdx.x; e < w.E; e += BA_T) {
        uint8_t f = B[w.flags + e];
        double er[2], pc[3];
        project_err(D + w.poses + (int64_t)I[w.cam + e] * 7, D + w.pts + (int64_t)I[w.pt + e] * 3, D + w.obs + (int64_t)e * 2, w.fx,
                    w.fy, w.cx, w.cy, er, pc);
        D[w.err + (int64_t)e * 2] = er[0];
        D[w.err + (int64_t)e * 2 + 1] = er[1];
        const double c = (er[0] * er[0] + er[1] * er[1]) * D[w.info + e];
        if (!final_only) {
            if (c > w.chi2_thr || !(pc[2] > 0.0)) f &= ~1;
            if (drop_robust) f &= ~2;
            B[w.flags + e] = f;
        }
        if (f & 1) acc += (f & 2) ? huber_rho0(c, w.delta) : c;
    }
    acc = block_sum(acc, s);
    if (threadIdx.x == 0) {
        D[w.st + ST_CHI] = acc;
        if (!final_only) D[w.st + ST_TERM] = 0.0; /* a new optimize() call starts */
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* host staging of one window before concatenation; kept in the solver so that a steady stream of similar batches
 * re-uses the vectors' capacity (allocating and freeing ~2 MB per window per call cost more than the staging) */
struct BAPiece {
    std::vector<double> O, D; /* outputs (poses, points, LM state) and inputs (observations, information) */
    std::vector<int32_t> I;
    std::vector<uint8_t> B;
    size_t scratch = 0;
    int rc = 0;
};

struct HostWin {
    BAWin w;
    std::vector<int> perm; /* sorted edge -> caller's edge index */
    std::vector<int> rank; /* caller's point -> position in the device layout */
};
/* Layout of a batch in the double buffer: [ outputs of every window | inputs of every window | scratch ].  The
 * output region (poses, points, LM state) is what comes back, as ONE download; inputs only go up; scratch never
 * crosses PCIe.  Until ba_stage() concatenates the windows, offsets are relative to the window's own piece. */
static int64_t BAWin::*const kOutDFields[] = {&BAWin::poses, &BAWin::pts, &BAWin::st};
static int64_t BAWin::*const kInDFields[] = {&BAWin::obs, &BAWin::info};
static int64_t BAWin::*const kIntFields[] = {&BAWin::cam,     &BAWin::pt,      &BAWin::pt_start, &BAWin::cam_start,
                                             &BAWin::cam_edges, &BAWin::slot,  &BAWin::blk_row,  &BAWin::blk_col,
                                             &BAWin::blk_diag, &BAWin::blk_mirror, &BAWin::blk_upper, &BAWin::blk_rowid, &BAWin::con_start, &BAWin::con_e1, &BAWin::con_e2, &BAWin::ch_first, &BAWin::ch_last,
                                             &BAWin::ch_desc, &BAWin::cb_blk, &BAWin::cb_start, &BAWin::cb_con, &BAWin::cg_slot, &BAWin::cg_start, &BAWin::cg_edge, &BAWin::gbar, &BAWin::blk_touch};
static int64_t BAWin::*const kScratchFields[] = {&BAWin::poses_t, &BAWin::pts_t, &BAWin::err, &BAWin::err_t, &BAWin::W,
                                                 &BAWin::tv,      &BAWin::Hpp,   &BAWin::bp,  &BAWin::Hll,   &BAWin::bl,
                                                 &BAWin::Dinv,    &BAWin::S,     &BAWin::g,   &BAWin::Minv,  &BAWin::x,
                                                 &BAWin::xl,      &BAWin::A,     &BAWin::Linv, &BAWin::gred,  &BAWin::pcgv,
                                                 &BAWin::S2,      &BAWin::g2,    &BAWin::Rc,   &BAWin::Uf,    &BAWin::tu};
#define BA_CHOL_LDS_BYTES (128 * 1024) /* dynamic LDS the blocked Cholesky may take beside the static 25 KB */

/* Point chunks or per-block lists for a window's Schur complement?  The chunked pass streams a window's edges once and
 * wins while the points of a chunk pile several contributions onto a block.  In a map with LONG tracks every point
 * touches k (k + 1) / 2 blocks: at 23 observing keyframes a single point no longer fits a chunk (BA_CHUNK_B blocks), and
 * cutting its pair list over several chunks loses badly (one contribution per (block, chunk) pair: 615 ms against 71 for
 * 32 windows of 40 keyframes x 1500 points with tracks of 30) -- there the per-block lists win, whose owners walk ~1000
 * contributions each.  Measured on that shape, kernel ms per 32-window batch, teams + lists / phase kernels + chunks:
 * tracks of 8: 25.4 / 8.0, 12: 35.4 / 10.4, 15: 40.9 / 15.6, 18: 48.3 / 22.6, 22: 50.1 / 38.5, 26 (with 480 blocks per
 * chunk): 70.2 / 69.4, 30: 71.2 / 122.5.  With the lists as a phase kernel too (k_ba_ph_schur_lists), same shape, phase
 * kernels + lists / phase kernels + chunks: tracks of 6: 6.7 / 7.3, 12: 10.2 / 10.5, 18: 16.6 / 22.8, 22: 23.0 / 38.9, 30:
 * 32.5 / - : with 32 such windows W (83 .. 150 MB) stays in the Infinity Cache and the lists' re-reads are cheap; at 256
 * windows of 100 x 5000 (1 GB of W) they are HBM traffic and the chunks win 41 to 94 ms.  So: chunks when no point is seen
 * by more than BA_CHUNK_MAX_TRACK free keyframes and the mean track length is at most BA_CHUNK_MEAN_TRACK_CACHED (the
 * batch's W fits the cache) or BA_CHUNK_MEAN_TRACK (it does not); lists otherwise. */
#define BA_CHUNK_MAX_TRACK 22
#define BA_CHUNK_MEAN_TRACK 22       /* ... when the batch's W does not fit the Infinity Cache */
#define BA_CHUNK_MEAN_TRACK_CACHED 12 /* ... when it does: lists are as good at short tracks and better beyond (below) */
#define BA_CACHED_W_BYTES (192u << 20)
static bool ba_window_chunkable(const BAProblem *p, int mean_track_limit) {
    static const int forced = getenv("ORBSLAM_AMD_BA_MEAN_TRACK") ? atoi(getenv("ORBSLAM_AMD_BA_MEAN_TRACK")) : 0; /* (diagnostics) */
    const int mean_track = forced > 0 ? forced : mean_track_limit;
    if (p->n_points <= 0 || (long)p->n_obs > (long)mean_track * p->n_points) return false;
    std::vector<int> deg(p->n_points, 0);
    for (int e = 0; e < p->n_obs; e++) {
        const int c = p->cam_idx[e], q = p->pt_idx[e];
        if (c < 0 || c >= p->n_cams || q < 0 || q >= p->n_points) return false; /* (stage_problem reports it) */
        if (!p->fixed[c] && ++deg[q] > BA_CHUNK_MAX_TRACK) return false;
    }
    return true;
}

/* Lay one problem out in the host staging vectors; returns an error code.  `team`: the window will be spread over
 * several workgroups (or is too irregular for point chunks) -> per-block contribution lists; otherwise point
 * chunks for the LDS-staged Schur pass of the batch path. */
static int stage_problem(const BAProblem *p, std::vector<double> &hO, std::vector<double> &hD, size_t &scratchD,
                         std::vector<int32_t> &hI, std::vector<uint8_t> &hB, HostWin &hw, bool team,
                         int mean_track_limit = BA_CHUNK_MEAN_TRACK) {
    if (!p || p->n_cams <= 0 || p->n_points < 0 || p->n_obs < 0 || !p->poses || !p->fixed ||
        (p->n_points && !p->points) || (p->n_obs && (!p->obs || !p->cam_idx || !p->pt_idx || !p->inv_sigma2)) ||
        p->n_rounds < 1 || p->n_rounds > 4)
        return ORB_E_INVALID;
    const int Nc = p->n_cams, Np = p->n_points, E = p->n_obs;
    BAWin &w = hw.w;
    memset(&w, 0, sizeof(w));
    w.Nc = Nc;
    w.Np = Np;
    w.E = E;
    w.fx = p->fx; w.fy = p->fy; w.cx = p->cx; w.cy = p->cy;
    w.delta = p->huber_delta;
    w.chi2_thr = p->chi2_threshold;
    for (int e = 0; e < E; e++)
        if (p->pt_idx[e] < 0 || p->pt_idx[e] >= Np || p->cam_idx[e] < 0 || p->cam_idx[e] >= Nc) return ORB_E_INVALID;
    /* free camera slots */
    std::vector<int> slot(Nc, -1), slot2cam;
    for (int c = 0; c < Nc; c++)
        if (!p->fixed[c]) {
            slot[c] = (int)slot2cam.size();
            slot2cam.push_back(c);
        }
    const int Ncf = (int)slot2cam.size();
    w.Ncf = Ncf;
    /* device order of the points: by the first free camera that sees them (then by id).  Consecutive points then
       contribute to a compact band of the reduced camera system -- local windows are chain-like, a point is seen by
       neighbouring keyframes -- which is what lets a chunk of points work on a handful of blocks from LDS. */
    std::vector<int> first_slot(Np, Ncf);
    for (int e = 0; e < E; e++) {
        const int s_ = slot[p->cam_idx[e]];
        if (s_ >= 0 && s_ < first_slot[p->pt_idx[e]]) first_slot[p->pt_idx[e]] = s_;
    }
    std::vector<int> order(Np), &rank = hw.rank;
    {
        std::vector<int> head(Ncf + 2, 0); /* counting sort: stable, O(Np) */
        for (int i = 0; i < Np; i++) head[first_slot[i] + 1]++;
        for (int i = 0; i <= Ncf; i++) head[i + 1] += head[i];
        for (int i = 0; i < Np; i++) order[head[first_slot[i]]++] = i;
    }
    rank.assign(Np, 0);
    for (int r = 0; r < Np; r++) rank[order[r]] = r;
    /* edges sorted by point position (stable) */
    std::vector<int> cnt(Np + 1, 0);
    for (int e = 0; e < E; e++) cnt[rank[p->pt_idx[e]] + 1]++;
    for (int i = 0; i < Np; i++) cnt[i + 1] += cnt[i];
    std::vector<int> pos(cnt.begin(), cnt.end() - 1);
    hw.perm.assign(E, 0);
    for (int e = 0; e < E; e++) hw.perm[pos[rank[p->pt_idx[e]]]++] = e;
    /* camera edge lists (sorted-edge indices) */
    std::vector<int> cstart(Nc + 1, 0), cedges(E);
    for (int k = 0; k < E; k++) cstart[p->cam_idx[hw.perm[k]] + 1]++;
    for (int c = 0; c < Nc; c++) cstart[c + 1] += cstart[c];
    {
        std::vector<int> cp(cstart.begin(), cstart.end() - 1);
        for (int k = 0; k < E; k++) cedges[cp[p->cam_idx[hw.perm[k]]]++] = k;
    }
    /* reduced-system block structure.  A dense Ncf x Ncf table maps a camera-slot pair to its block id (counts
       first, ids after a row-major sweep), so the O(sum k^2) passes over the co-observation pairs are plain table
       look-ups: no per-row vectors, no sorting, no binary searches. */
    std::vector<int> eslot(E); /* camera slot of every (sorted) edge, -1 = fixed camera */
    for (int k = 0; k < E; k++) eslot[k] = slot[p->cam_idx[hw.perm[k]]];
    std::vector<int> blk((size_t)Ncf * Ncf, 0); /* pass 1: contributions per pair; pass 2: block id */
    int max_deg = 0;
    for (int pt = 0; pt < Np; pt++) {
        max_deg = std::max(max_deg, cnt[pt + 1] - cnt[pt]);
        for (int a = cnt[pt]; a < cnt[pt + 1]; a++) {
            const int s1 = eslot[a];
            if (s1 < 0) continue;
            int *row = &blk[(size_t)s1 * Ncf];
            for (int b = cnt[pt]; b < cnt[pt + 1]; b++) {
                const int s2 = eslot[b];
                if (s2 >= 0) row[s2]++;
            }
        }
    }
    std::vector<int> brow(Ncf + 1, 0), bcol, bdiag(Ncf, 0), ccount(1, 0);
    for (int s1 = 0; s1 < Ncf; s1++) {
        int *row = &blk[(size_t)s1 * Ncf];
        for (int s2 = 0; s2 < Ncf; s2++) {
            const int nc = row[s2];
            if (nc > 0 || s2 == s1) { /* the diagonal block always exists (Hpp + lambda I) */
                if (s2 == s1) bdiag[s1] = (int)bcol.size();
                row[s2] = (int)bcol.size();
                bcol.push_back(s2);
                ccount.push_back(ccount.back() + (s2 >= s1 ? nc : 0)); /* only blocks on or above the diagonal are accumulated */
            } else
                row[s2] = -1;
        }
        brow[s1 + 1] = (int)bcol.size();
    }
    const int nnzb = (int)bcol.size();
    w.nnzb = nnzb;
    std::vector<int> bmirror(nnzb); /* id of the transposed block (S is symmetric) */
    for (int s1 = 0; s1 < Ncf; s1++)
        for (int bk = brow[s1]; bk < brow[s1 + 1]; bk++) bmirror[bk] = blk[(size_t)bcol[bk] * Ncf + s1];
    const int ncon = ccount[nnzb];
    const bool chunked = !team && max_deg <= BA_CHUNK_E && Ncf > 0 && ba_window_chunkable(p, mean_track_limit);
    /* ---- list path: per block, the edge pairs (a, b) of the points both cameras see ---- */
    std::vector<int> ce1, ce2;
    if (!chunked) {
        ce1.resize(ncon);
        ce2.resize(ncon);
        std::vector<int> cp(ccount.begin(), ccount.end() - 1);
        for (int pt = 0; pt < Np; pt++)
            for (int a = cnt[pt]; a < cnt[pt + 1]; a++) {
                const int s1 = eslot[a];
                if (s1 < 0) continue;
                const int *row = &blk[(size_t)s1 * Ncf];
                for (int b = cnt[pt]; b < cnt[pt + 1]; b++) {
                    const int s2 = eslot[b];
                    if (s2 < s1) continue; /* also skips fixed cameras (-1) */
                    const int k = cp[row[s2]]++;
                    ce1[k] = a;
                    ce2[k] = b;
                }
            }
    }
    /* ---- batch path: point chunks ---- */
    std::vector<int> chd, cbb, cbs(1, 0), cbc, cgs, cgst(1, 0), cge;
    if (chunked) {
        cbc.reserve(ncon);
        cge.reserve(E);
        std::vector<int> seen_blk(nnzb, -1), seen_slot(Ncf, -1); /* chunk id that last touched it -> index in the chunk's list */
        std::vector<int> blk_pos(nnzb, 0), slot_pos(Ncf, 0);
        std::vector<int> cb_cnt, cg_cnt, tmp_blk, tmp_slot;
        bool chunk_overflow = false;
        std::vector<int> grow_blk(nnzb, -1); /* growth attempt that last counted the block */
        int grow_stamp = 0;
        int p0 = 0;
        while (p0 < Np) {
            int p1 = p0;
            long first_pairs = 0;
            {
                /* as many consecutive points as the LDS tiles hold: edges, points, contributions (one per pair of a point's
                   free cameras) and distinct blocks touched.  The first point is taken whatever it needs (below). */
                long ncon_c = 0, nblk_c = 0;
                grow_stamp++;
                while (p1 < Np && p1 - p0 < BA_CHUNK_P && cnt[p1 + 1] - cnt[p0] <= BA_CHUNK_E) {
                    long pairs = 0, newb = 0;
                    for (int a = cnt[p1]; a < cnt[p1 + 1]; a++) {
                        const int s1 = eslot[a];
                        if (s1 < 0) continue;
                        const int *row = &blk[(size_t)s1 * Ncf];
                        for (int b = cnt[p1]; b < cnt[p1 + 1]; b++) {
                            const int s2 = eslot[b];
                            if (s2 < s1) continue;
                            pairs++;
                            if (grow_blk[row[s2]] != grow_stamp) newb++; /* (counted again below if the point is taken) */
                        }
                    }
                    if (p1 > p0 && (ncon_c + pairs > BA_CHUNK_C || nblk_c + newb > BA_CHUNK_B)) break;
                    for (int a = cnt[p1]; a < cnt[p1 + 1]; a++) {
                        const int s1 = eslot[a];
                        if (s1 < 0) continue;
                        const int *row = &blk[(size_t)s1 * Ncf];
                        for (int b = cnt[p1]; b < cnt[p1 + 1]; b++)
                            if (eslot[b] >= s1 && grow_blk[row[eslot[b]]] != grow_stamp) {
                                grow_blk[row[eslot[b]]] = grow_stamp;
                                nblk_c++;
                            }
                    }
                    ncon_c += pairs;
                    if (p1 == p0) first_pairs = pairs;
                    p1++;
                }
            }
            /* (a single point with more pairs than a chunk holds cannot occur: ba_window_chunkable() keeps such windows
               on the per-block lists; the check below stays as the safety net) */
            (void)first_pairs;
            const int e0 = cnt[p0], e1 = cnt[p1], chid = (int)chd.size() / BA_CH_DESC;
            /* pass 1: which blocks / cameras does the chunk touch, and how many contributions each */
            tmp_blk.clear();
            tmp_slot.clear();
            cb_cnt.clear();
            cg_cnt.clear();
            for (int pt = p0; pt < p1; pt++)
                for (int a = cnt[pt]; a < cnt[pt + 1]; a++) {
                    const int s1 = eslot[a];
                    if (s1 < 0) continue;
                    if (seen_slot[s1] != chid) {
                        seen_slot[s1] = chid;
                        slot_pos[s1] = (int)tmp_slot.size();
                        tmp_slot.push_back(s1);
                        cg_cnt.push_back(0);
                    }
                    cg_cnt[slot_pos[s1]]++;
                    const int *row = &blk[(size_t)s1 * Ncf];
                    for (int b = cnt[pt]; b < cnt[pt + 1]; b++) {
                        const int s2 = eslot[b];
                        if (s2 < s1) continue;
                        const int bk = row[s2];
                        if (seen_blk[bk] != chid) {
                            seen_blk[bk] = chid;
                            blk_pos[bk] = (int)tmp_blk.size();
                            tmp_blk.push_back(bk);
                            cb_cnt.push_back(0);
                        }
                        cb_cnt[blk_pos[bk]]++;
                    }
                }
            const int cb0 = (int)cbb.size(), cg0 = (int)cgs.size();
            if ((int)tmp_blk.size() > BA_CHUNK_B || std::accumulate(cb_cnt.begin(), cb_cnt.end(), 0L) > BA_CHUNK_C) {
                chunk_overflow = true;
                break;
            }
            /* heaviest blocks first: the 32 sixteen-lane groups of a workgroup take blocks u, u + 32, ... so each round
               of blocks has similar list lengths (diagonal blocks get a contribution from every edge of their camera) */
            std::vector<int> ord(tmp_blk.size());
            for (size_t i = 0; i < ord.size(); i++) ord[i] = (int)i;
            std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return cb_cnt[x] > cb_cnt[y]; });
            std::vector<int> cbp(tmp_blk.size()), cgp(tmp_slot.size());
            for (size_t j = 0; j < ord.size(); j++) {
                const int i = ord[j];
                cbb.push_back(tmp_blk[i]);
                cbp[i] = cbs.back();
                cbs.push_back(cbs.back() + cb_cnt[i]);
            }
            for (size_t i = 0; i < tmp_slot.size(); i++) {
                cgs.push_back(tmp_slot[i]);
                cgp[i] = cgst.back();
                cgst.push_back(cgst.back() + cg_cnt[i]);
            }
            cbc.resize(cbs.back());
            cge.resize(cgst.back());
            /* pass 2: fill, in point order (the summation order of every block) */
            for (int pt = p0; pt < p1; pt++)
                for (int a = cnt[pt]; a < cnt[pt + 1]; a++) {
                    const int s1 = eslot[a];
                    if (s1 < 0) continue;
                    cge[cgp[slot_pos[s1]]++] = a - e0;
                    const int *row = &blk[(size_t)s1 * Ncf];
                    for (int b = cnt[pt]; b < cnt[pt + 1]; b++) {
                        const int s2 = eslot[b];
                        if (s2 < s1) continue;
                        cbc[cbp[blk_pos[row[s2]]]++] = (a - e0) | ((b - e0) << 9) | ((pt - p0) << 18);
                    }
                }
            const int d[BA_CH_DESC] = {e0, e1, p0, p1, cb0, (int)cbb.size(), cg0, (int)cgs.size(),
                                       cbs[cb0], cbs.back() - cbs[cb0], cgst[cg0], cgst.back() - cgst[cg0], 0, 0, 0, 0};
            chd.insert(chd.end(), d, d + BA_CH_DESC);
            p0 = p1;
        }
        if (chunk_overflow) return stage_problem(p, hO, hD, scratchD, hI, hB, hw, true, mean_track_limit); /* nothing laid out yet */
    }
    /* ---- lay out ---- */
    auto allocO = [&](size_t n) { /* outputs: downloaded */
        size_t o = hO.size();
        hO.resize(o + ((n + 1) & ~(size_t)1), 0.0);
        return (int64_t)o;
    };
    auto allocD = [&](size_t n) { /* inputs: uploaded only */
        size_t o = hD.size();
        hD.resize(o + ((n + 1) & ~(size_t)1), 0.0);
        return (int64_t)o;
    };
    auto allocS = [&](size_t n) { /* device-only scratch */
        size_t o = scratchD;
        scratchD += (n + 1) & ~(size_t)1;
        return (int64_t)o;
    };
    auto allocI = [&](size_t n) {
        size_t o = hI.size();
        hI.resize(o + n, 0);
        return (int64_t)o;
    };
    auto putI = [&](const std::vector<int> &v) {
        const int64_t o = allocI(std::max<size_t>(v.size(), 1));
        for (size_t i = 0; i < v.size(); i++) hI[o + i] = v[i];
        return o;
    };
    w.poses = allocO((size_t)Nc * 7);
    w.poses_t = allocS((size_t)Nc * 7);
    w.pts = allocO((size_t)Np * 3);
    w.pts_t = allocS((size_t)Np * 3);
    w.obs = allocD((size_t)E * 2);
    w.info = allocD(E);
    w.err = allocS((size_t)E * 2);
    w.err_t = allocS((size_t)E * 2);
    w.W = allocS((size_t)E * 18);
    w.tv = allocS((size_t)Np * 3);
    w.Hpp = allocS((size_t)Ncf * 36);
    w.bp = allocS((size_t)Ncf * 6);
    w.Hll = allocS((size_t)Np * 9);
    w.bl = allocS((size_t)Np * 3);
    w.Dinv = allocS((size_t)Np * 9);
    w.S = allocS((size_t)nnzb * 36);
    w.g = allocS((size_t)Ncf * 6);
    w.Rc = allocS((size_t)Nc * 9);
    w.Uf = chunked ? allocS((size_t)Np * 6) : 0;
    w.tu = chunked ? allocS((size_t)Np * 3) : 0;
    w.S2 = chunked ? allocS((size_t)nnzb * 36) : w.S; /* (never read when the window is not chunked) */
    w.g2 = allocS((size_t)Ncf * 6); /* (also for list windows: the Cholesky of the phase path adds g and g2) */
    w.Minv = allocS((size_t)Ncf * 36);
    w.x = allocS((size_t)Ncf * 6);
    w.xl = allocS(2);
    w.gred = allocS(4 * BA_GMAX);
    w.pcgv = Ncf * 6 > BA_MAX_N ? allocS((size_t)5 * Ncf * 6) : 0; /* PCG vectors of a system too large for the LDS arrays */
    w.st = allocO(ST_COUNT);
    for (int c = 0; c < Nc; c++) {
        double q[4] = {p->poses[c * 7], p->poses[c * 7 + 1], p->poses[c * 7 + 2], p->poses[c * 7 + 3]};
        if (q[0] < 0) { /* SE3Quat ctor: normalizeRotation */
            q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3];
        }
        const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        for (int i = 0; i < 4; i++) hO[w.poses + c * 7 + i] = q[i] / nrm;
        for (int i = 0; i < 3; i++) hO[w.poses + c * 7 + 4 + i] = p->poses[c * 7 + 4 + i];
    }
    for (int i = 0; i < Np; i++)
        for (int j = 0; j < 3; j++) hO[w.pts + (int64_t)rank[i] * 3 + j] = p->points[i * 3 + j];
    for (int k = 0; k < E; k++) {
        const int e = hw.perm[k];
        hD[w.obs + 2 * k] = p->obs[2 * e];
        hD[w.obs + 2 * k + 1] = p->obs[2 * e + 1];
        hD[w.info + k] = p->inv_sigma2[e];
    }
    w.cam = allocI(E);
    w.pt = allocI(E);
    for (int k = 0; k < E; k++) {
        hI[w.cam + k] = p->cam_idx[hw.perm[k]];
        hI[w.pt + k] = rank[p->pt_idx[hw.perm[k]]];
    }
    w.pt_start = putI(cnt);
    w.cam_start = putI(cstart);
    w.cam_edges = putI(cedges);
    w.slot = putI(slot);
    w.blk_row = allocI((size_t)Ncf + 1 + Ncf); /* row starts, then slot -> camera */
    for (int i = 0; i <= Ncf; i++) hI[w.blk_row + i] = brow[i];
    for (int i = 0; i < Ncf; i++) hI[w.blk_row + Ncf + 1 + i] = slot2cam[i];
    w.blk_col = putI(bcol);
    w.blk_diag = putI(bdiag);
    w.blk_mirror = putI(bmirror);
    w.blk_rowid = allocI(std::max(nnzb, 1));
    for (int s1 = 0; s1 < Ncf; s1++)
        for (int bk = brow[s1]; bk < brow[s1 + 1]; bk++) hI[w.blk_rowid + bk] = s1;
    {
        /* envelope (skyline) of S in 16 x 16 tiles for the in-kernel Cholesky: fill-in never leaves the region
           [first(i), i] of each row, and local-BA windows are chain-like (a point is seen by neighbouring
           keyframes), so the envelope is a band -- at 100 KFs x 5000 points about a fifth of the triangle */
        const int nbt = (Ncf * 6 + 15) / 16;
        std::vector<int> first(std::max(nbt, 1), 0), last(std::max(nbt, 1), 0);
        for (int t = 0; t < nbt; t++) first[t] = t;
        for (int s1 = 0; s1 < Ncf; s1++) {
            int mincol = s1; /* lowest block column of this block row (the pattern is symmetric) */
            for (int bk = brow[s1]; bk < brow[s1 + 1]; bk++) mincol = std::min(mincol, bcol[bk]);
            const int tc = (mincol * 6) / 16;
            for (int t = (s1 * 6) / 16; t <= (s1 * 6 + 5) / 16 && t < nbt; t++) first[t] = std::min(first[t], tc);
        }
        for (int t = nbt - 2; t >= 0; t--) first[t] = std::min(first[t], first[t + 1]); /* non-decreasing */
        for (int k = 0; k < nbt; k++) {
            int l = k;
            while (l + 1 < nbt && first[l + 1] <= k) l++;
            last[k] = l;
        }
        int maxm = 1;
        for (int k = 0; k < nbt; k++) maxm = std::max(maxm, last[k] - k);
        w.ch_pm = maxm * 16;
        /* the dense matrix of the blocked Cholesky exists only where its LDS panel fits; a window that is both large
           and nearly dense is solved by PCG on the block-sparse S and needs no np x np storage */
        const size_t np = ((size_t)Ncf * 6 + 15) & ~(size_t)15;
        if (((size_t)w.ch_pm * 17 + 2 * 16 * 17 + np) * 8 <= BA_CHOL_LDS_BYTES) {
            w.A = allocS(np * np);
            w.Linv = allocS((np / 16) * 256);
            /* fill plan of the band Cholesky: nb + 1 starts and two ints per lower-triangle scalar of S (device-built, scratch) */
            w.fplan = -1;
            if (maxm <= CHB_MAX_M && nbt <= CHB_MAX_NB && (size_t)nnzb * 36 < (1u << 30)) {
                size_t nlow = 0;
                for (int i = 0; i < nnzb; i++) nlow += bmirror[i] > i ? 1 : 0; /* strictly upper blocks = strictly lower ones */
                w.fplan = allocS(((size_t)chb_ent0(nbt) + 2 * (nlow * 36 + (size_t)Ncf * 21)) / 2 + 2);
            }
        } else {
            w.ch_pm = -1; /* marks "no Cholesky" for the kernel and the launch code */
            w.A = w.Linv = 0;
            w.fplan = -1;
        }
        w.ch_first = putI(first);
        w.ch_last = putI(last);
    }
    {
        std::vector<int> upper;
        for (int i = 0; i < nnzb; i++)
            if (bmirror[i] >= i) upper.push_back(i);
        /* (row-major order stays: sorting the blocks by list length, heaviest first, was measured -- one window 7.6 -> 8.3 ms of
           kernel time: neighbouring blocks share a camera's W records, and that locality is worth more than the balance) */
        w.nub = (int)upper.size();
        w.blk_upper = putI(upper);
    }
    w.con_start = putI(ccount);
    w.con_e1 = putI(ce1);
    w.con_e2 = putI(ce2);
    w.nchunks = (int)chd.size() / BA_CH_DESC;
    {
        /* first-touch flags of the two-part Schur pass (part p = chunks [nchunks p / 2, nchunks (p + 1) / 2)) */
        std::vector<int> touch(std::max(nnzb, 1), 0);
        for (int part = 0; part < BA_SCHUR_PARTS; part++) {
            const int c0 = (int)((int64_t)w.nchunks * part / BA_SCHUR_PARTS), c1 = (int)((int64_t)w.nchunks * (part + 1) / BA_SCHUR_PARTS);
            for (int ch = c0; ch < c1; ch++)
                for (int k = chd[(size_t)ch * BA_CH_DESC + 4]; k < chd[(size_t)ch * BA_CH_DESC + 5]; k++) {
                    const int bk = cbb[k];
                    if (bcol[bk] == hI[w.blk_rowid + bk]) cbb[k] |= BA_BLK_DIAG;
                    if (!(touch[bk] & (1 << part))) {
                        touch[bk] |= 1 << part;
                        cbb[k] |= BA_BLK_FIRST;
                    }
                }
        }
        w.blk_touch = putI(touch);
    }
    w.ch_desc = putI(chd);
    w.cb_blk = putI(cbb);
    w.cb_start = putI(cbs);
    w.cb_con = putI(cbc);
    w.cg_slot = putI(cgs);
    w.cg_start = putI(cgst);
    w.cg_edge = putI(cge);
    w.gbar = allocI(4); /* two alternating arrival counters + the abort word */
    w.flags = (int64_t)hB.size();
    hB.resize(hB.size() + ((E + 15) & ~15), 0);
    for (int k = 0; k < E; k++) hB[w.flags + k] = (uint8_t)(1 | (p->huber_delta > 0 ? 2 : 0));
    return ORB_OK;
}

/* One batch in flight: its own pinned staging and device buffers, so that the host side of batch k+1 (index sorting,
 * contribution lists, concatenation, upload) and the download / unpacking of batch k-1 overlap the kernels of
 * batch k.  ba_solve_batch uses slot 0 synchronously; ba_submit_batch / ba_wait alternate between the two. */
struct BAJob {
    double *pD = nullptr; /* pinned */
    int32_t *pI = nullptr;
    uint8_t *pB = nullptr;
    size_t pcapD = 0, pcapI = 0, pcapB = 0;
    double *dD = nullptr; /* device */
    int32_t *dI = nullptr;
    uint8_t *dB = nullptr;
    BAWin *dW = nullptr;
    size_t capD = 0, capI = 0, capB = 0, capW = 0;
    std::vector<HostWin> hw;
    std::vector<BAWin> wins;
    size_t nOut = 0, nIn = 0, nI = 0, nB = 0, scratchD = 0; /* element counts */
    const BAProblem *probs = nullptr;
    BAResult *res = nullptr;
    int nwin = 0;
    bool pending = false;
    hipEvent_t uploaded = nullptr, solved = nullptr, downloaded = nullptr;
    std::vector<hipEvent_t> prof_ev; /* pairs around every k_ba_iteration launch (profiling only) */
    int prof_launches = 0;
    int team = 1;
    bool split = false; /* ran as phase kernels (k_ba_ph_*) */
    double t_start = 0, t_staged = 0, t_enqueued = 0;
};

/* batches in flight through ba_submit_batch / ba_wait: each slot has its own pinned staging and device buffers.  Two slots
   overlap the host side of batch k+1 with the kernels of batch k; with three the device queue always holds a staged batch
   beyond the running one, so a slow staging pass (a box with less CPU) no longer shows as an idle device between batches. */
#define BA_SLOTS 3
struct BASolver {
    int device = 0;
    hipStream_t stream = nullptr;   /* kernels */
    hipStream_t copy_in = nullptr;  /* uploads of the NEXT batch */
    hipStream_t copy_out = nullptr; /* downloads of the PREVIOUS batch */
    int max_cams, max_points, max_obs, max_batch;
    int cu_count = 256;
    BAJob job[BA_SLOTS];
    int next_slot = 0, oldest_slot = 0, in_flight = 0;
    std::vector<BAPiece> pieces;
    /* Optimizer::PoseOptimization has its own small buffers: Tracking may call it while a BA batch is in flight */
    double *poD = nullptr, *ppD = nullptr;
    int32_t *poI = nullptr, *ppI = nullptr;
    uint8_t *poB = nullptr, *ppB = nullptr;
    size_t pocapD = 0, pocapI = 0, pocapB = 0, ppcapD = 0, ppcapI = 0, ppcapB = 0;
    double pcg_tol2 = 1e-26; /* relative M^-1-norm of the residual, squared */
    int solver = 1;          /* 1 = dense blocked Cholesky (MFMA f64), 0 = PCG (ORBSLAM_AMD_BA_SOLVER=pcg) */
    int force_team = 0;      /* ORBSLAM_AMD_BA_TEAM: workgroups per window (0 = choose) */
    int force_split = -1;    /* ORBSLAM_AMD_BA_SPLIT: 0 fused kernel, 1 phase kernels when eligible, -1 by batch size */
    int spare_trials = BA_PH_SPARE; /* ORBSLAM_AMD_BA_SPARE */
    bool profiling = false;
    float last_iter_ms = 0.f; /* average k_ba_iteration launch of the last finished batch */
    int last_iter_launches = 0, last_team = 1;
};

extern "C" int ba_create(int max_cams, int max_points, int max_obs, int max_batch, int device_id, BASolver **out) {
    if (!out || max_cams <= 0 || max_points <= 0 || max_obs <= 0 || max_batch <= 0) return ORB_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return ORB_E_NO_DEVICE;
    BCHK(hipSetDevice(device_id));
    orb_trace_exit_late();
    BASolver *s = new BASolver();
    s->device = device_id;
    s->max_cams = max_cams;
    s->max_points = max_points;
    s->max_obs = max_obs;
    s->max_batch = max_batch;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) s->cu_count = prop.multiProcessorCount;
    bool ok = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&s->copy_in, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&s->copy_out, hipStreamNonBlocking) == hipSuccess;
    for (int j = 0; j < BA_SLOTS && ok; j++)
        ok = hipEventCreateWithFlags(&s->job[j].uploaded, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s->job[j].solved, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s->job[j].downloaded, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        b_err = "ba_create: stream / event creation failed";
        ba_destroy(s);
        return ORB_E_HIP;
    }
    const char *tol = getenv("ORBSLAM_AMD_PCG_TOL");
    if (tol) s->pcg_tol2 = atof(tol) * atof(tol);
    if (getenv("ORBSLAM_AMD_CHOL_SCALAR")) {
        int one = atoi(getenv("ORBSLAM_AMD_CHOL_SCALAR"));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_chol_scalar), &one, sizeof(int));
    }
    const char *sv = getenv("ORBSLAM_AMD_BA_SOLVER");
    if (sv && !strcmp(sv, "pcg")) s->solver = 0;
    if (getenv("ORBSLAM_AMD_BA_TEAM")) s->force_team = atoi(getenv("ORBSLAM_AMD_BA_TEAM"));
    /* tests / diagnostics: 0 / 1 = fused kernel / phase kernels (when eligible) whatever the batch size; trials enqueued per
       round beyond its iterations (a negative value leaves trials -- all of them at -100 -- to k_ba_ph_finish) */
    if (getenv("ORBSLAM_AMD_BA_SPLIT")) s->force_split = atoi(getenv("ORBSLAM_AMD_BA_SPLIT"));
    if (getenv("ORBSLAM_AMD_BA_SPARE")) s->spare_trials = atoi(getenv("ORBSLAM_AMD_BA_SPARE"));
    *out = s;
    return ORB_OK;
}

/* ORBSLAM_AMD_TRACE_EXIT=1: markers on stderr around the teardown (tools/exit_hang_probe.sh) */
static bool ba_trace_exit() {
    static const bool on = getenv("ORBSLAM_AMD_TRACE_EXIT") != nullptr;
    return on;
}
namespace {
struct BAExitMarker {
    ~BAExitMarker() {
        if (ba_trace_exit()) fprintf(stderr, "[liborbslam_amd] static destructor of ba_api (exit handlers are running)\n");
    }
} g_ba_exit_marker;
}

extern "C" void ba_destroy(BASolver *s) {
    if (!s) return;
    if (ba_trace_exit()) fprintf(stderr, "[liborbslam_amd] ba_destroy: begin\n");
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->copy_in) (void)hipStreamSynchronize(s->copy_in);
    if (s->copy_out) (void)hipStreamSynchronize(s->copy_out);
    for (BAJob &j : s->job) {
        (void)hipFree(j.dD);
        (void)hipFree(j.dI);
        (void)hipFree(j.dB);
        (void)hipFree(j.dW);
        if (j.pD) (void)hipHostFree(j.pD);
        if (j.pI) (void)hipHostFree(j.pI);
        if (j.pB) (void)hipHostFree(j.pB);
        if (j.uploaded) (void)hipEventDestroy(j.uploaded);
        if (j.solved) (void)hipEventDestroy(j.solved);
        if (j.downloaded) (void)hipEventDestroy(j.downloaded);
        for (hipEvent_t e : j.prof_ev) (void)hipEventDestroy(e);
    }
    (void)hipFree(s->poD);
    (void)hipFree(s->poI);
    (void)hipFree(s->poB);
    if (s->ppD) (void)hipHostFree(s->ppD);
    if (s->ppI) (void)hipHostFree(s->ppI);
    if (s->ppB) (void)hipHostFree(s->ppB);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->copy_in) (void)hipStreamDestroy(s->copy_in);
    if (s->copy_out) (void)hipStreamDestroy(s->copy_out);
    delete s;
    if (ba_trace_exit()) fprintf(stderr, "[liborbslam_amd] ba_destroy: end\n");
}

extern "C" int ba_set_profiling(BASolver *s, int enable) {
    if (!s) return ORB_E_INVALID;
    s->profiling = enable != 0;
    return ORB_OK;
}

extern "C" int ba_get_iteration_ms(BASolver *s, float *avg_ms, int32_t *n_launches, int32_t *team) {
    if (!s) return ORB_E_INVALID;
    if (avg_ms) *avg_ms = s->last_iter_ms;
    if (n_launches) *n_launches = s->last_iter_launches;
    if (team) *team = s->last_team;
    return ORB_OK;
}

template <typename T> static int ba_grow(T *&p, size_t &cap, size_t need) {
    if (need <= cap && p) return ORB_OK;
    if (p) BCHK(hipFree(p));
    p = nullptr;
    cap = 0;
    BCHK(hipMalloc((void **)&p, std::max<size_t>(need * sizeof(T), 256)));
    cap = need;
    return ORB_OK;
}

template <typename T> static int ba_grow_pinned(T *&p, size_t &cap, size_t need) {
    if (need <= cap && p) return ORB_OK;
    if (p) BCHK(hipHostFree(p));
    p = nullptr;
    cap = 0;
    need += need / 4; /* head-room: batches of similar windows do not re-allocate */
    BCHK(hipHostMalloc((void **)&p, std::max<size_t>(need * sizeof(T), 256)));
    cap = need;
    return ORB_OK;
}

static double ba_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

/* host threads of the staging / unpacking pools.  A one-GPU box has a CPU share of 16 (cgroup quota per 100 ms), but staging
   comes in bursts -- 24 ms of 16 threads per 256-window batch, then nothing until the next batch -- and a burst may run wider
   than the share as long as the period's budget holds: 32 threads measured 7050 windows/s streamed against 6320 with 16 (64:
   7490, too close to the quota to be the default).  ORBSLAM_AMD_BA_HOST_THREADS overrides. */
static int ba_host_threads() {
    static const int n = getenv("ORBSLAM_AMD_BA_HOST_THREADS") ? std::max(1, atoi(getenv("ORBSLAM_AMD_BA_HOST_THREADS"))) : 32;
    return n;
}
template <class F> static void ba_parallel(int nthr, F &&f) {
    if (nthr <= 1) {
        f(0);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < nthr; t++) pool.emplace_back(f, t);
    for (auto &th : pool) th.join();
}

/* workgroups per window: as many as fit co-resident AND as the window's size can feed (a team member needs a few
 * hundred edges per phase to be worth its barrier) */
static int ba_choose_team(BASolver *s, int nwin, int maxE) {
    /* k_ba_iteration holds 256 VGPRs per lane: two waves per SIMD = exactly ONE 512-thread workgroup per CU, whatever
       the LDS would allow.  Not asked of hipOccupancyMaxActiveBlocksPerMultiprocessor: that API can answer one block
       per CU too high (MI355X_MICROARCH.md, correctness boundaries) and a team barrier would then wait for a workgroup
       that is not resident. */
    const int resident = s->cu_count;
    /* measured (tools/ba_probe.py, one window in flight): 28 k edges 19.5 / 14.0 / 11.5 / 11.9 / 15.0 ms with 4 / 8 / 16 /
       32 / 64 workgroups, 11 k edges 9.2 / 7.3 / 6.8 / 7.9 ms with 4 / 8 / 16 / 32, 2.9 k edges 4.5 / 4.1 / 4.0 ms with
       1 / 4 / 8: a barrier costs more the more workgroups arrive at its counter, and past 16 that outweighs the split */
    int want = s->force_team > 0 ? s->force_team : std::max(1, std::min(16, maxE / 700));
    int G = 1;
    while (G * 2 <= BA_GMAX && G * 2 <= want && (long)nwin * (G * 2) <= resident) G *= 2;
    return G;
}

/* host side of a batch: per-window staging on a pool of host threads (index sorting and the contribution lists are
 * O(sum k^2) per window), then concatenation into the job's pinned buffers */
static int ba_stage(BASolver *s, BAJob &J, const BAProblem *probs, BAResult *res, int nwin) {
    const BAProblem &P0 = probs[0];
    for (int i = 0; i < nwin; i++) {
        if (probs[i].n_rounds != P0.n_rounds || P0.n_rounds < 0 || P0.n_rounds > (int)(sizeof(P0.rounds) / sizeof(P0.rounds[0])) ||
            memcmp(probs[i].rounds, P0.rounds, (size_t)P0.n_rounds * sizeof(P0.rounds[0])) ||   /* entries past n_rounds are the caller's to leave unset */
            probs[i].drop_robust_after != P0.drop_robust_after) {
            b_err = "ba_solve_batch: all windows of a batch must share one iteration schedule";
            return ORB_E_INVALID;
        }
        if (!res[i].poses || (probs[i].n_points && !res[i].points)) return ORB_E_INVALID;
    }
    J.t_start = ba_now();
    J.probs = probs;
    J.res = res;
    J.nwin = nwin;
    J.hw.resize(nwin);
    J.wins.resize(nwin);
    int maxE = 0;
    for (int i = 0; i < nwin; i++) maxE = std::max(maxE, probs[i].n_obs);
    J.team = ba_choose_team(s, nwin, maxE);
    /* Phase kernels instead of teams (or one workgroup per window) from a handful of windows on -- decided here because
       it decides the staging: point chunks for the phase kernels, per-block lists for teams.  Measured, ms of kernel
       time per batch, teams or fused / phase kernels:
           100 x 5000 (28 k edges): 1 window 9.0 / 16.0, 3: 11.4 / 16.8, 5: 14.7 / 17.1, 8: 19.6 / 16.9, 16: 40.4 / 18.5,
                                    32: 41.9 / 22.3, 64: 55.8 / 25.1, 128: 81.6 / 29.1
           50 x 2000 (11 k edges):  4 windows 8.8 / 8.2, 8: 16.4 / 9.0, 16: 34.0 / 9.7, 64: 36.7 / 10.7
           20 x 300:                8 windows 4.2 / 3.5, 32: 6.9 / 3.5, 128: 23.0 / 4.0          8 x 150: equal
       A team's barriers and per-block Schur lists do not scale to many teams at once, while the phase kernels spread
       every window's streaming phases over the machine by themselves. */
    {
        bool any_stop = false;
        for (int i = 0; i < nwin; i++) any_stop |= probs[i].stop != nullptr;
        const bool wanted = s->force_split == 1 || nwin >= BA_SPLIT_MIN_WINDOWS || (nwin >= BA_SPLIT_MIN_SMALL && maxE < BA_SPLIT_BIG_EDGES);
        if (wanted && s->force_split != 0 && s->force_team == 0 && s->solver == 1 && !any_stop && !getenv("ORBSLAM_AMD_BA_PROFILE") &&
            !getenv("ORBSLAM_AMD_BA_DEBUG")) {
            J.team = 1; /* (stage_problem lays a window out in point chunks or, long tracks, in per-block lists: both run as
                           phase kernels) */
        }
    }
    const bool team = J.team > 1;
    /* point chunks or per-block lists for the windows of a batch that runs as phase kernels: with the batch's W inside the
       Infinity Cache the lists are as good as chunks at short tracks and better beyond (ba_window_chunkable) */
    size_t total_obs = 0;
    for (int i = 0; i < nwin; i++) total_obs += (size_t)std::max(probs[i].n_obs, 0);
    const int mean_limit = total_obs * 144 <= (size_t)BA_CACHED_W_BYTES ? BA_CHUNK_MEAN_TRACK_CACHED : BA_CHUNK_MEAN_TRACK;
    if ((int)s->pieces.size() < nwin) s->pieces.resize(nwin);
    std::vector<BAPiece> &pieces = s->pieces;
    const int nthr = std::max(1, std::min<int>(nwin, std::min<int>(ba_host_threads(), (int)std::thread::hardware_concurrency())));
    ba_parallel(nthr, [&](int t) {
        for (int i = t; i < nwin; i += nthr) {
            BAPiece &pc = pieces[i];
            pc.O.clear();
            pc.D.clear();
            pc.I.clear();
            pc.B.clear();
            pc.scratch = 0;
            pc.rc = stage_problem(&probs[i], pc.O, pc.D, pc.scratch, pc.I, pc.B, J.hw[i], team, mean_limit);
        }
    });
    size_t nO = 0, nD = 0, nI = 0, nB = 0;
    std::vector<size_t> oO(nwin), oD(nwin), oI(nwin), oB(nwin);
    for (int i = 0; i < nwin; i++) {
        if (pieces[i].rc) return pieces[i].rc;
        oO[i] = nO;
        oD[i] = nD;
        oI[i] = nI;
        oB[i] = nB;
        nO += pieces[i].O.size();
        nD += pieces[i].D.size();
        nI += pieces[i].I.size();
        nB += pieces[i].B.size();
    }
    int rc;
    if ((rc = ba_grow_pinned(J.pD, J.pcapD, nO + nD))) return rc;
    if ((rc = ba_grow_pinned(J.pI, J.pcapI, std::max<size_t>(nI, 1)))) return rc;
    if ((rc = ba_grow_pinned(J.pB, J.pcapB, std::max<size_t>(nB, 1)))) return rc;
    J.nOut = nO;
    J.nIn = nD;
    J.nI = nI;
    J.nB = nB;
    ba_parallel(nthr, [&](int t) {
        for (int i = t; i < nwin; i += nthr) {
            BAPiece &pc = pieces[i];
            memcpy(J.pD + oO[i], pc.O.data(), pc.O.size() * sizeof(double));
            memcpy(J.pD + nO + oD[i], pc.D.data(), pc.D.size() * sizeof(double));
            memcpy(J.pI + oI[i], pc.I.data(), pc.I.size() * sizeof(int32_t));
            memcpy(J.pB + oB[i], pc.B.data(), pc.B.size());
        }
    });
    size_t scratchD = 0;
    for (int i = 0; i < nwin; i++) {
        BAWin &w = J.hw[i].w;
        for (auto f : kOutDFields) w.*f += (int64_t)oO[i];
        for (auto f : kInDFields) w.*f += (int64_t)(nO + oD[i]);
        for (auto f : kIntFields) w.*f += (int64_t)oI[i];
        w.flags += (int64_t)oB[i];
        for (auto f : kScratchFields) w.*f += (int64_t)(nO + nD + scratchD);
        if (w.fplan >= 0) w.fplan += (int64_t)(nO + nD + scratchD); /* (-1: the window has no fill plan) */
        scratchD += pieces[i].scratch;
        J.wins[i] = w;
    }
    J.scratchD = scratchD;
    J.t_staged = ba_now();
    return ORB_OK;
}

/* device side of a batch: upload on the copy-in stream, LM iterations on the compute stream, download of the output
 * region on the copy-out stream; returns without waiting unless a stop flag has to be polled between iterations */
static int ba_enqueue(BASolver *s, BAJob &J) {
    const int nwin = J.nwin;
    hipStream_t ks = s->stream; /* (a kernel stream per job in flight -- the thin tail of one batch beside the bulk of the next --
                                   was measured: 5141 windows/s against 5875 with the batches back to back on one stream) */
    const BAProblem *probs = J.probs;
    const BAProblem &P0 = probs[0];
    const bool prof = getenv("ORBSLAM_AMD_BA_PROFILE") != nullptr;
    int rc;
    if ((rc = ba_grow(J.dD, J.capD, J.nOut + J.nIn + J.scratchD))) return rc;
    if ((rc = ba_grow(J.dI, J.capI, std::max<size_t>(J.nI, 1)))) return rc;
    if ((rc = ba_grow(J.dB, J.capB, std::max<size_t>(J.nB, 1)))) return rc;
    if ((rc = ba_grow(J.dW, J.capW, (size_t)nwin))) return rc;
    BCHK(hipMemcpyAsync(J.dD, J.pD, (J.nOut + J.nIn) * 8, hipMemcpyHostToDevice, s->copy_in));
    if (J.nI) BCHK(hipMemcpyAsync(J.dI, J.pI, J.nI * 4, hipMemcpyHostToDevice, s->copy_in));
    if (J.nB) BCHK(hipMemcpyAsync(J.dB, J.pB, J.nB, hipMemcpyHostToDevice, s->copy_in));
    BCHK(hipMemcpyAsync(J.dW, J.wins.data(), (size_t)nwin * sizeof(BAWin), hipMemcpyHostToDevice, s->copy_in));
    BCHK(hipEventRecord(J.uploaded, s->copy_in));
    BCHK(hipStreamWaitEvent(ks, J.uploaded, 0));
    BABufs bufs{J.dD, J.dI, J.dB, J.dW};
    bool any_stop_ptr = false;
    for (int i = 0; i < nwin; i++) any_stop_ptr |= probs[i].stop != nullptr;
    auto stopped = [&]() {
        for (int i = 0; i < nwin; i++)
            if (probs[i].stop && *probs[i].stop) return true;
        return false;
    };
    int maxn = 0;
    for (int i = 0; i < nwin; i++) maxn = std::max(maxn, J.wins[i].Ncf * 6);
    int pcg_maxit = 6 * maxn + 100; /* CG needs > n steps in floating point when S is ill-conditioned */
    if (getenv("ORBSLAM_AMD_PCG_MAXIT")) pcg_maxit = atoi(getenv("ORBSLAM_AMD_PCG_MAXIT")); /* profiling only */
    /* dynamic LDS of the Cholesky solver: envelope-sized panel + two 16x16 blocks + rhs, for the largest window of the
       batch that has a dense matrix at all (ch_pm > 0); the others take PCG inside the kernel */
    int dyn_doubles = 0;
    if (s->solver == 1) {
        for (int i = 0; i < nwin; i++) {
            if (J.wins[i].ch_pm <= 0) continue;
            const size_t npw = ((size_t)J.wins[i].Ncf * 6 + 15) & ~(size_t)15;
            dyn_doubles = std::max(dyn_doubles, (int)((size_t)J.wins[i].ch_pm * 17 + 2 * 16 * 17 + npw));
        }
    }
    for (int i = 0; i < nwin; i++) /* the LDS tiles of the chunked Schur pass share the region with the Cholesky panel */
        if (J.team == 1 && J.wins[i].nchunks > 0) dyn_doubles = std::max(dyn_doubles, (int)BA_CHUNK_DOUBLES);
    /* the band Cholesky (LDS-resident window, ba_chol_band) inside the fused kernel: when every window with a dense matrix has
       a fill plan and a narrow band; its window must fit beside the static LDS (ORBSLAM_AMD_BA_CHOL_BAND=0: round 2's solver) */
    const int fused_band_env = getenv("ORBSLAM_AMD_BA_CHOL_BAND") ? atoi(getenv("ORBSLAM_AMD_BA_CHOL_BAND")) : 1;
    int fused_band = s->solver == 1 && fused_band_env != 0 ? 1 : 0;
    {
        size_t need = 0;
        bool any = false;
        for (int i = 0; i < nwin && fused_band; i++) {
            const BAWin &w = J.wins[i];
            if (w.ch_pm <= 0) continue;
            if (w.fplan < 0 || (w.ch_pm >> 4) > CHB_MAX_M) fused_band = 0;
            need = std::max(need, chb_lds_doubles(w.ch_pm >> 4, (w.Ncf * 6 + 15) & ~15));
            any = true;
        }
        if (!any || need * 8 > (size_t)BA_CHOL_LDS_BYTES) fused_band = 0;
        if (fused_band) dyn_doubles = std::max(dyn_doubles, (int)need);
    }
    {
        static PerDeviceLdsLimit lim;
        BCHK(lim.ensure((const void *)k_ba_iteration, (size_t)dyn_doubles * 8));
    }
    const int G = J.team;
    const int debug = getenv("ORBSLAM_AMD_BA_DEBUG") ? 2 : (prof ? 1 : 0);
    int total_its = 0;
    for (int r = 0; r < P0.n_rounds; r++) total_its += P0.rounds[r];
    if (s->profiling) {
        while ((int)J.prof_ev.size() < 2 * std::max(total_its, P0.n_rounds)) { /* (the phase-kernel path brackets rounds, also empty ones) */
            hipEvent_t e;
            BCHK(hipEventCreate(&e));
            J.prof_ev.push_back(e);
        }
    }
    J.prof_launches = 0;
    int seq = 0;
    bool halted = false;
    /* the phase-kernel path: a batch of whole workgroups per window, every window on the chunked Schur pass and the
       dense Cholesky, nobody polling a stop flag, no in-kernel timers asked for.  ORBSLAM_AMD_BA_SPLIT=0 / 1 forces
       the fused kernel / the phase kernels (when eligible) whatever the batch size. */
    bool split = G == 1 && s->solver == 1 && !any_stop_ptr && debug == 0;
    size_t chol_doubles = 0;
    for (int i = 0; i < nwin && split; i++) {
        const BAWin &w = J.wins[i];
        const size_t need = (size_t)std::max(w.ch_pm, 0) * CH_LD + 2 * 16 * CH_LD + (((size_t)w.Ncf * 6 + 15) & ~(size_t)15);
        if (w.ch_pm <= 0 || w.Ncf <= 0 || need > (size_t)dyn_doubles) split = false;
        chol_doubles = std::max(chol_doubles, need);
    }
    if (s->force_split == 0) split = false;
    {
        int maxE_ = 0;
        for (int i = 0; i < nwin; i++) maxE_ = std::max(maxE_, J.wins[i].E);
        if (s->force_split != 1 && nwin < BA_SPLIT_MIN_WINDOWS && !(nwin >= BA_SPLIT_MIN_SMALL && maxE_ < BA_SPLIT_BIG_EDGES)) split = false;
    }
    J.split = split;
    if (split) {
        static PerDeviceLdsLimit lim_schur, lim_chol, lim_fin;
        static const int schur_skip = getenv("ORBSLAM_AMD_BA_SCHUR_SKIP") ? atoi(getenv("ORBSLAM_AMD_BA_SCHUR_SKIP")) : 0; /* timing experiments (results invalid) */
        BCHK(lim_schur.ensure((const void *)k_ba_ph_schur2, (size_t)BA_CHUNK_DOUBLES * 8));
        BCHK(lim_chol.ensure((const void *)k_ba_ph_chol, chol_doubles * 8));
        /* band Cholesky with the LDS-resident window (ba_chol_band) when every window's band is narrow enough; 256 threads
           per window -- two windows per CU -- once the batch can fill the device that way.  ORBSLAM_AMD_BA_CHOL_BAND=0:
           the round-2 kernel; ORBSLAM_AMD_BA_CHOL_NT=256 / 512 forces the workgroup size (tests, measurements). */
        /* (read per call, not cached: the tests switch them inside one process) */
        const int band_env = getenv("ORBSLAM_AMD_BA_CHOL_BAND") ? atoi(getenv("ORBSLAM_AMD_BA_CHOL_BAND")) : 1;
        const int band_nt_env = getenv("ORBSLAM_AMD_BA_CHOL_NT") ? atoi(getenv("ORBSLAM_AMD_BA_CHOL_NT")) : 0;
        static const int band_skip = getenv("ORBSLAM_AMD_BA_CHOL_SKIP") ? atoi(getenv("ORBSLAM_AMD_BA_CHOL_SKIP")) : 0; /* timing experiments (results invalid) */
        size_t band_doubles = 0, fin_doubles = std::max<size_t>(chol_doubles, 2 * (size_t)BA_CHUNK_DOUBLES);
        int band_maxm = 0;
        bool band = band_env != 0;
        for (int i = 0; i < nwin && band; i++) {
            const BAWin &w = J.wins[i];
            const int M = w.ch_pm / 16;
            if (M < 1 || M > CHB_MAX_M || w.fplan < 0) band = false;
            band_maxm = std::max(band_maxm, M);
            band_doubles = std::max(band_doubles, chb_lds_doubles(M, (w.Ncf * 6 + 15) & ~15));
        }
        if (band_doubles * 8 > (size_t)BA_CHOL_LDS_BYTES) band = false;
        int band_nt = (nwin >= 2 * s->cu_count - s->cu_count / 4 && band_doubles * 8 <= 72 * 1024) ? 256 : 512;
        if (band_nt_env == 256 || band_nt_env == 512) band_nt = band_nt_env;
        const void *band_fn = band_nt == 256 ? (band_maxm <= 4 ? (const void *)k_ba_ph_chol_band<256, 4> : (const void *)k_ba_ph_chol_band<256, 8>)
                                             : (band_maxm <= 4 ? (const void *)k_ba_ph_chol_band<512, 4> : (const void *)k_ba_ph_chol_band<512, 8>);
        if (band) {
            static PerDeviceLdsLimit lim_band[4];
            BCHK(lim_band[(band_nt == 256 ? 0 : 2) + (band_maxm <= 4 ? 0 : 1)].ensure(band_fn, band_doubles * 8));
            fin_doubles = std::max(fin_doubles, band_doubles);
        }
        BCHK(lim_fin.ensure((const void *)k_ba_ph_finish, fin_doubles * 8));
        static const bool phase_prof = getenv("ORBSLAM_AMD_BA_PHASE_PROF") != nullptr; /* diagnostics: serialises the phases */
        double phase_ms[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        hipEvent_t pe0 = nullptr, pe1 = nullptr;
        if (phase_prof) {
            BCHK(hipEventCreate(&pe0));
            BCHK(hipEventCreate(&pe1));
        }
        auto timed = [&](int slot, auto &&launch) -> int {
            if (phase_prof) BCHK(hipEventRecord(pe0, ks));
            launch();
            BCHK(hipGetLastError());
            if (phase_prof) {
                BCHK(hipEventRecord(pe1, ks));
                BCHK(hipEventSynchronize(pe1));
                float ms = 0;
                BCHK(hipEventElapsedTime(&ms, pe0, pe1));
                phase_ms[slot] += ms;
            }
            return ORB_OK;
        };
        /* parts per window of the streaming phases: enough workgroups of BA_PH_T threads for the largest window, capped
           so that a batch of a few hundred windows is a few thousand workgroups */
        int maxE = 0, maxP = 0, maxC = 0;
        for (int i = 0; i < nwin; i++) {
            maxE = std::max(maxE, J.wins[i].E);
            maxP = std::max(maxP, J.wins[i].Np);
            maxC = std::max(maxC, J.wins[i].Nc);
        }
        const int partsE = ba_ph_parts_e(maxE), partsP = ba_ph_parts_p(maxP);
        bool any_lists = false;
        int max_nub = 0;
        for (int i = 0; i < nwin; i++)
            if (J.wins[i].nchunks <= 0) {
                any_lists = true;
                max_nub = std::max(max_nub, J.wins[i].nub);
            }
        const int parts_lists = std::max(1, std::min(16, (max_nub + 4 * BA_W - 1) / (4 * BA_W))); /* four blocks per wavefront and step */
        const int partsC = std::max(1, (maxC + BA_PH_T / 64 - 1) / (BA_PH_T / 64));
        int trial_seq = 0;
        if (band) { /* once per batch: the windows' patterns do not change between trials */
            hipLaunchKernelGGL(k_ba_ph_fillplan, dim3(nwin), dim3(256), 0, ks, bufs);
            BCHK(hipGetLastError());
        }
        for (int round = 0; round < P0.n_rounds; round++) {
            const int nit = P0.rounds[round];
            if (s->profiling) BCHK(hipEventRecord(J.prof_ev[2 * J.prof_launches], ks));
            hipLaunchKernelGGL(k_ba_ph_round, dim3((nwin + 63) / 64), dim3(64), 0, ks, bufs, nwin, nit);
            BCHK(hipGetLastError());
            const int trials = nit > 0 ? std::max(0, nit + s->spare_trials) : 0;
            for (int t = 0; t < trials; t++, trial_seq++) {
                const dim3 gE(nwin, partsE), gP(nwin, partsP), gC(nwin, partsC), tb(BA_PH_T);
                if ((rc = timed(0, [&] { hipLaunchKernelGGL(k_ba_ph_chi2, gE, tb, 0, ks, bufs); }))) return rc;
                if ((rc = timed(1, [&] { hipLaunchKernelGGL(k_ba_ph_hll, gP, tb, 0, ks, bufs); }))) return rc;
                if ((rc = timed(2, [&] { hipLaunchKernelGGL(k_ba_ph_hpp, gC, tb, 0, ks, bufs); }))) return rc;
                if ((rc = timed(3, [&] {
                         hipLaunchKernelGGL(k_ba_ph_prep, dim3(nwin), dim3(BA_T), 0, ks, bufs);
                         hipLaunchKernelGGL(k_ba_ph_dinv, gP, tb, 0, ks, bufs);
                     })))
                    return rc;
                if ((rc = timed(4, [&] {
                         hipLaunchKernelGGL(k_ba_ph_schur2, dim3(nwin, 2), dim3(BA_SH_T), (size_t)BA_CHUNK_DOUBLES * 8, ks, bufs, schur_skip);
                     })))
                    return rc;
                if (any_lists)
                    if ((rc = timed(4, [&] { hipLaunchKernelGGL(k_ba_ph_schur_lists, dim3(nwin, parts_lists), dim3(BA_T), 0, ks, bufs); }))) return rc;
                if ((rc = timed(5, [&] {
                         if (!band)
                             hipLaunchKernelGGL(k_ba_ph_chol, dim3(nwin), dim3(BA_T), chol_doubles * 8, ks, bufs);
                         else if (band_nt == 256 && band_maxm <= 4)
                             hipLaunchKernelGGL((k_ba_ph_chol_band<256, 4>), dim3(nwin), dim3(256), band_doubles * 8, ks, bufs, band_skip);
                         else if (band_nt == 256)
                             hipLaunchKernelGGL((k_ba_ph_chol_band<256, 8>), dim3(nwin), dim3(256), band_doubles * 8, ks, bufs, band_skip);
                         else if (band_maxm <= 4)
                             hipLaunchKernelGGL((k_ba_ph_chol_band<512, 4>), dim3(nwin), dim3(512), band_doubles * 8, ks, bufs, band_skip);
                         else
                             hipLaunchKernelGGL((k_ba_ph_chol_band<512, 8>), dim3(nwin), dim3(512), band_doubles * 8, ks, bufs, band_skip);
                     })))
                    return rc;
                if ((rc = timed(6, [&] { hipLaunchKernelGGL(k_ba_ph_backsub, gP, tb, 0, ks, bufs); }))) return rc;
                if ((rc = timed(7, [&] { hipLaunchKernelGGL(k_ba_ph_trial, gE, tb, 0, ks, bufs); }))) return rc;
                if ((rc = timed(8, [&] {
                         hipLaunchKernelGGL(k_ba_ph_decide, dim3((nwin + 63) / 64), dim3(64), 0, ks, bufs, nwin, trial_seq);
                         hipLaunchKernelGGL(k_ba_ph_accept, gE, tb, 0, ks, bufs, trial_seq);
                     })))
                    return rc;
            }
            if (nit > 0)
                if ((rc = timed(9, [&] { hipLaunchKernelGGL(k_ba_ph_finish, dim3(nwin), dim3(BA_T), fin_doubles * 8, ks, bufs, band ? 1 : 0); }))) return rc;
            if (s->profiling) BCHK(hipEventRecord(J.prof_ev[2 * J.prof_launches + 1], ks));
            J.prof_launches++;
            if (round + 1 < P0.n_rounds) {
                hipLaunchKernelGGL(k_ba_classify, dim3(nwin), dim3(BA_T), 0, ks, bufs, P0.drop_robust_after == round ? 1 : 0, 0);
                BCHK(hipGetLastError());
            }
        }
        if (phase_prof) {
            fprintf(stderr, "[ba] phase kernels, %d windows, ms: chi2 %.2f | Hll %.2f | Hpp+W %.2f | Dinv %.2f | Schur %.2f | Cholesky %.2f | backsub %.2f | trial %.2f | decide+accept %.2f | finish %.2f\n",
                    nwin, phase_ms[0], phase_ms[1], phase_ms[2], phase_ms[3], phase_ms[4], phase_ms[5], phase_ms[6], phase_ms[7], phase_ms[8], phase_ms[9]);
            (void)hipEventDestroy(pe0);
            (void)hipEventDestroy(pe1);
        }
    }
    if (!split && fused_band) { /* once per batch: the fill plans of the band Cholesky */
        hipLaunchKernelGGL(k_ba_ph_fillplan, dim3(nwin), dim3(256), 0, ks, bufs);
        BCHK(hipGetLastError());
    }
    for (int round = 0; round < P0.n_rounds && !halted && !split; round++) {
        /* one launch per optimize() round; per iteration only when the host has a stop flag to poll in between */
        const int per_launch = any_stop_ptr ? 1 : std::max(P0.rounds[round], 1);
        for (int it = 0; it < P0.rounds[round]; it += per_launch) {
            if (stopped()) { /* g2o polls the force-stop flag between iterations */
                halted = true;
                break;
            }
            if (s->profiling) BCHK(hipEventRecord(J.prof_ev[2 * J.prof_launches], ks));
            int first = it == 0 ? 1 : 0, solver = s->solver, g = G, sq = seq++, nit = std::min(per_launch, P0.rounds[round] - it);
            double tol2 = s->pcg_tol2;
            int dbg = debug;
            /* profiling sessions: launch teams non-cooperatively.  Under rocprofv3 a process that made a cooperative launch
               segfaults at exit inside the HIP runtime's code-object unregistration (failure A of
               profiles/r03_profiler_exit_hang.md).  It is NOT the cause of the profiled phase-kernel runs that hung in
               round 2 -- those made no cooperative launch; that was a race inside the profiler's counter-collection worker
               at exit (failure B, same file), handled by draining before exit. */
            static const bool no_coop = getenv("ORBSLAM_AMD_BA_NO_COOP") != nullptr;
            if (G > 1 && !no_coop) {
                /* a team needs all its workgroups resident at once: cooperative launch (the runtime checks the grid
                   against the device and never runs two cooperative kernels side by side) */
                void *args[] = {&bufs, &first, &tol2, &pcg_maxit, &dbg, &solver, &dyn_doubles, &g, &sq, &nit, &fused_band};
                BCHK(hipLaunchCooperativeKernel((const void *)k_ba_iteration, dim3(nwin * G), dim3(BA_T), args,
                                                (size_t)dyn_doubles * 8, ks));
            } else if (G > 1) {
                hipLaunchKernelGGL(k_ba_iteration, dim3(nwin * G), dim3(BA_T), (size_t)dyn_doubles * 8, ks, bufs, first, tol2,
                                   pcg_maxit, dbg, solver, dyn_doubles, g, sq, nit, fused_band);
                BCHK(hipGetLastError());
            } else {
                hipLaunchKernelGGL(k_ba_iteration, dim3(nwin), dim3(BA_T), (size_t)dyn_doubles * 8, ks, bufs, first, tol2,
                                   pcg_maxit, dbg, solver, dyn_doubles, 1, sq, nit, fused_band);
                BCHK(hipGetLastError());
            }
            if (s->profiling) BCHK(hipEventRecord(J.prof_ev[2 * J.prof_launches + 1], ks));
            J.prof_launches++;
            /* the stop flag must be observed between iterations, as the reference's optimizer would */
            if (any_stop_ptr) BCHK(hipStreamSynchronize(ks));
        }
        if (halted) break;
        if (round + 1 < P0.n_rounds) {
            if (stopped()) break; /* bDoMore == false */
            hipLaunchKernelGGL(k_ba_classify, dim3(nwin), dim3(BA_T), 0, ks, bufs,
                               P0.drop_robust_after == round ? 1 : 0, 0);
            BCHK(hipGetLastError());
        }
    }
    hipLaunchKernelGGL(k_ba_classify, dim3(nwin), dim3(BA_T), 0, ks, bufs, 0, 1);
    BCHK(hipGetLastError());
    BCHK(hipEventRecord(J.solved, ks));
    /* results: the output region only (poses, points, LM state of every window), plus the edge flags */
    BCHK(hipStreamWaitEvent(s->copy_out, J.solved, 0));
    BCHK(hipMemcpyAsync(J.pD, J.dD, J.nOut * 8, hipMemcpyDeviceToHost, s->copy_out));
    bool want_flags = false;
    for (int i = 0; i < nwin; i++) want_flags |= J.res[i].outlier != nullptr;
    if (J.nB && want_flags) BCHK(hipMemcpyAsync(J.pB, J.dB, J.nB, hipMemcpyDeviceToHost, s->copy_out));
    BCHK(hipEventRecord(J.downloaded, s->copy_out));
    J.t_enqueued = ba_now();
    J.pending = true;
    return ORB_OK;
}

static int ba_finish(BASolver *s, BAJob &J) {
    BCHK(hipEventSynchronize(J.downloaded));
    J.pending = false;
    const double t_done = ba_now();
    const int nwin = J.nwin;
    BAResult *res = J.res;
    const int nt = std::max(1, std::min<int>(nwin / 8, std::min<int>(ba_host_threads(), (int)std::thread::hardware_concurrency())));
    ba_parallel(nt, [&](int t) {
        for (int i = t; i < nwin; i += nt) {
            const BAWin &w = J.wins[i];
            memcpy(res[i].poses, &J.pD[w.poses], (size_t)w.Nc * 7 * 8);
            const int *rank = J.hw[i].rank.data(); /* the device holds the points sorted by first camera */
            for (int pt = 0; pt < w.Np; pt++) memcpy(res[i].points + (size_t)pt * 3, &J.pD[w.pts + (int64_t)rank[pt] * 3], 24);
            if (res[i].outlier)
                for (int k = 0; k < w.E; k++) res[i].outlier[J.hw[i].perm[k]] = (J.pB[w.flags + k] & 1) ? 0 : 1;
            res[i].chi2_initial = J.pD[w.st + ST_CHI0];
            res[i].chi2_final = J.pD[w.st + ST_CHI];
            res[i].iterations = (int)J.pD[w.st + ST_ITERS];
            res[i].lin_solves = (int)J.pD[w.st + ST_SOLVES];
        }
    });
    s->last_team = J.split ? 0 : J.team;
    for (int i = 0; i < nwin; i++)
        if (J.pD[J.wins[i].st + ST_ABORT] != 0.0) {
            b_err = "ba_solve: a workgroup-team barrier timed out (window results are invalid)";
            return ORB_E_HIP;
        }
    if (s->profiling && J.prof_launches > 0) {
        double sum = 0;
        for (int k = 0; k < J.prof_launches; k++) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, J.prof_ev[2 * k], J.prof_ev[2 * k + 1]) == hipSuccess) sum += ms;
        }
        s->last_iter_ms = (float)(sum / J.prof_launches);
        s->last_iter_launches = J.prof_launches;
    }
    if (getenv("ORBSLAM_AMD_BA_PROFILE")) {
        double pcg = 0;
        for (int i = 0; i < nwin; i++) pcg += J.pD[J.wins[i].st + ST_PCG_ITERS];
        fprintf(stderr, "[ba] %d windows x team %d: stage %.2f ms, enqueue %.2f ms, wait %.2f ms, unpack %.2f ms, up %.1f MB (D) + %.1f MB (I), down %.1f MB, scratch %.1f MB, pcg its/window %.0f\n",
                nwin, J.team, (J.t_staged - J.t_start) * 1e3, (J.t_enqueued - J.t_staged) * 1e3, (t_done - J.t_enqueued) * 1e3,
                (ba_now() - t_done) * 1e3, (J.nOut + J.nIn) * 8e-6, J.nI * 4e-6, J.nOut * 8e-6, J.scratchD * 8e-6, pcg / nwin);
        static const char *names[10] = {"chi2", "Hll+W", "Hpp", "Dinv+Y(+lambda0)", "Schur", "g+diag", "Minv", "PCG",
                                        "backsub+update", "trial chi2+decision"};
        fprintf(stderr, "[ba] window 0 phase ms:");
        for (int k = 0; k < 10; k++) fprintf(stderr, " %s %.2f |", names[k], J.pD[J.wins[0].st + ST_T0 + k] * 1e-5);
        fprintf(stderr, "\n[ba] window 0 chunked Schur ms: commit %.2f | fetch issue %.2f | block products %.2f | g + barrier %.2f (chunks %d)\n",
                J.pD[J.wins[0].st + ST_T0 + 10] * 1e-5, J.pD[J.wins[0].st + ST_T0 + 11] * 1e-5, J.pD[J.wins[0].st + ST_T0 + 12] * 1e-5,
                J.pD[J.wins[0].st + ST_T0 + 13] * 1e-5, J.wins[0].nchunks);
        fprintf(stderr, "[ba] window 0 Cholesky ms: assemble %.2f | load diag %.2f | factor diag %.2f | panel %.2f | trailing %.2f | forward %.2f | backward %.2f\n",
                J.pD[J.wins[0].st + ST_T0 + 16] * 1e-5, J.pD[J.wins[0].st + ST_T0 + 17] * 1e-5, J.pD[J.wins[0].st + ST_T0 + 18] * 1e-5,
                J.pD[J.wins[0].st + ST_T0 + 19] * 1e-5, J.pD[J.wins[0].st + ST_T0 + 20] * 1e-5, J.pD[J.wins[0].st + ST_T0 + 21] * 1e-5,
                J.pD[J.wins[0].st + ST_T0 + 22] * 1e-5);
    }
    return ORB_OK;
}

static int ba_check_batch(BASolver *s, const BAProblem *probs, BAResult *res, int nwin) {
    if (!s || !probs || !res || nwin <= 0) return ORB_E_INVALID;
    if (nwin > s->max_batch) return ORB_E_CAPACITY;
    for (int i = 0; i < nwin; i++)
        if (probs[i].stop && *probs[i].stop) return ORB_E_STOPPED;
    return ORB_OK;
}

/* asynchronous pair: ba_submit_batch stages a batch and enqueues its device work, ba_wait returns the oldest
 * submitted batch's results.  Two batches may be in flight; probs / res (and everything they point to) must stay
 * valid until the matching ba_wait.  A batch that carries a stop flag is polled between iterations and therefore
 * runs synchronously inside submit. */
extern "C" int ba_submit_batch(BASolver *s, const BAProblem *probs, BAResult *res, int nwin) {
    int rc = ba_check_batch(s, probs, res, nwin);
    if (rc) return rc;
    if (s->in_flight >= BA_SLOTS) {
        b_err = "ba_submit_batch: every batch slot is in flight, call ba_wait first";
        return ORB_E_BUSY;
    }
    BCHK(hipSetDevice(s->device));
    BAJob &J = s->job[s->next_slot];
    if ((rc = ba_stage(s, J, probs, res, nwin))) return rc;
    if ((rc = ba_enqueue(s, J))) return rc;
    s->next_slot = (s->next_slot + 1) % BA_SLOTS;
    s->in_flight++;
    return ORB_OK;
}

extern "C" int ba_wait(BASolver *s) {
    if (!s) return ORB_E_INVALID;
    if (s->in_flight <= 0) return ORB_OK;
    BAJob &J = s->job[s->oldest_slot]; /* the oldest batch is retired whatever happens below: callers that mirror */
    s->oldest_slot = (s->oldest_slot + 1) % BA_SLOTS; /* the queue (optimizer.py) pop their entry before looking at the status */
    s->in_flight--;
    BCHK(hipSetDevice(s->device));
    return ba_finish(s, J);
}

extern "C" int ba_solve_batch(BASolver *s, const BAProblem *probs, BAResult *res, int nwin) {
    int rc = ba_check_batch(s, probs, res, nwin);
    if (rc) return rc;
    while (s->in_flight > 0)
        if ((rc = ba_wait(s))) return rc;
    s->next_slot = s->oldest_slot = 0; /* the synchronous path always works in slot 0: one set of buffers */
    if ((rc = ba_submit_batch(s, probs, res, nwin))) return rc;
    return ba_wait(s);
}

extern "C" int ba_solve(BASolver *s, const BAProblem *p, BAResult *r) { return ba_solve_batch(s, p, r, 1); }


/* ================================================================================================
 * Optimizer::PoseOptimization (reference src/Optimizer.cpp:131-261): one free SE3 vertex, n unary
 * EdgeSE3ProjectXYZOnlyPose edges with fixed Xw, Huber delta = 5.99 [sic] (:169), 4 x optimize(10)
 * with chi2 > 5.991 relabelling (:213-247), robust kernel removed after the round with it == 2.
 * The whole schedule runs in ONE launch of ONE 256-thread workgroup: lanes stride over the edges,
 * the 6x6 system is reduced across the workgroup and solved by thread 0.
 * ================================================================================================ */
#define PO_T 256
struct POShared {
    double red[PO_T / 64][28];
    double pose[7], trial[7];
    double bc[4];
    double cand_x[PO_T / 64][6], cand_t[PO_T / 64][7]; /* the next PO_T / 64 trial steps and poses, one per wavefront */
    int cand_ok[PO_T / 64];
};

__device__ __forceinline__ void po_reduce(double *v, int nv, POShared &s, double *out) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); /* uniform: scalar unit */
    for (int i = 0; i < nv; i++) {
        double t = v[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        v[i] = t;
    }
    __syncthreads();
    if (lane == 0)
        for (int i = 0; i < nv; i++) s.red[wave][i] = v[i];
    __syncthreads();
    for (int i = 0; i < nv; i++) {
        double t = 0;
        for (int w = 0; w < PO_T / 64; w++) t += s.red[w][i];
        out[i] = t;
    }
}

/* Sum of 28 per-thread partials (the 21 + 6 entries of H and b, and the active-edge count) over the workgroup by an
 * LDS transpose: every thread stores its 28 values, 224 threads each add 32 of them (8 threads per value,
 * interleaved slices), three shuffle steps finish.  About 0.4 us against 2.8 us for 28 six-step shuffle butterflies
 * (each f64 step is two LDS-crossbar permutes). */
#define PO_TR_LD (PO_T + 2)
__device__ __forceinline__ void po_reduce28(const double *v, double (*tr)[PO_TR_LD], POShared &s, double *out) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 28; i++) tr[i][tid] = v[i];
    __syncthreads();
    if (tid < 28 * 8) {
        const int val = tid >> 3, seg = tid & 7;
        double t = 0;
#pragma unroll 8
        for (int j = 0; j < PO_T / 8; j++) t += tr[val][j * 8 + seg];
        t += __shfl_xor(t, 1);
        t += __shfl_xor(t, 2);
        t += __shfl_xor(t, 4);
        if (seg == 0) s.red[0][val] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 28; i++) out[i] = s.red[0][i];
}

__device__ __forceinline__ double po_chi2(const double *pose, const double *xw, const double *obs, const double *info, const uint8_t *fl,
                          int n, double fx, double fy, double cx, double cy, double delta, POShared &s) {
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += PO_T) {
        if (!(fl[i] & 1)) continue;
        double e[2], pc[3];
        project_err(pose, xw + (int64_t)i * 3, obs + (int64_t)i * 2, fx, fy, cx, cy, e, pc);
        double c = (e[0] * e[0] + e[1] * e[1]) * info[i];
        if (fl[i] & 2) c = huber_rho0(c, delta);
        acc += c;
    }
    double out;
    po_reduce(&acc, 1, s, &out);
    return out;
}

/* (H + lambda I) x = b by Cholesky, 6 x 6, every loop unrolled and no early exit so that the factor lives in
 * registers (with a `break` in the column loop the compiler kept A[] in scratch memory: 10 us per solve, two
 * thirds of the whole pose optimisation).  Same operations in the same order as the plain loops. */
__device__ __forceinline__ bool po_chol6_solve(const double *H, double lambda, const double *b, double *x) {
    /* No f64 square root and no f64 division (six and twenty-seven of them stood here: each expands to a few dozen
       dependent half-rate instructions, and every one of the 40 iterations of a pose problem waits for this solve): a
       column is scaled by r = 1 / sqrt(d) from v_rsq_f64 (about 2^-23) and one cubically convergent step
       r (1 + e / 2 + 3 e^2 / 8), e = 1 - d r^2 (relative error below 2^-60), the pivot is d r, and the substitutions
       multiply by the same r. */
    double A[36], rinv[6];
#pragma unroll
    for (int i = 0; i < 36; i++) A[i] = H[i];
#pragma unroll
    for (int j = 0; j < 6; j++) A[j * 7] += lambda;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[j * 6 + j];
#pragma unroll
        for (int kk = 0; kk < j; kk++) d -= A[j * 6 + kk] * A[j * 6 + kk];
        if (!(d > 0) || !isfinite(d)) {
            ok = false;
            d = 1.0; /* keep going with harmless values; the caller discards the result */
        }
        double r = __builtin_amdgcn_rsq(d);
        {
            const double e = __builtin_fma(-(d * r), r, 1.0);
            r = __builtin_fma(r, __builtin_fma(0.375, e, 0.5) * e, r);
        }
        rinv[j] = r;
        A[j * 6 + j] = d * r;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double v = A[i * 6 + j];
#pragma unroll
            for (int kk = 0; kk < j; kk++) v -= A[i * 6 + kk] * A[j * 6 + kk];
            A[i * 6 + j] = v * r;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = b[i];
#pragma unroll
        for (int kk = 0; kk < i; kk++) v -= A[i * 6 + kk] * x[kk];
        x[i] = v * rinv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = x[i];
#pragma unroll
        for (int kk = i + 1; kk < 6; kk++) v -= A[kk * 6 + i] * x[kk];
        x[i] = v * rinv[i];
    }
    return ok;
}

__global__ __launch_bounds__(PO_T) void k_pose_opt(double *pose7_all, const double *xw_all, const double *obs_all,
                                                    const double *info_all, uint8_t *fl_all, uint8_t *outlier_all,
                                                    const int32_t *offs, double fx, double fy, double cx, double cy,
                                                    int32_t *n_inliers_all, int lds_edges, int timing) {
    /* one workgroup per pose problem: problem p owns edges [offs[p], offs[p+1]) */
    __shared__ POShared s;
    __shared__ double po_tr[28][PO_TR_LD];
    extern __shared__ __attribute__((aligned(16))) double po_dyn[];
    const int tid = threadIdx.x;
    const int e0 = offs[blockIdx.x], n = offs[blockIdx.x + 1] - e0;
    double *pose7 = pose7_all + (int64_t)blockIdx.x * 7;
    const double *xw = xw_all + (int64_t)e0 * 3, *obs = obs_all + (int64_t)e0 * 2, *info = info_all + e0;
    uint8_t *fl = fl_all + e0, *outlier = outlier_all + e0;
    int32_t *n_inliers = n_inliers_all + blockIdx.x;
    if (n < 3) { /* src/Optimizer.cpp:206-207: returns 0, pose untouched */
        if (tid == 0) *n_inliers = 0;
        if (tid < n) outlier[tid] = 0;
        return;
    }
    const double delta = 5.99;
    /* the pose lives in registers, identical in every thread (all of them run the same deterministic update), so
       neither the exponential map nor an accepted step needs LDS broadcasts and their barriers */
    double pose[7], trial[7];
#pragma unroll
    for (int i = 0; i < 7; i++) pose[i] = pose7[i];
    if (n <= lds_edges) {
        /* the 40 iterations make ~120 passes over the same few hundred edges, each a dependent trip to L2/HBM per
           loop step: stage the correspondences (48 B each) and their flags in LDS once -- same arithmetic, same
           order, a quarter of the time */
        double *lx = po_dyn, *lo = lx + (size_t)lds_edges * 3, *li = lo + (size_t)lds_edges * 2;
        uint8_t *lf = reinterpret_cast<uint8_t *>(li + lds_edges);
        for (int i = tid; i < n * 3; i += PO_T) lx[i] = xw[i];
        for (int i = tid; i < n * 2; i += PO_T) lo[i] = obs[i];
        for (int i = tid; i < n; i += PO_T) li[i] = info[i];
        xw = lx;
        obs = lo;
        info = li;
        fl = lf;
    }
    for (int i = tid; i < n; i += PO_T) {
        fl[i] = 3; /* active + robust */
        outlier[i] = 0;
    }
    __syncthreads();
    quat_normalize(pose);
    /* (timing != 0: thread 0 times the sections of an LM iteration -- s_memrealtime ticks of 10 ns -- and prints them) */
    unsigned long long tJ = 0, tR = 0, tS = 0, tC = 0, tq0 = 0, tq1;
    int nits = 0;
#define PO_TM(acc)                                   \
    if (timing) {                                    \
        tq1 = __builtin_amdgcn_s_memrealtime();      \
        acc += tq1 - tq0;                            \
        tq0 = tq1;                                   \
    }
    int nBads = 0;
    double lambda = 0, ni = 2;
    for (int it = 0; it < 4; it++) {
        bool ok = true;
        bool have_chi = false; /* the relabelling below changes the active set and the kernels */
        double currentChi = 0;
        for (int k = 0; k < 10 && ok; k++) {
            /* the chi2 of the current pose is the trial chi2 of the step that was just accepted (same pose, same
               flags, same summation order): only the first step of a round has to evaluate it */
            if (!have_chi) currentChi = po_chi2(pose, xw, obs, info, fl, n, fx, fy, cx, cy, delta, s);
            have_chi = true;
            double acc[28];
#pragma unroll
            for (int i = 0; i < 28; i++) acc[i] = 0;
            if (timing) tq0 = __builtin_amdgcn_s_memrealtime();
            nits++;
            for (int i = tid; i < n; i += PO_T) {
                if (!(fl[i] & 1)) continue;
                double e[2], pc[3], Ji[6], Jj[12];
                project_err(pose, xw + (int64_t)i * 3, obs + (int64_t)i * 2, fx, fy, cx, cy, e, pc);
                jacobians(pose, pc, fx, fy, Ji, Jj);
                double wgt = info[i], r0 = -wgt * e[0], r1 = -wgt * e[1];
                if (fl[i] & 2) {
                    const double rw = huber_w((e[0] * e[0] + e[1] * e[1]) * info[i], delta);
                    wgt *= rw;
                    r0 *= rw;
                    r1 *= rw;
                }
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b = a; b < 6; b++) acc[q++] += wgt * (Jj[a] * Jj[b] + Jj[6 + a] * Jj[6 + b]);
#pragma unroll
                for (int a = 0; a < 6; a++) acc[21 + a] += Jj[a] * r0 + Jj[6 + a] * r1;
                acc[27] += 1.0;
            }
            double tot[28];
            PO_TM(tJ)
            po_reduce28(acc, po_tr, s, tot);
            PO_TM(tR)
            if (tot[27] == 0.0) break;
            double H[36], b[6];
            {
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int c = a; c < 6; c++) {
                        H[a * 6 + c] = tot[q];
                        H[c * 6 + a] = tot[q];
                        q++;
                    }
#pragma unroll
                for (int a = 0; a < 6; a++) b[a] = tot[21 + a];
            }
            if (k == 0) {
                double m = 0;
#pragma unroll
                for (int j = 0; j < 6; j++) m = fmax(m, fabs(H[j * 7]));
                lambda = 1e-5 * m;
                ni = 2;
            }
            double rho = 0;
            int qmax = 0;
            do {
                /* The trials of one iteration use lambda, lambda * ni, lambda * ni * 2 ni, ... -- known before the first one is
                   evaluated -- and a rejected trial near convergence is the common case.  The 6 x 6 solve and the exponential
                   map are a chain of ~300 dependent f64 operations that every wavefront used to run redundantly: now wavefront
                   w runs it for the trial w steps ahead, so four trials cost one chain (same operations on the same values:
                   the products lambda * ni * ... are formed in the sequential order). */
                constexpr int PO_NW = PO_T / 64;
                if ((qmax & (PO_NW - 1)) == 0) {
                    const int wv = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
                    double lam_w = lambda, ni_w = ni;
                    for (int m = 0; m < wv; m++) {
                        lam_w *= ni_w;
                        ni_w *= 2;
                    }
                    double xc[6], tc[7];
                    const bool okc = po_chol6_solve(H, lam_w, b, xc);
                    if (okc) {
                        se3_exp_mul_fast(xc, pose, tc);
                    } else {
#pragma unroll
                        for (int i = 0; i < 7; i++) tc[i] = pose[i];
                    }
                    if ((tid & 63) == 0) {
#pragma unroll
                        for (int i = 0; i < 6; i++) s.cand_x[wv][i] = xc[i];
#pragma unroll
                        for (int i = 0; i < 7; i++) s.cand_t[wv][i] = tc[i];
                        s.cand_ok[wv] = okc ? 1 : 0;
                    }
                    __syncthreads(); /* (the previous group's values were last read before the barriers of its trial's chi2 pass) */
                }
                double x[6];
                const int cq = qmax & (PO_NW - 1);
                const bool ok2 = s.cand_ok[cq] != 0;
#pragma unroll
                for (int i = 0; i < 6; i++) x[i] = s.cand_x[cq][i];
#pragma unroll
                for (int i = 0; i < 7; i++) trial[i] = s.cand_t[cq][i];
                PO_TM(tS)
                double tempChi = po_chi2(trial, xw, obs, info, fl, n, fx, fy, cx, cy, delta, s);
                PO_TM(tC)
                if (!ok2) tempChi = 1.7976931348623157e308;
                double scale = 0;
                if (ok2)
#pragma unroll
                    for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho = (currentChi - tempChi) / scale;
                if (rho > 0 && isfinite(tempChi)) {
                    const double t3 = 2 * rho - 1;
                    double alpha = 1. - t3 * t3 * t3; /* (pow(t, 3) is a few hundred instructions on the chain of every iteration) */
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
#pragma unroll
                    for (int i = 0; i < 7; i++) pose[i] = trial[i];
                } else {
                    lambda *= ni;
                    ni *= 2;
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) ok = false;
        }
        /* relabel (:218-244) */
        double bad = 0;
        for (int i = tid; i < n; i += PO_T) {
            double e[2], pc[3];
            project_err(pose, xw + (int64_t)i * 3, obs + (int64_t)i * 2, fx, fy, cx, cy, e, pc);
            const double c = (e[0] * e[0] + e[1] * e[1]) * info[i];
            uint8_t f = fl[i];
            if (c > 5.991) {
                f &= ~1;
                outlier[i] = 1;
                bad += 1.0;
            } else {
                f |= 1;
                outlier[i] = 0;
            }
            if (it == 2) f &= ~2;
            fl[i] = f;
        }
        double tb;
        po_reduce(&bad, 1, s, &tb);
        nBads = (int)tb;
        __syncthreads();
        if (n < 10) break;
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 7; i++) pose7[i] = pose[i];
    }
    if (tid == 0) *n_inliers = n - nBads;
    if (timing && tid == 0 && blockIdx.x == 0)
        printf("[pose opt] n %d, %d iterations, us per iteration: jacobians + H %.2f | reduce28 %.2f | solve + exp map %.2f | trial chi2 %.2f\n", n,
               nits, tJ * 0.01 / nits, tR * 0.01 / nits, tS * 0.01 / nits, tC * 0.01 / nits);
}

/* n_problems independent pose problems; problem p owns edges [offsets[p], offsets[p+1]) of xw/obs/inv_sigma2/
 * outlier; poses[7*n_problems] in/out; n_inliers[n_problems] out.  One launch, one workgroup per problem. */
extern "C" int ba_pose_optimization_batch(BASolver *s, double *poses, const double *xw, const double *obs,
                                          const double *inv_sigma2, const int32_t *offsets, int n_problems, double fx,
                                          double fy, double cx, double cy, uint8_t *outlier, int32_t *n_inliers) {
    if (!s || !poses || !offsets || !n_inliers || n_problems <= 0) return ORB_E_INVALID;
    const int n = offsets[n_problems];
    if (n < 0 || (n > 0 && (!xw || !obs || !inv_sigma2 || !outlier))) return ORB_E_INVALID;
    for (int p = 0; p < n_problems; p++)
        if (offsets[p + 1] < offsets[p]) return ORB_E_INVALID;
    BCHK(hipSetDevice(s->device));
    int rc;
    /* ONE device block and its pinned mirror, laid out so that the inputs go up in one copy and the outputs come back in one
       (six small copies cost ~8 us of host time each, and Tracking calls this up to three times per frame):
         [ xw 3n | obs 2n | info n | offsets (int32) | poses 7p ]   <- upload
                                               [ poses 7p | n_inliers (int32) | outlier (bytes) ]   <- download */
    const size_t npose = ((size_t)n_problems * 7 + 1) & ~(size_t)1, noffs = ((size_t)n_problems + 2) / 2, nninl = ((size_t)n_problems + 1) / 2,
                 nout = ((size_t)n + 7) / 8;
    const size_t o_offs = (size_t)n * 6, o_pose = o_offs + noffs, o_ninl = o_pose + npose, o_out = o_ninl + nninl, nd = o_out + nout + 2;
    if ((rc = ba_grow(s->poD, s->pocapD, nd))) return rc;
    if ((rc = ba_grow(s->poB, s->pocapB, (size_t)n + 16))) return rc;
    double *d_xw = s->poD, *d_obs = d_xw + (size_t)n * 3, *d_info = d_obs + (size_t)n * 2, *d_pose = s->poD + o_pose;
    int32_t *d_offs = (int32_t *)(s->poD + o_offs), *d_ninl = (int32_t *)(s->poD + o_ninl);
    uint8_t *d_fl = s->poB, *d_out = (uint8_t *)(s->poD + o_out);
    if ((rc = ba_grow_pinned(s->ppD, s->ppcapD, nd))) return rc;
    if (n > 0) {
        memcpy(s->ppD, xw, (size_t)n * 24);
        memcpy(s->ppD + (size_t)n * 3, obs, (size_t)n * 16);
        memcpy(s->ppD + (size_t)n * 5, inv_sigma2, (size_t)n * 8);
    }
    memcpy(s->ppD + o_offs, offsets, (size_t)(n_problems + 1) * 4);
    memcpy(s->ppD + o_pose, poses, (size_t)n_problems * 56);
    BCHK(hipMemcpyAsync(s->poD, s->ppD, (o_pose + npose) * 8, hipMemcpyHostToDevice, s->stream));
    int maxe = 0;
    for (int p = 0; p < n_problems; p++) maxe = std::max(maxe, offsets[p + 1] - offsets[p]);
    int lds_edges = 0;
    const size_t po_lds = ((size_t)maxe * 49 + 15) & ~(size_t)15; /* 6 doubles + 1 flag byte per correspondence */
    if (maxe > 0 && po_lds <= 96 * 1024) { /* beside the 58 KB of static LDS of the kernel */
        lds_edges = maxe;
        static PerDeviceLdsLimit po_lim;
        BCHK(po_lim.ensure((const void *)k_pose_opt, po_lds));
    }
    hipLaunchKernelGGL(k_pose_opt, dim3(n_problems), dim3(PO_T), lds_edges ? po_lds : 0, s->stream, d_pose, (const double *)d_xw,
                       (const double *)d_obs, (const double *)d_info, d_fl, d_out, (const int32_t *)d_offs, fx, fy, cx, cy,
                       d_ninl, lds_edges, getenv("ORBSLAM_AMD_PO_TIMING") ? 1 : 0);
    BCHK(hipGetLastError());
    BCHK(hipMemcpyAsync(s->ppD + o_pose, d_pose, (o_out + nout - o_pose) * 8, hipMemcpyDeviceToHost, s->stream));
    BCHK(hipStreamSynchronize(s->stream));
    memcpy(poses, s->ppD + o_pose, (size_t)n_problems * 56);
    if (n > 0) memcpy(outlier, s->ppD + o_out, (size_t)n);
    memcpy(n_inliers, s->ppD + o_ninl, (size_t)n_problems * 4);
    return ORB_OK;
}

/* ---- PoseOptimization of the correspondences a device-resident SearchByProjection produced (round 3) -------------------
 * The pose problem of Optimizer::PoseOptimization(Frame*) (src/Optimizer.cpp:131-261) built ON THE DEVICE: the current
 * frame's features j with cur_match[j] >= 0, in index order (the order Frame's slots are walked in, :161-200), each with the
 * world point of the matched last-frame feature, its keypoint position as the measurement and 1 / sigma^2 of its octave.
 * k_po_gather compacts them (order preserving, ballot prefix) into the solver's pose buffers, k_pose_opt runs unchanged,
 * k_po_scatter spreads the outlier labels back over the features.  Device pointers in and out, enqueued on `stream`,
 * no copy, no wait: the last link of extract -> SearchByProjection -> PoseOptimization without a host round trip. */
struct POLevels {
    double inv_sigma2[64];
};
__global__ __launch_bounds__(1024) void k_po_gather(const OrbKeyPoint *kps, const int32_t *n_cur_dev, int cap, const int32_t *match,
                                                     const double *last_xw, POLevels lv, int nlevels, double *xw, double *obs,
                                                     double *info, int32_t *idx, int32_t *offs, uint8_t *outlier_full, int32_t *n_corr) {
    __shared__ int s_wave[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(*n_cur_dev, cap);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 1024) {
        const int j = j0 + tid;
        const int mi = j < n ? match[j] : -1;
        if (j < n) outlier_full[j] = 0; /* (k_po_scatter2 sets the labels of the features that have a correspondence) */
        const bool has = mi >= 0;
        const uint64_t bal = __ballot(has);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int before = s_base;
        for (int w2 = 0; w2 < wave; w2++) before += s_wave[w2];
        if (has) {
            const int slot = before + __popcll(bal & ((1ull << lane) - 1ull));
            const OrbKeyPoint kp = kps[j];
            xw[3 * slot] = last_xw[3 * (int64_t)mi];
            xw[3 * slot + 1] = last_xw[3 * (int64_t)mi + 1];
            xw[3 * slot + 2] = last_xw[3 * (int64_t)mi + 2];
            obs[2 * slot] = (double)kp.x;
            obs[2 * slot + 1] = (double)kp.y;
            info[slot] = lv.inv_sigma2[(kp.octave >= 0 && kp.octave < nlevels) ? kp.octave : 0];
            idx[slot] = j;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w2 = 0; w2 < 16; w2++) tot += s_wave[w2];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        offs[0] = 0;
        offs[1] = s_base;
        if (n_corr) *n_corr = s_base;
    }
}
__global__ void k_po_scatter2(const int32_t *offs, const int32_t *idx, const uint8_t *out_compact, uint8_t *outlier_full) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < offs[1]) outlier_full[idx[i]] = out_compact[i];
}

extern "C" int ba_pose_optimization_matches_device(BASolver *s, const OrbKeyPoint *d_cur_kps, int cap_cur, const int32_t *d_n_cur,
                                                   const int32_t *d_cur_match, const double *d_last_xw, double *d_pose7,
                                                   const double *inv_sigma2_levels, int nlevels, double fx, double fy, double cx,
                                                   double cy, uint8_t *d_outlier, int32_t *d_n_inliers, int32_t *d_n_corr,
                                                   void *stream) {
    if (!s || !d_cur_kps || cap_cur <= 0 || !d_n_cur || !d_cur_match || !d_last_xw || !d_pose7 || !inv_sigma2_levels || nlevels <= 0 ||
        nlevels > 64 || !d_outlier || !d_n_inliers)
        return ORB_E_INVALID;
    BCHK(hipSetDevice(s->device));
    hipStream_t st = stream ? (hipStream_t)stream : s->stream;
    int rc;
    const size_t n = (size_t)cap_cur;
    /* scratch in the solver's pose buffers, sized for the slab capacity once (growing frees the old block: only while idle) */
    if (n * 6 + 8 > s->pocapD || n * 2 + 16 > s->pocapB || n + 8 > s->pocapI) {
        BCHK(hipStreamSynchronize(st));
        BCHK(hipStreamSynchronize(s->stream));
        if ((rc = ba_grow(s->poD, s->pocapD, n * 6 + 8))) return rc;
        if ((rc = ba_grow(s->poB, s->pocapB, n * 2 + 16))) return rc;
        if ((rc = ba_grow(s->poI, s->pocapI, n + 8))) return rc;
    }
    double *d_xw = s->poD, *d_obs = d_xw + n * 3, *d_info = d_obs + n * 2;
    uint8_t *d_fl = s->poB, *d_out = s->poB + n + 8;
    int32_t *d_offs = s->poI, *d_idx = s->poI + 4;
    POLevels lv;
    for (int l = 0; l < 64; l++) lv.inv_sigma2[l] = l < nlevels ? inv_sigma2_levels[l] : 0.0;
    hipLaunchKernelGGL(k_po_gather, dim3(1), dim3(1024), 0, st, d_cur_kps, d_n_cur, cap_cur, d_cur_match, d_last_xw, lv, nlevels, d_xw, d_obs,
                       d_info, d_idx, d_offs, d_outlier, d_n_corr);
    int lds_edges = 0;
    const size_t po_lds = (n * 49 + 15) & ~(size_t)15;
    if (po_lds <= 96 * 1024) {
        lds_edges = cap_cur;
        static PerDeviceLdsLimit po_lim2;
        BCHK(po_lim2.ensure((const void *)k_pose_opt, po_lds));
    }
    hipLaunchKernelGGL(k_pose_opt, dim3(1), dim3(PO_T), lds_edges ? po_lds : 0, st, d_pose7, (const double *)d_xw, (const double *)d_obs,
                       (const double *)d_info, d_fl, d_out, (const int32_t *)d_offs, fx, fy, cx, cy, d_n_inliers, lds_edges, 0);
    hipLaunchKernelGGL(k_po_scatter2, dim3((cap_cur + 255) / 256), dim3(256), 0, st, (const int32_t *)d_offs, (const int32_t *)d_idx,
                       (const uint8_t *)d_out, d_outlier);
    BCHK(hipGetLastError());
    return ORB_OK;
}

extern "C" int ba_pose_optimization(BASolver *s, double *pose, const double *xw, const double *obs, const double *inv_sigma2,
                                    int n, double fx, double fy, double cx, double cy, uint8_t *outlier, int32_t *n_inliers) {
    if (!s || !pose || !n_inliers || n < 0 || (n > 0 && (!xw || !obs || !inv_sigma2 || !outlier))) return ORB_E_INVALID;
    const int32_t offs[2] = {0, n};
    return ba_pose_optimization_batch(s, pose, xw, obs, inv_sigma2, offs, 1, fx, fy, cx, cy, outlier, n_inliers);
}
