/*
 * ba_oracle.cpp -- CPU ORACLE for Optimizer::BundleAdjustment / LocalBundleAdjustment /
 * PoseOptimization (test infrastructure only).
 *
 * The reference builds a g2o graph (src/Optimizer.cpp:23-129, :131-261) and all arithmetic happens
 * inside g2o, which is NOT under /root/reference and not installed.  This file restates
 * the published algorithm of the ORB-SLAM2 fork of g2o that the reference includes
 * (Thirdparty/g2o: OptimizationAlgorithmLevenberg, BlockSolver_6_3 with Schur complement,
 * EdgeSE3ProjectXYZ / EdgeSE3ProjectXYZOnlyPose, VertexSE3Expmap, RobustKernelHuber), version
 * unpinned: **parity unpinned** -- there is no reference test or golden vector for this path.
 * The linear solve is a dense Cholesky of the reduced camera system (what LinearSolverEigen's sparse
 * Cholesky computes).  tests/ additionally checks the converged optimum against an independent
 * scipy.optimize.least_squares run.
 *
 * Graph construction choices follow src/Optimizer.cpp: pose estimate = SE3Quat(Rwc, twc) used as
 * world->camera (:48), KF id 0 fixed (:52-53), points marginalised (:74), Huber delta sqrt(5.99)
 * (:61), information = I * invSigma2 (:96-97; uninitialised in the reference -> 1/sigma^2(octave),
 * SURVEY 8c-ii).
 */
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Quat {
    double w, x, y, z;
};
struct SE3 {
    Quat q;
    double t[3];
};

inline void quat_normalize(Quat &q) { /* SE3Quat::normalizeRotation */
    if (q.w < 0) {
        q.w = -q.w;
        q.x = -q.x;
        q.y = -q.y;
        q.z = -q.z;
    }
    double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    q.w /= n;
    q.x /= n;
    q.y /= n;
    q.z /= n;
}
inline Quat quat_mul(const Quat &a, const Quat &b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
inline void quat_to_R(const Quat &q, double R[9]) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
/* Eigen::Quaterniond(Matrix3d) */
inline Quat R_to_quat(const double R[9]) {
    Quat q;
    double t = R[0] + R[4] + R[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (R[7] - R[5]) * t;
        q.y = (R[2] - R[6]) * t;
        q.z = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (R[k * 3 + j] - R[j * 3 + k]) * t;
        v[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        v[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
        q.x = v[0];
        q.y = v[1];
        q.z = v[2];
    }
    return q;
}
inline void se3_map(const SE3 &T, const double R[9], const double X[3], double out[3]) {
    out[0] = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + T.t[0];
    out[1] = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + T.t[1];
    out[2] = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + T.t[2];
}
/* SE3Quat::exp(update) * T  (VertexSE3Expmap::oplusImpl); update = [omega(3), upsilon(3)] */
inline SE3 se3_exp_mul(const double u[6], const SE3 &T) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = std::sqrt(wx * wx + wy * wy + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i];
        for (int i = 0; i < 9; i++) V[i] = R[i];
    } else {
        const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta),
                     c = (theta - std::sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; i++) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i];
        }
    }
    SE3 E;
    E.q = R_to_quat(R);
    quat_normalize(E.q);
    for (int i = 0; i < 3; i++) E.t[i] = V[i * 3] * u[3] + V[i * 3 + 1] * u[4] + V[i * 3 + 2] * u[5];
    /* E * T */
    SE3 out;
    double RE[9];
    quat_to_R(E.q, RE);
    for (int i = 0; i < 3; i++) out.t[i] = E.t[i] + RE[i * 3] * T.t[0] + RE[i * 3 + 1] * T.t[1] + RE[i * 3 + 2] * T.t[2];
    out.q = quat_mul(E.q, T.q);
    quat_normalize(out.q);
    return out;
}

inline void huber(double e, double delta, double rho[3]) { /* RobustKernelHuber::robustify */
    const double dsqr = delta * delta;
    if (e <= dsqr) {
        rho[0] = e;
        rho[1] = 1.;
        rho[2] = 0.;
    } else {
        const double sqrte = std::sqrt(e);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e;
    }
}

/* dense Cholesky solve of A x = b (A symmetric n x n, row-major, destroyed); false if not PD */
bool chol_solve(std::vector<double> &A, std::vector<double> &b, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (!(d > 0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            const double *ri = &A[(size_t)i * n], *rj = &A[(size_t)j * n];
            for (int k = 0; k < j; k++) s -= ri[k] * rj[k];
            A[(size_t)i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k];
        b[i] = s / A[(size_t)i * n + i];
    }
    return true;
}

struct Edge {
    int cam, pt;
    double obs[2], info;
    bool robust;
    int level; /* 0 active, 1 switched off */
    double err[2];
};

struct BA {
    int Nc, Np;
    std::vector<SE3> poses;
    std::vector<uint8_t> fixed;
    std::vector<double> pts; /* Np x 3 */
    std::vector<Edge> edges;
    double fx, fy, cx, cy, delta;
    /* active sets / indexing */
    std::vector<int> camSlot, ptSlot; /* -1 = not in the system */
    int nCamFree = 0, nPtAct = 0;
    /* LM state (OptimizationAlgorithmLevenberg) */
    double lambda = 0, ni = 2;
    int linSolves = 0;

    void computeError(Edge &e) const {
        double R[9], pc[3];
        quat_to_R(poses[e.cam].q, R);
        se3_map(poses[e.cam], R, &pts[(size_t)e.pt * 3], pc);
        e.err[0] = e.obs[0] - (pc[0] / pc[2] * fx + cx);
        e.err[1] = e.obs[1] - (pc[1] / pc[2] * fy + cy);
    }
    double chi2(const Edge &e) const { return (e.err[0] * e.err[0] + e.err[1] * e.err[1]) * e.info; }
    bool depthPositive(const Edge &e) const {
        double R[9], pc[3];
        quat_to_R(poses[e.cam].q, R);
        se3_map(poses[e.cam], R, &pts[(size_t)e.pt * 3], pc);
        return pc[2] > 0.0;
    }
    void computeActiveErrors() {
        for (auto &e : edges)
            if (e.level == 0) computeError(e);
    }
    double activeRobustChi2() const {
        double s = 0;
        for (const auto &e : edges)
            if (e.level == 0) {
                double c = chi2(e);
                if (e.robust) {
                    double rho[3];
                    huber(c, delta, rho);
                    c = rho[0];
                }
                s += c;
            }
        return s;
    }
    /* SparseOptimizer::initializeOptimization(level 0): active vertices = those touched by active edges */
    void initializeOptimization() {
        camSlot.assign(Nc, -1);
        ptSlot.assign(Np, -1);
        std::vector<char> cu(Nc, 0), pu(Np, 0);
        for (const auto &e : edges)
            if (e.level == 0) {
                cu[e.cam] = 1;
                pu[e.pt] = 1;
            }
        nCamFree = 0;
        nPtAct = 0;
        for (int c = 0; c < Nc; c++)
            if (cu[c] && !fixed[c]) camSlot[c] = nCamFree++;
        for (int p = 0; p < Np; p++)
            if (pu[p]) ptSlot[p] = nPtAct++;
    }

    /* system at the current linearisation point */
    std::vector<double> Hpp, Hll, bp, bl; /* Hpp: nCamFree x 36 ; Hll: nPtAct x 9 */
    std::vector<double> W;                /* per edge 6x3 (pose rows, point cols), zero if cam fixed */

    void buildSystem() {
        Hpp.assign((size_t)nCamFree * 36, 0.0);
        Hll.assign((size_t)nPtAct * 9, 0.0);
        bp.assign((size_t)nCamFree * 6, 0.0);
        bl.assign((size_t)nPtAct * 3, 0.0);
        W.assign(edges.size() * 18, 0.0);
        for (size_t ei = 0; ei < edges.size(); ei++) {
            Edge &e = edges[ei];
            if (e.level != 0) continue;
            double R[9], pc[3];
            quat_to_R(poses[e.cam].q, R);
            se3_map(poses[e.cam], R, &pts[(size_t)e.pt * 3], pc);
            const double x = pc[0], y = pc[1], z = pc[2], z_2 = z * z;
            /* EdgeSE3ProjectXYZ::linearizeOplus */
            double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
            double Ji[6]; /* 2x3 d e / d point */
            for (int r = 0; r < 2; r++)
                for (int c = 0; c < 3; c++)
                    Ji[r * 3 + c] = -1. / z * (tmp[r * 3] * R[c] + tmp[r * 3 + 1] * R[3 + c] + tmp[r * 3 + 2] * R[6 + c]);
            double Jj[12]; /* 2x6 d e / d pose */
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
            /* BaseBinaryEdge::constructQuadraticForm */
            double w = e.info;
            double er[2] = {-e.info * e.err[0], -e.info * e.err[1]}; /* omega_r = -omega * error */
            if (e.robust) {
                double rho[3];
                huber(chi2(e), delta, rho);
                w = rho[1] * e.info;
                er[0] *= rho[1];
                er[1] *= rho[1];
            }
            const int ps = ptSlot[e.pt], cs = camSlot[e.cam];
            for (int a = 0; a < 3; a++) {
                bl[(size_t)ps * 3 + a] += Ji[a] * er[0] + Ji[3 + a] * er[1];
                for (int b = 0; b < 3; b++) Hll[(size_t)ps * 9 + a * 3 + b] += w * (Ji[a] * Ji[b] + Ji[3 + a] * Ji[3 + b]);
            }
            if (cs >= 0) {
                for (int a = 0; a < 6; a++) {
                    bp[(size_t)cs * 6 + a] += Jj[a] * er[0] + Jj[6 + a] * er[1];
                    for (int b = 0; b < 6; b++)
                        Hpp[(size_t)cs * 36 + a * 6 + b] += w * (Jj[a] * Jj[b] + Jj[6 + a] * Jj[6 + b]);
                    for (int b = 0; b < 3; b++) W[ei * 18 + a * 3 + b] = w * (Jj[a] * Ji[b] + Jj[6 + a] * Ji[3 + b]);
                }
            }
        }
    }
    double computeLambdaInit() const { /* tau * max |diag| over all free vertices */
        double m = 0;
        for (int c = 0; c < nCamFree; c++)
            for (int j = 0; j < 6; j++) m = std::max(m, std::fabs(Hpp[(size_t)c * 36 + j * 7]));
        for (int p = 0; p < nPtAct; p++)
            for (int j = 0; j < 3; j++) m = std::max(m, std::fabs(Hll[(size_t)p * 9 + j * 4]));
        return 1e-5 * m;
    }
    /* BlockSolver::solve with Schur complement; returns false if the reduced system is not PD */
    bool solve(double lam, std::vector<double> &xp, std::vector<double> &xl) {
        const int n = nCamFree * 6;
        std::vector<double> S((size_t)n * n, 0.0), g(n, 0.0);
        for (int c = 0; c < nCamFree; c++) {
            for (int a = 0; a < 6; a++) {
                for (int b = 0; b < 6; b++) S[(size_t)(c * 6 + a) * n + c * 6 + b] = Hpp[(size_t)c * 36 + a * 6 + b];
                S[(size_t)(c * 6 + a) * n + c * 6 + a] += lam;
                g[c * 6 + a] = bp[(size_t)c * 6 + a];
            }
        }
        std::vector<double> Dinv((size_t)nPtAct * 9);
        for (int p = 0; p < nPtAct; p++) {
            double D[9];
            for (int i = 0; i < 9; i++) D[i] = Hll[(size_t)p * 9 + i];
            D[0] += lam;
            D[4] += lam;
            D[8] += lam;
            const double det = D[0] * (D[4] * D[8] - D[5] * D[7]) - D[1] * (D[3] * D[8] - D[5] * D[6]) +
                               D[2] * (D[3] * D[7] - D[4] * D[6]);
            const double id = 1.0 / det;
            double *I = &Dinv[(size_t)p * 9];
            I[0] = (D[4] * D[8] - D[5] * D[7]) * id;
            I[1] = (D[2] * D[7] - D[1] * D[8]) * id;
            I[2] = (D[1] * D[5] - D[2] * D[4]) * id;
            I[3] = (D[5] * D[6] - D[3] * D[8]) * id;
            I[4] = (D[0] * D[8] - D[2] * D[6]) * id;
            I[5] = (D[2] * D[3] - D[0] * D[5]) * id;
            I[6] = (D[3] * D[7] - D[4] * D[6]) * id;
            I[7] = (D[1] * D[6] - D[0] * D[7]) * id;
            I[8] = (D[0] * D[4] - D[1] * D[3]) * id;
        }
        /* edges grouped by point */
        std::vector<std::vector<int>> byPt(nPtAct);
        for (size_t ei = 0; ei < edges.size(); ei++)
            if (edges[ei].level == 0 && camSlot[edges[ei].cam] >= 0) byPt[ptSlot[edges[ei].pt]].push_back((int)ei);
        for (int p = 0; p < nPtAct; p++) {
            const double *I = &Dinv[(size_t)p * 9];
            for (int e1 : byPt[p]) {
                const int c1 = camSlot[edges[e1].cam];
                double Y[18]; /* W_e1 * Dinv */
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < 3; b++)
                        Y[a * 3 + b] = W[(size_t)e1 * 18 + a * 3] * I[b] + W[(size_t)e1 * 18 + a * 3 + 1] * I[3 + b] +
                                       W[(size_t)e1 * 18 + a * 3 + 2] * I[6 + b];
                for (int a = 0; a < 6; a++)
                    g[c1 * 6 + a] -= Y[a * 3] * bl[(size_t)p * 3] + Y[a * 3 + 1] * bl[(size_t)p * 3 + 1] +
                                     Y[a * 3 + 2] * bl[(size_t)p * 3 + 2];
                for (int e2 : byPt[p]) {
                    const int c2 = camSlot[edges[e2].cam];
                    const double *W2 = &W[(size_t)e2 * 18];
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++)
                            S[(size_t)(c1 * 6 + a) * n + c2 * 6 + b] -=
                                Y[a * 3] * W2[b * 3] + Y[a * 3 + 1] * W2[b * 3 + 1] + Y[a * 3 + 2] * W2[b * 3 + 2];
                }
            }
        }
        linSolves++;
        if (n > 0 && !chol_solve(S, g, n)) return false;
        xp = g;
        xl.assign((size_t)nPtAct * 3, 0.0);
        for (int p = 0; p < nPtAct; p++) {
            double cl[3] = {bl[(size_t)p * 3], bl[(size_t)p * 3 + 1], bl[(size_t)p * 3 + 2]};
            for (int e : byPt[p]) {
                const int c = camSlot[edges[e].cam];
                const double *We = &W[(size_t)e * 18];
                for (int b = 0; b < 3; b++)
                    for (int a = 0; a < 6; a++) cl[b] -= We[a * 3 + b] * xp[c * 6 + a];
            }
            const double *I = &Dinv[(size_t)p * 9];
            for (int a = 0; a < 3; a++) xl[(size_t)p * 3 + a] = I[a * 3] * cl[0] + I[a * 3 + 1] * cl[1] + I[a * 3 + 2] * cl[2];
        }
        return true;
    }

    /* SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg::solve */
    int optimize(int iterations, volatile const uint8_t *stop) {
        int done = 0;
        bool ok = true;
        for (int it = 0; it < iterations && !(stop && *stop) && ok; it++) {
            computeActiveErrors();
            double currentChi = activeRobustChi2();
            double tempChi = currentChi;
            buildSystem();
            if (it == 0) {
                lambda = computeLambdaInit();
                ni = 2;
            }
            double rho = 0;
            int qmax = 0;
            do {
                std::vector<SE3> poseBackup = poses; /* push() */
                std::vector<double> ptBackup = pts;
                std::vector<double> xp, xl;
                bool ok2 = solve(lambda, xp, xl);
                if (ok2) { /* update(x) */
                    for (int c = 0; c < Nc; c++)
                        if (camSlot[c] >= 0) poses[c] = se3_exp_mul(&xp[(size_t)camSlot[c] * 6], poses[c]);
                    for (int p = 0; p < Np; p++)
                        if (ptSlot[p] >= 0)
                            for (int a = 0; a < 3; a++) pts[(size_t)p * 3 + a] += xl[(size_t)ptSlot[p] * 3 + a];
                }
                computeActiveErrors();
                tempChi = activeRobustChi2();
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                rho = (currentChi - tempChi);
                double scale = 0; /* computeScale */
                if (ok2) {
                    for (int c = 0; c < nCamFree * 6; c++) scale += xp[c] * (lambda * xp[c] + bp[c]);
                    for (int p = 0; p < nPtAct * 3; p++) scale += xl[p] * (lambda * xl[p] + bl[p]);
                }
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = std::min(alpha, 2. / 3.);
                    double scaleFactor = std::max(1. / 3., alpha);
                    lambda *= scaleFactor;
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    poses = poseBackup; /* pop() */
                    pts = ptBackup;
                }
                qmax++;
                if (getenv("ORC_BA_DEBUG"))
                    fprintf(stderr, "[orc] it %d trial %d ok %d cur %.10e temp %.10e rho %.6e lambda %.6e\n", it, qmax,
                            (int)ok2, currentChi, tempChi, rho, lambda);
            } while (rho < 0 && qmax < 10 && !(stop && *stop));
            done++;
            if (qmax == 10 || rho == 0) ok = false; /* Terminate */
        }
        return done;
    }
};

} // namespace

extern "C" {

int orc_ba_solve(const OrcBAProblem *p, OrcBAResult *r) {
    BA ba;
    ba.Nc = p->n_cams;
    ba.Np = p->n_points;
    ba.poses.resize(ba.Nc);
    ba.fixed.assign(p->fixed, p->fixed + ba.Nc);
    for (int c = 0; c < ba.Nc; c++) {
        const double *q = p->poses + (size_t)c * 7;
        ba.poses[c].q = {q[0], q[1], q[2], q[3]};
        quat_normalize(ba.poses[c].q); /* SE3Quat ctor normalises */
        ba.poses[c].t[0] = q[4];
        ba.poses[c].t[1] = q[5];
        ba.poses[c].t[2] = q[6];
    }
    ba.pts.assign(p->points, p->points + (size_t)ba.Np * 3);
    ba.fx = p->fx;
    ba.fy = p->fy;
    ba.cx = p->cx;
    ba.cy = p->cy;
    ba.delta = p->huber_delta;
    ba.edges.resize(p->n_obs);
    for (int e = 0; e < p->n_obs; e++) {
        Edge &E = ba.edges[e];
        E.cam = p->cam_idx[e];
        E.pt = p->pt_idx[e];
        E.obs[0] = p->obs[(size_t)e * 2];
        E.obs[1] = p->obs[(size_t)e * 2 + 1];
        E.info = p->inv_sigma2[e];
        E.robust = p->huber_delta > 0;
        E.level = 0;
        E.err[0] = E.err[1] = 0;
    }
    r->iterations = 0;
    ba.initializeOptimization();
    ba.computeActiveErrors();
    r->chi2_initial = ba.activeRobustChi2();
    for (int round = 0; round < p->n_rounds; round++) {
        if (p->stop && *p->stop) break;
        ba.initializeOptimization();
        r->iterations += ba.optimize(p->rounds[round], p->stop);
        if (round + 1 < p->n_rounds) {
            if (p->stop && *p->stop) break; /* bDoMore */
            /* errors at the accepted estimate (g2o may hold a rejected trial's errors here; that stale
               corner case is not reproduced) */
            for (auto &e : ba.edges) {
                ba.computeError(e);
                if (ba.chi2(e) > p->chi2_threshold || !ba.depthPositive(e)) e.level = 1;
                if (p->drop_robust_after == round) e.robust = false;
            }
        }
    }
    for (auto &e : ba.edges) ba.computeError(e);
    {
        double s = 0;
        for (const auto &e : ba.edges)
            if (e.level == 0) {
                double c = ba.chi2(e);
                if (e.robust) {
                    double rho[3];
                    huber(c, ba.delta, rho);
                    c = rho[0];
                }
                s += c;
            }
        r->chi2_final = s;
    }
    for (int c = 0; c < ba.Nc; c++) {
        double *q = r->poses + (size_t)c * 7;
        q[0] = ba.poses[c].q.w;
        q[1] = ba.poses[c].q.x;
        q[2] = ba.poses[c].q.y;
        q[3] = ba.poses[c].q.z;
        q[4] = ba.poses[c].t[0];
        q[5] = ba.poses[c].t[1];
        q[6] = ba.poses[c].t[2];
    }
    memcpy(r->points, ba.pts.data(), (size_t)ba.Np * 3 * sizeof(double));
    if (r->outlier)
        for (int e = 0; e < p->n_obs; e++) r->outlier[e] = ba.edges[e].level != 0;
    r->lin_solves = ba.linSolves;
    return 0;
}

/* Optimizer::PoseOptimization, src/Optimizer.cpp:131-261: one free pose, unary edges with fixed Xw
 * (EdgeSE3ProjectXYZOnlyPose), Huber delta = 5.99 [sic] (:169), dense 6x6 solver, 4 x optimize(10) with
 * chi2 > 5.991 relabelling (:213-247), robust kernel removed when it == 2 (:242-243).  The pose is NOT
 * reset between rounds (unlike upstream).  Intrinsics bug (:138-139) resolved per SURVEY 8c-v. */
int orc_pose_optimization(double *pose7, const double *xw, const double *obs, const double *inv_sigma2, int n,
                          double fx, double fy, double cx, double cy, uint8_t *outlier, int32_t *n_inliers) {
    *n_inliers = 0;
    if (n < 3) return 0; /* :206-207 */
    SE3 T;
    T.q = {pose7[0], pose7[1], pose7[2], pose7[3]};
    quat_normalize(T.q);
    T.t[0] = pose7[4];
    T.t[1] = pose7[5];
    T.t[2] = pose7[6];
    const double delta = 5.99;
    std::vector<int> level(n, 0);
    std::vector<char> robust(n, 1);
    std::vector<double> err((size_t)n * 2, 0.0);
    for (int i = 0; i < n; i++) outlier[i] = 0;
    auto computeError = [&](int i) {
        double R[9], pc[3];
        quat_to_R(T.q, R);
        se3_map(T, R, xw + (size_t)i * 3, pc);
        err[2 * i] = obs[2 * i] - (pc[0] / pc[2] * fx + cx);
        err[2 * i + 1] = obs[2 * i + 1] - (pc[1] / pc[2] * fy + cy);
    };
    auto chi2 = [&](int i) { return (err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]) * inv_sigma2[i]; };
    auto activeChi2 = [&]() {
        double s = 0;
        for (int i = 0; i < n; i++)
            if (level[i] == 0) {
                double c = chi2(i);
                if (robust[i]) {
                    double rho[3];
                    huber(c, delta, rho);
                    c = rho[0];
                }
                s += c;
            }
        return s;
    };
    int nBads = 0;
    double lambda = 0, ni = 2;
    for (int it = 0; it < 4; it++) {
        /* optimize(10) */
        bool ok = true;
        for (int k = 0; k < 10 && ok; k++) {
            for (int i = 0; i < n; i++)
                if (level[i] == 0) computeError(i);
            double currentChi = activeChi2(), tempChi = currentChi;
            double H[36] = {0}, b[6] = {0};
            int nact = 0;
            for (int i = 0; i < n; i++) {
                if (level[i] != 0) continue;
                nact++;
                double R[9], pc[3];
                quat_to_R(T.q, R);
                se3_map(T, R, xw + (size_t)i * 3, pc);
                const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz;
                double Jj[12];
                Jj[0] = x * y * invz_2 * fx;
                Jj[1] = -(1 + (x * x * invz_2)) * fx;
                Jj[2] = y * invz * fx;
                Jj[3] = -invz * fx;
                Jj[4] = 0;
                Jj[5] = x * invz_2 * fx;
                Jj[6] = (1 + y * y * invz_2) * fy;
                Jj[7] = -x * y * invz_2 * fy;
                Jj[8] = -x * invz * fy;
                Jj[9] = 0;
                Jj[10] = -invz * fy;
                Jj[11] = y * invz_2 * fy;
                double w = inv_sigma2[i], er[2] = {-w * err[2 * i], -w * err[2 * i + 1]};
                if (robust[i]) {
                    double rho[3];
                    huber(chi2(i), delta, rho);
                    w *= rho[1];
                    er[0] *= rho[1];
                    er[1] *= rho[1];
                }
                for (int a = 0; a < 6; a++) {
                    b[a] += Jj[a] * er[0] + Jj[6 + a] * er[1];
                    for (int c = 0; c < 6; c++) H[a * 6 + c] += w * (Jj[a] * Jj[c] + Jj[6 + a] * Jj[6 + c]);
                }
            }
            if (nact == 0) break;
            if (k == 0) {
                double m = 0;
                for (int j = 0; j < 6; j++) m = std::max(m, std::fabs(H[j * 7]));
                lambda = 1e-5 * m;
                ni = 2;
            }
            double rho = 0;
            int qmax = 0;
            do {
                SE3 backup = T;
                std::vector<double> A(H, H + 36), x(b, b + 6);
                for (int j = 0; j < 6; j++) A[j * 7] += lambda;
                bool ok2 = chol_solve(A, x, 6);
                if (ok2) T = se3_exp_mul(x.data(), T);
                for (int i = 0; i < n; i++)
                    if (level[i] == 0) computeError(i);
                tempChi = activeChi2();
                if (!ok2) tempChi = std::numeric_limits<double>::max();
                rho = currentChi - tempChi;
                double scale = 0;
                if (ok2)
                    for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho /= scale;
                if (rho > 0 && std::isfinite(tempChi)) {
                    double alpha = 1. - std::pow((2 * rho - 1), 3);
                    alpha = std::min(alpha, 2. / 3.);
                    lambda *= std::max(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni;
                    ni *= 2;
                    T = backup;
                }
                qmax++;
            } while (rho < 0 && qmax < 10);
            if (qmax == 10 || rho == 0) ok = false;
        }
        nBads = 0;
        for (int i = 0; i < n; i++) {
            computeError(i); /* :224-227 recomputes inactive edges; active ones are re-evaluated at the
                                accepted pose too (g2o's stale-trial corner case is not reproduced) */
            const double c = chi2(i);
            if (c > 5.991) {
                level[i] = 1;
                outlier[i] = 1;
                nBads++;
            } else {
                level[i] = 0;
                outlier[i] = 0;
            }
            if (it == 2) robust[i] = 0;
        }
        if (n < 10) break; /* optimizer.edges().size() < 10 (:246) */
    }
    pose7[0] = T.q.w;
    pose7[1] = T.q.x;
    pose7[2] = T.q.y;
    pose7[3] = T.q.z;
    pose7[4] = T.t[0];
    pose7[5] = T.t[1];
    pose7[6] = T.t[2];
    *n_inliers = n - nBads;
    return 0;
}

} /* extern "C" */
