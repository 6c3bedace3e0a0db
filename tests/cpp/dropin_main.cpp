/*
 * dropin_main.cpp -- exercises EVERY public method of the C++ drop-in classes (include/orbslam_amd/) the way the
 * reference's Frame / Tracking / LocalMapping call them, by the reference's own class names and signatures:
 * `Frame`, `KeyFrame`, `MapPoint`, `Map` come from tests/cpp/refmodel/ (stand-ins with the reference's accessor
 * surface), `ORBextractor`, `ORBmatcher`, `Optimizer` from the drop-in headers, which find "Frame.h" /
 * "KeyFrame.h" on the include path exactly as in the reference tree.  Results go to files that
 * tests/test_gpu_dropin.py compares with the C-ABI path and the oracle.
 *
 *   dropin_main <w> <h> <frame0.raw> <frame1.raw> <ba_window.bin> <out_prefix>
 *
 * ba_window.bin: int32 Nc, Np, E | f64 poses[Nc*7] (qw qx qy qz tx ty tz) | f64 points[Np*3] | f32 obs[E*2] |
 *                int32 cam[E], pt[E], octave[E] | f32 fx, fy, cx, cy
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "ORBmatcher.h" /* forwarders in tests/cpp/refmodel would be the reference's include/ORBmatcher.h etc. */
#include "Optimizer.h"

using namespace ORBSlam;

float KeyFrame::fx, KeyFrame::fy, KeyFrame::cx, KeyFrame::cy;
cv::Mat Frame::mcamk;

static std::vector<unsigned char> read_file(const char *p, size_t n) {
    std::vector<unsigned char> b(n);
    FILE *f = fopen(p, "rb");
    if (!f || fread(b.data(), 1, n, f) != n) {
        fprintf(stderr, "cannot read %s\n", p);
        exit(2);
    }
    fclose(f);
    return b;
}
static std::vector<unsigned char> read_all(const char *p) {
    FILE *f = fopen(p, "rb");
    if (!f) {
        fprintf(stderr, "cannot read %s\n", p);
        exit(2);
    }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b(n);
    if (fread(b.data(), 1, n, f) != (size_t)n) exit(2);
    fclose(f);
    return b;
}

struct Window {
    int Nc, Np, E;
    std::vector<double> poses, points;
    std::vector<float> obs;
    std::vector<int> cam, pt, oct;
    float fx, fy, cx, cy;
};
static Window load_window(const char *path) {
    std::vector<unsigned char> b = read_all(path);
    const unsigned char *p = b.data();
    Window w;
    auto take = [&](void *dst, size_t n) {
        memcpy(dst, p, n);
        p += n;
    };
    take(&w.Nc, 4);
    take(&w.Np, 4);
    take(&w.E, 4);
    w.poses.resize((size_t)w.Nc * 7);
    w.points.resize((size_t)w.Np * 3);
    w.obs.resize((size_t)w.E * 2);
    w.cam.resize(w.E);
    w.pt.resize(w.E);
    w.oct.resize(w.E);
    take(w.poses.data(), w.poses.size() * 8);
    take(w.points.data(), w.points.size() * 8);
    take(w.obs.data(), w.obs.size() * 4);
    take(w.cam.data(), (size_t)w.E * 4);
    take(w.pt.data(), (size_t)w.E * 4);
    take(w.oct.data(), (size_t)w.E * 4);
    take(&w.fx, 4);
    take(&w.fy, 4);
    take(&w.cx, 4);
    take(&w.cy, 4);
    return w;
}
static Eigen::Matrix3d quat_R(const double *q) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    Eigen::Matrix3d R;
    R(0, 0) = 1 - 2 * (y * y + z * z); R(0, 1) = 2 * (x * y - z * w);     R(0, 2) = 2 * (x * z + y * w);
    R(1, 0) = 2 * (x * y + z * w);     R(1, 1) = 1 - 2 * (x * x + z * z); R(1, 2) = 2 * (y * z - x * w);
    R(2, 0) = 2 * (x * z - y * w);     R(2, 1) = 2 * (y * z + x * w);     R(2, 2) = 1 - 2 * (x * x + y * y);
    return R;
}

/* a map the way Tracking::CreateInitialMap / LocalMapping::ProcessNewKeyFrame leave it: keyframes with their
 * undistorted keypoints and point slots, map points with their observation lists */
struct World {
    std::vector<KeyFrame> kfs;
    std::vector<MapPoint> mps;
    Map map;
    explicit World(const Window &w) : kfs(w.Nc), mps(w.Np) {
        for (int c = 0; c < w.Nc; c++) {
            kfs[c].mlnId = c;
            kfs[c].SetR(quat_R(&w.poses[c * 7]));
            kfs[c].Settrans(Eigen::Vector3d(w.poses[c * 7 + 4], w.poses[c * 7 + 5], w.poses[c * 7 + 6]));
        }
        for (int j = 0; j < w.Np; j++) {
            mps[j].mlnId = j;
            mps[j].SetPos(Eigen::Vector3d(w.points[j * 3], w.points[j * 3 + 1], w.points[j * 3 + 2]));
        }
        for (int e = 0; e < w.E; e++) {
            KeyFrame &k = kfs[w.cam[e]];
            const int id = (int)k.mvUnKeypts.size();
            k.mvUnKeypts.push_back(cv::KeyPoint(w.obs[2 * e], w.obs[2 * e + 1], 31.f, -1.f, 0.f, w.oct[e]));
            k.AddMapPoint(&mps[w.pt[e]], id);
            mps[w.pt[e]].AddObservation(&k, id);
        }
        for (auto &k : kfs) map.AddKeyFrame(&k);
        for (auto &m : mps) map.AddMapPoint(&m);
    }
    void dump(const std::string &path) const {
        FILE *f = fopen(path.c_str(), "wb");
        for (const KeyFrame &k : kfs) {
            const Eigen::Matrix3d R = k.GetR();
            const Eigen::Vector3d t = k.Gettrans();
            fwrite(R.v, 8, 9, f);
            fwrite(t.v, 8, 3, f);
        }
        for (const MapPoint &m : mps) {
            const Eigen::Vector3d p = m.Getpos();
            fwrite(p.v, 8, 3, f);
        }
        fclose(f);
    }
    std::vector<double> state() const {
        std::vector<double> s;
        for (const KeyFrame &k : kfs) {
            const Eigen::Matrix3d R = k.GetR();
            const Eigen::Vector3d t = k.Gettrans();
            s.insert(s.end(), R.v, R.v + 9);
            s.insert(s.end(), t.v, t.v + 3);
        }
        for (const MapPoint &m : mps) {
            const Eigen::Vector3d p = m.Getpos();
            s.insert(s.end(), p.v, p.v + 3);
        }
        return s;
    }
};

/* a tracked frame as Tracking::TrackReferenceKeyFrame hands it to PoseOptimization: the keypoints of camera c,
 * every third slot without a map point */
static void make_tracked_frame(const Window &w, World &world, int c, Frame &F) {
    F.mvUnKeypts.clear();
    F.mvpMappts.clear();
    for (int e = 0; e < w.E; e++)
        if (w.cam[e] == c) {
            F.mvUnKeypts.push_back(cv::KeyPoint(w.obs[2 * e], w.obs[2 * e + 1], 31.f, -1.f, 0.f, w.oct[e]));
            F.mvpMappts.push_back(F.mvpMappts.size() % 3 == 2 ? nullptr : &world.mps[w.pt[e]]);
        }
    F.mvbOutlier.assign(F.mvpMappts.size(), true); /* PoseOptimization must reset them (src/Optimizer.cpp:176) */
    Eigen::Matrix4d T;
    const Eigen::Matrix3d R = quat_R(&w.poses[c * 7]);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) T(i, j) = R(i, j);
        T(i, 3) = w.poses[c * 7 + 4 + i];
    }
    F.SetPose(T);
}
static std::vector<double> frame_state(const Frame &F, int ret) {
    std::vector<double> s;
    const Eigen::Matrix4d T = F.GetPose();
    s.insert(s.end(), T.v, T.v + 16);
    s.push_back((double)ret);
    for (size_t i = 0; i < F.mvbOutlier.size(); i++) s.push_back(F.mvbOutlier[i] ? 1.0 : 0.0);
    return s;
}

int main(int argc, char **argv) {
    if (argc < 7) return 2;
    const int w = atoi(argv[1]), h = atoi(argv[2]);
    std::vector<unsigned char> b0 = read_file(argv[3], (size_t)w * h), b1 = read_file(argv[4], (size_t)w * h);
    const Window win = load_window(argv[5]);
    const std::string pre(argv[6]);
    cv::Mat im0(h, w, CV_8UC1, b0.data()), im1(h, w, CV_8UC1, b1.data());

    /* ---------------- ORBextractor: src/Tracking.cpp:47-48, src/Frame.cpp:77 ---------------- */
    ORBextractor *mpIniOrbextractor = new ORBextractor(2 * 1000, 1.2f, 8, 20, 7);
    if (mpIniOrbextractor->GetLevels() != 8 || std::fabs(mpIniOrbextractor->GetScaleFactor() - 1.2f) > 1e-6f) return 3;
    const std::vector<float> sf = mpIniOrbextractor->GetScaleFactors(), isf = mpIniOrbextractor->GetInverseScaleFactors(),
                             s2 = mpIniOrbextractor->GetScaleSigmaSquares(), is2 = mpIniOrbextractor->GetInverseScaleSigmaSquares();
    if (sf.size() != 8 || isf.size() != 8 || s2.size() != 8 || is2.size() != 8 || sf[0] != 1.f || s2[1] != sf[1] * sf[1]) return 3;
    {
        /* empty image: outputs untouched (src/ORBextractor.cpp:1004-1005) */
        std::vector<cv::KeyPoint> kk(3);
        cv::Mat dd(2, 32, CV_8U);
        (*mpIniOrbextractor)(cv::Mat(), cv::Mat(), kk, dd);
        if (kk.size() != 3 || dd.rows != 2) return 4;
    }
    Frame F0(im0, mpIniOrbextractor), F1(im1, mpIniOrbextractor);
    FILE *f = fopen((pre + "_f0.bin").c_str(), "wb");
    const int n0 = (int)F0.mvKeypts.size();
    fwrite(&n0, 4, 1, f);
    fwrite(F0.mvKeypts.data(), sizeof(cv::KeyPoint), n0, f);
    for (int i = 0; i < n0; i++) fwrite(F0.mcvDescriptors.ptr(i), 1, 32, f);
    fclose(f);
    /* the public mvImagePyramid member, read like a std::vector<cv::Mat> (include/ORBextractor.h:71): holds frame 1's
       pyramid now; every level is the ROI at (19,19) of its padded image, so the apron is at negative offsets */
    f = fopen((pre + "_pyr.bin").c_str(), "wb");
    const int nl = (int)mpIniOrbextractor->mvImagePyramid.size();
    fwrite(&nl, 4, 1, f);
    for (int l = 0; l < nl; l++) {
        const cv::Mat &lv = mpIniOrbextractor->mvImagePyramid[l];
        const int dims[3] = {lv.cols, lv.rows, (int)lv.step};
        fwrite(dims, 4, 3, f);
        for (int y = -19; y < lv.rows + 19; y++) fwrite(lv.data + (ptrdiff_t)y * (ptrdiff_t)lv.step - 19, 1, lv.cols + 38, f);
    }
    fclose(f);

    /* ---------------- ORBmatcher: src/Tracking.cpp:170-189 (no bounds set by hand) ---------------- */
    std::vector<cv::Point2f> vbPrevMatched(F0.GetUnKeyPts().size());
    for (size_t i = 0; i < vbPrevMatched.size(); i++) vbPrevMatched[i] = F0.GetUnKeyPts()[i].pt;
    std::vector<int> vnMatches12;
    ORBmatcher matcher(0.9, true);
    const int nmatches = matcher.SearchForInitialization(F0, F1, vbPrevMatched, vnMatches12, 100);
    f = fopen((pre + "_match.bin").c_str(), "wb");
    fwrite(&nmatches, 4, 1, f);
    fwrite(vnMatches12.data(), 4, vnMatches12.size(), f);
    fclose(f);
    {
        int dist_self = matcher.DescriptorDistance(F0.mcvDescriptors.row(0), F0.mcvDescriptors.row(0));
        int dist01 = matcher.DescriptorDistance(F0.mcvDescriptors.row(0), F0.mcvDescriptors.row(1));
        int ref01 = 0;
        for (int i = 0; i < 32; i++) ref01 += __builtin_popcount(F0.mcvDescriptors.ptr(0)[i] ^ F0.mcvDescriptors.ptr(1)[i]);
        std::vector<int> histo[30];
        for (int i = 0; i < 30; i++) histo[i].assign((i * 7) % 11, 0);
        int i1 = -1, i2 = -1, i3 = -1;
        matcher.ComputeThreeMaxima(histo, 30, i1, i2, i3);
        if (dist_self != 0 || dist01 != ref01 || i1 < 0 || histo[i1].size() != 10) return 5;
    }
    Frame::SetCameraPara(win.fx, win.fy, win.cx, win.cy);
    {
        /* SearchByProjection (src/Tracking.cpp:344) and SearchByBoW (:298) on the two extracted frames.  Map
           points for frame 0: its keypoints lifted to depth 4 with the identity pose (three of every four features, every
           other one with an observation); frame 1 keeps the identity pose, the 2-D motion of the synthetic stream is
           what the search radius has to absorb */
        const int nA = (int)F0.mvKeypts.size();
        std::vector<MapPoint> pts(nA);
        KeyFrame kf0;
        kf0.mvUnKeypts = F0.GetUnKeyPts();
        kf0.mcvDescriptors = F0.mcvDescriptors;
        for (int i = 0; i < nA; i++) {
            const cv::KeyPoint &kp = F0.mvKeypts[i];
            pts[i].mlnId = i;
            pts[i].SetPos(Eigen::Vector3d((kp.pt.x - win.cx) / win.fx * 4.0, (kp.pt.y - win.cy) / win.fy * 4.0, 4.0));
            if (i % 2 == 0) pts[i].AddObservation(&kf0, i);
            if (i % 4 != 3) {
                F0.mvpMappts[i] = &pts[i];
                kf0.AddMapPoint(&pts[i], i);
            }
        }
        F0.mvbOutlier[5] = true;
        ORBmatcher m2(0.9, true);                                            /* src/Tracking.cpp:333 */
        const int nproj = m2.SearchByProjection(F1, F0, 30.f, true);
        std::vector<MapPoint *> vpMapPointMatches;
        ORBmatcher m3(0.7, true);                                            /* src/Tracking.cpp:293 */
        const int nbow = m3.SearchByBoW(&kf0, F1, vpMapPointMatches);
        f = fopen((pre + "_n4.bin").c_str(), "wb");
        const int n1 = (int)F1.mvKeypts.size();
        const int hdr[3] = {nproj, nbow, n1};
        fwrite(hdr, 4, 3, f);
        for (int j = 0; j < n1; j++) {
            const int id = F1.mvpMappts[j] ? (int)F1.mvpMappts[j]->mlnId : -1;
            fwrite(&id, 4, 1, f);
        }
        for (int j = 0; j < n1; j++) {
            const int id = vpMapPointMatches[j] ? (int)vpMapPointMatches[j]->mlnId : -1;
            fwrite(&id, 4, 1, f);
        }
        fclose(f);
        for (int j = 0; j < n1; j++) F1.mvpMappts[j] = nullptr; /* pts goes out of scope */
        for (int i = 0; i < nA; i++) F0.mvpMappts[i] = nullptr;
    }

    /* ---------------- Optimizer ---------------- */
    KeyFrame::fx = win.fx;
    KeyFrame::fy = win.fy;
    KeyFrame::cx = win.cx;
    KeyFrame::cy = win.cy;
    std::vector<double> gba_state, po_state;
    {
        World world(win); /* src/Tracking.cpp:254 */
        Optimizer::GlobalBundleAdjustemnt(&world.map, 20);
        world.dump(pre + "_gba.bin");
        gba_state = world.state();
    }
    {
        World world(win); /* the explicit-vector form, non-robust, 5 iterations, with a (clear) stop flag */
        bool stop = false;
        Optimizer::BundleAdjustment(world.map.GetKeyFrames(), world.map.GetMapPoints(), 5, &stop, 0, false);
        world.dump(pre + "_ba5.bin");
    }
    const int cur = win.Nc / 2;
    {
        World world(win); /* LocalMapping::Run after ProcessNewKeyFrame (src/LocalMapping.cpp:11-20) */
        bool mbAbortBA = false;
        std::vector<KeyFrame *> lk, fk;
        std::vector<MapPoint *> lp;
        Optimizer::LocalWindow(&world.kfs[cur], lk, fk, lp);
        f = fopen((pre + "_lwin.bin").c_str(), "wb");
        const int cnt[3] = {(int)lk.size(), (int)fk.size(), (int)lp.size()};
        fwrite(cnt, 4, 3, f);
        for (KeyFrame *k : lk) { const int id = (int)k->mlnId; fwrite(&id, 4, 1, f); }
        for (KeyFrame *k : fk) { const int id = (int)k->mlnId; fwrite(&id, 4, 1, f); }
        for (MapPoint *m : lp) { const int id = (int)m->mlnId; fwrite(&id, 4, 1, f); }
        fclose(f);
        Optimizer::LocalBundleAdjustment(&world.kfs[cur], &mbAbortBA, &world.map);
        world.dump(pre + "_lba.bin");
        /* window-explicit overloads give the same result */
        World w2(win);
        std::vector<KeyFrame *> lk2, fk2;
        std::vector<MapPoint *> lp2;
        Optimizer::LocalWindow(&w2.kfs[cur], lk2, fk2, lp2);
        Optimizer::LocalBundleAdjustment(lk2, fk2, lp2, &mbAbortBA);
        if (w2.state() != world.state()) return 6;
        World w3(win);
        Optimizer::LocalBundleAdjustment(w3.map.GetKeyFrames(), w3.map.GetMapPoints(), (bool *)NULL); /* whole map, no fixed observers */
        w3.dump(pre + "_lba_all.bin");
    }
    {
        World world(win); /* stop flag already set: nothing moves (src/Optimizer.cpp:39-40) */
        const std::vector<double> before = world.state();
        bool stop = true;
        Optimizer::LocalBundleAdjustment(&world.kfs[cur], &stop, &world.map);
        if (world.state() != before) return 7;
    }
    int po_ret = 0;
    {
        World world(win); /* src/Tracking.cpp:309,355,382 */
        Frame mCurframe;
        make_tracked_frame(win, world, cur, mCurframe);
        po_ret = Optimizer::PoseOptimization(&mCurframe);
        po_state = frame_state(mCurframe, po_ret);
        f = fopen((pre + "_po.bin").c_str(), "wb");
        fwrite(po_state.data(), 8, po_state.size(), f);
        fclose(f);
        /* explicit-intrinsics form: same answer */
        Frame F2;
        make_tracked_frame(win, world, cur, F2);
        const int r2 = Optimizer::PoseOptimization(&F2, win.fx, win.fy, win.cx, win.cy);
        if (frame_state(F2, r2) != po_state) return 8;
        /* fewer than 3 correspondences: returns 0, pose untouched (:206-207) */
        Frame F3;
        make_tracked_frame(win, world, cur, F3);
        for (size_t i = 2; i < F3.mvpMappts.size(); i++) F3.mvpMappts[i] = nullptr;
        const Eigen::Matrix4d T3 = F3.GetPose();
        if (Optimizer::PoseOptimization(&F3) != 0 || memcmp(T3.v, F3.GetPose().v, sizeof(T3.v))) return 9;
    }
    {
        /* Tracking optimises poses while LocalMapping runs BA (upstream's threading; every thread owns its solver):
           both results must equal the sequential ones bit for bit */
        bool ok_a = true, ok_b = true;
        std::thread ta([&] {
            for (int rep = 0; rep < 20 && ok_a; rep++) {
                World world(win);
                Frame F;
                make_tracked_frame(win, world, cur, F);
                const int r = Optimizer::PoseOptimization(&F);
                ok_a = frame_state(F, r) == po_state;
            }
        });
        std::thread tb([&] {
            for (int rep = 0; rep < 2 && ok_b; rep++) {
                World world(win);
                Optimizer::GlobalBundleAdjustemnt(&world.map, 20);
                ok_b = world.state() == gba_state;
            }
        });
        ta.join();
        tb.join();
        if (!ok_a || !ok_b) {
            fprintf(stderr, "threaded results differ: pose %d ba %d\n", (int)ok_a, (int)ok_b);
            return 10;
        }
    }
    delete mpIniOrbextractor;
    printf("dropin ok: %d keypoints, %d matches, pose inliers %d\n", n0, nmatches, po_ret);
    return 0;
}
