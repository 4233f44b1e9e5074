// The eleven reference-signature search methods of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h, GPU) against their
// sequential CPU restatements (oracle/orb_oracle_frames.h) on seeded synthetic scenes: a cloud of 3-D points seen
// from two poses, keypoints at the projections plus clutter, descriptors that are noisy copies of a per-point code.
// Every scenario builds the same scene twice (one copy per implementation), runs both and compares every container
// the reference writes.  All scenarios run twice: with plain Frames / KeyFrames (the matchers upload their arrays) and
// with Frames that carry a device-resident copy (mpDeviceFrame: the matchers read the train side on the GPU).
// Prints one line per scenario; exit status 0 only if all agree.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <string>

#include "ORBmatcher.h"
#include "orb_oracle_frames.h"
#include "slam_types.h"

using namespace slamtest;

// the KeyFrame fixtures remember the float bounds their grid was filled with
// (ORBGPU_TEST_MUTATE_ORIGIN: a self-check of this test -- the GPU side then bins a KeyFrame's keypoints with the
// truncated bounds while the oracle keeps the Frame's, and the scenarios with non-integer bounds must notice)
#ifndef ORBGPU_TEST_MUTATE_ORIGIN
namespace ORB_SLAM2 {
template <> struct KeyFrameGridOrigin<KeyFrame> {
    static void get(const KeyFrame &kf, float &x, float &y) { x = kf.gridOriginX; y = kf.gridOriginY; }
};
}  // namespace ORB_SLAM2
#endif
namespace orc_frames {
template <> struct KeyFrameOrigin<KeyFrame> {
    static void get(const KeyFrame &kf, float &x, float &y) { x = kf.gridOriginX; y = kf.gridOriginY; }
};
}  // namespace orc_frames

struct Cfg {
    uint32_t seed = 1;
    bool stereo = true;
    bool distorted = false;  // non-integer image bounds (a camera with distortion)
    float motion_z = 0.3f;   // translation of view B along the optical axis
    int npoints = 1400;
    int nclutter = 350;
};

struct View {
    float R[9], t[3];
    std::vector<orbcv::KeyPoint> keys;
    std::vector<uint8_t> desc;
    std::vector<float> uright;
    std::vector<int> point;  // world point per keypoint, -1 for clutter
    std::vector<unsigned> node;
    std::vector<float> dist;  // distance of the point from the camera centre
};

struct Scene {
    Cfg cfg;
    float fx = 520.f, fy = 523.f, cx = 321.5f, cy = 238.2f, bf = 40.f;
    float bounds[4];
    std::vector<float> scale, sigma2, inv_sigma2;
    std::vector<std::unique_ptr<MapPoint> > mps;
    View va, vb;
    Frame FA, FB;
    KeyFrame KA, KB;
    std::mt19937 rng;
    float uni(float a, float b) { return a + (b - a) * (float)(rng() >> 8) * (1.0f / 16777216.0f); }
    float gauss(float s) { float a = 0; for (int i = 0; i < 6; i++) a += uni(-1.f, 1.f); return a * s * 0.7071f; }
    int pick(int n) { return (int)(rng() % (uint32_t)n); }
    bool chance(float p) { return uni(0.f, 1.f) < p; }
};

static void rot_xy(float ax, float ay, float *R)
{
    const double cx = cos(ax), sx = sin(ax), cy = cos(ay), sy = sin(ay);
    const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += Rx[3 * i + k] * Ry[3 * k + j];
            R[3 * i + j] = (float)a;
        }
}

static FMat pose4(const float *R, const float *t)
{
    FMat T(4, 4);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.at<float>(i, j) = R[3 * i + j]; T.at<float>(i, 3) = t[i]; }
    T.at<float>(3, 3) = 1.f;
    return T;
}

struct WorldPoint { float p[3]; uint8_t code[32]; float angle; int levelA; };

static void make_view(Scene &S, const std::vector<WorldPoint> &W, View &V, float view_rot_deg, bool isA)
{
    const float *b = S.bounds;
    struct Tmp { orbcv::KeyPoint kp; uint8_t d[32]; float ur; int pt; unsigned node; float dist; };
    std::vector<Tmp> all;
    double C[3];  // camera centre
    for (int i = 0; i < 3; i++) C[i] = -((double)V.R[i] * V.t[0] + (double)V.R[3 + i] * V.t[1] + (double)V.R[6 + i] * V.t[2]);
    for (size_t j = 0; j < W.size(); j++) {
        const float *p = W[j].p;
        const float x = V.R[0] * p[0] + V.R[1] * p[1] + V.R[2] * p[2] + V.t[0];
        const float y = V.R[3] * p[0] + V.R[4] * p[1] + V.R[5] * p[2] + V.t[1];
        const float z = V.R[6] * p[0] + V.R[7] * p[1] + V.R[8] * p[2] + V.t[2];
        if (z < 0.5f) continue;
        const float u = S.fx * x / z + S.cx, v = S.fy * y / z + S.cy;
        if (u < b[0] + 2 || u > b[2] - 2 || v < b[1] + 2 || v > b[3] - 2) continue;
        if (S.chance(0.08f)) continue;  // not detected in this view
        Tmp t;
        const float d = (float)sqrt((p[0] - C[0]) * (p[0] - C[0]) + (p[1] - C[1]) * (p[1] - C[1]) + (p[2] - C[2]) * (p[2] - C[2]));
        int lvl = W[j].levelA;
        if (!isA) lvl += S.pick(3) - 1;
        lvl = std::min(7, std::max(0, lvl));
        t.kp.pt.x = u + S.gauss(0.4f); t.kp.pt.y = v + S.gauss(0.4f);
        t.kp.size = 31.f * S.scale[lvl];
        float ang = W[j].angle + view_rot_deg + S.gauss(2.0f);
        if (S.chance(0.1f)) ang = S.uni(0.f, 360.f);
        ang = fmodf(ang + 720.f, 360.f);
        t.kp.angle = ang; t.kp.response = (float)(20 + S.pick(100)); t.kp.octave = lvl; t.kp.class_id = -1;
        memcpy(t.d, W[j].code, 32);
        const int flips = S.pick(26);
        for (int k = 0; k < flips; k++) { const int bit = S.pick(256); t.d[bit >> 3] ^= (uint8_t)(1u << (bit & 7)); }
        t.ur = (S.cfg.stereo && S.chance(0.7f)) ? u - S.bf / z + S.gauss(0.3f) : -1.f;
        t.pt = (int)j; t.node = W[j].code[0] & 31u; t.dist = d;
        all.push_back(t);
        if (S.chance(0.15f)) {  // a twin a few pixels away with the same descriptor: which of the two wins a search is
            Tmp w = t;          // decided by the order in which the grid lists them
            w.kp.pt.x += (S.chance(0.5f) ? 1.f : -1.f) * S.uni(1.5f, 6.f);
            w.kp.pt.y += (S.chance(0.5f) ? 1.f : -1.f) * S.uni(1.5f, 6.f);
            w.kp.pt.x = std::min(b[2] - 1.f, std::max(b[0] + 1.f, w.kp.pt.x));
            w.kp.pt.y = std::min(b[3] - 1.f, std::max(b[1] + 1.f, w.kp.pt.y));
            w.pt = -1;
            all.push_back(w);
        }
    }
    for (int c = 0; c < S.cfg.nclutter; c++) {
        Tmp t;
        t.kp.pt.x = S.uni(b[0] + 1, b[2] - 1); t.kp.pt.y = S.uni(b[1] + 1, b[3] - 1);
        const int lvl = S.pick(8);
        t.kp.size = 31.f * S.scale[lvl]; t.kp.angle = S.uni(0.f, 359.9f); t.kp.response = 30.f; t.kp.octave = lvl; t.kp.class_id = -1;
        for (int k = 0; k < 32; k++) t.d[k] = (uint8_t)S.pick(256);
        t.ur = (S.cfg.stereo && S.chance(0.5f)) ? t.kp.pt.x - S.uni(1.f, 30.f) : -1.f;
        t.pt = -1; t.node = t.d[0] & 31u; t.dist = 0;
        all.push_back(t);
    }
    std::shuffle(all.begin(), all.end(), S.rng);
    for (const Tmp &t : all) {
        V.keys.push_back(t.kp); V.desc.insert(V.desc.end(), t.d, t.d + 32); V.uright.push_back(t.ur);
        V.point.push_back(t.pt); V.node.push_back(t.node); V.dist.push_back(t.dist);
    }
}

static bool g_device_frames = false;  // the scenes' Frames and KeyFrames carry a DeviceFrame

static void fill_frame(Scene &S, const View &V, Frame &F)
{
    F.N = (int)V.keys.size();
    F.mvKeys = V.keys; F.mvKeysUn = V.keys;
    F.mDescriptors.create(F.N, 32);
    memcpy(F.mDescriptors.data, V.desc.data(), V.desc.size());
    F.mvpMapPoints.assign(F.N, nullptr);
    F.mvbOutlier.assign(F.N, false);
    F.mvuRight = V.uright;
    F.mvScaleFactors = S.scale;
    for (int i = 0; i < F.N; i++) F.mFeatVec[V.node[i]].push_back((unsigned)i);
    F.mTcw = pose4(V.R, V.t);
    F.fx = S.fx; F.fy = S.fy; F.cx = S.cx; F.cy = S.cy; F.mbf = S.bf; F.mb = S.bf / S.fx;
    F.mnMinX = S.bounds[0]; F.mnMinY = S.bounds[1]; F.mnMaxX = S.bounds[2]; F.mnMaxY = S.bounds[3];
    F.mnScaleLevels = 8; F.mfLogScaleFactor = logf(1.2f);
    if (g_device_frames) {
        F.mpDeviceFrame = std::make_shared<ORB_SLAM2::DeviceFrame>(F.N > 0 ? F.N : 1);
        if (F.mpDeviceFrame->Upload(F.mvKeys.data(), F.mvKeysUn.data(), F.mDescriptors.data, F.mvuRight.data(), nullptr, F.N, 8, S.bounds) != F.N) {
            fprintf(stderr, "DeviceFrame::Upload: %s\n", F.mpDeviceFrame->LastError().c_str());
            exit(2);
        }
    }
}

static void fill_keyframe(Scene &S, const View &V, const Frame &F, KeyFrame &K)
{
    K.N = F.N; K.mvKeys = F.mvKeys; K.mvKeysUn = F.mvKeysUn; K.mvuRight = F.mvuRight;
    K.mDescriptors.create(F.N, 32);
    memcpy(K.mDescriptors.data, F.mDescriptors.data, (size_t)F.N * 32);
    K.mFeatVec = F.mFeatVec;
    K.mvScaleFactors = S.scale; K.mvLevelSigma2 = S.sigma2; K.mvInvLevelSigma2 = S.inv_sigma2;
    K.fx = S.fx; K.fy = S.fy; K.cx = S.cx; K.cy = S.cy; K.mbf = S.bf;
    K.mnMinX = (int)F.mnMinX; K.mnMinY = (int)F.mnMinY; K.mnMaxX = (int)F.mnMaxX; K.mnMaxY = (int)F.mnMaxY;
    K.mfGridElementWidthInv = 64.f / (F.mnMaxX - F.mnMinX);
    K.mfGridElementHeightInv = 48.f / (F.mnMaxY - F.mnMinY);
    K.gridOriginX = F.mnMinX; K.gridOriginY = F.mnMinY;
    K.mnScaleLevels = 8; K.mfLogScaleFactor = logf(1.2f);
    K.Tcw = pose4(V.R, V.t);
    K.Ow = FMat(3, 1);
    for (int i = 0; i < 3; i++) K.Ow.at<float>(i, 0) = -(V.R[i] * V.t[0] + V.R[3 + i] * V.t[1] + V.R[6 + i] * V.t[2]);
    K.mvpMapPoints.assign(K.N, nullptr);
    K.mpDeviceFrame = F.mpDeviceFrame;  // a KeyFrame keeps the arrays of its Frame (src/KeyFrame.cc:37-52)
}

static std::unique_ptr<Scene> build(const Cfg &cfg)
{
    std::unique_ptr<Scene> sp(new Scene());
    Scene &S = *sp;
    S.cfg = cfg;
    S.rng.seed(cfg.seed);
    if (cfg.distorted) { S.bounds[0] = -7.62f; S.bounds[1] = -5.31f; S.bounds[2] = 652.4f; S.bounds[3] = 488.7f; }
    else { S.bounds[0] = 0.f; S.bounds[1] = 0.f; S.bounds[2] = 640.f; S.bounds[3] = 480.f; }
    S.scale.resize(8); S.sigma2.resize(8); S.inv_sigma2.resize(8);
    S.scale[0] = 1.f;
    for (int l = 1; l < 8; l++) S.scale[l] = S.scale[l - 1] * 1.2f;
    for (int l = 0; l < 8; l++) { S.sigma2[l] = S.scale[l] * S.scale[l]; S.inv_sigma2[l] = 1.f / S.sigma2[l]; }
    std::vector<WorldPoint> W(cfg.npoints);
    for (WorldPoint &w : W) {
        w.p[2] = S.uni(3.f, 12.f);
        w.p[0] = S.uni(-0.7f, 0.7f) * w.p[2]; w.p[1] = S.uni(-0.52f, 0.52f) * w.p[2];
        for (int k = 0; k < 32; k++) w.code[k] = (uint8_t)S.pick(256);
        w.angle = S.uni(0.f, 360.f);
        w.levelA = S.chance(0.45f) ? 0 : S.pick(6);
    }
    rot_xy(0.004f, 0.02f, S.va.R); S.va.t[0] = 0.05f; S.va.t[1] = -0.02f; S.va.t[2] = 0.01f;
    rot_xy(0.012f, 0.045f, S.vb.R); S.vb.t[0] = -0.08f; S.vb.t[1] = 0.03f; S.vb.t[2] = -cfg.motion_z;  // camera moves +z
    make_view(S, W, S.va, 0.f, true);
    make_view(S, W, S.vb, 4.f, false);
    fill_frame(S, S.va, S.FA); fill_frame(S, S.vb, S.FB);
    fill_keyframe(S, S.va, S.FA, S.KA); fill_keyframe(S, S.vb, S.FB, S.KB);
    // one map point per world point, created from its observation in view A
    std::vector<int> idxA(cfg.npoints, -1);
    for (size_t i = 0; i < S.va.point.size(); i++)
        if (S.va.point[i] >= 0) idxA[S.va.point[i]] = (int)i;
    const float *Oa = S.KA.Ow.d.data();
    for (int j = 0; j < cfg.npoints; j++) {
        std::unique_ptr<MapPoint> mp(new MapPoint());
        mp->id = j;
        mp->pos = FMat(3, 1); mp->normal = FMat(3, 1);
        float n[3], len = 0;
        for (int k = 0; k < 3; k++) { mp->pos.at<float>(k, 0) = W[j].p[k]; n[k] = W[j].p[k] - Oa[k]; len += n[k] * n[k]; }
        len = sqrtf(len);
        for (int k = 0; k < 3; k++) mp->normal.at<float>(k, 0) = n[k] / len;
        mp->descriptor.create(1, 32);
        memcpy(mp->descriptor.data, W[j].code, 32);
        for (int k = S.pick(8); k > 0; k--) { const int bit = S.pick(256); mp->descriptor.data[bit >> 3] ^= (uint8_t)(1u << (bit & 7)); }
        const int lvl = idxA[j] >= 0 ? S.va.keys[idxA[j]].octave : W[j].levelA;
        mp->mfMaxDistance = len * S.scale[lvl];
        mp->mfMinDistance = mp->mfMaxDistance / S.scale[7];
        mp->bad = S.chance(0.03f);
        mp->nObs = S.chance(0.85f) ? 2 : 0;
        S.mps.push_back(std::move(mp));
    }
    return sp;
}

// attach map points to the keypoints of a view (fraction `p` of the keypoints that show a world point)
template <class T> static void attach(Scene &S, const View &V, T &f, float p, KeyFrame *asKF)
{
    for (size_t i = 0; i < V.point.size(); i++) {
        if (V.point[i] < 0 || !S.chance(p)) continue;
        MapPoint *mp = S.mps[V.point[i]].get();
        f.mvpMapPoints[i] = mp;
        if (asKF) mp->observations[asKF] = i;
    }
}

static int mpid(const MapPoint *p) { return p ? p->id : -1; }
static std::vector<int> ids(const std::vector<MapPoint *> &v)
{
    std::vector<int> o;
    for (const MapPoint *p : v) o.push_back(mpid(p));
    return o;
}

static int g_fail = 0;
static void report(const std::string &name0, bool same, int n_gpu, int n_cpu, int min_expected, const std::string &extra = "")
{
    const std::string name = name0 + (g_device_frames ? " [device frames]" : "");
    const bool ok = same && n_gpu == n_cpu && n_gpu >= min_expected;
    printf("%-44s gpu %5d  cpu %5d  %s %s\n", name.c_str(), n_gpu, n_cpu, ok ? "OK" : "FAIL", extra.c_str());
    if (!ok) g_fail++;
}

// state of all map points and of a keyframe's slots, for the Fuse scenarios
static std::vector<int> world_state(Scene &S, KeyFrame &K)
{
    std::vector<int> o = ids(K.mvpMapPoints);
    for (auto &mp : S.mps) { o.push_back(mp->bad ? 1 : 0); o.push_back(mp->nObs); o.push_back(mp->GetIndexInKeyFrame(&K)); }
    return o;
}

static void prefill_current(Scene &S, const View &V, Frame &F)
{
    // some keypoints of the current frame already hold a map point: observed ones block the keypoint, unobserved
    // ones (just created, no observation yet) do not
    for (size_t i = 0; i < V.point.size(); i++) {
        if (V.point[i] < 0) continue;
        if (S.chance(0.12f)) F.mvpMapPoints[i] = S.mps[V.point[i]].get();
    }
}

static void local_map_points(Scene &S, std::vector<MapPoint *> &out)
{
    // Tracking::SearchLocalPoints fills the tracking fields from Frame::isInFrustum (src/Frame.cc:320-386)
    const View &V = S.vb;
    for (auto &mpp : S.mps) {
        MapPoint *mp = mpp.get();
        const float *p = mp->pos.d.data();
        const float x = V.R[0] * p[0] + V.R[1] * p[1] + V.R[2] * p[2] + V.t[0];
        const float y = V.R[3] * p[0] + V.R[4] * p[1] + V.R[5] * p[2] + V.t[1];
        const float z = V.R[6] * p[0] + V.R[7] * p[1] + V.R[8] * p[2] + V.t[2];
        mp->mbTrackInView = false;
        if (z > 0.f) {
            const float u = S.fx * x / z + S.cx, v = S.fy * y / z + S.cy;
            if (u >= S.bounds[0] && u <= S.bounds[2] && v >= S.bounds[1] && v <= S.bounds[3] && S.chance(0.92f)) {
                const float *Ob = S.KB.Ow.d.data();
                const float d = sqrtf((p[0] - Ob[0]) * (p[0] - Ob[0]) + (p[1] - Ob[1]) * (p[1] - Ob[1]) + (p[2] - Ob[2]) * (p[2] - Ob[2]));
                mp->mbTrackInView = true;
                mp->mTrackProjX = u; mp->mTrackProjY = v; mp->mTrackProjXR = u - S.bf / z;
                mp->mnTrackScaleLevel = mp->PredictScale(d, &S.FB);
                mp->mTrackViewCos = S.chance(0.5f) ? 0.9995f : 0.99f;
            }
        }
        out.push_back(mp);
    }
}

static FMat sim3_of(const View &V, float s)
{
    FMat T(4, 4);
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T.at<float>(i, j) = s * V.R[3 * i + j]; T.at<float>(i, 3) = s * V.t[i]; }
    T.at<float>(3, 3) = 1.f;
    return T;
}

// x_a = R12 x_b + t12 between the two views; F12 = K^-T [t12]x R12 K^-1 (LocalMapping::ComputeF12)
static void relative(const Scene &S, FMat &R12, FMat &t12, FMat &F12)
{
    const View &A = S.va, &B = S.vb;
    double R[9], t[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += (double)A.R[3 * i + k] * B.R[3 * j + k];  // Ra Rb^T
            R[3 * i + j] = a;
        }
    for (int i = 0; i < 3; i++) t[i] = A.t[i] - (R[3 * i] * B.t[0] + R[3 * i + 1] * B.t[1] + R[3 * i + 2] * B.t[2]);
    R12 = FMat(3, 3); t12 = FMat(3, 1); F12 = FMat(3, 3);
    for (int i = 0; i < 9; i++) R12.d[i] = (float)R[i];
    for (int i = 0; i < 3; i++) t12.d[i] = (float)t[i];
    const double tx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    const double Ki[9] = {1.0 / S.fx, 0, -S.cx / S.fx, 0, 1.0 / S.fy, -S.cy / S.fy, 0, 0, 1};
    double E[9], M[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += tx[3 * i + k] * R[3 * k + j]; E[3 * i + j] = a; }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += Ki[3 * k + i] * E[3 * k + j]; M[3 * i + j] = a; }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { double a = 0; for (int k = 0; k < 3; k++) a += M[3 * i + k] * Ki[3 * k + j]; F12.d[3 * i + j] = (float)a; }
}

static void run_scenarios()
{
    using ORB_SLAM2::ORBmatcher;
    char name[128];

    // 1. SearchByProjection(Frame&, vector<MapPoint*>&, th)
    for (int v = 0; v < 4; v++) {
        Cfg c; c.seed = 10 + v; c.stereo = v != 1; c.distorted = v == 3;
        const float th = v == 0 ? 1.f : (v == 1 ? 3.f : 5.f);
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            prefill_current(S, S.vb, S.FB);
            std::vector<MapPoint *> pts;
            local_map_points(S, pts);
            if (k == 0) { ORBmatcher m(0.8f); n[k] = m.SearchByProjection(S.FB, pts, th); }
            else n[k] = orc_frames::SearchByProjection(S.FB, pts, th, 0.8f);
            out[k] = ids(S.FB.mvpMapPoints);
        }
        snprintf(name, sizeof name, "SearchByProjection(F,MapPoints) th=%g%s%s", th, c.stereo ? " stereo" : " mono", c.distorted ? " dist" : "");
        report(name, out[0] == out[1], n[0], n[1], 300);
    }

    // 2. SearchByProjection(Frame &Cur, const Frame &Last, th, bMono)
    for (int v = 0; v < 5; v++) {
        Cfg c; c.seed = 20 + v; c.stereo = v != 2;
        c.motion_z = v == 0 ? 0.3f : (v == 1 ? -0.3f : 0.02f);  // forward / backward / neither
        const bool bMono = v == 2;
        const float th = v == 3 ? 7.f : 15.f;
        const bool ori = v != 4;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            attach(S, S.va, S.FA, 0.8f, nullptr);
            for (int i = 0; i < S.FA.N; i++) S.FA.mvbOutlier[i] = S.chance(0.05f);
            prefill_current(S, S.vb, S.FB);
            if (k == 0) { ORBmatcher m(0.9f, ori); n[k] = m.SearchByProjection(S.FB, S.FA, th, bMono); }
            else n[k] = orc_frames::SearchByProjection(S.FB, S.FA, th, bMono, ori);
            out[k] = ids(S.FB.mvpMapPoints);
        }
        snprintf(name, sizeof name, "SearchByProjection(Cur,Last) th=%g dz=%g%s%s", th, c.motion_z, bMono ? " mono" : "", ori ? "" : " noOri");
        report(name, out[0] == out[1], n[0], n[1], 200);
    }

    // 3. SearchByProjection(Frame &Cur, KeyFrame*, sAlreadyFound, th, ORBdist)
    for (int v = 0; v < 3; v++) {
        Cfg c; c.seed = 30 + v; c.distorted = v == 2;
        const float th = v == 0 ? 10.f : 3.f;
        const int ORBdist = v == 0 ? 100 : 64;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            attach(S, S.va, S.KA, 0.85f, &S.KA);
            std::set<MapPoint *> found;
            for (size_t i = 0; i < S.vb.point.size(); i++)
                if (S.vb.point[i] >= 0 && S.chance(0.15f)) { found.insert(S.mps[S.vb.point[i]].get()); S.FB.mvpMapPoints[i] = S.mps[S.vb.point[i]].get(); }
            if (k == 0) { ORBmatcher m(0.9f, true); n[k] = m.SearchByProjection(S.FB, &S.KA, found, th, ORBdist); }
            else n[k] = orc_frames::SearchByProjection(S.FB, &S.KA, found, th, ORBdist, true);
            out[k] = ids(S.FB.mvpMapPoints);
        }
        snprintf(name, sizeof name, "SearchByProjection(Cur,KF,found) th=%g dist=%d%s", th, ORBdist, c.distorted ? " dist" : "");
        report(name, out[0] == out[1], n[0], n[1], 150);
    }

    // 4. SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th)
    for (int v = 0; v < 3; v++) {
        Cfg c; c.seed = 40 + v; c.distorted = v >= 1;
        const int th = v == 2 ? 4 : 10;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            std::vector<MapPoint *> pts, matched(S.KB.N, nullptr);
            for (auto &mp : S.mps) pts.push_back(mp.get());
            for (size_t i = 0; i < S.vb.point.size(); i++)
                if (S.vb.point[i] >= 0 && S.chance(0.2f)) matched[i] = S.mps[S.vb.point[i]].get();
            const FMat Scw = sim3_of(S.vb, 1.3f);
            if (k == 0) { ORBmatcher m(0.75f, true); n[k] = m.SearchByProjection(&S.KB, Scw, pts, matched, th); }
            else n[k] = orc_frames::SearchByProjection(&S.KB, Scw, pts, matched, th);
            out[k] = ids(matched);
        }
        snprintf(name, sizeof name, "SearchByProjection(KF,Scw) th=%d%s", th, c.distorted ? " dist" : "");
        report(name, out[0] == out[1], n[0], n[1], 150);
    }

    // 5. SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)
    for (int v = 0; v < 2; v++) {
        Cfg c; c.seed = 50 + v;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            attach(S, S.va, S.KA, 0.85f, &S.KA);
            std::vector<MapPoint *> matches;
            if (k == 0) { ORBmatcher m(0.7f + 0.2f * v, v == 0); n[k] = m.SearchByBoW(&S.KA, S.FB, matches); }
            else n[k] = orc_frames::SearchByBoW(&S.KA, S.FB, matches, 0.7f + 0.2f * v, v == 0);
            out[k] = ids(matches);
        }
        snprintf(name, sizeof name, "SearchByBoW(KF,F) ratio=%g", 0.7 + 0.2 * v);
        report(name, out[0] == out[1], n[0], n[1], 100);
    }

    // 6. SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)
    for (int v = 0; v < 2; v++) {
        Cfg c; c.seed = 60 + v;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            attach(S, S.va, S.KA, 0.85f, &S.KA);
            attach(S, S.vb, S.KB, 0.8f, &S.KB);
            std::vector<MapPoint *> m12;
            if (k == 0) { ORBmatcher m(0.8f, v == 0); n[k] = m.SearchByBoW(&S.KA, &S.KB, m12); }
            else n[k] = orc_frames::SearchByBoW(&S.KA, &S.KB, m12, 0.8f, v == 0);
            out[k] = ids(m12);
        }
        report(std::string("SearchByBoW(KF,KF)") + (v ? " noOri" : ""), out[0] == out[1], n[0], n[1], 100);
    }

    // 7. SearchForInitialization(Frame&, Frame&, vbPrevMatched, vnMatches12, windowSize)
    for (int v = 0; v < 2; v++) {
        Cfg c; c.seed = 70 + v; c.stereo = false; c.motion_z = 0.05f;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        std::vector<float> pv[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            std::vector<orbcv::Point2f> prev(S.FA.N);
            for (int i = 0; i < S.FA.N; i++) prev[i] = S.FA.mvKeysUn[i].pt;
            std::vector<int> m12;
            const int win = v == 0 ? 100 : 40;
            if (k == 0) { ORBmatcher m(0.9f, true); n[k] = m.SearchForInitialization(S.FA, S.FB, prev, m12, win); }
            else n[k] = orc_frames::SearchForInitialization(S.FA, S.FB, prev, m12, win, 0.9f, true);
            out[k] = m12;
            for (const orbcv::Point2f &p : prev) { pv[k].push_back(p.x); pv[k].push_back(p.y); }
        }
        report(std::string("SearchForInitialization window=") + (v ? "40" : "100"), out[0] == out[1] && pv[0] == pv[1], n[0], n[1], 100);
    }

    // 8. SearchForTriangulation(KeyFrame*, KeyFrame*, F12, vMatchedPairs, bOnlyStereo)
    for (int v = 0; v < 3; v++) {
        Cfg c; c.seed = 80 + v; c.stereo = v != 0; c.motion_z = 0.4f;
        const bool onlyStereo = v == 2;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<std::pair<size_t, size_t> > out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            attach(S, S.va, S.KA, 0.4f, &S.KA);
            attach(S, S.vb, S.KB, 0.4f, &S.KB);
            FMat R12, t12, F12;
            relative(S, R12, t12, F12);
            if (k == 0) { ORBmatcher m(0.6f, v != 1); n[k] = m.SearchForTriangulation(&S.KA, &S.KB, F12, out[k], onlyStereo); }
            else n[k] = orc_frames::SearchForTriangulation(&S.KA, &S.KB, F12, out[k], onlyStereo, v != 1);
        }
        snprintf(name, sizeof name, "SearchForTriangulation%s%s", c.stereo ? " stereo" : " mono", onlyStereo ? " onlyStereo" : "");
        report(name, out[0] == out[1], n[0], n[1], 60);
    }

    // 9. Fuse(KeyFrame*, vpMapPoints, th)
    for (int v = 0; v < 3; v++) {
        Cfg c; c.seed = 90 + v; c.stereo = v != 1; c.distorted = v == 2;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            attach(S, S.vb, S.KB, 0.4f, &S.KB);
            // the keyframe's slots mostly hold the right point; make some hold another one so that Replace runs
            for (int i = 0; i < S.KB.N; i++)
                if (S.KB.mvpMapPoints[i] && S.chance(0.3f)) {
                    MapPoint *other = S.mps[S.pick((int)S.mps.size())].get();
                    if (!other->IsInKeyFrame(&S.KB)) {
                        S.KB.mvpMapPoints[i]->observations.erase(&S.KB);
                        S.KB.mvpMapPoints[i] = other; other->observations[&S.KB] = i;
                    }
                }
            std::vector<MapPoint *> pts;
            for (auto &mp : S.mps) { pts.push_back(S.chance(0.05f) ? nullptr : mp.get()); if (S.chance(0.03f)) pts.push_back(mp.get()); }
            const float th = v == 0 ? 3.f : 5.f;
            if (k == 0) { ORBmatcher m; n[k] = m.Fuse(&S.KB, pts, th); }
            else n[k] = orc_frames::Fuse(&S.KB, pts, th);
            out[k] = world_state(S, S.KB);
        }
        snprintf(name, sizeof name, "Fuse(KF,MapPoints)%s%s", c.stereo ? " stereo" : " mono", c.distorted ? " dist" : "");
        report(name, out[0] == out[1], n[0], n[1], 150);
    }

    // 10. Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint)
    for (int v = 0; v < 2; v++) {
        Cfg c; c.seed = 100 + v; c.distorted = v == 1;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            attach(S, S.vb, S.KB, 0.35f, &S.KB);
            for (int i = 0; i < S.KB.N; i++)  // some slots hold a different point: those become replacements
                if (S.KB.mvpMapPoints[i] && S.chance(0.4f)) {
                    MapPoint *other = S.mps[S.pick((int)S.mps.size())].get();
                    if (!other->IsInKeyFrame(&S.KB)) {
                        S.KB.mvpMapPoints[i]->observations.erase(&S.KB);
                        S.KB.mvpMapPoints[i] = other; other->observations[&S.KB] = i;
                    }
                }
            std::vector<MapPoint *> pts;
            for (auto &mp : S.mps) pts.push_back(mp.get());
            std::vector<MapPoint *> repl(pts.size(), nullptr);
            const FMat Scw = sim3_of(S.vb, 0.8f);
            if (k == 0) { ORBmatcher m; n[k] = m.Fuse(&S.KB, Scw, pts, 4.f, repl); }
            else n[k] = orc_frames::Fuse(&S.KB, Scw, pts, 4.f, repl);
            out[k] = world_state(S, S.KB);
            const std::vector<int> r = ids(repl);
            out[k].insert(out[k].end(), r.begin(), r.end());
        }
        report(std::string("Fuse(KF,Scw)") + (c.distorted ? " dist" : ""), out[0] == out[1], n[0], n[1], 150);
    }

    // 11. SearchBySim3(KeyFrame*, KeyFrame*, vpMatches12, s12, R12, t12, th)
    for (int v = 0; v < 2; v++) {
        Cfg c; c.seed = 110 + v; c.distorted = v == 1;
        std::unique_ptr<Scene> s[2] = {build(c), build(c)};
        int n[2] = {0, 0};
        std::vector<int> out[2];
        for (int k = 0; k < 2; k++) {
            Scene &S = *s[k];
            attach(S, S.va, S.KA, 0.8f, &S.KA);
            attach(S, S.vb, S.KB, 0.8f, &S.KB);
            FMat R12, t12, F12;
            relative(S, R12, t12, F12);
            std::vector<MapPoint *> m12(S.KA.N, nullptr);
            for (int i = 0; i < S.KA.N; i++)  // matches found before (by SearchByBoW in the reference's loop closing)
                if (S.KA.mvpMapPoints[i] && S.KA.mvpMapPoints[i]->IsInKeyFrame(&S.KB) && S.chance(0.3f)) m12[i] = S.KA.mvpMapPoints[i];
            const float s12 = 1.0f;
            if (k == 0) { ORBmatcher m(0.75f, true); n[k] = m.SearchBySim3(&S.KA, &S.KB, m12, s12, R12, t12, 7.5f); }
            else n[k] = orc_frames::SearchBySim3(&S.KA, &S.KB, m12, s12, R12, t12, 7.5f);
            out[k] = ids(m12);
        }
        report(std::string("SearchBySim3") + (c.distorted ? " dist" : ""), out[0] == out[1], n[0], n[1], 100);
    }

    // ComputeThreeMaxima(vector<int>*, L, ...) against the oracle's rule, and the copyable matcher
    {
        std::mt19937 rng(5);
        bool same = true;
        for (int it = 0; it < 500; it++) {
            std::vector<int> h[30];
            int sizes[30];
            for (int i = 0; i < 30; i++) { sizes[i] = (rng() % 3 == 0) ? (int)(rng() % 9) : 0; h[i].assign(sizes[i], 0); }
            if (it == 0) { for (int i = 0; i < 30; i++) { sizes[i] = 0; h[i].clear(); } }
            if (it == 1) { sizes[0] = 100; h[0].assign(100, 0); for (int i = 1; i < 30; i++) { sizes[i] = 5; h[i].assign(5, 0); } }
            int a1 = -1, a2 = -1, a3 = -1, b1, b2, b3;
            ORBmatcher::ComputeThreeMaxima(h, 30, a1, a2, a3);
            orc_three_maxima(sizes, 30, &b1, &b2, &b3);
            same = same && a1 == b1 && a2 == b2 && a3 == b3;
        }
        ORBmatcher m1(0.7f, false);
        ORBmatcher m2 = m1;  // a value type like the reference's
        ORBmatcher m3;
        m3 = m1;
        report("ComputeThreeMaxima + copies", same && m2.LastError().empty() && m3.LastError().empty(), 1, 1, 1);
    }
}

int main()
{
    {
        ORB_SLAM2::ORBmatcher probe;
        if (!probe.LastError().empty()) { fprintf(stderr, "no GPU matcher: %s\n", probe.LastError().c_str()); return 2; }
    }
    g_device_frames = false;
    run_scenarios();
    g_device_frames = true;
    run_scenarios();
    printf("%s\n", g_fail ? "FAILED" : "ALL OK");
    return g_fail ? 1 : 0;
}
