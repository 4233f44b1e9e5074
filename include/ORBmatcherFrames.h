/*
 * ORBmatcherFrames.h -- the eleven search methods of ORB_SLAM2::ORBmatcher with the reference's own
 * signatures (reference: /root/reference/src/ORB_SLAM2/include/ORBmatcher.h:48-83), as member templates over
 * the Frame / KeyFrame / MapPoint types of the program that includes this file.  ORBmatcher.h includes it;
 * nothing else is needed to compile src/Tracking.cc, src/LocalMapping.cc and src/LoopClosing.cc unchanged.
 *
 * Each method keeps the host arithmetic of its reference counterpart (pose projection, frustum and
 * scale-invariance gates, predicted level, search radius), turns every surviving map point into one window
 * query, runs the descriptor search on the GPU through the array entry points of orbgpu.h, and writes the
 * result back into the containers the reference writes (mvpMapPoints, vpMatched, vpMatches12, ...).
 * Reference line numbers are those of /root/reference/src/ORB_SLAM2/src/ORBmatcher.cc.
 *
 * Members used.  Frame: N, mvKeys, mvKeysUn, mDescriptors, mvpMapPoints, mvbOutlier, mvuRight, mvScaleFactors,
 * mFeatVec, mTcw, fx, fy, cx, cy, mbf, mb, mnMinX, mnMinY, mnMaxX, mnMaxY.  KeyFrame: N, mvKeysUn, mvuRight,
 * mDescriptors, mFeatVec, mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2, fx, fy, cx, cy, mbf, mnMinX ... mnMaxY,
 * mfGridElementWidthInv, mfGridElementHeightInv, GetMapPointMatches(), GetMapPoints(), GetMapPoint(),
 * AddMapPoint(), GetRotation(), GetTranslation(), GetCameraCenter(), IsInImage().  MapPoint: mbTrackInView,
 * mnTrackScaleLevel, mTrackViewCos, mTrackProjX, mTrackProjY, mTrackProjXR, isBad(), Observations(),
 * GetDescriptor(), GetWorldPos(), GetNormal(), GetMinDistanceInvariance(), GetMaxDistanceInvariance(),
 * PredictScale(), IsInKeyFrame(), GetIndexInKeyFrame(), AddObservation(), Replace().  Matrices are read through
 * `template at<float>(row, col)`, descriptor rows through `ptr(row)`: cv::Mat has both.
 *
 * Matrix arithmetic.  The reference writes `Rcw*x3Dw+tcw`, `-Rcw.t()*tcw`, `cv::norm(PO)`, `PO.dot(Pn)` on
 * cv::Mat; OpenCV evaluates 3x3 float products with float multiply-adds summed left to right, norms and dot
 * products with double accumulators.  The helpers below do the same; like every OpenCV-defined detail of this
 * path this is restated from OpenCV's published algorithm, not checked against a build of it.
 */
#ifndef ORBGPU_ORBMATCHER_FRAMES_H
#define ORBGPU_ORBMATCHER_FRAMES_H

#ifndef ORBGPU_ORBMATCHER_H
#error "include ORBmatcher.h, which includes this file after the class definition"
#endif

#include <cmath>
#include <cstring>
#include <set>
#include <type_traits>

namespace ORB_SLAM2 {

/* Where the 64x48 grid of a KeyFrame was built.  A KeyFrame copies the grid of the Frame it is made from, and
 * that grid was filled with the Frame's float image bounds; the KeyFrame itself only keeps the bounds truncated
 * to int (include/KeyFrame.h:185).  With an undistorted camera both are the same integers and the default below
 * is exact.  With a distorted camera (bounds like -7.6) specialise this for the program's KeyFrame type to return
 * the Frame's bounds (they are static members of Frame there: `x = Frame::mnMinX; y = Frame::mnMinY;`). */
template <class KeyFrameT>
struct KeyFrameGridOrigin {
    static void get(const KeyFrameT &kf, float &x, float &y)
    {
        x = (float)kf.mnMinX;
        y = (float)kf.mnMinY;
    }
};

namespace orbgpu_detail {

/* A Frame / KeyFrame type that carries a device-resident copy of its arrays exposes it as a member
 * `mpDeviceFrame`: anything whose `->Handle()` is the orbgpu_frame (a std::shared_ptr<DeviceFrame> of
 * FrameDevice.h, or a plain pointer).  The searches then read the train side where it lies on the GPU; types
 * without the member, and frames whose pointer is null, take the array entry points. */
template <class F, class = void> struct HasDeviceFrame : std::false_type {};
template <class F> struct HasDeviceFrame<F, decltype((void)std::declval<const F &>().mpDeviceFrame, void())> : std::true_type {};
template <class F> inline const orbgpu_frame *device_frame(const F &f, std::true_type) { return f.mpDeviceFrame ? f.mpDeviceFrame->Handle() : nullptr; }
template <class F> inline const orbgpu_frame *device_frame(const F &, std::false_type) { return nullptr; }
template <class F> inline const orbgpu_frame *device_frame(const F &f) { return device_frame(f, HasDeviceFrame<F>()); }

struct V3 { float x, y, z; };
struct SE3 { float R[9]; float t[3]; };  // x_c = R x_w + t, R row-major

template <class M> inline float el(const M &m, int r, int c) { return m.template at<float>(r, c); }
template <class M> inline V3 vec3(const M &m) { V3 v = {el(m, 0, 0), el(m, 1, 0), el(m, 2, 0)}; return v; }

template <class M> inline SE3 se3_from_4x4(const M &T)  // mTcw.rowRange(0,3).colRange(0,3) and .col(3)
{
    SE3 P;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) P.R[3 * r + c] = el(T, r, c);
        P.t[r] = el(T, r, 3);
    }
    return P;
}
template <class MR, class MT> inline SE3 se3_from_Rt(const MR &R, const MT &t)
{
    SE3 P;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) P.R[3 * r + c] = el(R, r, c);
        P.t[r] = el(t, r, 0);
    }
    return P;
}
// Scw = [s R | s t]: scw = sqrt(row0 . row0) (double accumulation), then R = sR / scw, t = st / scw, each element
// multiplied by the float value of 1 / scw (:299-302, :1037-1040)
template <class M> inline SE3 se3_from_sim3(const M &S)
{
    double ss = 0.0;
    for (int c = 0; c < 3; c++) ss += (double)el(S, 0, c) * (double)el(S, 0, c);
    const float scw = (float)std::sqrt(ss);
    const float inv = (float)(1.0 / (double)scw);
    SE3 P;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) P.R[3 * r + c] = el(S, r, c) * inv;
        P.t[r] = el(S, r, 3) * inv;
    }
    return P;
}
inline V3 apply(const SE3 &P, const V3 &p)  // R p + t
{
    V3 o;
    o.x = ((P.R[0] * p.x + P.R[1] * p.y) + P.R[2] * p.z) + P.t[0];
    o.y = ((P.R[3] * p.x + P.R[4] * p.y) + P.R[5] * p.z) + P.t[1];
    o.z = ((P.R[6] * p.x + P.R[7] * p.y) + P.R[8] * p.z) + P.t[2];
    return o;
}
inline V3 centre(const SE3 &P)  // -R^T t
{
    V3 o;
    o.x = -((P.R[0] * P.t[0] + P.R[3] * P.t[1]) + P.R[6] * P.t[2]);
    o.y = -((P.R[1] * P.t[0] + P.R[4] * P.t[1]) + P.R[7] * P.t[2]);
    o.z = -((P.R[2] * P.t[0] + P.R[5] * P.t[1]) + P.R[8] * P.t[2]);
    return o;
}
inline V3 sub(const V3 &a, const V3 &b) { V3 o = {a.x - b.x, a.y - b.y, a.z - b.z}; return o; }
inline float norm3(const V3 &v) { return (float)std::sqrt((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z); }
inline double dot3(const V3 &a, const V3 &b) { return (double)a.x * b.x + (double)a.y * b.y + (double)a.z * b.z; }

// rows of an N x 32 descriptor matrix as one contiguous block
template <class M> inline const uint8_t *descriptor_block(const M &m, int n, std::vector<uint8_t> &tmp)
{
    if (n <= 0) return nullptr;
    if ((size_t)m.step == 32) return m.ptr(0);
    tmp.resize((size_t)n * 32);
    for (int i = 0; i < n; i++) std::memcpy(&tmp[(size_t)i * 32], m.ptr(i), 32);
    return tmp.data();
}
template <class MP> inline void push_descriptor(MP *pMP, std::vector<uint8_t> &qd)
{
    const auto d = pMP->GetDescriptor();
    const uint8_t *p = d.ptr(0);
    qd.insert(qd.end(), p, p + 32);
}
template <class KP> inline const orbgpu_keypoint *as_keypoints(const std::vector<KP> &v)
{
    static_assert(sizeof(KP) == sizeof(orbgpu_keypoint), "cv::KeyPoint layout");
    return reinterpret_cast<const orbgpu_keypoint *>(v.data());
}
inline orbgpu_wquery window(float u, float v, float r, int minLevel, int maxLevel, float angle, float ur, int flags)
{
    orbgpu_wquery w;
    w.u = u; w.v = v; w.r = r; w.minLevel = minLevel; w.maxLevel = maxLevel; w.angle = angle; w.ur = ur; w.flags = flags;
    return w;
}
// the grid of a KeyFrame: built at the Frame's origin, queried at the KeyFrame's truncated one (src/KeyFrame.cc:585-624)
template <class KF> inline void keyframe_geometry(const KF &kf, float geom6[6], float bounds4[4])
{
    KeyFrameGridOrigin<KF>::get(kf, geom6[0], geom6[1]);
    geom6[2] = (float)kf.mnMinX; geom6[3] = (float)kf.mnMinY;
    geom6[4] = kf.mfGridElementWidthInv; geom6[5] = kf.mfGridElementHeightInv;
    bounds4[0] = (float)kf.mnMinX; bounds4[1] = (float)kf.mnMinY; bounds4[2] = (float)kf.mnMaxX; bounds4[3] = (float)kf.mnMaxY;
}
// the two sorted FeatureVectors walked together (:175-264): common nodes become segments of side-2 indices
template <class FV, class Keep>
inline void intersect_feature_vectors(const FV &fv1, const FV &fv2, Keep keep1, std::vector<int32_t> &qIndex, std::vector<int32_t> &qSeg,
                                      std::vector<int32_t> &segStart, std::vector<int32_t> &segIndex)
{
    segStart.assign(1, 0);
    typename FV::const_iterator it1 = fv1.begin(), it2 = fv2.begin();
    while (it1 != fv1.end() && it2 != fv2.end()) {
        if (it1->first == it2->first) {
            const int32_t s = (int32_t)segStart.size() - 1;
            for (size_t k = 0; k < it2->second.size(); k++) segIndex.push_back((int32_t)it2->second[k]);
            segStart.push_back((int32_t)segIndex.size());
            for (size_t k = 0; k < it1->second.size(); k++) {
                const size_t i = it1->second[k];
                if (keep1(i)) { qIndex.push_back((int32_t)i); qSeg.push_back(s); }
            }
            ++it1; ++it2;
        } else if (it1->first < it2->first) {
            it1 = fv1.lower_bound(it2->first);
        } else {
            it2 = fv2.lower_bound(it1->first);
        }
    }
    if (segIndex.empty()) segIndex.push_back(0);  // never a null buffer; segStart bounds its use
}

}  // namespace orbgpu_detail

// ---- SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th)   :45-129 ----------------------
template <class FrameT, class MapPointT>
int ORBmatcher::SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th)
{
    namespace D = orbgpu_detail;
    if (!handle_) return 0;
    const bool bFactor = th != 1.0;
    std::vector<orbgpu_wquery> q;
    std::vector<uint8_t> qd;
    std::vector<int32_t> qIndex;
    for (size_t iMP = 0; iMP < vpMapPoints.size(); iMP++) {
        MapPointT *pMP = vpMapPoints[iMP];
        if (!pMP->mbTrackInView || pMP->isBad()) continue;  // :54-58
        const int level = pMP->mnTrackScaleLevel;
        float r = pMP->mTrackViewCos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos, :131-137
        if (bFactor) r *= th;
        // a match marks its keypoint occupied for later map points only if the point has observations (:87-89)
        q.push_back(D::window(pMP->mTrackProjX, pMP->mTrackProjY, r * F.mvScaleFactors[level], level - 1, level, 0.0f,
                              pMP->mTrackProjXR, 1 | (pMP->Observations() > 0 ? 2 : 0)));
        D::push_descriptor(pMP, qd);
        qIndex.push_back((int32_t)iMP);
    }
    if (q.empty() || F.N == 0) return 0;
    std::vector<unsigned char> occ((size_t)F.N);
    for (int t = 0; t < F.N; t++) occ[t] = F.mvpMapPoints[t] && F.mvpMapPoints[t]->Observations() > 0;
    const float b[4] = {(float)F.mnMinX, (float)F.mnMinY, (float)F.mnMaxX, (float)F.mnMaxY};
    std::vector<uint8_t> tmp;
    std::vector<int32_t> assign((size_t)F.N, -1);
    int32_t nmatches = 0;
    const orbgpu_frame *df = D::device_frame(F);
    if (check(df ? orbgpu_search_by_projection_frame(handle_, ORBGPU_MODE_LOCAL_MAP, q.data(), qd.data(), (int32_t)q.size(), df, occ.data(),
                                                     assign.data(), TH_HIGH, mfNNratio, 0, &nmatches)
                 : orbgpu_search_by_projection(handle_, ORBGPU_MODE_LOCAL_MAP, q.data(), qd.data(), (int32_t)q.size(),
                                               D::as_keypoints(F.mvKeysUn), D::descriptor_block(F.mDescriptors, F.N, tmp),
                                               F.mvuRight.data(), F.N, b, occ.data(), assign.data(), TH_HIGH, mfNNratio, 0, &nmatches)))
        return 0;
    for (int t = 0; t < F.N; t++)
        if (assign[t] >= 0) F.mvpMapPoints[t] = vpMapPoints[qIndex[assign[t]]];  // :123
    return nmatches;
}

// ---- SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono)   :1379-1521 ---------------
template <class FrameT>
int ORBmatcher::SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono)
{
    namespace D = orbgpu_detail;
    if (!handle_) return 0;
    const D::SE3 Tcw = D::se3_from_4x4(CurrentFrame.mTcw), Tlw = D::se3_from_4x4(LastFrame.mTcw);
    const D::V3 twc = D::centre(Tcw);
    const D::V3 tlc = D::apply(Tlw, twc);  // :1397
    const bool bForward = tlc.z > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.z > CurrentFrame.mb && !bMono;
    std::vector<orbgpu_wquery> q;
    std::vector<uint8_t> qd;
    std::vector<int32_t> qIndex;
    for (int i = 0; i < LastFrame.N; i++) {
        auto *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;
        const D::V3 x3Dc = D::apply(Tcw, D::vec3(pMP->GetWorldPos()));
        const float invzc = (float)(1.0 / x3Dc.z);
        if (invzc < 0) continue;
        const float u = CurrentFrame.fx * x3Dc.x * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * x3Dc.y * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        const int nLastOctave = LastFrame.mvKeys[i].octave;
        const float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
        int minLevel, maxLevel;  // :1436-1441
        if (bForward) { minLevel = nLastOctave; maxLevel = -1; }
        else if (bBackward) { minLevel = 0; maxLevel = nLastOctave; }
        else { minLevel = nLastOctave - 1; maxLevel = nLastOctave + 1; }
        const float ur = u - CurrentFrame.mbf * invzc;  // :1460
        q.push_back(D::window(u, v, radius, minLevel, maxLevel, LastFrame.mvKeysUn[i].angle, ur,
                              1 | (pMP->Observations() > 0 ? 2 : 0)));  // :1454-1456 as seen by later map points
        D::push_descriptor(pMP, qd);
        qIndex.push_back(i);
    }
    const int N2 = CurrentFrame.N;
    if (q.empty() || N2 == 0) return 0;
    std::vector<unsigned char> occ((size_t)N2);
    for (int t = 0; t < N2; t++) occ[t] = CurrentFrame.mvpMapPoints[t] && CurrentFrame.mvpMapPoints[t]->Observations() > 0;
    const float b[4] = {(float)CurrentFrame.mnMinX, (float)CurrentFrame.mnMinY, (float)CurrentFrame.mnMaxX, (float)CurrentFrame.mnMaxY};
    std::vector<uint8_t> tmp;
    std::vector<int32_t> assign((size_t)N2, -1);
    int32_t nmatches = 0;
    const orbgpu_frame *df = D::device_frame(CurrentFrame);
    if (check(df ? orbgpu_search_by_projection_frame(handle_, ORBGPU_MODE_LAST_FRAME, q.data(), qd.data(), (int32_t)q.size(), df, occ.data(),
                                                     assign.data(), TH_HIGH, mfNNratio, mbCheckOrientation ? 1 : 0, &nmatches)
                 : orbgpu_search_by_projection(handle_, ORBGPU_MODE_LAST_FRAME, q.data(), qd.data(), (int32_t)q.size(),
                                               D::as_keypoints(CurrentFrame.mvKeysUn),
                                               D::descriptor_block(CurrentFrame.mDescriptors, N2, tmp), CurrentFrame.mvuRight.data(), N2, b,
                                               occ.data(), assign.data(), TH_HIGH, mfNNratio, mbCheckOrientation ? 1 : 0, &nmatches)))
        return 0;
    for (int t = 0; t < N2; t++) {
        if (assign[t] >= 0) CurrentFrame.mvpMapPoints[t] = LastFrame.mvpMapPoints[qIndex[assign[t]]];  // :1479
        else if (assign[t] == -2) CurrentFrame.mvpMapPoints[t] = nullptr;                            // :1513
    }
    return nmatches;
}

// ---- SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist)   :1523-1650 -------
template <class FrameT, class KeyFrameT, class MapPointT>
int ORBmatcher::SearchByProjection(FrameT &CurrentFrame, KeyFrameT *pKF, const std::set<MapPointT *> &sAlreadyFound, const float th,
                                   const int ORBdist)
{
    namespace D = orbgpu_detail;
    if (!handle_) return 0;
    const D::SE3 Tcw = D::se3_from_4x4(CurrentFrame.mTcw);
    const D::V3 Ow = D::centre(Tcw);
    const std::vector<MapPointT *> vpMPs = pKF->GetMapPointMatches();
    std::vector<orbgpu_wquery> q;
    std::vector<uint8_t> qd;
    std::vector<int32_t> qIndex;
    for (size_t i = 0; i < vpMPs.size(); i++) {
        MapPointT *pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        const D::V3 x3Dw = D::vec3(pMP->GetWorldPos());
        const D::V3 x3Dc = D::apply(Tcw, x3Dw);
        const float invzc = (float)(1.0 / x3Dc.z);  // no sign test here (:1553)
        const float u = CurrentFrame.fx * x3Dc.x * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * x3Dc.y * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        const float dist3D = D::norm3(D::sub(x3Dw, Ow));
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
        const float radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];
        q.push_back(D::window(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, pKF->mvKeysUn[i].angle, -1.0f, 3));
        D::push_descriptor(pMP, qd);
        qIndex.push_back((int32_t)i);
    }
    const int N2 = CurrentFrame.N;
    if (q.empty() || N2 == 0) return 0;
    std::vector<unsigned char> occ((size_t)N2);
    for (int t = 0; t < N2; t++) occ[t] = CurrentFrame.mvpMapPoints[t] != nullptr;  // :1592
    const float b[4] = {(float)CurrentFrame.mnMinX, (float)CurrentFrame.mnMinY, (float)CurrentFrame.mnMaxX, (float)CurrentFrame.mnMaxY};
    std::vector<uint8_t> tmp;
    std::vector<int32_t> assign((size_t)N2, -1);
    int32_t nmatches = 0;
    const orbgpu_frame *df = D::device_frame(CurrentFrame);
    if (check(df ? orbgpu_search_by_projection_frame(handle_, ORBGPU_MODE_KEYFRAME, q.data(), qd.data(), (int32_t)q.size(), df, occ.data(),
                                                     assign.data(), ORBdist, mfNNratio, mbCheckOrientation ? 1 : 0, &nmatches)
                 : orbgpu_search_by_projection(handle_, ORBGPU_MODE_KEYFRAME, q.data(), qd.data(), (int32_t)q.size(),
                                               D::as_keypoints(CurrentFrame.mvKeysUn),
                                               D::descriptor_block(CurrentFrame.mDescriptors, N2, tmp), nullptr, N2, b, occ.data(),
                                               assign.data(), ORBdist, mfNNratio, mbCheckOrientation ? 1 : 0, &nmatches)))
        return 0;
    for (int t = 0; t < N2; t++) {
        if (assign[t] >= 0) CurrentFrame.mvpMapPoints[t] = vpMPs[qIndex[assign[t]]];  // :1608
        else if (assign[t] == -2) CurrentFrame.mvpMapPoints[t] = nullptr;           // :1642
    }
    return nmatches;
}

// ---- SearchByProjection(KeyFrame *pKF, cv::Mat Scw, vpPoints, vpMatched, th)   :290-403 ---------------------
template <class KeyFrameT, class MatT, class MapPointT>
int ORBmatcher::SearchByProjection(KeyFrameT *pKF, MatT Scw, const std::vector<MapPointT *> &vpPoints, std::vector<MapPointT *> &vpMatched,
                                   int th)
{
    namespace D = orbgpu_detail;
    if (!handle_) return 0;
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
    const D::SE3 Tcw = D::se3_from_sim3(Scw);
    const D::V3 Ow = D::centre(Tcw);
    std::set<MapPointT *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPointT *>(nullptr));
    std::vector<orbgpu_wquery> q;
    std::vector<uint8_t> qd;
    std::vector<int32_t> qIndex;
    for (int iMP = 0, iend = (int)vpPoints.size(); iMP < iend; iMP++) {
        MapPointT *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        const D::V3 p3Dw = D::vec3(pMP->GetWorldPos());
        const D::V3 p3Dc = D::apply(Tcw, p3Dw);
        if (p3Dc.z < 0.0) continue;
        const float invz = 1 / p3Dc.z;
        const float x = p3Dc.x * invz, y = p3Dc.y * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const D::V3 PO = D::sub(p3Dw, Ow);
        const float dist = D::norm3(PO);
        if (dist < minDistance || dist > maxDistance) continue;
        if (D::dot3(PO, D::vec3(pMP->GetNormal())) < 0.5 * dist) continue;  // viewing angle below 60 degrees
        const int nPredictedLevel = pMP->PredictScale(dist, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        q.push_back(D::window(u, v, radius, nPredictedLevel - 1, nPredictedLevel, 0.0f, -1.0f, 3));
        D::push_descriptor(pMP, qd);
        qIndex.push_back(iMP);
    }
    const int N2 = pKF->N;
    if (q.empty() || N2 == 0) return 0;
    std::vector<unsigned char> occ((size_t)N2);
    for (int t = 0; t < N2; t++) occ[t] = vpMatched[t] != nullptr;  // :375
    float g[6], b[4];
    D::keyframe_geometry(*pKF, g, b);
    std::vector<uint8_t> tmp;
    std::vector<int32_t> assign((size_t)N2, -1);
    int32_t nmatches = 0;
    if (check(orbgpu_matcher_set_grid_geometry(handle_, g))) return 0;
    const orbgpu_frame *df = D::device_frame(*pKF);
    if (check(df ? orbgpu_search_by_projection_frame(handle_, ORBGPU_MODE_KEYFRAME, q.data(), qd.data(), (int32_t)q.size(), df, occ.data(),
                                                     assign.data(), TH_LOW, mfNNratio, 0, &nmatches)
                 : orbgpu_search_by_projection(handle_, ORBGPU_MODE_KEYFRAME, q.data(), qd.data(), (int32_t)q.size(),
                                               D::as_keypoints(pKF->mvKeysUn), D::descriptor_block(pKF->mDescriptors, N2, tmp), nullptr, N2,
                                               b, occ.data(), assign.data(), TH_LOW, mfNNratio, 0, &nmatches)))
        return 0;
    for (int t = 0; t < N2; t++)
        if (assign[t] >= 0) vpMatched[t] = vpPoints[qIndex[assign[t]]];  // :396
    return nmatches;
}

// ---- SearchByBoW(KeyFrame *pKF, Frame &F, vpMapPointMatches)   :159-288 -------------------------------------
template <class KeyFrameT, class FrameT, class MapPointT>
typename std::enable_if<!std::is_pointer<FrameT>::value, int>::type
ORBmatcher::SearchByBoW(KeyFrameT *pKF, FrameT &F, std::vector<MapPointT *> &vpMapPointMatches)
{
    namespace D = orbgpu_detail;
    const std::vector<MapPointT *> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPointT *>((size_t)F.N, static_cast<MapPointT *>(nullptr));
    if (!handle_) return 0;
    std::vector<int32_t> qIndex, qSeg, segStart, segIndex;
    D::intersect_feature_vectors(pKF->mFeatVec, F.mFeatVec,
                                 [&](size_t i) { MapPointT *p = vpMapPointsKF[i]; return p && !p->isBad(); },  // :191-197
                                 qIndex, qSeg, segStart, segIndex);
    if (qIndex.empty() || F.N == 0) return 0;
    std::vector<orbgpu_wquery> q(qIndex.size());
    std::vector<uint8_t> qd(qIndex.size() * 32);
    for (size_t k = 0; k < qIndex.size(); k++) {
        q[k] = D::window(0, 0, 0, qSeg[k], 0, pKF->mvKeysUn[qIndex[k]].angle, 0, 1);
        std::memcpy(&qd[k * 32], pKF->mDescriptors.ptr(qIndex[k]), 32);
    }
    std::vector<uint8_t> tmp;
    std::vector<int32_t> assign((size_t)F.N, -1), qmatch(q.size(), -1);
    int32_t nmatches = 0;
    // the rotation histogram compares with F.mvKeys (:238); keypoint angles do not change with undistortion
    const orbgpu_frame *df = D::device_frame(F);
    if (check(df ? orbgpu_search_by_bow_frame(handle_, q.data(), qd.data(), (int32_t)q.size(), df, segStart.data(), segIndex.data(),
                                              (int32_t)segStart.size() - 1, nullptr, TH_LOW, mfNNratio, mbCheckOrientation ? 1 : 0,
                                              assign.data(), qmatch.data(), &nmatches)
                 : orbgpu_search_by_bow(handle_, q.data(), qd.data(), (int32_t)q.size(), D::as_keypoints(F.mvKeys),
                                        D::descriptor_block(F.mDescriptors, F.N, tmp), F.N, segStart.data(), segIndex.data(),
                                        (int32_t)segStart.size() - 1, nullptr, TH_LOW, mfNNratio, mbCheckOrientation ? 1 : 0, assign.data(),
                                        qmatch.data(), &nmatches)))
        return 0;
    for (size_t k = 0; k < q.size(); k++)
        if (qmatch[k] >= 0) vpMapPointMatches[qmatch[k]] = vpMapPointsKF[qIndex[k]];  // :232
    return nmatches;
}

// ---- SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12)   :573-706 ------------------------------------
template <class KeyFrameT, class MapPointT>
int ORBmatcher::SearchByBoW(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12)
{
    namespace D = orbgpu_detail;
    const std::vector<MapPointT *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPointT *>(vpMapPoints1.size(), static_cast<MapPointT *>(nullptr));
    if (!handle_) return 0;
    const int N2 = (int)vpMapPoints2.size();
    std::vector<int32_t> qIndex, qSeg, segStart, segIndex;
    D::intersect_feature_vectors(pKF1->mFeatVec, pKF2->mFeatVec,
                                 [&](size_t i) { MapPointT *p = vpMapPoints1[i]; return p && !p->isBad(); },  // :609-613
                                 qIndex, qSeg, segStart, segIndex);
    if (qIndex.empty() || N2 == 0) return 0;
    std::vector<unsigned char> excluded((size_t)N2);
    for (int t = 0; t < N2; t++) excluded[t] = !vpMapPoints2[t] || vpMapPoints2[t]->isBad();  // :625-631
    std::vector<orbgpu_wquery> q(qIndex.size());
    std::vector<uint8_t> qd(qIndex.size() * 32);
    for (size_t k = 0; k < qIndex.size(); k++) {
        q[k] = D::window(0, 0, 0, qSeg[k], 0, pKF1->mvKeysUn[qIndex[k]].angle, 0, 1);
        std::memcpy(&qd[k * 32], pKF1->mDescriptors.ptr(qIndex[k]), 32);
    }
    std::vector<uint8_t> tmp;
    std::vector<int32_t> assign((size_t)N2, -1), qmatch(q.size(), -1);
    int32_t nmatches = 0;
    const orbgpu_frame *df = D::device_frame(*pKF2);
    if (check(df ? orbgpu_search_by_bow_frame(handle_, q.data(), qd.data(), (int32_t)q.size(), df, segStart.data(), segIndex.data(),
                                              (int32_t)segStart.size() - 1, excluded.data(), TH_LOW - 1, mfNNratio,
                                              mbCheckOrientation ? 1 : 0, assign.data(), qmatch.data(), &nmatches)
                 : orbgpu_search_by_bow(handle_, q.data(), qd.data(), (int32_t)q.size(), D::as_keypoints(pKF2->mvKeysUn),
                                        D::descriptor_block(pKF2->mDescriptors, N2, tmp), N2, segStart.data(), segIndex.data(),
                                        (int32_t)segStart.size() - 1, excluded.data(), TH_LOW - 1 /* bestDist1 < TH_LOW, :649 */,
                                        mfNNratio, mbCheckOrientation ? 1 : 0, assign.data(), qmatch.data(), &nmatches)))
        return 0;
    for (size_t k = 0; k < q.size(); k++)
        if (qmatch[k] >= 0) vpMatches12[qIndex[k]] = vpMapPoints2[qmatch[k]];  // :653
    return nmatches;
}

// ---- SearchForInitialization(Frame &F1, Frame &F2, vbPrevMatched, vnMatches12, windowSize)   :422-571 ---------
template <class FrameT, class PointT>
int ORBmatcher::SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<PointT> &vbPrevMatched, std::vector<int> &vnMatches12,
                                        int windowSize)
{
    namespace D = orbgpu_detail;
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>((size_t)n1, -1);
    if (!handle_ || n1 == 0) return 0;
    std::vector<float> prev((size_t)n1 * 2);
    for (int i = 0; i < n1; i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
    const float b[4] = {(float)F2.mnMinX, (float)F2.mnMinY, (float)F2.mnMaxX, (float)F2.mnMaxY};
    std::vector<uint8_t> tmp1, tmp2;
    int32_t nmatches = 0;
    const orbgpu_frame *df1 = D::device_frame(F1), *df2 = D::device_frame(F2);
    if (check(df1 && df2
                  ? orbgpu_search_for_initialization_frames(handle_, df1, df2, prev.data(), vnMatches12.data(), windowSize, mfNNratio,
                                                            mbCheckOrientation ? 1 : 0, &nmatches)
                  : orbgpu_search_for_initialization(handle_, D::as_keypoints(F1.mvKeysUn), D::descriptor_block(F1.mDescriptors, n1, tmp1),
                                                     n1, D::as_keypoints(F2.mvKeysUn), D::descriptor_block(F2.mDescriptors, n2, tmp2), n2,
                                                     b, prev.data(), vnMatches12.data(), windowSize, mfNNratio,
                                                     mbCheckOrientation ? 1 : 0, &nmatches)))
        return 0;
    for (int i = 0; i < n1; i++)
        if (vnMatches12[i] >= 0) vbPrevMatched[i] = F2.mvKeysUn[vnMatches12[i]].pt;  // :566-568
    return nmatches;
}

// ---- SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, F12, vMatchedPairs, bOnlyStereo)   :708-874 -------
template <class KeyFrameT, class MatT>
int ORBmatcher::SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, MatT F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                                       const bool bOnlyStereo)
{
    namespace D = orbgpu_detail;
    vMatchedPairs.clear();
    if (!handle_) return 0;
    // epipole of camera 1 in image 2 (:714-721)
    const D::SE3 T2w = D::se3_from_Rt(pKF2->GetRotation(), pKF2->GetTranslation());
    const D::V3 C2 = D::apply(T2w, D::vec3(pKF1->GetCameraCenter()));
    const float invz = 1.0f / C2.z;
    const float ex = pKF2->fx * C2.x * invz + pKF2->cx;
    const float ey = pKF2->fy * C2.y * invz + pKF2->cy;
    const int N1 = pKF1->N, N2 = pKF2->N;
    std::vector<int32_t> qIndex, qSeg, segStart, segIndex;
    D::intersect_feature_vectors(pKF1->mFeatVec, pKF2->mFeatVec,
                                 [&](size_t i) {
                                     if (pKF1->GetMapPoint(i)) return false;                 // :750-754
                                     return !bOnlyStereo || pKF1->mvuRight[i] >= 0;          // :756-760
                                 },
                                 qIndex, qSeg, segStart, segIndex);
    if (qIndex.empty() || N2 == 0) return 0;
    std::vector<unsigned char> hasMapPoint2((size_t)N2);
    for (int t = 0; t < N2; t++) hasMapPoint2[t] = pKF2->GetMapPoint(t) != nullptr;  // :773-777
    std::vector<orbgpu_wquery> q(qIndex.size());
    std::vector<uint8_t> qd(qIndex.size() * 32);
    for (size_t k = 0; k < qIndex.size(); k++) {
        const int i = qIndex[k];
        q[k] = D::window(pKF1->mvKeysUn[i].pt.x, pKF1->mvKeysUn[i].pt.y, 0, qSeg[k], 0, pKF1->mvKeysUn[i].angle, 0,
                         pKF1->mvuRight[i] >= 0 ? 5 : 1);
        std::memcpy(&qd[k * 32], pKF1->mDescriptors.ptr(i), 32);
    }
    float F[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F[3 * r + c] = D::el(F12, r, c);
    std::vector<uint8_t> tmp;
    std::vector<int32_t> qmatch(q.size(), -1);
    int32_t nmatches = 0;
    const orbgpu_frame *df = D::device_frame(*pKF2);
    if (check(df ? orbgpu_search_for_triangulation_frame(handle_, q.data(), qd.data(), (int32_t)q.size(), df, 1, segStart.data(),
                                                         segIndex.data(), (int32_t)segStart.size() - 1, hasMapPoint2.data(), F, ex, ey,
                                                         pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(),
                                                         (int32_t)pKF2->mvScaleFactors.size(), bOnlyStereo ? 1 : 0,
                                                         mbCheckOrientation ? 1 : 0, qmatch.data(), &nmatches)
                 : orbgpu_search_for_triangulation(handle_, q.data(), qd.data(), (int32_t)q.size(), D::as_keypoints(pKF2->mvKeysUn),
                                                   D::descriptor_block(pKF2->mDescriptors, N2, tmp), pKF2->mvuRight.data(), N2,
                                                   segStart.data(), segIndex.data(), (int32_t)segStart.size() - 1, hasMapPoint2.data(), F,
                                                   ex, ey, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(),
                                                   (int32_t)pKF2->mvScaleFactors.size(), bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                                   qmatch.data(), &nmatches)))
        return 0;
    std::vector<int> vMatches12((size_t)N1, -1);
    for (size_t k = 0; k < q.size(); k++)
        if (qmatch[k] >= 0) vMatches12[qIndex[k]] = qmatch[k];
    vMatchedPairs.reserve(nmatches > 0 ? nmatches : 0);
    for (int i = 0; i < N1; i++)  // :866-871
        if (vMatches12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)vMatches12[i]));
    return nmatches;
}

// ---- Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, th)   :876-1026 ------------------------------
template <class KeyFrameT, class MapPointT>
int ORBmatcher::Fuse(KeyFrameT *pKF, const std::vector<MapPointT *> &vpMapPoints, const float th)
{
    namespace D = orbgpu_detail;
    if (!handle_) return 0;
    const D::SE3 Tcw = D::se3_from_Rt(pKF->GetRotation(), pKF->GetTranslation());
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy, bf = pKF->mbf;
    const D::V3 Ow = D::vec3(pKF->GetCameraCenter());
    const int nMPs = (int)vpMapPoints.size();
    // The searches are independent of each other and of the map surgery below (world position, normal, distance
    // range and descriptor of a point that is still to be processed do not change while others are fused), so
    // they run up front for every point that is admissible now; whether a point is still admissible when its turn
    // comes is decided in the second loop, in the reference's order.
    std::vector<orbgpu_wquery> q;
    std::vector<uint8_t> qd;
    std::vector<int32_t> qOf((size_t)nMPs, -1);
    for (int i = 0; i < nMPs; i++) {
        MapPointT *pMP = vpMapPoints[i];
        if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        const D::V3 p3Dw = D::vec3(pMP->GetWorldPos());
        const D::V3 p3Dc = D::apply(Tcw, p3Dw);
        if (p3Dc.z < 0.0f) continue;
        const float invz = 1 / p3Dc.z;
        const float x = p3Dc.x * invz, y = p3Dc.y * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float ur = u - bf * invz;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const D::V3 PO = D::sub(p3Dw, Ow);
        const float dist3D = D::norm3(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        if (D::dot3(PO, D::vec3(pMP->GetNormal())) < 0.5 * dist3D) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        qOf[i] = (int32_t)q.size();
        q.push_back(D::window(u, v, radius, nPredictedLevel - 1, nPredictedLevel, 0.0f, ur, 1));
        D::push_descriptor(pMP, qd);
    }
    if (q.empty() || pKF->N == 0) return 0;
    float g[6], b[4];
    D::keyframe_geometry(*pKF, g, b);
    std::vector<uint8_t> tmp;
    std::vector<int32_t> qmatch(q.size(), -1);
    int32_t found = 0;
    if (check(orbgpu_matcher_set_grid_geometry(handle_, g))) return 0;
    const orbgpu_frame *df = D::device_frame(*pKF);  // its grid was built with the Frame's own bounds
    if (check(df ? orbgpu_search_in_area_frame(handle_, q.data(), qd.data(), (int32_t)q.size(), df, pKF->mvInvLevelSigma2.data(),
                                               (int32_t)pKF->mvInvLevelSigma2.size(), 1, TH_LOW, qmatch.data(), &found)
                 : orbgpu_search_in_area(handle_, q.data(), qd.data(), (int32_t)q.size(), D::as_keypoints(pKF->mvKeysUn),
                                         D::descriptor_block(pKF->mDescriptors, pKF->N, tmp), pKF->mvuRight.data(), pKF->N, b,
                                         pKF->mvInvLevelSigma2.data(), (int32_t)pKF->mvInvLevelSigma2.size(), 1, TH_LOW, qmatch.data(),
                                         &found)))
        return 0;
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {  // :1003-1022
        if (qOf[i] < 0 || qmatch[qOf[i]] < 0) continue;
        MapPointT *pMP = vpMapPoints[i];
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;  // replaced or added to pKF by an earlier fusion (:900)
        const int bestIdx = qmatch[qOf[i]];
        MapPointT *pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// ---- Fuse(KeyFrame *pKF, cv::Mat Scw, vpPoints, th, vpReplacePoint)   :1028-1151 ----------------------------
template <class KeyFrameT, class MatT, class MapPointT>
int ORBmatcher::Fuse(KeyFrameT *pKF, MatT Scw, const std::vector<MapPointT *> &vpPoints, float th, std::vector<MapPointT *> &vpReplacePoint)
{
    namespace D = orbgpu_detail;
    if (!handle_) return 0;
    const float fx = pKF->fx, fy = pKF->fy, cx = pKF->cx, cy = pKF->cy;
    const D::SE3 Tcw = D::se3_from_sim3(Scw);
    const D::V3 Ow = D::centre(Tcw);
    const std::set<MapPointT *> spAlreadyFound = pKF->GetMapPoints();
    const int nPoints = (int)vpPoints.size();
    std::vector<orbgpu_wquery> q;
    std::vector<uint8_t> qd;
    std::vector<int32_t> qIndex;
    for (int iMP = 0; iMP < nPoints; iMP++) {
        MapPointT *pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        const D::V3 p3Dw = D::vec3(pMP->GetWorldPos());
        const D::V3 p3Dc = D::apply(Tcw, p3Dw);
        if (p3Dc.z < 0.0f) continue;
        const float invz = (float)(1.0 / p3Dc.z);
        const float x = p3Dc.x * invz, y = p3Dc.y * invz;
        const float u = fx * x + cx, v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
        const D::V3 PO = D::sub(p3Dw, Ow);
        const float dist3D = D::norm3(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        if (D::dot3(PO, D::vec3(pMP->GetNormal())) < 0.5 * dist3D) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        const float radius = th * pKF->mvScaleFactors[nPredictedLevel];
        q.push_back(D::window(u, v, radius, nPredictedLevel - 1, nPredictedLevel, 0.0f, -1.0f, 1));
        D::push_descriptor(pMP, qd);
        qIndex.push_back(iMP);
    }
    if (q.empty() || pKF->N == 0) return 0;
    float g[6], b[4];
    D::keyframe_geometry(*pKF, g, b);
    std::vector<uint8_t> tmp;
    std::vector<int32_t> qmatch(q.size(), -1);
    int32_t found = 0;
    if (check(orbgpu_matcher_set_grid_geometry(handle_, g))) return 0;
    const orbgpu_frame *df = D::device_frame(*pKF);
    if (check(df ? orbgpu_search_in_area_frame(handle_, q.data(), qd.data(), (int32_t)q.size(), df, nullptr, 0, 0, TH_LOW, qmatch.data(), &found)
                 : orbgpu_search_in_area(handle_, q.data(), qd.data(), (int32_t)q.size(), D::as_keypoints(pKF->mvKeysUn),
                                         D::descriptor_block(pKF->mDescriptors, pKF->N, tmp), nullptr, pKF->N, b, nullptr, 0, 0, TH_LOW,
                                         qmatch.data(), &found)))
        return 0;
    int nFused = 0;
    for (size_t k = 0; k < q.size(); k++) {  // :1133-1147, in the order of vpPoints
        if (qmatch[k] < 0) continue;
        const int iMP = qIndex[k], bestIdx = qmatch[k];
        MapPointT *pMP = vpPoints[iMP];
        MapPointT *pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}

// ---- SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)   :1153-1377 ----------------------------------
template <class KeyFrameT, class MapPointT, class MatT>
int ORBmatcher::SearchBySim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12, const float &s12, const MatT &R12,
                             const MatT &t12, const float th)
{
    namespace D = orbgpu_detail;
    if (!handle_) return 0;
    const float fx = pKF1->fx, fy = pKF1->fy, cx = pKF1->cx, cy = pKF1->cy;  // pKF1's intrinsics for both directions (:1156-1159)
    const D::SE3 T1w = D::se3_from_Rt(pKF1->GetRotation(), pKF1->GetTranslation());
    const D::SE3 T2w = D::se3_from_Rt(pKF2->GetRotation(), pKF2->GetTranslation());
    // sR12 = s12 * R12, sR21 = (1 / s12) * R12^T, t21 = -sR21 * t12 (:1170-1172)
    D::SE3 S12, S21;
    const float inv_s = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            S12.R[3 * r + c] = D::el(R12, r, c) * s12;
            S21.R[3 * r + c] = D::el(R12, c, r) * inv_s;
        }
    for (int r = 0; r < 3; r++) S12.t[r] = D::el(t12, r, 0);
    for (int r = 0; r < 3; r++)
        S21.t[r] = -((S21.R[3 * r] * S12.t[0] + S21.R[3 * r + 1] * S12.t[1]) + S21.R[3 * r + 2] * S12.t[2]);

    const std::vector<MapPointT *> vpMapPoints1 = pKF1->GetMapPointMatches(), vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size(), N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1((size_t)N1, false), vbAlreadyMatched2((size_t)N2, false);
    for (int i = 0; i < N1; i++) {  // :1183-1193
        MapPointT *pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            const int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    // one direction: the map points of `src` (camera Tsw) moved by S into the other camera and searched in `dst`
    auto direction = [&](const std::vector<MapPointT *> &points, const std::vector<bool> &already, const D::SE3 &Tsw, const D::SE3 &S,
                         KeyFrameT *dst, std::vector<int> &vnMatch) -> int {
        std::vector<orbgpu_wquery> q;
        std::vector<uint8_t> qd;
        std::vector<int32_t> qIndex;
        for (int i = 0; i < (int)points.size(); i++) {
            MapPointT *pMP = points[i];
            if (!pMP || already[i] || pMP->isBad()) continue;
            const D::V3 p = D::apply(S, D::apply(Tsw, D::vec3(pMP->GetWorldPos())));
            if (p.z < 0.0) continue;
            const float invz = (float)(1.0 / p.z);
            const float x = p.x * invz, y = p.y * invz;
            const float u = fx * x + cx, v = fy * y + cy;
            if (!dst->IsInImage(u, v)) continue;
            const float maxDistance = pMP->GetMaxDistanceInvariance(), minDistance = pMP->GetMinDistanceInvariance();
            const float dist3D = D::norm3(p);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const int nPredictedLevel = pMP->PredictScale(dist3D, dst);
            const float radius = th * dst->mvScaleFactors[nPredictedLevel];
            q.push_back(D::window(u, v, radius, nPredictedLevel - 1, nPredictedLevel, 0.0f, -1.0f, 1));
            D::push_descriptor(pMP, qd);
            qIndex.push_back(i);
        }
        if (q.empty() || dst->N == 0) return ORBGPU_OK;
        float g[6], b[4];
        D::keyframe_geometry(*dst, g, b);
        std::vector<uint8_t> tmp;
        std::vector<int32_t> qmatch(q.size(), -1);
        int32_t found = 0;
        int rc = check(orbgpu_matcher_set_grid_geometry(handle_, g));
        if (rc) return rc;
        const orbgpu_frame *df = D::device_frame(*dst);
        rc = check(df ? orbgpu_search_in_area_frame(handle_, q.data(), qd.data(), (int32_t)q.size(), df, nullptr, 0, 0, TH_HIGH, qmatch.data(),
                                                    &found)
                      : orbgpu_search_in_area(handle_, q.data(), qd.data(), (int32_t)q.size(), D::as_keypoints(dst->mvKeysUn),
                                              D::descriptor_block(dst->mDescriptors, dst->N, tmp), nullptr, dst->N, b, nullptr, 0, 0,
                                              TH_HIGH, qmatch.data(), &found));
        if (rc) return rc;
        for (size_t k = 0; k < q.size(); k++) vnMatch[qIndex[k]] = qmatch[k];
        return ORBGPU_OK;
    };
    std::vector<int> vnMatch1((size_t)N1, -1), vnMatch2((size_t)N2, -1);
    if (direction(vpMapPoints1, vbAlreadyMatched1, T1w, S21, pKF2, vnMatch1)) return 0;  // :1199-1276
    if (direction(vpMapPoints2, vbAlreadyMatched2, T2w, S12, pKF1, vnMatch2)) return 0;  // :1279-1356
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++) {  // mutual agreement, :1361-1374
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) {
            vpMatches12[i1] = vpMapPoints2[idx2];
            nFound++;
        }
    }
    return nFound;
}

}  // namespace ORB_SLAM2

#endif
