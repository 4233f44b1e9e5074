// Test fixtures: the smallest Frame / KeyFrame / MapPoint types that carry the members ORB_SLAM2::ORBmatcher touches
// (names as in include/Frame.h, include/KeyFrame.h, include/MapPoint.h of the reference), plus float matrices with
// cv::Mat's `at<float>(row, col)`.  They hold data and bookkeeping only; nothing here computes a match.
#ifndef ORBGPU_TEST_SLAM_TYPES_H
#define ORBGPU_TEST_SLAM_TYPES_H

#include <cmath>
#include <map>
#include <memory>
#include <set>
#include <type_traits>
#include <vector>

#include "FrameDevice.h"   // ORB_SLAM2::DeviceFrame
#include "ORBextractor.h"  // orbcv::KeyPoint, orbcv::Mat (8-bit), orbcv::Point2f

namespace slamtest {

struct FMat {  // float matrix
    int rows = 0, cols = 0;
    std::vector<float> d;
    FMat() {}
    FMat(int r, int c) : rows(r), cols(c), d((size_t)r * c, 0.0f) {}
    template <class T> T &at(int r, int c) { static_assert(std::is_same<T, float>::value, "float only"); return d[(size_t)r * cols + c]; }
    template <class T> const T &at(int r, int c) const { static_assert(std::is_same<T, float>::value, "float only"); return d[(size_t)r * cols + c]; }
};

typedef std::map<unsigned, std::vector<unsigned> > FeatureVector;  // DBoW2::FeatureVector

struct MapPoint;

struct Frame {
    int N = 0;
    std::vector<orbcv::KeyPoint> mvKeys, mvKeysUn;
    orbcv::Mat mDescriptors;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<float> mvuRight;
    std::vector<float> mvScaleFactors;
    FeatureVector mFeatVec;
    FMat mTcw;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0;
    float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
    int mnScaleLevels = 0;
    float mfLogScaleFactor = 0;
    std::shared_ptr<ORB_SLAM2::DeviceFrame> mpDeviceFrame;  // null: the matchers are given the host arrays
};

struct KeyFrame {
    int N = 0;
    std::vector<orbcv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight;
    orbcv::Mat mDescriptors;
    FeatureVector mFeatVec;
    std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;  // int in the reference (include/KeyFrame.h:185)
    float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    float gridOriginX = 0, gridOriginY = 0;  // the float bounds of the Frame the grid was filled with
    int mnScaleLevels = 0;
    float mfLogScaleFactor = 0;
    FMat Tcw;  // 4x4
    std::vector<MapPoint *> mvpMapPoints;
    std::shared_ptr<ORB_SLAM2::DeviceFrame> mpDeviceFrame;  // the handle of the Frame the KeyFrame was made from

    std::vector<MapPoint *> GetMapPointMatches() { return mvpMapPoints; }
    std::set<MapPoint *> GetMapPoints()
    {
        std::set<MapPoint *> s;
        for (MapPoint *p : mvpMapPoints)
            if (p) s.insert(p);
        return s;
    }
    MapPoint *GetMapPoint(const size_t &idx) { return mvpMapPoints[idx]; }
    void AddMapPoint(MapPoint *pMP, const size_t &idx) { mvpMapPoints[idx] = pMP; }
    void ReplaceMapPointMatch(const size_t &idx, MapPoint *pMP) { mvpMapPoints[idx] = pMP; }
    void EraseMapPointMatch(const size_t &idx) { mvpMapPoints[idx] = nullptr; }
    FMat GetRotation() const
    {
        FMat R(3, 3);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) R.at<float>(r, c) = Tcw.at<float>(r, c);
        return R;
    }
    FMat GetTranslation() const
    {
        FMat t(3, 1);
        for (int r = 0; r < 3; r++) t.at<float>(r, 0) = Tcw.at<float>(r, 3);
        return t;
    }
    FMat GetCameraCenter() const { return Ow; }
    FMat Ow;  // 3x1, set with the pose
    bool IsInImage(const float &x, const float &y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }
};

struct MapPoint {
    int id = -1;
    FMat pos, normal;  // 3x1
    orbcv::Mat descriptor;  // 1x32
    float mfMinDistance = 0, mfMaxDistance = 0;
    bool bad = false;
    int nObs = 0;
    std::map<KeyFrame *, size_t> observations;
    // tracking fields (include/MapPoint.h:92-97)
    float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 0;

    FMat GetWorldPos() const { return pos; }
    FMat GetNormal() const { return normal; }
    orbcv::Mat GetDescriptor() const { return descriptor; }
    bool isBad() const { return bad; }
    int Observations() const { return nObs; }
    float GetMinDistanceInvariance() const { return 0.8f * mfMinDistance; }
    float GetMaxDistanceInvariance() const { return 1.2f * mfMaxDistance; }
    template <class T> int PredictScale(const float &currentDist, T *p) const  // KeyFrame* or Frame*
    {
        const float ratio = mfMaxDistance / currentDist;
        int nScale = (int)std::ceil(std::log(ratio) / p->mfLogScaleFactor);
        if (nScale < 0) nScale = 0;
        else if (nScale >= p->mnScaleLevels) nScale = p->mnScaleLevels - 1;
        return nScale;
    }
    bool IsInKeyFrame(KeyFrame *pKF) const { return observations.count(pKF) != 0; }
    int GetIndexInKeyFrame(KeyFrame *pKF) const
    {
        std::map<KeyFrame *, size_t>::const_iterator it = observations.find(pKF);
        return it == observations.end() ? -1 : (int)it->second;
    }
    void AddObservation(KeyFrame *pKF, size_t idx)
    {
        if (observations.count(pKF)) return;
        observations[pKF] = idx;
        nObs += pKF->mvuRight[idx] >= 0 ? 2 : 1;
    }
    void Replace(MapPoint *pMP)  // this point hands its observations to pMP and turns bad
    {
        if (pMP == this) return;
        std::map<KeyFrame *, size_t> obs = observations;
        observations.clear();
        bad = true;
        for (std::map<KeyFrame *, size_t>::iterator it = obs.begin(); it != obs.end(); ++it) {
            KeyFrame *pKF = it->first;
            if (!pMP->IsInKeyFrame(pKF)) {
                pKF->ReplaceMapPointMatch(it->second, pMP);
                pMP->AddObservation(pKF, it->second);
            } else {
                pKF->EraseMapPointMatch(it->second);
            }
        }
    }
};

}  // namespace slamtest

#endif
