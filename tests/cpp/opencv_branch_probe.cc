// Type-checks the ORBGPU_WITH_OPENCV branch of the drop-in headers against declarations with OpenCV's signatures
// (tests/cpp/fake_opencv): compiled with -fsyntax-only, never linked or run.
#define ORBGPU_WITH_OPENCV
#include "FrameDevice.h"
#include "ORBVocabulary.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"

int probe(cv::Mat &im, cv::Mat &depth)
{
    ORB_SLAM2::ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> keys, keysUn;
    cv::Mat desc;
    ex(im, cv::Mat(), keys, desc);  // the call of Frame::ExtractORB (src/Frame.cc:306-312)
    const cv::Mat &lvl = ex.mvImagePyramid[1];
    ORB_SLAM2::ORBmatcher m(0.9f, true);
    std::vector<cv::Point2f> prev(keys.size());
    std::vector<int> m12;
    const float b[4] = {0, 0, 640, 480}, K4[4] = {500, 500, 320, 240}, D5[5] = {0, 0, 0, 0, 0};
    int n = m.SearchForInitialization(keys, desc, keys, desc, b, prev, m12, 100);
    n += ORB_SLAM2::ORBmatcher::DescriptorDistance(desc.row(0), desc.row(1));
    m.UndistortKeyPoints(keys, K4, D5, keysUn);
    std::vector<int> bi, bd, bd2;
    n += m.BruteForce(desc, desc, bi, bd, bd2);
    ORB_SLAM2::DeviceFrame f(ex.MaxKeypoints());
    n += f.BuildMono(ex, im.data, im.cols, im.rows, (int)im.step, K4, D5, b);
    n += f.BuildRGBD(ex, im.data, im.cols, im.rows, (int)im.step, (const float *)depth.data, (int)((size_t)depth.step / 4), 40.f, K4, D5, b);
    f.Keys(keys);
    f.KeysUn(keysUn);
    ORB_SLAM2::ORBVocabulary voc;
    DBoW2::BowVector bow;
    DBoW2::FeatureVector fv;
    voc.transform(desc, bow, fv, 4);
    return n + lvl.rows;
}

// The reference-signature templates on types that hold cv::Mat where the reference's do (include/Frame.h,
// include/KeyFrame.h, include/MapPoint.h): poses and points as CV_32F matrices read with at<float>, descriptors as
// CV_8U rows read with ptr().
#include <map>
#include <memory>
#include <set>
namespace probe_types {
struct KeyFrame;
struct MapPoint {
    cv::Mat GetWorldPos();
    cv::Mat GetNormal();
    cv::Mat GetDescriptor();
    bool isBad();
    int Observations();
    float GetMinDistanceInvariance();
    float GetMaxDistanceInvariance();
    template <class T> int PredictScale(const float &d, T *p);
    bool IsInKeyFrame(KeyFrame *pKF);
    int GetIndexInKeyFrame(KeyFrame *pKF);
    void AddObservation(KeyFrame *pKF, size_t idx);
    void Replace(MapPoint *pMP);
    float mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos;
    bool mbTrackInView;
    int mnTrackScaleLevel;
};
struct Frame {
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    cv::Mat mDescriptors, mTcw;
    std::vector<MapPoint *> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    std::vector<float> mvuRight, mvScaleFactors;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;
    float fx, fy, cx, cy, mbf, mb;
    static float mnMinX, mnMinY, mnMaxX, mnMaxY;
    int mnScaleLevels;
    float mfLogScaleFactor;
    std::shared_ptr<ORB_SLAM2::DeviceFrame> mpDeviceFrame;
};
struct KeyFrame {
    int N;
    std::vector<cv::KeyPoint> mvKeys, mvKeysUn;
    std::vector<float> mvuRight, mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    cv::Mat mDescriptors;
    std::map<unsigned, std::vector<unsigned> > mFeatVec;
    float fx, fy, cx, cy, mbf;
    int mnMinX, mnMinY, mnMaxX, mnMaxY;
    float mfGridElementWidthInv, mfGridElementHeightInv;
    int mnScaleLevels;
    float mfLogScaleFactor;
    std::vector<MapPoint *> GetMapPointMatches();
    std::set<MapPoint *> GetMapPoints();
    MapPoint *GetMapPoint(const size_t &idx);
    void AddMapPoint(MapPoint *pMP, const size_t &idx);
    void ReplaceMapPointMatch(const size_t &idx, MapPoint *pMP);
    cv::Mat GetRotation();
    cv::Mat GetTranslation();
    cv::Mat GetCameraCenter();
    bool IsInImage(const float &x, const float &y) const;
};
}  // namespace probe_types

int probe_templates(probe_types::Frame &cur, probe_types::Frame &last, probe_types::KeyFrame *kf1, probe_types::KeyFrame *kf2,
                    std::vector<probe_types::MapPoint *> &pts, cv::Mat Scw, cv::Mat R12, cv::Mat t12, cv::Mat F12)
{
    using probe_types::MapPoint;
    ORB_SLAM2::ORBmatcher m(0.9f, true);
    int n = m.SearchByProjection(cur, pts, 3.0f);
    n += m.SearchByProjection(cur, last, 15.0f, true);
    std::set<MapPoint *> found;
    n += m.SearchByProjection(cur, kf1, found, 10.0f, 100);
    std::vector<MapPoint *> matched(kf1->N, nullptr), m12, repl(pts.size(), nullptr);
    n += m.SearchByProjection(kf1, Scw, pts, matched, 10);
    n += m.SearchByBoW(kf1, cur, m12);
    n += m.SearchByBoW(kf1, kf2, m12);
    std::vector<cv::Point2f> prev(cur.mvKeysUn.size());
    std::vector<int> im12;
    n += m.SearchForInitialization(cur, last, prev, im12, 100);
    std::vector<std::pair<size_t, size_t> > pairs;
    n += m.SearchForTriangulation(kf1, kf2, F12, pairs, false);
    n += m.Fuse(kf1, pts, 3.0f);
    n += m.Fuse(kf1, Scw, pts, 4.0f, repl);
    const float s12 = 1.0f;
    n += m.SearchBySim3(kf1, kf2, m12, s12, R12, t12, 7.5f);
    return n;
}
