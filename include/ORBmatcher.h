/*
 * ORBmatcher.h -- host-side C++ mirror of the descriptor hot path of
 * ORB_SLAM2::ORBmatcher (reference:
 * /root/reference/src/ORB_SLAM2/include/ORBmatcher.h:41-103) on top of the C
 * ABI in orbgpu.h.
 *
 * Two layers.  (1) The eleven search methods with the reference's own
 * signatures (include/ORBmatcher.h:48-83 of the reference), as member templates
 * over the caller's Frame / KeyFrame / MapPoint types: declared here, defined in
 * ORBmatcherFrames.h, which this header includes.  (2) Array forms of the same
 * searches (keypoints, descriptors, mvuRight, "has a map point" flags plus one
 * window query per map point) for callers that hold arrays; layer (1) is built
 * on them.
 *
 * An ORBmatcher is a value: the reference creates one on the stack at every call
 * site (src/Tracking.cc:738,951,1060,1381,1568; src/LocalMapping.cc:215,483;
 * src/LoopClosing.cc:241,325,377,591).  Construction takes a recycled GPU context
 * from the library (no stream creation, no allocation after the first few), and
 * copies get a context of their own.
 */
#ifndef ORBGPU_ORBMATCHER_H
#define ORBGPU_ORBMATCHER_H

#include <map>
#include <set>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "ORBextractor.h"
#include "orbgpu.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    // reference: include/ORBmatcher.h:41
    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), device_(device)
    {
        if (orbgpu_matcher_create(device, &handle_) != ORBGPU_OK) { handle_ = nullptr; err_ = orbgpu_last_error(); }
    }
    ~ORBmatcher() { orbgpu_matcher_destroy(handle_); }
    ORBmatcher(const ORBmatcher &o) : mfNNratio(o.mfNNratio), mbCheckOrientation(o.mbCheckOrientation), device_(o.device_)
    {
        if (orbgpu_matcher_create(device_, &handle_) != ORBGPU_OK) { handle_ = nullptr; err_ = orbgpu_last_error(); }
    }
    ORBmatcher &operator=(const ORBmatcher &o)  // the context stays, the settings are copied
    {
        mfNNratio = o.mfNNratio;
        mbCheckOrientation = o.mbCheckOrientation;
        return *this;
    }

    // ---- the reference's search methods (include/ORBmatcher.h:48-83), defined in ORBmatcherFrames.h ----------
    template <class FrameT, class MapPointT>
    int SearchByProjection(FrameT &F, const std::vector<MapPointT *> &vpMapPoints, const float th = 3);
    template <class FrameT>
    int SearchByProjection(FrameT &CurrentFrame, const FrameT &LastFrame, const float th, const bool bMono);
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchByProjection(FrameT &CurrentFrame, KeyFrameT *pKF, const std::set<MapPointT *> &sAlreadyFound, const float th,
                           const int ORBdist);
    template <class KeyFrameT, class MatT, class MapPointT>
    int SearchByProjection(KeyFrameT *pKF, MatT Scw, const std::vector<MapPointT *> &vpPoints, std::vector<MapPointT *> &vpMatched, int th);
    template <class KeyFrameT, class FrameT, class MapPointT>
    typename std::enable_if<!std::is_pointer<FrameT>::value, int>::type
    SearchByBoW(KeyFrameT *pKF, FrameT &F, std::vector<MapPointT *> &vpMapPointMatches);
    template <class KeyFrameT, class MapPointT>
    int SearchByBoW(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12);
    template <class FrameT, class PointT>
    int SearchForInitialization(FrameT &F1, FrameT &F2, std::vector<PointT> &vbPrevMatched, std::vector<int> &vnMatches12,
                                int windowSize = 10);
    template <class KeyFrameT, class MatT>
    int SearchForTriangulation(KeyFrameT *pKF1, KeyFrameT *pKF2, MatT F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                               const bool bOnlyStereo);
    template <class KeyFrameT, class MapPointT, class MatT>
    int SearchBySim3(KeyFrameT *pKF1, KeyFrameT *pKF2, std::vector<MapPointT *> &vpMatches12, const float &s12, const MatT &R12,
                     const MatT &t12, const float th);
    template <class KeyFrameT, class MapPointT>
    int Fuse(KeyFrameT *pKF, const std::vector<MapPointT *> &vpMapPoints, const float th = 3.0);
    template <class KeyFrameT, class MatT, class MapPointT>
    int Fuse(KeyFrameT *pKF, MatT Scw, const std::vector<MapPointT *> &vpPoints, float th, std::vector<MapPointT *> &vpReplacePoint);

    // reference: include/ORBmatcher.h:44, src/ORBmatcher.cc:1710-1727 (rows of 32 bytes)
    static int DescriptorDistance(const orbcv::Mat &a, const orbcv::Mat &b) { return orbgpu_descriptor_distance(a.data, b.data); }

    // Brute-force scan used inside SearchByBoW / ComputeDistinctiveDescriptors
    // (src/ORBmatcher.cc:205-226, src/MapPoint.cc:302-313): best index, best and
    // second-best distance per query row.
    int BruteForce(const orbcv::Mat &queries, const orbcv::Mat &train, std::vector<int> &bestIdx, std::vector<int> &bestDist,
                   std::vector<int> &secondDist)
    {
        const int nq = queries.rows, nt = train.rows;
        bestIdx.assign(nq, -1); bestDist.assign(nq, 256); secondDist.assign(nq, 256);
        if (!handle_ || nq == 0) return 0;
        return check(orbgpu_hamming_bruteforce(handle_, queries.data, nq, train.data, nt, bestIdx.data(), bestDist.data(),
                                               secondDist.data()));
    }

    // SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th)      mode ORBGPU_MODE_LOCAL_MAP
    // SearchByProjection(Frame &Cur, const Frame &Last, th, bMono)                mode ORBGPU_MODE_LAST_FRAME
    // SearchByProjection(Frame &Cur, KeyFrame *pKF, sAlreadyFound, th, ORBdist)   mode ORBGPU_MODE_KEYFRAME
    // (include/ORBmatcher.h:48-57).  queries[i] is the projection window of map
    // point i, queryDesc row i its descriptor; the train side is the current
    // frame.  assign[t] = query whose map point lands on keypoint t, or -1.
    int SearchByProjection(int mode, const std::vector<orbgpu_wquery> &queries, const orbcv::Mat &queryDesc,
                           const std::vector<orbcv::KeyPoint> &keysUn, const orbcv::Mat &descriptors,
                           const std::vector<float> *uRight, const std::vector<unsigned char> *hasMapPoint,
                           const float bounds[4], std::vector<int> &assign, int thDist = TH_HIGH)
    {
        const int nq = (int)queries.size(), n2 = (int)keysUn.size();
        assign.assign(n2, -1);
        if (!handle_) return 0;
        int32_t nmatches = 0;
        if (check(orbgpu_search_by_projection(handle_, mode, queries.data(), queryDesc.data, nq,
                                              reinterpret_cast<const orbgpu_keypoint *>(keysUn.data()), descriptors.data,
                                              uRight ? uRight->data() : nullptr, n2, bounds,
                                              hasMapPoint ? hasMapPoint->data() : nullptr, assign.data(), thDist, mfNNratio,
                                              mbCheckOrientation ? 1 : 0, &nmatches)))
            return 0;
        return nmatches;
    }

    // SearchByBoW(KeyFrame *pKF, Frame &F, vpMapPointMatches)            strict = false  (src/ORBmatcher.cc:159-288)
    // SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12)           strict = true   (src/ORBmatcher.cc:573-706)
    // (include/ORBmatcher.h:62-63).  featVec1 / featVec2 are the DBoW2::FeatureVector of the two sides (node id ->
    // feature indices); valid1[i]: feature i of side 1 owns a good map point; excluded2 (optional): features of
    // side 2 that may not be matched (no good map point, :625-631).  matches12[i] = side-2 feature of side-1
    // feature i or -1; matches21 the inverse.
    int SearchByBoW(const std::map<unsigned, std::vector<unsigned> > &featVec1, const std::vector<orbcv::KeyPoint> &keys1,
                    const orbcv::Mat &desc1, const std::vector<unsigned char> &valid1,
                    const std::map<unsigned, std::vector<unsigned> > &featVec2, const std::vector<orbcv::KeyPoint> &keys2,
                    const orbcv::Mat &desc2, const std::vector<unsigned char> *excluded2, bool strict,
                    std::vector<int> &matches12, std::vector<int> &matches21)
    {
        const int n1 = (int)keys1.size(), n2 = (int)keys2.size();
        matches12.assign(n1, -1); matches21.assign(n2, -1);
        if (!handle_) return 0;
        // intersection of the two sorted node maps (:180-264)
        std::vector<orbgpu_wquery> q;
        std::vector<unsigned char> qd;
        std::vector<int32_t> qIndex, segStart(1, 0), segIndex;
        std::map<unsigned, std::vector<unsigned> >::const_iterator it1 = featVec1.begin(), it2 = featVec2.begin();
        while (it1 != featVec1.end() && it2 != featVec2.end()) {
            if (it1->first == it2->first) {
                const int32_t s = (int32_t)segStart.size() - 1;
                for (size_t k = 0; k < it2->second.size(); k++) segIndex.push_back((int32_t)it2->second[k]);
                segStart.push_back((int32_t)segIndex.size());
                for (size_t k = 0; k < it1->second.size(); k++) {
                    const unsigned i = it1->second[k];
                    if (i >= (unsigned)n1 || !valid1[i]) continue;
                    orbgpu_wquery w = orbgpu_wquery();
                    w.angle = keys1[i].angle; w.minLevel = s; w.flags = 1;
                    q.push_back(w);
                    qIndex.push_back((int32_t)i);
                    qd.insert(qd.end(), desc1.data + (size_t)i * 32, desc1.data + (size_t)i * 32 + 32);
                }
                ++it1; ++it2;
            } else if (it1->first < it2->first) {
                it1 = featVec1.lower_bound(it2->first);
            } else {
                it2 = featVec2.lower_bound(it1->first);
            }
        }
        if (segIndex.empty()) segIndex.push_back(0);
        std::vector<int32_t> assign(n2 > 0 ? n2 : 1, -1), qmatch(q.empty() ? 1 : q.size(), -1);
        int32_t nmatches = 0;
        if (check(orbgpu_search_by_bow(handle_, q.data(), qd.data(), (int32_t)q.size(),
                                       reinterpret_cast<const orbgpu_keypoint *>(keys2.data()), desc2.data, n2, segStart.data(),
                                       segIndex.data(), (int32_t)segStart.size() - 1, excluded2 ? excluded2->data() : nullptr,
                                       strict ? TH_LOW - 1 : TH_LOW, mfNNratio, mbCheckOrientation ? 1 : 0, assign.data(),
                                       qmatch.data(), &nmatches)))
            return 0;
        for (size_t k = 0; k < q.size(); k++)
            if (qmatch[k] >= 0) { matches12[qIndex[k]] = qmatch[k]; matches21[qmatch[k]] = qIndex[k]; }
        return nmatches;
    }

    // SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vMatchedPairs, bOnlyStereo)
    // (include/ORBmatcher.h:73-74, src/ORBmatcher.cc:708-874).  keys = mvKeysUn, uRight = mvuRight (empty: monocular),
    // hasMapPoint = GetMapPoint(i) != NULL, F12 row-major 3x3, (ex, ey) = epipole of camera 1 in image 2 (:714-721),
    // scale2 / sigma2 = mvScaleFactors / mvLevelSigma2 of keyframe 2.
    int SearchForTriangulation(const std::map<unsigned, std::vector<unsigned> > &featVec1, const std::vector<orbcv::KeyPoint> &keys1,
                               const orbcv::Mat &desc1, const std::vector<unsigned char> &hasMapPoint1,
                               const std::vector<float> &uRight1, const std::map<unsigned, std::vector<unsigned> > &featVec2,
                               const std::vector<orbcv::KeyPoint> &keys2, const orbcv::Mat &desc2,
                               const std::vector<unsigned char> &hasMapPoint2, const std::vector<float> &uRight2,
                               const float F12[9], float ex, float ey, const std::vector<float> &scale2,
                               const std::vector<float> &sigma2, std::vector<std::pair<size_t, size_t> > &vMatchedPairs,
                               bool bOnlyStereo)
    {
        const int n1 = (int)keys1.size(), n2 = (int)keys2.size();
        vMatchedPairs.clear();
        if (!handle_) return 0;
        std::vector<orbgpu_wquery> q;
        std::vector<unsigned char> qd;
        std::vector<int32_t> qIndex, segStart(1, 0), segIndex;
        std::map<unsigned, std::vector<unsigned> >::const_iterator it1 = featVec1.begin(), it2 = featVec2.begin();
        while (it1 != featVec1.end() && it2 != featVec2.end()) {
            if (it1->first == it2->first) {
                const int32_t s = (int32_t)segStart.size() - 1;
                for (size_t k = 0; k < it2->second.size(); k++) segIndex.push_back((int32_t)it2->second[k]);
                segStart.push_back((int32_t)segIndex.size());
                for (size_t k = 0; k < it1->second.size(); k++) {
                    const unsigned i = it1->second[k];
                    if (i >= (unsigned)n1 || hasMapPoint1[i]) continue;               // :750-754
                    const bool stereo1 = !uRight1.empty() && uRight1[i] >= 0;
                    if (bOnlyStereo && !stereo1) continue;                           // :758-760
                    orbgpu_wquery w = orbgpu_wquery();
                    w.u = keys1[i].pt.x; w.v = keys1[i].pt.y; w.angle = keys1[i].angle; w.minLevel = s; w.flags = stereo1 ? 5 : 1;
                    q.push_back(w);
                    qIndex.push_back((int32_t)i);
                    qd.insert(qd.end(), desc1.data + (size_t)i * 32, desc1.data + (size_t)i * 32 + 32);
                }
                ++it1; ++it2;
            } else if (it1->first < it2->first) {
                it1 = featVec1.lower_bound(it2->first);
            } else {
                it2 = featVec2.lower_bound(it1->first);
            }
        }
        if (segIndex.empty()) segIndex.push_back(0);
        std::vector<int32_t> qmatch(q.empty() ? 1 : q.size(), -1);
        int32_t nmatches = 0;
        if (check(orbgpu_search_for_triangulation(handle_, q.data(), qd.data(), (int32_t)q.size(),
                                                  reinterpret_cast<const orbgpu_keypoint *>(keys2.data()), desc2.data,
                                                  uRight2.empty() ? nullptr : uRight2.data(), n2, segStart.data(), segIndex.data(),
                                                  (int32_t)segStart.size() - 1, hasMapPoint2.data(), F12, ex, ey, scale2.data(),
                                                  sigma2.data(), (int32_t)scale2.size(), bOnlyStereo ? 1 : 0,
                                                  mbCheckOrientation ? 1 : 0, qmatch.data(), &nmatches)))
            return 0;
        std::vector<int> m12(n1, -1);
        for (size_t k = 0; k < q.size(); k++)
            if (qmatch[k] >= 0) m12[qIndex[k]] = qmatch[k];
        for (int i = 0; i < n1; i++)  // :866-871
            if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
        return nmatches;
    }

    // The keypoint search inside Fuse(KeyFrame*, vpMapPoints, th) (chi2Gate = true), Fuse(KeyFrame*, Scw, ...) and
    // both directions of SearchBySim3 (thDist = TH_HIGH) (include/ORBmatcher.h:77-89, src/ORBmatcher.cc:876-1377):
    // queries[i] is the projection window of map point i with minLevel = predicted-1, maxLevel = predicted; every
    // query independently takes the nearest keypoint.  bestIdx[i] = keypoint or -1.  The map surgery of Fuse and
    // the mutual-agreement step of SearchBySim3 (:1356-1374) stay with the caller.
    int SearchInArea(const std::vector<orbgpu_wquery> &queries, const orbcv::Mat &queryDesc,
                     const std::vector<orbcv::KeyPoint> &keysUn, const orbcv::Mat &descriptors, const std::vector<float> *uRight,
                     const float bounds[4], const std::vector<float> *invLevelSigma2, bool chi2Gate, int thDist,
                     std::vector<int> &bestIdx)
    {
        const int nq = (int)queries.size(), n2 = (int)keysUn.size();
        bestIdx.assign(nq, -1);
        if (!handle_ || nq == 0) return 0;
        int32_t nmatches = 0;
        if (check(orbgpu_search_in_area(handle_, queries.data(), queryDesc.data, nq,
                                        reinterpret_cast<const orbgpu_keypoint *>(keysUn.data()), descriptors.data,
                                        uRight ? uRight->data() : nullptr, n2, bounds,
                                        invLevelSigma2 ? invLevelSigma2->data() : nullptr,
                                        invLevelSigma2 ? (int32_t)invLevelSigma2->size() : 0, chi2Gate ? 1 : 0, thDist,
                                        bestIdx.data(), &nmatches)))
            return 0;
        return nmatches;
    }

    // MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:262-340) for many map points at once: the observations of
    // map point p are rows start[p] .. start[p+1]-1 of observed; bestIdx[p] = the representative's offset in that range.
    int ComputeDistinctiveDescriptors(const orbcv::Mat &observed, const std::vector<int> &start, std::vector<int> &bestIdx)
    {
        const int np = (int)start.size() - 1;
        bestIdx.assign(np > 0 ? np : 0, -1);
        if (!handle_ || np <= 0) return 0;
        return check(orbgpu_distinctive_descriptors(handle_, observed.data, start.data(), np, bestIdx.data(), nullptr));
    }

    // Frame::AssignFeaturesToGrid (src/Frame.cc:270-293): CSR over the 64x48 cells, cell = ix*48 + iy.
    int AssignFeaturesToGrid(const std::vector<orbcv::KeyPoint> &keysUn, const float bounds[4], std::vector<int> &cellStart,
                             std::vector<int> &cellIndex)
    {
        cellStart.assign(ORBGPU_GRID_COLS * ORBGPU_GRID_ROWS + 1, 0);
        cellIndex.assign(keysUn.empty() ? 1 : keysUn.size(), 0);
        if (!handle_) return 0;
        int32_t n = 0;
        const int rc = check(orbgpu_assign_features_to_grid(handle_, reinterpret_cast<const orbgpu_keypoint *>(keysUn.data()),
                                                            (int32_t)keysUn.size(), bounds, cellStart.data(), cellIndex.data(), &n));
        cellIndex.resize(rc ? 0 : n);
        return rc;
    }

    // Frame::UndistortKeyPoints (src/Frame.cc:517-558): K4 = {fx, fy, cx, cy}, dist5 = {k1, k2, p1, p2, k3}.
    int UndistortKeyPoints(const std::vector<orbcv::KeyPoint> &keys, const float K4[4], const float dist5[5],
                           std::vector<orbcv::KeyPoint> &keysUn)
    {
        keysUn = keys;
        if (!handle_ || keys.empty()) return 0;
        return check(orbgpu_undistort_keypoints(handle_, reinterpret_cast<const orbgpu_keypoint *>(keys.data()), (int32_t)keys.size(), K4,
                                                dist5, reinterpret_cast<orbgpu_keypoint *>(keysUn.data())));
    }

    // Frame::ComputeStereoFromRGBD (src/Frame.cc:777-798): depth = CV_32F image rows of strideFloats floats.
    int ComputeStereoFromRGBD(const std::vector<orbcv::KeyPoint> &keys, const std::vector<orbcv::KeyPoint> &keysUn, const float *depth,
                              int w, int h, int strideFloats, float bf, std::vector<float> &mvuRight, std::vector<float> &mvDepth)
    {
        mvuRight.assign(keys.size(), -1.0f); mvDepth.assign(keys.size(), -1.0f);
        if (!handle_ || keys.empty()) return 0;
        return check(orbgpu_stereo_from_rgbd(handle_, reinterpret_cast<const orbgpu_keypoint *>(keys.data()),
                                             reinterpret_cast<const orbgpu_keypoint *>(keysUn.data()), (int32_t)keys.size(), depth, w, h,
                                             strideFloats, bf, mvuRight.data(), mvDepth.data()));
    }

    // reference: include/ORBmatcher.h:70, src/ORBmatcher.cc:422-571
    int SearchForInitialization(const std::vector<orbcv::KeyPoint> &keys1, const orbcv::Mat &desc1,
                                const std::vector<orbcv::KeyPoint> &keys2, const orbcv::Mat &desc2, const float bounds[4],
                                std::vector<orbcv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10)
    {
        const int n1 = (int)keys1.size(), n2 = (int)keys2.size();
        vnMatches12.assign(n1, -1);
        if (!handle_) return 0;
        int32_t nmatches = 0;
        if (check(orbgpu_search_for_initialization(handle_, reinterpret_cast<const orbgpu_keypoint *>(keys1.data()), desc1.data,
                                                   n1, reinterpret_cast<const orbgpu_keypoint *>(keys2.data()), desc2.data, n2,
                                                   bounds, reinterpret_cast<float *>(vbPrevMatched.data()), vnMatches12.data(),
                                                   windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, &nmatches)))
            return 0;
        return nmatches;
    }

    // Frame::ComputeStereoMatches (src/Frame.cc:600-774); left/right hold the pyramids of the pair.
    int ComputeStereoMatches(ORBextractor &left, ORBextractor &right, const std::vector<orbcv::KeyPoint> &keysL,
                             const orbcv::Mat &descL, const std::vector<orbcv::KeyPoint> &keysR, const orbcv::Mat &descR,
                             float bf, float b, std::vector<float> &mvuRight, std::vector<float> &mvDepth)
    {
        const int nL = (int)keysL.size(), nR = (int)keysR.size();
        mvuRight.assign(nL, -1.0f); mvDepth.assign(nL, -1.0f);
        if (!handle_) return 0;
        int32_t nvalid = 0;
        if (check(orbgpu_stereo_match(handle_, left.Handle(), right.Handle(), 0,
                                      reinterpret_cast<const orbgpu_keypoint *>(keysL.data()), descL.data, nL,
                                      reinterpret_cast<const orbgpu_keypoint *>(keysR.data()), descR.data, nR, bf, b,
                                      mvuRight.data(), mvDepth.data(), &nvalid)))
            return 0;
        return nvalid;
    }

    // reference: include/ORBmatcher.h:98, src/ORBmatcher.cc:1661-1704: the three fullest bins of the rotation histogram,
    // the second and third only if they hold at least a tenth of the fullest.  Among equally full bins the lower index
    // ranks first (a bin must be strictly fuller to displace one).
    static void ComputeThreeMaxima(std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3)
    {
        int top[3] = {0, 0, 0}, at[3] = {-1, -1, -1};
        for (int i = 0; i < L; i++) {
            const int s = (int)histo[i].size();
            int rank = 0;
            while (rank < 3 && s <= top[rank]) rank++;
            for (int k = 2; k > rank; k--) { top[k] = top[k - 1]; at[k] = at[k - 1]; }
            if (rank < 3) { top[rank] = s; at[rank] = i; }
        }
        const float floor_ = 0.1f * (float)top[0];
        if ((float)top[1] < floor_) at[1] = at[2] = -1;
        else if ((float)top[2] < floor_) at[2] = -1;
        ind1 = at[0]; ind2 = at[1]; ind3 = at[2];
    }

    static const int TH_LOW = ORBGPU_TH_LOW;            // src/ORBmatcher.cc:38
    static const int TH_HIGH = ORBGPU_TH_HIGH;          // src/ORBmatcher.cc:37
    static const int HISTO_LENGTH = ORBGPU_HISTO_LENGTH;  // src/ORBmatcher.cc:39

    const std::string &LastError() const { return err_; }

protected:
    float mfNNratio;
    bool mbCheckOrientation;

private:
    int check(int rc)
    {
        if (rc != ORBGPU_OK) err_ = orbgpu_last_error();
        return rc;
    }
    int device_ = 0;
    orbgpu_matcher *handle_ = nullptr;
    std::string err_;
};

}  // namespace ORB_SLAM2

#include "ORBmatcherFrames.h"

#endif
