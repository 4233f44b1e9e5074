/*
 * ORBmatcher.h -- host-side C++ mirror of the descriptor hot path of
 * ORB_SLAM2::ORBmatcher (reference:
 * /root/reference/src/ORB_SLAM2/include/ORBmatcher.h:41-103) on top of the C
 * ABI in orbgpu.h.
 *
 * The reference's search methods take Frame / KeyFrame / MapPoint objects; the
 * pose projection that turns a map point into an image window is host glue
 * outside the descriptor path.  The methods here therefore take the arrays
 * those objects hold (keypoints, descriptors, mvuRight, "has a map point"
 * flags) plus one window query per map point; INTEGRATION.md shows the
 * few-line adapters that give the Frame-based overloads back.
 */
#ifndef ORBGPU_ORBMATCHER_H
#define ORBGPU_ORBMATCHER_H

#include <string>
#include <vector>

#include "ORBextractor.h"
#include "orbgpu.h"

namespace ORB_SLAM2 {

class ORBmatcher {
public:
    // reference: include/ORBmatcher.h:41
    ORBmatcher(float nnratio = 0.6, bool checkOri = true, int device = 0) : mfNNratio(nnratio), mbCheckOrientation(checkOri)
    {
        if (orbgpu_matcher_create(device, &handle_) != ORBGPU_OK) { handle_ = nullptr; err_ = orbgpu_last_error(); }
    }
    ~ORBmatcher() { orbgpu_matcher_destroy(handle_); }
    ORBmatcher(const ORBmatcher &) = delete;
    ORBmatcher &operator=(const ORBmatcher &) = delete;

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

    // reference: include/ORBmatcher.h:92, src/ORBmatcher.cc:1661-1704 (on bin sizes)
    static void ComputeThreeMaxima(const int *histoSizes, const int L, int &ind1, int &ind2, int &ind3)
    {
        int max1 = 0, max2 = 0, max3 = 0;
        ind1 = ind2 = ind3 = -1;
        for (int i = 0; i < L; i++) {
            const int s = histoSizes[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
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
    orbgpu_matcher *handle_ = nullptr;
    std::string err_;
};

}  // namespace ORB_SLAM2

#endif
