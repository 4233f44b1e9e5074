/*
 * ORBextractor.h -- host-side C++ mirror of ORB_SLAM2::ORBextractor
 * (reference: /root/reference/src/ORB_SLAM2/include/ORBextractor.h:44-113) on
 * top of the C ABI in orbgpu.h.  Same constructor, operator(), getters and the
 * public mvImagePyramid member, so Frame.cc / Tracking.cc compile unchanged.
 * operator() moves one image to the device and the keypoints and descriptors
 * back; pyramid levels cross the bus only if mvImagePyramid[level] is read.
 *
 * With OpenCV available define ORBGPU_WITH_OPENCV before including this file:
 * the class then takes cv::InputArray / cv::OutputArray / cv::KeyPoint exactly
 * like the reference.  Without it (this repository's build: OpenCV is absent)
 * a minimal stand-in for the three cv types used by the signature is provided
 * in namespace orbcv; it has the same field names and binary layout.
 *
 * Error behaviour: the reference's operator() is void and never throws; it
 * returns silently on an empty image (src/ORBextractor.cc:1459-1460).  Here a
 * failing GPU call leaves the outputs empty and the reason in LastError().
 *
 * Image size: the reference accepts any image.  The device working set is sized
 * for max_width x max_height; an image beyond it makes operator() build a larger
 * handle once (tables, buffers and launch graphs for the new size, a few
 * milliseconds) and go on with that one -- Handle() then returns the new handle.
 */
#ifndef ORBGPU_ORBEXTRACTOR_H
#define ORBGPU_ORBEXTRACTOR_H

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "orbgpu.h"

#ifdef ORBGPU_WITH_OPENCV
#include <opencv2/core/core.hpp>
namespace orbcv = cv;
#else
namespace orbcv {
struct Point2f { float x, y; };
struct KeyPoint {  // binary layout of cv::KeyPoint
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
// 8-bit single-channel matrix: owns its pixels unless constructed as a view
struct Mat {
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t *data = nullptr;
    std::vector<uint8_t> store;
    Mat() {}
    Mat(int r, int c, uint8_t *ext, size_t st) : rows(r), cols(c), step(st), data(ext) {}
    void create(int r, int c) { rows = r; cols = c; step = (size_t)c; store.assign((size_t)r * c, 0); data = store.data(); }
    void release() { rows = cols = 0; step = 0; data = nullptr; store.clear(); }
    bool empty() const { return rows == 0 || cols == 0 || !data; }
    uint8_t *ptr(int r) { return data + (size_t)r * step; }
    const uint8_t *ptr(int r) const { return data + (size_t)r * step; }
};
typedef const Mat &InputArray;
typedef Mat &OutputArray;
}  // namespace orbcv
#endif

namespace ORB_SLAM2 {

// mvImagePyramid of the reference is a std::vector<cv::Mat> that every extraction fills
// (include/ORBextractor.h:85).  Its only readers are `mvImagePyramid[level]` expressions in
// Frame::ComputeStereoMatches (src/Frame.cc:607,697,709,714) -- and a program whose stereo matching runs
// on the GPU (ORBmatcher::ComputeStereoMatches) never reads it.  This stand-in keeps that expression
// working but copies a level from the device only when it is first asked for after an extraction.
class ImagePyramidMirror {
public:
    size_t size() const { return levels_.size(); }
    bool empty() const { return levels_.empty(); }
    orbcv::Mat &operator[](size_t level) { fetch(level); return levels_[level]; }
    const orbcv::Mat &operator[](size_t level) const { fetch(level); return levels_[level]; }
    orbcv::Mat &at(size_t level) { fetch(level); return levels_.at(level); }
    // the handle has extracted another image: host copies of its levels are stale (every caller of an extract entry
    // point of the C ABI on Handle() says so here; operator() does it itself)
    void Invalidate() { fresh_.assign(fresh_.size(), 0); }

private:
    friend class ORBextractor;
    void bind(orbgpu_extractor *h, int nlevels) { handle_ = h; levels_.resize((size_t)nlevels); fresh_.assign((size_t)nlevels, 0); }
    void fetch(size_t l) const
    {
        if (!handle_ || l >= levels_.size() || fresh_[l]) return;
        int32_t lw = 0, lh = 0;
        if (orbgpu_pyramid_level(handle_, 0, (int32_t)l, 0, 0, nullptr, 0, &lw, &lh) != ORBGPU_OK) return;
#ifdef ORBGPU_WITH_OPENCV
        levels_[l].create(lh, lw, CV_8U);
#else
        levels_[l].create(lh, lw);
#endif
        if (orbgpu_pyramid_level(handle_, 0, (int32_t)l, 0, 0, levels_[l].data, (int)levels_[l].step, &lw, &lh) == ORBGPU_OK) fresh_[l] = 1;
    }
    orbgpu_extractor *handle_ = nullptr;
    mutable std::vector<orbcv::Mat> levels_;
    mutable std::vector<char> fresh_;
};

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // reference: include/ORBextractor.h:51-52.  max_width/max_height size the
    // device working set (defaults cover the shipped TUM/EuRoC/KITTI configs).
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0,
                 int max_width = 1280, int max_height = 720)
        : nfeatures(nfeatures), scaleFactor(scaleFactor), nlevels(nlevels), iniThFAST(iniThFAST), minThFAST(minThFAST),
          device_(device)
    {
        if (!Rebuild(max_width, max_height)) return;
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        int32_t n = 0;
        orbgpu_extractor_tables(handle_, &n, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                                mvInvLevelSigma2.data(), mnFeaturesPerLevel.data());
    }
    ~ORBextractor() { orbgpu_extractor_destroy(handle_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // reference: include/ORBextractor.h:59-61, src/ORBextractor.cc:1455-1521.  The mask is ignored.
    void operator()(orbcv::InputArray image, orbcv::InputArray /*mask*/, std::vector<orbcv::KeyPoint> &keypoints,
                    orbcv::OutputArray descriptors)
    {
#ifdef ORBGPU_WITH_OPENCV
        cv::Mat im = image.getMat();
        if (im.empty()) return;
        CV_Assert(im.type() == CV_8UC1);
        const uint8_t *pix = im.data; const int w = im.cols, h = im.rows; const int step = (int)im.step;
#else
        if (image.empty()) return;
        const uint8_t *pix = image.data; const int w = image.cols, h = image.rows; const int step = (int)image.step;
#endif
        keypoints.clear();
        if (!handle_) return;
        // an image larger than the working set: a larger handle, once (the reference takes any size)
        if ((w > max_w_ || h > max_h_) && !Rebuild(w > max_w_ ? w : max_w_, h > max_h_ ? h : max_h_)) {
            descriptors.release();
            return;
        }
        static_assert(sizeof(orbcv::KeyPoint) == sizeof(orbgpu_keypoint), "KeyPoint layout");
        kp_buf_.resize((size_t)max_kp_);
        desc_buf_.resize((size_t)max_kp_ * 32);
        int32_t n = 0;
        if (orbgpu_extract(handle_, pix, w, h, step, kp_buf_.data(), desc_buf_.data(), max_kp_, &n) != ORBGPU_OK) {
            err_ = orbgpu_last_error();
            n = 0;
        }
        keypoints.resize((size_t)n);
        if (n) std::memcpy((void *)keypoints.data(), kp_buf_.data(), (size_t)n * sizeof(orbgpu_keypoint));
#ifdef ORBGPU_WITH_OPENCV
        if (n == 0) descriptors.release();
        else { descriptors.create(n, 32, CV_8U); std::memcpy(descriptors.getMat().data, desc_buf_.data(), (size_t)n * 32); }
#else
        if (n == 0) descriptors.release();
        else { descriptors.create(n, 32); std::memcpy(descriptors.data, desc_buf_.data(), (size_t)n * 32); }
#endif
        mvImagePyramid.Invalidate();  // levels are copied from the device when somebody reads them
    }

    // reference: include/ORBextractor.h:63-83
    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    ImagePyramidMirror mvImagePyramid;  // include/ORBextractor.h:85; read as mvImagePyramid[level]

    orbgpu_extractor *Handle() { return handle_; }  // for ORBmatcher::ComputeStereoMatches and DeviceFrame
    int MaxKeypoints() const { return max_kp_; }   // upper bound of keypoints per image: capacity of a DeviceFrame
    const std::string &LastError() const { return err_; }

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

private:
    // (Re)creates the device handle for images up to max_width x max_height.  The new handle is built before the old
    // one is released, so a failure (LastError()) leaves the extractor as it was.
    bool Rebuild(int max_width, int max_height)
    {
        orbgpu_params p;
        std::memset(&p, 0, sizeof(p));
        p.nfeatures = nfeatures; p.scaleFactor = (float)scaleFactor; p.nlevels = nlevels;
        p.iniThFAST = iniThFAST; p.minThFAST = minThFAST; p.device = device_;
        p.max_width = max_width; p.max_height = max_height; p.max_batch = 1;
        orbgpu_extractor *h = nullptr;
        if (orbgpu_extractor_create(&p, &h) != ORBGPU_OK) { err_ = orbgpu_last_error(); return false; }
        if (handle_) orbgpu_extractor_destroy(handle_);
        handle_ = h;
        max_w_ = max_width; max_h_ = max_height;
        orbgpu_extractor_max_keypoints(handle_, &max_kp_);
        mvImagePyramid.bind(handle_, nlevels);
        return true;
    }

    orbgpu_extractor *handle_ = nullptr;
    int device_ = 0, max_w_ = 0, max_h_ = 0;
    int32_t max_kp_ = 0;
    std::vector<orbgpu_keypoint> kp_buf_;
    std::vector<uint8_t> desc_buf_;
    std::string err_;
};

}  // namespace ORB_SLAM2

#endif
