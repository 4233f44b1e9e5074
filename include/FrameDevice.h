/*
 * FrameDevice.h -- C++ holder of the device-resident frame state (orbgpu_frame in orbgpu.h).
 *
 * The constructors of ORB_SLAM2::Frame (reference: /root/reference/src/ORB_SLAM2/src/Frame.cc:58-116 stereo,
 * :119-172 RGB-D, :175-226 monocular) extract ORB features and then compute, on the CPU, the undistorted
 * keypoints, the stereo / depth measurements and the 64x48 feature grid.  DeviceFrame does those steps on the GPU
 * behind the extraction and keeps the results there; the matchers of ORBmatcher.h read them in place.  A Frame (or
 * KeyFrame) type opts in by carrying
 *
 *     std::shared_ptr<ORB_SLAM2::DeviceFrame> mpDeviceFrame;
 *
 * filled in its constructor (and copied with the Frame, and into the KeyFrame made from it):
 *
 *     mpDeviceFrame = std::make_shared<DeviceFrame>(extractor->MaxKeypoints());
 *     N = mpDeviceFrame->BuildMono(*extractor, im.data, im.cols, im.rows, (int)im.step, K4, dist5, bounds4);
 *     mpDeviceFrame->Keys(mvKeys); mpDeviceFrame->KeysUn(mvKeysUn); ...   // only the mirrors the program reads
 *
 * The host mirrors are ordinary copies fetched on request; nothing is downloaded that is not asked for.
 */
#ifndef ORBGPU_FRAME_DEVICE_H
#define ORBGPU_FRAME_DEVICE_H

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "ORBextractor.h"
#include "orbgpu.h"

namespace ORB_SLAM2 {

class DeviceFrame {
public:
    explicit DeviceFrame(int capacity, int device = 0)
    {
        if (orbgpu_frame_create(device, capacity, &handle_) != ORBGPU_OK) { handle_ = nullptr; err_ = orbgpu_last_error(); }
    }
    ~DeviceFrame() { orbgpu_frame_destroy(handle_); }
    DeviceFrame(const DeviceFrame &) = delete;
    DeviceFrame &operator=(const DeviceFrame &) = delete;

    // K4 = {fx, fy, cx, cy}; dist5 = {k1, k2, p1, p2, k3} (mDistCoef); bounds4 = {mnMinX, mnMinY, mnMaxX, mnMaxY}.
    // Each returns N (0 on failure, reason in LastError()).
    int BuildMono(ORBextractor &ex, const uint8_t *img, int w, int h, int stride, const float *K4, const float *dist5,
                  const float *bounds4)
    {
        int32_t n = 0;
        if (!handle_ || !check(orbgpu_frame_build_mono(handle_, ex.Handle(), img, w, h, stride, K4, dist5, bounds4, &n))) n = 0;
        ex.mvImagePyramid.Invalidate();
        return n;
    }
    int BuildRGBD(ORBextractor &ex, const uint8_t *img, int w, int h, int stride, const float *depth, int depth_stride_floats, float bf,
                  const float *K4, const float *dist5, const float *bounds4)
    {
        int32_t n = 0;
        if (!handle_ ||
            !check(orbgpu_frame_build_rgbd(handle_, ex.Handle(), img, w, h, stride, depth, depth_stride_floats, bf, K4, dist5, bounds4, &n)))
            n = 0;
        ex.mvImagePyramid.Invalidate();
        return n;
    }
    int BuildStereo(ORBextractor &left, ORBextractor &right, const uint8_t *imgL, const uint8_t *imgR, int w, int h, int stride, float bf,
                    float b, const float *K4, const float *dist5, const float *bounds4)
    {
        int32_t n = 0, nv = 0;
        if (!handle_ || !check(orbgpu_frame_build_stereo(handle_, left.Handle(), right.Handle(), imgL, imgR, w, h, stride, bf, b, K4, dist5,
                                                         bounds4, &n, &nv)))
            n = 0;
        left.mvImagePyramid.Invalidate();
        right.mvImagePyramid.Invalidate();
        return n;
    }

    // From host arrays (orbgpu_frame_upload): keys may be null (= keysUn), uRight / depth may be null (monocular).
    int Upload(const orbcv::KeyPoint *keys, const orbcv::KeyPoint *keysUn, const uint8_t *descriptors, const float *uRight,
               const float *depth, int n, int nlevels, const float *bounds4)
    {
        static_assert(sizeof(orbcv::KeyPoint) == sizeof(orbgpu_keypoint), "KeyPoint layout");
        if (!handle_ || !check(orbgpu_frame_upload(handle_, (const orbgpu_keypoint *)keys, (const orbgpu_keypoint *)keysUn, descriptors,
                                                   uRight, depth, n, nlevels, bounds4)))
            return 0;
        return n;
    }

    int N() const { int32_t n = 0; if (handle_) orbgpu_frame_size(handle_, &n, nullptr, nullptr); return n; }
    int NumWithDepth() const { int32_t n = 0; if (handle_) orbgpu_frame_size(handle_, nullptr, &n, nullptr); return n; }

    // host mirrors: mvKeys, mvKeysUn, mDescriptors (N rows of 32 bytes), mvuRight, mvDepth
    bool Keys(std::vector<orbcv::KeyPoint> &v) { v.resize((size_t)N()); return fetch((orbgpu_keypoint *)v.data(), nullptr, nullptr, nullptr, nullptr); }
    bool KeysUn(std::vector<orbcv::KeyPoint> &v) { v.resize((size_t)N()); return fetch(nullptr, (orbgpu_keypoint *)v.data(), nullptr, nullptr, nullptr); }
    bool Descriptors(std::vector<uint8_t> &v) { v.resize((size_t)N() * 32); return fetch(nullptr, nullptr, v.data(), nullptr, nullptr); }
    bool uRight(std::vector<float> &v) { v.resize((size_t)N()); return fetch(nullptr, nullptr, nullptr, v.data(), nullptr); }
    bool Depth(std::vector<float> &v) { v.resize((size_t)N()); return fetch(nullptr, nullptr, nullptr, nullptr, v.data()); }
    // mGrid[ix][iy] = cellIndex[cellStart[ix*48 + iy] .. cellStart[ix*48 + iy + 1])
    bool Grid(std::vector<int32_t> &cellStart, std::vector<int32_t> &cellIndex)
    {
        cellStart.assign((size_t)ORBGPU_GRID_COLS * ORBGPU_GRID_ROWS + 1, 0);
        cellIndex.assign((size_t)(N() > 0 ? N() : 1), 0);
        int32_t ng = 0;
        if (!handle_ || !check(orbgpu_frame_grid(handle_, cellStart.data(), cellIndex.data(), &ng))) return false;
        cellIndex.resize((size_t)ng);
        return true;
    }

    orbgpu_frame *Handle() const { return handle_; }
    const std::string &LastError() const { return err_; }

private:
    bool check(int rc)
    {
        if (rc != ORBGPU_OK) err_ = orbgpu_last_error();
        return rc == ORBGPU_OK;
    }
    bool fetch(orbgpu_keypoint *k, orbgpu_keypoint *ku, uint8_t *d, float *ur, float *dp)
    {
        if (!handle_ || N() == 0) return handle_ != nullptr;
        return check(orbgpu_frame_download(handle_, k, ku, d, ur, dp));
    }
    orbgpu_frame *handle_ = nullptr;
    std::string err_;
};

}  // namespace ORB_SLAM2

#endif
