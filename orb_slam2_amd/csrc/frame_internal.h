// The device-resident frame handle, shared by frame_api.hip (which builds it) and matcher_api.hip (whose *_frame
// entry points read it).
#pragma once
#include <vector>

#include "match_internal.h"

struct orbgpu_frame {
    int device = 0;
    int capacity = 0;     // keypoints the buffers hold
    int n = 0;            // keypoints of the last build (host copy)
    int nvalid = 0;       // keypoints with a stereo / depth measurement
    int nlevels = 0;      // pyramid levels of the extractor that built it: octaves are 0 .. nlevels-1
    bool built = false;
    float bounds4[4] = {0, 0, 0, 0};
    // one allocation: keypoints | undistorted keypoints | descriptors | uRight | depth | counts {n, nvalid, -, -} | grid
    void *block = nullptr;
    orbgpu_keypoint *kps = nullptr, *kps_un = nullptr;
    uint8_t *desc = nullptr;
    float *uright = nullptr, *depth = nullptr;
    int32_t *count = nullptr;
    // the 64x48 grid in the window matcher's layout: the scratch block of a pair with one query
    // (orbgpu::match_scratch_ints(1, capacity)); the grid proper starts kTopK ints in:
    // cell_start [3080] | grid records [4*capacity] | grid descriptors [8*capacity]
    int32_t *gscratch = nullptr;
    const int32_t *grid() const { return gscratch + orbgpu::kTopK; }
    // stereo working set, allocated by the first stereo / RGB-D build: best SAD per left keypoint, row table, depth image
    void *stereo_block = nullptr;
    size_t stereo_bytes = 0;
    hipEvent_t ev = nullptr;  // joins the right extractor's stream
    // host copies of the keypoints and descriptors, taken at build time from the extractor's page-locked output block
    // (which the extraction fills anyway): orbgpu_frame_download serves mvKeys / mDescriptors -- and mvKeysUn of an
    // undistorted camera -- from here without touching the GPU
    std::vector<orbgpu_keypoint> h_kps;
    std::vector<uint8_t> h_desc;
    bool h_valid = false, distorted = false;
};
