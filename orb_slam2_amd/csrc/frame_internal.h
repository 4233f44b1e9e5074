// The device-resident frame handle, shared by frame_api.hip (which builds it) and matcher_api.hip (whose *_frame
// entry points read it).
#pragma once
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "match_internal.h"

// One helper thread of a frame handle: the stereo build enqueues the right image's extraction on it while the calling
// thread enqueues the left one (staging copy + graph launch cost the host about 70 us per image; the reference runs
// the two extractions on two threads as well, src/Frame.cc:78-81).  Started by the first stereo build.
struct FrameWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = false, quit = false;
    void start()
    {
        if (th.joinable()) return;
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                std::function<void()> j = std::move(job);
                has_job = false;
                lk.unlock();
                j();
                lk.lock();
                done = true;
                cv.notify_all();
            }
        });
    }
    void submit(std::function<void()> j)
    {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
    }
    void stop()
    {
        if (!th.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
            cv.notify_all();
        }
        th.join();
    }
};

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
    FrameWorker worker;
};
