// Minimal C++ driver of the drop-in classes: reads two PGM-less raw frames
// (width height, then w*h bytes each) from a file, extracts ORB features with
// ORB_SLAM2::ORBextractor and matches them with
// ORB_SLAM2::ORBmatcher::SearchForInitialization -- the calls Tracking makes
// (src/Tracking.cc:160-168, 738-748 of the reference) -- and prints a summary.
// With a vocabulary text file as second argument it also runs Frame::ComputeBoW
// (src/Frame.cc:503-510) on the first frame through ORB_SLAM2::ORBVocabulary.
//
// Build: g++ -std=c++14 -Iinclude examples/mono_frontend.cc -Lorb_slam2_amd -lorbgpu -Wl,-rpath,$PWD/orb_slam2_amd
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <memory>

#include "FrameDevice.h"
#include "ORBVocabulary.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"

// What a Frame has to carry for ORBmatcher::SearchForInitialization(Frame&, Frame&, ...) when its arrays stay on the
// GPU: the undistorted keypoints on the host (the caller reads their positions), the bounds, and the device copy.
struct MiniFrame {
    std::vector<orbcv::KeyPoint> mvKeysUn;
    orbcv::Mat mDescriptors;  // stays empty: the descriptors are read on the device
    float mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0;
    std::shared_ptr<ORB_SLAM2::DeviceFrame> mpDeviceFrame;
};

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s frames.raw [vocabulary.txt]\n", argv[0]); return 2; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) { std::perror("open"); return 2; }
    int wh[2];
    if (std::fread(wh, sizeof(int), 2, f) != 2) return 2;
    const int w = wh[0], h = wh[1];
    std::vector<uint8_t> buf[2];
    orbcv::Mat im[2];
    for (int i = 0; i < 2; i++) {
        buf[i].resize((size_t)w * h);
        if (std::fread(buf[i].data(), 1, buf[i].size(), f) != buf[i].size()) return 2;
        im[i] = orbcv::Mat(h, w, buf[i].data(), (size_t)w);
    }
    std::fclose(f);

    ORB_SLAM2::ORBextractor extractor(1000, 1.2f, 8, 20, 7, 0, w, h);
    if (!extractor.Handle()) { std::fprintf(stderr, "extractor: %s\n", extractor.LastError().c_str()); return 1; }
    std::vector<orbcv::KeyPoint> keys[2];
    orbcv::Mat desc[2], nomask;
    for (int i = 0; i < 2; i++) extractor(im[i], nomask, keys[i], desc[i]);

    ORB_SLAM2::ORBmatcher matcher(0.9f, true);
    std::vector<orbcv::Point2f> prev(keys[0].size());
    for (size_t i = 0; i < keys[0].size(); i++) prev[i] = keys[0][i].pt;
    std::vector<int> matches12;
    const float bounds[4] = {0.f, 0.f, (float)w, (float)h};
    const int n = matcher.SearchForInitialization(keys[0], desc[0], keys[1], desc[1], bounds, prev, matches12, 100);
    long sum = 0;
    for (size_t i = 0; i < matches12.size(); i++) sum = (sum * 31 + (matches12[i] + 2)) % 1000000007L;
    std::printf("keypoints %zu %zu pyramid0 %dx%d matches %d checksum %ld\n", keys[0].size(), keys[1].size(),
                extractor.mvImagePyramid[0].cols, extractor.mvImagePyramid[0].rows, n, sum);
    if (argc > 2) {
        ORB_SLAM2::ORBVocabulary voc;
        if (!voc.loadFromTextFile(argv[2])) { std::fprintf(stderr, "vocabulary: %s\n", voc.LastError().c_str()); return 1; }
        DBoW2::BowVector bow;
        DBoW2::FeatureVector fv;
        voc.transform(desc[0], bow, fv, 2);
        long fsum = 0;
        size_t nfeat = 0;
        for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it)
            for (size_t i = 0; i < it->second.size(); i++, nfeat++) fsum = (fsum * 31 + it->first * 7 + it->second[i]) % 1000000007L;
        double total = 0;
        for (DBoW2::BowVector::const_iterator it = bow.begin(); it != bow.end(); ++it) total += it->second;
        std::printf("words %u bow %zu sum %.17g nodes %zu features %zu checksum %ld\n", voc.size(), bow.size(), total, fv.size(), nfeat,
                    fsum);
    }
    // Per-frame latency of the two calls as Tracking makes them (src/Tracking.cc:160-168, 738-748): operator() on a
    // resident extractor, and an ORBmatcher built on the stack for one search.  Median of 50 after 5 warm-ups, with
    // std::chrono::steady_clock around each call like the reference's drivers (Examples/Monocular/mono_tum.cc:79-96).
    typedef std::chrono::steady_clock clk;
    std::vector<double> te, tm, tc;
    for (int it = 0; it < 55; it++) {
        const clk::time_point t0 = clk::now();
        extractor(im[it & 1], nomask, keys[it & 1], desc[it & 1]);
        const clk::time_point t1 = clk::now();
        int nm;
        clk::time_point t2;
        {
            ORB_SLAM2::ORBmatcher onStack(0.9f, true);
            t2 = clk::now();
            for (size_t i = 0; i < keys[0].size(); i++) prev[i] = keys[0][i].pt;
            nm = onStack.SearchForInitialization(keys[0], desc[0], keys[1], desc[1], bounds, prev, matches12, 100);
        }
        const clk::time_point t3 = clk::now();
        if (nm != n) { std::fprintf(stderr, "match count changed between calls: %d vs %d\n", nm, n); return 1; }
        if (it >= 5) {
            te.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
            tc.push_back(std::chrono::duration<double, std::micro>(t2 - t1).count());
            tm.push_back(std::chrono::duration<double, std::milli>(t3 - t1).count());
        }
    }
    std::sort(te.begin(), te.end()); std::sort(tm.begin(), tm.end()); std::sort(tc.begin(), tc.end());
    std::printf("latency extract_ms %.4f stack_matcher_ms %.4f matcher_ctor_us %.2f\n", te[te.size() / 2], tm[tm.size() / 2], tc[tc.size() / 2]);

    // The same two steps with device-resident frames: the constructor part of Frame (extraction, undistortion, grid;
    // src/Frame.cc:175-226) leaves everything on the GPU and fetches mvKeysUn only, and the search reads both frames there.
    const float K4[4] = {517.3f, 516.5f, 318.6f, 255.3f}, D5[5] = {0, 0, 0, 0, 0};
    MiniFrame fr[2];
    for (int i = 0; i < 2; i++) {
        fr[i].mpDeviceFrame = std::make_shared<ORB_SLAM2::DeviceFrame>(extractor.MaxKeypoints());
        fr[i].mnMaxX = (float)w; fr[i].mnMaxY = (float)h;
    }
    std::vector<double> tb, ts;
    for (int it = 0; it < 55; it++) {
        MiniFrame &F = fr[it & 1];
        const clk::time_point t0 = clk::now();
        const int N = F.mpDeviceFrame->BuildMono(extractor, im[it & 1].data, w, h, w, K4, D5, bounds);
        F.mpDeviceFrame->KeysUn(F.mvKeysUn);
        const clk::time_point t1 = clk::now();
        if (N != (int)keys[it & 1].size()) { std::fprintf(stderr, "device frame: %d keypoints, operator() %zu (%s)\n", N, keys[it & 1].size(), F.mpDeviceFrame->LastError().c_str()); return 1; }
        if (it == 0) continue;  // the other frame is not built yet
        int nm;
        {
            ORB_SLAM2::ORBmatcher onStack(0.9f, true);
            for (size_t i = 0; i < fr[0].mvKeysUn.size(); i++) prev[i] = fr[0].mvKeysUn[i].pt;
            nm = onStack.SearchForInitialization(fr[0], fr[1], prev, matches12, 100);
        }
        const clk::time_point t2 = clk::now();
        if (nm != n) { std::fprintf(stderr, "device frames: %d matches, arrays %d\n", nm, n); return 1; }
        if (it >= 5) {
            tb.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
            ts.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
    }
    std::sort(tb.begin(), tb.end()); std::sort(ts.begin(), ts.end());
    std::printf("device_frames build_ms %.4f stack_matcher_ms %.4f\n", tb[tb.size() / 2], ts[ts.size() / 2]);

    // The pattern of a Frame constructor: a NEW DeviceFrame per image (make_shared), built, its keypoints fetched, and the
    // frame before the last one destroyed when its shared_ptr goes -- create and destroy recycle the frame's device block.
    std::vector<double> tp;
    for (int it = 0; it < 55; it++) {
        const clk::time_point t0 = clk::now();
        MiniFrame &F = fr[it & 1];
        F.mpDeviceFrame = std::make_shared<ORB_SLAM2::DeviceFrame>(extractor.MaxKeypoints());  // the old one is destroyed here
        const int N = F.mpDeviceFrame->BuildMono(extractor, im[it & 1].data, w, h, w, K4, D5, bounds);
        F.mpDeviceFrame->KeysUn(F.mvKeysUn);
        const clk::time_point t1 = clk::now();
        if (N != (int)keys[it & 1].size()) { std::fprintf(stderr, "recycled device frame: %d keypoints, operator() %zu\n", N, keys[it & 1].size()); return 1; }
        if (it >= 5) tp.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
    }
    std::sort(tp.begin(), tp.end());
    std::printf("device_frames_per_frame make_build_destroy_ms %.4f\n", tp[tp.size() / 2]);

    // An image larger than the extractor's working set: operator() grows the handle once instead of failing (the
    // reference accepts any image size).  Checked against an extractor that was created for the larger size.
    {
        const int W2 = w + 64, H2 = h + 32;
        std::vector<uint8_t> big((size_t)W2 * H2);
        for (int y = 0; y < H2; y++)
            for (int x = 0; x < W2; x++) big[(size_t)y * W2 + x] = buf[0][(size_t)(y % h) * w + (x % w)];
        orbcv::Mat imBig(H2, W2, big.data(), (size_t)W2);
        ORB_SLAM2::ORBextractor sized(1000, 1.2f, 8, 20, 7, 0, W2, H2);
        std::vector<orbcv::KeyPoint> kGrown, kSized;
        orbcv::Mat dGrown, dSized;
        extractor(imBig, nomask, kGrown, dGrown);  // created for w x h
        sized(imBig, nomask, kSized, dSized);
        const bool same = !kGrown.empty() && kGrown.size() == kSized.size() &&
                          std::memcmp(kGrown.data(), kSized.data(), kGrown.size() * sizeof(orbcv::KeyPoint)) == 0 &&
                          std::memcmp(dGrown.data, dSized.data, kGrown.size() * 32) == 0;
        extractor(im[0], nomask, keys[0], desc[0]);  // and the original size still works on the grown handle
        std::printf("grown_handle keypoints %zu equal %d small_again %zu\n", kGrown.size(), same ? 1 : 0, keys[0].size());
        if (!same) { std::fprintf(stderr, "grown handle: %s\n", extractor.LastError().c_str()); return 1; }
    }
    return 0;
}
