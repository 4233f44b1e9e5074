// The device-resident frame: what the constructors of Frame compute directly after the extraction
// (src/Frame.cc:58-116 stereo, :119-172 RGB-D, :175-226 monocular) -- UndistortKeyPoints, ComputeStereoMatches or
// ComputeStereoFromRGBD, AssignFeaturesToGrid -- enqueued behind the extraction on the extractor's stream, with the
// results kept in the frame's own device buffers.  The host waits once, for the keypoint count.
// Reference lines are relative to /root/reference/src/ORB_SLAM2/.
#include <math.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "frame_internal.h"
#include "orbgpu_internal.h"

namespace orbgpu {
int launch_grid_only(hipStream_t st, const MatchArgs &A);
int launch_stereo(hipStream_t st, const StereoArgs &A);
int launch_stereo_rows(hipStream_t st, const orbgpu_keypoint *kpsR, const int32_t *d_nR, int capR, const ScaleTab &scale, int nlevels, int nRows,
                       int32_t *spans, int32_t *row_start, int32_t *row_items);
int launch_stereo_outliers(hipStream_t st, const int32_t *best_sad, const int32_t *d_nL, int capL, float *uRight, float *depth,
                           int32_t *nvalid);

struct FrameOut {
    orbgpu_keypoint *kps, *kps_un;
    uint8_t *desc;
    float *uright, *depth;
    int32_t *count;
    int capacity;
};

// mode 0: monocular (mvuRight = mvDepth = -1, src/Frame.cc:203-204); 1: RGB-D (ComputeStereoFromRGBD, :777-798: depth
// image lookup at the distorted keypoint, truncated to integer pixel coordinates as cv::Mat::at<float>(float, float)
// does); 2: stereo (k_stereo fills uRight / depth afterwards).  The count arrives in device memory.
__global__ __launch_bounds__(256) void k_frame_finish(const orbgpu_keypoint *__restrict__ src_kps, const uint8_t *__restrict__ src_desc,
                                                      const int32_t *__restrict__ src_count, FrameOut F, UndistortArgs U, int distorted,
                                                      const float *__restrict__ depth_img, int w, int h, int stride_floats, float bf,
                                                      int mode)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = min(max(*src_count, 0), F.capacity);
    if (i == 0) {
        F.count[0] = n;
        if (mode != 1) F.count[1] = 0;  // RGB-D counts its depths with atomics: the host clears that word before the launch
    }
    if (i >= n) return;
    const orbgpu_keypoint kp = src_kps[i];
    orbgpu_keypoint un = kp;
    if (distorted) undistort_point(U, kp.x, kp.y, &un.x, &un.y);  // :519-523: a plain copy when k1 == 0
    F.kps[i] = kp;
    F.kps_un[i] = un;
    const uint4 *sd = reinterpret_cast<const uint4 *>(src_desc) + 2 * (size_t)i;
    uint4 *dd = reinterpret_cast<uint4 *>(F.desc) + 2 * (size_t)i;
    dd[0] = sd[0];
    dd[1] = sd[1];
    if (mode == 0) {
        F.uright[i] = -1.0f;
        F.depth[i] = -1.0f;
    } else if (mode == 1) {
        const int v = (int)kp.y, u = (int)kp.x;
        float d = 0.0f;
        if (u >= 0 && u < w && v >= 0 && v < h) d = depth_img[(size_t)v * stride_floats + u];
        if (d > 0) {
            F.depth[i] = d;
            F.uright[i] = __fsub_rn(un.x, __fdiv_rn(bf, d));
            atomicAdd(&F.count[1], 1);
        } else {
            F.depth[i] = -1.0f;
            F.uright[i] = -1.0f;
        }
    }
}

}  // namespace orbgpu

using namespace orbgpu;

// Frame blocks are recycled like matcher contexts: the reference constructs a Frame per image (src/Tracking.cc:240, 270,
// 330) and destroys the one before the last, so create / destroy must not go to the allocator (hipMalloc, hipFree and
// the device-wide waits they imply would stall every other stream of the process once per frame).
// orbgpu_frame_destroy parks an idle frame of the same device here, orbgpu_frame_create takes one of the wanted
// capacity back.
static std::mutex g_frame_pool_mutex;
static std::vector<orbgpu_frame *> g_frame_pool[64];
static const size_t kFramePoolPerDevice = 16;

// A build that fails after its extraction was enqueued must not return while the extractor's streams still run: the
// next call on those handles rewrites their staging and output blocks at once.
static int fail_after_enqueue(int rc, const HostExtraction *XL, const HostExtraction *XR)
{
    if (rc != ORBGPU_OK) {
        if (XL && XL->stream) (void)hipStreamSynchronize(XL->stream);
        if (XR && XR->stream) (void)hipStreamSynchronize(XR->stream);
    }
    return rc;
}

static int frame_check(const char *who, orbgpu_frame *f, const float *K4, const float *dist5, const float *bounds4, int32_t *n_out)
{
    if (!f || !K4 || !dist5 || !bounds4 || !n_out || K4[0] == 0.0f || K4[1] == 0.0f || !(bounds4[2] > bounds4[0]) ||
        !(bounds4[3] > bounds4[1])) {
        set_error("%s: invalid argument", who);
        return ORBGPU_E_INVALID;
    }
    return ORBGPU_OK;
}

static int frame_stereo_block(orbgpu_frame *f, size_t bytes)
{
    if (bytes <= f->stereo_bytes) return ORBGPU_OK;
    if (f->stereo_block) (void)hipFree(f->stereo_block);
    f->stereo_block = nullptr;
    f->stereo_bytes = 0;
    if (hipMalloc(&f->stereo_block, bytes) != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed", bytes);
        return ORBGPU_E_NOMEM;
    }
    f->stereo_bytes = bytes;
    return ORBGPU_OK;
}

// Frame::AssignFeaturesToGrid (src/Frame.cc:270-293) into the frame's grid block, in the window matcher's layout.
static int frame_grid_enqueue(orbgpu_frame *f, hipStream_t st)
{
    MatchArgs A;
    memset(&A, 0, sizeof(A));
    A.mode = 1;
    A.nq = f->count + 3;  // always 0
    A.q_cap = 1;
    A.kps2 = f->kps_un; A.desc2 = f->desc; A.n2 = f->count; A.t_cap = f->capacity;
    A.gminX = A.qminX = f->bounds4[0]; A.gminY = A.qminY = f->bounds4[1];  // src/Frame.cc:101-102
    A.invW = (float)ORBGPU_GRID_COLS / (f->bounds4[2] - f->bounds4[0]);
    A.invH = (float)ORBGPU_GRID_ROWS / (f->bounds4[3] - f->bounds4[1]);
    A.scratch = f->gscratch;
    return launch_grid_only(st, A);
}

static int frame_finish_enqueue(orbgpu_frame *f, const HostExtraction &X, const float *K4, const float *dist5, int mode,
                                const float *d_depth, int w, int h, float bf)
{
    FrameOut F = {f->kps, f->kps_un, f->desc, f->uright, f->depth, f->count, f->capacity};
    f->distorted = dist5[0] != 0.0f;
    if (mode == 1) ORBGPU_HIP_CHECK(hipMemsetAsync(f->count, 0, 16, X.stream));
    hipLaunchKernelGGL(k_frame_finish, dim3((f->capacity + 255) / 256), dim3(256), 0, X.stream, X.d_kps, X.d_desc, X.d_counts, F,
                       undistort_args(K4, dist5), dist5[0] != 0.0f ? 1 : 0, d_depth, w, h, w, bf, mode);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

// Wait for the build, check the extraction's flags and note the counts on the host.  The keypoint count, the
// keypoints and the descriptors are in the extractor's page-locked block by then; only a stereo / RGB-D build
// has a second number to fetch.
static int frame_complete(orbgpu_frame *f, const HostExtraction &X, const HostExtraction *XR, bool want_nvalid, int32_t *n_out)
{
    int32_t hc[4] = {0, 0, 0, 0};
    if (want_nvalid) ORBGPU_HIP_CHECK(hipMemcpyAsync(hc, f->count, 16, hipMemcpyDeviceToHost, X.stream));
    {
        const int rcw = wait_stream_short(X.stream, 400);
        if (rcw) return rcw;
    }
    if (*X.h_status || (XR && *XR->h_status)) {
        set_error("internal capacity exceeded (status %d)", *X.h_status ? *X.h_status : *XR->h_status);
        return ORBGPU_E_CAPACITY;
    }
    const int n = X.h_counts[0];
    if (n > f->capacity) {
        set_error("the frame holds %d keypoints, the image yields %d", f->capacity, n);
        return ORBGPU_E_CAPACITY;
    }
    f->h_kps.assign(X.h_kps, X.h_kps + n);
    f->h_desc.assign(X.h_desc, X.h_desc + (size_t)n * 32);
    f->h_valid = true;
    f->n = n;
    f->nvalid = hc[1];
    f->built = true;
    *n_out = f->n;
    return ORBGPU_OK;
}

static int frame_stereo_enqueue(orbgpu_frame *f, orbgpu_extractor *left, orbgpu_extractor *right, const HostExtraction &XL,
                                const HostExtraction &XR, StereoArgs &S, int32_t w, int32_t h, float bf, float b, const float *K4,
                                const float *dist5);

static int frame_begin(orbgpu_frame *f, orbgpu_extractor *e, const float *bounds4)
{
    ORBGPU_HIP_CHECK(hipSetDevice(f->device));
    int32_t nlev = 0;
    int rc = orbgpu_extractor_tables(e, &nlev, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    f->nlevels = nlev;
    f->built = false;
    f->h_valid = false;
    memcpy(f->bounds4, bounds4, sizeof(f->bounds4));
    return ORBGPU_OK;
}

extern "C" {

int orbgpu_frame_create(int32_t device, int32_t capacity, orbgpu_frame **out)
{
    if (!out || capacity < 1 || capacity > (1 << 20)) {
        set_error("orbgpu_frame_create: invalid argument");
        return ORBGPU_E_INVALID;
    }
    *out = nullptr;
    ORBGPU_HIP_CHECK(hipSetDevice(device));
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lock(g_frame_pool_mutex);
        std::vector<orbgpu_frame *> &pool = g_frame_pool[device];
        for (size_t i = 0; i < pool.size(); i++)
            if (pool[i]->capacity == capacity) {
                *out = pool[i];
                pool.erase(pool.begin() + (long)i);
                return ORBGPU_OK;
            }
    }
    orbgpu_frame *f = new (std::nothrow) orbgpu_frame;
    if (!f) return ORBGPU_E_NOMEM;
    f->device = device;
    f->capacity = capacity;
    const size_t cap = (size_t)capacity;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_kps = 0, o_un = o_kps + up(cap * sizeof(orbgpu_keypoint)), o_desc = o_un + up(cap * sizeof(orbgpu_keypoint)),
                 o_ur = o_desc + up(cap * 32), o_dep = o_ur + up(cap * 4), o_cnt = o_dep + up(cap * 4), o_grid = o_cnt + 256,
                 total = o_grid + up(match_scratch_ints(1, capacity) * 4);
    if (hipMalloc(&f->block, total) != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed", total);
        delete f;
        return ORBGPU_E_NOMEM;
    }
    uint8_t *b = (uint8_t *)f->block;
    f->kps = (orbgpu_keypoint *)(b + o_kps);
    f->kps_un = (orbgpu_keypoint *)(b + o_un);
    f->desc = b + o_desc;
    f->uright = (float *)(b + o_ur);
    f->depth = (float *)(b + o_dep);
    f->count = (int32_t *)(b + o_cnt);
    f->gscratch = (int32_t *)(b + o_grid);
    if (hipMemset(f->count, 0, 256) != hipSuccess || hipEventCreateWithFlags(&f->ev, hipEventDisableTiming) != hipSuccess) {
        set_error("orbgpu_frame_create: device setup failed");
        (void)hipFree(f->block);
        delete f;
        return ORBGPU_E_HIP;
    }
    *out = f;
    return ORBGPU_OK;
}

// Nothing of the library reads a frame after the call that was given it has returned (the build calls and the *_frame
// searches wait for their stream), so there is nothing to wait for here; a caller that runs kernels of its own on
// the views of orbgpu_frame_device finishes them first.
void orbgpu_frame_destroy(orbgpu_frame *f)
{
    if (!f) return;
    if (f->device >= 0 && f->device < 64) {
        f->built = false;
        f->h_valid = false;
        f->n = f->nvalid = 0;
        std::lock_guard<std::mutex> lock(g_frame_pool_mutex);
        if (g_frame_pool[f->device].size() < kFramePoolPerDevice) {
            g_frame_pool[f->device].push_back(f);
            return;
        }
    }
    (void)hipSetDevice(f->device);
    f->worker.stop();
    if (f->ev) (void)hipEventDestroy(f->ev);
    if (f->stereo_block) (void)hipFree(f->stereo_block);
    if (f->block) (void)hipFree(f->block);
    delete f;
}

int orbgpu_frame_build_mono(orbgpu_frame *f, orbgpu_extractor *e, const uint8_t *img, int32_t w, int32_t h, int32_t stride,
                            const float *K4, const float *dist5, const float *bounds4, int32_t *n_out)
{
    int rc = frame_check("orbgpu_frame_build_mono", f, K4, dist5, bounds4, n_out);
    if (rc) return rc;
    if ((rc = frame_begin(f, e, bounds4))) return rc;
    HostExtraction X;
    const uint8_t *one[1] = {img};
    if ((rc = extract_host_begin(e, one, 1, w, h, stride, &X))) return rc;
    if ((rc = frame_finish_enqueue(f, X, K4, dist5, 0, nullptr, w, h, 0.0f))) return fail_after_enqueue(rc, &X, nullptr);
    if ((rc = frame_grid_enqueue(f, X.stream))) return fail_after_enqueue(rc, &X, nullptr);
    return frame_complete(f, X, nullptr, false, n_out);
}

int orbgpu_frame_build_rgbd(orbgpu_frame *f, orbgpu_extractor *e, const uint8_t *img, int32_t w, int32_t h, int32_t stride,
                            const float *depth, int32_t depth_stride_floats, float bf, const float *K4, const float *dist5,
                            const float *bounds4, int32_t *n_out)
{
    int rc = frame_check("orbgpu_frame_build_rgbd", f, K4, dist5, bounds4, n_out);
    if (rc) return rc;
    if (!depth || depth_stride_floats < w) {
        set_error("orbgpu_frame_build_rgbd: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if ((rc = frame_begin(f, e, bounds4))) return rc;
    if ((rc = frame_stereo_block(f, (size_t)w * h * 4))) return rc;
    HostExtraction X;
    const uint8_t *one[1] = {img};
    if ((rc = extract_host_begin(e, one, 1, w, h, stride, &X))) return rc;
    if (hipMemcpy2DAsync(f->stereo_block, (size_t)w * 4, depth, (size_t)depth_stride_floats * 4, (size_t)w * 4, (size_t)h,
                         hipMemcpyHostToDevice, X.stream) != hipSuccess) {
        set_error("orbgpu_frame_build_rgbd: the upload of the depth image failed");
        return fail_after_enqueue(ORBGPU_E_HIP, &X, nullptr);
    }
    if ((rc = frame_finish_enqueue(f, X, K4, dist5, 1, (const float *)f->stereo_block, w, h, bf))) return fail_after_enqueue(rc, &X, nullptr);
    if ((rc = frame_grid_enqueue(f, X.stream))) return fail_after_enqueue(rc, &X, nullptr);
    return frame_complete(f, X, nullptr, true, n_out);
}

int orbgpu_frame_build_stereo(orbgpu_frame *f, orbgpu_extractor *left, orbgpu_extractor *right, const uint8_t *imgL,
                              const uint8_t *imgR, int32_t w, int32_t h, int32_t stride, float bf, float b, const float *K4,
                              const float *dist5, const float *bounds4, int32_t *n_out, int32_t *nvalid_out)
{
    int rc = frame_check("orbgpu_frame_build_stereo", f, K4, dist5, bounds4, n_out);
    if (rc) return rc;
    if (!left || !right || left == right || !(b > 0)) {
        set_error("orbgpu_frame_build_stereo: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if ((rc = frame_begin(f, left, bounds4))) return rc;
    StereoArgs S;
    memset(&S, 0, sizeof(S));
    int32_t nlevR = 0;
    float scaleR[ORBGPU_MAX_LEVELS];
    if ((rc = orbgpu_extractor_tables(left, &S.nlevels, S.scale, S.inv_scale, nullptr, nullptr, nullptr))) return rc;
    if ((rc = orbgpu_extractor_tables(right, &nlevR, scaleR, nullptr, nullptr, nullptr, nullptr))) return rc;
    if (nlevR != S.nlevels || memcmp(scaleR, S.scale, sizeof(float) * (size_t)nlevR) != 0) {
        set_error("orbgpu_frame_build_stereo: the two extractors have different pyramids");
        return ORBGPU_E_INVALID;
    }
    if (h > 8192) {
        set_error("orbgpu_frame_build_stereo: images of more than 8192 rows are not supported");
        return ORBGPU_E_UNSUPPORTED;
    }
    // the two extractions run side by side on the two handles' streams (two threads in the reference, :78-81)
    HostExtraction XL, XR;
    XL.stream = XR.stream = nullptr;
    const uint8_t *oneL[1] = {imgL}, *oneR[1] = {imgR};
    // the right image is staged and enqueued by the handle's helper thread while this one does the left image
    int rcR = ORBGPU_OK;
    std::string errR;
    f->worker.start();
    f->worker.submit([&] {
        rcR = extract_host_begin(right, oneR, 1, w, h, stride, &XR);
        if (rcR) errR = orbgpu_last_error();
    });
    rc = extract_host_begin(left, oneL, 1, w, h, stride, &XL);
    f->worker.wait();
    if (rc) return fail_after_enqueue(rc, nullptr, rcR ? nullptr : &XR);  // the right image may be in flight
    if (rcR) {
        set_error("right image: %s", errR.c_str());
        return fail_after_enqueue(rcR, &XL, nullptr);
    }
    rc = frame_stereo_enqueue(f, left, right, XL, XR, S, w, h, bf, b, K4, dist5);
    if (rc) return fail_after_enqueue(rc, &XL, &XR);
    if ((rc = frame_complete(f, XL, &XR, true, n_out))) return rc;
    if (nvalid_out) *nvalid_out = f->nvalid;
    return ORBGPU_OK;
}

}  // extern "C"

// The part of the stereo build behind the two extractions: join the right extractor's stream, row table, matching,
// outlier cut, grid -- all enqueued on the left extractor's stream.
static int frame_stereo_enqueue(orbgpu_frame *f, orbgpu_extractor *left, orbgpu_extractor *right, const HostExtraction &XL,
                                const HostExtraction &XR, StereoArgs &S, int32_t w, int32_t h, float bf, float b, const float *K4,
                                const float *dist5)
{
    int rc;
    hipStream_t st = XL.stream;
    if (XR.stream != st) {
        ORBGPU_HIP_CHECK(hipEventRecord(f->ev, XR.stream));
        ORBGPU_HIP_CHECK(hipStreamWaitEvent(st, f->ev, 0));
    }
    for (int l = 0; l < S.nlevels; l++) {
        int32_t w2, h2, p2;
        if ((rc = orbgpu_pyramid_level_device(left, 0, l, 0, &S.pyrL[l], &S.pitchL[l], &S.w[l], &S.h[l]))) return rc;
        if ((rc = orbgpu_pyramid_level_device(right, 0, l, 0, &S.pyrR[l], &p2, &w2, &h2))) return rc;
        S.pitchR[l] = p2;
    }
    // stereo working set: best SAD [capacity] | row spans [right capacity] | row_start [rows + 1] | row_items; a right keypoint spans at most
    // 2 r + 3 rows, r = 2 * its level's scale
    const int nRows = h;
    const size_t span = (size_t)ceilf(4.0f * S.scale[S.nlevels - 1]) + 3;
    const size_t items = (size_t)XR.cap * (span < (size_t)nRows ? span : (size_t)nRows);
    if ((rc = frame_stereo_block(f, ((size_t)f->capacity + (size_t)XR.cap + (size_t)nRows + 1 + items) * 4))) return rc;
    int32_t *best_sad = (int32_t *)f->stereo_block, *spans = best_sad + f->capacity, *row_start = spans + XR.cap,
            *row_items = row_start + nRows + 1;
    if ((rc = frame_finish_enqueue(f, XL, K4, dist5, 2, nullptr, w, h, bf))) return rc;
    ScaleTab sc;
    for (int l = 0; l < ORBGPU_MAX_LEVELS; l++) sc.s[l] = S.scale[l < S.nlevels ? l : S.nlevels - 1];
    if ((rc = launch_stereo_rows(st, XR.d_kps, XR.d_counts, XR.cap, sc, S.nlevels, nRows, spans, row_start, row_items))) return rc;
    S.kpsL = f->kps; S.descL = f->desc; S.nL = f->capacity; S.d_nL = f->count;
    S.kpsR = XR.d_kps; S.descR = XR.d_desc; S.nR = XR.cap;
    S.row_start = row_start; S.row_items = row_items; S.nRows = nRows;
    S.bf = bf; S.b = b;
    S.uRight = f->uright; S.depth = f->depth; S.best_sad = best_sad;
    if ((rc = launch_stereo(st, S))) return rc;
    if ((rc = launch_stereo_outliers(st, best_sad, f->count, f->capacity, f->uright, f->depth, f->count + 1))) return rc;
    return frame_grid_enqueue(f, st);
}

extern "C" {

// A frame from host arrays (a Frame the program computed itself, a KeyFrame read back from a saved map): the
// arrays are uploaded and the grid is built as AssignFeaturesToGrid does.
int orbgpu_frame_upload(orbgpu_frame *f, const orbgpu_keypoint *kps, const orbgpu_keypoint *kps_un, const uint8_t *desc,
                        const float *uright, const float *depth, int32_t n, int32_t nlevels, const float *bounds4)
{
    if (!f || n < 0 || (n > 0 && (!kps_un || !desc)) || !bounds4 || !(bounds4[2] > bounds4[0]) || !(bounds4[3] > bounds4[1]) ||
        nlevels < 1 || nlevels > ORBGPU_MAX_LEVELS) {
        set_error("orbgpu_frame_upload: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (n > f->capacity) {
        set_error("orbgpu_frame_upload: the frame holds %d keypoints, %d given", f->capacity, n);
        return ORBGPU_E_CAPACITY;
    }
    for (int i = 0; i < n; i++)
        if (kps_un[i].octave < 0 || kps_un[i].octave >= nlevels) {
            set_error("orbgpu_frame_upload: keypoint %d has octave %d of %d levels", i, kps_un[i].octave, nlevels);
            return ORBGPU_E_INVALID;
        }
    ORBGPU_HIP_CHECK(hipSetDevice(f->device));
    f->built = false;
    f->nlevels = nlevels;
    memcpy(f->bounds4, bounds4, sizeof(f->bounds4));
    int32_t hc[4] = {n, 0, 0, 0};
    if (n > 0) {
        ORBGPU_HIP_CHECK(hipMemcpy(f->kps_un, kps_un, (size_t)n * sizeof(orbgpu_keypoint), hipMemcpyHostToDevice));
        ORBGPU_HIP_CHECK(hipMemcpy(f->kps, kps ? kps : kps_un, (size_t)n * sizeof(orbgpu_keypoint), hipMemcpyHostToDevice));
        ORBGPU_HIP_CHECK(hipMemcpy(f->desc, desc, (size_t)n * 32, hipMemcpyHostToDevice));
        std::vector<float> minus((uright && depth) ? 0 : (size_t)n, -1.0f);
        ORBGPU_HIP_CHECK(hipMemcpy(f->uright, uright ? uright : minus.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        ORBGPU_HIP_CHECK(hipMemcpy(f->depth, depth ? depth : minus.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        if (depth)
            for (int i = 0; i < n; i++) hc[1] += depth[i] > 0 ? 1 : 0;
    }
    ORBGPU_HIP_CHECK(hipMemcpy(f->count, hc, 16, hipMemcpyHostToDevice));
    int rc = frame_grid_enqueue(f, nullptr);
    if (rc) return rc;
    ORBGPU_HIP_CHECK(hipStreamSynchronize(nullptr));
    const orbgpu_keypoint *hk = kps ? kps : kps_un;
    f->h_kps.assign(hk, hk + n);
    f->h_desc.assign(desc, desc + (size_t)n * 32);
    f->h_valid = true;
    f->distorted = kps != nullptr && kps != kps_un;  // the two arrays may differ
    f->n = n;
    f->nvalid = hc[1];
    f->built = true;
    return ORBGPU_OK;
}

int orbgpu_frame_size(const orbgpu_frame *f, int32_t *n, int32_t *nvalid, int32_t *capacity)
{
    if (!f) return ORBGPU_E_INVALID;
    if (n) *n = f->built ? f->n : 0;
    if (nvalid) *nvalid = f->built ? f->nvalid : 0;
    if (capacity) *capacity = f->capacity;
    return ORBGPU_OK;
}

int orbgpu_frame_download(orbgpu_frame *f, orbgpu_keypoint *kps, orbgpu_keypoint *kps_un, uint8_t *desc, float *uright, float *depth)
{
    if (!f || !f->built) {
        set_error("orbgpu_frame_download: the frame has not been built");
        return ORBGPU_E_INVALID;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(f->device));
    const size_t n = (size_t)f->n;
    if (n == 0) return ORBGPU_OK;
    // keypoints and descriptors came to the host with the extraction: no GPU round trip for them
    if (kps) {
        if (f->h_valid) memcpy(kps, f->h_kps.data(), n * sizeof(orbgpu_keypoint));
        else ORBGPU_HIP_CHECK(hipMemcpy(kps, f->kps, n * sizeof(orbgpu_keypoint), hipMemcpyDeviceToHost));
    }
    if (kps_un) {
        if (f->h_valid && !f->distorted) memcpy(kps_un, f->h_kps.data(), n * sizeof(orbgpu_keypoint));
        else ORBGPU_HIP_CHECK(hipMemcpy(kps_un, f->kps_un, n * sizeof(orbgpu_keypoint), hipMemcpyDeviceToHost));
    }
    if (desc) {
        if (f->h_valid) memcpy(desc, f->h_desc.data(), n * 32);
        else ORBGPU_HIP_CHECK(hipMemcpy(desc, f->desc, n * 32, hipMemcpyDeviceToHost));
    }
    if (uright) ORBGPU_HIP_CHECK(hipMemcpy(uright, f->uright, n * 4, hipMemcpyDeviceToHost));
    if (depth) ORBGPU_HIP_CHECK(hipMemcpy(depth, f->depth, n * 4, hipMemcpyDeviceToHost));
    return ORBGPU_OK;
}

int orbgpu_frame_grid(orbgpu_frame *f, int32_t *cell_start, int32_t *cell_index, int32_t *n_in_grid)
{
    if (!f || !cell_start) {
        set_error("orbgpu_frame_grid: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (!f->built) {
        set_error("orbgpu_frame_grid: the frame has not been built");
        return ORBGPU_E_INVALID;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(f->device));
    const int ncell = ORBGPU_GRID_COLS * ORBGPU_GRID_ROWS;
    const int32_t *g = f->grid();
    ORBGPU_HIP_CHECK(hipMemcpy(cell_start, g, (size_t)(ncell + 1) * 4, hipMemcpyDeviceToHost));
    const int ng = cell_start[ncell];
    if (ng > 0 && cell_index)  // the keypoint index is the fourth word of every 16-byte grid record
        ORBGPU_HIP_CHECK(hipMemcpy2D(cell_index, 4, g + 3080 + 3, 16, 4, (size_t)ng, hipMemcpyDeviceToHost));
    if (n_in_grid) *n_in_grid = ng;
    return ORBGPU_OK;
}

int orbgpu_frame_device(const orbgpu_frame *f, const orbgpu_keypoint **d_kps, const orbgpu_keypoint **d_kps_un, const uint8_t **d_desc,
                        const float **d_uright, const float **d_depth, const int32_t **d_count)
{
    if (!f) return ORBGPU_E_INVALID;
    if (d_kps) *d_kps = f->kps;
    if (d_kps_un) *d_kps_un = f->kps_un;
    if (d_desc) *d_desc = f->desc;
    if (d_uright) *d_uright = f->uright;
    if (d_depth) *d_depth = f->depth;
    if (d_count) *d_count = f->count;
    return ORBGPU_OK;
}

}  // extern "C"
