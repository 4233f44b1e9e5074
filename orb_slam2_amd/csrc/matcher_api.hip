// C-ABI entry points of the matcher half of liborbgpu (see include/orbgpu.h).
// Reference lines are relative to /root/reference/src/ORB_SLAM2/.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "frame_internal.h"
#include "match_internal.h"
#include "orbgpu_internal.h"

namespace orbgpu {
int launch_bruteforce(hipStream_t st, const uint8_t *d_Q, int nq, const uint8_t *d_T, int nt, int nsplit, int tchunk,
                      uint2 *part, int32_t *bi, int32_t *b1, int32_t *b2);
int launch_match_window(hipStream_t st, const MatchArgs &A, int npairs);
int launch_grid_only(hipStream_t st, const MatchArgs &A);
int launch_shift_queries(hipStream_t st, int npairs, const orbgpu_keypoint *kps, const int32_t *n, int capacity, float du,
                         float dv, float th, const ScaleTab &sc, const float *b4, orbgpu_wquery *q);
int launch_stereo(hipStream_t st, const StereoArgs &A);
int launch_stereo_rows(hipStream_t st, const orbgpu_keypoint *kpsR, const int32_t *d_nR, int capR, const ScaleTab &scale, int nlevels, int nRows,
                       int32_t *spans, int32_t *row_start, int32_t *row_items);
int launch_stereo_outliers(hipStream_t st, const int32_t *best_sad, const int32_t *d_nL, int capL, float *uRight, float *depth,
                           int32_t *nvalid);
int launch_stereo_batch(hipStream_t st, const StereoArgs &A, int npairs);
int launch_stereo_rows_batch(hipStream_t st, int npairs, const orbgpu_keypoint *kpsR, const int32_t *d_nR, int capR, const ScaleTab &scale,
                             int nlevels, int nRows, int32_t *spans, int32_t *row_start, int32_t *row_items, int items_stride);
int launch_stereo_outliers_batch(hipStream_t st, int npairs, const int32_t *best_sad, const int32_t *d_nL, int nL_stride, int capL,
                                 float *uRight, float *depth, int32_t *nvalid, int nvalid_stride);
int launch_init_queries(hipStream_t st, const orbgpu_keypoint *kps1, int n1, const float *prev_xy, int window, orbgpu_wquery *q);
int launch_init_update(hipStream_t st, const int32_t *matches12, int n1, const orbgpu_keypoint *kps2, float *prev_xy);
int launch_init_queries_batch(hipStream_t st, int npairs, const orbgpu_keypoint *kps1, const int32_t *n1, int capacity, int window,
                              orbgpu_wquery *q);
int launch_distinctive(hipStream_t st, const uint8_t *d_desc, const int32_t *d_start, int npoints, int32_t *d_best_idx,
                       int32_t *d_best_median);
int launch_undistort(hipStream_t st, const orbgpu_keypoint *kps, int n, const float *K4, const float *D5, orbgpu_keypoint *out);
int launch_stereo_from_rgbd(hipStream_t st, const orbgpu_keypoint *kps, const orbgpu_keypoint *kps_un, int n, const float *depth, int w,
                            int h, int stride_floats, float bf, float *uright, float *out_depth);

struct MBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return ORBGPU_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        // grow geometrically so repeated calls with slowly growing sizes do not reallocate each time
        size_t want = need + need / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            set_error("hipMalloc of %zu bytes failed", want);
            return ORBGPU_E_NOMEM;
        }
        bytes = want;
        return ORBGPU_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};
}  // namespace orbgpu

using namespace orbgpu;

// Partial results of the allocation-free brute-force entry point: one fixed array in every
// GPU's memory for the life of the process.  Host code never uses the array's name as an
// address: each device's copy is looked up once with hipGetSymbolAddress while that device is
// current (resolve_noalloc_scratch, called from orbgpu_matcher_create) and the pointer kept per
// device.
#define ORBGPU_NOALLOC_SCRATCH_BYTES (16u << 20)
#define ORBGPU_NOALLOC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[ORBGPU_NOALLOC_SCRATCH_BYTES];
static void *g_noalloc_scratch[ORBGPU_NOALLOC_MAX_DEVICES];
static std::mutex g_noalloc_mutex;

// With `device` current: the address of this device's copy of `scratch`.
static int resolve_noalloc_scratch(int device, void **out)
{
    if (device < 0 || device >= ORBGPU_NOALLOC_MAX_DEVICES) {
        set_error("device %d is beyond the %d devices the allocation-free entry point serves", device, ORBGPU_NOALLOC_MAX_DEVICES);
        return ORBGPU_E_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (!g_noalloc_scratch[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            set_error("hipGetSymbolAddress(scratch) on device %d failed: %s", device, hipGetErrorString(e));
            return ORBGPU_E_HIP;
        }
        g_noalloc_scratch[device] = ptr;
    }
    if (out) *out = g_noalloc_scratch[device];
    return ORBGPU_OK;
}

struct orbgpu_matcher {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    MBuf part;     // brute-force partial results
    MBuf bfws;     // matrix-core brute force: byte-per-bit images of both descriptor sets and the chunks' partial results
    int bf_kernel = 0;  // 0: chosen by size, 1: vector-unit kernel, 2: matrix-core kernel (orbgpu_matcher_set_bruteforce_kernel)
    MBuf scratch;  // window matcher scratch
    MBuf stereo;   // batched ComputeStereoMatches: best SADs, row spans, row tables
    MBuf initws;   // batched SearchForInitialization: queries and the per-train-keypoint assignment
    MBuf stats;    // resolver statistics of the last window call
    int stats_pairs = 0;
    MBuf stage[12];  // staging for the host entry points
    // the windowed / segmented searches move their host operands in one block each way: a device block and its
    // page-locked twin, laid out per call (window_host)
    MBuf io;
    void *h_io = nullptr;
    size_t h_io_bytes = 0;
    bool geom_set = false;  // one-shot grid geometry of the next windowed call (orbgpu_matcher_set_grid_geometry)
    float geom[6] = {0, 0, 0, 0, 0, 0};
};

// Matcher contexts (stream + scratch + staging) are recycled: the reference builds an ORBmatcher on the stack at
// every call site (src/Tracking.cc:738,951,1060 ...), so create / destroy must not cost a stream and a dozen
// hipMallocs each time.  orbgpu_matcher_destroy parks an idle context here, orbgpu_matcher_create takes one back.
static std::mutex g_pool_mutex;
static std::vector<orbgpu_matcher *> g_pool[ORBGPU_NOALLOC_MAX_DEVICES];
static const size_t kPoolPerDevice = 8;

// The grid geometry of one windowed call: from the bounds, or the caller's one-shot override.
static void grid_geometry(orbgpu_matcher *m, const float *b4, MatchArgs *A)
{
    if (m->geom_set) {
        A->gminX = m->geom[0]; A->gminY = m->geom[1]; A->qminX = m->geom[2]; A->qminY = m->geom[3];
        A->invW = m->geom[4]; A->invH = m->geom[5];
        m->geom_set = false;
    } else {  // src/Frame.cc:101-102
        A->gminX = A->qminX = b4[0]; A->gminY = A->qminY = b4[1];
        A->invW = (float)ORBGPU_GRID_COLS / (b4[2] - b4[0]);
        A->invH = (float)ORBGPU_GRID_ROWS / (b4[3] - b4[1]);
    }
}

static const int64_t kBfMfmaMinPairs = 4 << 20;

static void choose_split(int nq, int nt, int *nsplit, int *tchunk)
{
    // Aim at >= ~4096 workgroups so 256 CUs stay balanced; chunks of at most
    // 2^20 train descriptors (the local index has 20 bits in the packed key).
    const int qb = (nq + 255) / 256;
    int ns = (4096 + qb - 1) / qb;
    if (ns < 1) ns = 1;
    int maxns = (nt + 255) / 256;  // keep chunks >= 256 descriptors
    if (maxns < 1) maxns = 1;
    if (ns > maxns) ns = maxns;
    int ch = (nt + ns - 1) / ns;
    if (ch < 1) ch = 1;
    if (ch > (1 << 20)) ch = 1 << 20;
    ns = nt > 0 ? (nt + ch - 1) / ch : 1;
    *nsplit = ns;
    *tchunk = ch;
}

extern "C" {

int orbgpu_matcher_create(int32_t device, orbgpu_matcher **out)
{
    if (!out) return ORBGPU_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available: the ORB matcher needs a GPU");
        return ORBGPU_E_HIP;
    }
    if (device < 0 || device >= ndev) {
        set_error("device %d out of range (%d devices)", device, ndev);
        return ORBGPU_E_INVALID;
    }
    if (device < ORBGPU_NOALLOC_MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        if (!g_pool[device].empty()) {
            *out = g_pool[device].back();
            g_pool[device].pop_back();
            return ORBGPU_OK;
        }
    }
    orbgpu_matcher *m = new (std::nothrow) orbgpu_matcher();
    if (!m) return ORBGPU_E_NOMEM;
    m->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("stream creation failed");
        delete m;
        return ORBGPU_E_HIP;
    }
    m->stream = m->own_stream;
    int rc = resolve_noalloc_scratch(device, nullptr);  // device is current here
    if (rc != ORBGPU_OK && rc != ORBGPU_E_UNSUPPORTED) {
        orbgpu_matcher_destroy(m);
        return rc;
    }
    *out = m;
    return ORBGPU_OK;
}

void orbgpu_matcher_destroy(orbgpu_matcher *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->own_stream && m->device < ORBGPU_NOALLOC_MAX_DEVICES) {
        m->stream = m->own_stream;  // a caller's stream does not outlive its use of the handle
        m->geom_set = false;
        m->stats_pairs = 0;
        m->bf_kernel = 0;
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        if (g_pool[m->device].size() < kPoolPerDevice) {
            g_pool[m->device].push_back(m);
            return;
        }
    }
    m->part.release();
    m->bfws.release();
    m->scratch.release();
    m->stereo.release();
    m->initws.release();
    m->stats.release();
    for (auto &b : m->stage) b.release();
    m->io.release();
    if (m->h_io) (void)hipHostFree(m->h_io);
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    delete m;
}

int orbgpu_matcher_set_stream(orbgpu_matcher *m, void *hip_stream)
{
    if (!m) return ORBGPU_E_INVALID;
    m->stream = hip_stream ? (hipStream_t)hip_stream : m->own_stream;
    return ORBGPU_OK;
}

int orbgpu_matcher_set_grid_geometry(orbgpu_matcher *m, const float *geom6)
{
    if (!m) return ORBGPU_E_INVALID;
    m->geom_set = false;
    if (!geom6) return ORBGPU_OK;
    if (!(geom6[4] > 0.0f) || !(geom6[5] > 0.0f)) {
        set_error("orbgpu_matcher_set_grid_geometry: the inverse cell sizes must be positive");
        return ORBGPU_E_INVALID;
    }
    memcpy(m->geom, geom6, sizeof(m->geom));
    m->geom_set = true;
    return ORBGPU_OK;
}

int orbgpu_matcher_sync(orbgpu_matcher *m)
{
    if (!m) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipStreamSynchronize(m->stream));
    return ORBGPU_OK;
}

// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1710-1727).
int orbgpu_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int d = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

int orbgpu_hamming_bruteforce_device(orbgpu_matcher *m, const uint8_t *d_Q, int32_t nq, const uint8_t *d_T, int32_t nt,
                                     int32_t *d_best_idx, int32_t *d_best_d, int32_t *d_second_d)
{
    TraceRange trace_range("orbgpu.hamming_bruteforce");
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!d_Q || !d_best_idx || !d_best_d || !d_second_d)) || (nt > 0 && !d_T)) {
        set_error("orbgpu_hamming_bruteforce: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (nq == 0) return ORBGPU_OK;
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    // the matrix-core kernel pays for spreading both sets to a byte per bit: worth it from a few million pairs on
    const bool mfma = nt > 0 && (m->bf_kernel == 2 || (m->bf_kernel == 0 && (int64_t)nq * nt >= kBfMfmaMinPairs));
    if (mfma) {
        const BfMfmaPlan P = bf_mfma_plan(nq, nt);
        int rc = m->bfws.ensure(P.bytes);
        if (rc) return rc;
        return launch_bruteforce_mfma(m->stream, P, d_Q, nq, d_T, nt, (uint8_t *)m->bfws.p, d_best_idx, d_best_d, d_second_d);
    }
    int nsplit, tchunk;
    choose_split(nq, nt, &nsplit, &tchunk);
    int rc = m->part.ensure((size_t)nsplit * nq * sizeof(uint2));
    if (rc) return rc;
    return launch_bruteforce(m->stream, d_Q, nq, d_T, nt, nsplit, tchunk, (uint2 *)m->part.p, d_best_idx, d_best_d, d_second_d);
}

int orbgpu_matcher_set_bruteforce_kernel(orbgpu_matcher *m, int32_t which)
{
    if (!m || which < 0 || which > 2) {
        set_error("orbgpu_matcher_set_bruteforce_kernel: which must be 0 (by size), 1 (vector unit) or 2 (matrix cores)");
        return ORBGPU_E_INVALID;
    }
    m->bf_kernel = which;
    return ORBGPU_OK;
}

// Same scan without any allocation: the per-chunk partial results go to this device's fixed
// `scratch` array instead of a buffer the handle grows on demand.
int orbgpu_hamming_bruteforce_device_noalloc(orbgpu_matcher *m, const uint8_t *d_Q, int32_t nq, const uint8_t *d_T, int32_t nt,
                                             int32_t *d_best_idx, int32_t *d_best_d, int32_t *d_second_d)
{
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!d_Q || !d_best_idx || !d_best_d || !d_second_d)) || (nt > 0 && !d_T)) {
        set_error("orbgpu_hamming_bruteforce_device_noalloc: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (nq == 0) return ORBGPU_OK;
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    void *ptr = nullptr;
    int rc = resolve_noalloc_scratch(m->device, &ptr);  // a table lookup after orbgpu_matcher_create
    if (rc) return rc;
    const size_t max_entries = ORBGPU_NOALLOC_SCRATCH_BYTES / sizeof(uint2);
    int nsplit, tchunk;
    choose_split(nq, nt, &nsplit, &tchunk);
    // choose_split keeps nsplit * nq below 2^21 unless the query set alone (nq > 2^21) or the 2^20-descriptor cap
    // on a chunk forces more partial results than the array holds
    if ((size_t)nsplit * (size_t)nq > max_entries) {
        set_error("orbgpu_hamming_bruteforce_device_noalloc: %d queries x %d train descriptors need more than the fixed %u-byte "
                  "scratch (nq * ceil(nt / 2^20) <= %zu)", nq, nt, ORBGPU_NOALLOC_SCRATCH_BYTES, max_entries);
        return ORBGPU_E_CAPACITY;
    }
    return launch_bruteforce(m->stream, d_Q, nq, d_T, nt, nsplit, tchunk, (uint2 *)ptr, d_best_idx, d_best_d, d_second_d);
}

int orbgpu_hamming_bruteforce(orbgpu_matcher *m, const uint8_t *Q, int32_t nq, const uint8_t *T, int32_t nt,
                              int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    if (!m || nq < 0 || nt < 0 || (nq > 0 && (!Q || !best_idx || !best_d || !second_d)) || (nt > 0 && !T)) {
        set_error("orbgpu_hamming_bruteforce: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (nq == 0) return ORBGPU_OK;
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    int rc;
    if ((rc = m->stage[0].ensure((size_t)nq * 32))) return rc;
    if ((rc = m->stage[1].ensure((size_t)(nt > 0 ? nt : 1) * 32))) return rc;
    if ((rc = m->stage[2].ensure((size_t)nq * 12))) return rc;
    hipStream_t st = m->stream;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(m->stage[0].p, Q, (size_t)nq * 32, hipMemcpyHostToDevice, st));
    if (nt > 0) ORBGPU_HIP_CHECK(hipMemcpyAsync(m->stage[1].p, T, (size_t)nt * 32, hipMemcpyHostToDevice, st));
    int32_t *o = (int32_t *)m->stage[2].p;
    rc = orbgpu_hamming_bruteforce_device(m, (const uint8_t *)m->stage[0].p, nq, (const uint8_t *)m->stage[1].p, nt, o, o + nq,
                                          o + 2 * (size_t)nq);
    if (rc) return rc;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(best_idx, o, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipMemcpyAsync(best_d, o + nq, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipMemcpyAsync(second_d, o + 2 * (size_t)nq, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    return ORBGPU_OK;
}

static int run_window(orbgpu_matcher *m, int32_t mode, int32_t npairs, const orbgpu_wquery *d_q, const uint8_t *d_qdesc,
                      const int32_t *d_nq, int32_t q_cap, const orbgpu_keypoint *d_kps2, const uint8_t *d_desc2,
                      const float *d_uright2, const int32_t *d_n2, int32_t t_cap, const float *b4, const uint8_t *d_occ,
                      int32_t *d_assign, int32_t *d_qmatch, int32_t th_dist, float nnratio, int32_t ori, int32_t *d_nmatches,
                      const int32_t *d_seg_start = nullptr, const int32_t *d_seg_index = nullptr, const MatchArgs *extra = nullptr,
                      const int32_t *d_grid = nullptr)
{
    if (npairs < 1 || q_cap < 1 || t_cap < 1 || t_cap > (1 << 20) || !(b4[2] > b4[0]) || !(b4[3] > b4[1])) {
        set_error("window matcher: invalid sizes or bounds");
        return ORBGPU_E_INVALID;
    }
    int rc = m->scratch.ensure((size_t)npairs * match_scratch_ints(q_cap, t_cap) * 4);
    if (rc) return rc;
    MatchArgs A;
    memset(&A, 0, sizeof(A));
    if (extra) A = *extra;  // mode-specific fields (epipolar geometry of mode 5)
    A.mode = mode;
    A.q = d_q; A.qdesc = d_qdesc; A.nq = d_nq; A.q_cap = q_cap;
    A.kps2 = d_kps2; A.desc2 = d_desc2; A.uright2 = d_uright2; A.n2 = d_n2; A.t_cap = t_cap;
    grid_geometry(m, b4, &A);
    A.occupied = d_occ; A.assign = d_assign; A.qmatch = d_qmatch; A.scratch = (int32_t *)m->scratch.p;
    A.th_dist = th_dist; A.nnratio = nnratio; A.check_orientation = ori; A.nmatches = d_nmatches;
    A.seg_start = d_seg_start; A.seg_index = d_seg_index;
    A.grid = npairs == 1 ? d_grid : nullptr;
    A.grid_levels = (mode == 3 && !A.grid) ? 1 : 0;  // SearchForInitialization looks at level 0 only
    if ((rc = m->stats.ensure((size_t)npairs * 16))) return rc;
    ORBGPU_HIP_CHECK(hipMemsetAsync(m->stats.p, 0, (size_t)npairs * 16, m->stream));
    A.stats = (int32_t *)m->stats.p;
    m->stats_pairs = npairs;
    return launch_match_window(m->stream, A, npairs);
}

int orbgpu_debug_window_stats(orbgpu_matcher *m, int32_t pair, int32_t *iterations, int32_t *rescans)
{
    if (!m || pair < 0 || pair >= m->stats_pairs) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(m->stream));
    int32_t v[4];
    ORBGPU_HIP_CHECK(hipMemcpy(v, (int32_t *)m->stats.p + 4 * pair, 16, hipMemcpyDeviceToHost));
    if (iterations) *iterations = v[0];
    if (rescans) *rescans = v[1];
    return ORBGPU_OK;
}

int orbgpu_search_by_projection_batch_device(orbgpu_matcher *m, int32_t mode, int32_t npairs, const orbgpu_wquery *d_q,
                                             const uint8_t *d_qdesc, const int32_t *d_nq, int32_t q_capacity,
                                             const orbgpu_keypoint *d_kps2, const uint8_t *d_desc2, const float *d_uright2,
                                             const int32_t *d_n2, int32_t t_capacity, const float *bounds4_host,
                                             const uint8_t *d_occupied, int32_t *d_assign, int32_t th_dist, float nnratio,
                                             int32_t check_orientation, int32_t *d_nmatches)
{
    TraceRange trace_range("orbgpu.search_by_projection");
    if (!m || mode < 0 || mode > 2 || !d_q || !d_qdesc || !d_nq || !d_kps2 || !d_desc2 || !d_n2 || !bounds4_host || !d_assign ||
        !d_nmatches) {
        set_error("orbgpu_search_by_projection_batch_device: invalid argument");
        return ORBGPU_E_INVALID;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    return run_window(m, mode, npairs, d_q, d_qdesc, d_nq, q_capacity, d_kps2, d_desc2, d_uright2, d_n2, t_capacity, bounds4_host,
                      d_occupied, d_assign, nullptr, th_dist, nnratio, check_orientation, d_nmatches);
}

// Operands of a windowed call that are on the device already.
struct WindowDev {
    const orbgpu_frame *train = nullptr;   // the train side is this frame (keypoints, descriptors, uRight, grid)
    const orbgpu_wquery *d_q = nullptr;    // the queries and their descriptors (both or neither)
    const uint8_t *d_qdesc = nullptr;
    // SearchForInitialization between two frames: the queries are made on the device from init_f1's undistorted
    // keypoints and prev_xy (vbPrevMatched, in / out: it travels inside the block, updated on the device)
    const orbgpu_frame *init_f1 = nullptr;
    float *prev_xy = nullptr;
    int window = 0;
};

// Front end shared by the windowed and segmented searches.  Host arrays travel in ONE block each way (a call with a
// thousand keypoints is a dozen small arrays: copied one by one, the copies cost more than the kernels): the inputs
// are packed into the handle's page-locked block and uploaded with one copy, the outputs (assign | qmatch | counts)
// come back with one.  Operands named in `dev` are used where they lie on the device.
static int window_host(orbgpu_matcher *m, int32_t mode, const orbgpu_wquery *q, const uint8_t *qdesc, int32_t nq,
                       const orbgpu_keypoint *kps2, const uint8_t *desc2, const float *uright2, int32_t n2, const float *b4,
                       const uint8_t *occupied, int32_t *assign, int32_t *qmatch, int32_t th_dist, float nnratio, int32_t ori,
                       int32_t *nmatches, const int32_t *seg_start = nullptr, const int32_t *seg_index = nullptr, int32_t nseg = 0,
                       const MatchArgs *extra = nullptr, const WindowDev *dev = nullptr)
{
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    const orbgpu_frame *train = dev ? dev->train : nullptr;
    const orbgpu_frame *f1 = dev ? dev->init_f1 : nullptr;
    const bool dev_q = dev && dev->d_q;
    // a frame's grid is laid out for the frame's capacity
    const size_t qc = nq > 0 ? nq : 1, tc = train ? train->capacity : (n2 > 0 ? n2 : 1);
    const int32_t nidx = seg_start ? seg_start[nseg] : 0;
    // block layout (offsets in bytes, every part 256-byte aligned): inputs first, outputs last
    size_t off = 0;
    auto part = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_cnt = part(16);
    const size_t o_q = part(dev_q ? 0 : qc * sizeof(orbgpu_wquery)), o_qd = part(dev_q || f1 ? 0 : qc * 32);
    const size_t o_k2 = part(train ? 0 : tc * sizeof(orbgpu_keypoint)), o_d2 = part(train ? 0 : tc * 32);
    const size_t o_ur = part(train || !uright2 ? 0 : tc * 4), o_occ = part(occupied ? tc : 0);
    const size_t o_ss = part(seg_start ? (size_t)(nseg + 1) * 4 : 0), o_si = part(seg_start ? (size_t)(nidx > 0 ? nidx : 1) * 4 : 0);
    const size_t o_prev = part(f1 ? qc * 8 : 0);  // last input, first output
    const size_t in_bytes = off;
    const size_t o_as = part(tc * 4), o_qm = part(qc * 4), o_out = part(16);
    const size_t total = off;
    int rc;
    if ((rc = m->io.ensure(total))) return rc;
    if (total > m->h_io_bytes) {
        if (m->h_io) (void)hipHostFree(m->h_io);
        m->h_io = nullptr;
        m->h_io_bytes = 0;
        const size_t want = total + total / 4 + 4096;
        if (hipHostMalloc(&m->h_io, want, hipHostMallocDefault) != hipSuccess) {
            set_error("hipHostMalloc of %zu bytes failed", want);
            return ORBGPU_E_NOMEM;
        }
        m->h_io_bytes = want;
    }
    uint8_t *h = (uint8_t *)m->h_io, *d = (uint8_t *)m->io.p;
    int32_t *hc = (int32_t *)(h + o_cnt);
    hc[0] = nq; hc[1] = n2; hc[2] = 0; hc[3] = 0;
    if (nq > 0 && f1) {
        memcpy(h + o_prev, dev->prev_xy, (size_t)nq * 8);
    } else if (nq > 0 && !dev_q) {
        memcpy(h + o_q, q, (size_t)nq * sizeof(orbgpu_wquery));
        memcpy(h + o_qd, qdesc, (size_t)nq * 32);
    }
    if (n2 > 0) {
        if (!train) {
            memcpy(h + o_k2, kps2, (size_t)n2 * sizeof(orbgpu_keypoint));
            memcpy(h + o_d2, desc2, (size_t)n2 * 32);
            if (uright2) memcpy(h + o_ur, uright2, (size_t)n2 * 4);
        }
        if (occupied) memcpy(h + o_occ, occupied, (size_t)n2);
    }
    if (seg_start) {
        memcpy(h + o_ss, seg_start, (size_t)(nseg + 1) * 4);
        if (nidx > 0) memcpy(h + o_si, seg_index, (size_t)nidx * 4);
    }
    ORBGPU_HIP_CHECK(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
    if (f1 && nq > 0 && (rc = launch_init_queries(st, f1->kps_un, nq, (const float *)(d + o_prev), dev->window, (orbgpu_wquery *)(d + o_q))))
        return rc;
    int32_t *dc = (int32_t *)(d + o_cnt);
    const orbgpu_keypoint *d_k2 = train ? train->kps_un : (const orbgpu_keypoint *)(d + o_k2);
    const uint8_t *d_d2 = train ? train->desc : d + o_d2;
    const float *d_ur = !uright2 ? nullptr : train ? train->uright : (const float *)(d + o_ur);
    rc = run_window(m, mode, 1, dev_q ? dev->d_q : (const orbgpu_wquery *)(d + o_q), dev_q ? dev->d_qdesc : f1 ? f1->desc : d + o_qd, dc,
                    (int32_t)qc,
                    d_k2, d_d2, d_ur, dc + 1, (int32_t)tc, b4, occupied ? d + o_occ : nullptr, (int32_t *)(d + o_as),
                    (int32_t *)(d + o_qm), th_dist, nnratio, ori, (int32_t *)(d + o_out),
                    seg_start ? (const int32_t *)(d + o_ss) : nullptr, seg_start ? (const int32_t *)(d + o_si) : nullptr, extra,
                    train ? train->grid() : nullptr);
    if (rc) return rc;
    // :566-568 on the device: vbPrevMatched[i] = position of the matched keypoint of frame 2
    if (f1 && nq > 0 && (rc = launch_init_update(st, (const int32_t *)(d + o_qm), nq, d_k2, (float *)(d + o_prev)))) return rc;
    const size_t o_back = f1 ? o_prev : o_as;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(h + o_back, d + o_back, total - o_back, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    if (f1 && nq > 0) memcpy(dev->prev_xy, h + o_prev, (size_t)nq * 8);
    if (assign && n2 > 0) memcpy(assign, h + o_as, (size_t)n2 * 4);
    if (qmatch && nq > 0) memcpy(qmatch, h + o_qm, (size_t)nq * 4);
    *nmatches = *(const int32_t *)(h + o_out);
    return ORBGPU_OK;
}

// A frame as the train side of a matcher call: built, and on the matcher's device.
static int check_train_frame(const char *who, const orbgpu_matcher *m, const orbgpu_frame *f)
{
    if (!f || !f->built) { set_error("%s: the frame has not been built", who); return ORBGPU_E_INVALID; }
    if (f->device != m->device) { set_error("%s: the frame lives on device %d, the matcher on %d", who, f->device, m->device); return ORBGPU_E_INVALID; }
    return ORBGPU_OK;
}

int orbgpu_search_by_projection(orbgpu_matcher *m, int32_t mode, const orbgpu_wquery *q, const uint8_t *qdesc, int32_t nq,
                                const orbgpu_keypoint *kps2, const uint8_t *desc2, const float *uright2, int32_t n2,
                                const float *bounds4, const uint8_t *occupied, int32_t *assign, int32_t th_dist, float nnratio,
                                int32_t check_orientation, int32_t *nmatches)
{
    TraceRange trace_range("orbgpu.search_by_projection");
    if (!m || mode < 0 || mode > 2 || nq < 0 || n2 < 0 || (nq > 0 && (!q || !qdesc)) || (n2 > 0 && (!kps2 || !desc2 || !assign)) ||
        !bounds4 || !nmatches) {
        set_error("orbgpu_search_by_projection: invalid argument");
        return ORBGPU_E_INVALID;
    }
    return window_host(m, mode, q, qdesc, nq, kps2, desc2, uright2, n2, bounds4, occupied, assign, nullptr, th_dist, nnratio,
                       check_orientation, nmatches);
}

// The kernels trust the segment tables of the BoW-guided matchers: check them on the host.
static int check_segments(const char *who, const orbgpu_wquery *q, int32_t nq, int32_t n2, const int32_t *seg_start,
                          const int32_t *seg_index, int32_t nseg)
{
    if (seg_start[0] != 0) { set_error("%s: seg_start[0] must be 0", who); return ORBGPU_E_INVALID; }
    for (int s = 0; s < nseg; s++)
        if (seg_start[s + 1] < seg_start[s]) { set_error("%s: seg_start is not monotone", who); return ORBGPU_E_INVALID; }
    for (int k = 0; k < seg_start[nseg]; k++)
        if (seg_index[k] < 0 || seg_index[k] >= n2) { set_error("%s: segment entry %d out of range", who, k); return ORBGPU_E_INVALID; }
    for (int i = 0; i < nq; i++)
        if ((q[i].flags & 1) && (q[i].minLevel < 0 || q[i].minLevel >= nseg)) {
            set_error("%s: query %d names segment %d of %d", who, i, q[i].minLevel, nseg);
            return ORBGPU_E_INVALID;
        }
    return ORBGPU_OK;
}

// ORBmatcher::SearchByBoW, both overloads (src/ORBmatcher.cc:159-288, 573-706): query i is compared
// with the train features of vocabulary node segment queries[i].minLevel only.
int orbgpu_search_by_bow(orbgpu_matcher *m, const orbgpu_wquery *q, const uint8_t *qdesc, int32_t nq, const orbgpu_keypoint *kps2,
                         const uint8_t *desc2, int32_t n2, const int32_t *seg_start, const int32_t *seg_index, int32_t nseg,
                         const uint8_t *excluded, int32_t th_dist, float nnratio, int32_t check_orientation, int32_t *assign,
                         int32_t *qmatch, int32_t *nmatches)
{
    if (!m || nq < 0 || n2 < 0 || nseg < 0 || (nq > 0 && (!q || !qdesc)) || (n2 > 0 && (!kps2 || !desc2)) || !seg_start ||
        (nseg > 0 && !seg_index) || !nmatches || (n2 > 0 && !assign)) {
        set_error("orbgpu_search_by_bow: invalid argument");
        return ORBGPU_E_INVALID;
    }
    int rc = check_segments("orbgpu_search_by_bow", q, nq, n2, seg_start, seg_index, nseg);
    if (rc) return rc;
    const float b4[4] = {0.f, 0.f, 1.f, 1.f};
    return window_host(m, 4, q, qdesc, nq, kps2, desc2, nullptr, n2, b4, excluded, assign, qmatch, th_dist, nnratio,
                       check_orientation, nmatches, seg_start, seg_index, nseg);
}

// ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:708-874).
int orbgpu_search_for_triangulation(orbgpu_matcher *m, const orbgpu_wquery *q, const uint8_t *qdesc, int32_t nq,
                                    const orbgpu_keypoint *kps2, const uint8_t *desc2, const float *uright2, int32_t n2,
                                    const int32_t *seg_start, const int32_t *seg_index, int32_t nseg, const uint8_t *excluded,
                                    const float *F12, float ex, float ey, const float *scale2, const float *sigma2,
                                    int32_t nlevels, int32_t only_stereo, int32_t check_orientation, int32_t *qmatch,
                                    int32_t *nmatches)
{
    if (!m || nq < 0 || n2 < 0 || nseg < 0 || (nq > 0 && (!q || !qdesc || !qmatch)) || (n2 > 0 && (!kps2 || !desc2)) || !seg_start ||
        (nseg > 0 && !seg_index) || !F12 || !scale2 || !sigma2 || nlevels < 1 || nlevels > ORBGPU_MAX_LEVELS || !nmatches) {
        set_error("orbgpu_search_for_triangulation: invalid argument");
        return ORBGPU_E_INVALID;
    }
    int rc = check_segments("orbgpu_search_for_triangulation", q, nq, n2, seg_start, seg_index, nseg);
    if (rc) return rc;
    for (int t = 0; t < n2; t++)
        if (kps2[t].octave < 0 || kps2[t].octave >= nlevels) {
            set_error("orbgpu_search_for_triangulation: keypoint %d has octave %d of %d levels", t, kps2[t].octave, nlevels);
            return ORBGPU_E_INVALID;
        }
    MatchArgs X;
    memset(&X, 0, sizeof(X));
    for (int k = 0; k < 9; k++) X.F12[k] = F12[k];
    X.ex = ex; X.ey = ey; X.only_stereo = only_stereo;
    for (int l = 0; l < nlevels; l++) { X.scale2.s[l] = scale2[l]; X.sigma2.s[l] = sigma2[l]; }
    const float b4[4] = {0.f, 0.f, 1.f, 1.f};
    return window_host(m, 5, q, qdesc, nq, kps2, desc2, uright2, n2, b4, excluded, nullptr, qmatch, ORBGPU_TH_LOW, 1.0f,
                       check_orientation, nmatches, seg_start, seg_index, nseg, &X);
}

// The window search shared by Fuse (src/ORBmatcher.cc:876-1010, 1012-1151) and SearchBySim3
// (:1153-1377): every query independently takes the nearest keypoint of its window.
int orbgpu_search_in_area(orbgpu_matcher *m, const orbgpu_wquery *q, const uint8_t *qdesc, int32_t nq, const orbgpu_keypoint *kps2,
                          const uint8_t *desc2, const float *uright2, int32_t n2, const float *bounds4, const float *inv_sigma2,
                          int32_t nlevels, int32_t chi2_gate, int32_t th_dist, int32_t *qmatch, int32_t *nmatches)
{
    if (!m || nq < 0 || n2 < 0 || (nq > 0 && (!q || !qdesc || !qmatch)) || (n2 > 0 && (!kps2 || !desc2)) || !bounds4 || !nmatches ||
        (chi2_gate && (!inv_sigma2 || nlevels < 1 || nlevels > ORBGPU_MAX_LEVELS))) {
        set_error("orbgpu_search_in_area: invalid argument");
        return ORBGPU_E_INVALID;
    }
    MatchArgs X;
    memset(&X, 0, sizeof(X));
    X.chi2_gate = chi2_gate ? 1 : 0;
    if (chi2_gate) {
        for (int t = 0; t < n2; t++)
            if (kps2[t].octave < 0 || kps2[t].octave >= nlevels) {
                set_error("orbgpu_search_in_area: keypoint %d has octave %d of %d levels", t, kps2[t].octave, nlevels);
                return ORBGPU_E_INVALID;
            }
        for (int l = 0; l < nlevels; l++) X.inv_sigma2.s[l] = inv_sigma2[l];
    }
    return window_host(m, 6, q, qdesc, nq, kps2, desc2, chi2_gate ? uright2 : nullptr, n2, bounds4, nullptr, nullptr, qmatch, th_dist,
                       1.0f, 0, nmatches, nullptr, nullptr, 0, &X);
}

// Frame::AssignFeaturesToGrid (src/Frame.cc:270-293) with Frame::PosInGrid (:485-500).
int orbgpu_assign_features_to_grid(orbgpu_matcher *m, const orbgpu_keypoint *kps, int32_t n, const float *bounds4, int32_t *cell_start,
                                   int32_t *cell_index, int32_t *n_in_grid)
{
    if (!m || n < 0 || (n > 0 && (!kps || !cell_index)) || !bounds4 || !cell_start || !(bounds4[2] > bounds4[0]) ||
        !(bounds4[3] > bounds4[1]) || n > (1 << 20)) {
        set_error("orbgpu_assign_features_to_grid: invalid argument");
        return ORBGPU_E_INVALID;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    const int tc = n > 0 ? n : 1;
    int rc;
    if ((rc = m->stage[2].ensure((size_t)tc * sizeof(orbgpu_keypoint)))) return rc;
    if ((rc = m->stage[3].ensure((size_t)tc * 32))) return rc;
    if ((rc = m->stage[8].ensure(64))) return rc;
    if ((rc = m->scratch.ensure(match_scratch_ints(1, tc) * 4))) return rc;
    if (n > 0) ORBGPU_HIP_CHECK(hipMemcpyAsync(m->stage[2].p, kps, (size_t)n * sizeof(orbgpu_keypoint), hipMemcpyHostToDevice, st));
    int32_t hc[4] = {0, n, 0, 0};
    ORBGPU_HIP_CHECK(hipMemcpyAsync(m->stage[8].p, hc, 16, hipMemcpyHostToDevice, st));
    MatchArgs A;
    memset(&A, 0, sizeof(A));
    A.mode = 1;
    A.nq = (const int32_t *)m->stage[8].p; A.q_cap = 1;
    A.kps2 = (const orbgpu_keypoint *)m->stage[2].p; A.desc2 = (const uint8_t *)m->stage[3].p;  // descriptors are carried along, unused here
    A.n2 = (const int32_t *)m->stage[8].p + 1; A.t_cap = tc;
    grid_geometry(m, bounds4, &A);
    A.scratch = (int32_t *)m->scratch.p;
    if ((rc = launch_grid_only(st, A))) return rc;
    // scratch layout of one pair (match_scratch_ints): candidates [kTopK*q_cap] | cell_start [3080] | grid records [4*t_cap] ...
    const int32_t *d_cs = A.scratch + kTopK * 1;
    const int32_t *d_rec = d_cs + 3080;
    const int ncell = ORBGPU_GRID_COLS * ORBGPU_GRID_ROWS;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(cell_start, d_cs, (size_t)(ncell + 1) * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    const int ng = cell_start[ncell];
    if (ng > 0)  // the keypoint index is the fourth word of every 16-byte grid record
        ORBGPU_HIP_CHECK(hipMemcpy2DAsync(cell_index, 4, d_rec + 3, 16, 4, (size_t)ng, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    if (n_in_grid) *n_in_grid = ng;
    return ORBGPU_OK;
}

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:262-340), batched over map points.
int orbgpu_distinctive_descriptors(orbgpu_matcher *m, const uint8_t *desc, const int32_t *start, int32_t npoints, int32_t *best_idx,
                                   int32_t *best_median)
{
    if (!m || npoints < 0 || !start || (npoints > 0 && !best_idx) || start[0] != 0) {
        set_error("orbgpu_distinctive_descriptors: invalid argument");
        return ORBGPU_E_INVALID;
    }
    for (int p = 0; p < npoints; p++)
        if (start[p + 1] < start[p] || start[p + 1] - start[p] >= (1 << 20)) {
            set_error("orbgpu_distinctive_descriptors: bad observation range of map point %d", p);
            return ORBGPU_E_INVALID;
        }
    if (npoints == 0) return ORBGPU_OK;
    const int32_t total = start[npoints];
    if (total > 0 && !desc) { set_error("orbgpu_distinctive_descriptors: invalid argument"); return ORBGPU_E_INVALID; }
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    int rc;
    if ((rc = m->stage[0].ensure((size_t)(total > 0 ? total : 1) * 32))) return rc;
    if ((rc = m->stage[1].ensure((size_t)(npoints + 1) * 4))) return rc;
    if ((rc = m->stage[2].ensure((size_t)npoints * 8))) return rc;
    if (total > 0) ORBGPU_HIP_CHECK(hipMemcpyAsync(m->stage[0].p, desc, (size_t)total * 32, hipMemcpyHostToDevice, st));
    ORBGPU_HIP_CHECK(hipMemcpyAsync(m->stage[1].p, start, (size_t)(npoints + 1) * 4, hipMemcpyHostToDevice, st));
    int32_t *o = (int32_t *)m->stage[2].p;
    if ((rc = launch_distinctive(st, (const uint8_t *)m->stage[0].p, (const int32_t *)m->stage[1].p, npoints, o, o + npoints))) return rc;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(best_idx, o, (size_t)npoints * 4, hipMemcpyDeviceToHost, st));
    if (best_median) ORBGPU_HIP_CHECK(hipMemcpyAsync(best_median, o + npoints, (size_t)npoints * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    return ORBGPU_OK;
}

// Frame::ComputeStereoFromRGBD (src/Frame.cc:777-798).
int orbgpu_stereo_from_rgbd(orbgpu_matcher *m, const orbgpu_keypoint *kps, const orbgpu_keypoint *kps_un, int32_t n, const float *depth,
                            int32_t w, int32_t h, int32_t stride_floats, float bf, float *uright, float *out_depth)
{
    if (!m || n < 0 || (n > 0 && (!kps || !kps_un || !uright || !out_depth)) || !depth || w < 1 || h < 1 || stride_floats < w) {
        set_error("orbgpu_stereo_from_rgbd: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (n == 0) return ORBGPU_OK;
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    int rc;
    if ((rc = m->stage[0].ensure((size_t)n * sizeof(orbgpu_keypoint)))) return rc;
    if ((rc = m->stage[1].ensure((size_t)n * sizeof(orbgpu_keypoint)))) return rc;
    if ((rc = m->stage[2].ensure((size_t)w * h * 4))) return rc;
    if ((rc = m->stage[3].ensure((size_t)n * 8))) return rc;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(m->stage[0].p, kps, (size_t)n * sizeof(orbgpu_keypoint), hipMemcpyHostToDevice, st));
    ORBGPU_HIP_CHECK(hipMemcpyAsync(m->stage[1].p, kps_un, (size_t)n * sizeof(orbgpu_keypoint), hipMemcpyHostToDevice, st));
    ORBGPU_HIP_CHECK(hipMemcpy2DAsync(m->stage[2].p, (size_t)w * 4, depth, (size_t)stride_floats * 4, (size_t)w * 4, (size_t)h,
                                      hipMemcpyHostToDevice, st));
    float *o = (float *)m->stage[3].p;
    if ((rc = launch_stereo_from_rgbd(st, (const orbgpu_keypoint *)m->stage[0].p, (const orbgpu_keypoint *)m->stage[1].p, n,
                                      (const float *)m->stage[2].p, w, h, w, bf, o, o + n)))
        return rc;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(uright, o, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipMemcpyAsync(out_depth, o + n, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    return ORBGPU_OK;
}

// Frame::UndistortKeyPoints (src/Frame.cc:517-558).
int orbgpu_undistort_keypoints(orbgpu_matcher *m, const orbgpu_keypoint *kps, int32_t n, const float *K4, const float *dist5,
                               orbgpu_keypoint *kps_un)
{
    if (!m || n < 0 || (n > 0 && (!kps || !kps_un)) || !K4 || !dist5 || K4[0] == 0.0f || K4[1] == 0.0f) {
        set_error("orbgpu_undistort_keypoints: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (n == 0) return ORBGPU_OK;
    if (dist5[0] == 0.0f) {  // :519-523: no distortion, plain copy
        if (kps_un != kps) memcpy(kps_un, kps, (size_t)n * sizeof(orbgpu_keypoint));
        return ORBGPU_OK;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    int rc;
    if ((rc = m->stage[0].ensure((size_t)n * sizeof(orbgpu_keypoint)))) return rc;
    if ((rc = m->stage[1].ensure((size_t)n * sizeof(orbgpu_keypoint)))) return rc;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(m->stage[0].p, kps, (size_t)n * sizeof(orbgpu_keypoint), hipMemcpyHostToDevice, st));
    if ((rc = launch_undistort(st, (const orbgpu_keypoint *)m->stage[0].p, n, K4, dist5, (orbgpu_keypoint *)m->stage[1].p))) return rc;
    ORBGPU_HIP_CHECK(hipMemcpyAsync(kps_un, m->stage[1].p, (size_t)n * sizeof(orbgpu_keypoint), hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    return ORBGPU_OK;
}

int orbgpu_search_for_initialization(orbgpu_matcher *m, const orbgpu_keypoint *kps1, const uint8_t *desc1, int32_t n1,
                                     const orbgpu_keypoint *kps2, const uint8_t *desc2, int32_t n2, const float *bounds4,
                                     float *prev_matched_xy, int32_t *matches12, int32_t window_size, float nnratio,
                                     int32_t check_orientation, int32_t *nmatches)
{
    TraceRange trace_range("orbgpu.search_for_initialization");
    if (!m || n1 < 0 || n2 < 0 || (n1 > 0 && (!kps1 || !desc1 || !prev_matched_xy || !matches12)) || (n2 > 0 && (!kps2 || !desc2)) ||
        !bounds4 || !nmatches) {
        set_error("orbgpu_search_for_initialization: invalid argument");
        return ORBGPU_E_INVALID;
    }
    // Window queries around vbPrevMatched, level-0 keypoints only (:444-453).
    std::vector<orbgpu_wquery> q((size_t)(n1 > 0 ? n1 : 1));
    for (int i = 0; i < n1; i++) {
        orbgpu_wquery &w = q[i];
        w.u = prev_matched_xy[2 * i];
        w.v = prev_matched_xy[2 * i + 1];
        w.r = (float)window_size;
        w.minLevel = kps1[i].octave;
        w.maxLevel = kps1[i].octave;
        w.angle = kps1[i].angle;
        w.ur = -1.0f;
        w.flags = kps1[i].octave > 0 ? 0 : 1;
    }
    int rc = window_host(m, 3, q.data(), desc1, n1, kps2, desc2, nullptr, n2, bounds4, nullptr, nullptr, matches12, ORBGPU_TH_LOW,
                         nnratio, check_orientation, nmatches);
    if (rc) return rc;
    for (int i = 0; i < n1; i++)  // :566-568
        if (matches12[i] >= 0) {
            prev_matched_xy[2 * i] = kps2[matches12[i]].x;
            prev_matched_xy[2 * i + 1] = kps2[matches12[i]].y;
        }
    return ORBGPU_OK;
}

// SearchForInitialization (src/ORBmatcher.cc:422-571) for a batch of frame pairs on the device, each with
// vbPrevMatched = frame 1's own keypoint positions (the state at the start of an initialisation).
int orbgpu_search_for_initialization_batch_device(orbgpu_matcher *m, int32_t npairs, const orbgpu_keypoint *d_kps1,
                                                  const uint8_t *d_desc1, const int32_t *d_n1, int32_t capacity1,
                                                  const orbgpu_keypoint *d_kps2, const uint8_t *d_desc2, const int32_t *d_n2,
                                                  int32_t capacity2, const float *bounds4_host, int32_t window_size, float nnratio,
                                                  int32_t check_orientation, int32_t *d_matches12, int32_t *d_nmatches)
{
    TraceRange trace_range("orbgpu.search_for_initialization");
    if (!m || npairs < 1 || capacity1 < 1 || capacity2 < 1 || !d_kps1 || !d_desc1 || !d_n1 || !d_kps2 || !d_desc2 || !d_n2 ||
        !bounds4_host || !d_matches12 || !d_nmatches) {
        set_error("orbgpu_search_for_initialization_batch_device: invalid argument");
        return ORBGPU_E_INVALID;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    const size_t qbytes = ((size_t)npairs * capacity1 * sizeof(orbgpu_wquery) + 255) & ~(size_t)255;
    int rc = m->initws.ensure(qbytes + (size_t)npairs * capacity2 * 4);
    if (rc) return rc;
    orbgpu_wquery *q = (orbgpu_wquery *)m->initws.p;
    int32_t *assign = (int32_t *)((uint8_t *)m->initws.p + qbytes);
    if ((rc = launch_init_queries_batch(m->stream, npairs, d_kps1, d_n1, capacity1, window_size, q))) return rc;
    return run_window(m, 3, npairs, q, d_desc1, d_n1, capacity1, d_kps2, d_desc2, nullptr, d_n2, capacity2, bounds4_host, nullptr,
                      assign, d_matches12, ORBGPU_TH_LOW, nnratio, check_orientation, d_nmatches);
}

int orbgpu_make_shift_queries_device(orbgpu_matcher *m, int32_t npairs, const orbgpu_keypoint *d_kps_last,
                                     const int32_t *d_n_last, int32_t capacity, float du, float dv, float th,
                                     const float *scale_host, int32_t nlevels, const float *bounds4_host, orbgpu_wquery *d_q)
{
    if (!m || npairs < 1 || !d_kps_last || !d_n_last || capacity < 1 || !scale_host || nlevels < 1 || nlevels > ORBGPU_MAX_LEVELS ||
        !bounds4_host || !d_q) {
        set_error("orbgpu_make_shift_queries_device: invalid argument");
        return ORBGPU_E_INVALID;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    ScaleTab sc;
    for (int i = 0; i < ORBGPU_MAX_LEVELS; i++) sc.s[i] = i < nlevels ? scale_host[i] : scale_host[nlevels - 1];
    return launch_shift_queries(m->stream, npairs, d_kps_last, d_n_last, capacity, du, dv, th, sc, bounds4_host, d_q);
}

int orbgpu_stereo_match(orbgpu_matcher *m, orbgpu_extractor *left, orbgpu_extractor *right, int32_t frame,
                        const orbgpu_keypoint *kpsL, const uint8_t *descL, int32_t nL, const orbgpu_keypoint *kpsR,
                        const uint8_t *descR, int32_t nR, float bf, float b, float *uRight, float *depth, int32_t *nvalid)
{
    TraceRange trace_range("orbgpu.stereo_match");
    if (!m || !left || !right || nL < 0 || nR < 0 || (nL > 0 && (!kpsL || !descL || !uRight || !depth)) ||
        (nR > 0 && (!kpsR || !descR)) || !nvalid || !(b > 0)) {
        set_error("orbgpu_stereo_match: invalid argument");
        return ORBGPU_E_INVALID;
    }
    *nvalid = 0;
    if (nL == 0) return ORBGPU_OK;
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    StereoArgs A;
    memset(&A, 0, sizeof(A));
    int32_t nlev = 0;
    int rc;
    if ((rc = orbgpu_extractor_tables(left, &nlev, A.scale, A.inv_scale, nullptr, nullptr, nullptr))) return rc;
    A.nlevels = nlev;
    for (int l = 0; l < nlev; l++) {
        int32_t w2, h2, p2;
        if ((rc = orbgpu_pyramid_level_device(left, frame, l, 0, &A.pyrL[l], &A.pitchL[l], &A.w[l], &A.h[l]))) return rc;
        if ((rc = orbgpu_pyramid_level_device(right, frame, l, 0, &A.pyrR[l], &p2, &w2, &h2))) return rc;
        A.pitchR[l] = p2;
        if (w2 != A.w[l] || h2 != A.h[l]) {
            set_error("left and right pyramids differ at level %d", l);
            return ORBGPU_E_INVALID;
        }
    }
    for (int i = 0; i < nL; i++)
        if (kpsL[i].octave < 0 || kpsL[i].octave >= nlev) {
            set_error("orbgpu_stereo_match: left keypoint %d has octave %d of %d levels", i, kpsL[i].octave, nlev);
            return ORBGPU_E_INVALID;
        }
    for (int i = 0; i < nR; i++)
        if (kpsR[i].octave < 0 || kpsR[i].octave >= nlev) {
            set_error("orbgpu_stereo_match: right keypoint %d has octave %d of %d levels", i, kpsR[i].octave, nlev);
            return ORBGPU_E_INVALID;
        }
    // the extractors may run on other streams: make their work visible first
    if ((rc = orbgpu_extractor_sync(left))) return rc;
    if ((rc = orbgpu_extractor_sync(right))) return rc;
    if (A.h[0] > 8192) {
        set_error("orbgpu_stereo_match: images of more than 8192 rows are not supported");
        return ORBGPU_E_UNSUPPORTED;
    }
    // One block each way, like the windowed searches: counts | left keypoints | left descriptors | right keypoints |
    // right descriptors go up; uRight | depth | matches with a depth come back.  Between them, on the device: the
    // row table (:610-627), the matching with its sub-pixel refinement, and the outlier cut on the SADs (:760-773).
    const int nRows = A.h[0];
    const size_t capR = nR > 0 ? nR : 1;
    const size_t span = (size_t)ceilf(4.0f * A.scale[nlev - 1]) + 3;  // a right keypoint lies in at most 2 r + 3 rows, r = 2 * scale
    const size_t items = capR * (span < (size_t)nRows ? span : (size_t)nRows);
    size_t off = 0;
    auto part = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_cnt = part(16), o_kL = part((size_t)nL * sizeof(orbgpu_keypoint)), o_dL = part((size_t)nL * 32);
    const size_t o_kR = part(capR * sizeof(orbgpu_keypoint)), o_dR = part(capR * 32);
    const size_t in_bytes = off;
    const size_t o_sp = part(capR * 4), o_rs = part(((size_t)nRows + 1) * 4), o_ri = part(items * 4), o_sad = part((size_t)nL * 4);
    const size_t o_ur = part((size_t)nL * 4), o_dp = part((size_t)nL * 4), o_nv = part(16);
    const size_t total = off, pinned = in_bytes + (total - o_ur);
    if ((rc = m->io.ensure(total))) return rc;
    if (pinned > m->h_io_bytes) {
        if (m->h_io) (void)hipHostFree(m->h_io);
        m->h_io = nullptr;
        m->h_io_bytes = 0;
        const size_t want = pinned + pinned / 4 + 4096;
        if (hipHostMalloc(&m->h_io, want, hipHostMallocDefault) != hipSuccess) {
            set_error("hipHostMalloc of %zu bytes failed", want);
            return ORBGPU_E_NOMEM;
        }
        m->h_io_bytes = want;
    }
    uint8_t *h = (uint8_t *)m->h_io, *d = (uint8_t *)m->io.p;
    int32_t *hc = (int32_t *)(h + o_cnt);
    hc[0] = nL; hc[1] = nR; hc[2] = 0; hc[3] = 0;
    memcpy(h + o_kL, kpsL, (size_t)nL * sizeof(orbgpu_keypoint));
    memcpy(h + o_dL, descL, (size_t)nL * 32);
    if (nR > 0) {
        memcpy(h + o_kR, kpsR, (size_t)nR * sizeof(orbgpu_keypoint));
        memcpy(h + o_dR, descR, (size_t)nR * 32);
    }
    ORBGPU_HIP_CHECK(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
    const int32_t *dc = (const int32_t *)(d + o_cnt);
    ScaleTab sc;
    for (int l = 0; l < ORBGPU_MAX_LEVELS; l++) sc.s[l] = A.scale[l < nlev ? l : nlev - 1];
    if ((rc = launch_stereo_rows(st, (const orbgpu_keypoint *)(d + o_kR), dc + 1, (int)capR, sc, nlev, nRows, (int32_t *)(d + o_sp),
                                 (int32_t *)(d + o_rs), (int32_t *)(d + o_ri))))
        return rc;
    A.kpsL = (const orbgpu_keypoint *)(d + o_kL); A.descL = d + o_dL; A.nL = nL;
    A.kpsR = (const orbgpu_keypoint *)(d + o_kR); A.descR = d + o_dR; A.nR = nR;
    A.row_start = (const int32_t *)(d + o_rs); A.row_items = (const int32_t *)(d + o_ri); A.nRows = nRows;
    A.bf = bf; A.b = b;
    A.uRight = (float *)(d + o_ur); A.depth = (float *)(d + o_dp); A.best_sad = (int32_t *)(d + o_sad);
    if ((rc = launch_stereo(st, A))) return rc;
    if ((rc = launch_stereo_outliers(st, A.best_sad, dc, nL, A.uRight, A.depth, (int32_t *)(d + o_nv)))) return rc;
    uint8_t *hb = h + in_bytes;  // the outputs' place in the page-locked block
    ORBGPU_HIP_CHECK(hipMemcpyAsync(hb, d + o_ur, total - o_ur, hipMemcpyDeviceToHost, st));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    memcpy(uRight, hb, (size_t)nL * 4);
    memcpy(depth, hb + (o_dp - o_ur), (size_t)nL * 4);
    *nvalid = *(const int32_t *)(hb + (o_nv - o_ur));
    return ORBGPU_OK;
}

// Frame::ComputeStereoMatches (src/Frame.cc:600-774) for a batch of stereo pairs whose keypoints, descriptors and
// pyramids are on the device: five launches for the whole batch (row spans + row starts, row lists, matching with the
// sub-pixel refinement, outlier cut), every one with the pair as a grid dimension.
int orbgpu_stereo_match_batch_device(orbgpu_matcher *m, orbgpu_extractor *left, orbgpu_extractor *right, int32_t npairs,
                                     const orbgpu_keypoint *d_kpsL, const uint8_t *d_descL, const int32_t *d_nL, int32_t capL,
                                     const orbgpu_keypoint *d_kpsR, const uint8_t *d_descR, const int32_t *d_nR, int32_t capR,
                                     float bf, float b, float *d_uRight, float *d_depth, int32_t *d_nvalid)
{
    TraceRange trace_range("orbgpu.stereo_match");
    if (!m || !left || !right || left == right || npairs < 1 || capL < 1 || capR < 1 || !d_kpsL || !d_descL || !d_nL || !d_kpsR ||
        !d_descR || !d_nR || !d_uRight || !d_depth || !d_nvalid || !(b > 0)) {
        set_error("orbgpu_stereo_match_batch_device: invalid argument");
        return ORBGPU_E_INVALID;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(m->device));
    StereoArgs A;
    memset(&A, 0, sizeof(A));
    int32_t nlev = 0, nlevR = 0;
    float scaleR[ORBGPU_MAX_LEVELS];
    int rc;
    if ((rc = orbgpu_extractor_tables(left, &nlev, A.scale, A.inv_scale, nullptr, nullptr, nullptr))) return rc;
    if ((rc = orbgpu_extractor_tables(right, &nlevR, scaleR, nullptr, nullptr, nullptr, nullptr))) return rc;
    if (nlevR != nlev || memcmp(scaleR, A.scale, sizeof(float) * (size_t)nlev) != 0) {
        set_error("orbgpu_stereo_match_batch_device: the two extractors have different pyramids");
        return ORBGPU_E_INVALID;
    }
    A.nlevels = nlev;
    for (int l = 0; l < nlev; l++) {
        int32_t w2, h2, p2;
        // frame npairs - 1 must exist in both extractors' last call
        const uint8_t *lastL, *lastR;
        if ((rc = orbgpu_pyramid_level_device(left, npairs - 1, l, 0, &lastL, nullptr, nullptr, nullptr))) return rc;
        if ((rc = orbgpu_pyramid_level_device(right, npairs - 1, l, 0, &lastR, nullptr, nullptr, nullptr))) return rc;
        if ((rc = orbgpu_pyramid_level_device(left, 0, l, 0, &A.pyrL[l], &A.pitchL[l], &A.w[l], &A.h[l]))) return rc;
        if ((rc = orbgpu_pyramid_level_device(right, 0, l, 0, &A.pyrR[l], &p2, &w2, &h2))) return rc;
        A.pitchR[l] = p2;
        if (w2 != A.w[l] || h2 != A.h[l]) {
            set_error("left and right pyramids differ at level %d", l);
            return ORBGPU_E_INVALID;
        }
        if (l == 0 && npairs > 1) {
            A.pyr_strideL = (lastL - A.pyrL[0]) / (npairs - 1);
            A.pyr_strideR = (lastR - A.pyrR[0]) / (npairs - 1);
        }
    }
    if (A.h[0] > 8192) {
        set_error("orbgpu_stereo_match_batch_device: images of more than 8192 rows are not supported");
        return ORBGPU_E_UNSUPPORTED;
    }
    const int nRows = A.h[0];
    const size_t span = (size_t)ceilf(4.0f * A.scale[nlev - 1]) + 3;  // a right keypoint lies in at most 2 r + 3 rows, r = 2 * scale
    const size_t items = (size_t)capR * (span < (size_t)nRows ? span : (size_t)nRows);
    if (items > 0x7FFFFFFF) {
        set_error("orbgpu_stereo_match_batch_device: the row tables of one pair exceed 2^31 entries");
        return ORBGPU_E_UNSUPPORTED;
    }
    // per pair: best SAD [capL] | row spans [capR] | row starts [nRows + 1] | row lists [items]
    const size_t n_sad = (size_t)npairs * capL, n_sp = (size_t)npairs * capR, n_rs = (size_t)npairs * (nRows + 1), n_ri = (size_t)npairs * items;
    if ((rc = m->stereo.ensure((n_sad + n_sp + n_rs + n_ri) * 4))) return rc;
    int32_t *best_sad = (int32_t *)m->stereo.p, *spans = best_sad + n_sad, *row_start = spans + n_sp, *row_items = row_start + n_rs;
    hipStream_t st = m->stream;
    ScaleTab sc;
    for (int l = 0; l < ORBGPU_MAX_LEVELS; l++) sc.s[l] = A.scale[l < nlev ? l : nlev - 1];
    if ((rc = launch_stereo_rows_batch(st, npairs, d_kpsR, d_nR, capR, sc, nlev, nRows, spans, row_start, row_items, (int)items))) return rc;
    A.kpsL = d_kpsL; A.descL = d_descL; A.nL = capL; A.d_nL = d_nL;
    A.kpsR = d_kpsR; A.descR = d_descR; A.nR = capR;
    A.row_start = row_start; A.row_items = row_items; A.nRows = nRows;
    A.bf = bf; A.b = b;
    A.uRight = d_uRight; A.depth = d_depth; A.best_sad = best_sad;
    A.strideL = capL; A.strideR = capR; A.row_stride = nRows + 1; A.items_stride = (int32_t)items;
    if ((rc = launch_stereo_batch(st, A, npairs))) return rc;
    return launch_stereo_outliers_batch(st, npairs, best_sad, d_nL, 1, capL, d_uRight, d_depth, d_nvalid, 1);
}

// ---- the same searches with a device-resident frame as the train side ------------------------------------------
// The frame's undistorted keypoints, descriptors, uRight and grid are read where orbgpu_frame_build_* left them.
// When the matcher carries a one-shot grid geometry (a KeyFrame's truncated bounds), its build origin must be the
// frame's: the frame's grid is used as built.

int orbgpu_search_by_projection_frame(orbgpu_matcher *m, int32_t mode, const orbgpu_wquery *q, const uint8_t *qdesc, int32_t nq,
                                      const orbgpu_frame *train, const uint8_t *occupied, int32_t *assign, int32_t th_dist,
                                      float nnratio, int32_t check_orientation, int32_t *nmatches)
{
    if (!m || mode < 0 || mode > 2 || nq < 0 || (nq > 0 && (!q || !qdesc)) || !nmatches || !train || (train->n > 0 && !assign)) {
        set_error("orbgpu_search_by_projection_frame: invalid argument");
        return ORBGPU_E_INVALID;
    }
    int rc = check_train_frame("orbgpu_search_by_projection_frame", m, train);
    if (rc) return rc;
    WindowDev dev;
    dev.train = train;
    return window_host(m, mode, q, qdesc, nq, nullptr, nullptr, train->uright, train->n, train->bounds4, occupied, assign, nullptr,
                       th_dist, nnratio, check_orientation, nmatches, nullptr, nullptr, 0, nullptr, &dev);
}

int orbgpu_search_in_area_frame(orbgpu_matcher *m, const orbgpu_wquery *q, const uint8_t *qdesc, int32_t nq, const orbgpu_frame *train,
                                const float *inv_sigma2, int32_t nlevels, int32_t chi2_gate, int32_t th_dist, int32_t *qmatch,
                                int32_t *nmatches)
{
    if (!m || nq < 0 || (nq > 0 && (!q || !qdesc || !qmatch)) || !train || !nmatches ||
        (chi2_gate && (!inv_sigma2 || nlevels < 1 || nlevels > ORBGPU_MAX_LEVELS))) {
        set_error("orbgpu_search_in_area_frame: invalid argument");
        return ORBGPU_E_INVALID;
    }
    int rc = check_train_frame("orbgpu_search_in_area_frame", m, train);
    if (rc) return rc;
    MatchArgs X;
    memset(&X, 0, sizeof(X));
    X.chi2_gate = chi2_gate ? 1 : 0;
    if (chi2_gate) {
        if (nlevels < train->nlevels) {
            set_error("orbgpu_search_in_area_frame: %d levels given, the frame's keypoints span %d", nlevels, train->nlevels);
            return ORBGPU_E_INVALID;
        }
        for (int l = 0; l < nlevels; l++) X.inv_sigma2.s[l] = inv_sigma2[l];
    }
    WindowDev dev;
    dev.train = train;
    return window_host(m, 6, q, qdesc, nq, nullptr, nullptr, chi2_gate ? train->uright : nullptr, train->n, train->bounds4, nullptr,
                       nullptr, qmatch, th_dist, 1.0f, 0, nmatches, nullptr, nullptr, 0, &X, &dev);
}

int orbgpu_search_by_bow_frame(orbgpu_matcher *m, const orbgpu_wquery *q, const uint8_t *qdesc, int32_t nq, const orbgpu_frame *train,
                               const int32_t *seg_start, const int32_t *seg_index, int32_t nseg, const uint8_t *excluded,
                               int32_t th_dist, float nnratio, int32_t check_orientation, int32_t *assign, int32_t *qmatch,
                               int32_t *nmatches)
{
    if (!m || nq < 0 || nseg < 0 || (nq > 0 && (!q || !qdesc)) || !train || !seg_start || (nseg > 0 && !seg_index) || !nmatches ||
        (train->n > 0 && !assign)) {
        set_error("orbgpu_search_by_bow_frame: invalid argument");
        return ORBGPU_E_INVALID;
    }
    int rc = check_train_frame("orbgpu_search_by_bow_frame", m, train);
    if (rc) return rc;
    if ((rc = check_segments("orbgpu_search_by_bow_frame", q, nq, train->n, seg_start, seg_index, nseg))) return rc;
    const float b4[4] = {0.f, 0.f, 1.f, 1.f};
    WindowDev dev;
    dev.train = train;
    return window_host(m, 4, q, qdesc, nq, nullptr, nullptr, nullptr, train->n, b4, excluded, assign, qmatch, th_dist, nnratio,
                       check_orientation, nmatches, seg_start, seg_index, nseg, nullptr, &dev);
}

int orbgpu_search_for_triangulation_frame(orbgpu_matcher *m, const orbgpu_wquery *q, const uint8_t *qdesc, int32_t nq,
                                          const orbgpu_frame *train, int32_t use_uright, const int32_t *seg_start,
                                          const int32_t *seg_index, int32_t nseg, const uint8_t *excluded, const float *F12, float ex,
                                          float ey, const float *scale2, const float *sigma2, int32_t nlevels, int32_t only_stereo,
                                          int32_t check_orientation, int32_t *qmatch, int32_t *nmatches)
{
    if (!m || nq < 0 || nseg < 0 || (nq > 0 && (!q || !qdesc || !qmatch)) || !train || !seg_start || (nseg > 0 && !seg_index) || !F12 ||
        !scale2 || !sigma2 || nlevels < 1 || nlevels > ORBGPU_MAX_LEVELS || !nmatches) {
        set_error("orbgpu_search_for_triangulation_frame: invalid argument");
        return ORBGPU_E_INVALID;
    }
    int rc = check_train_frame("orbgpu_search_for_triangulation_frame", m, train);
    if (rc) return rc;
    if ((rc = check_segments("orbgpu_search_for_triangulation_frame", q, nq, train->n, seg_start, seg_index, nseg))) return rc;
    if (nlevels < train->nlevels) {
        set_error("orbgpu_search_for_triangulation_frame: %d levels given, the frame's keypoints span %d", nlevels, train->nlevels);
        return ORBGPU_E_INVALID;
    }
    MatchArgs X;
    memset(&X, 0, sizeof(X));
    for (int k = 0; k < 9; k++) X.F12[k] = F12[k];
    X.ex = ex; X.ey = ey; X.only_stereo = only_stereo;
    for (int l = 0; l < nlevels; l++) { X.scale2.s[l] = scale2[l]; X.sigma2.s[l] = sigma2[l]; }
    const float b4[4] = {0.f, 0.f, 1.f, 1.f};
    WindowDev dev;
    dev.train = train;
    return window_host(m, 5, q, qdesc, nq, nullptr, nullptr, use_uright ? train->uright : nullptr, train->n, b4, excluded, nullptr, qmatch,
                       ORBGPU_TH_LOW, 1.0f, check_orientation, nmatches, seg_start, seg_index, nseg, &X, &dev);
}

// SearchForInitialization between two frames: the queries are made on the device from frame 1's undistorted
// keypoints and the caller's vbPrevMatched, and vbPrevMatched is updated there too (:566-568).
int orbgpu_search_for_initialization_frames(orbgpu_matcher *m, const orbgpu_frame *f1, const orbgpu_frame *f2, float *prev_matched_xy,
                                            int32_t *matches12, int32_t window_size, float nnratio, int32_t check_orientation,
                                            int32_t *nmatches)
{
    if (!m || !f1 || !f2 || !nmatches || (f1->n > 0 && (!prev_matched_xy || !matches12))) {
        set_error("orbgpu_search_for_initialization_frames: invalid argument");
        return ORBGPU_E_INVALID;
    }
    int rc = check_train_frame("orbgpu_search_for_initialization_frames", m, f1);
    if (rc) return rc;
    if ((rc = check_train_frame("orbgpu_search_for_initialization_frames", m, f2))) return rc;
    WindowDev dev;
    dev.train = f2;
    dev.init_f1 = f1;
    dev.prev_xy = prev_matched_xy;
    dev.window = window_size;
    return window_host(m, 3, nullptr, nullptr, f1->n, nullptr, nullptr, nullptr, f2->n, f2->bounds4, nullptr, nullptr, matches12, ORBGPU_TH_LOW,
                       nnratio, check_orientation, nmatches, nullptr, nullptr, 0, nullptr, &dev);
}

}  // extern "C"
