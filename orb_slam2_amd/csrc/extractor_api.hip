// C-ABI entry points of the extractor half of liborbgpu (see include/orbgpu.h).
// Reference lines are relative to /root/reference/src/ORB_SLAM2/.
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "orbgpu_internal.h"

namespace orbgpu {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int launch_halo_full(hipStream_t st, int nimg, uint8_t *pyr, const Geom &G);
int launch_pyramid(hipStream_t st, const uint8_t *d_imgs, int nimg, int stride, size_t frame_stride, uint8_t *pyr,
                   const Geom &G, const XTab *xtab, const YTab *ytab, const FuseArgs *fuse, int32_t *status_clear);
int launch_fast(hipStream_t st, int nimg, const uint8_t *pyr, const Geom &G, const uint32_t *cell_tab, uint32_t *slots,
                int32_t *cell_counts, const LaunchTuning &tune);
int launch_octree(hipStream_t st, int nimg, const Geom &G, const uint32_t *slots, const int32_t *cell_counts, uint32_t *cand,
                  QNode *nodes, int32_t *scratch, uint32_t *sel, int32_t *sel_count, int32_t *cand_count, int32_t *status,
                  const LaunchTuning &tune, const BlurWork *with_blur, bool *blur_done);
int launch_blur(hipStream_t st, int nimg, const uint8_t *pyr, uint8_t *blur, const Geom &G, const uint32_t *tile_tab, int ntiles);
int blur_tile_table(const Geom &G, std::vector<uint32_t> *tab);
int launch_describe(hipStream_t st, int nimg, const uint8_t *pyr, const uint8_t *blur, const Geom &G, const uint32_t *sel,
                    const int32_t *sel_count, const int8_t *pattern, orbgpu_keypoint *kps, uint8_t *desc, int32_t *counts,
                    int capacity, int32_t *status);

namespace {
typedef int (*roctx_push_fn)(const char *);
typedef int (*roctx_pop_fn)();
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
std::once_flag g_roctx_once;
void roctx_resolve()
{
    const char *on = getenv("ORBGPU_ROCTX");
    if (!on || atoi(on) == 0) return;
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
        void *hnd = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
        if (!hnd) continue;
        roctx_push_fn push = (roctx_push_fn)dlsym(hnd, "roctxRangePushA");
        roctx_pop_fn pop = (roctx_pop_fn)dlsym(hnd, "roctxRangePop");
        if (push && pop) {
            g_roctx_push = push;
            g_roctx_pop = pop;
            return;
        }
    }
}
}  // namespace

TraceRange::TraceRange(const char *name)
{
    std::call_once(g_roctx_once, roctx_resolve);
    on_ = g_roctx_push != nullptr;
    if (on_) (void)g_roctx_push(name);
}
TraceRange::~TraceRange()
{
    if (on_) (void)g_roctx_pop();
}

int wait_stream_short(hipStream_t st, int spin_us)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return ORBGPU_OK;
        if (q != hipErrorNotReady) {
            set_error("hipStreamQuery failed: %s", hipGetErrorString(q));
            return ORBGPU_E_HIP;
        }
        if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
    }
    ORBGPU_HIP_CHECK(hipStreamSynchronize(st));
    return ORBGPU_OK;
}

static const int8_t kPattern[1024] = {
#include "brief_pattern.inc"
};

static inline int cv_round(float v) { return (int)lrintf(v); }  // round half to even, like cvRound
static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need)
    {
        if (need <= bytes) return ORBGPU_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        if (hipMalloc(&p, need) != hipSuccess) {
            set_error("hipMalloc of %zu bytes failed", need);
            return ORBGPU_E_NOMEM;
        }
        bytes = need;
        return ORBGPU_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

// A few persistent host threads for the per-frame copies of the host batch entry points (staging pageable images
// into pinned memory, handing results out): started with the handle's first large batch, joined with the handle.
// One thread moves only a few GB/s, and starting threads per call costs more than the copies of a small chunk.
class CopyPool {
public:
    explicit CopyPool(int nthreads)
    {
        for (int t = 0; t < nthreads; t++) workers_.emplace_back([this]() { loop(); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        wake_.notify_all();
        for (auto &w : workers_) w.join();
    }
    // fn(i) for i in [0, n), on the workers and the calling thread; returns when all are done
    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 0) return;
        unsigned gen;
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; n_ = n; next_ = 0; pending_ = n;
            gen = ++generation_;
        }
        wake_.notify_all();
        work(gen);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this]() { return pending_ == 0; });
    }

private:
    // items are whole frames (hundreds of KB each), so handing them out under the mutex costs nothing; a worker that
    // is still leaving an earlier run can never take an item of the current one under the old run's terms
    void work(unsigned gen)
    {
        for (;;) {
            const std::function<void(int)> *fn;
            int i;
            {
                std::lock_guard<std::mutex> lk(m_);
                if (generation_ != gen || next_ >= n_) return;
                i = next_++;
                fn = fn_;
            }
            (*fn)(i);
            std::lock_guard<std::mutex> lk(m_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop()
    {
        unsigned seen = 0;
        for (;;) {
            unsigned gen;
            {
                std::unique_lock<std::mutex> lk(m_);
                wake_.wait(lk, [&]() { return quit_ || (generation_ != seen && next_ < n_); });
                if (quit_) return;
                seen = gen = generation_;
            }
            work(gen);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(int)> *fn_ = nullptr;
    int next_ = 0, n_ = 0, pending_ = 0;
    unsigned generation_ = 0;
    bool quit_ = false;
};

}  // namespace orbgpu

using namespace orbgpu;

struct orbgpu_extractor {
    orbgpu_params p;
    LaunchTuning tune;  // environment hooks, read once in orbgpu_extractor_create
    float scale[ORBGPU_MAX_LEVELS], inv_scale[ORBGPU_MAX_LEVELS], sigma2[ORBGPU_MAX_LEVELS], inv_sigma2[ORBGPU_MAX_LEVELS];
    int32_t quota[ORBGPU_MAX_LEVELS];
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t aux_stream = nullptr;          // the quadtree runs here, beside the blur on the main stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    Geom G;
    int geom_w = 0, geom_h = 0;
    int last_nimg = 0;
    bool halo_full = false;  // the last extraction's levels carry their halo columns (filled on demand)
    // the host-pointer entry point replays its launches (upload, eleven kernels on two streams, downloads) as one
    // HIP graph, captured the first time a (frames, width, height) combination is seen: a single frame is bound by
    // launch overhead, not by the kernels
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_nimg = 0, graph_w = 0, graph_h = 0;
    bool graph_broken = false;  // capture failed once: plain launches from then on
    DevBuf pyr, blur, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, pattern, xtab, ytab, cell_tab, blur_tab;
    DevBuf fuse_tab;
    FuseArgs fuse = {nullptr, 0, 0};  // the single-launch pyramid of a few frames (k_pyr_fused), if the geometry allows it
    int blur_tiles = 0;
    // staging for the host entry points: images; and one block for keypoints | descriptors | counts | status word, so
    // that a small call brings everything back with one copy
    DevBuf st_img, st_out;
    int32_t *status_cur = nullptr;  // the status word of the call in flight (status.p, or the one inside st_out)
    void *h_pinned = nullptr;
    size_t h_pinned_bytes = 0;
    // chunked host batches: uploads and downloads on their own streams beside the kernels
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    std::vector<hipEvent_t> chunk_ev;  // three per chunk: uploaded, extracted, downloaded
    CopyPool *pool = nullptr;
    // optional stage timing: a ring of event sets, one set per extract call
    bool profile = false;
    bool serialize = false;  // profiling mode 2: the quadtree stays on the main stream so every stage is timed alone
    int last_schedule = 0;  // the last call: 0 = blur, quadtree one after the other on the main stream; 1 = quadtree, 2 = blur on the second stream
    std::vector<hipEvent_t> ev;  // kProfSets * kProfEvents
    int prof_next = 0, prof_used = 0;
};

// start, pyramid, fast, blur (octree when serial), describe | second stage start, end: the quadtree on its own stream, or (serial) the blur
static constexpr int kProfSets = 128, kProfEvents = 7;

// ORBextractor::ORBextractor tables (src/ORBextractor.cc:469-512).
static void compute_tables(orbgpu_extractor *e)
{
    const int L = e->p.nlevels;
    e->scale[0] = 1.0f;
    e->sigma2[0] = 1.0f;
    for (int i = 1; i < L; i++) {
        e->scale[i] = e->scale[i - 1] * e->p.scaleFactor;
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < L; i++) {
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    const float factor = 1.0f / e->p.scaleFactor;
    float nDesired = (float)e->p.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
        e->quota[l] = cv_round(nDesired);
        sum += e->quota[l];
        nDesired *= factor;
    }
    e->quota[L - 1] = e->p.nfeatures - sum > 0 ? e->p.nfeatures - sum : 0;
}

static short sat_short(float v)
{
    const int i = cv_round(v);
    return (short)(i < -32768 ? -32768 : i > 32767 ? 32767 : i);
}

// Level geometry for a w x h input, plus the bilinear coefficient tables
// (cv::resize INTER_LINEAR, fixed point: 11-bit weights from float fractions).
static int build_geometry(const orbgpu_extractor *e, int w, int h, Geom *G, std::vector<XTab> *xt, std::vector<YTab> *yt)
{
    memset(G, 0, sizeof(*G));
    const int L = e->p.nlevels;
    G->nlevels = L;
    G->iniTh = e->p.iniThFAST < 0 ? 0 : e->p.iniThFAST > 255 ? 255 : e->p.iniThFAST;
    G->minTh = e->p.minThFAST < 0 ? 0 : e->p.minThFAST > 255 ? 255 : e->p.minThFAST;
    G->gauss_variant = e->p.gauss_variant;
    int64_t pyr = 0, slot = 0, cand = 0, node = 0;
    int cells = 0, kp = 0;
    if (xt) xt->clear();
    if (yt) yt->clear();
    for (int l = 0; l < L; l++) {
        LevelGeom &g = G->lv[l];
        g.w = cv_round((float)w * e->inv_scale[l]);  // :1536
        g.h = cv_round((float)h * e->inv_scale[l]);
        if (g.w < 62 || g.h < 62 || g.w - 2 * kMinBorder > 4095 || g.h - 2 * kMinBorder > 4095) {
            set_error("level %d is %dx%d: supported level sizes are 62..4127 pixels per side", l, g.w, g.h);
            return ORBGPU_E_UNSUPPORTED;
        }
        g.pitch = (int32_t)align_up(kLeftMargin + g.w + kEdge, 64);
        g.plane_off = pyr;
        pyr = align_up(pyr + (int64_t)g.pitch * (g.h + 2 * kEdge), 256);
        // cell grid (:1033-1040)
        const float width = (float)(g.w - 2 * kMinBorder), height = (float)(g.h - 2 * kMinBorder);
        g.nCols = (int)(width / 30.f);
        g.nRows = (int)(height / 30.f);
        g.wCell = (int)ceilf(width / (float)g.nCols);
        g.hCell = (int)ceilf(height / (float)g.nRows);
        if (g.wCell + 6 > kMaxTile || g.hCell + 6 > kMaxTile) {
            set_error("level %d: cell %dx%d too large", l, g.wCell, g.hCell);
            return ORBGPU_E_UNSUPPORTED;
        }
        g.cell_off = cells;
        cells += g.nCols * g.nRows;
        g.cell_cap = ((g.wCell + 1) / 2) * ((g.hCell + 1) / 2);
        g.slot_off = slot;
        slot += (int64_t)g.nCols * g.nRows * g.cell_cap;
        g.cand_cap = g.nCols * g.nRows * g.cell_cap;
        g.cand_off = cand;
        cand += 2 * (int64_t)g.cand_cap;
        g.quota = e->quota[l];
        // quadtree roots (:696-699)
        g.nIni = (int)roundf(width / height);
        if (g.nIni < 1 || g.nIni > kMaxRoots) {
            set_error("level %d: aspect ratio %dx%d gives %d quadtree roots (supported 1..%d)", l, g.w, g.h, g.nIni, kMaxRoots);
            return ORBGPU_E_UNSUPPORTED;
        }
        g.hX = width / (float)g.nIni;
        int nc = g.quota + 3 > 4 * g.nIni ? g.quota + 3 : 4 * g.nIni;
        g.node_cap = (int32_t)align_up(nc + 1, 4);
if (g.node_cap > (1 << 20)) {
            set_error("level %d: %d features exceed the quadtree capacity of 2^20 per level", l, g.quota);
            return ORBGPU_E_UNSUPPORTED;
        }
        g.node_off = node;
        node += 2 * (int64_t)g.node_cap;
        g.kp_cap = g.node_cap;
        g.kp_off = kp;
        kp += g.kp_cap;
        g.scale = e->scale[l];
        g.patch_size = (int)(31 * e->scale[l]);  // :1120
        if (l > 0 && xt && yt) {
            const LevelGeom &s = G->lv[l - 1];
            g.xtab_off = (int32_t)xt->size();
            g.ytab_off = (int32_t)yt->size();
            const double scale_x = 1. / ((double)g.w / s.w), scale_y = 1. / ((double)g.h / s.h);
            for (int dx = 0; dx < g.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = (int)floorf(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= s.w - 1) { fx = 0; sx = s.w - 1; }
                XTab t;
                t.sx = sx;
                t.a0 = sat_short((1.f - fx) * 2048);
                t.a1 = sat_short(fx * 2048);
                xt->push_back(t);
            }
            for (int dy = 0; dy < g.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = (int)floorf(fy);
                fy -= sy;
                YTab t;
                t.sy0 = sy < 0 ? 0 : (sy >= s.h ? s.h - 1 : sy);
                t.sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= s.h ? s.h - 1 : sy + 1);
                t.b0 = sat_short((1.f - fy) * 2048);
                t.b1 = sat_short(fy * 2048);
                // bits 0..15: the row that mirrors dy in the halo rows above / below, plus 32 (dy itself when it has none);
                // bit 16: the upper source row is the previous output row's lower one (its sums can be kept)
                const int ym = (dy >= 1 && dy <= kEdge) ? -dy : ((dy >= g.h - 1 - kEdge && dy <= g.h - 2) ? 2 * (g.h - 1) - dy : dy);
                const bool keep = dy > 0 && yt->back().sy1 == t.sy0;
                t.pad = (ym + 32) | (keep ? 1 << 16 : 0);
                yt->push_back(t);
            }
            // source window staged by the largest kRW x kRH output tile (dword-aligned on the left)
            const XTab *X = xt->data() + g.xtab_off;
            const YTab *Y = yt->data() + g.ytab_off;
            int maxdw = 1, maxrows = 1;
            for (int tx0 = 0; tx0 < g.w; tx0 += kRW) {
                const int last = X[tx0 + kRW - 1 < g.w ? tx0 + kRW - 1 : g.w - 1].sx + 1;
                const int ndw = ((last - (X[tx0].sx & ~3)) >> 2) + 1;
                maxdw = ndw > maxdw ? ndw : maxdw;
            }
            for (int ty0 = 0; ty0 < g.h; ty0 += kRH) {
                const int rows = Y[ty0 + kRH - 1 < g.h ? ty0 + kRH - 1 : g.h - 1].sy1 - Y[ty0].sy0 + 1;
                maxrows = rows > maxrows ? rows : maxrows;
            }
            int maxrows8 = 1;
            for (int ty0 = 0; ty0 < g.h; ty0 += 8) {
                const int rows = Y[ty0 + 7 < g.h ? ty0 + 7 : g.h - 1].sy1 - Y[ty0].sy0 + 1;
                maxrows8 = rows > maxrows8 ? rows : maxrows8;
            }
            g.rs_rows8 = maxrows8;
            g.rs_rows = maxrows;
            g.rs_pitch_dw = maxdw | 1;
            g.rs_magic = 0xFFFFFFFFu / (uint32_t)g.rs_pitch_dw + 1u;
            if ((int64_t)g.rs_rows * g.rs_pitch_dw * 4 > 64 * 1024) {
                set_error("level %d: scale factor too large for the resize tile (%d x %d source dwords)", l, g.rs_rows, g.rs_pitch_dw);
                return ORBGPU_E_UNSUPPORTED;
            }
        }
    }
    for (int l = 0; l < ORBGPU_MAX_LEVELS; l++) {
        const LevelGeom &g = G->lv[l < L ? l : L - 1];
        G->t_kp_off[l] = l < L ? g.kp_off : 0x7FFFFFFF;
        G->t_pitch[l] = g.pitch;
        G->t_interior[l] = (int32_t)interior_off(g);
        G->t_patch[l] = g.patch_size;
        G->t_scale[l] = g.scale;
    }
    if (pyr + 64 > 0x7FFFFFFF) {
        set_error("pyramid of one frame exceeds 2 GiB");
        return ORBGPU_E_UNSUPPORTED;
    }
    G->total_cells = cells;
    G->total_kp_cap = kp;
    G->pyr_frame_bytes = align_up(pyr + 64, 256);
    G->slot_frame_u32 = slot;
    G->cand_frame_u32 = cand;
    G->node_frame = node;
    return ORBGPU_OK;
}

static int ensure_buffers(orbgpu_extractor *e, const Geom &G, int batch, int img_w, int img_h)
{
    int rc;
    const size_t B = (size_t)batch;
#define ENS(buf, bytes) if ((rc = e->buf.ensure(bytes)) != ORBGPU_OK) return rc
    ENS(pyr, B * (size_t)G.pyr_frame_bytes);
    ENS(blur, B * (size_t)G.pyr_frame_bytes);
    ENS(slots, B * (size_t)G.slot_frame_u32 * 4);
    ENS(cell_counts, B * (size_t)G.total_cells * 4);
    ENS(cand, B * (size_t)G.cand_frame_u32 * 4);
    ENS(nodes, B * (size_t)G.node_frame * sizeof(QNode));
    ENS(scratch, B * (size_t)G.node_frame * kScratchPerNode * sizeof(int32_t));
    ENS(sel, B * (size_t)G.total_kp_cap * 4);
    ENS(sel_count, B * ORBGPU_MAX_LEVELS * 4);
    ENS(cand_count, B * ORBGPU_MAX_LEVELS * 4);
    ENS(status, 64);
    ENS(st_img, B * (size_t)img_w * img_h);
    ENS(st_out, B * (size_t)G.total_kp_cap * (sizeof(orbgpu_keypoint) + 32) + B * 4 + 64);
#undef ENS
    // pinned staging: outputs (keypoints, descriptors, counts, status) followed by the input images
    const size_t pin = B * (size_t)G.total_kp_cap * (sizeof(orbgpu_keypoint) + 32) + B * 4 + 64 + B * (size_t)img_w * img_h;
    if (pin > e->h_pinned_bytes) {
        if (e->h_pinned) (void)hipHostFree(e->h_pinned);
        e->h_pinned = nullptr;
        e->h_pinned_bytes = 0;
        if (hipHostMalloc(&e->h_pinned, pin, hipHostMallocDefault) != hipSuccess) {
            set_error("hipHostMalloc of %zu bytes failed", pin);
            return ORBGPU_E_NOMEM;
        }
        e->h_pinned_bytes = pin;
    }
    return ORBGPU_OK;
}

// Select the geometry for w x h (recomputing tables when the size changes).
static void drop_graph(orbgpu_extractor *e);

static int set_geometry_unchecked(orbgpu_extractor *e, int w, int h);

static int set_geometry(orbgpu_extractor *e, int w, int h)
{
    if (w == e->geom_w && h == e->geom_h) return ORBGPU_OK;
    if (w > e->p.max_width || h > e->p.max_height) {
        set_error("image %dx%d exceeds the handle's max %dx%d", w, h, e->p.max_width, e->p.max_height);
        return ORBGPU_E_INVALID;
    }
    drop_graph(e);  // the captured launches name the old geometry's tables and buffers
    // Buffers may be reallocated and tables overwritten before the new geometry is complete: until it is, the handle
    // has no geometry, so that a failure here makes the next call start over instead of launching on a mixture.
    e->geom_w = e->geom_h = 0;
    e->last_nimg = 0;  // the pyramid of the last call is gone with the old buffers
    return set_geometry_unchecked(e, w, h);
}

static int set_geometry_unchecked(orbgpu_extractor *e, int w, int h)
{
    Geom G;
    std::vector<XTab> xt;
    std::vector<YTab> yt;
    int rc = build_geometry(e, w, h, &G, &xt, &yt);
    if (rc) return rc;
    if ((rc = ensure_buffers(e, G, e->p.max_batch, w, h))) return rc;
    if ((rc = e->xtab.ensure(xt.size() * sizeof(XTab) + 16))) return rc;
    if ((rc = e->ytab.ensure(yt.size() * sizeof(YTab) + 16))) return rc;
    ORBGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    // FAST cell table, eight words per cell (the geometry of src/ORBextractor.cc:1043-1072 and where the cell's data
    // lie, so that the kernel needs neither divisions nor the level table):
    //   0: x | y << 16 of the cell's first interior pixel relative to (16,16) (3 + column * wCell, 3 + row * hCell)
    //   1: state | cw << 8 | ch << 16 (state 1: skipped by the rules of :1053 / :1068, 2: searched; cw x ch = cell
    //      image incl. the 3 px apron)
    //   2: byte offset, within a frame's pyramid, of the byte left of the cell image's first pixel
    //   3: ceil(2^20 / interior width)    4: ceil(2^20 / groups of four interior pixels per row)
    //   5: row pitch of the level | slot capacity of the cell << 16
    //   6: offset (in words) of the cell's candidate slot within a frame's slots     7: unused
    std::vector<uint32_t> ctab((size_t)G.total_cells * 8, 0u);
    for (int l = 0; l < G.nlevels; l++) {
        const LevelGeom &g = G.lv[l];
        const int maxBX = g.w - kMinBorder, maxBY = g.h - kMinBorder;
        for (int i = 0; i < g.nRows; i++)
            for (int j = 0; j < g.nCols; j++) {
                const size_t ci = (size_t)i * g.nCols + j;
                uint32_t *r = &ctab[((size_t)g.cell_off + ci) * 8];
                r[0] = (uint32_t)(3 + j * g.wCell) | ((uint32_t)(3 + i * g.hCell) << 16);
                const int iniY = kMinBorder + i * g.hCell, iniX = kMinBorder + j * g.wCell;
                int state = 1, cw = 0, ch = 0;
                if (!(iniY >= maxBY - 3 || iniX >= maxBX - 6)) {
                    cw = std::min(iniX + g.wCell + 6, maxBX) - iniX;
                    ch = std::min(iniY + g.hCell + 6, maxBY) - iniY;
                    if (cw > 6 && ch > 6) state = 2;
                }
                r[1] = (uint32_t)state | ((uint32_t)cw << 8) | ((uint32_t)ch << 16);
                r[2] = (uint32_t)(interior_off(g) + (int64_t)iniY * g.pitch + (iniX - 1));
                if (state == 2) {
                    const int wi = cw - 6, ngx = (wi + 3) >> 2;
                    r[3] = (uint32_t)(((1 << 20) + wi - 1) / wi);
                    r[4] = (uint32_t)(((1 << 20) + ngx - 1) / ngx);
                }
                r[5] = (uint32_t)g.pitch | ((uint32_t)g.cell_cap << 16);
                r[6] = (uint32_t)(g.slot_off + (int64_t)ci * g.cell_cap);
            }
    }
    if ((rc = e->cell_tab.ensure(ctab.size() * 4 + 16))) return rc;
    ORBGPU_HIP_CHECK(hipMemcpy(e->cell_tab.p, ctab.data(), ctab.size() * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> btab;
    e->blur_tiles = blur_tile_table(G, &btab);
    if ((rc = e->blur_tab.ensure(btab.size() * 4 + 16))) return rc;
    ORBGPU_HIP_CHECK(hipMemcpy(e->blur_tab.p, btab.data(), btab.size() * 4, hipMemcpyHostToDevice));
    ORBGPU_HIP_CHECK(hipMemcpy(e->xtab.p, xt.data(), xt.size() * sizeof(XTab), hipMemcpyHostToDevice));
    ORBGPU_HIP_CHECK(hipMemcpy(e->ytab.p, yt.data(), yt.size() * sizeof(YTab), hipMemcpyHostToDevice));
    // Regions of the fused pyramid: a kFW x kFH tile of the last level, then level by level the taps of the region
    // above (first tap of its first column / row .. second tap of its last), extended to the level's edge where the
    // region above touches its own.  While every scale step is at most 2 each column (row) of a level is a tap of
    // the next, so the regions of all tiles cover every level.  The region of level 0 is read from global memory in
    // whole dwords (the rows have that much room behind the level).
    e->fuse.tab = nullptr;
    if (G.nlevels > 1) {
        const LevelGeom &gl = G.lv[G.nlevels - 1];
        const int tX = (gl.w + kFW - 1) / kFW, tY = (gl.h + kFH - 1) / kFH;
        std::vector<FuseRegion> ftab((size_t)tX * tY * G.nlevels);
        bool ok = true;
        int64_t half = 0, entries = 0;
        for (int l = 1; l < G.nlevels; l++)
            if (G.lv[l - 1].w > 2 * G.lv[l].w || G.lv[l - 1].h > 2 * G.lv[l].h) ok = false;
        for (int t = 0; t < tX * tY && ok; t++) {
            const int tx = t % tX, ty = t / tX;
            int x0 = tx * kFW, y0 = ty * kFH, x1 = std::min(tx * kFW + kFW, gl.w) - 1, y1 = std::min(ty * kFH + kFH, gl.h) - 1;
            int64_t ne = 0;
            for (int l = G.nlevels - 1; l >= 0; l--) {
                if (l == 0) { x0 &= ~3; x1 |= 3; }
                const FuseRegion r = {(int16_t)x0, (int16_t)y0, (int16_t)x1, (int16_t)y1};
                ftab[(size_t)t * G.nlevels + l] = r;
                const int64_t rw = x1 - x0 + 1, rh = y1 - y0 + 1;
                if (l >= 1) ne += std::max(rw, rh);
                if (l < G.nlevels - 1) half = std::max(half, rw * rh);
                if (rw >= 500 || rw * rh >= 65536) ok = false;  // the kernel's index arithmetic
                if (l == 0) break;
                const LevelGeom &gd = G.lv[l], &gs = G.lv[l - 1];
                const XTab *xl = &xt[gd.xtab_off];
                const YTab *yl = &yt[gd.ytab_off];
                const int nx0 = x0 == 0 ? 0 : xl[x0].sx, nx1 = x1 == gd.w - 1 ? gs.w - 1 : std::min(xl[x1].sx + 1, gs.w - 1);
                const int ny0 = y0 == 0 ? 0 : yl[y0].sy0, ny1 = y1 == gd.h - 1 ? gs.h - 1 : std::min(yl[y1].sy1, gs.h - 1);
                x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1;
            }
            entries = std::max(entries, ne);
        }
        half = (half + 15) & ~(int64_t)15;
        entries = (entries + 1) & ~(int64_t)1;
        if (ok && entries * 24 + 2 * half <= 60 * 1024) {
            if ((rc = e->fuse_tab.ensure(ftab.size() * sizeof(FuseRegion) + 16))) return rc;
            ORBGPU_HIP_CHECK(hipMemcpy(e->fuse_tab.p, ftab.data(), ftab.size() * sizeof(FuseRegion), hipMemcpyHostToDevice));
            e->fuse.tab = (const FuseRegion *)e->fuse_tab.p;
            e->fuse.lds_entries = (int)entries;
            e->fuse.lds_half = (int)half;
        }
    }
    e->G = G;
    e->geom_w = w;
    e->geom_h = h;
    return ORBGPU_OK;
}

extern "C" {

const char *orbgpu_last_error(void) { return g_err; }

int orbgpu_device_count(int *count)
{
    if (!count) return ORBGPU_E_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return ORBGPU_OK;
}

int orbgpu_pinned_alloc(size_t bytes, void **out)
{
    if (!out || bytes == 0) return ORBGPU_E_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipHostMalloc of %zu bytes failed", bytes);
        return ORBGPU_E_NOMEM;
    }
    return ORBGPU_OK;
}

void orbgpu_pinned_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int orbgpu_extractor_create(const orbgpu_params *p, orbgpu_extractor **out)
{
    if (!p || !out) return ORBGPU_E_INVALID;
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > ORBGPU_MAX_LEVELS || p->nfeatures < 0 || !(p->scaleFactor > 1.0f) ||
        p->max_batch < 1 || p->max_width < 1 || p->max_height < 1) {
        set_error("invalid extractor parameters");
        return ORBGPU_E_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device available: the ORB extractor needs a GPU");
        return ORBGPU_E_HIP;
    }
    if (p->device < 0 || p->device >= ndev) {
        set_error("device %d out of range (%d devices)", p->device, ndev);
        return ORBGPU_E_INVALID;
    }
    orbgpu_extractor *e = new (std::nothrow) orbgpu_extractor();
    if (!e) return ORBGPU_E_NOMEM;
    e->p = *p;
    if (const char *ov = getenv("ORBGPU_FAST_QUEUE_CAP")) e->tune.fast_queue_cap = atoi(ov) > 0 ? atoi(ov) : 0;
    if (const char *ov = getenv("ORBGPU_SCHEDULE")) e->tune.schedule = atoi(ov) == 1 ? 1 : atoi(ov) == 2 ? 2 : 0;
    if (const char *ov = getenv("ORBGPU_OCT_THREADS")) e->tune.oct_threads = atoi(ov) == 128 ? 128 : atoi(ov) == 512 ? 512 : atoi(ov) == 256 ? 256 : 0;
    compute_tables(e);
    int rc = ORBGPU_OK;
    do {
        if (hipSetDevice(p->device) != hipSuccess) { set_error("hipSetDevice failed"); rc = ORBGPU_E_HIP; break; }
        if (hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); rc = ORBGPU_E_HIP; break; }
        e->stream = e->own_stream;
        // An ordinary stream: created with the highest stream priority (so that the quadtree's few long workgroups are
        // placed ahead of the throughput kernels), the chunked host path, which has four streams and several calls in
        // flight, stopped making progress on this runtime.
        if (hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) { set_error("aux stream/event creation failed"); rc = ORBGPU_E_HIP; break; }
        if ((rc = e->pattern.ensure(1024))) break;
        if (hipMemcpy(e->pattern.p, kPattern, 1024, hipMemcpyHostToDevice) != hipSuccess) { set_error("pattern upload failed"); rc = ORBGPU_E_HIP; break; }
        // validate and size for the largest image up front
        if ((rc = set_geometry(e, p->max_width, p->max_height))) break;
        if (hipMemset(e->status.p, 0, 64) != hipSuccess) { rc = ORBGPU_E_HIP; break; }
    } while (0);
    if (rc) {
        orbgpu_extractor_destroy(e);
        return rc;
    }
    *out = e;
    return ORBGPU_OK;
}

void orbgpu_extractor_destroy(orbgpu_extractor *e)
{
    if (!e) return;
    (void)hipSetDevice(e->p.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    drop_graph(e);
    DevBuf *bufs[] = {&e->pyr, &e->blur, &e->slots, &e->cell_counts, &e->cand, &e->nodes, &e->scratch, &e->sel, &e->sel_count,
                      &e->cand_count, &e->status, &e->pattern, &e->xtab, &e->ytab, &e->cell_tab, &e->blur_tab, &e->fuse_tab, &e->st_img, &e->st_out};
    for (DevBuf *b : bufs) b->release();
    if (e->h_pinned) (void)hipHostFree(e->h_pinned);
    for (hipEvent_t x : e->ev) (void)hipEventDestroy(x);
    for (hipEvent_t x : e->chunk_ev) (void)hipEventDestroy(x);
    delete e->pool;
    if (e->up_stream) (void)hipStreamDestroy(e->up_stream);
    if (e->down_stream) (void)hipStreamDestroy(e->down_stream);
    if (e->aux_stream) { (void)hipStreamSynchronize(e->aux_stream); (void)hipStreamDestroy(e->aux_stream); }
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
    delete e;
}

int orbgpu_extractor_set_stream(orbgpu_extractor *e, void *hip_stream)
{
    if (!e) return ORBGPU_E_INVALID;
    e->stream = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
    return ORBGPU_OK;
}

int orbgpu_extractor_sync(orbgpu_extractor *e)
{
    if (!e) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    return ORBGPU_OK;
}

int orbgpu_extractor_status(orbgpu_extractor *e, int32_t *flags)
{
    if (!e || !flags) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipSetDevice(e->p.device));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    ORBGPU_HIP_CHECK(hipMemcpy(flags, e->status_cur ? e->status_cur : (int32_t *)e->status.p, 4, hipMemcpyDeviceToHost));
    return ORBGPU_OK;
}

int orbgpu_extractor_profile(orbgpu_extractor *e, int32_t enable)
{
    if (!e) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipSetDevice(e->p.device));
    if (enable && e->ev.empty()) {
        e->ev.resize((size_t)kProfSets * kProfEvents);
        for (auto &x : e->ev) ORBGPU_HIP_CHECK(hipEventCreate(&x));
    }
    e->profile = enable != 0;
    e->serialize = enable == 2;
    e->prof_next = 0;
    e->prof_used = 0;
    return ORBGPU_OK;
}

int orbgpu_extractor_stage_times(orbgpu_extractor *e, float *ms5, int32_t *ncalls)
{
    if (!e || !ms5) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipSetDevice(e->p.device));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    // {pyramid, fast, octree, blur, describe}.  Two streams: the quadtree runs on its own stream (events 5 -> 6) beside
    // the blur on the main stream (2 -> 3), and describe's interval (3 -> 4) includes waiting for the quadtree.
    // One stream (serialized mode, one or two frames): FAST, blur (5 -> 6), quadtree (6 -> 3), describe.
    // The blur on the second stream instead: blur 5 -> 6, quadtree 2 -> 3.
    const int sch = e->last_schedule;
    const int from[5] = {0, 1, sch == 0 ? 6 : sch == 1 ? 5 : 2, sch == 1 ? 2 : 5, 3}, to[5] = {1, 2, sch == 1 ? 6 : 3, sch == 1 ? 3 : 6, 4};
    for (int i = 0; i < 5; i++) ms5[i] = 0.f;
    for (int s = 0; s < e->prof_used; s++)
        for (int i = 0; i < 5; i++) {
            float t = 0.f;
            ORBGPU_HIP_CHECK(hipEventElapsedTime(&t, e->ev[(size_t)s * kProfEvents + from[i]], e->ev[(size_t)s * kProfEvents + to[i]]));
            ms5[i] += t;
        }
    if (e->prof_used > 0)
        for (int i = 0; i < 5; i++) ms5[i] /= (float)e->prof_used;
    if (ncalls) *ncalls = e->prof_used;
    return ORBGPU_OK;
}

int orbgpu_extractor_tables(const orbgpu_extractor *e, int32_t *nlevels, float *scale, float *inv_scale, float *sigma2,
                            float *inv_sigma2, int32_t *features_per_level)
{
    if (!e) return ORBGPU_E_INVALID;
    const int L = e->p.nlevels;
    if (nlevels) *nlevels = L;
    for (int i = 0; i < L; i++) {
        if (scale) scale[i] = e->scale[i];
        if (inv_scale) inv_scale[i] = e->inv_scale[i];
        if (sigma2) sigma2[i] = e->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
        if (features_per_level) features_per_level[i] = e->quota[i];
    }
    return ORBGPU_OK;
}

int orbgpu_extractor_max_keypoints(const orbgpu_extractor *e, int32_t *max_kp)
{
    if (!e || !max_kp) return ORBGPU_E_INVALID;
    *max_kp = e->G.total_kp_cap;
    return ORBGPU_OK;
}

// Host-side copies of a batch, frame by frame: in the calling thread for a few frames, else on up to eight threads.
static void parallel_frames(int nimg, const std::function<void(int)> &fn)
{
    const int nthreads = std::min(8, nimg / 16);
    if (nthreads < 2) {
        for (int f = 0; f < nimg; f++) fn(f);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(nthreads);
    for (int t = 0; t < nthreads; t++)
        pool.emplace_back([&, t]() {
            for (int f = t; f < nimg; f += nthreads) fn(f);
        });
    for (auto &th : pool) th.join();
}

static constexpr int kGraphMaxFrames = 8;  // beyond a few frames the kernels dominate and plain launches do as well

static void drop_graph(orbgpu_extractor *e)
{
    if (e->graph_exec) (void)hipGraphExecDestroy(e->graph_exec);
    if (e->graph) (void)hipGraphDestroy(e->graph);
    e->graph_exec = nullptr;
    e->graph = nullptr;
    e->graph_nimg = 0;
}

// Enqueue the extraction of nimg device-resident frames as frames frame0 .. frame0 + nimg - 1 of the handle's working
// set (pyramids, candidate slots, quadtree storage are per frame), on the handle's stream.
static int enqueue_extract(orbgpu_extractor *e, int frame0, const uint8_t *d_imgs, int32_t nimg, int32_t stride, size_t frame_stride,
                           orbgpu_keypoint *d_kps, uint8_t *d_desc, int32_t capacity, int32_t *d_counts, bool clear_status);

int orbgpu_extract_batch_device(orbgpu_extractor *e, const uint8_t *d_imgs, int32_t nimg, int32_t w, int32_t h,
                                int32_t stride, size_t frame_stride, orbgpu_keypoint *d_kps, uint8_t *d_desc,
                                int32_t capacity, int32_t *d_counts)
{
    if (!e || !d_imgs || !d_kps || !d_desc || !d_counts || nimg < 1 || capacity < 1 || stride < w) {
        set_error("orbgpu_extract_batch_device: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (nimg > e->p.max_batch) {
        set_error("batch of %d exceeds max_batch %d", nimg, e->p.max_batch);
        return ORBGPU_E_INVALID;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(e->p.device));
    int rc = set_geometry(e, w, h);
    if (rc) return rc;
    e->status_cur = (int32_t *)e->status.p;
    return enqueue_extract(e, 0, d_imgs, nimg, stride, frame_stride, d_kps, d_desc, capacity, d_counts, true);
}

static int enqueue_extract(orbgpu_extractor *e, int frame0, const uint8_t *d_imgs, int32_t nimg, int32_t stride, size_t frame_stride,
                           orbgpu_keypoint *d_kps, uint8_t *d_desc, int32_t capacity, int32_t *d_counts, bool clear_status)
{
    int rc;
    const Geom &G = e->G;
    hipStream_t st = e->stream;
    const size_t f0 = (size_t)frame0;
    uint8_t *pyr = (uint8_t *)e->pyr.p + f0 * (size_t)G.pyr_frame_bytes, *blur = (uint8_t *)e->blur.p + f0 * (size_t)G.pyr_frame_bytes;
    uint32_t *slots = (uint32_t *)e->slots.p + f0 * (size_t)G.slot_frame_u32, *cand = (uint32_t *)e->cand.p + f0 * (size_t)G.cand_frame_u32;
    int32_t *cell_counts = (int32_t *)e->cell_counts.p + f0 * (size_t)G.total_cells;
    QNode *nodes = (QNode *)e->nodes.p + f0 * (size_t)G.node_frame;
    int32_t *scratch = (int32_t *)e->scratch.p + f0 * (size_t)G.node_frame * kScratchPerNode;
    uint32_t *sel = (uint32_t *)e->sel.p + f0 * (size_t)G.total_kp_cap;
    int32_t *sel_count = (int32_t *)e->sel_count.p + f0 * (size_t)G.nlevels, *cand_count = (int32_t *)e->cand_count.p + f0 * (size_t)G.nlevels;
    hipEvent_t *ev = nullptr;
    if (e->profile) {
        ev = &e->ev[(size_t)e->prof_next * kProfEvents];
        e->prof_next = (e->prof_next + 1) % kProfSets;
        if (e->prof_used < kProfSets) e->prof_used++;
    }
#define STAGE_MARK(i) do { if (ev) ORBGPU_HIP_CHECK(hipEventRecord(ev[i], st)); } while (0)
    // the status word describes one call alone: the first kernel clears it (in a captured sequence on every replay)
    int32_t *status = e->status_cur;
    STAGE_MARK(0);
    TraceRange tr_all("orbgpu.extract");
    {
    TraceRange tr("orbgpu.pyramid");
    if ((rc = launch_pyramid(st, d_imgs, nimg, stride, frame_stride, pyr, G, (const XTab *)e->xtab.p, (const YTab *)e->ytab.p, &e->fuse,
                             clear_status ? status : nullptr)))
        return rc;
    }
    STAGE_MARK(1);
    {
    TraceRange tr("orbgpu.fast_cells");
    if ((rc = launch_fast(st, nimg, pyr, G, (const uint32_t *)e->cell_tab.p, slots, cell_counts, e->tune))) return rc;
    }
    STAGE_MARK(2);
    // fork: the blur depends on the pyramid only, the quadtree on FAST.  The quadtree is a few long, latency-bound
    // workgroups, the blur keeps every SIMD busy: the quadtree goes first, onto a stream of its own, and the blur
    // fills the chip around it on the main stream.  For one or two frames the blur is a few
    // microseconds and handing work to a second queue and back costs more than that.
    const bool serial = e->serialize || nimg <= 2;

    // Which of the two goes to the second stream, measured both ways on one GPU: with levels of a few hundred
    // features (640x480 at 1000 features, 512 frames per launch) the quadtree enqueued first on the second stream gives
    // the step 270 k frames/s against 267 k; with levels of thousands (3840x2160 at 8000) the blur enqueued first on
    // the second stream gives 32-frame launches 12.4 k frames/s against 11.7 k, 256-frame launches 14.7 k against 14.2 k.
    int maxcap = 0;
    for (int l = 0; l < G.nlevels; l++) maxcap = G.lv[l].node_cap > maxcap ? G.lv[l].node_cap : maxcap;
    const bool blur_second = e->tune.schedule == 1 || (e->tune.schedule == 0 && maxcap > 512);
    e->last_schedule = serial ? 0 : blur_second ? 2 : 1;
    {
    TraceRange tr("orbgpu.quadtree+blur");
    if (serial && !ev && !e->serialize) {
        // one launch for both (unless the stages are being timed one by one)
        const BlurWork bw = {pyr, blur, (const uint32_t *)e->blur_tab.p, e->blur_tiles};
        bool blur_done = false;
        if ((rc = launch_octree(st, nimg, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, e->tune, &bw, &blur_done)))
            return rc;
        if (!blur_done && (rc = launch_blur(st, nimg, pyr, blur, G, (const uint32_t *)e->blur_tab.p, e->blur_tiles))) return rc;
    } else if (serial) {
        if (ev) ORBGPU_HIP_CHECK(hipEventRecord(ev[5], st));
        if ((rc = launch_blur(st, nimg, pyr, blur, G, (const uint32_t *)e->blur_tab.p, e->blur_tiles))) return rc;
        if (ev) ORBGPU_HIP_CHECK(hipEventRecord(ev[6], st));
        if ((rc = launch_octree(st, nimg, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, e->tune, nullptr, nullptr)))
            return rc;
        STAGE_MARK(3);
    } else if (blur_second) {
        // the blur on the second stream, enqueued ahead of the quadtree on the main stream (the event intervals then
        // read: "octree" = the blur's, "blur" = the quadtree's)
        hipStream_t ax = e->aux_stream;
        ORBGPU_HIP_CHECK(hipEventRecord(e->ev_fork, st));
        ORBGPU_HIP_CHECK(hipStreamWaitEvent(ax, e->ev_fork, 0));
        if (ev) ORBGPU_HIP_CHECK(hipEventRecord(ev[5], ax));
        if ((rc = launch_blur(ax, nimg, pyr, blur, G, (const uint32_t *)e->blur_tab.p, e->blur_tiles))) return rc;
        if (ev) ORBGPU_HIP_CHECK(hipEventRecord(ev[6], ax));
        ORBGPU_HIP_CHECK(hipEventRecord(e->ev_join, ax));
        if ((rc = launch_octree(st, nimg, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, e->tune, nullptr, nullptr)))
            return rc;
        STAGE_MARK(3);
        ORBGPU_HIP_CHECK(hipStreamWaitEvent(st, e->ev_join, 0));  // join
    } else {
        hipStream_t ax = e->aux_stream;
        ORBGPU_HIP_CHECK(hipEventRecord(e->ev_fork, st));
        ORBGPU_HIP_CHECK(hipStreamWaitEvent(ax, e->ev_fork, 0));
        if (ev) ORBGPU_HIP_CHECK(hipEventRecord(ev[5], ax));
        if ((rc = launch_octree(ax, nimg, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, e->tune, nullptr, nullptr)))
            return rc;
        if (ev) ORBGPU_HIP_CHECK(hipEventRecord(ev[6], ax));
        ORBGPU_HIP_CHECK(hipEventRecord(e->ev_join, ax));
        if ((rc = launch_blur(st, nimg, pyr, blur, G, (const uint32_t *)e->blur_tab.p, e->blur_tiles))) return rc;
        STAGE_MARK(3);
        ORBGPU_HIP_CHECK(hipStreamWaitEvent(st, e->ev_join, 0));  // join
    }
    }
    TraceRange tr_desc("orbgpu.describe");
    if ((rc = launch_describe(st, nimg, pyr, blur, G, sel, sel_count, (const int8_t *)e->pattern.p, d_kps, d_desc, d_counts, capacity,
                              status)))
        return rc;
    STAGE_MARK(4);
#undef STAGE_MARK
    e->last_nimg = frame0 + nimg;
    e->halo_full = false;
    return ORBGPU_OK;
}

// Is p host memory that the GPU can reach by DMA (hipHostMalloc / hipHostRegister)?  Pageable memory is unknown to
// the runtime, which reports an error for it.
static bool is_pinned_host(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    const hipError_t rc = hipPointerGetAttributes(&a, p);
    (void)hipGetLastError();
    return rc == hipSuccess && a.type == hipMemoryTypeHost;
}

static constexpr int kChunkMin = 32;  // host batches of at least this many frames are cut into chunks

// Large host batches: the frames are cut into chunks, and while the kernels of chunk k run, chunk k + 1 is uploaded
// and the results of chunk k - 1 are downloaded on two further streams; the host meanwhile stages chunk k + 2 into
// pinned memory and hands out the results of chunk k - 2.  Caller memory that is already pinned (hipHostMalloc,
// hipHostRegister) is used in place: contiguous pinned frames go to the device in one copy per chunk without
// staging, and with capacity == the handle's own bound the results land in the caller's arrays directly.
static int extract_batch_chunked(orbgpu_extractor *e, const uint8_t *const *imgs, int32_t nimg, int32_t w, int32_t h, int32_t stride,
                                 orbgpu_keypoint *kps, uint8_t *desc, int32_t capacity, int32_t *counts)
{
    const int cap = e->G.total_kp_cap;
    const size_t fbytes = (size_t)w * h;
    int chunk = std::max(kChunkMin / 2, (nimg + 5) / 6);
    const int nchunks = (nimg + chunk - 1) / chunk;
    if (!e->up_stream) {
        ORBGPU_HIP_CHECK(hipStreamCreateWithFlags(&e->up_stream, hipStreamNonBlocking));
        ORBGPU_HIP_CHECK(hipStreamCreateWithFlags(&e->down_stream, hipStreamNonBlocking));
    }
    if (!e->pool) {
        const unsigned hw = std::thread::hardware_concurrency();
        e->pool = new (std::nothrow) CopyPool((int)std::min(15u, hw > 1 ? hw - 1 : 1u));
        if (!e->pool) return ORBGPU_E_NOMEM;
    }
    while ((int)e->chunk_ev.size() < 3 * nchunks) {
        hipEvent_t x;
        ORBGPU_HIP_CHECK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
        e->chunk_ev.push_back(x);
    }
    bool contiguous = stride == w;
    for (int f = 1; f < nimg && contiguous; f++) contiguous = imgs[f] == imgs[f - 1] + fbytes;
    const bool in_pinned = contiguous && is_pinned_host(imgs[0]) && is_pinned_host(imgs[0] + fbytes * (size_t)nimg - 1);
    const bool out_direct = capacity == cap && is_pinned_host(kps) && is_pinned_host(desc);
    uint8_t *hp = (uint8_t *)e->h_pinned;
    orbgpu_keypoint *h_kps = (orbgpu_keypoint *)hp;
    uint8_t *h_desc = hp + (size_t)e->p.max_batch * cap * sizeof(orbgpu_keypoint);
    int32_t *h_counts = (int32_t *)(h_desc + (size_t)e->p.max_batch * cap * 32);
    int32_t *h_status = h_counts + e->p.max_batch;
    uint8_t *h_in = (uint8_t *)(h_status + 16);
    hipStream_t st = e->stream;
    e->status_cur = (int32_t *)e->status.p;
    ORBGPU_HIP_CHECK(hipMemsetAsync(e->status.p, 0, 4, st));
    int rc = ORBGPU_OK;
    orbgpu_keypoint *st_kps = (orbgpu_keypoint *)e->st_out.p;
    uint8_t *st_desc = (uint8_t *)(st_kps + (size_t)e->p.max_batch * cap);
    int32_t *st_counts = (int32_t *)(st_desc + (size_t)e->p.max_batch * cap * 32);

    auto stage_in = [&](int c) {  // pageable frames of chunk c -> pinned
        const int f0 = c * chunk, n = std::min(chunk, nimg - f0);
        e->pool->run(n, [&](int i) {
            const int f = f0 + i;
            uint8_t *dst = h_in + (size_t)f * fbytes;
            if (stride == w) memcpy(dst, imgs[f], fbytes);
            else for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, imgs[f] + (size_t)y * stride, (size_t)w);
        });
    };
    auto hand_out = [&](int c) -> int {  // results of chunk c (downloaded) -> the caller's arrays
        const int f0 = c * chunk, n = std::min(chunk, nimg - f0);
        ORBGPU_HIP_CHECK(hipEventSynchronize(e->chunk_ev[3 * c + 2]));
        for (int i = 0; i < n; i++) {
            const int cnt = h_counts[f0 + i];
            if (cnt > capacity) {
                set_error("frame %d has %d keypoints but capacity is %d", f0 + i, cnt, capacity);
                return ORBGPU_E_CAPACITY;
            }
            counts[f0 + i] = cnt;
        }
        if (!out_direct)
            e->pool->run(n, [&](int i) {
                const int f = f0 + i, cnt = h_counts[f];
                memcpy(kps + (size_t)f * capacity, h_kps + (size_t)f * cap, (size_t)cnt * sizeof(orbgpu_keypoint));
                memcpy(desc + (size_t)f * capacity * 32, h_desc + (size_t)f * cap * 32, (size_t)cnt * 32);
            });
        return ORBGPU_OK;
    };

    if (!in_pinned) stage_in(0);
    for (int c = 0; c < nchunks && rc == ORBGPU_OK; c++) {
        const int f0 = c * chunk, n = std::min(chunk, nimg - f0);
        hipEvent_t ev_up = e->chunk_ev[3 * c], ev_done = e->chunk_ev[3 * c + 1], ev_down = e->chunk_ev[3 * c + 2];
        uint8_t *d_img = (uint8_t *)e->st_img.p + (size_t)f0 * fbytes;
        orbgpu_keypoint *d_k = st_kps + (size_t)f0 * cap;
        uint8_t *d_d = st_desc + (size_t)f0 * cap * 32;
        int32_t *d_c = st_counts + f0;
        ORBGPU_HIP_CHECK(hipMemcpyAsync(d_img, in_pinned ? imgs[f0] : h_in + (size_t)f0 * fbytes, (size_t)n * fbytes, hipMemcpyHostToDevice,
                                        e->up_stream));
        ORBGPU_HIP_CHECK(hipEventRecord(ev_up, e->up_stream));
        ORBGPU_HIP_CHECK(hipStreamWaitEvent(st, ev_up, 0));
        if ((rc = enqueue_extract(e, f0, d_img, n, w, fbytes, d_k, d_d, cap, d_c, false))) break;
        ORBGPU_HIP_CHECK(hipEventRecord(ev_done, st));
        ORBGPU_HIP_CHECK(hipStreamWaitEvent(e->down_stream, ev_done, 0));
        ORBGPU_HIP_CHECK(hipMemcpyAsync(out_direct ? kps + (size_t)f0 * cap : h_kps + (size_t)f0 * cap, d_k, (size_t)n * cap * sizeof(orbgpu_keypoint),
                                        hipMemcpyDeviceToHost, e->down_stream));
        ORBGPU_HIP_CHECK(hipMemcpyAsync(out_direct ? desc + (size_t)f0 * cap * 32 : h_desc + (size_t)f0 * cap * 32, d_d, (size_t)n * cap * 32,
                                        hipMemcpyDeviceToHost, e->down_stream));
        ORBGPU_HIP_CHECK(hipMemcpyAsync(h_counts + f0, d_c, (size_t)n * 4, hipMemcpyDeviceToHost, e->down_stream));
        if (c == nchunks - 1) ORBGPU_HIP_CHECK(hipMemcpyAsync(h_status, e->status.p, 4, hipMemcpyDeviceToHost, e->down_stream));
        ORBGPU_HIP_CHECK(hipEventRecord(ev_down, e->down_stream));
        // host work beside the GPU: the next chunk's frames, the results before last
        if (!in_pinned && c + 1 < nchunks) stage_in(c + 1);
        if (c >= 1) rc = hand_out(c - 1);
    }
    if (rc == ORBGPU_OK) rc = hand_out(nchunks - 1);
    // everything enqueued has to finish before the handle's buffers may be reused, also after an error
    (void)hipStreamSynchronize(e->up_stream);
    (void)hipStreamSynchronize(st);
    (void)hipStreamSynchronize(e->down_stream);
    if (rc) return rc;
    if (*h_status) {
        set_error("internal capacity exceeded (status %d)", *h_status);
        return ORBGPU_E_CAPACITY;
    }
    return ORBGPU_OK;
}

}  // extern "C"

// The host-image path up to, but not including, the wait for the GPU: the images go through the handle's page-locked
// staging block to the device, the extraction is enqueued (as a HIP graph for a few frames: upload, kernels and the
// download of the output block replay as one launch) and *X says where the results are: on the device in the handle's
// output block until the handle's next call, and in its page-locked block once the stream has been waited for.
int orbgpu::extract_host_enqueue(orbgpu_extractor *e, const uint8_t *const *imgs, int32_t nimg, int32_t w, int32_t h, int32_t stride,
                                 HostExtraction *X)
{
    int rc;
    hipStream_t st = e->stream;
    const int cap = e->G.total_kp_cap;
    uint8_t *h_in = (uint8_t *)e->h_pinned + (size_t)e->p.max_batch * cap * (sizeof(orbgpu_keypoint) + 32) + (size_t)e->p.max_batch * 4 + 64;
    // the previous call's device-to-host copies have completed (every host entry point synchronises)
    // pageable -> pinned on the host (a 2-D copy from pageable memory stalls for milliseconds); a batch is copied by
    // a few threads, one host thread moves only a few GB/s
    parallel_frames(nimg, [&](int f) {
        uint8_t *hin = h_in + (size_t)f * w * h;
        if (stride == w) memcpy(hin, imgs[f], (size_t)w * h);
        else for (int y = 0; y < h; y++) memcpy(hin + (size_t)y * w, imgs[f] + (size_t)y * stride, (size_t)w);
    });
    uint8_t *hp = (uint8_t *)e->h_pinned;
    X->h_kps = (const orbgpu_keypoint *)hp;
    X->h_desc = hp + (size_t)nimg * cap * sizeof(orbgpu_keypoint);
    X->h_counts = (const int32_t *)(X->h_desc + (size_t)nimg * cap * 32);
    X->h_status = X->h_counts + nimg;
    X->cap = cap;
    X->stream = st;
    // upload, extraction, downloads: every address involved belongs to the handle, so the sequence is the same for
    // every call with this geometry and batch size
    // device block of this call, laid out like the pinned one: keypoints | descriptors | counts | status word
    orbgpu_keypoint *d_k = (orbgpu_keypoint *)e->st_out.p;
    uint8_t *d_d = (uint8_t *)(d_k + (size_t)nimg * cap);
    int32_t *d_c = (int32_t *)(d_d + (size_t)nimg * cap * 32);
    X->d_kps = d_k; X->d_desc = d_d; X->d_counts = d_c;
    e->status_cur = d_c + nimg;
    const size_t block_bytes = (size_t)nimg * cap * (sizeof(orbgpu_keypoint) + 32) + (size_t)nimg * 4 + 4;
    // One or two frames: the first kernel reads the page-locked staging block in place (it is mapped into the device's
    // address space; the kernel streams the image over PCIe as fast as a copy would, and the copy's own launch and
    // the second pass over the image are saved).  Larger batches upload first: their kernels start on resident data.
    const bool in_place = nimg <= 2;
    auto enqueue = [&]() -> int {
        if (!in_place) ORBGPU_HIP_CHECK(hipMemcpyAsync(e->st_img.p, h_in, (size_t)nimg * w * h, hipMemcpyHostToDevice, st));
        int r = enqueue_extract(e, 0, in_place ? h_in : (const uint8_t *)e->st_img.p, nimg, w, (size_t)w * h, d_k, d_d, cap, d_c, true);
        if (r) return r;
        ORBGPU_HIP_CHECK(hipMemcpyAsync(hp, d_k, block_bytes, hipMemcpyDeviceToHost, st));
        return ORBGPU_OK;
    };
    const bool want_graph = !e->profile && !e->graph_broken && nimg <= kGraphMaxFrames;
    if (want_graph && e->graph_exec && e->graph_nimg == nimg && e->graph_w == w && e->graph_h == h) {
        ORBGPU_HIP_CHECK(hipGraphLaunch(e->graph_exec, st));
        e->last_nimg = nimg;  // what orbgpu_extract_batch_device notes on the host
        e->halo_full = false;
    } else if (want_graph) {
        drop_graph(e);
        bool captured = false;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            rc = enqueue();
            hipGraph_t g = nullptr;
            const hipError_t end = hipStreamEndCapture(st, &g);
            if (rc == ORBGPU_OK && end == hipSuccess && g && hipGraphInstantiate(&e->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
                e->graph = g;
                e->graph_nimg = nimg; e->graph_w = w; e->graph_h = h;
                captured = true;
            } else if (g) {
                (void)hipGraphDestroy(g);
            }
        }
        (void)hipGetLastError();
        if (captured) {
            ORBGPU_HIP_CHECK(hipGraphLaunch(e->graph_exec, st));
        } else {
            e->graph_broken = true;
            e->graph_exec = nullptr;
            if ((rc = enqueue())) return rc;
        }
    } else {
        if ((rc = enqueue())) return rc;
    }
    return ORBGPU_OK;
}

// Argument checks and geometry of the host-image entry points; then extract_host_enqueue.
int orbgpu::extract_host_begin(orbgpu_extractor *e, const uint8_t *const *imgs, int32_t nimg, int32_t w, int32_t h, int32_t stride,
                               HostExtraction *X)
{
    if (!e || !imgs || nimg < 1 || stride < w || w < 1 || h < 1) {
        set_error("extract: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (nimg > e->p.max_batch) {
        set_error("batch of %d exceeds max_batch %d", nimg, e->p.max_batch);
        return ORBGPU_E_INVALID;
    }
    for (int f = 0; f < nimg; f++)
        if (!imgs[f]) { set_error("image %d is NULL", f); return ORBGPU_E_INVALID; }
    ORBGPU_HIP_CHECK(hipSetDevice(e->p.device));
    int rc = set_geometry(e, w, h);
    if (rc) return rc;
    return extract_host_enqueue(e, imgs, nimg, w, h, stride, X);
}

extern "C" {

int orbgpu_extract_batch(orbgpu_extractor *e, const uint8_t *const *imgs, int32_t nimg, int32_t w, int32_t h,
                         int32_t stride, orbgpu_keypoint *kps, uint8_t *desc, int32_t capacity, int32_t *counts)
{
    if (!e || !imgs || !kps || !desc || !counts || nimg < 1 || capacity < 0 || stride < w) {
        set_error("orbgpu_extract_batch: invalid argument");
        return ORBGPU_E_INVALID;
    }
    if (nimg > e->p.max_batch) {
        set_error("batch of %d exceeds max_batch %d", nimg, e->p.max_batch);
        return ORBGPU_E_INVALID;
    }
    ORBGPU_HIP_CHECK(hipSetDevice(e->p.device));
    int rc = set_geometry(e, w, h);
    if (rc) return rc;
    for (int f = 0; f < nimg; f++)
        if (!imgs[f]) { set_error("image %d is NULL", f); return ORBGPU_E_INVALID; }
    if (nimg >= kChunkMin && !e->profile) return extract_batch_chunked(e, imgs, nimg, w, h, stride, kps, desc, capacity, counts);
    HostExtraction X;
    if ((rc = extract_host_enqueue(e, imgs, nimg, w, h, stride, &X))) return rc;
    if ((rc = wait_stream_short(X.stream, nimg <= 2 ? 400 : 0))) return rc;
    if (*X.h_status) {
        set_error("internal capacity exceeded (status %d)", *X.h_status);
        return ORBGPU_E_CAPACITY;
    }
    const int cap = X.cap;
    for (int f = 0; f < nimg; f++) {
        const int n = X.h_counts[f];
        if (n > capacity) {
            set_error("frame %d has %d keypoints but capacity is %d", f, n, capacity);
            return ORBGPU_E_CAPACITY;
        }
        counts[f] = n;
    }
    parallel_frames(nimg, [&](int f) {
        const int n = X.h_counts[f];
        memcpy(kps + (size_t)f * capacity, X.h_kps + (size_t)f * cap, (size_t)n * sizeof(orbgpu_keypoint));
        memcpy(desc + (size_t)f * capacity * 32, X.h_desc + (size_t)f * cap * 32, (size_t)n * 32);
    });
    return ORBGPU_OK;
}

int orbgpu_extract(orbgpu_extractor *e, const uint8_t *img, int32_t w, int32_t h, int32_t stride, orbgpu_keypoint *kps,
                   uint8_t *desc, int32_t capacity, int32_t *n_out)
{
    if (!e || !n_out) return ORBGPU_E_INVALID;
    *n_out = 0;
    if (!img || w <= 0 || h <= 0) return ORBGPU_OK;  // empty image: silent return (:1459-1460)
    const uint8_t *one[1] = {img};
    int32_t cnt = 0;
    int rc = orbgpu_extract_batch(e, one, 1, w, h, stride, kps, desc, capacity, &cnt);
    if (rc) return rc;
    *n_out = cnt;
    return ORBGPU_OK;
}

int orbgpu_pyramid_level_device(orbgpu_extractor *e, int32_t frame, int32_t level, int32_t blurred, const uint8_t **d_ptr,
                                int32_t *pitch, int32_t *w, int32_t *h)
{
    if (!e || level < 0 || level >= e->p.nlevels || frame < 0 || frame >= e->last_nimg) {
        set_error("orbgpu_pyramid_level: no such frame/level");
        return ORBGPU_E_INVALID;
    }
    if (blurred && d_ptr) {
        set_error("the blurred levels are stored in tiles: no row-major device view (use orbgpu_pyramid_level)");
        return ORBGPU_E_UNSUPPORTED;
    }
    const LevelGeom &g = e->G.lv[level];
    const uint8_t *base = (const uint8_t *)(blurred ? e->blur.p : e->pyr.p);
    if (d_ptr) *d_ptr = base + (int64_t)frame * e->G.pyr_frame_bytes + interior_off(g);
    if (pitch) *pitch = g.pitch;
    if (w) *w = g.w;
    if (h) *h = g.h;
    return ORBGPU_OK;
}

int orbgpu_pyramid_level(orbgpu_extractor *e, int32_t frame, int32_t level, int32_t blurred, int32_t border, uint8_t *dst,
                         int32_t dst_stride, int32_t *w, int32_t *h)
{
    const uint8_t *src;
    int32_t pitch, lw, lh;
    int rc = orbgpu_pyramid_level_device(e, frame, level, 0, &src, &pitch, &lw, &lh);  // sizes; the blurred level is read below
    if (rc) return rc;
    if (border < 0 || border > kEdge || (blurred && border != 0)) {
        set_error("border must be 0..19 (0 for blurred levels)");
        return ORBGPU_E_INVALID;
    }
    if (w) *w = lw;
    if (h) *h = lh;
    if (!dst) return ORBGPU_OK;
    if (dst_stride < lw + 2 * border) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipSetDevice(e->p.device));
    if (border > 0 && !e->halo_full) {
        // the extraction fills the halo rows only (it reads no halo column); the columns on first request
        if ((rc = launch_halo_full(e->stream, e->last_nimg, (uint8_t *)e->pyr.p, e->G))) return rc;
        e->halo_full = true;
    }
    ORBGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (blurred) {
        // the blurred level is stored in tiles of 4 x 32 pixels (blur_tiled_off): fetch the level's block and put the rows together
        const LevelGeom &g = e->G.lv[level];
        std::vector<uint8_t> blk((size_t)g.pitch * (g.h + 2 * kEdge));
        ORBGPU_HIP_CHECK(hipMemcpy(blk.data(), (const uint8_t *)e->blur.p + (int64_t)frame * e->G.pyr_frame_bytes + g.plane_off, blk.size(),
                                   hipMemcpyDeviceToHost));
        for (int y = 0; y < lh; y++)
            for (int x = 0; x < lw; x++) dst[(size_t)y * dst_stride + x] = blk[(size_t)blur_tiled_off(g.pitch, x, y)];
        return ORBGPU_OK;
    }
    ORBGPU_HIP_CHECK(hipMemcpy2D(dst, dst_stride, src - (int64_t)border * pitch - border, pitch, lw + 2 * border,
                                 lh + 2 * border, hipMemcpyDeviceToHost));
    return ORBGPU_OK;
}

int orbgpu_debug_level_candidates(orbgpu_extractor *e, int32_t frame, int32_t level, int32_t *xs, int32_t *ys,
                                  int32_t *resp, int32_t capacity, int32_t *n_out)
{
    if (!e || !n_out || level < 0 || level >= e->p.nlevels || frame < 0 || frame >= e->last_nimg) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipSetDevice(e->p.device));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    const Geom &G = e->G;
    const LevelGeom &g = G.lv[level];
    const int nCells = g.nCols * g.nRows;
    std::vector<int32_t> cc(nCells);
    std::vector<uint32_t> sl((size_t)nCells * g.cell_cap);
    ORBGPU_HIP_CHECK(hipMemcpy(cc.data(), (int32_t *)e->cell_counts.p + (int64_t)frame * G.total_cells + g.cell_off,
                               (size_t)nCells * 4, hipMemcpyDeviceToHost));
    ORBGPU_HIP_CHECK(hipMemcpy(sl.data(), (uint32_t *)e->slots.p + (int64_t)frame * G.slot_frame_u32 + g.slot_off,
                               sl.size() * 4, hipMemcpyDeviceToHost));
    int n = 0;
    for (int c = 0; c < nCells; c++)
        for (int k = 0; k < cc[c]; k++) {
            const uint32_t p = sl[(size_t)c * g.cell_cap + k];
            if (n < capacity) { xs[n] = cand_x(p); ys[n] = cand_y(p); resp[n] = cand_s(p); }
            n++;
        }
    *n_out = n;
    return n > capacity ? ORBGPU_E_CAPACITY : ORBGPU_OK;
}

int orbgpu_debug_level_selected(orbgpu_extractor *e, int32_t frame, int32_t level, int32_t *xs, int32_t *ys, int32_t *resp,
                                int32_t capacity, int32_t *n_out)
{
    if (!e || !n_out || level < 0 || level >= e->p.nlevels || frame < 0 || frame >= e->last_nimg) return ORBGPU_E_INVALID;
    ORBGPU_HIP_CHECK(hipSetDevice(e->p.device));
    ORBGPU_HIP_CHECK(hipStreamSynchronize(e->stream));
    const Geom &G = e->G;
    const LevelGeom &g = G.lv[level];
    int32_t n = 0;
    ORBGPU_HIP_CHECK(hipMemcpy(&n, (int32_t *)e->sel_count.p + frame * G.nlevels + level, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> s(n > 0 ? n : 1);
    if (n > 0)
        ORBGPU_HIP_CHECK(hipMemcpy(s.data(), (uint32_t *)e->sel.p + (int64_t)frame * G.total_kp_cap + g.kp_off, (size_t)n * 4,
                                   hipMemcpyDeviceToHost));
    for (int i = 0; i < n && i < capacity; i++) { xs[i] = cand_x(s[i]); ys[i] = cand_y(s[i]); resp[i] = cand_s(s[i]); }
    *n_out = n;
    return n > capacity ? ORBGPU_E_CAPACITY : ORBGPU_OK;
}

}  // extern "C"
