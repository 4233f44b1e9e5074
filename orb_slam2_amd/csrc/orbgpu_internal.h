// Internal definitions shared by the HIP translation units of liborbgpu.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/orbgpu.h"

namespace orbgpu {

constexpr int kEdge = 19;        // EDGE_THRESHOLD: halo kept around every pyramid level
constexpr int kLeftMargin = 32;  // bytes in front of each interior row (>= kEdge, keeps rows 32 B aligned)
constexpr int kMinBorder = 16;   // EDGE_THRESHOLD - 3: origin of the FAST search area
constexpr int kMaxRoots = 128;   // initial quadtree nodes (round(width/height)): the widest strip of the smallest level
constexpr int kScratchPerNode = 6;  // int32 words of quadtree scratch per entry of a level's two node lists (see octree_level)
constexpr int kMaxTile = 66;     // largest FAST cell incl. 3 px apron (cells are < 60 px)
constexpr int kRW = 128, kRH = 32;  // output tile of one resize workgroup: a wave makes 8 rows, kRH / 8 waves per workgroup (one wave per workgroup, kRH = 8, measured equal)

// Fused pyramid (k_pyr_fused): tile of the last level owned by a workgroup, and what the host tabulates for it: for
// every tile and level the region {x0, y0, x1, y1} (inclusive) of the level that the tile depends on.
constexpr int kFW = 16, kFH = 8;
struct FuseRegion { int16_t x0, y0, x1, y1; };
struct FuseArgs {
    const FuseRegion *tab;  // [tile][nlevels], device memory; null: one launch per level
    int lds_entries;        // column / row table entries a workgroup stages (sum over levels of max(region width, height))
    int lds_half;           // bytes of the largest region a workgroup holds in LDS
};

// Horizontal / vertical coefficient tables of the fixed-point bilinear resize.
struct XTab { int32_t sx; int16_t a0, a1; };
struct YTab { int32_t sy0, sy1; int16_t b0, b1; int32_t pad; };

// Packed FAST candidate: x (13 bits) | y (13 bits) << 13 | score (6 low bits spill) ...
// Coordinates are relative to (16,16) of the level; both fit in 12 bits for
// levels up to 4111 px, the score in 8 bits.
__host__ __device__ inline uint32_t pack_cand(int x, int y, int s) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24); }
__host__ __device__ inline int cand_x(uint32_t c) { return (int)(c & 0xFFFu); }
__host__ __device__ inline int cand_y(uint32_t c) { return (int)((c >> 12) & 0xFFFu); }
__host__ __device__ inline int cand_s(uint32_t c) { return (int)(c >> 24); }

struct LevelGeom {
    int32_t w, h, pitch;     // interior size and row pitch in bytes
    int32_t nCols, nRows, wCell, hCell;
    int32_t cell_off;        // index of this level's first cell within a frame
    int32_t cell_cap;        // slot entries per cell
    int32_t cand_cap;        // candidate capacity of the level
    int32_t quota;           // mnFeaturesPerLevel
    int32_t node_cap;        // quadtree node capacity
    int32_t kp_cap, kp_off;  // retained-keypoint capacity / offset within a frame
    int32_t nIni;            // quadtree roots
    int32_t xtab_off, ytab_off;
    float hX;                // root width
    float scale;             // mvScaleFactor[level]
    int32_t patch_size;      // (int)(31*scale)
    int32_t rs_rows, rs_pitch_dw;  // resize into this level: staged source rows / dwords per row of the largest tile
    uint32_t rs_magic;       // floor(2^32 / rs_pitch_dw) + 1: dword index -> row of the staged window by one multiply
    int32_t rs_rows8;        // staged source rows of the largest 8-row tile (k_pyr_walk)
    int64_t plane_off;       // byte offset of the level's allocation within a frame's pyramid
    int64_t slot_off;        // u32 offset of the level's cell slots within a frame
    int64_t cand_off;        // u32 offset of the level's candidate arrays within a frame
    int64_t node_off;        // node offset within a frame
};

struct Geom {
    int32_t nlevels;
    int32_t total_cells;
    int32_t total_kp_cap;
    int32_t iniTh, minTh;
    int32_t gauss_variant;
    int32_t nimg;            // frames of the current launch (set by the launchers)
    int64_t pyr_frame_bytes;
    int64_t slot_frame_u32;
    int64_t cand_frame_u32;
    int64_t node_frame;
    // per-level tables of the fields a per-lane level lookup needs (contiguous: one scalar load each)
    int32_t t_kp_off[ORBGPU_MAX_LEVELS], t_pitch[ORBGPU_MAX_LEVELS], t_interior[ORBGPU_MAX_LEVELS], t_patch[ORBGPU_MAX_LEVELS];
    float t_scale[ORBGPU_MAX_LEVELS];
    LevelGeom lv[ORBGPU_MAX_LEVELS];
};

// Quadtree node. Points of a node are the packed candidates at
// buf[start .. start+count) of candidate array `buf` (0 or 1).
struct QNode {
    int16_t x0, x1, y0, y1;
    int32_t start;
    int32_t count;  // bit 31: buffer select
};

__host__ __device__ inline int64_t interior_off(const LevelGeom &g) { return g.plane_off + (int64_t)kEdge * g.pitch + kLeftMargin; }

// The blurred levels are kept in tiles of 4 x 32 pixels = one 128-byte line each (a pixel column group of four next
// to the same group of the next row), so that the 37 x 37 patch of a descriptor touches about 21 lines instead of 48:
// the descriptor kernel is bound by the lines it pulls into L1.  A level's tiles fill the block its row-major plane
// would (ceil32(h) <= h + 2 * kEdge rows): byte offset of interior pixel (x, y) from the level's plane_off; the
// pitch is a multiple of 64.  Eight rows of a column group that start at a multiple of 8 are 32 contiguous bytes.
__host__ __device__ inline int32_t blur_tiled_off(int32_t pitch, int32_t x, int32_t y)
{
    const int32_t X = x + kLeftMargin;
    return ((((y >> 5) * (pitch >> 2) + (X >> 2)) << 7) + ((y & 31) << 2)) + (X & 3);
}

void set_error(const char *fmt, ...);

// Test / measurement hooks of an extractor handle, read from the environment once, when the handle is created
// (never on a launch path): ORBGPU_FAST_QUEUE_CAP caps the FAST kernel's survivor queue (0: by geometry),
// ORBGPU_OCT_THREADS = 128 / 256 / 512 forces the batched quadtree's workgroup size (0: by geometry).
// ORBGPU_SCHEDULE = 1 / 2 forces the blur / the quadtree onto the second stream (0: by geometry).
// The blur of a launch, for the quadtree launcher to take along (one or two frames).
struct BlurWork {
    const uint8_t *pyr;
    uint8_t *blur;
    const uint32_t *tile_tab;
    int ntiles;
};

struct LaunchTuning { int fast_queue_cap = 0, oct_threads = 0, schedule = 0; };

// Where the host-image path of an extractor handle leaves its results (extractor_api.hip).
struct HostExtraction {
    orbgpu_keypoint *d_kps;   // device, frame f at + f * cap; valid until the handle's next call
    uint8_t *d_desc;
    int32_t *d_counts;
    const orbgpu_keypoint *h_kps;  // the handle's page-locked block, valid once `stream` has been waited for
    const uint8_t *h_desc;
    const int32_t *h_counts, *h_status;
    int cap;                  // entries per frame in all of the above
    hipStream_t stream;
};
int extract_host_enqueue(orbgpu_extractor *e, const uint8_t *const *imgs, int32_t nimg, int32_t w, int32_t h, int32_t stride,
                         HostExtraction *X);
int extract_host_begin(orbgpu_extractor *e, const uint8_t *const *imgs, int32_t nimg, int32_t w, int32_t h, int32_t stride,
                       HostExtraction *X);

// Named ranges for tracing tools (rocprofv3 --marker-trace): with ORBGPU_ROCTX=1 in the environment when the library
// first needs them, roctxRangePushA / roctxRangePop are looked up in librocprofiler-sdk-roctx.so (or libroctx64.so)
// and every stage's enqueue is bracketed; otherwise a range is two predictable branches.  The ranges mark where the
// host enqueues a stage (the kernels run later on the stream; the tools correlate them by their launch calls).
struct TraceRange {
    explicit TraceRange(const char *name);
    ~TraceRange();
    TraceRange(const TraceRange &) = delete;
    TraceRange &operator=(const TraceRange &) = delete;
private:
    bool on_;
};

// Wait for a stream whose work is a single short call (a frame or two: ~0.1 ms).  The blocking wait returns 10-15 us
// after the GPU is done; asking for the stream's state in a loop returns within a microsecond or two.  After spin_us
// the call falls back to the blocking wait, so a long batch does not burn a core.
int wait_stream_short(hipStream_t st, int spin_us);

// Grids are (blocks, frames).  Workgroups are dealt to the 8 XCDs round-robin in
// linear order; for batches of 8 or more frames the grid is (8 * blocks, frame
// groups of 8) instead, so that all blocks of a frame run on one XCD and share
// its L2 (frame % 8 picks the XCD) and no kernel divides to find its frame.
// Workgroups of the frames that pad the last group get false.  Affects speed only.
inline dim3 frame_grid(int blocks, int nimg) { return nimg < 8 ? dim3(blocks, nimg) : dim3(8 * blocks, (nimg + 7) / 8); }

#ifdef __HIPCC__
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ bool frame_block(int nimg, int *f, int *b)
{
    if (nimg < 8) {
        *f = blockIdx.y;
        *b = blockIdx.x;
        return true;
    }
    // workgroups go to the XCDs round-robin by linear index and gridDim.x is a multiple of 8: the low three bits of
    // blockIdx.x choose the XCD, i.e. the frame within the group of eight
    *f = (int)(blockIdx.y * 8 + (blockIdx.x & 7));
    *b = (int)(blockIdx.x >> 3);
    return *f < nimg;
}

// Inclusive prefix sum over the 64 lanes of a wave (all lanes active) with data-parallel-primitive moves: shifts by
// 1, 2, 4, 8 inside every row of 16 lanes, then lane 15 of a row added to the next row (rows 1 and 3) and lane 31 to
// rows 2 and 3.  Six adds with a few wait states each; the same through lane permutes goes through the LDS crossbar
// six times, about a hundred cycles apiece.
__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// Sum and unsigned minimum over the 64 lanes of a wave (all lanes active), the result wave-uniform: the scan above
// (and its twin with min, whose shifted-in and masked-out lanes read the identity) ends in lane 63.
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_scan_incl(v), 63); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    const int id = -1;  // 0xFFFFFFFF
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x118, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x142, 0xa, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Exclusive scan over the block (blockDim.x a multiple of 64, <= 1024).
// lds needs 33 ints.  Every thread must call it.
__device__ __forceinline__ int block_scan_excl(int v, int *total, int *lds)
{
    const int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int inc = wave_scan_incl(v);
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        const int wv = lane < nw ? lds[lane] : 0;
        const int winc = wave_scan_incl(wv);  // at most 16 waves: the first row's part of the scan
        if (lane < nw) lds[16 + lane] = winc - wv;
        if (lane == nw - 1) lds[32] = winc;
    }
    __syncthreads();
    const int res = inc - v + lds[16 + wid];
    *total = lds[32];
    __syncthreads();
    return res;
}

// The same scan with one barrier.  Every wave adds up the wave totals itself (at most 16 values: one more
// data-parallel scan) instead of waiting for wave 0 to do it, and consecutive calls alternate between two halves of
// lds (64 ints: 32 apart), so no barrier is needed before the next call writes: a wave can only be two calls ahead of another
// after that other has passed the barrier of the call in between, i.e. has finished reading this half.  `par` is
// the caller's call parity (any workgroup-uniform 0 / 1 variable that only these calls change).
__device__ __forceinline__ int block_scan1(int v, int *total, int *lds, int &par)
{
    const int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int *half = lds + (par << 5);
    par ^= 1;
    const int inc = wave_scan_incl(v);
    if (lane == 63) half[wid] = inc;
    __syncthreads();
    const int wv = lane < nw ? half[lane] : 0;
    const int winc = wave_scan_incl(wv);
    *total = __builtin_amdgcn_readlane(winc, 63);
    return inc - v + __shfl(winc - wv, wid);
}

#endif

}  // namespace orbgpu

#define ORBGPU_HIP_CHECK(expr)                                                             \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            orbgpu::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return ORBGPU_E_HIP;                                                           \
        }                                                                                  \
    } while (0)
