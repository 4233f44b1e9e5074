// HIP kernels of the ORB extractor for gfx950 (wave64).
//
// Stage order for a batch of frames (one launch per stage covers every frame;
// the frame index is a grid dimension):
//   pyr_copy0 -> pyr_resize (levels 1..L-1, dependent chain) -> fast_cells ->
//   octree -> blur -> describe
// Reference lines are relative to /root/reference/src/ORB_SLAM2/.
#include <atomic>
#include <vector>

#include <stdlib.h>

#include "orbgpu_internal.h"

namespace orbgpu {

// ---------------------------------------------------------------------------
// Pyramid (ORBextractor::ComputePyramid, src/ORBextractor.cc:1528-1592)
// ---------------------------------------------------------------------------
// Store pixels x0 .. x0+n-1 of one row (packed little-endian in v[], NW words).
// A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits until the wave's global stores have
// been acknowledged; where the data the other waves need is in LDS and the global stores are results nobody in the
// kernel reads back, that wait (about a microsecond per barrier for a lone workgroup) buys nothing.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Bits 32 .. 47 of the product of two unsigned 24-bit values, a full-rate instruction (the 32-bit multiply-high is a
// quarter-rate one).  The vertical pass of the resize needs (b * h) >> 16 with the weight b <= 2^11 and the
// horizontal sum h < 2^15 kept with its four fraction bits cleared (x & ~15 = h << 4): (b << 12) * (x & ~15) =
// b * h * 2^16, whose bits 32 .. 47 are (b * h) >> 16.
__device__ __forceinline__ uint32_t mulhi_u24(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int NW>
__device__ __forceinline__ void store_row_span(uint8_t *row, int x0, int n, const uint32_t (&v)[NW])
{
    if (n == NW * 4) {
        if (NW == 4) *reinterpret_cast<uint4 *>(row + x0) = make_uint4(v[0], v[NW > 1 ? 1 : 0], v[NW > 2 ? 2 : 0], v[NW > 3 ? 3 : 0]);
        else *reinterpret_cast<uint32_t *>(row + x0) = v[0];
    } else {
#pragma unroll
        for (int k = 0; k < NW * 4; k++)
            if (k < n) row[x0 + k] = (uint8_t)(v[k >> 2] >> (8 * (k & 3)));
    }
}

// A span of a level plus its reflect-101 image in the halo rows above/below
// (copyMakeBorder BORDER_REFLECT_101, src/ORBextractor.cc:1567-1586); the halo
// columns are filled by k_halo_sides when a caller asks for them.  Needs h >= 2*kEdge+2.
template <int NW>
__device__ __forceinline__ void store_span_with_halo(uint8_t *in, int pitch, int w, int h, int x0, int y, int n,
                                                     const uint32_t (&v)[NW])
{
    store_row_span<NW>(in + (int64_t)y * pitch, x0, n, v);
    const int ym = (y >= 1 && y <= kEdge) ? -y : ((y >= h - 1 - kEdge && y <= h - 2) ? 2 * (h - 1) - y : y);
    if (ym != y) store_row_span<NW>(in + (int64_t)ym * pitch, x0, n, v);
}

// Left/right halo columns of every level and every row (halo rows included),
// one thread per pixel, all levels in one launch.  The extraction itself reads no halo column (the blur forms
// the three reflected columns it needs in registers): this runs when a caller asks for a level with its border.
__global__ __launch_bounds__(256) void k_halo_sides(uint8_t *__restrict__ pyr, Geom G)
{
    int f, b;  // a workgroup = 16 rows (38 columns padded to 64 lanes) of one level
    if (!frame_block(G.nimg, &f, &b)) return;
    int l = 0;
    while (true) {
        const int n = (G.lv[l].h + 2 * kEdge + 15) >> 4;
        if (b < n) break;
        b -= n;
        if (++l >= G.nlevels) return;
    }
    const LevelGeom &g = G.lv[l];
    const int c = threadIdx.x & 63, y0 = b * 16 + (threadIdx.x >> 6) * 4 - kEdge;
    if (c >= 2 * kEdge) return;
    const int x = c < kEdge ? c - kEdge : g.w + (c - kEdge);
    const int sx = x < 0 ? -x : 2 * (g.w - 1) - x;
    uint8_t *row = pyr + (int64_t)f * G.pyr_frame_bytes + interior_off(g) + (int64_t)y0 * g.pitch;
    uint8_t v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = (y0 + r < g.h + kEdge) ? row[(int64_t)r * g.pitch + sx] : 0;
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (y0 + r < g.h + kEdge) row[(int64_t)r * g.pitch + x] = v[r];
}

// Level 0 = the input image; 16 pixels per thread, 128-bit accesses when the
// source row is 16-byte aligned.  Threads are numbered over (row, 16-pixel group).
// As the first kernel of an extraction it also clears the call's status word (status_clear, may be null): a memset
// node of its own adds 2-3 us to a single-frame call.
__global__ __launch_bounds__(256) void k_pyr_copy0(const uint8_t *__restrict__ img, int stride, size_t frame_stride,
                                                   uint8_t *__restrict__ pyr, Geom G, int32_t *status_clear)
{
    if (status_clear && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *status_clear = 0;
    const LevelGeom &g = G.lv[0];
    int f, b;
    if (!frame_block(G.nimg, &f, &b)) return;
    const int groups = (g.w + 15) >> 4;
    const int t = b * 256 + threadIdx.x;
    const int y = t / groups, x = (t - y * groups) * 16;
    if (y >= g.h) return;
    const uint8_t *src = img + (size_t)f * frame_stride + (size_t)y * stride + x;
    const int n = min(16, g.w - x);
    uint32_t v[4] = {0, 0, 0, 0};
    if (n == 16 && ((uintptr_t)src & 15) == 0) {
        const uint4 q = *reinterpret_cast<const uint4 *>(src);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k < n) v[k >> 2] |= (uint32_t)src[k] << (8 * (k & 3));
    }
    store_span_with_halo<4>(pyr + (int64_t)f * G.pyr_frame_bytes + interior_off(g), g.pitch, g.w, g.h, x, y, n, v);
}

// cv::resize INTER_LINEAR 8UC1 fixed-point arithmetic (called at :1550) from
// level l-1 into level l.  A workgroup produces a kRW x kRH output tile: the
// source window of the tile is copied into LDS by direct global -> LDS loads (no
// registers; a wave instruction fills 64 consecutive dwords of the window, rows of
// rs_pitch_dw dwords), then a thread produces a 2 x 8 block of output pixels from LDS bytes.
typedef __attribute__((address_space(3))) void rs_lvoid;
__global__ __launch_bounds__(kRH / 8 * 64) void k_pyr_resize(uint8_t *__restrict__ pyr, Geom G, int level,
                                                    const XTab *__restrict__ xtab, const YTab *__restrict__ ytab)
{
    extern __shared__ __align__(16) uint32_t rs_src[];
    const LevelGeom &gd = G.lv[level];
    const LevelGeom &gs = G.lv[level - 1];
    const int tilesX = (gd.w + kRW - 1) / kRW;
    int f, b;
    if (!frame_block(G.nimg, &f, &b)) return;
    const int tyi = b / tilesX, ty0 = tyi * kRH, tx0 = (b - tyi * tilesX) * kRW;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const XTab *xt = xtab + gd.xtab_off;
    const YTab *yt = ytab + gd.ytab_off;
    // source window: columns sx_al .. (last tap + 1), rows sy_min .. sy1 of the last output row
    const int sx_al = xt[tx0].sx & ~3;
    const int ndw = ((xt[min(tx0 + kRW - 1, gd.w - 1)].sx + 1 - sx_al) >> 2) + 1;
    const int sy_min = yt[ty0].sy0;
    const int nrows = yt[min(ty0 + kRH - 1, gd.h - 1)].sy1 - sy_min + 1;
    const int P = gd.rs_pitch_dw;
    uint8_t *base = pyr + (int64_t)f * G.pyr_frame_bytes;
    const uint8_t *S = base + interior_off(gs) + (int64_t)sy_min * gs.pitch + sx_al;  // uniform, dword-aligned
    {
        // dword d of the window = (row d / P, column d % P); columns past the tile's last dword and rows past its last
        // row re-read the last one (the buffer is padded to whole 64-dword pieces)
        const int npieces = (nrows * P + 63) >> 6;
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(rs_lvoid *)rs_src);
        const uint32_t magic = (uint32_t)gd.rs_magic;  // floor(2^32 / P) + 1: exact quotients for d * P < 2^32
        for (int p = wv; p < npieces; p += kRH / 8) {
            const uint32_t d = ((uint32_t)p << 6) + (uint32_t)lane;
            const uint32_t r = __umulhi(d, magic), c = d - r * (uint32_t)P;
            const uint32_t off = min(r, (uint32_t)(nrows - 1)) * (uint32_t)gs.pitch + 4u * min(c, (uint32_t)(ndw - 1));
            const uint32_t dst = lds0 + ((uint32_t)p << 8);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(off), "s"(S), "s"(dst)
                         : "memory");
        }
    }
    // A wave produces 8 output rows x 128 columns, a lane 2 columns of each row.  The rows are uniform
    // over the wave, so the vertical tap table is scalar data and the horizontal sums of a source row,
    // shared by consecutive output rows, are computed once and kept (rows A and B below).  The tables are
    // fetched while the window is on its way.
    const int dx0 = tx0 + 2 * lane, dy0 = ty0 + 8 * wv;
    const bool live = dy0 < gd.h;  // uniform
    XTab t0 = {0, 0, 0}, t1 = {0, 0, 0};
    YTab ytv[8];  // all eight (uniform) loads in flight at once
    if (live) {
        t0 = xt[min(dx0, gd.w - 1)];
        t1 = xt[min(dx0 + 1, gd.w - 1)];
#pragma unroll
        for (int r = 0; r < 8; r++) ytv[r] = yt[min(dy0 + r, gd.h - 1)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!live) return;
    // the right tap is the next byte: its weight is 0 where the reference clamps it
    const int xo0 = t0.sx - sx_al, xo1 = t1.sx - sx_al;
    int xr0 = xo0 + 1, xr1 = xo1 + 1;  // kept opaque: two byte reads, not one 16-bit read at an odd LDS address
    asm("" : "+v"(xr0), "+v"(xr1));
    const int a00 = t0.a0, a01 = t0.a1, a10 = t1.a0, a11 = t1.a1;
    const uint8_t *L = reinterpret_cast<const uint8_t *>(rs_src);
    uint8_t *D = base + interior_off(gd);  // uniform: row bases stay scalar, the lane adds its 32-bit column offset
    const uint32_t dcol = (uint32_t)dx0;
    const int n = gd.w - dx0;  // columns of this lane inside the level: <= 0 none, 1 the first only
    int hA0 = 0, hA1 = 0, hB0 = 0, hB1 = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int y = dy0 + r;
        if (y >= gd.h) break;
        const YTab yv = ytv[r];
        // the upper source row is usually the previous output row's lower one (the host marks it): keep its sums
        if (r > 0 && (yv.pad & 0x10000)) {
            hA0 = hB0; hA1 = hB1;
        } else {
            const uint8_t *R = L + (yv.sy0 - sy_min) * P * 4;
            hA0 = (R[xo0] * a00 + R[xr0] * a01) & ~15;
            hA1 = (R[xo1] * a10 + R[xr1] * a11) & ~15;
        }
        {
            const uint8_t *R = L + (yv.sy1 - sy_min) * P * 4;
            hB0 = (R[xo0] * a00 + R[xr0] * a01) & ~15;
            hB1 = (R[xo1] * a10 + R[xr1] * a11) & ~15;
        }
        // (b * h) >> 16 by the 24-bit multiply-high (mulhi_u24): the weights are 0 .. 2048 and uniform over the wave, the
        // horizontal sums are kept as h << 4
        const uint32_t B0 = (uint32_t)yv.b0 << 12, B1 = (uint32_t)yv.b1 << 12;
        const int p0 = (int)(mulhi_u24(B0, (uint32_t)hA0) + mulhi_u24(B1, (uint32_t)hB0) + 2u) >> 2;
        const int p1 = (int)(mulhi_u24(B0, (uint32_t)hA1) + mulhi_u24(B1, (uint32_t)hB1) + 2u) >> 2;
        // the row and, for rows next to the top / bottom edge, its reflect-101 image in the halo (tabulated)
        const int ym = (yv.pad & 0xFFFF) - 32;
        if (n >= 1) {
            // an odd width's last lane also writes the first halo column (a copy of the last pixel): nothing in the
            // extraction reads it, and k_halo_sides overwrites it when a caller asks for the border
            const uint16_t v = (uint16_t)(p0 | (p1 << 8));
            *reinterpret_cast<uint16_t *>(D + y * gd.pitch + dcol) = v;
            if (ym != y) *reinterpret_cast<uint16_t *>(D + ym * gd.pitch + dcol) = v;
        }
    }
}

// ---------------------------------------------------------------------------
// Batches: all levels above 0 of a frame by ONE workgroup, one launch for the whole batch.  A launch per level
// (k_pyr_resize) makes a batch wait seven times for its slowest workgroup, and the small levels neither fill the
// chip nor amortise their launch.  Here a workgroup of kWalkWaves waves walks its frame's levels: a wave takes
// 128 x 8 output tiles of the level (its own LDS window, filled by direct global -> LDS loads and read by nobody
// else, so tiles need no barrier), and the workgroup meets once per level, when the level it has just written (its
// stores acknowledged, visible to the waves of its CU) becomes the next one's source.  The arithmetic is that of
// k_pyr_resize.  The other workgroups of the CU fill the waits.
// ---------------------------------------------------------------------------
constexpr int kWalkWaves = 16;
__global__ __launch_bounds__(kWalkWaves * 64) void k_pyr_walk(uint8_t *__restrict__ pyr, Geom G, const XTab *__restrict__ xtab,
                                                              const YTab *__restrict__ ytab, int lds_wave_dw)
{
    extern __shared__ __align__(16) uint32_t walk_lds[];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    uint8_t *base = pyr + (int64_t)f * G.pyr_frame_bytes;
    uint32_t *mine = walk_lds + (size_t)wv * lds_wave_dw;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(rs_lvoid *)mine);
    const uint8_t *L = reinterpret_cast<const uint8_t *>(mine);
#pragma unroll 1
    for (int level = 1; level < G.nlevels; level++) {
        const LevelGeom &gd = G.lv[level];
        const LevelGeom &gs = G.lv[level - 1];
        const XTab *xt = xtab + gd.xtab_off;
        const YTab *yt = ytab + gd.ytab_off;
        const int tilesX = (gd.w + kRW - 1) / kRW, ntiles = tilesX * ((gd.h + 7) >> 3);
        // the window's rows are staged 16 bytes per lane: PQ quads per row (the last one may run up to 12 bytes past the
        // row's last tap: inside the plane's right margin)
        const int PQ = (gd.rs_pitch_dw + 3) >> 2, P = PQ * 4;
        const uint32_t magic = 0xFFFFFFFFu / (uint32_t)PQ + 1u;
        uint8_t *D = base + interior_off(gd);
#pragma unroll 1
        for (int t = wv; t < ntiles; t += kWalkWaves) {
            const int tyi = t / tilesX, dy0 = tyi * 8, tx0 = (t - tyi * tilesX) * kRW;
            // source window of the tile: columns sx_al .. (last tap + 1), rows sy_min .. sy1 of the tile's last row
            const int sx_al = xt[tx0].sx & ~3;
            const int ndw = ((xt[min(tx0 + kRW - 1, gd.w - 1)].sx + 1 - sx_al) >> 2) + 1;
            const int sy_min = yt[dy0].sy0;
            const int nrows = yt[min(dy0 + 7, gd.h - 1)].sy1 - sy_min + 1;
            const uint8_t *S = base + interior_off(gs) + (int64_t)sy_min * gs.pitch + sx_al;  // uniform, dword-aligned
            const int nq = (ndw + 3) >> 2;
            const int npieces = (nrows * PQ + 63) >> 6;
            // the previous tile's LDS reads have returned before the copy engine overwrites the window
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (int p = 0; p < npieces; p++) {
                const uint32_t d = ((uint32_t)p << 6) + (uint32_t)lane;
                const uint32_t r = __umulhi(d, magic), c = d - r * (uint32_t)PQ;
                const uint32_t off = min(r, (uint32_t)(nrows - 1)) * (uint32_t)gs.pitch + 16u * min(c, (uint32_t)(nq - 1));
                const uint32_t dst = lds0 + ((uint32_t)p << 10);
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep)
                             : "v"(off), "s"(S), "s"(dst)
                             : "memory");
            }
            const int dx0 = tx0 + 2 * lane;
            const XTab t0 = xt[min(dx0, gd.w - 1)], t1 = xt[min(dx0 + 1, gd.w - 1)];
            YTab ytv[8];  // uniform loads, all in flight at once, fetched while the window is on its way
#pragma unroll
            for (int r = 0; r < 8; r++) ytv[r] = yt[min(dy0 + r, gd.h - 1)];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the right tap is the next byte: its weight is 0 where the reference clamps it
            const int xo0 = t0.sx - sx_al, xo1 = t1.sx - sx_al;
            int xr0 = xo0 + 1, xr1 = xo1 + 1;  // kept opaque: two byte reads, not one 16-bit read at an odd LDS address
            asm("" : "+v"(xr0), "+v"(xr1));
            const int a00 = t0.a0, a01 = t0.a1, a10 = t1.a0, a11 = t1.a1;
            const uint32_t dcol = (uint32_t)dx0;
            const int n = gd.w - dx0;  // columns of this lane inside the level: <= 0 none, 1 the first only
            int hA0 = 0, hA1 = 0, hB0 = 0, hB1 = 0;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int y = dy0 + r;
                if (y >= gd.h) break;
                const YTab yv = ytv[r];
                if (r > 0 && (yv.pad & 0x10000)) {
                    hA0 = hB0; hA1 = hB1;
                } else {
                    const uint8_t *R = L + (yv.sy0 - sy_min) * P * 4;
                    hA0 = (R[xo0] * a00 + R[xr0] * a01) & ~15;
                    hA1 = (R[xo1] * a10 + R[xr1] * a11) & ~15;
                }
                {
                    const uint8_t *R = L + (yv.sy1 - sy_min) * P * 4;
                    hB0 = (R[xo0] * a00 + R[xr0] * a01) & ~15;
                    hB1 = (R[xo1] * a10 + R[xr1] * a11) & ~15;
                }
                const uint32_t B0 = (uint32_t)yv.b0 << 12, B1 = (uint32_t)yv.b1 << 12;  // (b * h) >> 16: see mulhi_u24
                const int p0 = (int)(mulhi_u24(B0, (uint32_t)hA0) + mulhi_u24(B1, (uint32_t)hB0) + 2u) >> 2;
                const int p1 = (int)(mulhi_u24(B0, (uint32_t)hA1) + mulhi_u24(B1, (uint32_t)hB1) + 2u) >> 2;
                const int ym = (yv.pad & 0xFFFF) - 32;
                if (n >= 1) {
                    const uint16_t v = (uint16_t)(p0 | (p1 << 8));
                    *reinterpret_cast<uint16_t *>(D + y * gd.pitch + dcol) = v;
                    if (ym != y) *reinterpret_cast<uint16_t *>(D + ym * gd.pitch + dcol) = v;
                }
            }
        }
        // the level is complete and visible to this workgroup's waves before any of them reads it as a source
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Levels l_first .. l_last in ONE launch, for a single frame or a few: a frame's pyramid is a chain of dependent
// resizes, and with a handful of small launches the chain costs more in launch-to-launch gaps than in arithmetic.
// A workgroup owns a kFW x kFH tile of the LAST level and works backwards to the region of every earlier level
// that the tile depends on (the taps of the taps ...), then produces those regions level by level in LDS, each from
// the one before; the first from level l_first - 1 in global memory.  Every region is also written to the level's
// plane.  Regions of neighbouring tiles overlap: the overlap is computed twice, by the same integer arithmetic, and
// stored twice with equal bytes.  Together the regions cover every level: a level's columns (rows) are all tapped by
// the next level's while the scale step is at most 2, the regions of edge tiles are extended to the level's edge,
// and the launcher falls back to one launch per level for other geometries.
// ---------------------------------------------------------------------------
// LDS: column entries {tap, next tap (both relative to the source region), a0, a1} and row entries {upper / lower
// source row relative to the source region, b0, b1, row of the reflect-101 image, row} of every level's region, then
// two region buffers used alternately; the first holds the staged region of level l_first - 1.  The regions come
// from the host's table (extractor_api.hip, set_geometry).
__global__ __launch_bounds__(1024) void k_pyr_fused(uint8_t *__restrict__ pyr, Geom G, int l_first, int l_last, const XTab *__restrict__ xtab,
                                                    const YTab *__restrict__ ytab, const FuseRegion *__restrict__ fuse_tab, int lds_entries,
                                                    int lds_half)
{
    extern __shared__ __align__(16) uint8_t fuse_lds[];
    uint2 *xe = reinterpret_cast<uint2 *>(fuse_lds);                                  // [lds_entries]
    uint4 *ye = reinterpret_cast<uint4 *>(fuse_lds + (size_t)lds_entries * 8);          // [lds_entries]
    uint8_t *bufs = fuse_lds + (size_t)lds_entries * 24;                                // 2 x lds_half
    __shared__ FuseRegion R[ORBGPU_MAX_LEVELS];
    __shared__ int eoff[ORBGPU_MAX_LEVELS];        // first column / row entry of a level (rows follow the same numbering in ye)
    __shared__ int tfirst[ORBGPU_MAX_LEVELS + 1];  // running count of table entries (columns + rows) before a level
    const int tid = threadIdx.x, nt = blockDim.x, f = blockIdx.y;
    if (tid < G.nlevels) R[tid] = fuse_tab[(size_t)blockIdx.x * G.nlevels + tid];
    __syncthreads();
    if (tid == 0) {
        int ne = 0, nte = 0;
        for (int l = l_last; l >= l_first; l--) {
            const FuseRegion r = R[l];
            eoff[l] = ne;
            tfirst[l] = nte;
            ne += max(r.x1 - r.x0 + 1, r.y1 - r.y0 + 1);
            nte += (r.x1 - r.x0 + 1) + (r.y1 - r.y0 + 1);
        }
        tfirst[l_first - 1] = nte;  // total
    }
    __syncthreads();
    uint8_t *base = pyr + (int64_t)f * G.pyr_frame_bytes;
    // stage the tables of every level and the region of the source level: all loads first, then the LDS writes, so
    // that the workgroup (the only one on its CU) waits for global memory once
    {
        constexpr int kTE = 1, kSD = 2;  // table entries / source dwords per thread and batch
        const int T = tfirst[l_first - 1];
        const LevelGeom &gs0 = G.lv[l_first - 1];
        const FuseRegion q0 = R[l_first - 1];
        const int qdw = (q0.x1 - q0.x0 + 1) >> 2, nsd = qdw * (q0.y1 - q0.y0 + 1);
        const uint8_t *S0 = base + interior_off(gs0) + (int64_t)q0.y0 * gs0.pitch + q0.x0;
        const float inv_qdw = __fdiv_rn(1.0f, (float)qdw);
        for (int t0 = 0, s0 = 0; t0 < T || s0 < nsd; t0 += kTE * nt, s0 += kSD * nt) {
            XTab tx_[kTE];
            YTab ty_[kTE];
            int lv_[kTE], k_[kTE];
            uint32_t sd_[kSD];
#pragma unroll
            for (int u = 0; u < kTE; u++) {
                const int e = t0 + u * nt + tid;
                lv_[u] = -1;
                if (e < T) {
                    int l = l_last;
                    while (l > l_first && e >= tfirst[l - 1]) l--;  // levels are numbered downwards in tfirst
                    const FuseRegion r = R[l];
                    const int k = e - tfirst[l], rw = r.x1 - r.x0 + 1;
                    lv_[u] = l; k_[u] = k;
                    if (k < rw) tx_[u] = xtab[G.lv[l].xtab_off + r.x0 + k];
                    else ty_[u] = ytab[G.lv[l].ytab_off + r.y0 + k - rw];
                }
            }
#pragma unroll
            for (int u = 0; u < kSD; u++) {
                const int i = s0 + u * nt + tid;
                if (i < nsd) {
                    const int ry = (int)__fmul_rn((float)i + 0.5f, inv_qdw), rx = i - ry * qdw;  // exact: i < 2^16
                    sd_[u] = *reinterpret_cast<const uint32_t *>(S0 + (int64_t)ry * gs0.pitch + 4 * rx);
                }
            }
#pragma unroll
            for (int u = 0; u < kTE; u++) {
                const int l = lv_[u];
                if (l < 0) continue;
                const FuseRegion r = R[l], q = R[l - 1];
                const int rw = r.x1 - r.x0 + 1, k = k_[u];
                if (k < rw) {
                    const XTab t = tx_[u];
                    // the right tap is the next byte: its weight is 0 where the reference clamps it
                    xe[eoff[l] + k] = make_uint2((uint32_t)(t.sx - q.x0) | ((uint32_t)(min(t.sx + 1, G.lv[l - 1].w - 1) - q.x0) << 16),
                                                 (uint32_t)(uint16_t)t.a0 | ((uint32_t)(uint16_t)t.a1 << 16));
                } else {
                    const YTab v = ty_[u];
                    ye[eoff[l] + k - rw] = make_uint4((uint32_t)(v.sy0 - q.y0) | ((uint32_t)(v.sy1 - q.y0) << 16),
                                                      (uint32_t)(uint16_t)v.b0 | ((uint32_t)(uint16_t)v.b1 << 16),
                                                      (uint32_t)((v.pad & 0xFFFF) - 32), (uint32_t)(r.y0 + k - rw));
                }
            }
#pragma unroll
            for (int u = 0; u < kSD; u++) {
                const int i = s0 + u * nt + tid;
                if (i < nsd) reinterpret_cast<uint32_t *>(bufs)[i] = sd_[u];
            }
        }
    }
    __syncthreads();
    for (int l = l_first; l <= l_last; l++) {
        const LevelGeom &gd = G.lv[l];
        const FuseRegion r = R[l], q = R[l - 1];
        const int rw = r.x1 - r.x0 + 1, rh = r.y1 - r.y0 + 1, sp = q.x1 - q.x0 + 1, e0 = eoff[l];
        const uint8_t *S = bufs + ((l - l_first) & 1 ? lds_half : 0);
        uint8_t *Dl = bufs + ((l - l_first) & 1 ? 0 : lds_half);
        uint8_t *Dg = base + interior_off(gd) + r.x0;
        const float inv_rw = __fdiv_rn(1.0f, (float)rw);
        for (int i = tid; i < rw * rh; i += nt) {
            const int ry = (int)__fmul_rn((float)i + 0.5f, inv_rw), rx = i - ry * rw;  // exact: i < 2^16, rw < 2^9
            const uint2 cx = xe[e0 + rx];
            const uint4 cy = ye[e0 + ry];
            const int c0 = (int)(cx.x & 0xFFFF), c1 = (int)(cx.x >> 16);
            const int a0 = (int16_t)(cx.y & 0xFFFF), a1 = (int16_t)(cx.y >> 16);
            const uint8_t *A = S + (int)(cy.x & 0xFFFF) * sp, *B = S + (int)(cy.x >> 16) * sp;
            const int hA = (A[c0] * a0 + A[c1] * a1) >> 4, hB = (B[c0] * a0 + B[c1] * a1) >> 4;
            // (b * h) >> 16 as the high word of (b << 16) * h (weights 0 .. 2048): b0 << 16 and b1 << 16 are the halves of cy.y
            const uint8_t p = (uint8_t)((__umulhi(cy.y << 16, (uint32_t)hA) + __umulhi(cy.y & 0xFFFF0000u, (uint32_t)hB) + 2u) >> 2);
            if (l < l_last) Dl[i] = p;
            const int y = (int)cy.w, ym = (int)cy.z;  // the row and its reflect-101 image in the halo rows, if it has one
            Dg[y * gd.pitch + rx] = p;
            if (ym != y) Dg[ym * gd.pitch + rx] = p;
        }
        lds_barrier();  // the next level reads this one from LDS; nothing here reads the planes back
    }
}

// ---------------------------------------------------------------------------
// FAST-9/16 per 30 px cell with ini/min threshold fallback and in-cell
// non-maximum suppression (src/ORBextractor.cc:1043-1107; cv::FAST semantics).
// One workgroup per (run of cells, frame).  Output: per-cell ordered
// list of packed candidates in a fixed-capacity slot + the count.
// ---------------------------------------------------------------------------
#ifdef ORBGPU_STAMPS
// Debug build only (make CXXFLAGS+=-DORBGPU_STAMPS, tools/fast_stamps.py): cycle counter deltas of the kernel's
// phases per workgroup.
constexpr int kStampWgs = 1 << 19;
__device__ unsigned g_fast_stamps[8 * kStampWgs];
#endif
// LDS tile pitch in bytes, a template parameter of the FAST kernels: 72 (18 dwords: one lead byte + a 66 px row, rounded
// up) holds every cell the reference can make; 48 holds cell images up to 44 px wide (grids of ~30 px cells on levels
// wider than a few cells, the usual case) in two thirds of the LDS and of the copy's instructions.
constexpr int kTPWide = 72, kTPNarrow = 48;

// Strength A of the best 9-long arc: max over the 16 arcs of the minimum
// signed contrast, both polarities.  A pixel is a FAST corner at threshold t
// iff A > t, and cv::FAST's response is then A-1.  Both polarities run in one
// packed half-precision chain on (v - p, p - v): the integers -255 .. 255 are exact there, a byte b under the
// exponent byte 0x64 is the half 1024 + b (so the contrasts come out of one packed subtraction with no conversion),
// and gfx950 has three-operand packed minimum / maximum for halves, which the 16-bit integers lack.
typedef short short2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <int kTP>
__device__ __forceinline__ int fast_arc_strength(const uint8_t *c)
{
    const uint32_t v = c[0];
    const half2v cv = __builtin_bit_cast(half2v, (0x6400u | v) | ((0xE400u | v) << 16));  // (1024 + v, -(1024 + v))
    uint32_t p[16];
    p[0] = c[3 * kTP + 0];   p[1] = c[3 * kTP + 1];   p[2] = c[2 * kTP + 2];   p[3] = c[1 * kTP + 3];
    p[4] = c[0 * kTP + 3];   p[5] = c[-1 * kTP + 3];  p[6] = c[-2 * kTP + 2];  p[7] = c[-3 * kTP + 1];
    p[8] = c[-3 * kTP + 0];  p[9] = c[-3 * kTP - 1];  p[10] = c[-2 * kTP - 2]; p[11] = c[-1 * kTP - 3];
    p[12] = c[0 * kTP - 3];  p[13] = c[1 * kTP - 3];  p[14] = c[2 * kTP - 2];  p[15] = c[3 * kTP - 1];
    // min over an arc of (v - p, p - v) = (v, -v) + min over the arc of (-p, p): the ring goes in as
    // (-(1024 + p), 1024 + p), the byte under the exponent bytes 0xE4 / 0x64 in one permute, and the centre is
    // added once at the end
    half2v d[16], m3[16], m9[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = __builtin_bit_cast(half2v, __builtin_amdgcn_perm(0x0000E464u, p[k], 0x04000500u));
#define ORB_MIN3(a_, b_, c_) __builtin_elementwise_minimum(a_, __builtin_elementwise_minimum(b_, c_))
#define ORB_MAX3(a_, b_, c_) __builtin_elementwise_maximum(a_, __builtin_elementwise_maximum(b_, c_))
#pragma unroll
    for (int k = 0; k < 16; k++) m3[k] = ORB_MIN3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) m9[k] = ORB_MIN3(m3[k], m3[(k + 3) & 15], m3[(k + 6) & 15]);  // the arc k .. k+8
    half2v best = ORB_MAX3(m9[0], m9[1], m9[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) best = ORB_MAX3(best, m9[k], m9[k + 1]);
    best = __builtin_elementwise_maximum(best, m9[15]) + cv;
#undef ORB_MIN3
#undef ORB_MAX3
    return (int)fmaxf((float)best.x, (float)best.y);
}

// i / d for 0 <= i < 4096, 1 <= d <= 66, with m = ceil(2^20 / d)
__device__ __forceinline__ int div_small(int i, int m) { return (int)(__umul24((unsigned)i, (unsigned)m) >> 20); }

constexpr int kFastQuad = 4;  // consecutive cells per workgroup in batch launches (one cell per workgroup for a few frames)

// One FAST cell: geometry of :1043-1072 and where its pixels lie.
struct FastCell {
    int state;  // 0: past the last cell, 1: skipped by the rules of :1053 / :1068 (count 0), 2: to be searched
    int x0, y0, cw, ch, ndw, pitch, mdiv, mdivg, cell_cap;
    uint32_t slot_off;
    const uint8_t *src;  // the byte left of the cell image's first pixel
};

// The host tabulates every cell in eight words (extractor_api.hip, set_geometry): first interior pixel, state and
// size, source offset, the two reciprocals, pitch and slot capacity, slot offset.
__device__ __forceinline__ FastCell fast_cell_info(const uint8_t *__restrict__ pyr, const Geom &G, const uint32_t *__restrict__ cell_tab, int f,
                                                   int cell)
{
    FastCell c;
    c.state = 0;
    if (cell >= G.total_cells) return c;
    const uint4 t0 = reinterpret_cast<const uint4 *>(cell_tab)[2 * (size_t)cell];
    const uint4 t1 = reinterpret_cast<const uint4 *>(cell_tab)[2 * (size_t)cell + 1];
    c.x0 = t0.x & 0xFFFF; c.y0 = t0.x >> 16;
    c.state = t0.y & 0xFF;
    c.cw = (t0.y >> 8) & 0xFF;
    c.ch = t0.y >> 16;
    c.mdiv = (int)t0.w;
    c.mdivg = (int)t1.x;
    c.pitch = t1.y & 0xFFFF;
    c.cell_cap = t1.y >> 16;
    c.slot_off = t1.z;
    // The cell image goes one byte into the tile rows (tile byte 1 + x of row y = cell image (x, y)): the interior,
    // cell image column 3, then starts on a dword boundary.  The copy reads dwords at byte addresses.
    c.ndw = (c.cw + 4) >> 2;
    c.src = pyr + (int64_t)f * G.pyr_frame_bytes + t0.z;
    return c;
}

typedef __attribute__((address_space(3))) void fast_lvoid;

// Copy a cell image into a tile buffer (rows of kTP bytes) with direct global -> LDS loads: no registers, and the
// copy proceeds while the wave computes.  Rows and columns past the cell re-read its last row / dword (the buffer
// is padded for the last piece).  The loads are issued from an asm statement: issued through the builtin, the
// compiler orders every later LDS write and atomic of the wave behind them, which would stall the wave until the
// copy has landed.  The caller therefore waits itself (fast_stage_wait) before the barrier that precedes the first
// read of the buffer.
template <int kTP>
__device__ __forceinline__ void fast_stage_cell(const FastCell &c, uint8_t *buf, int wave, int lane, int nwaves)
{
    // A piece = one wave instruction = 64 consecutive dwords of the buffer.  Pieces start on whole tile rows, kRPP
    // rows apart: a lane's (row within the piece, column) is then the same for every piece, and a piece costs three
    // vector instructions.  The lanes past the piece's kRPP rows load the first dwords of the next piece's first
    // row, exactly what that piece's first lanes write there, so the two writers of those dwords agree.
    constexpr int RP = kTP / 4, kRPP = 64 / RP;
    const int lr = lane / RP, lc = lane - lr * RP;
    const uint32_t coloff = 4u * (uint32_t)min(lc, c.ndw - 1);
    const int npieces = (c.ch + kRPP - 1) / kRPP;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(fast_lvoid *)buf);
    const uint8_t *src = c.src;  // uniform: the loads take it as their scalar base and a 32-bit lane offset
    for (int p = wave; p < npieces; p += nwaves) {
        const uint32_t off = (uint32_t)min(kRPP * p + lr, c.ch - 1) * (uint32_t)c.pitch + coloff;
        const uint32_t dst = lds0 + (uint32_t)p * (uint32_t)(kRPP * kTP);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(off), "s"(src), "s"(dst)
                     : "memory");
    }
}
__device__ __forceinline__ void fast_stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int kTP, int kFastThreads>
__global__ __launch_bounds__(kFastThreads) void k_fast_cells(const uint8_t *__restrict__ pyr, Geom G, const uint32_t *__restrict__ cell_tab,
                                                    uint32_t *__restrict__ slots, int32_t *__restrict__ cell_counts,
                                                    int cells_per_wg, int lds_score_off, int SP, int lds_queue_off, int qcap)
{
    // dynamic LDS, sized by the launcher for the largest cell of the geometry:
    // tile | score (interior + 1 px zero border, rows of SP bytes) | queue; the list of maxima takes the tile's place
    // once the arc stage has read it for the last time
    extern __shared__ __align__(16) uint8_t fast_lds[];
    uint8_t *tile = fast_lds;
    uint8_t *score = tile + lds_score_off;
    uint32_t *maxima = reinterpret_cast<uint32_t *>(tile);  // strict 3x3 maxima are never adjacent: at most a quarter of the pixels
    uint16_t *queue = reinterpret_cast<uint16_t *>(tile + lds_queue_off);
    __shared__ int qn, gqn, nmax, nini;
    uint32_t *gqueue = reinterpret_cast<uint32_t *>(score);  // the group queue lives where the score plane is cleared afterwards
#ifdef ORBGPU_STAMPS
    unsigned long long stamp_prev = __builtin_readcyclecounter();
    const unsigned stamp_wg = blockIdx.y * gridDim.x + blockIdx.x;
#define ORB_STAMP(k_) do { if (threadIdx.x == 0 && stamp_wg < (unsigned)kStampWgs) { const unsigned long long now_ = __builtin_readcyclecounter(); g_fast_stamps[stamp_wg * 8 + k_] += (unsigned)(now_ - stamp_prev); stamp_prev = now_; } } while (0)
#else
#define ORB_STAMP(k_) do { } while (0)
#endif

    // A workgroup searches cells_per_wg consecutive cells, one after the other.  In batches it is a single wave (the
    // barriers below then cost nothing) with little LDS, so that a CU holds many of them and the wait of one for its
    // cell image is filled by the others; a second tile buffer that prefetches the next cell and a second wave
    // measured slower (fewer workgroups per CU, waves idling at the barriers between the stages).
    const int f = blockIdx.y, tid = threadIdx.x;
    const int cell0 = blockIdx.x * cells_per_wg;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#pragma unroll 1
    for (int k = 0; k < cells_per_wg; k++) {
        const int cell = cell0 + k;
        const FastCell cu = fast_cell_info(pyr, G, cell_tab, f, cell);
        if (cu.state == 0) break;
        int32_t *cnt_out = cell_counts + (int64_t)f * G.total_cells + cell;
        uint8_t *cur = tile;
        if (cu.state == 1) {
            if (tid == 0) *cnt_out = 0;
            continue;
        }
        const int wi = cu.cw - 6, hi = cu.ch - 6;
    fast_stage_cell<kTP>(cu, cur, wave, lane, kFastThreads / 64);
    if (tid == 0) { qn = 0; gqn = 0; nmax = 0; nini = 0; }
    fast_stage_wait();
    __syncthreads();
    ORB_STAMP(1);

    // stage 1: cheap rejection at the lower threshold, four horizontally adjacent pixels per lane: any 9-long
    // arc contains one pixel of each opposite pair (k, k+8), so a corner needs the brighter of every even pair
    // below v - t or the darker of every even pair above v + t.  The ring bytes come out of 11 aligned dword
    // reads through byte permutes (column 0: a mask) as 16-bit pairs (pixels 0|2 and 1|3 of the group), the test is
    // packed 16-bit min / max: the 16-bit values are taken as half-precision denormals (a byte b is b * 2^-24; the
    // order of such values is that of the bytes), which have the three-operand packed minimum / maximum; the two
    // comparisons with v -+ t are 32-bit integer additions under guard bits.  Survivors go to a dense pixel queue
    // for the arc evaluation.
    const int t = G.minTh;
    const int mdiv = cu.mdiv;
    const uint8_t *t0 = cur + 1;  // pixel (x, y) of the cell image is t0[y * kTP + x]
    const uint32_t *tile32 = reinterpret_cast<const uint32_t *>(cur);
    const int ngx = (wi + 3) >> 2;  // groups of four interior pixels = dwords 1 .. ngx of a tile row
    {
        const uint32_t tg = ((uint32_t)t * 0x10001u) | 0x80008000u;  // the threshold under a guard bit in both halves
        constexpr int RP = kTP / 4;
        constexpr uint32_t kSel13 = 0x0c030c01u, kSel24 = 0x0c040c02u, kSel35 = 0x0c050c03u, kSel46 = 0x0c060c04u;
        // thread -> (row, group of the row) once per cell; an iteration of the workgroup covers rows_it whole rows, so
        // the loop only adds constants (threads past rows_it * ngx idle: at most ngx - 1 of them)
        const int rows_it = div_small(kFastThreads, cu.mdivg);
        const int ty = div_small(tid, cu.mdivg), gx = tid - (int)__umul24(ty, ngx);
        const uint32_t *r = tile32 + ty * RP + gx;
        uint32_t yg = ((uint32_t)ty << 4) | (uint32_t)gx;
        for (int y = ty < rows_it ? ty : hi; y < hi; y += rows_it, r += rows_it * RP, yg += (uint32_t)rows_it << 4) {
            // rows y .. y+6 of the tile, three dwords each; the group's pixels are the bytes of the middle dword of row y+3
            const uint32_t P8 = r[1], Am = r[RP], A0 = r[RP + 1], A1 = r[RP + 2];
            const uint32_t Cm = r[3 * RP], C0 = r[3 * RP + 1], C1 = r[3 * RP + 2];
            const uint32_t Bm = r[5 * RP], B0 = r[5 * RP + 1], B1 = r[5 * RP + 2], P0 = r[6 * RP + 1];
            uint32_t verdict[2];
#pragma unroll
            for (int odd = 0; odd < 2; odd++) {
                // byte offsets of the ring from the group's first byte: pixels j = odd, odd + 2
                const uint32_t selp3 = odd ? kSel46 : kSel35;  // column +3 from (C1, C0)
                const uint32_t selm3 = odd ? kSel24 : kSel13;  // column -3 from (C0, Cm)
                const uint32_t selp2 = odd ? kSel35 : kSel24;  // column +2 from (x1, x0); column -2 from (x0, xm)
#define ORB_P(hi_, lo_, sel_) __builtin_bit_cast(half2v, __builtin_amdgcn_perm(hi_, lo_, sel_))
                // column 0 needs no permute: bytes 0 | 2 or 1 | 3 of the dword by mask (and shift)
#define ORB_C(x_) __builtin_bit_cast(half2v, ((odd ? (x_) >> 8 : (x_)) & 0x00FF00FFu))
                const half2v v = ORB_C(C0), p0 = ORB_C(P0), p8 = ORB_C(P8);
#undef ORB_C
                const half2v p4 = ORB_P(C1, C0, selp3), p12 = ORB_P(C0, Cm, selm3);
                const half2v p2 = ORB_P(B1, B0, selp2), p14 = ORB_P(B0, Bm, selp2);
                const half2v p6 = ORB_P(A1, A0, selp2), p10 = ORB_P(A0, Am, selp2);
#undef ORB_P
#define ORB_MIN(a_, b_) __builtin_elementwise_minimum(a_, b_)
#define ORB_MAX(a_, b_) __builtin_elementwise_maximum(a_, b_)
                const half2v dmax = ORB_MAX(ORB_MAX(ORB_MIN(p0, p8), ORB_MIN(p4, p12)), ORB_MAX(ORB_MIN(p2, p10), ORB_MIN(p6, p14)));
                const half2v bmin = ORB_MIN(ORB_MIN(ORB_MAX(p0, p8), ORB_MAX(p4, p12)), ORB_MIN(ORB_MAX(p2, p10), ORB_MAX(p6, p14)));
#undef ORB_MIN
#undef ORB_MAX
                // dmax < v - t  or  bmin > v + t.  The halves are small non-negative integers, so the comparisons are
                // plain 32-bit additions with a guard bit on top of each half (no borrow crosses the halves): bit 15 of
                // a half of (dmax + t + 0x8000) - v is set unless dmax < v - t, likewise for (v + t + 0x8000) - bmin
                const uint32_t vi = __builtin_bit_cast(uint32_t, v);
                const uint32_t r1 = __builtin_bit_cast(uint32_t, dmax) + tg - vi, r2 = vi + tg - __builtin_bit_cast(uint32_t, bmin);
                verdict[odd] = ~(r1 & r2);
            }
            // bits 15 / 14 / 31 / 30 of the entry = pixels 0 / 1 / 2 / 3.  Groups with a survivor go to the group queue as
            // verdict bits | row << 4 | group: one ballot and one LDS atomic per wave (no order is needed: emission
            // ranks by pixel index).  Pixels beyond the interior's last column were tested on whatever lies there;
            // the expansion below drops them.
            const bool any = ((verdict[0] | verdict[1]) & 0x80008000u) != 0;
            const uint64_t m = __ballot(any);
            if (m) {  // wave-uniform
                int base = 0;
                const uint64_t act = __ballot(true);
                if (lane_id() == __ffsll((long long)act) - 1) base = atomicAdd(&gqn, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (any)
                    gqueue[base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] =
                        (verdict[0] & 0x80008000u) | ((verdict[1] & 0x80008000u) >> 1) | yg;
            }
        }
    }
    __syncthreads();
    ORB_STAMP(2);
    // the surviving pixels of the queued groups, densely into the pixel queue: a lane per group, positions from a
    // wave scan of the survivor counts (a wave takes chunks of 64 groups, all its lanes run: the scan needs them);
    // entries past the pixel queue's capacity are dropped: the cell is then evaluated densely (below)
    {
        const int ng = gqn;
        for (int g0 = wave << 6; g0 < ng; g0 += kFastThreads) {
            const int g = g0 + lane;
            const uint32_t e = g < ng ? gqueue[g] : 0u;
            const int y = (e >> 4) & 63, x0 = (e & 15) << 2, left = wi - x0;
            const int pix0 = (int)__umul24(y, wi) + x0;
            const bool s0 = (e & 0x8000u) != 0, s1 = (e & 0x4000u) != 0 && left > 1, s2 = (e & 0x80000000u) != 0 && left > 2,
                       s3 = (e & 0x40000000u) != 0 && left > 3;
            const int cnt = (int)s0 + (int)s1 + (int)s2 + (int)s3;
            const int incl = wave_scan_incl(cnt);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (total) {  // wave-uniform
                int base = 0;
                if (lane_id() == 0) base = atomicAdd(&qn, total);
                int pos = __builtin_amdgcn_readfirstlane(base) + incl - cnt;
                if (s0 && pos < qcap) queue[pos] = (uint16_t)pix0;
                pos += (int)s0;
                if (s1 && pos < qcap) queue[pos] = (uint16_t)(pix0 + 1);
                pos += (int)s1;
                if (s2 && pos < qcap) queue[pos] = (uint16_t)(pix0 + 2);
                pos += (int)s2;
                if (s3 && pos < qcap) queue[pos] = (uint16_t)(pix0 + 3);
            }
        }
    }
    __syncthreads();
    // the score plane (interior + 1 px border) starts from zero; it takes the group queue's place
    for (int i = tid; i < ((hi + 2) * SP + 15) >> 4; i += kFastThreads) reinterpret_cast<uint4 *>(score)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    ORB_STAMP(3);
    // The queue holds qcap entries, which keeps the workgroup's LDS small enough for a CU to hold 32 of them; a cell
    // with more survivors (noise at a low threshold) is evaluated densely instead, every pixel of it: the arc score
    // decides alone, so the result is the same.
    const bool dense = qn > qcap;  // uniform
    const int nq = dense ? (int)__umul24(wi, hi) : qn;
    for (int q = tid; q < nq; q += kFastThreads) {
        const int i = dense ? q : (int)queue[q];
        const int y = div_small(i, mdiv), x = i - (int)__umul24(y, wi);
        const int A = fast_arc_strength<kTP>(t0 + (y + 3) * kTP + (x + 3));
        if (A > t) score[(y + 1) * SP + (x + 1)] = (uint8_t)(A - 1);
    }
    __syncthreads();
    ORB_STAMP(4);
    // stage 2: strict 3x3 maximum inside the cell, evaluated for the corners only;
    // a maximum is listed as pixel index | initial-threshold flag << 12 | score << 16
    for (int q = tid; q < nq; q += kFastThreads) {
        const int i = dense ? q : (int)queue[q];
        const int y = div_small(i, mdiv), x = i - (int)__umul24(y, wi);
        const uint8_t *s = score + (y + 1) * SP + (x + 1);
        const int sc = s[0];
        if (sc == 0) continue;
        const int nb = max(max(max((int)s[-1], (int)s[1]), max((int)s[-SP - 1], (int)s[-SP])),
                           max(max((int)s[-SP + 1], (int)s[SP - 1]), max((int)s[SP], (int)s[SP + 1])));
        if (sc > nb) {
            const int ini = sc >= G.iniTh ? 1 : 0;
            maxima[atomicAdd(&nmax, 1)] = (uint32_t)i | ((uint32_t)ini << 12) | ((uint32_t)sc << 16);
            if (ini) atomicAdd(&nini, 1);
        }
    }
    __syncthreads();
    ORB_STAMP(5);

    // stage 3: threshold choice (:1071-1082) and row-major emission: the position of
    // a maximum in the output is its rank by pixel index among the chosen ones
    const int n = nmax, use_ini = nini > 0;
    const int count = use_ini ? nini : n;
    uint32_t *slot = slots + (int64_t)f * G.slot_frame_u32 + cu.slot_off;
    if (n <= 64) {
        // the usual case: the maxima fit one wave, which ranks them through lane reads instead of LDS reads
        if (tid < 64) {
            const uint32_t e = tid < n ? maxima[tid] : 0u;
            const bool chosen = tid < n && (!use_ini || (e & 0x1000u));
            const uint32_t pix = e & 0xFFFu, key = chosen ? pix : 0xFFFFu;  // the others rank behind every chosen one
            int pos = 0;
            for (int j = 0; j < n; j++) pos += (uint32_t)__builtin_amdgcn_readlane((int)key, j) < key ? 1 : 0;
            if (chosen) {
                const int y = div_small((int)pix, mdiv), x = (int)pix - (int)__umul24(y, wi);
                if (pos < cu.cell_cap) slot[pos] = pack_cand(x + cu.x0, y + cu.y0, (int)(e >> 16));
            }
        }
    } else
    for (int k = tid; k < n; k += kFastThreads) {
        const uint32_t e = maxima[k];
        if (use_ini && !(e & 0x1000u)) continue;
        const uint32_t need = use_ini ? 0x1000u : 0u, pix = e & 0xFFFu;
        int pos = 0;
        for (int j = 0; j < n; j++) {
            const uint32_t o = maxima[j];
            pos += ((o & 0x1000u) >= need && (o & 0xFFFu) < pix) ? 1 : 0;
        }
        const int y = div_small((int)pix, mdiv), x = (int)pix - (int)__umul24(y, wi);
        // keypoint coordinates relative to (16,16): position in the cell image + the cell's origin with its
        // 3 px apron, which the host table holds as (x0, y0) (:1099-1100)
        if (pos < cu.cell_cap) slot[pos] = pack_cand(x + cu.x0, y + cu.y0, (int)(e >> 16));
    }
    if (tid == 0) *cnt_out = min(count, cu.cell_cap);
    __syncthreads();  // the next cell reuses the tile (the maxima), the score plane and the counters
    ORB_STAMP(6);
    }
}

// ---------------------------------------------------------------------------
// DistributeOctTree (src/ORBextractor.cc:606-674, 688-1006), one workgroup of
// 1024 threads per (level, frame).
//
// The reference keeps a std::list of nodes and always inserts at the front, so
// list order == descending creation order; the node arrays here are kept in
// list order.  A "split pass" divides a set S of nodes given in processing
// order: the new list is reverse(children in creation order) ++ (nodes not in
// S, in their old order).  Points of a node are a contiguous, order-preserving
// range of packed candidates in one of two candidate buffers; a split is a
// 4-way partition of that range into the other buffer.
//
// The order of the points inside a node does not matter to the tree: boxes and
// populations decide every step.  It matters once, at the end, where the
// reference keeps the first of several points that share the best response in
// a node's vector (:984-996, strict >), and that vector is in the order the
// cells were visited (:1087-1130: cell row, cell column, then cv::FAST's
// row-major order inside the cell).  That rank is a function of the point's
// position alone, so the last step orders ties by (cell row, cell column, y, x)
// and the partitions are free to be unstable: the roots and every node of more
// than kBigNode points are split by all waves at once through atomic cursors
// (a stable split of such a node costs two workgroup scans per 1024 points).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int qn_count(const QNode &n) { return n.count & 0x7FFFFFFF; }
__device__ __forceinline__ int qn_buf(const QNode &n) { return (int)((uint32_t)n.count >> 31); }

constexpr int kBigNode = 512;  // nodes above this many points are split by the whole workgroup
constexpr int kQuadNode = 48;  // four nodes of at most this many points each share a wave, sixteen lanes per node
constexpr int kBigGroup = 16;  // big nodes scattered between two barriers (four LDS cursors and sixteen counters each)
constexpr int kOctUnroll = 4;  // 64-point pieces a wave has in flight
constexpr int kOctTmp = 20 * kBigGroup;  // LDS words shared by the roots' tables and the big nodes' cursors and counters

// bits 16 g .. 16 g + 15 of a ballot: the lanes of sixteen-lane group g
__device__ __forceinline__ uint32_t sub16(uint64_t b, int g)
{
    const uint32_t h = (g & 2) ? (uint32_t)(b >> 32) : (uint32_t)b;
    return (h >> ((g & 1) << 4)) & 0xFFFFu;
}

// The populations of a node's four quadrants travel with the node (three of them, 21 bits each; the fourth is the
// rest of the node's count): the pass that moves a node's points into its children also counts, for every child,
// the points of the child's own quadrants, so a round needs no counting pass of its own.
typedef unsigned long long QKids;
__device__ __forceinline__ QKids pack_kids(int k0, int k1, int k2) { return (QKids)k0 | ((QKids)k1 << 21) | ((QKids)k2 << 42); }
__device__ __forceinline__ void unpack_kids(QKids v, int cnt, int *k0, int *k1, int *k2, int *k3)
{
    *k0 = (int)(v & 0x1FFFFFu);
    *k1 = (int)((v >> 21) & 0x1FFFFFu);
    *k2 = (int)((v >> 42) & 0x1FFFFFu);
    *k3 = cnt - *k0 - *k1 - *k2;
}

struct OctCtx {
    uint32_t *buf0, *buf1;
    QNode *nl0, *nl1;
    QKids *kc0, *kc1;  // node_cap each: the quadrant populations of the nodes of nl0 / nl1
    int32_t *S;       // node_cap: positions (in the current list) of the nodes to split
    int32_t *cbase;   // node_cap: exclusive prefix of non-empty children per S entry
    int32_t *mark;    // node_cap: rank among the nodes that stay (a pass over all nodes) / 1 where the node is in S (a prefix of the sorted candidates)
    int32_t *big;     // node_cap: S entries whose node is split by all waves
    int *scan_lds;    // 64 ints (block_scan1, block_scan1x2)
    int *bigcur;      // 4 * kBigGroup LDS cursors
    int *bigk;        // 16 * kBigGroup LDS counters: points per (child, quadrant of the child)
    int *nbig;        // LDS counter of c.big (zero between passes)
    int par;          // call parity of the workgroup scans
    __device__ __forceinline__ uint32_t *buf(int i) const { return i ? buf1 : buf0; }
    __device__ __forceinline__ QNode *nl(int i) const { return i ? nl1 : nl0; }
    __device__ __forceinline__ QKids *kc(int i) const { return i ? kc1 : kc0; }
};

__device__ __forceinline__ int quadrant_of(uint32_t p, int midx, int midy)
{
    return (cand_x(p) < midx ? 0 : 1) | (cand_y(p) < midy ? 0 : 2);
}

// What a split of node n needs to know about boxes: the node's centre lines (ExtractorNode::DivideNode, :606-641:
// halfX = ceil((x1 - x0) / 2)) and those of its four children.
struct SplitBox {
    int midx, midy, cxl, cxr, cyt, cyb;
    __device__ __forceinline__ SplitBox(int x0, int x1, int y0, int y1)
    {
        midx = x0 + ((x1 - x0 + 1) >> 1);
        midy = y0 + ((y1 - y0 + 1) >> 1);
        cxl = x0 + ((midx - x0 + 1) >> 1);
        cxr = midx + ((x1 - midx + 1) >> 1);
        cyt = y0 + ((midy - y0 + 1) >> 1);
        cyb = midy + ((y1 - midy + 1) >> 1);
    }
    // 4 * quadrant + quadrant within that child
    __device__ __forceinline__ int code(uint32_t p) const
    {
        const int x = cand_x(p), y = cand_y(p);
        const int qx = x < midx ? 0 : 1, qy = y < midy ? 0 : 1;
        const int gx = x < (qx ? cxr : cxl) ? 0 : 1, gy = y < (qy ? cyb : cyt) ? 0 : 1;
        return ((qx | (qy << 1)) << 2) | gx | (gy << 1);
    }
};

// Two exclusive workgroup scans behind one barrier (as block_scan1; lds: 64 ints).
__device__ __forceinline__ void block_scan1x2(int a, int b, int *exa, int *exb, int *tota, int *totb, int *lds, int &par)
{
    const int lane = lane_id(), wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int *half = lds + (par << 5);
    par ^= 1;
    const int ia = wave_scan_incl(a), ib = wave_scan_incl(b);
    if (lane == 63) { half[wid] = ia; half[16 + wid] = ib; }
    __syncthreads();
    const int wa = lane < nw ? half[lane] : 0, wb = lane < nw ? half[16 + lane] : 0;
    const int sa = wave_scan_incl(wa), sb = wave_scan_incl(wb);
    *tota = __builtin_amdgcn_readlane(sa, 63);
    *totb = __builtin_amdgcn_readlane(sb, 63);
    *exa = ia - a + __shfl(sa - wa, wid);
    *exb = ib - b + __shfl(sb - wb, wid);
}

// The quadrant populations of the first nodes (the roots), list positions 0 .. size-1 of nl0: every wave takes
// 64-point pieces of every root, adds up in registers and sends one atomic add per quadrant.  tmp: 4 * size words.
__device__ void oct_count_roots(OctCtx &c, int size, int32_t *tmp)
{
    const int tid = threadIdx.x, nt = blockDim.x, lane = lane_id(), wid = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6;
    for (int i = tid; i < 4 * size; i += nt) tmp[i] = 0;
    __syncthreads();
    int acc = 0;  // pieces before the current root: the pieces of all roots go round the waves
    for (int r = 0; r < size; r++) {
        const QNode n = c.nl0[r];
        const int cnt = qn_count(n), nc = (cnt + 63) >> 6;
        const int j0 = (wid - acc) & (nw - 1);
        acc += nc;
        if (j0 >= nc) continue;
        const uint32_t *pts = c.buf(qn_buf(n)) + n.start;
        const int midx = n.x0 + ((n.x1 - n.x0 + 1) >> 1), midy = n.y0 + ((n.y1 - n.y0 + 1) >> 1);
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int j = j0; j < nc; j += kOctUnroll * nw) {
            uint32_t pv[kOctUnroll];
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) {
                const int i = ((j + u * nw) << 6) + lane;
                pv[u] = i < cnt ? pts[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) {
                const int i = ((j + u * nw) << 6) + lane;
                const int q = i < cnt ? quadrant_of(pv[u], midx, midy) : -1;
                c0 += __popcll(__ballot(q == 0));
                c1 += __popcll(__ballot(q == 1));
                c2 += __popcll(__ballot(q == 2));
                c3 += __popcll(__ballot(q == 3));
            }
        }
        if (lane < 4) {
            const int v = lane == 0 ? c0 : lane == 1 ? c1 : lane == 2 ? c2 : c3;
            if (v) atomicAdd(&tmp[4 * r + lane], v);
        }
    }
    __syncthreads();
    for (int r = tid; r < size; r += nt) c.kc0[r] = pack_kids(tmp[4 * r], tmp[4 * r + 1], tmp[4 * r + 2]);
    __syncthreads();
}

// Child q (0..3) of node n created by S entry e (ExtractorNode::DivideNode, :606-641); g0..g2: the populations of
// the child's own first three quadrants.
__device__ __forceinline__ void emit_child(const OctCtx &c, int cur, const QNode &n, int e, int q, int midx, int midy,
                                           int k0, int k1, int k2, int k3, int s0, int nchild, int db, int g0, int g1, int g2)
{
    const int kq = q == 0 ? k0 : q == 1 ? k1 : q == 2 ? k2 : k3;
    if (kq <= 0) return;
    const int before = (q > 0 && k0 > 0) + (q > 1 && k1 > 0) + (q > 2 && k2 > 0);
    const int created = c.cbase[e] + before;  // creation index within this pass
    QNode ch;
    ch.x0 = (q & 1) ? (int16_t)midx : n.x0;
    ch.x1 = (q & 1) ? n.x1 : (int16_t)midx;
    ch.y0 = (q & 2) ? (int16_t)midy : n.y0;
    ch.y1 = (q & 2) ? n.y1 : (int16_t)midy;
    ch.start = s0 + (q > 0 ? k0 : 0) + (q > 1 ? k1 : 0) + (q > 2 ? k2 : 0);
    ch.count = kq | (db << 31);
    c.nl(cur ^ 1)[nchild - 1 - created] = ch;  // push_front: later creations sit nearer the front
    c.kc(cur ^ 1)[nchild - 1 - created] = pack_kids(g0, g1, g2);
}

// A pass over every node of the list (:778-887): S = the nodes of more than one point, in list order; the others
// move behind the children-to-be in their old order.  Returns the size of S, the number of children the split of S
// creates and how many of those hold more than one point.
__device__ void oct_scan_all(OctCtx &c, int cur, int size, int *out_nS, int *out_children, int *out_expand)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const QNode *nl = c.nl(cur);
    const QKids *kc = c.kc(cur);
    int nS = 0, nstay = 0, nchild = 0, nexp = 0;
    for (int b = 0; b < size; b += nt) {
        const int i = b + tid;
        int inS = 0, stay = 0, ne = 0, nx = 0;
        if (i < size) {
            const int cnt = qn_count(nl[i]);
            inS = cnt > 1;
            stay = !inS;
            if (inS) {
                int k0, k1, k2, k3;
                unpack_kids(kc[i], cnt, &k0, &k1, &k2, &k3);
                ne = (k0 > 0) + (k1 > 0) + (k2 > 0) + (k3 > 0);
                nx = (k0 > 1) + (k1 > 1) + (k2 > 1) + (k3 > 1);
            }
        }
        int exa, exb, ta, tb;
        block_scan1x2(inS | (stay << 16), ne | (nx << 16), &exa, &exb, &ta, &tb, c.scan_lds, c.par);
        if (inS) {
            const int e = nS + (exa & 0xFFFF);
            c.S[e] = i;
            c.cbase[e] = nchild + (exb & 0xFFFF);
        }
        if (stay) c.mark[i] = nstay + (exa >> 16);
        nS += ta & 0xFFFF; nstay += ta >> 16;
        nchild += tb & 0xFFFF; nexp += tb >> 16;
    }
    __syncthreads();
    for (int i = tid; i < size; i += nt) {
        const QNode n = nl[i];
        if (qn_count(n) <= 1) c.nl(cur ^ 1)[nchild + c.mark[i]] = n;  // (its quadrant populations are never looked at)
    }
    *out_nS = nS;
    *out_children = nchild;
    *out_expand = nexp;
}

// The same for S = the first nS of the sorted candidates of a final pass (:893-968).
__device__ void oct_scan_prefix(OctCtx &c, int cur, int size, int nS, int *out_children, int *out_expand)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const QNode *nl = c.nl(cur);
    const QKids *kc = c.kc(cur);
    int nchild = 0, nexp = 0;
    for (int b = 0; b < nS; b += nt) {
        const int e = b + tid;
        int ne = 0, nx = 0;
        if (e < nS) {
            const int i = c.S[e];
            int k0, k1, k2, k3;
            unpack_kids(kc[i], qn_count(nl[i]), &k0, &k1, &k2, &k3);
            ne = (k0 > 0) + (k1 > 0) + (k2 > 0) + (k3 > 0);
            nx = (k0 > 1) + (k1 > 1) + (k2 > 1) + (k3 > 1);
        }
        int total;
        const int ex = block_scan1(ne | (nx << 16), &total, c.scan_lds, c.par);
        if (e < nS) c.cbase[e] = nchild + (ex & 0xFFFF);
        nchild += total & 0xFFFF;
        nexp += total >> 16;
    }
    for (int i = tid; i < size; i += nt) c.mark[i] = 0;
    __syncthreads();
    for (int e = tid; e < nS; e += nt) c.mark[c.S[e]] = 1;
    __syncthreads();
    int carry = 0;
    for (int b = 0; b < size; b += nt) {
        const int i = b + tid;
        const int keep = (i < size) ? (c.mark[i] == 0) : 0;
        int total;
        const int ex = block_scan1(keep, &total, c.scan_lds, c.par);
        if (keep) {
            c.nl(cur ^ 1)[nchild + carry + ex] = nl[i];
            c.kc(cur ^ 1)[nchild + carry + ex] = kc[i];
        }
        carry += total;
    }
    *out_children = nchild;
    *out_expand = nexp;
}

// Split the nodes S[0..nS): their points go to the other candidate buffer, partitioned by quadrant, the child
// nodes (with the populations of their own quadrants) to the front of the other node list.  Needs S and cbase
// complete (a barrier since they were written); ends with a barrier.
__device__ void oct_scatter(OctCtx &c, int cur, int nS, int nchild)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = lane_id(), wid = __builtin_amdgcn_readfirstlane(tid >> 6), nw = nt >> 6;  // nw: a power of two
    const QNode *nl = c.nl(cur);
    const QKids *kc = c.kc(cur);
    // few nodes (the early rounds): a wave per node would leave most waves idle, so every node of more than a
    // wave's worth of points is cut into 64-point pieces that are dealt to all waves
    const int thr = nS < 4 * nw ? 64 : kBigNode;
    for (int e = tid; e < nS; e += nt)
        if (qn_count(nl[c.S[e]]) > thr) c.big[atomicAdd(c.nbig, 1)] = e;
    __syncthreads();
    const int nB = *c.nbig;
    // small nodes: a wave takes four S entries (nw apart, so that a handful of nodes spreads over the waves).  While
    // none of the four holds more than kQuadNode points (the late rounds: many nodes of a few points each) sixteen
    // lanes work on each; otherwise the wave takes them one after the other.
    for (int base = 0; base < nS; base += 4 * nw) {
        const int grp = lane >> 4, sl = lane & 15, eg = base + grp * nw + wid;
        QNode ng = {};
        int cntg = 0;
        if (eg < nS) {
            ng = nl[c.S[eg]];
            cntg = qn_count(ng);
        }
        if (__ballot(cntg > kQuadNode) == 0ull) {
            const int sb = qn_buf(ng), db = sb ^ 1;
            const uint32_t *pts = c.buf(sb) + ng.start;
            uint32_t *dst = c.buf(db);
            const SplitBox sbx(ng.x0, ng.x1, ng.y0, ng.y1);
            int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
            if (eg < nS) unpack_kids(kc[c.S[eg]], cntg, &k0, &k1, &k2, &k3);
            const int s0 = ng.start, s1 = s0 + k0, s2 = s1 + k1, s3 = s2 + k2;
            int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            int gk[12];  // populations of the first three quadrants of each child
#pragma unroll
            for (int k = 0; k < 12; k++) gk[k] = 0;
            const uint32_t lt = (1u << sl) - 1u;
            for (int b = 0; __ballot(b < cntg) != 0ull; b += 16) {
                const int i = b + sl;
                int code = -1;
                uint32_t p = 0;
                if (i < cntg) {
                    p = pts[i];
                    code = sbx.code(p);
                }
                const int q = code >> 2;  // -1 stays -1
                const uint32_t m0 = sub16(__ballot(q == 0), grp), m1 = sub16(__ballot(q == 1), grp), m2 = sub16(__ballot(q == 2), grp),
                               m3 = sub16(__ballot(q == 3), grp);
                if (q == 0) dst[s0 + r0 + __popc(m0 & lt)] = p;
                else if (q == 1) dst[s1 + r1 + __popc(m1 & lt)] = p;
                else if (q == 2) dst[s2 + r2 + __popc(m2 & lt)] = p;
                else if (q == 3) dst[s3 + r3 + __popc(m3 & lt)] = p;
                r0 += __popc(m0); r1 += __popc(m1); r2 += __popc(m2); r3 += __popc(m3);
#pragma unroll
                for (int k = 0; k < 12; k++) gk[k] += __popc(sub16(__ballot(code == (k / 3) * 4 + k % 3), grp));
            }
            if (sl < 4 && eg < nS)
                emit_child(c, cur, ng, eg, sl, sbx.midx, sbx.midy, k0, k1, k2, k3, s0, nchild, db,
                           sl == 0 ? gk[0] : sl == 1 ? gk[3] : sl == 2 ? gk[6] : gk[9], sl == 0 ? gk[1] : sl == 1 ? gk[4] : sl == 2 ? gk[7] : gk[10],
                           sl == 0 ? gk[2] : sl == 1 ? gk[5] : sl == 2 ? gk[8] : gk[11]);
            continue;
        }
        for (int e = base + wid; e < min(base + 4 * nw, nS); e += nw) {
            const QNode n = nl[c.S[e]];
            const int cnt = qn_count(n), sb = qn_buf(n), db = sb ^ 1;
            if (cnt > thr) continue;
            const uint32_t *pts = c.buf(sb) + n.start;
            uint32_t *dst = c.buf(db);
            const SplitBox sbx(n.x0, n.x1, n.y0, n.y1);
            int k0, k1, k2, k3;
            unpack_kids(kc[c.S[e]], cnt, &k0, &k1, &k2, &k3);
            const int s0 = n.start, s1 = s0 + k0, s2 = s1 + k1, s3 = s2 + k2;
            int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            int gk[12];
#pragma unroll
            for (int k = 0; k < 12; k++) gk[k] = 0;
            const uint64_t lt = (1ull << lane) - 1ull;
            for (int b = 0; b < cnt; b += 64) {
                const int i = b + lane;
                int code = -1;
                uint32_t p = 0;
                if (i < cnt) {
                    p = pts[i];
                    code = sbx.code(p);
                }
                const int q = code >> 2;
                const uint64_t b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
                if (q == 0) dst[s0 + r0 + __popcll(b0 & lt)] = p;
                else if (q == 1) dst[s1 + r1 + __popcll(b1 & lt)] = p;
                else if (q == 2) dst[s2 + r2 + __popcll(b2 & lt)] = p;
                else if (q == 3) dst[s3 + r3 + __popcll(b3 & lt)] = p;
                r0 += __popcll(b0); r1 += __popcll(b1); r2 += __popcll(b2); r3 += __popcll(b3);
#pragma unroll
                for (int k = 0; k < 12; k++) gk[k] += __popcll(__ballot(code == (k / 3) * 4 + k % 3));
            }
            if (lane < 4)
                emit_child(c, cur, n, e, lane, sbx.midx, sbx.midy, k0, k1, k2, k3, s0, nchild, db,
                           lane == 0 ? gk[0] : lane == 1 ? gk[3] : lane == 2 ? gk[6] : gk[9],
                           lane == 0 ? gk[1] : lane == 1 ? gk[4] : lane == 2 ? gk[7] : gk[10],
                           lane == 0 ? gk[2] : lane == 1 ? gk[5] : lane == 2 ? gk[8] : gk[11]);
        }
    }
    // big nodes: their 64-point pieces are numbered through and dealt round the waves.  Lane l fetches what the
    // pieces of the l-th node of a group need and the prefix of the piece counts; a piece finds its node by one ballot
    // over the prefixes and gets the node's data by lane reads.  A piece claims its place in each child's range with
    // one atomic add per non-empty quadrant; the populations of the children's quadrants are counted by one LDS
    // atomic per point.  kOctUnroll pieces are fetched before the first is placed (the points of a large level are in
    // global memory: a wave that fetched them one by one would spend its time waiting).
    for (int g0 = 0; g0 < nB; g0 += kBigGroup) {
        const int ng = min(kBigGroup, nB - g0);
        for (int i = tid; i < 20 * kBigGroup; i += nt) c.bigk[i] = 0;  // counters and cursors (bigcur follows bigk)
        int me = 0, mstart = 0, mcb = 0, mk0 = 0, mk1 = 0, mk2 = 0, mnc = 0, mx = 0, my = 0;
        if (lane < ng) {
            me = c.big[g0 + lane];
            const QNode n = nl[c.S[me]];
            mstart = n.start;
            mcb = n.count;
            mx = (uint16_t)n.x0 | ((int)n.x1 << 16);
            my = (uint16_t)n.y0 | ((int)n.y1 << 16);
            int k3;
            unpack_kids(kc[c.S[me]], qn_count(n), &mk0, &mk1, &mk2, &k3);
            mnc = (qn_count(n) + 63) >> 6;
        }
        const int mincl = wave_scan_incl(mnc), mexcl = mincl - mnc;
        const int npieces = __builtin_amdgcn_readlane(mincl, 63);
        __syncthreads();
        const uint64_t lt = (1ull << lane) - 1ull;
        for (int P0 = wid; P0 < npieces; P0 += kOctUnroll * nw) {
            uint32_t pv[kOctUnroll];
            int bjv[kOctUnroll], qv[kOctUnroll], basev[kOctUnroll];
            uint64_t mv[kOctUnroll];
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) {
                const int P = P0 + u * nw;
                const int bj = P < npieces ? (int)__popcll(__ballot(mincl <= P)) : 0;  // the lanes behind the group hold npieces
                const int cb = __builtin_amdgcn_readlane(mcb, bj);
                const int i = ((P - __builtin_amdgcn_readlane(mexcl, bj)) << 6) + lane;
                bjv[u] = bj;
                qv[u] = (P < npieces && i < (cb & 0x7FFFFFFF)) ? 0 : -1;
                pv[u] = qv[u] == 0 ? (c.buf((int)((uint32_t)cb >> 31)) + __builtin_amdgcn_readlane(mstart, bj))[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) {
                const int xs = __builtin_amdgcn_readlane(mx, bjv[u]), ys = __builtin_amdgcn_readlane(my, bjv[u]);
                const SplitBox sbx((int16_t)(xs & 0xFFFF), (int16_t)(xs >> 16), (int16_t)(ys & 0xFFFF), (int16_t)(ys >> 16));
                const int code = qv[u] == 0 ? sbx.code(pv[u]) : -1;
                const int q = code >> 2;
                qv[u] = q;
                if (code >= 0) atomicAdd(&c.bigk[16 * bjv[u] + code], 1);
                const uint64_t b0 = __ballot(q == 0), b1 = __ballot(q == 1), b2 = __ballot(q == 2), b3 = __ballot(q == 3);
                mv[u] = q == 0 ? b0 : q == 1 ? b1 : q == 2 ? b2 : b3;  // the lanes of this lane's quadrant
                int base = 0;
                if (lane < 4) {
                    const int v = __popcll(lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3);
                    if (v) base = atomicAdd(&c.bigcur[4 * bjv[u] + lane], v);
                }
                basev[u] = base;
            }
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) {
                const int bj = bjv[u], q = qv[u];
                const int s0 = __builtin_amdgcn_readlane(mstart, bj), s1 = s0 + __builtin_amdgcn_readlane(mk0, bj),
                          s2 = s1 + __builtin_amdgcn_readlane(mk1, bj), s3 = s2 + __builtin_amdgcn_readlane(mk2, bj);
                const int o0 = __builtin_amdgcn_readlane(basev[u], 0), o1 = __builtin_amdgcn_readlane(basev[u], 1),
                          o2 = __builtin_amdgcn_readlane(basev[u], 2), o3 = __builtin_amdgcn_readlane(basev[u], 3);
                const int at = q == 0 ? s0 + o0 : q == 1 ? s1 + o1 : q == 2 ? s2 + o2 : s3 + o3;
                uint32_t *dst = c.buf((int)((uint32_t)__builtin_amdgcn_readlane(mcb, bj) >> 31) ^ 1);
                if (q >= 0) dst[at + __popcll(mv[u] & lt)] = pv[u];
            }
        }
        __syncthreads();
        // the child nodes: thread t makes child t & 3 of the group's node t >> 2 (every wave holds all of them)
        for (int t0 = 0; t0 < 4 * ng; t0 += nt) {
            const int t = t0 + tid, nd = (t >> 2) & 63;
            const int e = __shfl(me, nd), cb = __shfl(mcb, nd), xs = __shfl(mx, nd), ys = __shfl(my, nd);
            const int k0 = __shfl(mk0, nd), k1 = __shfl(mk1, nd), k2 = __shfl(mk2, nd), s0 = __shfl(mstart, nd);
            if (t < 4 * ng) {
                QNode n;
                n.x0 = (int16_t)(xs & 0xFFFF); n.x1 = (int16_t)(xs >> 16); n.y0 = (int16_t)(ys & 0xFFFF); n.y1 = (int16_t)(ys >> 16);
                n.start = s0; n.count = cb;
                const int q = t & 3;
                const int *gkp = c.bigk + 16 * (t >> 2) + 4 * q;
                emit_child(c, cur, n, e, q, n.x0 + ((n.x1 - n.x0 + 1) >> 1), n.y0 + ((n.y1 - n.y0 + 1) >> 1), k0, k1, k2,
                           (cb & 0x7FFFFFFF) - k0 - k1 - k2, s0, nchild, (int)((uint32_t)cb >> 31) ^ 1, gkp[0], gkp[1], gkp[2]);
            }
        }
        if (g0 + kBigGroup < nB) __syncthreads();  // the next group clears the counters
    }
    __syncthreads();
    if (tid == 0) *c.nbig = 0;  // for the next pass (its appends come after another barrier)
}

template <bool kLdsNodes, bool kLdsPts>
__device__ __forceinline__ void octree_level(const int level, const int f, const Geom &G, const uint32_t *__restrict__ slots,
                                             const int32_t *__restrict__ cell_counts, uint32_t *cand, QNode *nodes,
                                             int32_t *scratch, uint32_t *__restrict__ sel,
                                             int32_t *__restrict__ sel_count, int32_t *__restrict__ cand_count,
                                             int32_t *__restrict__ status, int keys_cap, int dyn_coff_off,
                                             int pts_cap, int dyn_pts_off)
{
    // dynamic LDS: keys_cap 64-bit sort keys, then (lds_nodes != 0) the node
    // arrays and per-node work lists of this level -- keeping them on chip takes
    // a dozen dependent global-memory round trips out of every subdivision round.
    // keys_cap == 0: the level holds more nodes than LDS has room for sort keys (more than 4092 features on a
    // level); the keys then live behind the level's work lists in global memory.
    extern __shared__ __align__(16) unsigned char dyn_lds[];
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(dyn_lds);
    int *coff = reinterpret_cast<int *>(dyn_lds + dyn_coff_off);  // blockDim.x + 1 cell offsets of the gather
    __shared__ int scan_lds[64];
    __shared__ int oct_tmp[kOctTmp];  // the roots' populations and starts; later the big nodes' counters and cursors
    __shared__ int sh_a, sh_nbig;
    static_assert(2 * kMaxRoots <= kOctTmp, "the roots' tables share oct_tmp");
    int *const root_cnt = oct_tmp, *const root_start = oct_tmp + kMaxRoots;

    const int tid = threadIdx.x, nt = blockDim.x;
    const LevelGeom &g = G.lv[level];
    const int N = g.quota;
#ifdef ORBGPU_OCT_TRACE
    // phase times of level 0 of frame 0 (100 MHz clock), printed at the end: a measurement build only
    unsigned long long tr_t[96];
    int tr_id[96], ntr = 0;
    const bool trace_on = level == ORBGPU_OCT_TRACE && f == 0 && tid == 0;
    const unsigned long long tr_begin = wall_clock64();
#define OCT_T(id_) do { if (trace_on && ntr < 96) { tr_id[ntr] = id_; tr_t[ntr++] = wall_clock64(); } } while (0)
#else
#define OCT_T(id_) do { } while (0)
#endif
    OCT_T(0);
    OctCtx c;
    c.buf0 = cand + (int64_t)f * G.cand_frame_u32 + g.cand_off;
    c.buf1 = c.buf0 + g.cand_cap;
    // kLdsNodes: the node lists and work lists are in LDS, and this instance derives their pointers from the LDS
    // array alone, so that every access compiles to an LDS instruction (a pointer that may be either, decided at
    // run time, makes them flat accesses: longer latency, and they tie the LDS and the vector-memory counters)
    if constexpr (kLdsNodes) {
        c.nl0 = reinterpret_cast<QNode *>(dyn_lds + (size_t)keys_cap * 8);
        c.nl1 = c.nl0 + g.node_cap;
        c.kc0 = reinterpret_cast<QKids *>(c.nl1 + g.node_cap);
        c.kc1 = c.kc0 + g.node_cap;
        int32_t *ip = reinterpret_cast<int32_t *>(c.kc1 + g.node_cap);
        c.S = ip;
        c.cbase = ip + g.node_cap;
        c.mark = ip + 2 * g.node_cap;
        c.big = ip + 3 * g.node_cap;
    } else {
        c.nl0 = nodes + (int64_t)f * G.node_frame + g.node_off;
        c.nl1 = c.nl0 + g.node_cap;
        int32_t *scr = scratch + ((int64_t)f * G.node_frame + g.node_off) * kScratchPerNode;  // 12 * node_cap words
        if (keys_cap == 0) keys = reinterpret_cast<unsigned long long *>(scr + 8 * g.node_cap);  // 2 * node_cap keys
        c.kc0 = reinterpret_cast<QKids *>(scr);
        c.kc1 = c.kc0 + g.node_cap;
        c.S = scr + 4 * g.node_cap;
        c.cbase = scr + 5 * g.node_cap;
        c.mark = scr + 6 * g.node_cap;
        c.big = scr + 7 * g.node_cap;
    }
    c.scan_lds = scan_lds;
    c.bigk = oct_tmp;  // free once the roots exist
    c.bigcur = oct_tmp + 16 * kBigGroup;
    c.nbig = &sh_nbig;
    c.par = 0;
    if (tid == 0) sh_nbig = 0;  // (barriers follow before its first use)
    int32_t *out_count = sel_count + f * G.nlevels + level;
    uint32_t *out_sel = sel + (int64_t)f * G.total_kp_cap + g.kp_off;

    // ---- gather the per-cell lists into one ordered candidate array ----------
    const int nCells = g.nCols * g.nRows;
    const int32_t *cc = cell_counts + (int64_t)f * G.total_cells + g.cell_off;
    const uint32_t *sl = slots + (int64_t)f * G.slot_frame_u32 + g.slot_off;
    uint32_t *const gathered = c.buf0;  // the gathered candidates are also kept in global memory (orbgpu_debug_level_candidates)
    if constexpr (kLdsPts) {
        // a level whose candidates fit keeps both candidate buffers in LDS: every subdivision round reads and
        // re-partitions the points, and in global memory each of those steps is a dependent round trip.  (Its own
        // instance, as for the node lists: the point accesses compile to LDS instructions.)
        c.buf0 = reinterpret_cast<uint32_t *>(dyn_lds + dyn_pts_off);
        c.buf1 = c.buf0 + pts_cap;
    }
    int M = 0;
    for (int b = 0; b < nCells; b += nt) {
        const int ci = b + tid;
        const int cnt = ci < nCells ? cc[ci] : 0;
        int total;
        const int ex = block_scan1(cnt, &total, scan_lds, c.par);
        coff[tid] = ex;
        if (tid == 0) coff[nt] = total;
        __syncthreads();
        // one thread per candidate: find its cell by bisection of the offsets
        for (int j = tid; j < total; j += nt) {
            int lo = 0, hi = nt - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (coff[mid] <= j) lo = mid; else hi = mid - 1;
            }
            const uint32_t v = sl[(int64_t)(b + lo) * g.cell_cap + (j - coff[lo])];
            c.buf0[M + j] = v;
            if constexpr (kLdsPts) gathered[M + j] = v;
        }
        __syncthreads();
        M += total;
    }
    OCT_T(1);
    if (M >= (1 << 21)) {
        // the quadrant populations travel in 21-bit fields (pack_kids): a level with two million candidates or more
        // (a 4127 x 4127 level of isolated dots) is reported, not mis-counted
        if (tid == 0) atomicOr(status, 1);
        M = (1 << 21) - 1;
    }
    if (tid == 0) cand_count[f * G.nlevels + level] = M;
    if (M == 0) {
        if (tid == 0) *out_count = 0;
        return;
    }
    __syncthreads();

    // ---- roots (:696-763) -----------------------------------------------------
    const int nIni = g.nIni;
    const float hX = g.hX;
    int rb = 0;  // buffer holding the root ranges
    if (nIni > 1) {
        // a point's root is (int)(x / hX) (:707-711); the ranges of the roots are filled in any order
        const int lane = lane_id(), wid = __builtin_amdgcn_readfirstlane(tid >> 6);
        int *root_cur = coff;  // the gather is done with it; nt + 1 >= kMaxRoots entries
        for (int r = tid; r < nIni; r += nt) { root_cnt[r] = 0; root_cur[r] = 0; }
        __syncthreads();
        for (int b = wid * 64; b < M; b += kOctUnroll * nt) {
            uint32_t pv[kOctUnroll];
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) pv[u] = b + u * nt + lane < M ? c.buf0[b + u * nt + lane] : 0u;
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) {
                int r = b + u * nt + lane < M ? (int)((float)cand_x(pv[u]) / hX) : -1;
                if (r >= nIni) r = -1;
                uint64_t todo = __ballot(r >= 0);
                while (todo) {  // one turn per distinct root among the 64 points (neighbours in x: one or two)
                    const int leader = __ffsll((unsigned long long)todo) - 1;
                    const int r0 = __builtin_amdgcn_readlane(r, leader);
                    const uint64_t m = __ballot(r == r0);
                    if (lane == leader) atomicAdd(&root_cnt[r0], __popcll(m));
                    todo &= ~m;
                }
            }
        }
        __syncthreads();
        if (wid == 0) {  // kMaxRoots = 128: two roots per lane
            const int a = 2 * lane < nIni ? root_cnt[2 * lane] : 0, b2 = 2 * lane + 1 < nIni ? root_cnt[2 * lane + 1] : 0;
            const int ex = wave_scan_incl(a + b2) - (a + b2);
            if (2 * lane < nIni) root_start[2 * lane] = ex;
            if (2 * lane + 1 < nIni) root_start[2 * lane + 1] = ex + a;
        }
        __syncthreads();
        const uint64_t lt = (1ull << lane) - 1ull;
        for (int b = wid * 64; b < M; b += kOctUnroll * nt) {
            uint32_t pv[kOctUnroll];
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) pv[u] = b + u * nt + lane < M ? c.buf0[b + u * nt + lane] : 0u;
#pragma unroll
            for (int u = 0; u < kOctUnroll; u++) {
                int r = b + u * nt + lane < M ? (int)((float)cand_x(pv[u]) / hX) : -1;
                if (r >= nIni) r = -1;
                uint64_t todo = __ballot(r >= 0);
                while (todo) {
                    const int leader = __ffsll((unsigned long long)todo) - 1;
                    const int r0 = __builtin_amdgcn_readlane(r, leader);
                    const uint64_t m = __ballot(r == r0);
                    int base = 0;
                    if (lane == leader) base = atomicAdd(&root_cur[r0], __popcll(m));
                    base = __builtin_amdgcn_readlane(base, leader);
                    if (r == r0) c.buf1[root_start[r0] + base + __popcll(m & lt)] = pv[u];
                    todo &= ~m;
                }
            }
        }
        rb = 1;
    } else if (tid == 0) {
        root_cnt[0] = M; root_start[0] = 0;
    }
    __syncthreads();
    int size = 0;
    for (int r = 0; r < nIni; r++) {
        if (root_cnt[r] > 0) {
            if (tid == 0) {
                QNode n;
                n.x0 = (int16_t)(int)(hX * (float)r);
                n.x1 = (int16_t)(int)(hX * (float)(r + 1));
                n.y0 = 0;
                n.y1 = (int16_t)(g.h - 2 * kMinBorder);
                n.start = root_start[r];
                n.count = root_cnt[r] | (rb << 31);
                c.nl0[size] = n;
            }
            size++;
        }
    }
    __syncthreads();
    OCT_T(2);

    // the roots' quadrant populations; from here on every split counts those of the nodes it creates.  (S, cbase,
    // mark and big are one block of 4 * node_cap words, idle until the first pass.)
    oct_count_roots(c, size, c.S);
    OCT_T(3);

    // ---- subdivision rounds (:778-970) -----------------------------------------
    int cur = 0;
    bool finish = false;
    while (!finish) {
        const int prevSize = size;
        int nS, nchild, nexp;
        oct_scan_all(c, cur, size, &nS, &nchild, &nexp);
        if (nS == 0) break;  // nothing can be divided: size == prevSize
        OCT_T(10);
        oct_scatter(c, cur, nS, nchild);
        size = nchild + (size - nS);
        cur ^= 1;
        OCT_T(12);
        if (size >= N || size == prevSize) break;
        if (size + nexp * 3 > N) {
            // ---- final partial passes (:893-968): largest nodes first --------
            int lastChildren = nchild;
            while (true) {
                const int prev2 = size;
                // candidates: children of the previous pass holding > 1 point.
                // key = (count, creation index); processed in descending order.
                int nC = 0;
                for (int b = 0; b < lastChildren; b += nt) {
                    const int i = b + tid;
                    int flag = 0, cnt = 0;
                    if (i < lastChildren) {
                        cnt = qn_count(c.nl(cur)[i]);
                        flag = cnt > 1;
                    }
                    int total;
                    const int ex = block_scan1(flag, &total, scan_lds, c.par);
                    if (flag) keys[nC + ex] = ((unsigned long long)cnt << 32) | (unsigned)(lastChildren - 1 - i);
                    nC += total;
                }
                if (nC == 0) { finish = true; break; }
                OCT_T(20);
                if ((long long)nC * nC <= (long long)nt * 2048) {
                    // the keys are distinct: a key's place is the number of larger keys (every lane reads the same
                    // key at a time, which LDS broadcasts); one barrier instead of the sorting network's dozens
                    __syncthreads();
                    for (int i = tid; i < nC; i += nt) {
                        const unsigned long long k = keys[i];
                        int rank = 0;
#pragma unroll 8
                        for (int j = 0; j < nC; j++) rank += keys[j] > k ? 1 : 0;
                        c.S[rank] = lastChildren - 1 - (int)(k & 0xFFFFFFFFull);
                    }
                } else {
                    int np2 = 1;
                    while (np2 < nC) np2 <<= 1;
                    for (int i = nC + tid; i < np2; i += nt) keys[i] = 0ull;
                    __syncthreads();
                    for (int k = 2; k <= np2; k <<= 1) {
                        for (int j = k >> 1; j > 0; j >>= 1) {
                            for (int i = tid; i < np2; i += nt) {
                                const int ixj = i ^ j;
                                if (ixj > i) {
                                    const unsigned long long a = keys[i], bb = keys[ixj];
                                    const bool desc = ((i & k) == 0);
                                    if (desc ? (a < bb) : (a > bb)) { keys[i] = bb; keys[ixj] = a; }
                                }
                            }
                            __syncthreads();
                        }
                    }
                    for (int i = tid; i < nC; i += nt) c.S[i] = lastChildren - 1 - (int)(keys[i] & 0xFFFFFFFFull);
                }
                __syncthreads();
                OCT_T(21);
                // first j with size + sum_{i<=j}(children_i - 1) >= N
                // (a split never shrinks the list, so the running size crosses N at one entry only)
                int acc = size, jstar = -1;
                if (tid == 0) sh_a = -1;  // ordered before the first write below by the barrier of the scan
                for (int b = 0; b < nC && jstar < 0; b += nt) {
                    const int e = b + tid;
                    int gain = 0;
                    if (e < nC) {
                        const int i = c.S[e];
                        int k0, k1, k2, k3;
                        unpack_kids(c.kc(cur)[i], qn_count(c.nl(cur)[i]), &k0, &k1, &k2, &k3);
                        gain = (k0 > 0) + (k1 > 0) + (k2 > 0) + (k3 > 0) - 1;
                    }
                    int total;
                    const int ex = block_scan1(gain, &total, scan_lds, c.par);
                    if (e < nC && acc + ex < N && acc + ex + gain >= N) sh_a = e;
                    __syncthreads();
                    jstar = sh_a;
                    acc += total;
                }
                const int nS2 = jstar >= 0 ? jstar + 1 : nC;
                OCT_T(23);
                oct_scan_prefix(c, cur, size, nS2, &nchild, &nexp);
                __syncthreads();
                oct_scatter(c, cur, nS2, nchild);
                size = nchild + (size - nS2);
                cur ^= 1;
                OCT_T(24);
                if (size >= N || size == prev2) { finish = true; break; }
                lastChildren = nchild;
            }
        }
    }

    // ---- best response per node, list order (:975-1003) --------------------------
    if (size > g.kp_cap) {
        if (tid == 0) atomicOr(status, 1);
        size = g.kp_cap;
    }
    // eight lanes per node.  Highest response; among equals the first in the reference's visiting order: cell row,
    // cell column ((y - 3) / hCell and (x - 3) / wCell: a cell's keypoints start 3 px inside it, :1090-1124), y, x
    const uint32_t mx = ((1u << 20) + (uint32_t)g.wCell - 1u) / (uint32_t)g.wCell,  // exact quotients below 4096 for
                   my = ((1u << 20) + (uint32_t)g.hCell - 1u) / (uint32_t)g.hCell;  // divisors up to 256
    constexpr unsigned long long kRankMask = (1ull << 38) - 1ull;
    for (int e = (tid >> 3); e < size; e += (nt >> 3)) {
        const QNode n = c.nl(cur)[e];
        const int cnt = qn_count(n);
        const uint32_t *pts = c.buf(qn_buf(n)) + n.start;
        unsigned long long best = 0;
        for (int i = tid & 7; i < cnt; i += 8) {
            const uint32_t p = pts[i], x = (uint32_t)cand_x(p), y = (uint32_t)cand_y(p);
            const uint32_t cx = ((x - 3u) * mx) >> 20, cy = ((y - 3u) * my) >> 20;
            const unsigned long long rank = ((unsigned long long)cy << 31) | ((unsigned long long)cx << 24) | (y << 12) | x;
            const unsigned long long key = ((unsigned long long)cand_s(p) << 38) | (kRankMask - rank);
            best = key > best ? key : best;
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            best = other > best ? other : best;
        }
        if ((tid & 7) == 0) {
            const unsigned long long rank = kRankMask - (best & kRankMask);
            out_sel[e] = pack_cand((int)(rank & 0xFFFu), (int)((rank >> 12) & 0xFFFu), (int)(best >> 38));
        }
    }
    if (tid == 0) *out_count = size;
#ifdef ORBGPU_OCT_TRACE
    OCT_T(30);
    if (f == 0 && tid == 0) printf("level %d: %6.2f us, M %d size %d N %d\n", level, (double)(wall_clock64() - tr_begin) * 0.01, M, size, N);
    if (trace_on) {
        printf("octree trace: M %d size %d N %d\n", M, size, N);
        for (int i = 1; i < ntr; i++) printf("  phase %2d  +%5.2f us\n", tr_id[i], (double)(tr_t[i] - tr_t[i - 1]) * 0.01);
    }
#endif
#undef OCT_T
}

#define OCTREE_ARGS                                                                                                          \
    Geom G, const uint32_t *__restrict__ slots, const int32_t *__restrict__ cell_counts, uint32_t *cand, QNode *nodes,       \
        int32_t *scratch, uint32_t *__restrict__ sel, int32_t *__restrict__ sel_count, int32_t *__restrict__ cand_count,     \
        int32_t *__restrict__ status, int keys_cap, int lds_nodes, int dyn_coff_off, int pts_cap, int dyn_pts_off
// one frame: a few 1024-thread workgroups, each with most of a CU's LDS (candidate buffers included)
__device__ __forceinline__ void octree_single(const int level, const int f, const Geom &G, const uint32_t *__restrict__ slots,
                                              const int32_t *__restrict__ cell_counts, uint32_t *cand, QNode *nodes, int32_t *scratch,
                                              uint32_t *__restrict__ sel, int32_t *__restrict__ sel_count,
                                              int32_t *__restrict__ cand_count, int32_t *__restrict__ status, int keys_cap,
                                              int lds_nodes, int dyn_coff_off, int pts_cap, int dyn_pts_off)
{
    bool lds_pts = false;
    if (pts_cap > 0 && lds_nodes) {  // do this level's candidates fit the LDS left for them?
        __shared__ int fit_scan[64];
        const LevelGeom &g = G.lv[level];
        const int32_t *cc = cell_counts + (int64_t)f * G.total_cells + g.cell_off;
        int mine = 0, total, par = 0;
        for (int ci = threadIdx.x; ci < g.nCols * g.nRows; ci += blockDim.x) mine += cc[ci];
        (void)block_scan1(mine, &total, fit_scan, par);
        lds_pts = total <= pts_cap;
    }
    if (lds_pts)
        octree_level<true, true>(level, f, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, keys_cap,
                                 dyn_coff_off, pts_cap, dyn_pts_off);
    else if (lds_nodes)
        octree_level<true, false>(level, f, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, keys_cap,
                                  dyn_coff_off, 0, 0);
    else
        octree_level<false, false>(level, f, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, keys_cap,
                                   dyn_coff_off, 0, 0);
}
__global__ __launch_bounds__(1024) void k_octree(OCTREE_ARGS)
{
    octree_single(blockIdx.x, blockIdx.y, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, keys_cap, lds_nodes,
                  dyn_coff_off, pts_cap, dyn_pts_off);
}
// a batch: small workgroups, many per CU.  The grid is (frames, levels): workgroups start in linear order, so the
// levels with the most candidates (the lowest) start first and the short ones fill the tail.  The size follows the
// work of a level: 128 threads (LDS lets nine share a CU: five waves per SIMD, hence the register budget) for
// levels of a few hundred features -- fewer waves then wait at every barrier of the subdivision rounds (TUM frames:
// 256 threads with registers capped at 64 measured 7 % slower, 64 threads 20 %); for levels of thousands of features
// (3840 x 2160, 8000 features: 20 000 candidates on level 0) 256 threads when the launch has 2048 workgroups or
// more, else 512 (64 such frames: 1.30 / 0.81 / 0.49 ms with 128 / 256 / 512 threads).
template <int kThreads, int kWavesPerSimd>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void k_octree_batch(OCTREE_ARGS)
{
    if (lds_nodes)
        octree_level<true, false>(blockIdx.y, blockIdx.x, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, keys_cap,
                                  dyn_coff_off, 0, 0);
    else
        octree_level<false, false>(blockIdx.y, blockIdx.x, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, keys_cap,
                                   dyn_coff_off, 0, 0);
}

// ---------------------------------------------------------------------------
// GaussianBlur 7x7, sigma 2, 8-bit fixed point (called at :1502).  All levels
// in one launch; the halo rows of the source supply the reflect-101 border above and below.
// Tile = 64 x 32 pixels per one-wave workgroup; every thread moves and
// computes four horizontally adjacent pixels (32-bit global accesses, 128-bit
// LDS accesses for the 16-bit row sums).
// ---------------------------------------------------------------------------
constexpr int kBW = 64, kBH = 32;        // output tile of one blur workgroup
constexpr int kBlurThreads = (kBW / 4) * (kBH / 8);  // a thread of the vertical pass: 4 columns x 8 rows; = one wave
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t w, uint32_t acc)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, a), __builtin_bit_cast(ushort2v, w), acc, false);
}

// Horizontal 7-tap sums of the four pixels that start at byte 4 of (a, b, d): pixel j
// needs bytes 1+j .. 7+j, taken as two 4-byte windows for two byte dot products.
__device__ __forceinline__ void blur_row4(uint32_t a, uint32_t b, uint32_t d, uint32_t klo, uint32_t khi, uint32_t (&o)[4])
{
    o[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d, b, 1), khi, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 1), klo, 0u, false), false);
    o[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d, b, 2), khi, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 2), klo, 0u, false), false);
    o[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(d, b, 3), khi, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(b, a, 3), klo, 0u, false), false);
    o[3] = __builtin_amdgcn_udot4(d, khi, __builtin_amdgcn_udot4(b, klo, 0u, false), false);
}

// {sat_u8(x[15:0]), sat_u8(x[31:16])} as the low 16 bits (signed 16-bit inputs saturate to 0 .. 255)
__device__ __forceinline__ uint32_t sat_pk_u8(uint32_t x)
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(x));
    return r & 0xFFFFu;
}

__device__ __forceinline__ uint32_t u4_at(const uint4 &v, int q) { return q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w; }

// Horizontal pass of one tile straight from global memory (rows ty0-3 .. ty0+kBH+2, clamped into the halo): a
// thread takes two source rows x four columns (three aligned dwords per row: neighbours' loads overlap in L1) and
// writes four packed dwords.  EDGE: the tile touches the left or the right edge of the level.
template <bool EDGE>
__device__ __forceinline__ void blur_rows(const uint8_t *__restrict__ in, const LevelGeom &g, int tx0, int ty0, uint32_t klo,
                                          uint32_t khi, uint32_t *hpair, const int tid)
{
    const int xlim = (g.w + 15) & ~3, ylim = g.h + 2;
    constexpr int CG = kBW / 4;  // column groups of a tile row
    const bool left_edge = tx0 == 0, right_edge = g.w < tx0 + kBW + 4;  // uniform
    if (!EDGE && ty0 + kBH + 2 <= ylim) {
        // an inner tile above the level's last rows: no coordinate is clamped, so a lane's source address advances by a
        // constant from one of its row pairs to the next and the three dwords of a row are immediate offsets of one
        // pointer (the loop below derives and clamps every coordinate anew: half of its instructions)
        static_assert(kBlurThreads % CG == 0, "a lane keeps its column group");
        constexpr int kStep = kBlurThreads / CG;  // row pairs between a lane's tasks
        const int c = tid % CG;
        const uint8_t *r0 = in + (int64_t)(ty0 + 2 * (tid / CG) - 3) * g.pitch + (tx0 - 4 + 4 * c);
        const int64_t step = (int64_t)(2 * kStep) * g.pitch;
        uint32_t *dst = hpair + (tid / CG) * kBW + 4 * c;
        for (int rp = tid / CG; rp < (kBH + 6) / 2; rp += kStep, r0 += step, dst += kStep * kBW) {
            const uint8_t *r1 = r0 + g.pitch;
            const uint32_t a0 = *reinterpret_cast<const uint32_t *>(r0), b0 = *reinterpret_cast<const uint32_t *>(r0 + 4),
                           d0 = *reinterpret_cast<const uint32_t *>(r0 + 8);
            const uint32_t a1 = *reinterpret_cast<const uint32_t *>(r1), b1 = *reinterpret_cast<const uint32_t *>(r1 + 4),
                           d1 = *reinterpret_cast<const uint32_t *>(r1 + 8);
            uint32_t e[4], o[4];
            blur_row4(a0, b0, d0, klo, khi, e);
            blur_row4(a1, b1, d1, klo, khi, o);
            *reinterpret_cast<uint4 *>(dst) = make_uint4(e[0] | (o[0] << 16), e[1] | (o[1] << 16), e[2] | (o[2] << 16), e[3] | (o[3] << 16));
        }
        return;
    }
    for (int i = tid; i < (kBH + 6) / 2 * CG; i += kBlurThreads) {
        const int rp = i / CG, c = i % CG;
        const int y0 = min(ty0 + 2 * rp - 3, ylim), y1 = min(ty0 + 2 * rp - 2, ylim);
        const int xa = min(tx0 - 4 + 4 * c, xlim), xb = min(tx0 + 4 * c, xlim), xd = min(tx0 + 4 + 4 * c, xlim);
        const uint8_t *r0 = in + (int64_t)y0 * g.pitch, *r1 = in + (int64_t)y1 * g.pitch;
        // right edge: columns w .. w+3 = columns w-2, w-3, w-4, w-5; w is arbitrary, so these come out of the two
        // aligned dwords that hold them, loaded by every lane of an edge tile together with the tile's own dwords.
        const int hx = (g.w - 5) & ~3, hs = (g.w - 5) & 3;
        uint32_t hl0 = 0, hh0 = 0, hl1 = 0, hh1 = 0;
        if (EDGE && right_edge) {
            hl0 = *reinterpret_cast<const uint32_t *>(r0 + hx); hh0 = *reinterpret_cast<const uint32_t *>(r0 + hx + 4);
            hl1 = *reinterpret_cast<const uint32_t *>(r1 + hx); hh1 = *reinterpret_cast<const uint32_t *>(r1 + hx + 4);
        }
        uint32_t a0 = *reinterpret_cast<const uint32_t *>(r0 + xa), b0 = *reinterpret_cast<const uint32_t *>(r0 + xb),
                 d0 = *reinterpret_cast<const uint32_t *>(r0 + xd);
        uint32_t a1 = *reinterpret_cast<const uint32_t *>(r1 + xa), b1 = *reinterpret_cast<const uint32_t *>(r1 + xb),
                 d1 = *reinterpret_cast<const uint32_t *>(r1 + xd);
        if (EDGE && left_edge && c == 0) {  // columns -4 .. -1 = columns 4, 3, 2, 1
            a0 = __builtin_amdgcn_perm(d0, b0, 0x01020304u);
            a1 = __builtin_amdgcn_perm(d1, b1, 0x01020304u);
        }
        if (EDGE && right_edge) {
            const uint32_t h0 = __builtin_amdgcn_perm(0u, __builtin_amdgcn_alignbyte(hh0, hl0, hs), 0x00010203u);
            const uint32_t h1 = __builtin_amdgcn_perm(0u, __builtin_amdgcn_alignbyte(hh1, hl1, hs), 0x00010203u);
            int tr = g.w - xb;  // interior columns from this group's first one to the edge
            asm("" : "+v"(tr));  // opaque: the conditions below do not change along the loop, and a branch on them
                                 // would put the loads of h behind a second wait
            const int n = 4 - (tr & 3), sh = (8 * n) & 31;
            const bool in_b = tr >= 1 && tr <= 3;  // the edge lies in the middle dword: its first tr bytes stay
            const bool in_d = tr >= 5 && tr <= 7;  // the edge lies in the third dword: its first tr-4 bytes stay
            const uint32_t mb0 = __builtin_amdgcn_alignbyte(h0, b0 << sh, n), mb1 = __builtin_amdgcn_alignbyte(h1, b1 << sh, n);
            const uint32_t md0 = __builtin_amdgcn_alignbyte(h0, d0 << sh, n), md1 = __builtin_amdgcn_alignbyte(h1, d1 << sh, n);
            // the third dword: continues the reflection behind a patched middle dword, is the reflection (tr = 4),
            // or keeps its first bytes
            d0 = in_b ? h0 >> sh : tr == 4 ? h0 : in_d ? md0 : d0;
            d1 = in_b ? h1 >> sh : tr == 4 ? h1 : in_d ? md1 : d1;
            b0 = in_b ? mb0 : b0;
            b1 = in_b ? mb1 : b1;
        }
        uint32_t e[4], o[4];
        blur_row4(a0, b0, d0, klo, khi, e);
        blur_row4(a1, b1, d1, klo, khi, o);
        *reinterpret_cast<uint4 *>(hpair + rp * kBW + 4 * c) = make_uint4(e[0] | (o[0] << 16), e[1] | (o[1] << 16), e[2] | (o[2] << 16), e[3] | (o[3] << 16));
    }
}

// One wave per 64 x 32 tile: narrow tiles waste few lanes on the level widths, and the 19 x 16 horizontal tasks
// fill five passes of the wave almost completely.
constexpr int kBlurLdsWords = (kBH + 6) / 2 * kBW;  // LDS of one tile (one wave)
// The tile tt (level | tile column << 8 | tile row << 20) of frame f by one wave; tid: the lane; hpair: the wave's
// kBlurLdsWords of LDS.  Every wave of the workgroup must call it together (one barrier inside); active = false
// for a wave that only keeps the others company.
__device__ __forceinline__ void blur_tile(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, const Geom &G, const int f,
                                          const uint32_t tt, uint32_t *hpair, const int tid, const bool active)
{
    if (!active) {
        __syncthreads();
        return;
    }
    const LevelGeom &g = G.lv[tt & 0xFF];
    const int ty0 = (int)(tt >> 20) * kBH, tx0 = (int)((tt >> 8) & 0xFFF) * kBW;
    // taps (:1487, cv::GaussianBlur 7x7 sigma 2 in 8-bit fixed point) as byte / 16-bit dot-product operands
    const uint32_t k0 = 18, k1 = 34, k2 = G.gauss_variant ? 48 : 49, k3 = G.gauss_variant ? 56 : 55;
    const uint32_t klo = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), khi = k2 | (k1 << 8) | (k0 << 16);
    const uint8_t *in = pyr + (int64_t)f * G.pyr_frame_bytes + interior_off(g);
    // The halo rows hold the reflect-101 image of the interior (the producers write them); the three reflected
    // columns on either side are formed here from the row's own interior, in the tiles that touch an edge (their
    // own copy of the loop: the other tiles run without the checks).
    if (tx0 == 0 || g.w < tx0 + kBW + 4) blur_rows<true>(in, g, tx0, ty0, klo, khi, hpair, tid);
    else blur_rows<false>(in, g, tx0, ty0, klo, khi, hpair, tid);
    __syncthreads();
    // vertical pass: thread = 4 columns x 8 rows; output row y0+rr sums source rows y0+rr .. y0+rr+6,
    // i.e. four row pairs with the taps aligned for an even or an odd first row
    constexpr int CG = kBW / 4;  // column groups of a tile row
    const int c = tid % CG, rg = tid / CG;
    const int x = tx0 + 4 * c;
    uint8_t *out = blur + (int64_t)f * G.pyr_frame_bytes + g.plane_off;  // tiled: blur_tiled_off
    const uint32_t we0 = k0 | (k1 << 16), we1 = k2 | (k3 << 16), we2 = k2 | (k1 << 16), we3 = k0;
    const uint32_t wo0 = k0 << 16, wo1 = k1 | (k2 << 16), wo2 = k3 | (k2 << 16), wo3 = k1 | (k0 << 16);
    uint4 h[7];
#pragma unroll
    for (int k = 0; k < 7; k++) h[k] = *reinterpret_cast<const uint4 *>(hpair + (rg * 4 + k) * kBW + 4 * c);
    uint32_t rows8[8];
#pragma unroll
    for (int rr = 0; rr < 8; rr++) {
        const int b = rr >> 1;
        uint32_t sacc[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t p0 = u4_at(h[b], q), p1 = u4_at(h[b + 1], q), p2 = u4_at(h[b + 2], q), p3 = u4_at(h[b + 3], q);
            if (rr & 1) sacc[q] = dot2_u16(p3, wo3, dot2_u16(p2, wo2, dot2_u16(p1, wo1, dot2_u16(p0, wo0, 32768u))));
            else sacc[q] = dot2_u16(p3, we3, dot2_u16(p2, we2, dot2_u16(p1, we1, dot2_u16(p0, we0, 32768u))));
        }
        // the four results are the high halves (<= 257 with the 49 / 55 taps): two permutes gather them as 16-bit
        // pairs, the pack instruction saturates each pair into two bytes
        rows8[rr] = sat_pk_u8(__builtin_amdgcn_perm(sacc[1], sacc[0], 0x07060302u)) |
                    (sat_pk_u8(__builtin_amdgcn_perm(sacc[3], sacc[2], 0x07060302u)) << 16);
    }
    // the thread's 4 columns x 8 rows are 32 contiguous bytes of a tile (the first row is a multiple of 8); rows past
    // the level's last one and the up to 3 columns past its width land in unused bytes of the level's block
    const int y0 = ty0 + rg * 8;
    if (x < g.w && y0 < g.h) {
        uint4 *dst = reinterpret_cast<uint4 *>(out + blur_tiled_off(g.pitch, x, y0));
        dst[0] = make_uint4(rows8[0], rows8[1], rows8[2], rows8[3]);
        dst[1] = make_uint4(rows8[4], rows8[5], rows8[6], rows8[7]);
    }
}

__global__ __launch_bounds__(kBlurThreads) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur, Geom G,
                                                      const uint32_t *__restrict__ tile_tab)
{
    // horizontal sums (16 bit) of source rows 2i and 2i+1 share a dword: hpair[i][x] = h[2i][x] | h[2i+1][x] << 16
    __shared__ __align__(16) uint32_t hpair[kBlurLdsWords];
    int f, t;
    if (!frame_block(G.nimg, &f, &t)) return;
    // the host tabulates the tiles: level | tile column << 8 | tile row << 20 (no searches or divisions here)
    blur_tile(pyr, blur, G, f, tile_tab[t], hpair, threadIdx.x, true);
}

// One or two frames: the quadtree's few long workgroups (blockIdx.x < levels) and the blur's tiles (one per wave of
// a workgroup, in the dynamic LDS) in one launch.  Both depend on the pyramid and FAST only; as two
// launches on one stream the blur's few microseconds come on top of the quadtree's, and a second stream costs
// more than that in hand-overs.
__global__ __launch_bounds__(1024) void k_octree_blur(OCTREE_ARGS, const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                                      const uint32_t *__restrict__ tile_tab, int ntiles)
{
    if ((int)blockIdx.x < G.nlevels) {
        octree_single(blockIdx.x, blockIdx.y, G, slots, cell_counts, cand, nodes, scratch, sel, sel_count, cand_count, status, keys_cap,
                      lds_nodes, dyn_coff_off, pts_cap, dyn_pts_off);
        return;
    }
    extern __shared__ __align__(16) unsigned char dyn_lds_blur[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = ((int)blockIdx.x - G.nlevels) * (int)(blockDim.x >> 6) + wv;
    uint32_t *hpair = reinterpret_cast<uint32_t *>(dyn_lds_blur) + wv * kBlurLdsWords;
    blur_tile(pyr, blur, G, blockIdx.y, t < ntiles ? tile_tab[t] : 0u, hpair, threadIdx.x & 63, t < ntiles);
}
#undef OCTREE_ARGS

// ---------------------------------------------------------------------------
// Orientation (IC_Angle, :85-130), steered BRIEF (:142-191) and the final
// keypoint record (:1120-1135, :1511-1517).  One wave per retained keypoint.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    // cv::fastAtan2: float polynomial without fused multiply-add
    const float rad2deg = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * rad2deg, p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg, p7 = -0.04432655554792128f * rad2deg;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, cq, c2;
    if (ax >= ay) {
        cq = __fdiv_rn(ay, __fadd_rn(ax, 2.220446049250313e-16f));
        c2 = __fmul_rn(cq, cq);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), cq);
    } else {
        cq = __fdiv_rn(ax, __fadd_rn(ay, 2.220446049250313e-16f));
        c2 = __fmul_rn(cq, cq);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), cq));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

// sin and cos of a float angle in [0, 2*pi], evaluated in double (error below one
// double ulp) and rounded to float -- the correctly rounded float result except
// in vanishingly rare near-tie cases, i.e. what a good libm sinf/cosf returns.
// Quadrant reduction with a two-part pi/2, then the classic degree-13/14
// minimax kernels on |r| <= pi/4.
__device__ __forceinline__ void sincos_rounded(float x, float *s, float *c)
{
    const double xd = (double)x;
    const double k = rint(xd * 6.36619772367581382433e-01);
    double r = __fma_rn(-k, 1.57079632673412561417e+00, xd);
    r = __fma_rn(-k, 6.07710050650619224932e-11, r);
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double sr = r + r * z * ps;
    const double cr = 1.0 - 0.5 * z + z * z * pc;
    const int q = (int)k & 3;
    const double sv = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
    const double cv = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
    *s = (float)sv;
    *c = (float)cv;
}

// Sum over the wave with DPP lane permutations; the result is uniform.
__device__ __forceinline__ int wave_sum_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

struct __attribute__((aligned(4))) DwordPair { uint32_t lo, hi; };  // one 64-bit load at 32-bit alignment

constexpr int kDescGroup = 8;   // keypoint slots per wave in batches (a few frames: kDescGroupFew, shorter chains on more waves)
constexpr int kDescGroupFew = 2;
constexpr int kBR = 18;         // largest rotated pattern coordinate: round(hypot(13, 13))
constexpr int kBPitch = 40;     // bytes per staged row of the blurred patch: 37 + up to 3 of alignment

template <int kGroup>
__global__ __launch_bounds__(64) void k_describe(const uint8_t *__restrict__ pyr, const uint8_t *__restrict__ blur, Geom G,
                                                  const uint32_t *__restrict__ sel, const int32_t *__restrict__ sel_count,
                                                  const int8_t *__restrict__ pattern, orbgpu_keypoint *__restrict__ kps,
                                                  uint8_t *__restrict__ desc, int32_t *__restrict__ counts, int capacity,
                                                  int32_t *__restrict__ status)
{
    // One wave per workgroup (waves never cooperate here; four-wave workgroups take the same time alone but pack
    // worse beside the other streams' kernels: 303 k against 309 k frames/s for the whole step).
    // A wave owns kGroup consecutive keypoint slots.  Stage A: intensity
    // centroid of each (aligned 32-bit loads, dot products against per-lane
    // weights).  Stage B: lane j turns the moments of keypoint j into angle,
    // sine and cosine -- once for the whole group.  Stage C: the blurred patch
    // of each keypoint goes through LDS and the 256 tests read it from there.
    __shared__ __align__(16) uint8_t patch[1][(2 * kBR + 1) * kBPitch];
    int f, blk;
    if (!frame_block(G.nimg, &f, &blk)) return;
    const int lane = lane_id();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot0 = (blk * 1 + wv) * kGroup;
    if (slot0 >= G.total_kp_cap) return;
    const int32_t *sc = sel_count + f * G.nlevels;
#ifdef ORBGPU_STAMPS
    unsigned long long dstamp_prev = __builtin_readcyclecounter();
    const unsigned dstamp_id = kStampWgs / 2 + ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6));  // upper half: FAST uses the lower
#define ORB_DSTAMP(k_) do { if ((threadIdx.x & 63) == 0 && dstamp_id < (unsigned)kStampWgs) { const unsigned long long now_ = __builtin_readcyclecounter(); g_fast_stamps[dstamp_id * 8 + k_] += (unsigned)(now_ - dstamp_prev); dstamp_prev = now_; } } while (0)
#else
#define ORB_DSTAMP(k_) do { } while (0)
#endif
    // the per-level counts of the frame: uniform loads, all in flight at once (entries past nlevels belong to later
    // frames or to the padding behind the last one: masked wherever they are used)
    int n[ORBGPU_MAX_LEVELS];
#pragma unroll
    for (int k = 0; k < ORBGPU_MAX_LEVELS; k++) n[k] = sc[k];
    {
        // most slots are empty (a level rarely fills its quota): leave on scalar arithmetic alone when none of
        // this wave's slots holds a keypoint
        int any = slot0 == 0;  // the wave of slot 0 also writes the frame's count
#pragma unroll
        for (int k = 0; k < ORBGPU_MAX_LEVELS; k++)
            any |= (k < G.nlevels && max(slot0, G.t_kp_off[k]) < min(slot0 + kGroup, G.t_kp_off[k] + n[k])) ? 1 : 0;
        if (!any) return;
    }

    // lane j < kGroup: geometry of slot slot0 + j (invalid slots compute on pixel (19, 19) of
    // level 0 and store nothing); every table below is read with uniform, independent loads
    int my_l = 0, my_valid = 0, my_oi = 0, my_x = kEdge, my_y = kEdge, my_resp = 0;
    int my_pitch, my_off, my_patch, my_plane;
    float my_scale;
    {
        const int slot = slot0 + (lane & (kGroup - 1));
        int l = 0;
#pragma unroll
        for (int k = 1; k < ORBGPU_MAX_LEVELS; k++) l += slot >= G.t_kp_off[k] ? 1 : 0;  // unused levels hold INT_MAX
        int total = 0, offl = 0, nl = 0, kpo = 0;
#pragma unroll
        for (int k = 0; k < ORBGPU_MAX_LEVELS; k++) {
            const int nk = k < G.nlevels ? n[k] : 0;
            total += nk;
            offl += k < l ? nk : 0;
            if (k == l) { nl = nk; kpo = G.t_kp_off[k]; }
        }
        if (slot0 == 0 && lane == 0) {
            counts[f] = min(total, capacity);
            if (total > capacity) atomicOr(status, 2);
        }
        const int i = slot - kpo;
        my_oi = offl + i;
        if (slot < G.total_kp_cap && i < nl && my_oi < capacity) {
            const uint32_t p = sel[(int64_t)f * G.total_kp_cap + slot];
            my_l = l;
            my_valid = 1;
            my_x = cand_x(p) + kMinBorder;  // level coordinates (:1129-1130)
            my_y = cand_y(p) + kMinBorder;
            my_resp = cand_s(p);
        }
        my_pitch = G.t_pitch[0]; my_off = G.t_interior[0]; my_patch = G.t_patch[0]; my_scale = G.t_scale[0];
#pragma unroll
        for (int k = 1; k < ORBGPU_MAX_LEVELS; k++)
            if (k == my_l) { my_pitch = G.t_pitch[k]; my_off = G.t_interior[k]; my_patch = G.t_patch[k]; my_scale = G.t_scale[k]; }
        my_plane = my_off - kEdge * my_pitch - kLeftMargin;  // the level's block: the blurred level is stored in tiles
        my_off += my_y * my_pitch + my_x;
    }
    if (__ballot(my_valid && lane < kGroup) == 0) return;
    ORB_DSTAMP(1);
    const uint8_t *fpyr = pyr + (int64_t)f * G.pyr_frame_bytes, *fblur = blur + (int64_t)f * G.pyr_frame_bytes;

    // ---- stage A: m10 = sum u*I, m01 = sum v*I over the disc (:85-130) -----------------
    // the 31 x 32 px bounding box is 248 dwords (v = -15..15, u = -15+4j .. -12+4j); lane owns 4 of them
    int icv[4], icu0[4];
    uint32_t wU[4], wM[4];
    {
        const unsigned long long umax_nibbles = 0x3689ABCDDEEEFFFFull;  // umax[v] = nibble v: 15 15 15 15 14 14 14 13 13 12 11 10 9 8 6 3
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int p = lane + 64 * m;
            const bool live = p < 248;
            const int v = live ? (p >> 3) - 15 : 0, u0 = live ? -15 + 4 * (p & 7) : -15;  // idle lanes re-read an aligned dword, weight 0
            const int um = (int)((umax_nibbles >> (4 * (v < 0 ? -v : v))) & 15ull);
            uint32_t a = 0, b = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int u = u0 + k;
                const bool in = live && u >= -um && u <= um;
                a |= (in ? (uint32_t)(u + 15) : 0u) << (8 * k);
                b |= (in ? 1u : 0u) << (8 * k);
            }
            icv[m] = v; icu0[m] = u0; wU[m] = a; wM[m] = b;
        }
    }
    int my_m10 = 0, my_m01 = 0;
#pragma unroll 2
    for (int j = 0; j < kGroup; j++) {
        const int pitch = __builtin_amdgcn_readlane(my_pitch, j), off = __builtin_amdgcn_readlane(my_off, j);
        const int sh = (off - 15) & 3;  // planes are 4-byte aligned: the alignment of x - 15 is that of its byte offset
        const uint8_t *c = fpyr + off - sh;
        int sU = 0, sM = 0, sV = 0;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const DwordPair q = *reinterpret_cast<const DwordPair *>(c + icv[m] * pitch + icu0[m]);
            const uint32_t px = __builtin_amdgcn_alignbyte(q.hi, q.lo, sh);
            const int msum = (int)__builtin_amdgcn_udot4(px, wM[m], 0u, false);
            sU = (int)__builtin_amdgcn_udot4(px, wU[m], (uint32_t)sU, false);
            sM += msum;
            sV += icv[m] * msum;
        }
        const int m10 = wave_sum_dpp(sU - 15 * sM), m01 = wave_sum_dpp(sV);
        if (lane == j) { my_m10 = m10; my_m01 = m01; }
    }

    ORB_DSTAMP(2);
    // ---- stage B: orientation, once per group ---------------------------------------------
    const float my_angle = fast_atan2_deg((float)my_m01, (float)my_m10);
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.0);
    float my_a, my_b;
    sincos_rounded(__fmul_rn(my_angle, factorPI), &my_b, &my_a);
    if (lane < kGroup && my_valid) {
        orbgpu_keypoint k;
        const float fx = (float)my_x, fy = (float)my_y;
        k.x = my_l ? __fmul_rn(fx, my_scale) : fx;
        k.y = my_l ? __fmul_rn(fy, my_scale) : fy;
        k.size = (float)my_patch;
        k.angle = my_angle;
        k.response = (float)my_resp;
        k.octave = my_l;
        k.class_id = -1;
        kps[(int64_t)f * capacity + my_oi] = k;
    }

    ORB_DSTAMP(3);
    // ---- stage C: steered BRIEF on the blurred level (:142-191) -----------------------------
    char4 pt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) pt[k] = reinterpret_cast<const char4 *>(pattern)[k * 64 + lane];
    // staging: 37 rows x 10 dwords = 370 dwords, lane owns 6 of them
    int srow[6], scol[6];
#pragma unroll
    for (int m = 0; m < 6; m++) {
        // column-major over the patch: the blurred level is stored in 4-pixel-wide, 32-row tiles of one line each, so
        // the 64 dwords of one load instruction (1.7 columns of 37 rows) lie in three or four lines
        const int i = min(lane + 64 * m, (2 * kBR + 1) * (kBPitch / 4) - 1);
        scol[m] = i / (2 * kBR + 1);
        srow[m] = i - scol[m] * (2 * kBR + 1);
    }
    uint32_t *pw = reinterpret_cast<uint32_t *>(patch[wv]);
    const uint8_t *pb = patch[wv];
    uint32_t pre[6];
    // dword m of keypoint j's patch: columns (x - 18 rounded down to 4) + 4 scol .., row y - 18 + srow, in the tiled plane
    // blur_tiled_off(pitch, xb + 4 scol, yb + srow) with its uniform parts on the scalar side: the first column is a
    // multiple of 4, so the address is ((y >> 5) * (pitch >> 2) << 7) + ((y & 31) << 2) + a lane constant + a scalar
    int scol7[6];
#pragma unroll
    for (int m = 0; m < 6; m++) scol7[m] = scol[m] << 7;
    auto patch_loads = [&](int j, uint32_t (&dst)[6]) {
        const int pitch = __builtin_amdgcn_readlane(my_pitch, j), kx = __builtin_amdgcn_readlane(my_x, j);
        const int ky = __builtin_amdgcn_readlane(my_y, j);
        const int xb = (kx - kBR) & ~3, yb = ky - kBR;
        const uint8_t *c = fblur + __builtin_amdgcn_readlane(my_plane, j) + (((xb + kLeftMargin) >> 2) << 7);  // uniform
        const int k128 = (pitch >> 2) << 7;
#pragma unroll
        for (int m = 0; m < 6; m++) {
            const int y = yb + srow[m];
            dst[m] = *reinterpret_cast<const uint32_t *>(c + (uint32_t)((y >> 5) * k128 + ((y & 31) << 2) + scol7[m]));
        }
    };
    patch_loads(0, pre);
    // the pattern as floats, once per wave
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { px0[k] = (float)pt[k].x; py0[k] = (float)pt[k].y; px1[k] = (float)pt[k].z; py1[k] = (float)pt[k].w; }
    for (int j = 0; j < kGroup; j++) {
#pragma unroll
        for (int m = 0; m < 6; m++) pw[srow[m] * (kBPitch / 4) + scol[m]] = pre[m];
        const int ox = (__builtin_amdgcn_readlane(my_off, j) - kBR) & 3;
        if (j + 1 < kGroup) patch_loads(j + 1, pre);
        __builtin_amdgcn_wave_barrier();
        const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_a), j));
        const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_b), j));
        const uint8_t *ctr = pb + kBR * kBPitch + kBR + ox;
        int tap0[4], tap1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float x0 = px0[k], y0 = py0[k], x1 = px1[k], y1 = py1[k];
            const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
            const int q0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
            const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
            const int q1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
            tap0[k] = ctr[r0 * kBPitch + q0];
            tap1[k] = ctr[r1 * kBPitch + q1];
        }
        uint64_t bits[4];
#pragma unroll
        for (int k = 0; k < 4; k++) bits[k] = __ballot(tap0[k] < tap1[k]);
        __builtin_amdgcn_wave_barrier();
        const int valid = __builtin_amdgcn_readlane(my_valid, j), oi = __builtin_amdgcn_readlane(my_oi, j);
        if (valid && lane < 4) {
            const uint64_t w = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
            reinterpret_cast<uint64_t *>(desc + ((int64_t)f * capacity + oi) * 32)[lane] = w;
        }
    }
    ORB_DSTAMP(4);
}

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
constexpr int kWalkMinFrames = 128;  // from here on a workgroup per frame fills the chip

int launch_pyramid(hipStream_t st, const uint8_t *d_imgs, int nimg, int stride, size_t frame_stride, uint8_t *pyr,
                   const Geom &G_, const XTab *xtab, const YTab *ytab, const FuseArgs *fuse, int32_t *status_clear)
{
    // LDS window of one wave's 128 x 8 tile in k_pyr_walk: the largest over the levels, padded to whole 64-dword pieces
    int walk_lds_dw = 0;
    for (int l = 1; l < G_.nlevels; l++) {
        const int dw = (G_.lv[l].rs_rows8 * ((G_.lv[l].rs_pitch_dw + 3) & ~3) + 255 + 256) & ~255;  // rows of whole quads, whole 1 KiB pieces
        walk_lds_dw = dw > walk_lds_dw ? dw : walk_lds_dw;
        if (G_.lv[l].rs_rows8 <= 0) walk_lds_dw = -(1 << 30);
    }
    // the XCD-pinned frame mapping (nimg >= 8) pays off for the kernels that re-read shared rows
    // (resize, blur, describe); the streaming ones keep the plain mapping (nimg = 0)
    Geom G = G_;
    G.nimg = nimg;
    const LevelGeom &g0 = G.lv[0];
    hipLaunchKernelGGL(k_pyr_copy0, dim3((((g0.w + 15) / 16) * g0.h + 255) / 256, nimg), dim3(256), 0, st, d_imgs, stride, frame_stride, pyr, G_, status_clear);
    // a few frames: all levels above 0 in one launch (k_pyr_fused) when the host could tabulate the regions
    if (nimg <= 4 && G.nlevels > 1 && fuse && fuse->tab) {
        const LevelGeom &gl = G.lv[G.nlevels - 1];
        const size_t lds = (size_t)fuse->lds_entries * 24 + 2 * (size_t)fuse->lds_half;
        hipLaunchKernelGGL(k_pyr_fused, dim3(((gl.w + kFW - 1) / kFW) * ((gl.h + kFH - 1) / kFH), nimg), dim3(1024), lds, st, pyr, G, 1,
                           G.nlevels - 1, xtab, ytab, fuse->tab, fuse->lds_entries, fuse->lds_half);
        ORBGPU_HIP_CHECK(hipGetLastError());
        return ORBGPU_OK;
    }
    // a batch: one workgroup per frame walks the levels (one launch); the per-wave windows must fit the CU's LDS twice
    if (nimg >= kWalkMinFrames && G.nlevels > 1 && walk_lds_dw > 0 && (size_t)walk_lds_dw * 4 * kWalkWaves <= 64 * 1024) {
        hipLaunchKernelGGL(k_pyr_walk, dim3(nimg), dim3(kWalkWaves * 64), (size_t)walk_lds_dw * 4 * kWalkWaves, st, pyr, G, xtab, ytab, walk_lds_dw);
        ORBGPU_HIP_CHECK(hipGetLastError());
        return ORBGPU_OK;
    }
    for (int l = 1; l < G.nlevels; l++) {
        const LevelGeom &g = G.lv[l];
        hipLaunchKernelGGL(k_pyr_resize, frame_grid(((g.w + kRW - 1) / kRW) * ((g.h + kRH - 1) / kRH), nimg), dim3(kRH / 8 * 64),
                           ((size_t)g.rs_rows * g.rs_pitch_dw * 4 + 255) & ~(size_t)255, st, pyr, G, l, xtab, ytab);
    }
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

// All 19 halo columns of every level (the extraction writes none).
int launch_halo_full(hipStream_t st, int nimg, uint8_t *pyr, const Geom &G)
{
    int rows = 0;
    for (int l = 0; l < G.nlevels; l++) rows += (G.lv[l].h + 2 * kEdge + 15) / 16;
    Geom Gp = G;
    Gp.nimg = 0;  // plain (block, frame) mapping
    hipLaunchKernelGGL(k_halo_sides, dim3(rows, nimg), dim3(256), 0, st, pyr, Gp);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

#ifdef ORBGPU_STAMPS
extern "C" int orbgpu_debug_fast_stamps(unsigned *out, int nwg, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (nwg > kStampWgs) nwg = kStampWgs;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fast_stamps), (size_t)nwg * 32) != hipSuccess) return -1;
    if (reset) {
        void *p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_fast_stamps)) != hipSuccess || hipMemset(p, 0, sizeof(unsigned) * 8 * kStampWgs) != hipSuccess) return -1;
    }
    return 0;
}
#endif

int launch_fast(hipStream_t st, int nimg, const uint8_t *pyr, const Geom &G, const uint32_t *cell_tab, uint32_t *slots,
                int32_t *cell_counts, const LaunchTuning &tune)
{
    // LDS regions for the largest cell: rows of kTP bytes (all offsets multiples of 16)
    int wmax = 0, hmax = 0;
    for (int l = 0; l < G.nlevels; l++) {
        wmax = G.lv[l].wCell > wmax ? G.lv[l].wCell : wmax;
        hmax = G.lv[l].hCell > hmax ? G.lv[l].hCell : hmax;
    }
    // a tile row holds the cell image (cell + 6) behind one lead byte, in whole dwords
    const int kTP = wmax + 6 + 4 <= kTPNarrow ? kTPNarrow : kTPWide;
    // the direct global -> LDS copy writes 64 dwords per piece, a piece every 64 / (kTP / 4) rows: the last one runs
    // past the tile's last row
    const int rpp = 64 / (kTP / 4), tile_bytes = ((((hmax + 6 + rpp - 1) / rpp - 1) * rpp * kTP + 256) + 15) & ~15;
    const int score_pitch = (wmax + 2 + 3) & ~3, score_off = tile_bytes;
    // the region of the score plane first holds the group queue (a word per group of four interior pixels)
    const int score_bytes = ((hmax + 2) * score_pitch + 15) & ~15, gqueue_bytes = ((wmax + 3) / 4 * hmax * 4 + 15) & ~15;
    const int queue_off = score_off + (score_bytes > gqueue_bytes ? score_bytes : gqueue_bytes);
    // the survivor queue: all pixels of the largest cell if that keeps the workgroup within a 32nd of a CU's LDS (5120
    // bytes, 16 of them static), else what is left of that share but at least 256 entries
    constexpr int kLdsShare = 160 * 1024 / 32 - 16;
    int qcap = wmax * hmax;
    if (queue_off + 2 * qcap > kLdsShare) qcap = (kLdsShare - queue_off) / 2 > 256 ? (kLdsShare - queue_off) / 2 : 256;
    if (qcap > wmax * hmax) qcap = wmax * hmax;
    // test hook: a smaller queue sends ordinary cells down the dense path (tests/test_gpu_extract.py)
    if (tune.fast_queue_cap >= 1 && tune.fast_queue_cap < qcap) qcap = tune.fast_queue_cap;
    const size_t lds = (size_t)queue_off + (size_t)qcap * 2;
    // A batch: one wave per workgroup and runs of cells, so that a CU holds as many independent workgroups as its
    // LDS allows; a few frames: the cells spread over all CUs, four waves on each (a single 640 x 480 frame: 8.6 us
    // with two waves per cell, 7.2 with four; eight made the whole call 2 us slower than four)
    const bool batch = (int64_t)G.total_cells * nimg >= 16384;
    const int per_wg = batch ? kFastQuad : 1;
    const dim3 grid((G.total_cells + per_wg - 1) / per_wg, nimg);
#define ORB_FAST_LAUNCH(tp_, th_)                                                                                                     \
    hipLaunchKernelGGL((k_fast_cells<tp_, th_>), grid, dim3(th_), lds, st, pyr, G, cell_tab, slots, cell_counts, per_wg, score_off, \
                       score_pitch, queue_off, qcap)
    if (kTP == kTPNarrow) {
        if (batch) ORB_FAST_LAUNCH(kTPNarrow, 64);
        else ORB_FAST_LAUNCH(kTPNarrow, 256);
    } else {
        if (batch) ORB_FAST_LAUNCH(kTPWide, 64);
        else ORB_FAST_LAUNCH(kTPWide, 256);
    }
#undef ORB_FAST_LAUNCH
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

int launch_octree(hipStream_t st, int nimg, const Geom &G, const uint32_t *slots, const int32_t *cell_counts, uint32_t *cand,
                  QNode *nodes, int32_t *scratch, uint32_t *sel, int32_t *sel_count, int32_t *cand_count, int32_t *status,
                  const LaunchTuning &tune, const BlurWork *with_blur, bool *blur_done)
{
    if (blur_done) *blur_done = false;
    int maxcap = 0;
    for (int l = 0; l < G.nlevels; l++) maxcap = G.lv[l].node_cap > maxcap ? G.lv[l].node_cap : maxcap;
    int keys_cap = 2;
    while (keys_cap < maxcap) keys_cap <<= 1;  // the bitonic sort pads to a power of two: < 2 * node_cap keys
    if (keys_cap > 4096) keys_cap = 0;         // no room in LDS: the keys go to the level's global scratch
    const size_t node_bytes = (size_t)maxcap * (2 * sizeof(QNode) + 8 * sizeof(int32_t));
    const int lds_nodes = keys_cap && (size_t)keys_cap * 8 + node_bytes <= 56 * 1024;
    // one frame: few large workgroups (latency); a batch: many small ones so every CU holds several
    const bool batch = nimg * G.nlevels >= 256;
    // (one frame of 640 x 480 at 1000 features: 512 threads make the call 2.7 us shorter than 1024, 256 are no better than 1024)
    int threads = !batch ? (maxcap <= 512 ? 512 : 1024) : maxcap <= 512 ? 128 : nimg * G.nlevels >= 2048 ? 256 : 512;
    if (tune.oct_threads) threads = tune.oct_threads;  // measurement hook
    const size_t coff_off = (size_t)keys_cap * 8 + (lds_nodes ? node_bytes : 0);
    size_t dyn = coff_off + (size_t)((threads > kMaxRoots ? threads : kMaxRoots) + 1) * 4;
    if (batch) {
        if (threads == 128)
            hipLaunchKernelGGL((k_octree_batch<128, 5>), dim3(nimg, G.nlevels), dim3(threads), dyn, st, G, slots, cell_counts, cand, nodes,
                               scratch, sel, sel_count, cand_count, status, keys_cap, lds_nodes, (int)coff_off, 0, 0);
        else if (threads == 256)
            hipLaunchKernelGGL((k_octree_batch<256, 4>), dim3(nimg, G.nlevels), dim3(threads), dyn, st, G, slots, cell_counts, cand, nodes,
                               scratch, sel, sel_count, cand_count, status, keys_cap, lds_nodes, (int)coff_off, 0, 0);
        else
            hipLaunchKernelGGL((k_octree_batch<512, 4>), dim3(nimg, G.nlevels), dim3(threads), dyn, st, G, slots, cell_counts, cand, nodes,
                               scratch, sel, sel_count, cand_count, status, keys_cap, lds_nodes, (int)coff_off, 0, 0);
    } else {
        // the two candidate buffers of a level in LDS too, as far as 144 KiB go (a workgroup may use the whole CU's)
        const size_t pts_off = (dyn + 15) & ~(size_t)15, budget = 144 * 1024;
        int maxcand = 0;
        for (int l = 0; l < G.nlevels; l++) maxcand = G.lv[l].cand_cap > maxcand ? G.lv[l].cand_cap : maxcand;
        int pts_cap = pts_off + 64 < budget ? (int)((budget - pts_off) / 8) : 0;
        if (pts_cap > maxcand) pts_cap = maxcand;
        pts_cap &= ~3;
        if (pts_cap < 256) pts_cap = 0;
        if (pts_cap) dyn = pts_off + (size_t)pts_cap * 8;
        // one or two frames: the blur's tiles ride in the same launch (a tile and its LDS per wave of a workgroup)
        bool fuse_blur = with_blur && blur_done && with_blur->ntiles > 0;
        const int nwaves = threads / 64;
        const size_t blur_lds = (size_t)nwaves * kBlurLdsWords * 4;
        if (fuse_blur && blur_lds > dyn) dyn = blur_lds;
        // dynamic LDS beyond 64 KiB is asked for once per device and kernel (the attribute belongs to the device's copy
        // of the kernel); handles on several devices, or the two extractors of a stereo build on two host threads, may
        // get here at the same time
        if (dyn > 64 * 1024) {
            static std::atomic<bool> granted[2][64];
            int dev = -1;
            const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
            std::atomic<bool> *flag = known ? &granted[fuse_blur ? 1 : 0][dev] : nullptr;
            if (!(flag && flag->load(std::memory_order_acquire))) {
                const void *fn = fuse_blur ? reinterpret_cast<const void *>(k_octree_blur) : reinterpret_cast<const void *>(k_octree);
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget + 8192) == hipSuccess) {
                    if (flag) flag->store(true, std::memory_order_release);
                } else {
                    (void)hipGetLastError();
                    pts_cap = 0;
                    fuse_blur = false;
                    dyn = coff_off + (size_t)((threads > kMaxRoots ? threads : kMaxRoots) + 1) * 4;
                }
            }
        }
        if (fuse_blur) {
            hipLaunchKernelGGL(k_octree_blur, dim3(G.nlevels + (with_blur->ntiles + nwaves - 1) / nwaves, nimg), dim3(threads), dyn, st, G, slots, cell_counts,
                               cand, nodes, scratch, sel, sel_count, cand_count, status, keys_cap, lds_nodes, (int)coff_off, pts_cap, (int)pts_off,
                               with_blur->pyr, with_blur->blur, with_blur->tile_tab, with_blur->ntiles);
            *blur_done = true;
        } else {
            hipLaunchKernelGGL(k_octree, dim3(G.nlevels, nimg), dim3(threads), dyn, st, G, slots, cell_counts, cand, nodes, scratch, sel,
                               sel_count, cand_count, status, keys_cap, lds_nodes, (int)coff_off, pts_cap, (int)pts_off);
        }
    }
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

// Tile table of the blur: level | tile column << 8 | tile row << 20 for every kBW x kBH tile of every level.
int blur_tile_table(const Geom &G, std::vector<uint32_t> *tab)
{
    tab->clear();
    for (int l = 0; l < G.nlevels; l++)
        for (int ty = 0; ty < (G.lv[l].h + kBH - 1) / kBH; ty++)
            for (int tx = 0; tx < (G.lv[l].w + kBW - 1) / kBW; tx++) tab->push_back((uint32_t)l | ((uint32_t)tx << 8) | ((uint32_t)ty << 20));
    return (int)tab->size();
}

int launch_blur(hipStream_t st, int nimg, const uint8_t *pyr, uint8_t *blur, const Geom &G_, const uint32_t *tile_tab, int ntiles)
{
    Geom G = G_;
    G.nimg = nimg;
    hipLaunchKernelGGL(k_blur, frame_grid(ntiles, nimg), dim3(kBlurThreads), 0, st, pyr, blur, G, tile_tab);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

int launch_describe(hipStream_t st, int nimg, const uint8_t *pyr, const uint8_t *blur, const Geom &G_, const uint32_t *sel,
                    const int32_t *sel_count, const int8_t *pattern, orbgpu_keypoint *kps, uint8_t *desc, int32_t *counts,
                    int capacity, int32_t *status)
{
    Geom G = G_;
    G.nimg = nimg;
    if (nimg <= 2)
        hipLaunchKernelGGL(k_describe<kDescGroupFew>, frame_grid((G.total_kp_cap + kDescGroupFew - 1) / kDescGroupFew, nimg), dim3(64), 0, st,
                           pyr, blur, G, sel, sel_count, pattern, kps, desc, counts, capacity, status);
    else
        hipLaunchKernelGGL(k_describe<kDescGroup>, frame_grid((G.total_kp_cap + kDescGroup - 1) / kDescGroup, nimg), dim3(64), 0, st, pyr, blur,
                           G, sel, sel_count, pattern, kps, desc, counts, capacity, status);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

}  // namespace orbgpu
