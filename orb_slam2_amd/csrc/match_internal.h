// Argument blocks shared by the matcher kernels and their host launchers.
#pragma once
#include "orbgpu_internal.h"

namespace orbgpu {

constexpr int kTopK = 16;  // cached best candidates per windowed query
__host__ __device__ inline size_t match_scratch_ints(int q_cap, int t_cap)
{
    // cached candidates [kTopK*q_cap] | cell_start [3080] | grid records [4*t_cap] | grid descriptors [8*t_cap] | head [t_cap] |
    // ncand, sel x2, sd x2, nxt, full [q_cap each]; a multiple of 4 ints keeps every pair 16-byte aligned
    return (3080 + 13 * (size_t)t_cap + (kTopK + 7) * (size_t)q_cap + 3) & ~(size_t)3;
}

struct ScaleTab { float s[ORBGPU_MAX_LEVELS]; };

// mode 0..2: ORBGPU_MODE_*; mode 3: SearchForInitialization; mode 4: SearchByBoW (the candidates of a
// query are the train indices of one segment, seg_index[seg_start[s] .. seg_start[s+1]), s = q.minLevel);
// mode 5: SearchForTriangulation (segments as in mode 4, epipolar gates, queries independent);
// mode 6: best keypoint in a window per query, queries independent (Fuse x2, SearchBySim3), optional
// reprojection chi-square gate of Fuse(pKF, vpMapPoints, th).
struct MatchArgs {
    int32_t mode;
    const orbgpu_wquery *q;
    const uint8_t *qdesc;
    const int32_t *nq;
    int32_t q_cap;
    const orbgpu_keypoint *kps2;
    const uint8_t *desc2;
    const float *uright2;
    const int32_t *n2;
    int32_t t_cap;
    // 64x48 grid geometry: the train keypoints are binned with origin (gminX, gminY), the query cell ranges are
    // computed with origin (qminX, qminY), both with the inverse cell sizes invW / invH.  A Frame uses one origin
    // (src/Frame.cc:397-500); a KeyFrame keeps its Frame's grid and inverse cell sizes but queries with its own
    // integer-truncated bounds (include/KeyFrame.h:185, src/KeyFrame.cc:585-624).
    float gminX, gminY, qminX, qminY, invW, invH;
    // a grid built earlier with the same t_cap and build origin (one pair): cell_start [3080] | grid records
    // [4*t_cap] | grid descriptors [8*t_cap]; null: the call builds the grid in its scratch
    const int32_t *grid;
    // grid built by the call: keep only train keypoints of octave < grid_levels (0: all).  SearchForInitialization
    // asks GetFeaturesInArea for level 0 alone (src/ORBmatcher.cc:448-452), so the other levels' keypoints -- four
    // fifths of a window's entries -- need not enter its grid; their order among themselves is unchanged.
    int32_t grid_levels;
    const uint8_t *occupied;
    int32_t *assign;   // [pair][t_cap]
    int32_t *qmatch;   // [pair][q_cap] or null
    int32_t *scratch;  // [pair][match_scratch_ints(q_cap, t_cap)]
    int32_t th_dist;
    float nnratio;
    int32_t check_orientation;
    int32_t *nmatches;  // [pair]
    int32_t *stats;     // optional [pair][4]: resolver iterations, window rescans
    const int32_t *seg_start, *seg_index;  // modes 4 and 5 (one pair)
    // mode 5 (SearchForTriangulation): fundamental matrix (row-major), epipole in the train image,
    // per-level scale factors and sigma^2 of the train keyframe, bOnlyStereo
    float F12[9], ex, ey;
    ScaleTab scale2, sigma2;
    int32_t only_stereo;
    // mode 6: chi-square gate on/off and mvInvLevelSigma2 of the keyframe
    int32_t chi2_gate;
    ScaleTab inv_sigma2;
};

#ifdef __HIPCC__
// 256-bit Hamming distance (ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:1710-1727)
__device__ __forceinline__ int hamming8(const uint32_t (&a)[8], const uint32_t *__restrict__ b)
{
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d += __popc(a[k] ^ b[k]);
    return d;
}
#endif

// Matrix-core brute force (bf_mfma_kernels.hip): sizes of one call -- query tiles of 32 per wave, padded counts, the
// chunking of the train set and the layout of the workspace (byte-per-bit images, accumulator constants, partial results).
struct BfMfmaPlan {
    int nq_tile;
    int nq_pad, nt_pad, ntiles, tiles_per_chunk, nchunks, chunk;
    size_t off_qx, off_tx, off_cinit, off_part, bytes;
};
BfMfmaPlan bf_mfma_plan(int nq, int nt);
int launch_bruteforce_mfma(hipStream_t st, const BfMfmaPlan &P, const uint8_t *d_Q, int nq, const uint8_t *d_T, int nt, uint8_t *ws,
                           int32_t *bi, int32_t *b1, int32_t *b2);

// Frame::ComputeStereoMatches inputs (src/Frame.cc:600-774).
struct StereoArgs {
    const orbgpu_keypoint *kpsL; const uint8_t *descL; int32_t nL;
    const orbgpu_keypoint *kpsR; const uint8_t *descR; int32_t nR;
    const int32_t *row_start; const int32_t *row_items; int32_t nRows;
    const uint8_t *pyrL[ORBGPU_MAX_LEVELS]; const uint8_t *pyrR[ORBGPU_MAX_LEVELS];
    int32_t pitchL[ORBGPU_MAX_LEVELS], pitchR[ORBGPU_MAX_LEVELS], w[ORBGPU_MAX_LEVELS], h[ORBGPU_MAX_LEVELS];
    float scale[ORBGPU_MAX_LEVELS], inv_scale[ORBGPU_MAX_LEVELS];
    int32_t nlevels;
    float bf, b;
    float *uRight; float *depth; int32_t *best_sad;
    const int32_t *d_nL;  // non-null: the number of left keypoints is read on the device (nL is then the launch bound)
    // Batches (grid.y = stereo pair p; all zero for a single pair): entries per pair of the left-side arrays (kpsL,
    // descL, uRight, depth, best_sad) and of the right-side arrays, ints per pair of row_start / row_items, bytes
    // between two frames of the extractors' pyramids; d_nL[p] is pair p's count.
    int32_t strideL, strideR, row_stride, items_stride;
    int64_t pyr_strideL, pyr_strideR;
};

// Frame::UndistortKeyPoints (src/Frame.cc:517-558): cv::undistortPoints(pts, K, D, R = I, P = K),
// i.e. five fixed-point iterations of the radial + tangential model in double precision
// (OpenCV 2.4 / 3.x cvUndistortPoints), then re-projection with K; results rounded to float.
struct UndistortArgs { double fx, fy, cx, cy, k1, k2, p1, p2, k3; };
inline UndistortArgs undistort_args(const float *K4, const float *D5)
{
    UndistortArgs U = {(double)K4[0], (double)K4[1], (double)K4[2], (double)K4[3], (double)D5[0], (double)D5[1],
                       (double)D5[2], (double)D5[3], (double)D5[4]};
    return U;
}
#ifdef __HIPCC__
__device__ __forceinline__ void undistort_point(const UndistortArgs &U, float px, float py, float *ox, float *oy)
{
    const double ifx = 1.0 / U.fx, ify = 1.0 / U.fy;
    double x = ((double)px - U.cx) * ifx, y = ((double)py - U.cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0.0 * r2 + 0.0) * r2 + 0.0) * r2) / (1 + ((U.k3 * r2 + U.k2) * r2 + U.k1) * r2);
        const double deltaX = 2 * U.p1 * x * y + U.p2 * (r2 + 2 * x * x);
        const double deltaY = U.p1 * (r2 + 2 * y * y) + 2 * U.p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = U.fx * x + 0.0 * y + U.cx, yy = 0.0 * x + U.fy * y + U.cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    *ox = (float)(xx * ww);
    *oy = (float)(yy * ww);
}
#endif

}  // namespace orbgpu
