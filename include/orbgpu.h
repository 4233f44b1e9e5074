/*
 * orbgpu.h -- C ABI of the MI355X (gfx950) ORB front-end library.
 *
 * This is the drop-in boundary for the feature path of ORB-SLAM2: the host C++
 * classes ORBextractor / ORBmatcher keep their signatures and forward to these
 * entry points.  Reference citations are relative to
 * /root/reference/src/ORB_SLAM2/.
 *
 * Conventions
 *  - every function returns 0 (ORBGPU_OK) or a negative ORBGPU_E_* code; nothing
 *    throws, prints or exits (the reference's operator() is void/no-throw:
 *    include/ORBextractor.h:59-61);
 *  - "host" entry points take ordinary host pointers and synchronise before
 *    returning; "_device" entry points take device pointers, enqueue on the
 *    handle's stream and return without synchronising;
 *  - a handle owns its device memory and one HIP stream; different handles may
 *    be used from different threads concurrently (the reference runs the left
 *    and right extractor in two threads, src/Frame.cc:78-81); one handle must
 *    not be used from two threads at once;
 *  - keypoints use the binary layout of cv::KeyPoint (28 bytes).
 */
#ifndef ORBGPU_H
#define ORBGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBGPU_OK 0
#define ORBGPU_E_INVALID (-1)     /* bad argument */
#define ORBGPU_E_NOMEM (-2)       /* host or device allocation failed */
#define ORBGPU_E_HIP (-3)         /* a HIP runtime call failed (see orbgpu_last_error) */
#define ORBGPU_E_CAPACITY (-4)    /* an output or internal buffer is too small */
#define ORBGPU_E_UNSUPPORTED (-5) /* geometry outside the supported range */

#define ORBGPU_MAX_LEVELS 16
#define ORBGPU_GRID_COLS 64 /* include/Frame.h:38 */
#define ORBGPU_GRID_ROWS 48 /* include/Frame.h:37 */
#define ORBGPU_TH_HIGH 100  /* src/ORBmatcher.cc:37 */
#define ORBGPU_TH_LOW 50    /* src/ORBmatcher.cc:38 */
#define ORBGPU_HISTO_LENGTH 30 /* src/ORBmatcher.cc:39 */

typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbgpu_keypoint;

/* Constructor arguments of ORBextractor (include/ORBextractor.h:51-52) plus the
 * sizing of the device-side working set. */
typedef struct {
    int32_t nfeatures;
    float scaleFactor;
    int32_t nlevels;
    int32_t iniThFAST;
    int32_t minThFAST;
    int32_t gauss_variant; /* 0: 8-bit Gaussian taps 18,34,49,55,49,34,18; 1: 18,34,48,56,48,34,18 */
    int32_t device;        /* HIP device ordinal */
    int32_t max_width;     /* largest image the handle will see */
    int32_t max_height;
    int32_t max_batch;     /* frames processed per launch by the batch entry points (>=1) */
} orbgpu_params;

typedef struct orbgpu_extractor orbgpu_extractor;
typedef struct orbgpu_matcher orbgpu_matcher;

const char *orbgpu_last_error(void); /* thread-local description of the last failure */
int orbgpu_device_count(int *count);

/* Page-locked host memory for the batch entry points (no counterpart in the
 * reference): the GPU reads and writes it directly, without a staging copy. */
int orbgpu_pinned_alloc(size_t bytes, void **out);
void orbgpu_pinned_free(void *p);

/* ---- ORBextractor ------------------------------------------------------- */

/* ORBextractor::ORBextractor (src/ORBextractor.cc:456-577). */
int orbgpu_extractor_create(const orbgpu_params *p, orbgpu_extractor **out);
void orbgpu_extractor_destroy(orbgpu_extractor *e);
/* Use the caller's hipStream_t (as void*) instead of the handle's own stream. */
int orbgpu_extractor_set_stream(orbgpu_extractor *e, void *hip_stream);
int orbgpu_extractor_sync(orbgpu_extractor *e);

/* Overflow flags of the handle's last extract call (synchronises): bit 0: the
 * quadtree ran out of node storage; bit 1: a frame yielded more keypoints than
 * the `capacity` of orbgpu_extract_batch_device (its outputs and counts[] are
 * truncated to capacity).  The host-pointer entry points check the flags
 * themselves and return ORBGPU_E_CAPACITY; the asynchronous device entry point
 * cannot, so its callers ask here.  Every extract call starts with the flags
 * cleared.  (No counterpart in the reference, whose std::vector outputs grow.) */
int orbgpu_extractor_status(orbgpu_extractor *e, int32_t *flags);

/* Stage timing (the reference has only wall-clock timing in its example
 * drivers, Examples/Monocular/mono_tum.cc:79-96).  When enabled every extract
 * call records HIP events between its stages on the handle's stream;
 * orbgpu_extractor_stage_times synchronises and returns the mean milliseconds
 * of {pyramid, FAST cells, quadtree, blur, describe} over the calls since
 * profiling was enabled (at most the last 128).  enable = 1 times the normal
 * schedule (the quadtree runs on a stream of its own beside the
 * blur, so those two intervals overlap and describe's includes the wait for the
 * quadtree); enable = 2 keeps everything on the main stream so that every stage
 * is timed alone; 0 switches timing off. */
int orbgpu_extractor_profile(orbgpu_extractor *e, int32_t enable);
int orbgpu_extractor_stage_times(orbgpu_extractor *e, float *ms5, int32_t *ncalls);

/* GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:63-83) and the per-level
 * feature quotas mnFeaturesPerLevel.  Any pointer may be NULL. */
int orbgpu_extractor_tables(const orbgpu_extractor *e, int32_t *nlevels, float *scale,
                            float *inv_scale, float *sigma2, float *inv_sigma2,
                            int32_t *features_per_level);
/* Upper bound of keypoints one frame can yield (sizing of output buffers). */
int orbgpu_extractor_max_keypoints(const orbgpu_extractor *e, int32_t *max_kp);

/* ORBextractor::operator() (src/ORBextractor.cc:1455-1521) for one host image.
 * The mask argument of the reference is ignored there too and has no
 * counterpart.  An empty image (w or h == 0, or img NULL) yields *n_out = 0. */
int orbgpu_extract(orbgpu_extractor *e, const uint8_t *img, int32_t w, int32_t h,
                   int32_t stride, orbgpu_keypoint *kps, uint8_t *desc, int32_t capacity,
                   int32_t *n_out);

/* Batch of host images, all w x h: frame f outputs start at kps + f*capacity,
 * desc + f*capacity*32; counts[f] keypoints are valid.  Batches of 32 frames and
 * more are cut into chunks whose uploads, kernels and downloads overlap.  Memory
 * that is already page-locked (orbgpu_pinned_alloc, hipHostMalloc,
 * hipHostRegister) is used in place: contiguous page-locked frames (imgs[f + 1]
 * == imgs[f] + w*h, stride == w) are uploaded without a staging copy, and with
 * capacity == orbgpu_extractor_max_keypoints the results are written straight
 * into page-locked kps / desc (rows beyond counts[f] then hold unspecified
 * bytes).  Pageable memory goes through the handle's own staging buffers, copied
 * by a few host threads. */
int orbgpu_extract_batch(orbgpu_extractor *e, const uint8_t *const *imgs, int32_t nimg,
                         int32_t w, int32_t h, int32_t stride, orbgpu_keypoint *kps,
                         uint8_t *desc, int32_t capacity, int32_t *counts);

/* Same, device-resident and asynchronous: d_imgs holds nimg images, image f at
 * d_imgs + f*frame_stride (row pitch `stride`); outputs are device buffers.
 * nimg <= max_batch. */
int orbgpu_extract_batch_device(orbgpu_extractor *e, const uint8_t *d_imgs, int32_t nimg,
                                int32_t w, int32_t h, int32_t stride, size_t frame_stride,
                                orbgpu_keypoint *d_kps, uint8_t *d_desc, int32_t capacity,
                                int32_t *d_counts);

/* mvImagePyramid[level] of frame `frame` of the last extract call
 * (include/ORBextractor.h:85, read by src/Frame.cc:607,697,709,714).
 * border = 0..19 extra reflect-101 pixels on each side (the reference keeps a
 * 19 px halo around every level, src/ORBextractor.cc:1538-1586).  blurred != 0
 * returns the 7x7 Gaussian-blurred level used for descriptors instead
 * (border must be 0).  dst may be NULL to query the size only. */
int orbgpu_pyramid_level(orbgpu_extractor *e, int32_t frame, int32_t level, int32_t blurred,
                         int32_t border, uint8_t *dst, int32_t dst_stride, int32_t *w,
                         int32_t *h);
/* Device view of the same plane (interior pointer + pitch); valid until the
 * next extract call on the handle.  Above and below the interior the plane
 * carries the 19 halo rows; the 19 halo columns per side, which the extraction
 * never reads, are filled by the first orbgpu_pyramid_level call with
 * border > 0 after an extraction.  blurred != 0 is refused with
 * ORBGPU_E_UNSUPPORTED: the blurred levels are stored in 4 x 32 pixel tiles,
 * not row by row (orbgpu_pyramid_level returns them as images). */
int orbgpu_pyramid_level_device(orbgpu_extractor *e, int32_t frame, int32_t level,
                                int32_t blurred, const uint8_t **d_ptr, int32_t *pitch,
                                int32_t *w, int32_t *h);

/* Stage outputs of the last extract call, for parity tests: FAST candidates of
 * one level in the order ComputeKeyPointsOctTree collects them
 * (src/ORBextractor.cc:1043-1107; coordinates relative to (16,16)), and the
 * keypoints DistributeOctTree retained (src/ORBextractor.cc:1116-1117). */
int orbgpu_debug_level_candidates(orbgpu_extractor *e, int32_t frame, int32_t level,
                                  int32_t *xs, int32_t *ys, int32_t *resp, int32_t capacity,
                                  int32_t *n_out);
int orbgpu_debug_level_selected(orbgpu_extractor *e, int32_t frame, int32_t level,
                                int32_t *xs, int32_t *ys, int32_t *resp, int32_t capacity,
                                int32_t *n_out);

/* ---- ORBmatcher ---------------------------------------------------------- */

/* A matcher context: one HIP stream plus scratch and staging buffers.  The
 * reference builds an ORBmatcher on the stack at every call site
 * (src/Tracking.cc:738,951,1060; src/LocalMapping.cc:215; src/LoopClosing.cc:241),
 * so contexts are recycled: destroy parks an idle context (up to 8 per device),
 * create takes one back, and neither then touches the HIP allocator. */
int orbgpu_matcher_create(int32_t device, orbgpu_matcher **out);
void orbgpu_matcher_destroy(orbgpu_matcher *m);
/* Grid geometry of the NEXT windowed search on this handle (consumed by that one
 * call; NULL clears it): geom6 = {build origin x, y, query origin x, y, inverse
 * cell width, inverse cell height}.  Without it a call derives everything from
 * its bounds4 as a Frame does (src/Frame.cc:101-102, 397-500).  A KeyFrame keeps
 * the grid and the inverse cell sizes of the Frame it was made from but stores
 * mnMinX / mnMinY truncated to int and uses those in GetFeaturesInArea
 * (include/KeyFrame.h:130-133,185, src/KeyFrame.cc:37-52, 585-624): with a distorted
 * camera the two origins differ by a fraction of a pixel, which can reorder
 * candidates of equal distance. */
int orbgpu_matcher_set_grid_geometry(orbgpu_matcher *m, const float *geom6);
int orbgpu_matcher_set_stream(orbgpu_matcher *m, void *hip_stream);
int orbgpu_matcher_sync(orbgpu_matcher *m);

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1710-1727) for one pair of
 * 32-byte descriptors in host memory. */
int orbgpu_descriptor_distance(const uint8_t *a, const uint8_t *b);

/* Brute-force scan (the inner pattern of src/ORBmatcher.cc:205-226, :621-647,
 * src/MapPoint.cc:302-313): for every query the train descriptor with the
 * smallest Hamming distance (lowest index wins ties), that distance and the
 * second-smallest distance.  nt == 0 gives idx -1 and distances 256. */
int orbgpu_hamming_bruteforce(orbgpu_matcher *m, const uint8_t *Q, int32_t nq,
                              const uint8_t *T, int32_t nt, int32_t *best_idx,
                              int32_t *best_d, int32_t *second_d);
int orbgpu_hamming_bruteforce_device(orbgpu_matcher *m, const uint8_t *d_Q, int32_t nq,
                                     const uint8_t *d_T, int32_t nt, int32_t *d_best_idx,
                                     int32_t *d_best_d, int32_t *d_second_d);

/* Which kernel the brute-force scan of this handle runs on (no counterpart in the
 * reference; results are identical): 0 = chosen by size (the default), 1 = the
 * vector-unit kernel (xor + population count per pair), 2 = the matrix-core kernel
 * (descriptors spread to one byte per bit, v_mfma_i32_32x32x32_i8; distances follow
 * from |a| + |b| - 2 * popcount(a & b)).  The allocation-free entry point below always
 * runs the vector-unit kernel. */
int orbgpu_matcher_set_bruteforce_kernel(orbgpu_matcher *m, int32_t which);

/* orbgpu_hamming_bruteforce_device for callers that cannot allocate (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  It has
 * no counterpart in the reference.  Same arguments, same results; it calls no
 * hipMalloc / hipFree: the per-chunk partial results live in a fixed 16 MiB
 * array in each GPU's memory (held until the process ends; its address is looked
 * up per device in orbgpu_matcher_create).  Limit: nq * ceil(nt / 2^20) <= 2^21
 * (e.g. 100 000 x 100 000, or 2 M queries against up to 1 M train descriptors);
 * beyond it the call returns ORBGPU_E_CAPACITY.  All calls on one GPU share that
 * array: calls that overlap in time on the same GPU (different handles or
 * streams) race and must be serialised by the caller.  Nothing in it has to
 * start at zero: every partial result a call reads is written by that call. */
int orbgpu_hamming_bruteforce_device_noalloc(orbgpu_matcher *m, const uint8_t *d_Q, int32_t nq,
                                             const uint8_t *d_T, int32_t nt, int32_t *d_best_idx,
                                             int32_t *d_best_d, int32_t *d_second_d);

/* One windowed query of the SearchByProjection family: projection (u,v), search
 * radius r (already multiplied by the level scale), octave window as
 * Frame::GetFeaturesInArea takes it (src/Frame.cc:397), the query keypoint's
 * angle for the rotation histogram, and the predicted right-image coordinate
 * ur (compared only against train keypoints with uRight > 0). */
typedef struct {
    float u, v, r;
    int32_t minLevel, maxLevel;
    float angle;
    float ur;
    int32_t flags; /* bit0: query is valid; bit1: its match marks the train keypoint occupied */
} orbgpu_wquery;

#define ORBGPU_MODE_LOCAL_MAP 0  /* SearchByProjection(Frame&, vector<MapPoint*>&, th)   src/ORBmatcher.cc:45-129   */
#define ORBGPU_MODE_LAST_FRAME 1 /* SearchByProjection(Cur, Last, th, bMono)            src/ORBmatcher.cc:1379-1521 */
#define ORBGPU_MODE_KEYFRAME 2   /* SearchByProjection(Cur, KF, found, th, ORBdist)      src/ORBmatcher.cc:1523-1650 */

/* Windowed Hamming matching with the reference's sequential semantics (queries
 * are resolved in index order; a train keypoint claimed by an earlier query is
 * skipped by later ones).  bounds4 = {mnMinX, mnMinY, mnMaxX, mnMaxY} of the
 * train frame.  uright (nullable) = mvuRight of the train frame; occupied
 * (nullable) = 1 where the train keypoint already holds a map point.  Output:
 * assign[n2] = index of the query whose map point ends up on each train
 * keypoint, -1 where the search left the keypoint alone, -2 where the
 * rotation-consistency filter of modes LAST_FRAME / KEYFRAME removed a match
 * (the reference then sets the keypoint's map point to NULL, :1513, :1642);
 * *nmatches = the reference's return value. */
int orbgpu_search_by_projection(orbgpu_matcher *m, int32_t mode, const orbgpu_wquery *q,
                                const uint8_t *qdesc, int32_t nq, const orbgpu_keypoint *kps2,
                                const uint8_t *desc2, const float *uright2, int32_t n2,
                                const float *bounds4, const uint8_t *occupied,
                                int32_t *assign, int32_t th_dist, float nnratio,
                                int32_t check_orientation, int32_t *nmatches);

/* Device-resident batch: pair p matches queries [p] against train frame [p].
 * Arrays are strided by q_capacity / t_capacity entries per pair; nq[p], n2[p]
 * are device arrays of valid counts. uright/occupied may be NULL. */
int orbgpu_search_by_projection_batch_device(
    orbgpu_matcher *m, int32_t mode, int32_t npairs, const orbgpu_wquery *d_q,
    const uint8_t *d_qdesc, const int32_t *d_nq, int32_t q_capacity,
    const orbgpu_keypoint *d_kps2, const uint8_t *d_desc2, const float *d_uright2,
    const int32_t *d_n2, int32_t t_capacity, const float *bounds4_host,
    const uint8_t *d_occupied, int32_t *d_assign, int32_t th_dist, float nnratio,
    int32_t check_orientation, int32_t *d_nmatches);

/* Work counters of the last windowed-matching call for one pair: fixed-point
 * iterations of the sequential-claim resolution and full window rescans (no
 * counterpart in the reference; for tests and tuning). */
int orbgpu_debug_window_stats(orbgpu_matcher *m, int32_t pair, int32_t *iterations,
                              int32_t *rescans);

/* Builds SearchByProjection(Cur, Last) queries on the device from the last
 * frame's keypoints displaced by a known image-plane motion (du, dv): u = x+du,
 * v = y+dv, r = th*scale[octave], levels octave-1..octave+1, all valid and
 * claiming.  This stands in for the pose-based projection of
 * src/ORBmatcher.cc:1410-1441, which is outside the descriptor path. */
int orbgpu_make_shift_queries_device(orbgpu_matcher *m, int32_t npairs,
                                     const orbgpu_keypoint *d_kps_last, const int32_t *d_n_last,
                                     int32_t capacity, float du, float dv, float th,
                                     const float *scale_host, int32_t nlevels,
                                     const float *bounds4_host, orbgpu_wquery *d_q);

/* ORBmatcher::SearchByBoW, both overloads (src/ORBmatcher.cc:159-288 KeyFrame x Frame,
 * :573-706 KeyFrame x KeyFrame).  The two FeatureVectors are intersected by the
 * caller: segment s lists the train-side feature indices of the s-th common
 * vocabulary node (seg_index[seg_start[s] .. seg_start[s+1])), and queries[] holds
 * the query-side features that own a good map point in processing order (nodes
 * ascending, index order inside a node) with flags bit0 = 1, angle = the
 * keypoint angle and minLevel = the segment of their node (u, v, r, maxLevel, ur
 * unused).  excluded[n2] (or NULL) marks train features that may not be matched
 * (:625-631: no map point).  A query takes the nearest train feature of its
 * segment that no earlier query took, if its distance is <= th_dist (TH_LOW for
 * :228, TH_LOW-1 for the strict test of :649) and < nnratio * second best (256 if
 * none); then the rotation-consistency filter (:267-285).  assign[n2]: query per
 * train feature or -1; qmatch[nq] (or NULL): train feature per query or -1. */
int orbgpu_search_by_bow(orbgpu_matcher *m, const orbgpu_wquery *queries, const uint8_t *query_desc,
                         int32_t nq, const orbgpu_keypoint *kps2, const uint8_t *desc2, int32_t n2,
                         const int32_t *seg_start, const int32_t *seg_index, int32_t nseg,
                         const uint8_t *excluded, int32_t th_dist, float nnratio,
                         int32_t check_orientation, int32_t *assign, int32_t *qmatch,
                         int32_t *nmatches);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:708-874): BoW-guided search
 * between two keyframes for features without a map point, gated by the epipolar
 * line.  Segments as in orbgpu_search_by_bow.  queries[]: the side-1 features
 * without a map point (and with a stereo measurement if only_stereo), in
 * processing order; u, v = undistorted keypoint position, angle, minLevel =
 * segment, flags = 1 | (4 if mvuRight >= 0).  excluded[n2]: side-2 features that
 * hold a map point; uright2[n2] or NULL (monocular).  F12: 3x3 row-major; (ex, ey):
 * epipole in image 2 (:714-721); scale2 / sigma2: mvScaleFactors / mvLevelSigma2
 * of keyframe 2.  A query takes the candidate of its node with the smallest
 * distance <= TH_LOW that passes the gates (:789-806; the last one among equals);
 * side-2 features are NOT made exclusive (the reference never sets vbMatched2).
 * qmatch[nq]: side-2 feature per query or -1 after the rotation filter. */
int orbgpu_search_for_triangulation(orbgpu_matcher *m, const orbgpu_wquery *queries,
                                    const uint8_t *query_desc, int32_t nq,
                                    const orbgpu_keypoint *kps2, const uint8_t *desc2,
                                    const float *uright2, int32_t n2, const int32_t *seg_start,
                                    const int32_t *seg_index, int32_t nseg, const uint8_t *excluded,
                                    const float *F12, float ex, float ey, const float *scale2,
                                    const float *sigma2, int32_t nlevels, int32_t only_stereo,
                                    int32_t check_orientation, int32_t *qmatch, int32_t *nmatches);

/* The keypoint search shared by ORBmatcher::Fuse (src/ORBmatcher.cc:876-1010 and
 * :1012-1151) and ORBmatcher::SearchBySim3 (:1153-1377): query i (a projected map
 * point: u, v, r = th*scale[predicted level], minLevel = predicted-1, maxLevel =
 * predicted, ur, flags bit0) takes the keypoint of KeyFrame::GetFeaturesInArea
 * (src/KeyFrame.cc:585-624) with the smallest descriptor distance (first one among
 * equals) if it is <= th_dist (TH_LOW for Fuse, TH_HIGH for SearchBySim3).  Queries
 * do not influence each other.  chi2_gate != 0 adds the reprojection test of
 * :945-975 (7.8 with a stereo measurement uright2[t] >= 0, else 5.99, scaled by
 * inv_sigma2[octave] = mvInvLevelSigma2).  qmatch[nq]: keypoint per query or -1;
 * the map surgery of Fuse and the mutual-agreement step of SearchBySim3
 * (:1356-1374) stay with the caller. */
int orbgpu_search_in_area(orbgpu_matcher *m, const orbgpu_wquery *queries, const uint8_t *query_desc,
                          int32_t nq, const orbgpu_keypoint *kps2, const uint8_t *desc2,
                          const float *uright2, int32_t n2, const float *bounds4,
                          const float *inv_sigma2, int32_t nlevels, int32_t chi2_gate,
                          int32_t th_dist, int32_t *qmatch, int32_t *nmatches);

/* Frame::AssignFeaturesToGrid (src/Frame.cc:270-293) with Frame::PosInGrid
 * (:485-500): the 64x48 grid over bounds4 = {mnMinX, mnMinY, mnMaxX, mnMaxY} as
 * CSR.  Cell (ix, iy) has number ix*48 + iy; cell_start[64*48 + 1];
 * cell_index[n] lists the keypoints of every cell in insertion (index) order;
 * keypoints outside the grid are left out (n_in_grid of n are listed). */
int orbgpu_assign_features_to_grid(orbgpu_matcher *m, const orbgpu_keypoint *kps_un, int32_t n,
                                   const float *bounds4, int32_t *cell_start, int32_t *cell_index,
                                   int32_t *n_in_grid);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:262-340) for a batch of
 * map points: the observations of map point p are descriptors
 * desc[start[p] .. start[p+1]) (32 bytes each).  best_idx[p] = offset within
 * that range of the descriptor with the smallest median distance to all of them
 * (first one among equals; median = element (int)(0.5*(N-1)) of the sorted row,
 * self distance included), -1 if the range is empty; best_median[p] (or NULL). */
int orbgpu_distinctive_descriptors(orbgpu_matcher *m, const uint8_t *desc, const int32_t *start,
                                   int32_t npoints, int32_t *best_idx, int32_t *best_median);

/* Frame::ComputeStereoFromRGBD (src/Frame.cc:777-798): depth[h][stride_floats] is
 * the float depth image; kps the distorted keypoints (lookup position, truncated
 * to int), kps_un the undistorted ones.  uright[i] = kps_un[i].x - bf/d and
 * out_depth[i] = d where d > 0, else both -1. */
int orbgpu_stereo_from_rgbd(orbgpu_matcher *m, const orbgpu_keypoint *kps, const orbgpu_keypoint *kps_un,
                            int32_t n, const float *depth, int32_t w, int32_t h, int32_t stride_floats,
                            float bf, float *uright, float *out_depth);

/* Frame::UndistortKeyPoints (src/Frame.cc:517-558): cv::undistortPoints with R = I and
 * P = K.  K4 = {fx, fy, cx, cy}, dist5 = {k1, k2, p1, p2, k3} (mDistCoef; k3 = 0 when
 * the calibration has four coefficients).  Copies the keypoints when k1 == 0, as the
 * reference does.  Only pt.x / pt.y change. */
int orbgpu_undistort_keypoints(orbgpu_matcher *m, const orbgpu_keypoint *kps, int32_t n, const float *K4,
                               const float *dist5, orbgpu_keypoint *kps_un);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:422-571).
 * prev_matched_xy[2*n1] in/out (vbPrevMatched), matches12[n1] out. */
int orbgpu_search_for_initialization(orbgpu_matcher *m, const orbgpu_keypoint *kps1,
                                     const uint8_t *desc1, int32_t n1,
                                     const orbgpu_keypoint *kps2, const uint8_t *desc2,
                                     int32_t n2, const float *bounds4, float *prev_matched_xy,
                                     int32_t *matches12, int32_t window_size, float nnratio,
                                     int32_t check_orientation, int32_t *nmatches);

/* SearchForInitialization for npairs frame pairs on the device, asynchronous on
 * the matcher's stream: pair p matches frame 1 [p] (capacity1 entries per pair,
 * d_n1[p] valid) against frame 2 [p], with vbPrevMatched = frame 1's own keypoint
 * positions, the state in which Tracking starts an initialisation
 * (src/Tracking.cc:592-594) -- the batch form has no carried vbPrevMatched.
 * d_matches12: capacity1 entries per pair (vnMatches12; entries beyond d_n1[p] are
 * not written); d_nmatches[p] = the reference's return value. */
int orbgpu_search_for_initialization_batch_device(
    orbgpu_matcher *m, int32_t npairs, const orbgpu_keypoint *d_kps1, const uint8_t *d_desc1,
    const int32_t *d_n1, int32_t capacity1, const orbgpu_keypoint *d_kps2, const uint8_t *d_desc2,
    const int32_t *d_n2, int32_t capacity2, const float *bounds4_host, int32_t window_size,
    float nnratio, int32_t check_orientation, int32_t *d_matches12, int32_t *d_nmatches);

/* Frame::ComputeStereoMatches (src/Frame.cc:600-774).  left/right are
 * extractor handles whose last extract call processed the stereo pair (frame
 * index `frame` of that call); keypoints/descriptors are the host outputs of
 * those calls.  Outputs mvuRight / mvDepth (nL floats each). */
int orbgpu_stereo_match(orbgpu_matcher *m, orbgpu_extractor *left, orbgpu_extractor *right,
                        int32_t frame, const orbgpu_keypoint *kpsL, const uint8_t *descL,
                        int32_t nL, const orbgpu_keypoint *kpsR, const uint8_t *descR,
                        int32_t nR, float bf, float b, float *uRight, float *depth,
                        int32_t *nvalid);

/* Frame::ComputeStereoMatches (src/Frame.cc:600-774) for npairs stereo pairs that
 * are on the device already, asynchronous on the matcher's stream: left / right
 * are extractor handles whose last orbgpu_extract_batch_device call processed the
 * npairs left / right images (pair p = frame p of both calls; their pyramids are
 * read where they lie) into d_kpsL / d_descL / d_nL with `capL` entries per pair
 * and d_kpsR / d_descR / d_nR with `capR`.  The caller orders this call after the
 * two extractions (same stream, or an event).  Outputs, capL entries per pair:
 * d_uRight, d_depth (mvuRight / mvDepth; entries beyond d_nL[p] are not written)
 * and d_nvalid[p] = left keypoints of pair p that keep a depth.  Keypoint octaves
 * outside the pyramid are clamped into it. */
int orbgpu_stereo_match_batch_device(orbgpu_matcher *m, orbgpu_extractor *left, orbgpu_extractor *right,
                                     int32_t npairs, const orbgpu_keypoint *d_kpsL, const uint8_t *d_descL,
                                     const int32_t *d_nL, int32_t capL, const orbgpu_keypoint *d_kpsR,
                                     const uint8_t *d_descR, const int32_t *d_nR, int32_t capR, float bf,
                                     float b, float *d_uRight, float *d_depth, int32_t *d_nvalid);

/* ---- Frame: device-resident frame state ----------------------------------- */

/* What the constructors of Frame compute directly after the extraction
 * (src/Frame.cc:58-116 stereo, :119-172 RGB-D, :175-226 monocular), kept on the
 * device: mvKeys, mvKeysUn, mDescriptors, mvuRight, mvDepth and the 64x48 mGrid.
 * The build calls run UndistortKeyPoints (:517-558), ComputeStereoMatches
 * (:600-774) or ComputeStereoFromRGBD (:777-798) and AssignFeaturesToGrid
 * (:270-293) behind the extraction on the extractor's stream and wait once; the
 * *_frame matcher entry points then read the frame where it lies, and the host
 * fetches only what it asks for (orbgpu_frame_download, orbgpu_frame_grid).  A
 * frame owns its buffers: it stays valid while the extractor goes on to the next
 * image (Tracking keeps mLastFrame; a KeyFrame copies its Frame's arrays,
 * src/KeyFrame.cc:37-52, and can keep the handle instead).  capacity >=
 * orbgpu_extractor_max_keypoints of the extractors it will be built from. */
typedef struct orbgpu_frame orbgpu_frame;
int orbgpu_frame_create(int32_t device, int32_t capacity, orbgpu_frame **out);
void orbgpu_frame_destroy(orbgpu_frame *f);

/* Frame::Frame, monocular (src/Frame.cc:175-226): ExtractORB(0, imGray),
 * UndistortKeyPoints, mvuRight = mvDepth = -1, AssignFeaturesToGrid.  K4 = {fx,
 * fy, cx, cy}, dist5 = {k1, k2, p1, p2, k3} as for orbgpu_undistort_keypoints;
 * bounds4 = {mnMinX, mnMinY, mnMaxX, mnMaxY} (ComputeImageBounds, :560-597,
 * computed once per camera by the caller as the reference does).  *n_out = N. */
int orbgpu_frame_build_mono(orbgpu_frame *f, orbgpu_extractor *e, const uint8_t *img, int32_t w,
                            int32_t h, int32_t stride, const float *K4, const float *dist5,
                            const float *bounds4, int32_t *n_out);
/* Frame::Frame, RGB-D (src/Frame.cc:119-172): depth[h][depth_stride_floats] is
 * imDepth (float, already scaled by mDepthMapFactor); bf = mbf. */
int orbgpu_frame_build_rgbd(orbgpu_frame *f, orbgpu_extractor *e, const uint8_t *img, int32_t w,
                            int32_t h, int32_t stride, const float *depth,
                            int32_t depth_stride_floats, float bf, const float *K4,
                            const float *dist5, const float *bounds4, int32_t *n_out);
/* Frame::Frame, stereo (src/Frame.cc:58-116): the two extractions run side by side
 * on the two handles' streams (two threads in the reference, :78-81), then
 * ComputeStereoMatches with its row table, sub-pixel refinement and median-based
 * outlier rejection all on the device.  The right image's keypoints stay in the
 * right extractor's buffers (the reference keeps mvKeysRight only for this step).
 * b = mb, bf = mbf; *nvalid_out (nullable) = left keypoints with a depth. */
int orbgpu_frame_build_stereo(orbgpu_frame *f, orbgpu_extractor *left, orbgpu_extractor *right,
                              const uint8_t *imgL, const uint8_t *imgR, int32_t w, int32_t h,
                              int32_t stride, float bf, float b, const float *K4,
                              const float *dist5, const float *bounds4, int32_t *n_out,
                              int32_t *nvalid_out);

/* A frame from host arrays (a Frame the program computed itself, a KeyFrame read
 * back from a saved map): kps may be NULL (then = kps_un), uright / depth may be
 * NULL (then -1: monocular); octaves must lie in 0 .. nlevels-1.  The grid is built
 * over bounds4 as AssignFeaturesToGrid does (src/Frame.cc:270-293). */
int orbgpu_frame_upload(orbgpu_frame *f, const orbgpu_keypoint *kps, const orbgpu_keypoint *kps_un,
                        const uint8_t *desc, const float *uright, const float *depth, int32_t n,
                        int32_t nlevels, const float *bounds4);

/* N, the number of keypoints with a stereo / depth measurement, and the capacity.
 * Any pointer may be NULL. */
int orbgpu_frame_size(const orbgpu_frame *f, int32_t *n, int32_t *nvalid, int32_t *capacity);
/* Host copies of mvKeys, mvKeysUn, mDescriptors, mvuRight, mvDepth (N entries
 * each; any pointer may be NULL). */
int orbgpu_frame_download(orbgpu_frame *f, orbgpu_keypoint *kps, orbgpu_keypoint *kps_un,
                          uint8_t *desc, float *uright, float *depth);
/* mGrid as CSR, laid out as orbgpu_assign_features_to_grid returns it. */
int orbgpu_frame_grid(orbgpu_frame *f, int32_t *cell_start, int32_t *cell_index,
                      int32_t *n_in_grid);
/* Device views (valid until the frame is rebuilt or destroyed); d_count[0] = N. */
int orbgpu_frame_device(const orbgpu_frame *f, const orbgpu_keypoint **d_kps,
                        const orbgpu_keypoint **d_kps_un, const uint8_t **d_desc,
                        const float **d_uright, const float **d_depth, const int32_t **d_count);

/* The matcher entry points above with a frame as the train side: its undistorted
 * keypoints, descriptors, uRight, bounds and grid are used where they lie, nothing
 * of the train side is uploaded and no grid is built.  Arguments, semantics and
 * outputs are those of the array versions (assign[N], qmatch[nq]).  A one-shot
 * grid geometry (orbgpu_matcher_set_grid_geometry) still moves the query origin;
 * its build origin must be the frame's bounds. */
int orbgpu_search_by_projection_frame(orbgpu_matcher *m, int32_t mode, const orbgpu_wquery *q,
                                      const uint8_t *qdesc, int32_t nq, const orbgpu_frame *train,
                                      const uint8_t *occupied, int32_t *assign, int32_t th_dist,
                                      float nnratio, int32_t check_orientation, int32_t *nmatches);
int orbgpu_search_in_area_frame(orbgpu_matcher *m, const orbgpu_wquery *queries,
                                const uint8_t *query_desc, int32_t nq, const orbgpu_frame *train,
                                const float *inv_sigma2, int32_t nlevels, int32_t chi2_gate,
                                int32_t th_dist, int32_t *qmatch, int32_t *nmatches);
int orbgpu_search_by_bow_frame(orbgpu_matcher *m, const orbgpu_wquery *queries,
                               const uint8_t *query_desc, int32_t nq, const orbgpu_frame *train,
                               const int32_t *seg_start, const int32_t *seg_index, int32_t nseg,
                               const uint8_t *excluded, int32_t th_dist, float nnratio,
                               int32_t check_orientation, int32_t *assign, int32_t *qmatch,
                               int32_t *nmatches);
/* use_uright = 0 for a monocular keyframe (uright2 == NULL in the array version). */
int orbgpu_search_for_triangulation_frame(orbgpu_matcher *m, const orbgpu_wquery *queries,
                                          const uint8_t *query_desc, int32_t nq,
                                          const orbgpu_frame *train, int32_t use_uright,
                                          const int32_t *seg_start, const int32_t *seg_index,
                                          int32_t nseg, const uint8_t *excluded, const float *F12,
                                          float ex, float ey, const float *scale2,
                                          const float *sigma2, int32_t nlevels, int32_t only_stereo,
                                          int32_t check_orientation, int32_t *qmatch,
                                          int32_t *nmatches);
/* SearchForInitialization(F1, F2, ...) (src/ORBmatcher.cc:422-571) between two
 * frames: only vbPrevMatched travels (prev_matched_xy[2*N1] in/out), matches12[N1]
 * comes back. */
int orbgpu_search_for_initialization_frames(orbgpu_matcher *m, const orbgpu_frame *f1,
                                            const orbgpu_frame *f2, float *prev_matched_xy,
                                            int32_t *matches12, int32_t window_size, float nnratio,
                                            int32_t check_orientation, int32_t *nmatches);

/* ---- ORBVocabulary (DBoW2) ---------------------------------------------- */
/* Citations in this block are relative to Thirdparty/DBoW2/DBoW2/ of the reference. */

typedef struct orbgpu_vocabulary orbgpu_vocabulary;

/* WeightingType and ScoringType (BowVector.h:36-55). */
#define ORBGPU_WEIGHT_TF_IDF 0
#define ORBGPU_WEIGHT_TF 1
#define ORBGPU_WEIGHT_IDF 2
#define ORBGPU_WEIGHT_BINARY 3
#define ORBGPU_SCORE_L1_NORM 0
#define ORBGPU_SCORE_L2_NORM 1
#define ORBGPU_SCORE_CHI_SQUARE 2
#define ORBGPU_SCORE_KL 3
#define ORBGPU_SCORE_BHATTACHARYYA 4
#define ORBGPU_SCORE_DOT_PRODUCT 5

/* A vocabulary tree from the arrays TemplatedVocabulary::loadFromTextFile reads
 * (TemplatedVocabulary.h:1338-1424): node i+1 (i = 0 .. n_nodes-1, file order; node 0
 * is the root) has parent[i], is_leaf[i], desc[i*32 ..], weight[i]; children keep
 * file order and words are numbered in file order of the leaves.  k, L, scoring and
 * weighting are the header fields, accepted in the ranges of :1359.  Unlike the
 * reference, which does not check, a tree in which a parent is not an earlier
 * node, a leaf has children or an inner node has none is refused
 * (ORBGPU_E_INVALID); more than 32 children per node: ORBGPU_E_UNSUPPORTED. */
int orbgpu_vocabulary_create(int32_t device, int32_t k, int32_t L, int32_t scoring, int32_t weighting,
                             int32_t n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                             const uint8_t *desc, const double *weight, orbgpu_vocabulary **out);
/* ORBVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1424; called at
 * src/System.cc:77).  Blank lines add no node: the reference's `while(!f.eof())`
 * loop turns the empty line after the final newline into one more child of the
 * root with an uninitialised descriptor; that node is not reproduced. */
int orbgpu_vocabulary_load_text(int32_t device, const char *path, orbgpu_vocabulary **out);
void orbgpu_vocabulary_destroy(orbgpu_vocabulary *v);
int orbgpu_vocabulary_set_stream(orbgpu_vocabulary *v, void *hip_stream);
int orbgpu_vocabulary_sync(orbgpu_vocabulary *v);
/* getBranchingFactor, getDepthLevels, getScoringType, getWeightingType, node count
 * without the root, size() = number of words (TemplatedVocabulary.h:100-223).  Any
 * pointer may be NULL. */
int orbgpu_vocabulary_info(const orbgpu_vocabulary *v, int32_t *k, int32_t *L, int32_t *scoring,
                           int32_t *weighting, int32_t *n_nodes, int32_t *n_words);

/* TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
 * (TemplatedVocabulary.h:1127-1194) = Frame::ComputeBoW / KeyFrame::ComputeBoW
 * (src/Frame.cc:503-510, src/KeyFrame.cc:65-76; levelsup = 4) for n descriptors
 * of 32 bytes.  Per feature (either may be NULL): word_ids[n] and node_ids[n],
 * the node at level L - levelsup on the feature's path (0 when that level is
 * <= 0; where the leaf lies above that level the reference leaves the id unset
 * and this library reports the leaf).  BowVector: ascending bow_ids[*n_bow] with
 * bow_vals, accumulated and normalised in the order std::map gives
 * (BowVector.cpp:34-84), so the doubles are bit-identical.  FeatureVector:
 * ascending fv_nodes[*n_fv]; node j lists the features
 * fv_feat[fv_start[j] .. fv_start[j+1]) in ascending order (FeatureVector.cpp:31-45).
 * Features whose word has weight 0 are in neither.  Capacities: n entries each,
 * fv_start n+1.  n <= 16384. */
int orbgpu_vocabulary_transform(orbgpu_vocabulary *v, const uint8_t *desc, int32_t n, int32_t levelsup,
                                int32_t *word_ids, int32_t *node_ids, int32_t *bow_ids,
                                double *bow_vals, int32_t *n_bow, int32_t *fv_nodes,
                                int32_t *fv_start, int32_t *fv_feat, int32_t *n_fv);

/* Same for nsets descriptor sets resident on the device, asynchronous on the
 * handle's stream: set s has min(d_counts[s], capacity) descriptors at
 * d_desc + s*capacity*32 (the layout orbgpu_extract_batch_device writes); every
 * output uses the stride `capacity` per set, d_fv_start capacity+1. */
int orbgpu_vocabulary_transform_batch_device(orbgpu_vocabulary *v, const uint8_t *d_desc,
                                             const int32_t *d_counts, int32_t nsets, int32_t capacity,
                                             int32_t levelsup, int32_t *d_word_ids, int32_t *d_node_ids,
                                             int32_t *d_bow_ids, double *d_bow_vals, int32_t *d_n_bow,
                                             int32_t *d_fv_nodes, int32_t *d_fv_start,
                                             int32_t *d_fv_feat, int32_t *d_n_fv);

#ifdef __cplusplus
}
#endif
#endif
