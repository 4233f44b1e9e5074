/*
 * orb_oracle.h -- CPU restatement of the ORB-SLAM2 front-end feature path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped HIP library links, loads or
 * calls this code; it exists so tests (and the bench's cpu_baseline leg) can
 * check the GPU path against a scalar statement of the reference algorithm.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for
 * this path and its arithmetic lives in OpenCV (not vendored, pinned only as
 * ">= 2.4.3", tested 2.4.11 / 3.2), which is absent here.  The OpenCV routines
 * (resize INTER_LINEAR 8U, copyMakeBorder REFLECT_101, FAST 9_16 + NMS,
 * fastAtan2, GaussianBlur 7x7 sigma 2 on 8U, cvRound) are restated from their
 * published generic-C++ algorithms (OpenCV 3.2, no IPP/OpenCL).  Only the
 * constants the reference itself states (umax table, TH_LOW/TH_HIGH/HISTO_LENGTH,
 * border example, DescriptorDistance arithmetic) are pinned.
 *
 * Every function cites the reference lines it follows, relative to
 * /root/reference/src/ORB_SLAM2/.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16
#define ORC_EDGE 19          /* EDGE_THRESHOLD, src/ORBextractor.cc:74 */
#define ORC_GRID_COLS 64     /* include/Frame.h:38 */
#define ORC_GRID_ROWS 48     /* include/Frame.h:37 */
#define ORC_TH_HIGH 100      /* src/ORBmatcher.cc:37 */
#define ORC_TH_LOW 50        /* src/ORBmatcher.cc:38 */
#define ORC_HISTO_LENGTH 30  /* src/ORBmatcher.cc:39 */

/* Binary layout of cv::KeyPoint (28 bytes). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

typedef struct {
    int nfeatures;
    float scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    int gauss_variant; /* 0: kernel 18,34,49,55,... (sum 257); 1: 18,34,48,56,... (sum 256) */
} orc_params;

/* ---- scalar helpers -------------------------------------------------- */
int orc_cvround_f(float v);
float orc_fast_atan2(float y, float x);
int orc_hamming256(const uint8_t *a, const uint8_t *b);

/* ---- per-extractor tables (ORBextractor ctor, src/ORBextractor.cc:456-577) */
int orc_tables(const orc_params *p, float *scale, float *inv_scale, float *sigma2,
               float *inv_sigma2, int *quota, int *umax16);
void orc_level_size(const orc_params *p, int w, int h, int level, int *lw, int *lh);

/* ---- image primitives -------------------------------------------------- */
int orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride,
                         uint8_t *dst, int dw, int dh, int dstride);
void orc_border_reflect101(uint8_t *interior, int w, int h, int stride, int border);
void orc_gauss7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride,
                int variant);

/* cv::FAST(TYPE_9_16) on a (sub-)image; returns number of keypoints, writes up
 * to cap entries in row-major order.  scores = FAST response. */
int orc_fast9(const uint8_t *img, int w, int h, int stride, int threshold, int nonmax,
              int *xs, int *ys, int *scores, int cap);
/* Threshold-free score A-1 (clamped at 0) of one pixel; needs 3 px margin. */
int orc_fast_score_pixel(const uint8_t *p, int stride);

/* Per-level FAST over 30-px cells with ini/min threshold fallback
 * (src/ORBextractor.cc:1018-1107).  Coordinates relative to (16,16). */
int orc_level_candidates(const uint8_t *plane, int w, int h, int stride, int iniTh,
                         int minTh, float *xs, float *ys, float *resp, int cap);

/* DistributeOctTree (src/ORBextractor.cc:688-1006) with the deterministic
 * tie-break (count, creation sequence).  Writes indices (into the input arrays)
 * of the retained points in output order.  Returns count or <0 on error. */
int orc_distribute_octree(const float *xs, const float *ys, const float *resp, int n,
                          int minX, int maxX, int minY, int maxY, int N, int *out_idx,
                          int cap, int *tie_straddle);

float orc_ic_angle(const uint8_t *plane, int stride, int cx, int cy, const int *umax16);
void orc_brief(const uint8_t *blurred, int stride, int cx, int cy, float angle_deg,
               uint8_t *desc32);

/* ---- the extractor ------------------------------------------------------ */
typedef struct orc_extractor orc_extractor;
orc_extractor *orc_extractor_create(const orc_params *p);
void orc_extractor_destroy(orc_extractor *e);
/* ORBextractor::operator() (src/ORBextractor.cc:1455-1521).  Returns the number
 * of keypoints (<= cap) or <0. */
int orc_extract(orc_extractor *e, const uint8_t *img, int w, int h, int stride,
                orc_keypoint *kps, uint8_t *desc, int cap);
/* Views into the pyramid of the last orc_extract call (interior pointer; the
 * allocation carries a 19 px reflect-101 halo around it). blurred=1 returns the
 * Gaussian-blurred plane used for descriptors (no halo). */
int orc_extractor_level(orc_extractor *e, int level, int blurred, const uint8_t **ptr,
                        int *w, int *h, int *stride);
int orc_extractor_level_count(orc_extractor *e, int level);
int orc_extractor_tie_straddles(orc_extractor *e);

/* ---- matching ----------------------------------------------------------- */
void orc_bruteforce(const uint8_t *Q, int nq, const uint8_t *T, int nt, int32_t *best_idx,
                    int32_t *best_d, int32_t *second_d);

/* Frame::AssignFeaturesToGrid + PosInGrid (src/Frame.cc:270-293,485-500).
 * cell_start has 64*48+1 entries, cells indexed ix*48+iy; cell_items has n. */
void orc_grid_build(const orc_keypoint *kps, int n, const float *bounds4,
                    int32_t *cell_start, int32_t *cell_items);
/* Frame::GetFeaturesInArea (src/Frame.cc:397-474). Returns count. */
int orc_features_in_area(const orc_keypoint *kps, const float *bounds4,
                         const int32_t *cell_start, const int32_t *cell_items, float x,
                         float y, float r, int minLevel, int maxLevel, int32_t *out, int cap);

/* The same with an explicit grid geometry geom6 = {build origin x, y, query origin x, y, inverse cell width,
 * inverse cell height}: a KeyFrame keeps its Frame's grid and inverse cell sizes but queries with int-truncated
 * bounds (include/KeyFrame.h:185, src/KeyFrame.cc:37-52, 585-624). */
void orc_geom_from_bounds(const float *bounds4, float *geom6);
void orc_grid_build_geom(const orc_keypoint *kps, int n, const float *geom6, int32_t *cell_start, int32_t *cell_items);
int orc_features_in_area_geom(const orc_keypoint *kps, const float *geom6, const int32_t *cell_start,
                              const int32_t *cell_items, float x, float y, float r, int minLevel, int maxLevel,
                              int32_t *out, int cap);

/* Tie order of equal-sized nodes in the partial passes of orc_distribute_octree: 0 (default) = latest created first,
 * 1 = earliest created first.  The reference orders them by heap address (src/ORBextractor.cc:906); rule 1 exists to
 * measure how much of the selection depends on that order.  Process-wide, not thread-safe: measurement only. */
void orc_set_tie_rule(int rule);

/* ORBmatcher::ComputeThreeMaxima (src/ORBmatcher.cc:1661-1704) on bin sizes. */
void orc_three_maxima(const int *sizes, int L, int *ind1, int *ind2, int *ind3);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:422-571). */
int orc_search_for_initialization(const orc_keypoint *kps1, const uint8_t *desc1, int n1,
                                  const orc_keypoint *kps2, const uint8_t *desc2, int n2,
                                  const float *bounds4, float *prev_matched_xy,
                                  int32_t *matches12, int windowSize, float nnratio,
                                  int checkOri);

/* One windowed query: projection (u,v), radius, octave window, descriptor index
 * is the query's own index; angle for the rotation histogram; ur = predicted
 * right coordinate (used only when the train keypoint has uRight > 0). */
typedef struct {
    float u, v, r;
    int32_t minLevel, maxLevel;
    float angle;
    float ur;
    int32_t flags; /* bit0: valid; bit1: a match by this query marks the train keypoint occupied */
} orc_wquery;

/* mode 0: SearchByProjection(Frame&, vector<MapPoint*>, th)  (src/ORBmatcher.cc:45-129)
 * mode 1: SearchByProjection(Cur, Last, th, mono)            (src/ORBmatcher.cc:1379-1521)
 * mode 2: SearchByProjection(Cur, KF, found, th, ORBdist)    (src/ORBmatcher.cc:1523-1650)
 * occupied[n2]: in = train keypoints already holding a map point; assign[n2]:
 * out = query index assigned to each train keypoint or -1. Returns nmatches. */
int orc_search_by_projection(int mode, const orc_wquery *q, const uint8_t *qdesc, int nq,
                             const orc_keypoint *kps2, const uint8_t *desc2,
                             const float *uright2, int n2, const float *bounds4,
                             const uint8_t *occupied, int32_t *assign, int th_dist,
                             float nnratio, int checkOri);

/* ORBmatcher::SearchByBoW, both overloads (src/ORBmatcher.cc:159-288, 573-706), with
 * the FeatureVector intersection done by the caller: q[] = query-side features
 * owning a good map point, in processing order, q[i].minLevel = segment (common
 * vocabulary node) of the query; segment s lists train features
 * seg_index[seg_start[s] .. seg_start[s+1]).  excluded[n2] or NULL (:625-631).
 * th_dist = TH_LOW for :228 (<=), TH_LOW-1 for :649 (<).  assign[n2], qmatch[nq] out. */
int orc_search_by_bow(const orc_wquery *q, const uint8_t *qdesc, int nq, const orc_keypoint *kps2,
                      const uint8_t *desc2, int n2, const int32_t *seg_start,
                      const int32_t *seg_index, const uint8_t *excluded, int th_dist,
                      float nnratio, int checkOri, int32_t *assign, int32_t *qmatch);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:708-874) with CheckDistEpipolarLine
 * (:140-157), segment form as orc_search_by_bow.  q[i]: u, v = kp1 position, angle,
 * minLevel = segment, flags = 1 | (4 if stereo).  qmatch[nq] out.  Returns nmatches. */
int orc_search_for_triangulation(const orc_wquery *q, const uint8_t *qdesc, int nq,
                                 const orc_keypoint *kps2, const uint8_t *desc2, const float *uright2,
                                 int n2, const int32_t *seg_start, const int32_t *seg_index,
                                 const uint8_t *excluded, const float *F12, float ex, float ey,
                                 const float *scale2, const float *sigma2, int onlyStereo,
                                 int checkOri, int32_t *qmatch);

/* The window search of ORBmatcher::Fuse (src/ORBmatcher.cc:876-1010, 1012-1151) and
 * ORBmatcher::SearchBySim3 (:1153-1377) on KeyFrame::GetFeaturesInArea
 * (src/KeyFrame.cc:585-624): per query the nearest keypoint with octave in
 * [minLevel, maxLevel], optionally behind the chi-square gate of :945-975.
 * qmatch[nq] out; returns the number of queries with a match. */
int orc_search_in_area(const orc_wquery *q, const uint8_t *qdesc, int nq, const orc_keypoint *kps2,
                       const uint8_t *desc2, const float *uright2, int n2, const float *bounds4,
                       const float *inv_sigma2, int chi2_gate, int th_dist, int32_t *qmatch);
/* Both with an explicit grid geometry (see orc_grid_build_geom). */
int orc_search_by_projection_geom(int mode, const orc_wquery *q, const uint8_t *qdesc, int nq,
                                  const orc_keypoint *kps2, const uint8_t *desc2, const float *uright2, int n2,
                                  const float *geom6, const uint8_t *occupied, int32_t *assign, int th_dist,
                                  float nnratio, int checkOri);
int orc_search_in_area_geom(const orc_wquery *q, const uint8_t *qdesc, int nq, const orc_keypoint *kps2,
                            const uint8_t *desc2, const float *uright2, int n2, const float *geom6,
                            const float *inv_sigma2, int chi2_gate, int th_dist, int32_t *qmatch);


/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:297-334) for map points whose
 * observations are desc[start[p] .. start[p+1]).  best_idx[p] relative to start[p] (-1: none). */
void orc_distinctive_descriptors(const uint8_t *desc, const int32_t *start, int npoints,
                                 int32_t *best_idx, int32_t *best_median);

/* Frame::ComputeStereoFromRGBD (src/Frame.cc:777-798). */
void orc_stereo_from_rgbd(const orc_keypoint *kps, const orc_keypoint *kps_un, int n, const float *depth,
                          int w, int h, int stride_floats, float bf, float *uright, float *out_depth);

/* Frame::UndistortKeyPoints (src/Frame.cc:517-558) = cv::undistortPoints(pts, K, D, I, K): OpenCV's
 * cvUndistortPoints restated (double precision, five iterations).  K4 = fx, fy, cx, cy; D5 = k1, k2, p1, p2, k3. */
void orc_undistort_keypoints(const orc_keypoint *kps, int n, const float *K4, const float *D5, orc_keypoint *out);

/* DBoW2 TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, BowVector, FeatureVector,
 * levelsup) (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1194) with the per-feature tree
 * descent of :1217-1259 and FORB::distance (Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101); this is
 * Frame::ComputeBoW (src/Frame.cc:503-510, levelsup = 4).
 * The vocabulary is given as the arrays loadFromTextFile (:1338-1424) reads: node i+1 (i = 0 ..
 * n_nodes-1, in file order; node 0 is the root) has parent[i], is_leaf[i], node_desc[i*32..],
 * weight[i]; words are numbered in file order of the leaves.  scoring / weighting are the
 * enumerators of BowVector.h:36-55.
 * Outputs: word_id[n], node_id[n] per feature (node at level L - levelsup, 0 if that is <= 0);
 * BowVector as ascending bow_ids[*n_bow] / bow_vals (std::map order, BowVector.cpp:34-84);
 * FeatureVector as ascending fv_nodes[*n_fv] with features fv_feat[fv_start[j] .. fv_start[j+1])
 * in insertion (= ascending feature) order (FeatureVector.cpp:31-45).  Returns 0, or -1 on an
 * inconsistent tree (a parent that is not an earlier node, a leaf with children, an inner node
 * without). */
int orc_voc_transform(int L, int scoring, int weighting, int n_nodes, const int32_t *parent,
                      const uint8_t *is_leaf, const uint8_t *node_desc, const double *weight,
                      const uint8_t *feat_desc, int n, int levelsup, int32_t *word_id,
                      int32_t *node_id, int32_t *bow_ids, double *bow_vals, int32_t *n_bow,
                      int32_t *fv_nodes, int32_t *fv_start, int32_t *fv_feat, int32_t *n_fv);

/* Frame::ComputeStereoMatches (src/Frame.cc:600-774). pyrL/pyrR: arrays of
 * nlevels interior pointers with strides. */
int orc_stereo_matches(const orc_keypoint *kpsL, const uint8_t *descL, int nL,
                       const orc_keypoint *kpsR, const uint8_t *descR, int nR,
                       const uint8_t *const *pyrL, const uint8_t *const *pyrR,
                       const int *pyr_w, const int *pyr_h, const int *pyr_strideL,
                       const int *pyr_strideR, int nlevels, const float *scale,
                       const float *inv_scale, float bf, float b, float *uRight,
                       float *depth);

#ifdef __cplusplus
}
#endif
#endif
