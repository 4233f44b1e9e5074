// Windowed (grid-guided) Hamming matching for gfx950 (wave64): the
// SearchByProjection overloads of Tracking and SearchForInitialization.
// Reference lines are relative to /root/reference/src/ORB_SLAM2/.
//
// Three kernels per batch of (query set, train frame) pairs:
//   k_grid_build     one workgroup per pair: the 64x48 keypoint grid of the train
//                    frame (Frame::AssignFeaturesToGrid / PosInGrid,
//                    src/Frame.cc:270-293, 485-500), stored column-major with the
//                    keypoint coordinates re-ordered into grid order;
//   k_window_topk_lanes  one lane per query, whole chip: enumerates the window
//                    (Frame::GetFeaturesInArea, src/Frame.cc:397-474), evaluates
//                    the Hamming distances and caches the kTopK best candidates
//                    (k_window_topk: the same for the vocabulary-node segments of
//                    SearchByBoW / SearchForTriangulation, a wave per query);
//   k_window_resolve one workgroup per pair: reproduces the reference's
//                    sequential semantics (a train keypoint claimed by an earlier
//                    query is skipped by later ones) as the fixed point of
//                    "query i takes its best candidate not claimed by an accepted
//                    query j < i" -- unique because query i depends on j < i only --
//                    then the rotation-histogram check.
// Candidate order inside a window is the reference's (grid column outer, row
// inner, insertion order in the cell): with column-major cells one window
// column is a contiguous run of grid entries, and the ordinal of an entry in
// the concatenated runs is its position in the reference's candidate vector.
#include "match_internal.h"
#include "orbgpu_internal.h"

namespace orbgpu {

constexpr int kCells = ORBGPU_GRID_COLS * ORBGPU_GRID_ROWS;
constexpr unsigned long long kNoKey = ~0ull;

struct Top2 {
    unsigned long long a, b;  // two smallest keys, a <= b
};
__device__ __forceinline__ void top2_push(Top2 &t, unsigned long long k)
{
    if (k < t.a) { t.b = t.a; t.a = k; }
    else if (k < t.b) t.b = k;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o)
{
    const int lo = __shfl_xor((int)(v & 0xFFFFFFFFull), o), hi = __shfl_xor((int)(v >> 32), o);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src)
{
    const int lo = __shfl((int)(v & 0xFFFFFFFFull), src), hi = __shfl((int)(v >> 32), src);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ void top2_wave_reduce(Top2 &t2)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long oa = shfl_xor_u64(t2.a, o), ob = shfl_xor_u64(t2.b, o);
        const unsigned long long m1 = min(t2.a, oa);
        const unsigned long long m2 = min(max(t2.a, oa), min(t2.b, ob));
        t2.a = m1;
        t2.b = m2;
    }
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, shfl_xor_u64(v, o));
    return v;
}

// key = distance << 40 | window ordinal << 20 | train index
__device__ __forceinline__ int key_dist(unsigned long long k) { return (int)(k >> 40); }
__device__ __forceinline__ int key_train(unsigned long long k) { return (int)(k & 0xFFFFFull); }
// cached candidates are stored in key order as distance << 20 | train index
constexpr uint32_t kNoCand = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t cand_pack(unsigned long long k)
{
    return k == kNoKey ? kNoCand : ((uint32_t)key_dist(k) << 20) | (uint32_t)key_train(k);
}
__device__ __forceinline__ int cand_dist(uint32_t c) { return (int)(c >> 20); }
__device__ __forceinline__ int cand_train(uint32_t c) { return (int)(c & 0xFFFFFu); }

struct PairCtx {
    int nq, n2;
    const orbgpu_wquery *Q;
    const uint32_t *QD, *TD;
    const orbgpu_keypoint *K2;
    const float *UR;
    const uint8_t *OCC;
    int32_t *assign, *qmatch;
    uint32_t *topk;    // [query][kTopK] cached candidates, ascending key order, kNoCand padded
    int32_t *cell_start, *head, *ncand, *sel0, *sel1, *sd0, *sd1, *nxt, *full;
    const int32_t *sd_cur;  // distances of the current selections (SearchForInitialization claims)
    int4 *grec;        // grid-ordered train keypoints: {x bits, y bits, octave, index}
    uint32_t *gdesc;   // grid-ordered train descriptors
    float invW, invH;
};

__device__ __forceinline__ PairCtx make_ctx(const MatchArgs &A, int pair)
{
    PairCtx P;
    // the counts come from device memory; every buffer of the pair is sized by the capacities
    P.nq = min(max(A.nq[pair], 0), A.q_cap);
    P.n2 = min(max(A.n2[pair], 0), A.t_cap);
    P.Q = A.q + (size_t)pair * A.q_cap;
    P.QD = (const uint32_t *)(A.qdesc + (size_t)pair * A.q_cap * 32);
    P.K2 = A.kps2 + (size_t)pair * A.t_cap;
    P.TD = (const uint32_t *)(A.desc2 + (size_t)pair * A.t_cap * 32);
    P.UR = A.uright2 ? A.uright2 + (size_t)pair * A.t_cap : nullptr;
    P.OCC = A.occupied ? A.occupied + (size_t)pair * A.t_cap : nullptr;
    P.assign = A.assign + (size_t)pair * A.t_cap;
    P.qmatch = A.qmatch ? A.qmatch + (size_t)pair * A.q_cap : nullptr;
    int32_t *s = A.scratch + (size_t)pair * match_scratch_ints(A.q_cap, A.t_cap);
    P.topk = (uint32_t *)s;            s += kTopK * (size_t)A.q_cap;
    P.cell_start = s;                  s += 3080;
    P.grec = (int4 *)s;                s += 4 * (size_t)A.t_cap;
    P.gdesc = (uint32_t *)s;           s += 8 * (size_t)A.t_cap;
    P.head = s;                        s += A.t_cap;
    P.ncand = s;                       s += A.q_cap;
    P.sel0 = s;                        s += A.q_cap;
    P.sel1 = s;                        s += A.q_cap;
    P.sd0 = s;                         s += A.q_cap;
    P.sd1 = s;                         s += A.q_cap;
    P.sd_cur = P.sd0;
    P.nxt = s;                         s += A.q_cap;
    P.full = s;
    if (A.grid) {  // the train frame's own grid (built by k_grid_build with the same t_cap)
        int32_t *g = const_cast<int32_t *>(A.grid);
        P.cell_start = g;
        P.grec = (int4 *)(g + 3080);
        P.gdesc = (uint32_t *)(g + 3080 + 4 * (size_t)A.t_cap);
    }
    P.invW = A.invW;
    P.invH = A.invH;
    return P;
}

// ---------------------------------------------------------------------------
// grid
// ---------------------------------------------------------------------------
constexpr int kGridLdsIdx = 4096;  // keypoints of a frame whose grid order is sorted in LDS
__global__ __launch_bounds__(1024) void k_grid_build(MatchArgs A)
{
    __shared__ int cs[kCells + 1];
    __shared__ int cursor[kCells];
    __shared__ int scan_lds[33];
    __shared__ int32_t gidx_lds[kGridLdsIdx];
    const int pair = blockIdx.x, tid = threadIdx.x, nt_ = blockDim.x;
    const PairCtx P = make_ctx(A, pair);
    const int n2 = P.n2;
    for (int i = tid; i <= kCells; i += nt_) cs[i] = 0;
    for (int i = tid; i < kCells; i += nt_) cursor[i] = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += nt_) {
        const int px = (int)roundf(__fmul_rn(__fsub_rn(P.K2[i].x, A.gminX), P.invW));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(P.K2[i].y, A.gminY), P.invH));
        if (px >= 0 && px < ORBGPU_GRID_COLS && py >= 0 && py < ORBGPU_GRID_ROWS && (A.grid_levels == 0 || P.K2[i].octave < A.grid_levels))
            atomicAdd(&cs[px * ORBGPU_GRID_ROWS + py], 1);
    }
    __syncthreads();
    {
        int carry = 0;
        for (int b = 0; b < kCells; b += nt_) {
            const int cidx = b + tid;
            const int v = cidx < kCells ? cs[cidx] : 0;
            int total;
            const int ex = block_scan_excl(v, &total, scan_lds);
            if (cidx < kCells) cs[cidx] = carry + ex;
            carry += total;
        }
        if (tid == 0) cs[kCells] = carry;
    }
    __syncthreads();
    // fetch-and-add gives an arbitrary in-cell order, restored below.  The index list lives in LDS when the frame's
    // keypoints fit (every access of the per-cell sort would otherwise be a round trip to global memory), else in
    // the head array, which is free until the resolver runs.
    auto order_and_write = [&](auto *gidx) {
        for (int i = tid; i < n2; i += nt_) {
            const int px = (int)roundf(__fmul_rn(__fsub_rn(P.K2[i].x, A.gminX), P.invW));
            const int py = (int)roundf(__fmul_rn(__fsub_rn(P.K2[i].y, A.gminY), P.invH));
            if (px >= 0 && px < ORBGPU_GRID_COLS && py >= 0 && py < ORBGPU_GRID_ROWS && (A.grid_levels == 0 || P.K2[i].octave < A.grid_levels)) {
                const int cidx = px * ORBGPU_GRID_ROWS + py;
                gidx[cs[cidx] + atomicAdd(&cursor[cidx], 1)] = i;
            }
        }
        __syncthreads();
        // ... so restore insertion (ascending index) order inside every cell
        for (int cidx = tid; cidx < kCells; cidx += nt_) {
            const int s = cs[cidx], e = cs[cidx + 1];
            for (int i = s + 1; i < e; i++) {
                const int v = gidx[i];
                int j = i - 1;
                while (j >= s && gidx[j] > v) { gidx[j + 1] = gidx[j]; j--; }
                gidx[j + 1] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i <= kCells; i += nt_) P.cell_start[i] = cs[i];
        const int ng = cs[kCells];
        for (int k = tid; k < ng; k += nt_) {
            const int t = gidx[k];
            const orbgpu_keypoint kp = P.K2[t];
            P.grec[k] = make_int4(__float_as_int(kp.x), __float_as_int(kp.y), kp.octave, t);
        }
        // descriptors in grid order: 8 lanes copy one 32-byte descriptor
        for (int w = tid; w < ng * 8; w += nt_) P.gdesc[w] = P.TD[(size_t)gidx[w >> 3] * 8 + (w & 7)];
    };
    if (n2 <= kGridLdsIdx) order_and_write(gidx_lds);
    else order_and_write(P.head);
}

// ---------------------------------------------------------------------------
// window enumeration (one wave per query)
// ---------------------------------------------------------------------------
// Is train keypoint t unavailable to query i given the accepted selections of
// earlier queries?  SearchByProjection: head[t] = lowest index of an accepted
// query whose match marks t occupied (the keypoint then holds a map point).
// SearchForInitialization: head/nxt = list of the accepted selectors of t; t is
// skipped when an earlier selector matched it at a distance <= dist
// (vMatchedDistance[i2] <= dist, src/ORBmatcher.cc:477).
__device__ __forceinline__ bool claimed_by_earlier(const PairCtx &P, int mode, int cur, int i, int t, int dist)
{
    if (mode != 3) return P.head[t] < i;
    for (int j = P.head[t]; j >= 0; j = P.nxt[j])
        if (j < i && P.sd_cur[j] <= dist) return true;
    return false;
}

// Can a candidate at this distance influence the decision of its query?  A
// candidate beyond the acceptance threshold is never selected, and as the
// runner-up it only matters while it can still fail the ratio test of decide():
// dropping the others here leaves every decision unchanged and keeps the
// per-query caches short.
__device__ __forceinline__ bool cand_useful(const MatchArgs &A, int dist)
{
    if (A.mode == 3) return dist <= ORBGPU_TH_LOW || !((float)ORBGPU_TH_LOW < __fmul_rn((float)dist, A.nnratio));
    if (dist <= A.th_dist) return true;
    if (A.mode == 4) return !((float)A.th_dist < __fmul_rn(A.nnratio, (float)dist));
    if (A.mode >= 5) return false;  // modes 5 and 6: only candidates within the threshold matter
    return A.mode == 0 && (float)A.th_dist > __fmul_rn(A.nnratio, (float)dist);
}

// Reprojection chi-square gate of Fuse (src/ORBmatcher.cc:945-975): 7.8 with a stereo
// measurement, 5.99 without; the products are float, the comparisons double.
__device__ __forceinline__ bool reprojection_gate(const MatchArgs &A, const PairCtx &P, const orbgpu_wquery &q, const int4 &rec, int t)
{
    const float ex = __fsub_rn(q.u, __int_as_float(rec.x)), ey = __fsub_rn(q.v, __int_as_float(rec.y));
    const float inv = A.inv_sigma2.s[rec.z];
    float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
    if (P.UR && P.UR[t] >= 0) {
        const float er = __fsub_rn(q.ur, P.UR[t]);
        e2 = __fadd_rn(e2, __fmul_rn(er, er));
        return !((double)__fmul_rn(e2, inv) > 7.8);
    }
    return !((double)__fmul_rn(e2, inv) > 5.99);
}

// One 64-candidate chunk into the per-query cache wk (kTopK keys, kNoKey = free): nk = entries in
// use, npass = candidates seen so far.
__device__ __forceinline__ void cache_insert_wave(unsigned long long key, unsigned long long *wk, int &nk, int &npass)
{
    const int lane = lane_id();
    // The cache is an unordered set (the resolver takes minima): while it has
    // room the chunk's candidates are appended by ballot rank; once it would
    // overflow, candidates are taken in ascending order and evict the
    // largest cached key, so the cache ends as the kTopK smallest keys.
    const unsigned long long vb = __ballot(key != kNoKey);
    const int nv = __popcll(vb);
    npass += nv;
    if (nk + nv <= kTopK) {
        if (key != kNoKey) wk[nk + __popcll(vb & ((1ull << lane) - 1ull))] = key;
        nk += nv;
        __builtin_amdgcn_wave_barrier();
    } else {
        uint32_t k32 = key == kNoKey ? 0xFFFFFFFFu : ((uint32_t)(key >> 40) << 22) | (uint32_t)((key >> 20) & 0xFFFFFu);
        for (int rnd = 0; rnd < 64; rnd++) {
            uint32_t m = k32;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
            if (m == 0xFFFFFFFFu) break;
            const int src = __ffsll((long long)__ballot(k32 == m)) - 1;
            const unsigned long long mk = shfl_u64(key, src);
            if (lane == src) k32 = 0xFFFFFFFFu;
            if (nk < kTopK) {
                if (lane == 0) wk[nk] = mk;
                nk++;
                __builtin_amdgcn_wave_barrier();
                continue;
            }
            // largest cached key and its slot
            unsigned long long mx = lane < kTopK ? wk[lane] : 0ull;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx = max(mx, shfl_xor_u64(mx, o));
            mx = shfl_u64(mx, 0);
            if (mk >= mx) break;  // later candidates of this chunk are larger still
            const int slot = __ffsll((long long)__ballot(lane < kTopK && wk[lane] == mx)) - 1;
            if (lane == 0) wk[slot] = mk;
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// DYN = false: cache the kTopK best candidates of the window in wk (LDS, sorted)
// and count the candidates; DYN = true: apply the claims and return the two best.
template <bool DYN>
__device__ __forceinline__ void scan_window(const MatchArgs &A, const PairCtx &P, int cur, int i, const orbgpu_wquery &q,
                                            int *wl, int *ws, unsigned long long *wk, Top2 &t2, int &npass)
{
    const int lane = lane_id(), mode = A.mode;
    const float r = q.r;
    t2.a = t2.b = kNoKey;
    npass = 0;
    int nk = 0;  // valid entries in wk
    const int c0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.u, A.qminX), r), P.invW)));
    const int c1 = min(ORBGPU_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.u, A.qminX), r), P.invW)));
    const int r0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.v, A.qminY), r), P.invH)));
    const int r1 = min(ORBGPU_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.v, A.qminY), r), P.invH)));
    if (c0 >= ORBGPU_GRID_COLS || c1 < 0 || r0 >= ORBGPU_GRID_ROWS || r1 < 0) return;
    const int ncols = c1 - c0 + 1;
    int s = 0, len = 0;
    if (lane < ncols) {
        s = P.cell_start[(c0 + lane) * ORBGPU_GRID_ROWS + r0];
        len = P.cell_start[(c0 + lane) * ORBGPU_GRID_ROWS + r1 + 1] - s;
    }
    int inc = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    const int total = __shfl(inc, 63);
    if (total == 0) return;
    ws[lane] = s;
    wl[lane + 1] = inc;
    if (lane == 0) wl[0] = 0;
    __builtin_amdgcn_wave_barrier();
    const bool check = (q.minLevel > 0) || (q.maxLevel >= 0);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) qd[k] = P.QD[(size_t)i * 8 + k];
    for (int base = 0; base < total; base += 64) {
        const int j = base + lane;
        unsigned long long key = kNoKey;
        if (j < total) {
            int lo = 0, hi = ncols - 1;  // column c with wl[c] <= j < wl[c+1]
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (wl[mid] <= j) lo = mid; else hi = mid - 1;
            }
            const int k = ws[lo] + (j - wl[lo]);
            // record and descriptor of grid entry k: independent loads
            const int4 rec = P.grec[k];
            const uint4 d0 = reinterpret_cast<const uint4 *>(P.gdesc)[2 * (size_t)k];
            const uint4 d1 = reinterpret_cast<const uint4 *>(P.gdesc)[2 * (size_t)k + 1];
            const int oct = rec.z, t = rec.w;
            bool ok = true;
            if (check) {
                if (oct < q.minLevel) ok = false;
                if (q.maxLevel >= 0 && oct > q.maxLevel) ok = false;
            }
            if (ok && !(fabsf(__fsub_rn(__int_as_float(rec.x), q.u)) < r && fabsf(__fsub_rn(__int_as_float(rec.y), q.v)) < r)) ok = false;
            if (ok && mode != 3) {
                if (P.OCC && P.OCC[t]) ok = false;
                if (ok && mode < 2 && P.UR && P.UR[t] > 0 && fabsf(__fsub_rn(q.ur, P.UR[t])) > r) ok = false;
                if (ok && mode == 6 && A.chi2_gate) ok = reprojection_gate(A, P, q, rec, t);
            }
            if (ok) {
                const int dist = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) +
                                 __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
                if (cand_useful(A, dist) && (!DYN || !claimed_by_earlier(P, mode, cur, i, t, dist)))
                    key = ((unsigned long long)dist << 40) | ((unsigned long long)j << 20) | (unsigned)t;
            }
        }
        if (DYN) {
            top2_push(t2, key);
        } else {
            cache_insert_wave(key, wk, nk, npass);
        }
    }
    if (DYN) top2_wave_reduce(t2);
}

// The candidates of a SearchByBoW query are the train features of its vocabulary node, in the
// node's index order (src/ORBmatcher.cc:205-226, 621-647); one wave per query.
template <bool DYN>
__device__ __forceinline__ void scan_segment(const MatchArgs &A, const PairCtx &P, int i, const orbgpu_wquery &q,
                                             unsigned long long *wk, Top2 &t2, int &npass)
{
    const int lane = lane_id();
    t2.a = t2.b = kNoKey;
    npass = 0;
    int nk = 0;
    const int b = A.seg_start[q.minLevel], total = A.seg_start[q.minLevel + 1] - b;
    if (total <= 0) return;
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) qd[k] = P.QD[(size_t)i * 8 + k];
    // SearchForTriangulation: epipolar line of the query in the train image, l = x1' F12 (src/ORBmatcher.cc:143-145)
    const bool tri = A.mode == 5, stereo1 = (q.flags & 4) != 0;
    float la = 0.f, lb = 0.f, lc = 0.f;
    if (tri) {
        la = __fadd_rn(__fadd_rn(__fmul_rn(q.u, A.F12[0]), __fmul_rn(q.v, A.F12[3])), A.F12[6]);
        lb = __fadd_rn(__fadd_rn(__fmul_rn(q.u, A.F12[1]), __fmul_rn(q.v, A.F12[4])), A.F12[7]);
        lc = __fadd_rn(__fadd_rn(__fmul_rn(q.u, A.F12[2]), __fmul_rn(q.v, A.F12[5])), A.F12[8]);
    }
    for (int base = 0; base < total; base += 64) {
        const int j = base + lane;
        unsigned long long key = kNoKey;
        if (j < total) {
            const int t = A.seg_index[b + j];
            bool ok = !(P.OCC && P.OCC[t]);
            const bool stereo2 = tri && P.UR && P.UR[t] >= 0;
            if (tri && A.only_stereo && !stereo2) ok = false;  // :779-783
            if (ok) {
                const uint4 d0 = reinterpret_cast<const uint4 *>(P.TD)[2 * (size_t)t];
                const uint4 d1 = reinterpret_cast<const uint4 *>(P.TD)[2 * (size_t)t + 1];
                const int dist = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) +
                                 __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
                if (tri) {
                    if (dist <= A.th_dist) {  // :789
                        const orbgpu_keypoint k2 = P.K2[t];
                        bool pass = true;
                        if (!stereo1 && !stereo2) {  // too close to the epipole (:794-800)
                            const float dx = __fsub_rn(A.ex, k2.x), dy = __fsub_rn(A.ey, k2.y);
                            if (__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)) < __fmul_rn(100.0f, A.scale2.s[k2.octave])) pass = false;
                        }
                        if (pass) {  // CheckDistEpipolarLine (:147-156), the 3.84 comparison is done in double
                            const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, k2.x), __fmul_rn(lb, k2.y)), lc);
                            const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
                            pass = den != 0.0f && (double)__fdiv_rn(__fmul_rn(num, num), den) < __dmul_rn(3.84, (double)A.sigma2.s[k2.octave]);
                        }
                        // `dist > bestDist` skips (:789): among equal distances the last candidate wins
                        if (pass) key = ((unsigned long long)dist << 40) | ((unsigned long long)(0xFFFFFu - (unsigned)j) << 20) | (unsigned)t;
                    }
                } else if (cand_useful(A, dist) && (!DYN || !claimed_by_earlier(P, A.mode, 0, i, t, dist))) {
                    key = ((unsigned long long)dist << 40) | ((unsigned long long)j << 20) | (unsigned)t;
                }
            }
        }
        if (DYN) top2_push(t2, key);
        else cache_insert_wave(key, wk, nk, npass);
    }
    if (DYN) top2_wave_reduce(t2);
}

// Two queries per wave, one per 32-lane half: the window setup and the final
// ordering of the cache are shared by both, and most windows hold fewer than 32
// candidates.  Window c0 .. c0+ncols-1 x r0 .. r1 (ncols <= 32; 0 = nothing to
// scan); wl[33], ws[32] and wk[kTopK] belong to this half.
__device__ __forceinline__ void scan_window_half(const MatchArgs &A, const PairCtx &P, int i, const orbgpu_wquery &q, int c0,
                                                 int ncols, int r0, int r1, int *wl, int *ws, unsigned long long *wk, int &npass)
{
    const int lane = lane_id(), sub = lane & 31, hs = lane & 32, mode = A.mode;
    const float r = q.r;
    npass = 0;
    int nk = 0;  // valid entries in wk
    int s = 0, len = 0;
    if (sub < ncols) {
        s = P.cell_start[(c0 + sub) * ORBGPU_GRID_ROWS + r0];
        len = P.cell_start[(c0 + sub) * ORBGPU_GRID_ROWS + r1 + 1] - s;
    }
    int inc = len;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        const int t = __shfl_up(inc, o, 32);
        if (sub >= o) inc += t;
    }
    const int total = __shfl(inc, 31, 32);
    const int tmax = max(total, __shfl_xor(total, 32));
    if (tmax == 0) return;
    ws[sub] = s;
    wl[sub + 1] = inc;
    if (sub == 0) wl[0] = 0;
    __builtin_amdgcn_wave_barrier();
    const bool check = (q.minLevel > 0) || (q.maxLevel >= 0);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) qd[k] = total ? P.QD[(size_t)i * 8 + k] : 0u;
    for (int base = 0; base < tmax; base += 32) {
        const int j = base + sub;
        unsigned long long key = kNoKey;
        if (j < total) {
            int lo = 0, hi = ncols - 1;  // column c with wl[c] <= j < wl[c+1]
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (wl[mid] <= j) lo = mid; else hi = mid - 1;
            }
            const int k = ws[lo] + (j - wl[lo]);
            const int4 rec = P.grec[k];
            const int oct = rec.z, t = rec.w;
            bool ok = true;
            if (check) {
                if (oct < q.minLevel) ok = false;
                if (q.maxLevel >= 0 && oct > q.maxLevel) ok = false;
            }
            if (ok && !(fabsf(__fsub_rn(__int_as_float(rec.x), q.u)) < r && fabsf(__fsub_rn(__int_as_float(rec.y), q.v)) < r)) ok = false;
            if (ok && mode != 3) {
                if (P.OCC && P.OCC[t]) ok = false;
                if (ok && mode < 2 && P.UR && P.UR[t] > 0 && fabsf(__fsub_rn(q.ur, P.UR[t])) > r) ok = false;
                if (ok && mode == 6 && A.chi2_gate) ok = reprojection_gate(A, P, q, rec, t);
            }
            if (ok) {
                // the descriptor is fetched only for entries inside the window and the level range (a minority)
                const uint4 d0 = reinterpret_cast<const uint4 *>(P.gdesc)[2 * (size_t)k];
                const uint4 d1 = reinterpret_cast<const uint4 *>(P.gdesc)[2 * (size_t)k + 1];
                const int dist = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) +
                                 __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
                if (cand_useful(A, dist)) key = ((unsigned long long)dist << 40) | ((unsigned long long)j << 20) | (unsigned)t;
            }
        }
        // unordered cache, as in scan_window<false>, with every cross-lane step confined to the half
        const uint32_t vh = (uint32_t)(__ballot(key != kNoKey) >> hs);
        const int nv = __popc(vh);
        npass += nv;
        const bool fits = nk + nv <= kTopK;
        if (fits) {
            if (key != kNoKey) wk[nk + __popc(vh & ((1u << sub) - 1u))] = key;
            nk += nv;
        }
        __builtin_amdgcn_wave_barrier();
        if (__ballot(!fits) != 0ull) {
            bool active = !fits;
            uint32_t k32 = (active && key != kNoKey) ? ((uint32_t)(key >> 40) << 22) | (uint32_t)((key >> 20) & 0xFFFFFu) : 0xFFFFFFFFu;
            for (int rnd = 0; rnd < 32; rnd++) {
                uint32_t m = k32;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
                if (m == 0xFFFFFFFFu) active = false;
                if (__ballot(active) == 0ull) break;
                const uint32_t eq = (uint32_t)(__ballot(active && k32 == m) >> hs);
                const int src = hs + (eq ? __ffs((int)eq) - 1 : 0);
                const unsigned long long mk = shfl_u64(key, src);
                if (active && lane == src) k32 = 0xFFFFFFFFu;
                // largest cached key and its slot (used once the cache is full)
                unsigned long long mx = sub < kTopK ? wk[sub] : 0ull;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) mx = max(mx, shfl_xor_u64(mx, o));
                mx = shfl_u64(mx, hs);
                const uint32_t sl = (uint32_t)(__ballot(sub < kTopK && wk[sub] == mx) >> hs);
                if (active) {
                    if (nk < kTopK) {
                        if (sub == 0) wk[nk] = mk;
                        nk++;
                    } else if (mk >= mx) {
                        active = false;  // later candidates of this chunk are larger still
                    } else if (sub == 0) {
                        wk[__ffs((int)sl) - 1] = mk;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// Two queries per wave: the segment modes (SearchByBoW, SearchForTriangulation) always, the window modes for small
// launches (a frame or two: the candidates of a window are spread over 32 lanes, which keeps the latency of a
// single frame low; large batches use k_window_topk_lanes).
__global__ __launch_bounds__(256) void k_window_topk(MatchArgs A)
{
    __shared__ int wl[4][66], ws[4][64];
    __shared__ unsigned long long wk[4][2][kTopK];
    __shared__ uint32_t wp[4][2][kTopK];
    const int pair = blockIdx.y, w = threadIdx.x >> 6, lane = lane_id(), sub = lane & 31, half = lane >> 5;
    const int i0 = (blockIdx.x * 4 + w) * 2, i = i0 + half;
    const PairCtx P = make_ctx(A, pair);
    if (i0 >= P.nq) return;
    const bool inq = i < P.nq;
    if (sub < kTopK) wk[w][half][sub] = kNoKey;
    __builtin_amdgcn_wave_barrier();
    const orbgpu_wquery q = P.Q[inq ? i : i0];
    int npass = 0;
    if (A.mode == 4 || A.mode == 5) {
        // vocabulary-node segments: the full-wave scan, one query after the other
        for (int h = 0; h < 2 && i0 + h < P.nq; h++) {
            const orbgpu_wquery qh = P.Q[i0 + h];
            Top2 t2;
            int np = 0;
            if (qh.flags & 1) scan_segment<false>(A, P, i0 + h, qh, wk[w][h], t2, np);
            if (half == h) npass = np;
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        // window of this half's query (Frame::GetFeaturesInArea, src/Frame.cc:402-420)
        const float r = q.r;
        const int c0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.u, A.qminX), r), P.invW)));
        const int c1 = min(ORBGPU_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.u, A.qminX), r), P.invW)));
        const int r0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.v, A.qminY), r), P.invH)));
        const int r1 = min(ORBGPU_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.v, A.qminY), r), P.invH)));
        const bool live = inq && (q.flags & 1) && !(c0 >= ORBGPU_GRID_COLS || c1 < 0 || r0 >= ORBGPU_GRID_ROWS || r1 < 0);
        const int ncols = live ? c1 - c0 + 1 : 0;
        if (__ballot(ncols > 32) == 0ull) {
            scan_window_half(A, P, i, q, c0, ncols, r0, r1, wl[w] + 33 * half, ws[w] + 32 * half, wk[w][half], npass);
        } else {
            // a window wider than 32 grid columns: the full-wave scan, one query after the other
            for (int h = 0; h < 2 && i0 + h < P.nq; h++) {
                const orbgpu_wquery qh = P.Q[i0 + h];
                Top2 t2;
                int np = 0;
                if (qh.flags & 1) scan_window<false>(A, P, 0, i0 + h, qh, wl[w], ws[w], wk[w][h], t2, np);
                if (half == h) npass = np;
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    // the cache is a set of distinct keys: rank = position in ascending key order
    const unsigned long long my = sub < kTopK ? wk[w][half][sub] : kNoKey;
    int rank = 0;
#pragma unroll
    for (int k = 0; k < kTopK; k++) rank += wk[w][half][k] < my;
    if (sub < kTopK) wp[w][half][sub] = kNoCand;
    __builtin_amdgcn_wave_barrier();
    if (my != kNoKey) wp[w][half][rank] = cand_pack(my);
    __builtin_amdgcn_wave_barrier();
    if (inq && sub < kTopK) P.topk[(size_t)i * kTopK + sub] = wp[w][half][sub];
    if (inq && sub == 0) P.ncand[i] = npass;
}

// Window modes (SearchByProjection x3, SearchForInitialization, the area search): one lane per query.  A window
// holds few grid entries (a handful at tracking radii), so a lane walks its own window -- grid column after grid
// column, each a contiguous run of grid entries, i.e. the reference's candidate order -- and keeps its kTopK
// best candidates as a sorted list in registers; the wave runs as long as its busiest lane.
// cs / grec / gdesc: the pair's cell starts, grid records and grid-ordered descriptors, in global memory or staged in LDS.
__device__ __forceinline__ void lanes_query(const MatchArgs &A, const PairCtx &P, int i, const int32_t *cs, const int4 *grec,
                                            const uint4 *gdesc)
{
    const int mode = A.mode;
    const orbgpu_wquery q = P.Q[i];
    uint32_t c[kTopK];  // distance << 20 | train index, ascending distance, equal distances in scan order
#pragma unroll
    for (int s = 0; s < kTopK; s++) c[s] = kNoCand;
    int npass = 0;
    // window of the query (Frame::GetFeaturesInArea, src/Frame.cc:402-420)
    const float r = q.r;
    const int c0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.u, A.qminX), r), P.invW)));
    const int c1 = min(ORBGPU_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.u, A.qminX), r), P.invW)));
    const int r0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.v, A.qminY), r), P.invH)));
    const int r1 = min(ORBGPU_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.v, A.qminY), r), P.invH)));
    const bool live = (q.flags & 1) && !(c0 >= ORBGPU_GRID_COLS || c1 < 0 || r0 >= ORBGPU_GRID_ROWS || r1 < 0);
    const bool check = (q.minLevel > 0) || (q.maxLevel >= 0);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) qd[k] = live ? P.QD[(size_t)i * 8 + k] : 0u;
    // The walk is latency-bound (few waves, dependent loads), so every load is issued as early as its address is
    // known: the bounds of the next column's run while the current one is walked, the next entry's record and the
    // current entry's descriptor before the gates are evaluated.
    int col = live ? c0 : c1 + 1, k = 0, e = 0;
    int kn = 0, en = 0;  // run of column `col`
    if (col <= c1) {
        kn = cs[col * ORBGPU_GRID_ROWS + r0];
        en = cs[col * ORBGPU_GRID_ROWS + r1 + 1];
    }
    int4 rec = make_int4(0, 0, 0, 0);
    // opens the next non-empty grid column of the window (a short loop of its own: an iteration of the walk below
    // pays for its whole body, so it is spent on entries only)
    auto next_column = [&]() {
        while (k >= e && col <= c1) {
            k = kn;
            e = en;
            col++;
            if (col <= c1) {
                kn = cs[col * ORBGPU_GRID_ROWS + r0];
                en = cs[col * ORBGPU_GRID_ROWS + r1 + 1];
            }
        }
        if (k < e) rec = grec[k];
    };
    next_column();
    while (k < e) {
        int4 rec_next = grec[min(k + 1, e - 1)];
        const int oct = rec.z, t = rec.w;
        bool ok = true;
        if (check) {
            if (oct < q.minLevel) ok = false;
            if (q.maxLevel >= 0 && oct > q.maxLevel) ok = false;
        }
        if (ok && !(fabsf(__fsub_rn(__int_as_float(rec.x), q.u)) < r && fabsf(__fsub_rn(__int_as_float(rec.y), q.v)) < r)) ok = false;
        if (ok && mode != 3) {
            if (P.OCC && P.OCC[t]) ok = false;
            if (ok && mode < 2 && P.UR && P.UR[t] > 0 && fabsf(__fsub_rn(q.ur, P.UR[t])) > r) ok = false;
            if (ok && mode == 6 && A.chi2_gate) ok = reprojection_gate(A, P, q, rec, t);
        }
        uint32_t key = kNoCand;
        if (ok) {
            // only the lanes whose entry passed the gates fetch its descriptor: the loads of a wave are priced by the
            // lines they touch, and every lane's entry lies in a line of its own
            const uint4 d0 = gdesc[2 * (size_t)k];
            const uint4 d1 = gdesc[2 * (size_t)k + 1];
            const int dist = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) +
                             __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
            if (cand_useful(A, dist)) {
                key = ((uint32_t)dist << 20) | (uint32_t)t;
                npass++;
            }
        }
        k++;
        rec = rec_next;
        if (k >= e) next_column();
        // insert behind the entries of smaller or equal distance; slots are visited only while some lane of the
        // wave still has something to place (a lane is through once it carries an empty slot's content)
        uint32_t carry = key;
        const uint32_t kk = key | 0xFFFFFu;
        bool shifting = false;
#pragma unroll
        for (int s = 0; s < kTopK; s++) {
            if (__ballot(carry != kNoCand) == 0ull) break;
            shifting = shifting || kk < c[s];
            const uint32_t old = c[s];
            c[s] = shifting ? carry : old;
            carry = shifting ? old : carry;
        }
    }
    uint4 *out = reinterpret_cast<uint4 *>(P.topk + (size_t)i * kTopK);
#pragma unroll
    for (int s = 0; s < kTopK / 4; s++) out[s] = make_uint4(c[4 * s], c[4 * s + 1], c[4 * s + 2], c[4 * s + 3]);
    P.ncand[i] = npass;
}


__global__ __launch_bounds__(64) void k_window_topk_lanes(MatchArgs A)
{
    const int pair = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
    const PairCtx P = make_ctx(A, pair);
    if (i >= P.nq) return;
    lanes_query(A, P, i, P.cell_start, P.grec, reinterpret_cast<const uint4 *>(P.gdesc));
}

// The same walk with the pair's grid staged in LDS: one workgroup per pair copies the cell starts, the records and
// the descriptors of the train frame (coalesced, once), then every lane walks its window out of LDS.  Walking global
// memory, the 64 lanes of a load touch 64 different lines, and the wave is priced by those lines.
constexpr int kStagedThreads = 512;
__host__ __device__ inline size_t staged_lds_bytes(int t_cap) { return (size_t)t_cap * 48 + (size_t)(kCells + 1 + 3) / 4 * 16; }

__global__ __launch_bounds__(kStagedThreads) void k_window_topk_staged(MatchArgs A)
{
    extern __shared__ __align__(16) uint8_t staged[];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const PairCtx P = make_ctx(A, pair);
    uint4 *l_desc = reinterpret_cast<uint4 *>(staged);                                // [2 * t_cap]
    int4 *l_rec = reinterpret_cast<int4 *>(staged + (size_t)A.t_cap * 32);            // [t_cap]
    int32_t *l_cs = reinterpret_cast<int32_t *>(staged + (size_t)A.t_cap * 48);       // [kCells + 1]
    const int ng = P.cell_start[kCells];  // entries of the grid (keypoints inside the bounds)
    for (int k = tid; k < 2 * ng; k += kStagedThreads) l_desc[k] = reinterpret_cast<const uint4 *>(P.gdesc)[k];
    for (int k = tid; k < ng; k += kStagedThreads) l_rec[k] = P.grec[k];
    for (int k = tid; k <= kCells; k += kStagedThreads) l_cs[k] = P.cell_start[k];
    __syncthreads();
    // A lane takes a query from the front and its mirror from the back of the list: the extractor lists keypoints
    // level by level and the windows grow with the level, so the two together cost about the same for every lane,
    // and neighbouring lanes still walk windows of like size.
    const int nq = P.nq, half = (nq + 1) >> 1;
    for (int t = tid; t < half; t += kStagedThreads) {
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
            const int i = h ? nq - 1 - t : t;
            if (h && i == t) break;
            lanes_query(A, P, i, l_cs, l_rec, l_desc);
        }
    }
}

// Acceptance rules of the four matchers, given the best two available candidates.
__device__ __forceinline__ void decide(const MatchArgs &A, const PairCtx &P, uint32_t b1, uint32_t b2, int &newsel, int &newd)
{
    newsel = -1;
    newd = 0;
    if (b1 == kNoCand) return;
    const int best = cand_dist(b1), bidx = cand_train(b1);
    bool ok;
    if (A.mode == 3) {  // src/ORBmatcher.cc:496-499
        const float best2 = b2 != kNoCand ? (float)cand_dist(b2) : 2147483648.0f;
        ok = best <= ORBGPU_TH_LOW && (float)best < __fmul_rn(best2, A.nnratio);
    } else if (A.mode == 5) {  // src/ORBmatcher.cc:809: any surviving candidate is a match
        ok = true;
    } else if (A.mode == 6) {  // src/ORBmatcher.cc:981, 1117, 1258
        ok = best <= A.th_dist;
    } else if (A.mode == 4) {  // src/ORBmatcher.cc:228-230, 649-651 (bestDist2 starts at 256)
        const float best2 = b2 != kNoCand ? (float)cand_dist(b2) : 256.0f;
        ok = best <= A.th_dist && (float)best < __fmul_rn(A.nnratio, best2);
    } else {
        ok = best <= A.th_dist;
        if (ok && A.mode == 0 && b2 != kNoCand) {  // ratio test only within one pyramid level (src/ORBmatcher.cc:118-121)
            const int best2 = cand_dist(b2), bidx2 = cand_train(b2);
            if (P.K2[bidx].octave == P.K2[bidx2].octave && (float)best > __fmul_rn(A.nnratio, (float)best2)) ok = false;
        }
    }
    if (ok) { newsel = bidx; newd = best; }
}

__global__ __launch_bounds__(1024) void k_window_resolve(MatchArgs A, int lds_state)
{
    // dynamic LDS (lds_state != 0): selection state, claim words and flags of this
    // pair, so an iteration of the fixed-point loop touches no global memory except
    // the (rare) window rescans
    extern __shared__ __align__(16) int dyn_state[];
    __shared__ int wl[16][65], ws[16][64];
    __shared__ int hist[ORBGPU_HISTO_LENGTH];
    __shared__ int sh_changed, sh_nrescan, sh_keep[3];

    const int pair = blockIdx.x, tid = threadIdx.x, nt_ = blockDim.x;
    const int lane = lane_id(), wid = tid >> 6, nw = nt_ >> 6;
    PairCtx P = make_ctx(A, pair);
    const int nq = P.nq, n2 = P.n2, mode = A.mode;
    const bool need2 = (mode == 0 || mode == 3 || mode == 4);
    if (lds_state) {
        int32_t *s = dyn_state;
        P.sel0 = s; s += A.q_cap;
        P.sel1 = s; s += A.q_cap;
        P.sd0 = s;  s += A.q_cap;
        P.sd1 = s;  s += A.q_cap;
        P.nxt = s;  s += A.q_cap;
        P.full = s; s += A.q_cap;
        P.head = s;
        P.sd_cur = P.sd0;
    }

    int32_t *selc = P.sel0, *seln = P.sel1, *sdc = P.sd0, *sdn = P.sd1;  // current / next state
    for (int i = tid; i < nq; i += nt_) { selc[i] = -1; sdc[i] = 0; }
    // the cached candidates never change: keep those of the thread's first query in registers
    uint32_t key0[kTopK];
#pragma unroll
    for (int k = 0; k < kTopK; k++) key0[k] = kNoCand;
    int ncand0 = 0;
    if (tid < nq) {
        const uint4 *kp = reinterpret_cast<const uint4 *>(P.topk + (size_t)tid * kTopK);
#pragma unroll
        for (int k = 0; k < kTopK / 4; k++) {
            const uint4 v = kp[k];
            key0[4 * k] = v.x; key0[4 * k + 1] = v.y; key0[4 * k + 2] = v.z; key0[4 * k + 3] = v.w;
        }
        ncand0 = P.ncand[tid];
    }
    const int qflags0 = tid < nq ? P.Q[tid].flags : 0;
    __syncthreads();

    // one query against the current claims: the first one or two cached candidates
    // (in key order) that no earlier query has claimed
    auto evaluate = [&](int i, const uint32_t (&key)[kTopK], int ncand) {
        uint32_t b1 = kNoCand, b2 = kNoCand;
        if (mode != 3) {
            int h[kTopK];  // claim words of all cached candidates: independent loads
#pragma unroll
            for (int k = 0; k < kTopK; k++) h[k] = P.head[key[k] == kNoCand ? 0 : cand_train(key[k])];
#pragma unroll
            for (int k = kTopK - 1; k >= 0; k--)
                if (key[k] != kNoCand && h[k] >= i) { b2 = b1; b1 = key[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < kTopK; k++) {
                const uint32_t kk = key[k];
                if (kk == kNoCand || b2 != kNoCand) continue;
                if (claimed_by_earlier(P, mode, 0, i, cand_train(kk), cand_dist(kk))) continue;
                if (b1 == kNoCand) b1 = kk; else b2 = kk;
            }
        }
        // a truncated cache that cannot supply what the decision needs says nothing about the rest of the window
        if (ncand > kTopK && (b1 == kNoCand || (need2 && b2 == kNoCand))) {
            P.full[atomicAdd(&sh_nrescan, 1)] = i;  // rescan the whole window below
        } else {
            int ns, nd;
            decide(A, P, b1, b2, ns, nd);
            if (ns != selc[i] || nd != sdc[i]) sh_changed = 1;
            seln[i] = ns;
            sdn[i] = nd;
        }
    };

    const int cur = 0;
    int n_iter = 0, n_full = 0;
    for (int it = 0; it <= nq + 1; it++) {
        n_iter++;
        for (int i = tid; i < n2; i += nt_) P.head[i] = (mode == 3) ? -1 : 0x7FFFFFFF;
        if (tid == 0) { sh_changed = 0; sh_nrescan = 0; }
        __syncthreads();
        for (int i = tid; i < nq; i += nt_) {
            const int s = selc[i];
            if (s < 0) continue;
            if (mode == 3) P.nxt[i] = atomicExch(&P.head[s], i);
            else if (mode < 5 && (mode == 2 || mode == 4 || ((i == tid ? qflags0 : P.Q[i].flags) & 2))) atomicMin(&P.head[s], i);
        }
        __syncthreads();
        if (tid < nq) evaluate(tid, key0, ncand0);
        for (int i = tid + nt_; i < nq; i += nt_) {
            uint32_t key[kTopK];
            const uint4 *kp = reinterpret_cast<const uint4 *>(P.topk + (size_t)i * kTopK);
#pragma unroll
            for (int k = 0; k < kTopK / 4; k++) {
                const uint4 v = kp[k];
                key[4 * k] = v.x; key[4 * k + 1] = v.y; key[4 * k + 2] = v.z; key[4 * k + 3] = v.w;
            }
            evaluate(i, key, P.ncand[i]);
        }
        __syncthreads();
        // queries the cache could not decide: one wave per listed query rescans its window
        const int nrescan = sh_nrescan;
        for (int r = wid; r < nrescan; r += nw) {
            const int i = P.full[r];
            n_full++;
            Top2 t2;
            int npass;
            if (mode == 4 || mode == 5) scan_segment<true>(A, P, i, P.Q[i], nullptr, t2, npass);
            else scan_window<true>(A, P, cur, i, P.Q[i], wl[wid], ws[wid], nullptr, t2, npass);
            if (lane == 0) {
                int ns, nd;
                decide(A, P, cand_pack(t2.a), cand_pack(t2.b), ns, nd);
                if (ns != selc[i] || nd != sdc[i]) sh_changed = 1;
                seln[i] = ns;
                sdn[i] = nd;
            }
        }
        __syncthreads();
        { int32_t *t1 = selc; selc = seln; seln = t1; int32_t *t2 = sdc; sdc = sdn; sdn = t2; }
        P.sd_cur = sdc;
        const int ch = sh_changed;
        __syncthreads();
        if (!ch) break;
    }

    // ---- outputs + rotation consistency ------------------------------------------
    int32_t *assign = P.assign;
    for (int i = tid; i < n2; i += nt_) assign[i] = -1;
    for (int i = tid; i < ORBGPU_HISTO_LENGTH; i += nt_) hist[i] = 0;
    if (tid == 0) sh_changed = 0;  // reused as the match counter
    __syncthreads();
    const float factor = 1.0f / ORBGPU_HISTO_LENGTH;  // the reference's 1/30 (src/ORBmatcher.cc:434)
    const bool ori = A.check_orientation && mode != 0;
    for (int i = tid; i < nq; i += nt_) {
        const int s = selc[i];
        if (s < 0) continue;
        atomicMax(&assign[s], i);  // a later selector overwrites (mode 3: steals) an earlier one
        if (mode != 3) atomicAdd(&sh_changed, 1);
        if (ori) {
            float rot = __fsub_rn(P.Q[i].angle, P.K2[s].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == ORBGPU_HISTO_LENGTH) bin = 0;
            P.nxt[i] = bin;  // the claim lists are no longer needed
            atomicAdd(&hist[bin], 1);
        }
    }
    __syncthreads();
    if (mode == 3) {
        // SearchForInitialization counts a train keypoint once (stolen matches are decremented, :502-506)
        for (int i = tid; i < n2; i += nt_)
            if (assign[i] >= 0) atomicAdd(&sh_changed, 1);
        __syncthreads();
    }
    if (ori) {
        if (tid == 0) {  // ComputeThreeMaxima (src/ORBmatcher.cc:1661-1704)
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < ORBGPU_HISTO_LENGTH; i++) {
                const int s = hist[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
                else if (s > max3) { max3 = s; i3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
            sh_keep[0] = i1; sh_keep[1] = i2; sh_keep[2] = i3;
        }
        __syncthreads();
        const int k1 = sh_keep[0], k2 = sh_keep[1], k3 = sh_keep[2];
        for (int i = tid; i < nq; i += nt_) {
            const int s = selc[i];
            if (s < 0) continue;
            const int bin = P.nxt[i];
            if (bin == k1 || bin == k2 || bin == k3) continue;
            if (mode == 3) {
                // only a live match is removed (:551-559); i is the only writer of assign[s] here
                if (assign[s] == i) { assign[s] = -1; atomicSub(&sh_changed, 1); }
            } else if (mode >= 5) {
                selc[i] = -1;  // vMatches12[idx1] = -1 (:856); train features are not exclusive in these matchers
                atomicSub(&sh_changed, 1);
            } else {
                // every entry of a losing bin clears its train keypoint and decrements (:1507-1517); -2 tells the
                // caller that the reference has set this keypoint's map point to NULL, -1 that it never touched it
                assign[s] = (mode == 1 || mode == 2) ? -2 : -1;
                atomicSub(&sh_changed, 1);
            }
        }
        __syncthreads();
    }
    if (P.qmatch) {
        for (int i = tid; i < nq; i += nt_) {
            const int s = selc[i];
            P.qmatch[i] = (s >= 0 && (mode >= 5 || assign[s] == i)) ? s : -1;
        }
    }
    if (tid == 0) A.nmatches[pair] = sh_changed;
    if (A.stats && lane == 0) {
        if (tid == 0) A.stats[4 * pair] = n_iter;
        atomicAdd(&A.stats[4 * pair + 1], n_full);
    }
}

__global__ __launch_bounds__(256) void k_shift_queries(const orbgpu_keypoint *__restrict__ kps, const int32_t *__restrict__ n,
                                                       int capacity, float du, float dv, float th, ScaleTab sc, float minX,
                                                       float minY, float maxX, float maxY, orbgpu_wquery *__restrict__ q)
{
    const int pair = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(n[pair], capacity)) return;
    const orbgpu_keypoint kp = kps[(size_t)pair * capacity + i];
    orbgpu_wquery w;
    w.u = __fadd_rn(kp.x, du);
    w.v = __fadd_rn(kp.y, dv);
    w.r = __fmul_rn(th, sc.s[min(max(kp.octave, 0), ORBGPU_MAX_LEVELS - 1)]);  // the octave is caller data
    w.minLevel = kp.octave - 1;
    w.maxLevel = kp.octave + 1;
    w.angle = kp.angle;
    w.ur = -1.0f;
    const bool inside = !(w.u < minX || w.u > maxX || w.v < minY || w.v > maxY);  // src/ORBmatcher.cc:1424-1427
    w.flags = inside ? 3 : 0;
    q[(size_t)pair * capacity + i] = w;
}

// SearchForInitialization queries (src/ORBmatcher.cc:444-453): a window around vbPrevMatched for every level-0
// keypoint of frame 1; and the update of vbPrevMatched with the matched keypoints of frame 2 (:566-568).
__global__ __launch_bounds__(256) void k_init_queries(const orbgpu_keypoint *__restrict__ kps1, int n1, const float *__restrict__ prev_xy,
                                                      int window, orbgpu_wquery *__restrict__ q)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n1) return;
    const orbgpu_keypoint kp = kps1[i];
    orbgpu_wquery w;
    w.u = prev_xy[2 * i];
    w.v = prev_xy[2 * i + 1];
    w.r = (float)window;
    w.minLevel = kp.octave;
    w.maxLevel = kp.octave;
    w.angle = kp.angle;
    w.ur = -1.0f;
    w.flags = kp.octave > 0 ? 0 : 1;
    q[i] = w;
}

__global__ __launch_bounds__(256) void k_init_update(const int32_t *__restrict__ matches12, int n1, const orbgpu_keypoint *__restrict__ kps2,
                                                     float *__restrict__ prev_xy)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n1) return;
    const int t = matches12[i];
    if (t >= 0) {
        prev_xy[2 * i] = kps2[t].x;
        prev_xy[2 * i + 1] = kps2[t].y;
    }
}

int launch_init_queries(hipStream_t st, const orbgpu_keypoint *kps1, int n1, const float *prev_xy, int window, orbgpu_wquery *q)
{
    hipLaunchKernelGGL(k_init_queries, dim3((n1 + 255) / 256), dim3(256), 0, st, kps1, n1, prev_xy, window, q);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

// The same queries for a batch of frame pairs, with vbPrevMatched = the keypoints' own positions (what Tracking sets
// when it starts an initialisation, src/Tracking.cc:592-594): pair p's keypoints at kps1 + p * capacity, n1[p] valid.
__global__ __launch_bounds__(256) void k_init_queries_batch(const orbgpu_keypoint *__restrict__ kps1, const int32_t *__restrict__ n1,
                                                            int capacity, int window, orbgpu_wquery *__restrict__ q)
{
    const int pair = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= min(n1[pair], capacity)) return;
    const orbgpu_keypoint kp = kps1[(size_t)pair * capacity + i];
    orbgpu_wquery w;
    w.u = kp.x;
    w.v = kp.y;
    w.r = (float)window;
    w.minLevel = kp.octave;
    w.maxLevel = kp.octave;
    w.angle = kp.angle;
    w.ur = -1.0f;
    w.flags = kp.octave > 0 ? 0 : 1;
    q[(size_t)pair * capacity + i] = w;
}

int launch_init_queries_batch(hipStream_t st, int npairs, const orbgpu_keypoint *kps1, const int32_t *n1, int capacity, int window,
                              orbgpu_wquery *q)
{
    hipLaunchKernelGGL(k_init_queries_batch, dim3((capacity + 255) / 256, npairs), dim3(256), 0, st, kps1, n1, capacity, window, q);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

int launch_init_update(hipStream_t st, const int32_t *matches12, int n1, const orbgpu_keypoint *kps2, float *prev_xy)
{
    hipLaunchKernelGGL(k_init_update, dim3((n1 + 255) / 256), dim3(256), 0, st, matches12, n1, kps2, prev_xy);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

int launch_match_window(hipStream_t st, const MatchArgs &A, int npairs)
{
    if (A.mode != 4 && A.mode != 5 && !A.grid) hipLaunchKernelGGL(k_grid_build, dim3(npairs), dim3(1024), 0, st, A);
    // a lane per query pays once there are enough queries to fill the chip; below that the half-wave scan is quicker
    if (A.mode == 4 || A.mode == 5 || (int64_t)A.q_cap * npairs < 65536)
        hipLaunchKernelGGL(k_window_topk, dim3((A.q_cap + 7) / 8, npairs), dim3(256), 0, st, A);
    else if (staged_lds_bytes(A.t_cap) <= 64 * 1024)  // two workgroups per CU
        hipLaunchKernelGGL(k_window_topk_staged, dim3(npairs), dim3(kStagedThreads), staged_lds_bytes(A.t_cap), st, A);
    else
        hipLaunchKernelGGL(k_window_topk_lanes, dim3((A.q_cap + 63) / 64, npairs), dim3(64), 0, st, A);
    // selection state in LDS when it fits (6 words per query + 1 per train keypoint)
    const size_t state = (6 * (size_t)A.q_cap + (size_t)A.t_cap) * 4;
    const int lds_state = state <= 48 * 1024;
    // the kernel holds 128 registers: 1024 threads fill a CU, 512 let two pairs share it (a batch of pairs then runs in
    // one round; the second query of a thread re-reads its candidates in every iteration)
    hipLaunchKernelGGL(k_window_resolve, dim3(npairs), dim3(npairs >= 256 ? 512 : 1024), lds_state ? state : 0, st, A, lds_state);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

// The keypoint grid of one frame alone (Frame::AssignFeaturesToGrid).
int launch_grid_only(hipStream_t st, const MatchArgs &A)
{
    hipLaunchKernelGGL(k_grid_build, dim3(1), dim3(1024), 0, st, A);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

int launch_shift_queries(hipStream_t st, int npairs, const orbgpu_keypoint *kps, const int32_t *n, int capacity, float du,
                         float dv, float th, const ScaleTab &sc, const float *b4, orbgpu_wquery *q)
{
    hipLaunchKernelGGL(k_shift_queries, dim3((capacity + 255) / 256, npairs), dim3(256), 0, st, kps, n, capacity, du, dv, th, sc,
                       b4[0], b4[1], b4[2], b4[3], q);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

}  // namespace orbgpu
