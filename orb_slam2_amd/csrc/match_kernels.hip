// HIP kernels of the ORB matcher for gfx950 (wave64): brute-force Hamming,
// windowed (grid-guided) Hamming matching with the reference's sequential
// semantics, and the stereo row-band matcher.
// Reference lines are relative to /root/reference/src/ORB_SLAM2/.
#include "orbgpu_internal.h"
#include "match_internal.h"

namespace orbgpu {

// median of three unsigned values (v_med3_u32)
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ int hamming8(const uint32_t (&a)[8], const uint32_t *__restrict__ b)
{
    int d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d += __popc(a[k] ^ b[k]);
    return d;
}

// ---------------------------------------------------------------------------
// Brute force: every query against every train descriptor.  A thread owns one
// query (8 VGPRs); the train descriptor index is wave-uniform so its 8 words
// come through the scalar path.  Per pair: 8 xor + 8 popcount-accumulate, one
// key build and two min/median updates.  key = dist << 20 | local index, so the
// two smallest keys are (first-min, second-min) in the reference's scan order
// (src/ORBmatcher.cc:216-225: strict '<', ascending index).
// The train set is cut into `nsplit` chunks (grid.y) to fill the machine;
// k_bf_merge folds the per-chunk partial results.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bf_partial(const uint32_t *__restrict__ Q, int nq, const uint32_t *__restrict__ T,
                                                    int nt, int tchunk, uint2 *__restrict__ part)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int t0 = blockIdx.y * tchunk, t1 = min(nt, t0 + tchunk);
    uint32_t a[8];
    const int qq = min(q, nq - 1);
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = Q[(size_t)qq * 8 + k];
    uint32_t k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu;
    int j = t0;
    for (; j + 4 <= t1; j += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t *tp = T + (size_t)(j + u) * 8;
            const uint32_t key = ((uint32_t)hamming8(a, tp) << 20) | (uint32_t)(j + u - t0);
            k2 = med3_u32(k1, k2, key);  // second smallest of {k1<=k2, key}
            k1 = min(k1, key);
        }
    }
    for (; j < t1; j++) {
        const uint32_t *tp = T + (size_t)j * 8;
        const uint32_t key = ((uint32_t)hamming8(a, tp) << 20) | (uint32_t)(j - t0);
        k2 = med3_u32(k1, k2, key);
        k1 = min(k1, key);
    }
    if (q < nq) part[(size_t)blockIdx.y * nq + q] = make_uint2(k1, k2);
}

__global__ __launch_bounds__(256) void k_bf_merge(const uint2 *__restrict__ part, int nq, int nsplit, int tchunk,
                                                  int32_t *__restrict__ best_idx, int32_t *__restrict__ best_d,
                                                  int32_t *__restrict__ second_d)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    int b1 = 256, b2 = 256, bi = -1;
    for (int s = 0; s < nsplit; s++) {
        const uint2 p = part[(size_t)s * nq + q];
        if (p.x != 0xFFFFFFFFu) {
            const int d = (int)(p.x >> 20);
            if (d < b1) { b2 = b1; b1 = d; bi = s * tchunk + (int)(p.x & 0xFFFFFu); }
            else if (d < b2) b2 = d;
        }
        if (p.y != 0xFFFFFFFFu) {
            const int d = (int)(p.y >> 20);
            if (d < b2) b2 = d;
        }
    }
    best_idx[q] = bi;
    best_d[q] = b1;
    second_d[q] = b2;
}

int launch_bruteforce(hipStream_t st, const uint8_t *d_Q, int nq, const uint8_t *d_T, int nt, int nsplit, int tchunk,
                      uint2 *part, int32_t *bi, int32_t *b1, int32_t *b2)
{
    const int qb = (nq + 255) / 256;
    if (nt > 0)
        hipLaunchKernelGGL(k_bf_partial, dim3(qb, nsplit), dim3(256), 0, st, (const uint32_t *)d_Q, nq, (const uint32_t *)d_T, nt,
                           tchunk, part);
    hipLaunchKernelGGL(k_bf_merge, dim3(qb), dim3(256), 0, st, part, nq, nt > 0 ? nsplit : 0, tchunk, bi, b1, b2);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

// ---------------------------------------------------------------------------
// Windowed matching.  One workgroup of 1024 threads per (query set, train
// frame) pair:
//   1. build the 64x48 keypoint grid of the train frame
//      (Frame::AssignFeaturesToGrid / PosInGrid, src/Frame.cc:270-293, 485-500);
//   2. iterate to the fixed point of "query i takes its best candidate among
//      those not claimed by an accepted query j < i".  The reference resolves
//      queries sequentially; its result is the unique fixed point of that map,
//      because query i depends on queries j < i only (induction on i);
//   3. rotation-histogram consistency (ComputeThreeMaxima, src/ORBmatcher.cc:1661-1704).
// Candidate order inside a window is the reference's (grid column outer, row
// inner, insertion order in the cell; src/Frame.cc:436-471): cells are stored
// column-major so one window column is a contiguous run of grid entries.
// ---------------------------------------------------------------------------
constexpr int kCells = ORBGPU_GRID_COLS * ORBGPU_GRID_ROWS;

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
constexpr unsigned long long kNoKey = ~0ull;

__global__ __launch_bounds__(1024) void k_match_window(MatchArgs A)
{
    __shared__ int cell_start[kCells + 1];
    __shared__ int scan_lds[33];
    __shared__ int hist[ORBGPU_HISTO_LENGTH];
    __shared__ int sh_changed, sh_keep[3], sh_conflict;
    __shared__ int cursor[kCells];

    const int pair = blockIdx.x, tid = threadIdx.x, nt_ = blockDim.x;
    const int lane = lane_id(), wid = tid >> 6, nw = nt_ >> 6;
    const int nq = A.nq[pair], n2 = A.n2[pair];
    const orbgpu_wquery *Q = A.q + (size_t)pair * A.q_cap;
    const uint32_t *QD = (const uint32_t *)(A.qdesc + (size_t)pair * A.q_cap * 32);
    const orbgpu_keypoint *K2 = A.kps2 + (size_t)pair * A.t_cap;
    const uint32_t *TD = (const uint32_t *)(A.desc2 + (size_t)pair * A.t_cap * 32);
    const float *UR = A.uright2 ? A.uright2 + (size_t)pair * A.t_cap : nullptr;
    const uint8_t *OCC = A.occupied ? A.occupied + (size_t)pair * A.t_cap : nullptr;
    int32_t *assign = A.assign + (size_t)pair * A.t_cap;
    int32_t *qmatch = A.qmatch ? A.qmatch + (size_t)pair * A.q_cap : nullptr;
    int32_t *scr = A.scratch + (size_t)pair * (5 * (size_t)A.q_cap + 2 * (size_t)A.t_cap);
    int32_t *sel[2] = {scr, scr + A.q_cap};
    int32_t *sd[2] = {scr + 2 * A.q_cap, scr + 3 * A.q_cap};
    int32_t *nxt = scr + 4 * A.q_cap;
    int32_t *head = scr + 5 * A.q_cap;
    int32_t *items = head + A.t_cap;
    const float invW = __fdiv_rn((float)ORBGPU_GRID_COLS, __fsub_rn(A.maxX, A.minX));
    const float invH = __fdiv_rn((float)ORBGPU_GRID_ROWS, __fsub_rn(A.maxY, A.minY));
    const int mode = A.mode;

    // ---- 1. grid -----------------------------------------------------------
    for (int i = tid; i <= kCells; i += nt_) cell_start[i] = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += nt_) {
        const int px = (int)roundf(__fmul_rn(__fsub_rn(K2[i].x, A.minX), invW));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(K2[i].y, A.minY), invH));
        if (px >= 0 && px < ORBGPU_GRID_COLS && py >= 0 && py < ORBGPU_GRID_ROWS) atomicAdd(&cell_start[px * ORBGPU_GRID_ROWS + py], 1);
    }
    __syncthreads();
    {
        int carry = 0;
        for (int b = 0; b < kCells; b += nt_) {
            const int cidx = b + tid;
            const int v = cidx < kCells ? cell_start[cidx] : 0;
            int total;
            const int ex = block_scan_excl(v, &total, scan_lds);
            if (cidx < kCells) cell_start[cidx] = carry + ex;
            carry += total;
        }
        if (tid == 0) cell_start[kCells] = carry;
    }
    __syncthreads();
    // fill: fetch-and-add gives an arbitrary in-cell order; head[] doubles as the per-cell cursor
    for (int i = tid; i < kCells; i += nt_) cursor[i] = 0;
    __syncthreads();
    for (int i = tid; i < n2; i += nt_) {
        const int px = (int)roundf(__fmul_rn(__fsub_rn(K2[i].x, A.minX), invW));
        const int py = (int)roundf(__fmul_rn(__fsub_rn(K2[i].y, A.minY), invH));
        if (px >= 0 && px < ORBGPU_GRID_COLS && py >= 0 && py < ORBGPU_GRID_ROWS) {
            const int cidx = px * ORBGPU_GRID_ROWS + py;
            items[cell_start[cidx] + atomicAdd(&cursor[cidx], 1)] = i;
        }
    }
    __syncthreads();
    // restore insertion (ascending index) order inside every cell
    for (int cidx = tid; cidx < kCells; cidx += nt_) {
        const int s = cell_start[cidx], e = cell_start[cidx + 1];
        for (int i = s + 1; i < e; i++) {
            const int v = items[i];
            int j = i - 1;
            while (j >= s && items[j] > v) { items[j + 1] = items[j]; j--; }
            items[j + 1] = v;
        }
    }
    for (int i = tid; i < nq; i += nt_) { sel[0][i] = -1; sd[0][i] = 0; }
    __syncthreads();

    // ---- 2. fixed-point iteration ----------------------------------------------
    int cur = 0;
    for (int it = 0; it <= nq + 1; it++) {
        for (int i = tid; i < n2; i += nt_) head[i] = -1;
        if (tid == 0) { sh_changed = 0; sh_conflict = 0; }
        __syncthreads();
        for (int i = tid; i < nq; i += nt_) {
            const int s = sel[cur][i];
            if (s >= 0) {
                const int old = atomicExch(&head[s], i);
                nxt[i] = old;
                if (old >= 0) sh_conflict = 1;
            }
        }
        __syncthreads();
        // Modes without a second-best test: if the unconstrained bests (state after
        // the first pass) select no train keypoint twice, no query's best is
        // claimed by an earlier one, so that state already is the fixed point.
        if (it == 1 && !sh_conflict && (mode == 1 || mode == 2)) break;
        for (int i = wid; i < nq; i += nw) {
            const orbgpu_wquery q = Q[i];
            int newsel = -1, newd = 0;
            if (q.flags & 1) {
                const float r = q.r;
                int c0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.u, A.minX), r), invW)));
                int c1 = min(ORBGPU_GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.u, A.minX), r), invW)));
                int r0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(q.v, A.minY), r), invH)));
                int r1 = min(ORBGPU_GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(q.v, A.minY), r), invH)));
                if (c0 < ORBGPU_GRID_COLS && c1 >= 0 && r0 < ORBGPU_GRID_ROWS && r1 >= 0) {
                    const bool check = (q.minLevel > 0) || (q.maxLevel >= 0);
                    uint32_t qd[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) qd[k] = QD[(size_t)i * 8 + k];
                    Top2 t2 = {kNoKey, kNoKey};
                    int pos = 0;
                    for (int ix = c0; ix <= c1; ix++) {
                        const int s = cell_start[ix * ORBGPU_GRID_ROWS + r0], e = cell_start[ix * ORBGPU_GRID_ROWS + r1 + 1];
                        for (int k = s + lane; k < e; k += 64) {
                            const int t = items[k];
                            const orbgpu_keypoint kp = K2[t];
                            if (check) {
                                if (kp.octave < q.minLevel) continue;
                                if (q.maxLevel >= 0 && kp.octave > q.maxLevel) continue;
                            }
                            if (!(fabsf(__fsub_rn(kp.x, q.u)) < r && fabsf(__fsub_rn(kp.y, q.v)) < r)) continue;
                            if (mode != 3) {
                                if (OCC && OCC[t]) continue;
                                if (mode != 2 && UR && UR[t] > 0) {
                                    if (fabsf(__fsub_rn(q.ur, UR[t])) > r) continue;
                                }
                            }
                            const int dist = hamming8(qd, TD + (size_t)t * 8);
                            // claims of accepted earlier queries
                            bool skip = false;
                            for (int j = head[t]; j >= 0; j = nxt[j]) {
                                if (j >= i) continue;
                                if (mode == 3) { if (sd[cur][j] <= dist) skip = true; }
                                else if (mode == 2 || (Q[j].flags & 2)) skip = true;
                            }
                            if (skip) continue;
                            top2_push(t2, ((unsigned long long)dist << 40) | ((unsigned long long)(pos + k - s) << 20) | (unsigned)t);
                        }
                        pos += e - s;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const unsigned long long oa = shfl_xor_u64(t2.a, o), ob = shfl_xor_u64(t2.b, o);
                        const unsigned long long m1 = min(t2.a, oa);
                        const unsigned long long m2 = min(max(t2.a, oa), min(t2.b, ob));
                        t2.a = m1;
                        t2.b = m2;
                    }
                    if (t2.a != kNoKey) {
                        const int best = (int)(t2.a >> 40), bidx = (int)(t2.a & 0xFFFFFull);
                        bool ok;
                        if (mode == 3) {
                            const float best2 = t2.b != kNoKey ? (float)(int)(t2.b >> 40) : 2147483648.0f;
                            ok = best <= ORBGPU_TH_LOW && (float)best < __fmul_rn(best2, A.nnratio);
                        } else {
                            ok = best <= A.th_dist;
                            if (ok && mode == 0 && t2.b != kNoKey) {
                                const int best2 = (int)(t2.b >> 40), bidx2 = (int)(t2.b & 0xFFFFFull);
                                if (K2[bidx].octave == K2[bidx2].octave && (float)best > __fmul_rn(A.nnratio, (float)best2)) ok = false;
                            }
                        }
                        if (ok) { newsel = bidx; newd = best; }
                    }
                }
            }
            if (lane == 0) {
                if (newsel != sel[cur][i] || newd != sd[cur][i]) sh_changed = 1;
                sel[cur ^ 1][i] = newsel;
                sd[cur ^ 1][i] = newd;
            }
        }
        __syncthreads();
        cur ^= 1;
        const int ch = sh_changed;
        __syncthreads();
        if (!ch) break;
    }

    // ---- 3. outputs + rotation consistency ---------------------------------------
    for (int i = tid; i < n2; i += nt_) assign[i] = -1;
    for (int i = tid; i < ORBGPU_HISTO_LENGTH; i += nt_) hist[i] = 0;
    if (tid == 0) sh_changed = 0;  // reused as the match counter
    __syncthreads();
    const float factor = 1.0f / ORBGPU_HISTO_LENGTH;  // the reference's 1/30 (src/ORBmatcher.cc:434)
    const bool ori = A.check_orientation && mode != 0;
    for (int i = tid; i < nq; i += nt_) {
        const int s = sel[cur][i];
        if (s < 0) continue;
        atomicMax(&assign[s], i);  // a later selector overwrites (mode 3: steals) an earlier one
        if (mode != 3) atomicAdd(&sh_changed, 1);
        if (ori) {
            float rot = __fsub_rn(Q[i].angle, K2[s].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == ORBGPU_HISTO_LENGTH) bin = 0;
            nxt[i] = bin;  // the claim lists are no longer needed
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
        if (tid == 0) {
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
            const int s = sel[cur][i];
            if (s < 0) continue;
            const int bin = nxt[i];
            if (bin == k1 || bin == k2 || bin == k3) continue;
            if (mode == 3) {
                // only a live match is removed (:551-559); i is the only writer of assign[s] here
                if (assign[s] == i) { assign[s] = -1; atomicSub(&sh_changed, 1); }
            } else {
                // every entry of a losing bin clears its train keypoint and decrements (:1507-1517)
                assign[s] = -1;
                atomicSub(&sh_changed, 1);
            }
        }
        __syncthreads();
    }
    if (qmatch) {
        for (int i = tid; i < nq; i += nt_) {
            const int s = sel[cur][i];
            qmatch[i] = (s >= 0 && assign[s] == i) ? s : -1;
        }
    }
    if (tid == 0) A.nmatches[pair] = sh_changed;
}

__global__ __launch_bounds__(256) void k_shift_queries(const orbgpu_keypoint *__restrict__ kps, const int32_t *__restrict__ n,
                                                       int capacity, float du, float dv, float th, ScaleTab sc, float minX,
                                                       float minY, float maxX, float maxY, orbgpu_wquery *__restrict__ q)
{
    const int pair = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n[pair]) return;
    const orbgpu_keypoint kp = kps[(size_t)pair * capacity + i];
    orbgpu_wquery w;
    w.u = __fadd_rn(kp.x, du);
    w.v = __fadd_rn(kp.y, dv);
    w.r = __fmul_rn(th, sc.s[kp.octave]);
    w.minLevel = kp.octave - 1;
    w.maxLevel = kp.octave + 1;
    w.angle = kp.angle;
    w.ur = -1.0f;
    const bool inside = !(w.u < minX || w.u > maxX || w.v < minY || w.v > maxY);  // src/ORBmatcher.cc:1424-1427
    w.flags = inside ? 3 : 0;
    q[(size_t)pair * capacity + i] = w;
}

int launch_match_window(hipStream_t st, const MatchArgs &A, int npairs)
{
    hipLaunchKernelGGL(k_match_window, dim3(npairs), dim3(1024), 0, st, A);
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

// ---------------------------------------------------------------------------
// Frame::ComputeStereoMatches (src/Frame.cc:600-757): one wave per left
// keypoint.  Lanes scan the row-band candidates (Hamming), then evaluate the
// 11 SAD window positions on the un-blurred pyramid level of both images and
// fit the parabola.  The median-based outlier rejection (:760-773) is done by
// the caller on the returned SAD values.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stereo(StereoArgs A)
{
    const int lane = lane_id(), iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= A.nL) return;
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    const orbgpu_keypoint kpL = A.kpsL[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int row = (int)vL;
    const float maxD = __fdiv_rn(A.bf, A.b);
    const float minU = __fsub_rn(uL, maxD), maxU = uL;
    do {
        if (row < 0 || row >= A.nRows) break;
        const int s = A.row_start[row], e = A.row_start[row + 1];
        if (s == e || maxU < 0) break;
        uint32_t qd[8];
#pragma unroll
        for (int k = 0; k < 8; k++) qd[k] = reinterpret_cast<const uint32_t *>(A.descL)[(size_t)iL * 8 + k];
        uint32_t best = 0xFFFFFFFFu;
        for (int k = s + lane; k < e; k += 64) {
            const int iR = A.row_items[k];
            const orbgpu_keypoint kpR = A.kpsR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            if (!(kpR.x >= minU && kpR.x <= maxU)) continue;
            const int dist = hamming8(qd, reinterpret_cast<const uint32_t *>(A.descR) + (size_t)iR * 8);
            if (dist < ORBGPU_TH_HIGH) best = min(best, ((uint32_t)dist << 20) | (uint32_t)(k - s));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
        if (best == 0xFFFFFFFFu) break;
        const int bestDist = (int)(best >> 20);
        if (!(bestDist < (ORBGPU_TH_HIGH + ORBGPU_TH_LOW) / 2)) break;
        const int bestIdxR = A.row_items[s + (int)(best & 0xFFFFFu)];

        // sub-pixel refinement by correlation (:686-757)
        const float uR0 = A.kpsR[bestIdxR].x;
        const float sf = A.inv_scale[levelL];
        const float scaleduL = roundf(__fmul_rn(kpL.x, sf)), scaledvL = roundf(__fmul_rn(kpL.y, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int w = 5, L = 5;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= (float)A.w[levelL]) break;
        const uint8_t *PL = A.pyrL[levelL], *PR = A.pyrR[levelL];
        const int sL = A.pitchL[levelL], sR = A.pitchR[levelL];
        const int y0 = (int)(scaledvL - w), xL0 = (int)(scaleduL - w);
        const int cL = PL[(int64_t)(y0 + w) * sL + xL0 + w];
        // lane -> up to two pixels of the 11x11 patch
        const int p0 = lane, p1 = lane + 64;
        const int py0 = p0 / 11, px0 = p0 - py0 * 11, py1 = p1 / 11, px1 = p1 - py1 * 11;
        const int a0 = (int)PL[(int64_t)(y0 + py0) * sL + xL0 + px0] - cL;
        const int a1 = p1 < 121 ? (int)PL[(int64_t)(y0 + py1) * sL + xL0 + px1] - cL : 0;
        int bestSad = 0x7FFFFFFF, bestinc = 0;
        float vd[11];
#pragma unroll
        for (int inc = -L; inc <= L; inc++) {
            const int xR0 = (int)(scaleduR0 + (float)inc - w);
            const int cR = PR[(int64_t)(y0 + w) * sR + xR0 + w];
            int acc = abs(a0 - ((int)PR[(int64_t)(y0 + py0) * sR + xR0 + px0] - cR));
            if (p1 < 121) acc += abs(a1 - ((int)PR[(int64_t)(y0 + py1) * sR + xR0 + px1] - cR));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (acc < bestSad) { bestSad = acc; bestinc = inc; }
            vd[inc + L] = (float)acc;
        }
        if (bestinc == -L || bestinc == L) break;
        float d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
        for (int k = 1; k < 10; k++)
            if (k == bestinc + L) { d1 = vd[k - 1]; d2 = vd[k]; d3 = vd[k + 1]; }
        const float deltaR = __fdiv_rn(__fsub_rn(d1, d3), __fmul_rn(2.0f, __fsub_rn(__fadd_rn(d1, d3), __fmul_rn(2.0f, d2))));
        if (deltaR < -1 || deltaR > 1) break;
        float bestuR = __fmul_rn(A.scale[levelL], __fadd_rn(__fadd_rn(scaleduR0, (float)bestinc), deltaR));
        float disparity = __fsub_rn(uL, bestuR);
        if (disparity >= 0 && disparity < maxD) {
            if (disparity <= 0) {
                disparity = (float)0.01;
                bestuR = (float)((double)uL - 0.01);
            }
            out_d = __fdiv_rn(A.bf, disparity);
            out_u = bestuR;
            out_sad = bestSad;
        }
    } while (0);
    if (lane == 0) {
        A.uRight[iL] = out_u;
        A.depth[iL] = out_d;
        A.best_sad[iL] = out_sad;
    }
}

int launch_stereo(hipStream_t st, const StereoArgs &A)
{
    hipLaunchKernelGGL(k_stereo, dim3((A.nL + 3) / 4), dim3(256), 0, st, A);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

}  // namespace orbgpu
