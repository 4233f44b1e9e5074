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

// popcount(x) + acc in one instruction (v_bcnt_u32_b32 accumulates)
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t hamming8_chain(const uint32_t (&a)[8], const uint32_t *__restrict__ b)
{
    uint32_t d = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) d = bcnt_acc(a[k] ^ b[k], d);
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
            const uint32_t key = (hamming8_chain(a, tp) << 20) | (uint32_t)(j + u - t0);
            k2 = med3_u32(k1, k2, key);  // second smallest of {k1<=k2, key}
            k1 = min(k1, key);
        }
    }
    for (; j < t1; j++) {
        const uint32_t *tp = T + (size_t)j * 8;
        const uint32_t key = (hamming8_chain(a, tp) << 20) | (uint32_t)(j - t0);
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
// Frame::ComputeStereoMatches (src/Frame.cc:600-757): one wave per left
// keypoint.  Lanes scan the row-band candidates (Hamming), then evaluate the
// 11 SAD window positions on the un-blurred pyramid level of both images and
// fit the parabola.  The row table (:610-627) and the median-based outlier rejection (:760-773) are
// the kernels after this one.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stereo(const StereoArgs A)
{
    const int lane = lane_id(), iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int pair = blockIdx.y;
    if (iL >= (A.d_nL ? min(A.d_nL[pair], A.nL) : A.nL)) return;
    // this pair's slices of the batch (all strides are zero for a single pair); the argument block itself stays
    // untouched in constant memory, where its per-level tables are indexed at run time
    const orbgpu_keypoint *__restrict__ kpsL = A.kpsL + (size_t)pair * A.strideL, *__restrict__ kpsR = A.kpsR + (size_t)pair * A.strideR;
    const uint8_t *__restrict__ descL = A.descL + (size_t)pair * A.strideL * 32, *__restrict__ descR = A.descR + (size_t)pair * A.strideR * 32;
    const int32_t *__restrict__ row_start = A.row_start + (size_t)pair * A.row_stride;
    const int32_t *__restrict__ row_items = A.row_items + (size_t)pair * A.items_stride;
    const size_t outL = (size_t)pair * A.strideL;
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    const orbgpu_keypoint kpL = kpsL[iL];
    const int levelL = min(max(kpL.octave, 0), A.nlevels - 1);  // the host entry point rejects octaves outside the pyramid
    const float vL = kpL.y, uL = kpL.x;
    const int row = (int)vL;
    const float maxD = __fdiv_rn(A.bf, A.b);
    const float minU = __fsub_rn(uL, maxD), maxU = uL;
    do {
        if (row < 0 || row >= A.nRows) break;
        const int s = row_start[row], e = row_start[row + 1];
        if (s == e || maxU < 0) break;
        uint32_t qd[8];
#pragma unroll
        for (int k = 0; k < 8; k++) qd[k] = reinterpret_cast<const uint32_t *>(descL)[(size_t)iL * 8 + k];
        uint32_t best = 0xFFFFFFFFu;
        for (int k = s + lane; k < e; k += 64) {
            const int iR = row_items[k];
            const orbgpu_keypoint kpR = kpsR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            if (!(kpR.x >= minU && kpR.x <= maxU)) continue;
            const int dist = hamming8(qd, reinterpret_cast<const uint32_t *>(descR) + (size_t)iR * 8);
            if (dist < ORBGPU_TH_HIGH) best = min(best, ((uint32_t)dist << 20) | (uint32_t)(k - s));
        }
        best = wave_min_u32(best);
        if (best == 0xFFFFFFFFu) break;
        const int bestDist = (int)(best >> 20);
        if (!(bestDist < (ORBGPU_TH_HIGH + ORBGPU_TH_LOW) / 2)) break;
        const int bestIdxR = row_items[s + (int)(best & 0xFFFFFu)];

        // sub-pixel refinement by correlation (:686-757)
        const float uR0 = kpsR[bestIdxR].x;
        const float sf = A.inv_scale[levelL];
        const float scaleduL = roundf(__fmul_rn(kpL.x, sf)), scaledvL = roundf(__fmul_rn(kpL.y, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int w = 5, L = 5;
        const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
        if (iniu < 0 || endu >= (float)A.w[levelL]) break;
        const uint8_t *PL = A.pyrL[levelL] + pair * A.pyr_strideL, *PR = A.pyrR[levelL] + pair * A.pyr_strideR;
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
            acc = wave_sum(acc);
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
        A.uRight[outL + iL] = out_u;
        A.depth[outL + iL] = out_d;
        A.best_sad[outL + iL] = out_sad;
    }
}

int launch_stereo_batch(hipStream_t st, const StereoArgs &A, int npairs);
int launch_stereo(hipStream_t st, const StereoArgs &A)
{
    return launch_stereo_batch(st, A, 1);
}

// `npairs` stereo pairs in one launch (grid.y = pair; the strides of StereoArgs describe the batch).
int launch_stereo_batch(hipStream_t st, const StereoArgs &A, int npairs)
{
    hipLaunchKernelGGL(k_stereo, dim3((A.nL + 3) / 4, npairs), dim3(256), 0, st, A);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

// Row table of ComputeStereoMatches (src/Frame.cc:610-627): right keypoint iR is listed in rows floor(y - r) ..
// ceil(y + r), r = 2 * scale[octave]; every row lists its keypoints in ascending iR.  Two launches: one workgroup
// writes every keypoint's row span, counts the rows' entries with LDS atomics and scans them into the row starts;
// then a wave per row walks the spans 64 at a time and appends the keypoints that cover its row at ballot ranks,
// which keeps them in index order.
__global__ __launch_bounds__(1024) void k_stereo_rows_count(const orbgpu_keypoint *__restrict__ kpsR, const int32_t *__restrict__ d_nR,
                                                            int capR, ScaleTab scale, int nlevels, int nRows, int32_t *__restrict__ spans,
                                                            int32_t *__restrict__ row_start)
{
    extern __shared__ int rows_lds[];  // nRows + 1 counts
    __shared__ int scan_lds[33];
    const int tid = threadIdx.x;
    const int pair = blockIdx.x;  // a workgroup per stereo pair: arrays advance by capR / nRows + 1 entries per pair
    kpsR += (size_t)pair * capR; spans += (size_t)pair * capR; row_start += (size_t)pair * (nRows + 1);
    const int nR = min(max(d_nR[pair], 0), capR);
    for (int y = tid; y <= nRows; y += 1024) rows_lds[y] = 0;
    __syncthreads();
    for (int iR = tid; iR < nR; iR += 1024) {
        const orbgpu_keypoint kp = kpsR[iR];
        const float r = __fmul_rn(2.0f, scale.s[min(max(kp.octave, 0), nlevels - 1)]);
        const int maxr = (int)ceilf(__fadd_rn(kp.y, r)), minr = (int)floorf(__fsub_rn(kp.y, r));
        const int lo = max(minr, 0), hi = min(maxr, nRows - 1);  // an empty span (above row 0 or past the last row) has lo > hi
        spans[iR] = (lo & 0xFFFF) | (hi << 16);
        for (int y = lo; y <= hi; y++) atomicAdd(&rows_lds[y], 1);
    }
    __syncthreads();
    int carry = 0;
    for (int b = 0; b < nRows; b += 1024) {
        const int y = b + tid;
        const int v = y < nRows ? rows_lds[y] : 0;
        int total;
        const int ex = block_scan_excl(v, &total, scan_lds);
        if (y < nRows) row_start[y] = carry + ex;
        carry += total;
    }
    if (tid == 0) row_start[nRows] = carry;
}

__global__ __launch_bounds__(256) void k_stereo_rows_fill(const int32_t *__restrict__ spans, const int32_t *__restrict__ d_nR, int capR,
                                                          int nRows, const int32_t *__restrict__ row_start, int32_t *__restrict__ row_items,
                                                          int items_stride)
{
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6), lane = lane_id();
    if (y >= nRows) return;
    const int pair = blockIdx.y;
    spans += (size_t)pair * capR; row_start += (size_t)pair * (nRows + 1); row_items += (size_t)pair * items_stride;
    const int nR = min(max(d_nR[pair], 0), capR);
    int cur = row_start[y];
    for (int base = 0; base < nR; base += 64) {
        const int iR = base + lane;
        bool covers = false;
        if (iR < nR) {
            const int s = spans[iR];
            const int lo = s & 0xFFFF, hi = s >> 16;  // hi is negative for keypoints above the image
            covers = lo <= y && y <= hi;
        }
        const uint64_t m = __ballot(covers);
        if (covers) row_items[cur + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = iR;
        cur += __popcll(m);
    }
}

// Outlier rejection of ComputeStereoMatches (src/Frame.cc:760-773): the median of the matched SADs is element
// size / 2 of their ascending order; matches with SAD >= 1.5 * 1.4 * median lose their measurement.  The SAD of an
// 11 x 11 window of centre-subtracted bytes is at most 121 * 510 = 61 710 < 2^16: the median comes out of a
// histogram of the high eight bits and one of the low eight bits inside the bucket that holds it.  A workgroup per
// stereo pair (the arrays advance by capL entries per pair).
__global__ __launch_bounds__(1024) void k_stereo_outliers(const int32_t *__restrict__ best_sad, const int32_t *__restrict__ d_nL, int capL,
                                                          float *__restrict__ uRight, float *__restrict__ depth, int32_t *__restrict__ nvalid_out,
                                                          int nL_stride, int nvalid_stride)
{
    __shared__ int hist[256];
    __shared__ int removed;
    const int tid = threadIdx.x;
    const int pair = blockIdx.x;
    best_sad += (size_t)pair * capL; uRight += (size_t)pair * capL; depth += (size_t)pair * capL; nvalid_out += pair * nvalid_stride;
    const int n = min(max(d_nL[pair * nL_stride], 0), capL);
    if (tid < 256) hist[tid] = 0;
    if (tid == 0) removed = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int s = best_sad[i];
        if (s >= 0) atomicAdd(&hist[min(s >> 8, 255)], 1);
    }
    __syncthreads();
    // every thread walks the histogram itself (broadcast reads): the number of matches, the bucket of the median and
    // the matches below that bucket
    int total = 0;
    for (int k = 0; k < 256; k++) total += hist[k];
    if (total == 0) {
        if (tid == 0) *nvalid_out = 0;
        return;
    }
    const int kmed = total / 2;
    int b = 0, acc = 0;
    for (; b < 255; b++) {
        if (kmed < acc + hist[b]) break;
        acc += hist[b];
    }
    __syncthreads();
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int s = best_sad[i];
        if (s >= 0 && min(s >> 8, 255) == b) atomicAdd(&hist[s & 255], 1);
    }
    __syncthreads();
    int lo = 0;
    for (; lo < 255; lo++) {
        if (kmed < acc + hist[lo]) break;
        acc += hist[lo];
    }
    const int bucket = (b << 8) | lo;
    const float median = (float)bucket;
    const float thDist = __fmul_rn(1.5f * 1.4f, median);
    int mine = 0;
    for (int i = tid; i < n; i += 1024) {
        const int s = best_sad[i];
        if (s >= 0 && !((float)s < thDist)) {
            uRight[i] = -1.0f;
            depth[i] = -1.0f;
            mine++;
        }
    }
    if (mine) atomicAdd(&removed, mine);
    __syncthreads();
    if (tid == 0) *nvalid_out = total - removed;
}

// npairs row tables in two launches: pair p's right keypoints at kpsR + p * capR with count d_nR[p], its spans at
// spans + p * capR, row starts at row_start + p * (nRows + 1), row lists at row_items + p * items_stride.
int launch_stereo_rows_batch(hipStream_t st, int npairs, const orbgpu_keypoint *kpsR, const int32_t *d_nR, int capR, const ScaleTab &scale,
                             int nlevels, int nRows, int32_t *spans, int32_t *row_start, int32_t *row_items, int items_stride)
{
    hipLaunchKernelGGL(k_stereo_rows_count, dim3(npairs), dim3(1024), ((size_t)nRows + 1) * 4, st, kpsR, d_nR, capR, scale, nlevels, nRows,
                       spans, row_start);
    hipLaunchKernelGGL(k_stereo_rows_fill, dim3((nRows + 3) / 4, npairs), dim3(256), 0, st, spans, d_nR, capR, nRows, row_start, row_items,
                       items_stride);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

int launch_stereo_rows(hipStream_t st, const orbgpu_keypoint *kpsR, const int32_t *d_nR, int capR, const ScaleTab &scale, int nlevels, int nRows,
                       int32_t *spans, int32_t *row_start, int32_t *row_items)
{
    return launch_stereo_rows_batch(st, 1, kpsR, d_nR, capR, scale, nlevels, nRows, spans, row_start, row_items, 0);
}

// npairs outlier cuts: pair p's arrays at + p * capL, its count at d_nL[p * nL_stride], its result at nvalid[p * nvalid_stride].
int launch_stereo_outliers_batch(hipStream_t st, int npairs, const int32_t *best_sad, const int32_t *d_nL, int nL_stride, int capL,
                                 float *uRight, float *depth, int32_t *nvalid, int nvalid_stride)
{
    hipLaunchKernelGGL(k_stereo_outliers, dim3(npairs), dim3(1024), 0, st, best_sad, d_nL, capL, uRight, depth, nvalid, nL_stride,
                       nvalid_stride);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

int launch_stereo_outliers(hipStream_t st, const int32_t *best_sad, const int32_t *d_nL, int capL, float *uRight, float *depth,
                           int32_t *nvalid)
{
    return launch_stereo_outliers_batch(st, 1, best_sad, d_nL, 0, capL, uRight, depth, nvalid, 0);
}

// ---------------------------------------------------------------------------
// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:297-334), batched: map point p
// observes descriptors desc[start[p] .. start[p+1]); the representative is the one whose
// median distance to all of them (itself included) is smallest, first one among equals.
// One wave per map point, a lane per row of the distance matrix; the median of a row is its
// k-th smallest value, k = (int)(0.5*(N-1)), found by a 9-step binary search on the value
// (the largest v with #{d < v} <= k) with the distances recomputed in every step.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_distinctive(const uint32_t *__restrict__ desc, const int32_t *__restrict__ start, int npoints,
                                                     int32_t *__restrict__ best_idx, int32_t *__restrict__ best_median)
{
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= npoints) return;
    const int b = start[p], N = start[p + 1] - b;
    if (N <= 0) {
        if (lane == 0) { best_idx[p] = -1; if (best_median) best_median[p] = -1; }
        return;
    }
    const int k = (N - 1) >> 1;
    const uint32_t *D = desc + (size_t)b * 8;
    unsigned best = 0xFFFFFFFFu;  // median << 20 | row
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        const bool live = i < N;
        uint32_t mine[8];
#pragma unroll
        for (int w = 0; w < 8; w++) mine[w] = D[(size_t)(live ? i : 0) * 8 + w];
        int lo = 0;
        for (int bit = 256; bit > 0; bit >>= 1) {
            const int cand = lo | bit;
            int below = 0;
            for (int j = 0; j < N; j++) {
                int d = 0;
#pragma unroll
                for (int w = 0; w < 8; w++) d += __popc(mine[w] ^ D[(size_t)j * 8 + w]);  // uniform address: scalar loads
                below += d < cand ? 1 : 0;
            }
            if (below <= k) lo = cand;
        }
        unsigned key = live ? ((unsigned)lo << 20) | (unsigned)i : 0xFFFFFFFFu;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) key = min(key, (unsigned)__shfl_xor((int)key, o));
        best = min(best, key);
    }
    if (lane == 0) {
        best_idx[p] = (int)(best & 0xFFFFFu);
        if (best_median) best_median[p] = (int)(best >> 20);
    }
}

int launch_distinctive(hipStream_t st, const uint8_t *d_desc, const int32_t *d_start, int npoints, int32_t *d_best_idx,
                       int32_t *d_best_median)
{
    hipLaunchKernelGGL(k_distinctive, dim3((npoints + 3) / 4), dim3(256), 0, st, (const uint32_t *)d_desc, d_start, npoints, d_best_idx,
                       d_best_median);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

// ---------------------------------------------------------------------------
// Frame::ComputeStereoFromRGBD (src/Frame.cc:777-798): depth image lookup at the (distorted)
// keypoint, truncated to integer pixel coordinates as cv::Mat::at<float>(float, float) does.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stereo_from_rgbd(const orbgpu_keypoint *__restrict__ kps, const orbgpu_keypoint *__restrict__ kps_un,
                                                          int n, const float *__restrict__ depth, int w, int h, int stride_floats,
                                                          float bf, float *__restrict__ uright, float *__restrict__ out_depth)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int v = (int)kps[i].y, u = (int)kps[i].x;
    float d = 0.0f;
    if (u >= 0 && u < w && v >= 0 && v < h) d = depth[(size_t)v * stride_floats + u];
    if (d > 0) {
        out_depth[i] = d;
        uright[i] = __fsub_rn(kps_un[i].x, __fdiv_rn(bf, d));
    } else {
        out_depth[i] = -1.0f;
        uright[i] = -1.0f;
    }
}

int launch_stereo_from_rgbd(hipStream_t st, const orbgpu_keypoint *kps, const orbgpu_keypoint *kps_un, int n, const float *depth, int w,
                            int h, int stride_floats, float bf, float *uright, float *out_depth)
{
    hipLaunchKernelGGL(k_stereo_from_rgbd, dim3((n + 255) / 256), dim3(256), 0, st, kps, kps_un, n, depth, w, h, stride_floats, bf, uright,
                       out_depth);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

// ---------------------------------------------------------------------------
// Frame::UndistortKeyPoints (src/Frame.cc:517-558); the arithmetic is undistort_point (match_internal.h).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_undistort(const orbgpu_keypoint *__restrict__ kps, int n, UndistortArgs U,
                                                   orbgpu_keypoint *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    orbgpu_keypoint kp = kps[i];
    undistort_point(U, kp.x, kp.y, &kp.x, &kp.y);
    out[i] = kp;
}

int launch_undistort(hipStream_t st, const orbgpu_keypoint *kps, int n, const float *K4, const float *D5, orbgpu_keypoint *out)
{
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, st, kps, n, undistort_args(K4, D5), out);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

}  // namespace orbgpu
