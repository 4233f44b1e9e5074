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
// fit the parabola.  The median-based outlier rejection (:760-773) is done by
// the caller on the returned SAD values.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stereo(StereoArgs A)
{
    const int lane = lane_id(), iL = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iL >= (A.d_nL ? min(*A.d_nL, A.nL) : A.nL)) return;
    float out_u = -1.0f, out_d = -1.0f;
    int out_sad = -1;
    const orbgpu_keypoint kpL = A.kpsL[iL];
    const int levelL = min(max(kpL.octave, 0), A.nlevels - 1);  // the host entry point rejects octaves outside the pyramid
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
