// Bag-of-words transform of DBoW2 on the device: the per-feature descent of the vocabulary tree and the
// construction of the BowVector / FeatureVector of a descriptor set.
// Reference lines are relative to /root/reference/src/ORB_SLAM2/Thirdparty/DBoW2/DBoW2/.
#include "bow_internal.h"

namespace orbgpu {

// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (TemplatedVocabulary.h:1217-1259) with
// FORB::distance (FORB.cpp:81-101).  G lanes own one feature: lane c measures child c of the current node, the
// group keeps the smallest distance << 8 | c (the reference's strict `d < best_d` keeps the first minimum), all
// lanes step to that child.  Groups of one wave may leave the loop at different depths.
template <int G>
__global__ __launch_bounds__(256) void k_voc_descend(BowArgs A)
{
    const int set = blockIdx.y;
    const int n = min(A.counts[set], A.capacity);
    const int c = threadIdx.x % G;
    const int f = blockIdx.x * (256 / G) + threadIdx.x / G;
    if (f >= n) return;
    const size_t at = (size_t)set * A.capacity + f;
    const uint4 *qp = (const uint4 *)(A.desc + at * 32);
    const uint4 q0 = qp[0], q1 = qp[1];
    uint32_t node = 0, link = A.V.link[0];
    int level = 0, nid = A.nid_level <= 0 ? 0 : -1;  // :1226-1227
    while ((link & 63u) != 0) {
        ++level;
        const uint32_t nch = link & 63u, first = link >> 6;
        uint32_t key = 0xFFFFFFFFu;
        if ((uint32_t)c < nch) {
            const uint4 *dp = A.V.desc + (size_t)(first + c) * 2;
            const uint4 d0 = dp[0], d1 = dp[1];
            const int d = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
                          __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
            key = ((uint32_t)d << 8) | (uint32_t)c;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, o, G));
        node = first + (key & 255u);
        link = A.V.link[node];
        if (level == A.nid_level) nid = A.V.orig[node];  // :1251-1252
    }
    if (c == 0) {
        // a leaf above nid_level: the reference leaves *nid unset there; the leaf itself is reported
        A.node_ids[at] = nid < 0 ? A.V.orig[node] : nid;
        A.word_ids[at] = A.V.word[node];
        A.wts[at] = A.V.weight[node];
    }
}

// Ascending bitonic sort of P (a power of two) 64-bit keys in LDS by the whole workgroup.
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *keys, int P)
{
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += blockDim.x) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const uint64_t a = keys[i], b = keys[l];
                if ((a > b) == ((i & k) == 0)) {
                    keys[i] = b;
                    keys[l] = a;
                }
            }
            __syncthreads();
        }
}

constexpr uint64_t kNoKey = ~0ull;

// Runs of equal high words in the sorted keys: for run r (in order) id_out[r] = the high word and
// start_out[r] = its first position; pos_out[p] (optional) = low word of key p.  Returns the number of runs
// and, in *nkept, the number of keys that are not kNoKey.  Every thread must call it.
__device__ __forceinline__ int list_runs(const uint64_t *keys, int P, int32_t *id_out, int32_t *start_out, int32_t *pos_out,
                                         int *nkept, int *scan_lds)
{
    int runs = 0, kept = 0;
    for (int base = 0; base < P; base += blockDim.x) {
        const int p = base + threadIdx.x;
        const uint64_t k = p < P ? keys[p] : kNoKey;
        const bool valid = k != kNoKey;
        const bool head = valid && (p == 0 || (uint32_t)(keys[p - 1] >> 32) != (uint32_t)(k >> 32));
        int tot;
        const int r = runs + block_scan_excl(head ? 1 : 0, &tot, scan_lds);
        if (head) {
            id_out[r] = (int32_t)(k >> 32);
            start_out[r] = p;
        }
        if (valid && pos_out) pos_out[p] = (int32_t)(uint32_t)k;
        runs += tot;
        kept += __syncthreads_count(valid);
    }
    *nkept = kept;
    return runs;
}

// transform(features, BowVector, FeatureVector, levelsup) (TemplatedVocabulary.h:1127-1194) after the descent:
// one workgroup per descriptor set.  std::map order = ascending id, so both containers are runs of a sort by
// (id, feature index); the value of a word is accumulated feature by feature as addWeight does
// (BowVector.cpp:34-46), the norm is summed in map order by one lane (BowVector.cpp:62-84) so that every
// double rounds as in the reference.
__global__ __launch_bounds__(1024) void k_bow_build(BowArgs A, int P)
{
    extern __shared__ uint64_t keys[];
    __shared__ int scan_lds[33];
    __shared__ double sh_norm;
    const int set = blockIdx.x;
    const int n = min(A.counts[set], A.capacity);
    const size_t at = (size_t)set * A.capacity;
    const int32_t *word = A.word_ids + at, *node = A.node_ids + at;
    const double *wts = A.wts + at;
    int32_t *bow_ids = A.bow_ids + at, *fv_nodes = A.fv_nodes + at, *fv_feat = A.fv_feat + at;
    int32_t *fv_start = A.fv_start + (size_t)set * (A.capacity + 1);
    double *bow_vals = A.bow_vals + at;

    // ---- BowVector ----
    for (int p = threadIdx.x; p < P; p += blockDim.x)
        keys[p] = (p < n && wts[p] > 0) ? ((uint64_t)(uint32_t)word[p] << 32) | (uint32_t)p : kNoKey;  // :1157 w > 0
    __syncthreads();
    bitonic_sort_lds(keys, P);
    int nk;
    // fv_start serves as scratch for the run starts until the FeatureVector pass rewrites it
    const int nw = list_runs(keys, P, bow_ids, fv_start, nullptr, &nk, scan_lds);
    __threadfence_block();
    __syncthreads();
    for (int r = threadIdx.x; r < nw; r += blockDim.x) {
        const int p0 = fv_start[r], p1 = r + 1 < nw ? fv_start[r + 1] : nk;
        const double w = wts[(uint32_t)keys[p0]];
        double v = w;  // insert
        if (A.tf)
            for (int i = p0 + 1; i < p1; i++) v += w;  // vit->second += v, once per further feature of the word
        bow_vals[r] = v;
    }
    __threadfence_block();
    __syncthreads();
    double *vals = (double *)keys;  // the keys are dead from here to the next pass
    for (int r = threadIdx.x; r < nw; r += blockDim.x) {
        double v = bow_vals[r];
        if (A.tf && !A.must) v = __ddiv_rn(v, (double)nw);  // :1164-1170
        vals[r] = v;
    }
    __syncthreads();
    if (A.must) {
        if (threadIdx.x == 0) {
            double norm = 0.0;
            int r = 0;
            for (; r + 8 <= nw; r += 8) {
                double t[8];
#pragma unroll
                for (int u = 0; u < 8; u++) t[u] = A.l2 ? __dmul_rn(vals[r + u], vals[r + u]) : fabs(vals[r + u]);
#pragma unroll
                for (int u = 0; u < 8; u++) norm = __dadd_rn(norm, t[u]);
            }
            for (; r < nw; r++) norm = __dadd_rn(norm, A.l2 ? __dmul_rn(vals[r], vals[r]) : fabs(vals[r]));
            sh_norm = A.l2 ? __dsqrt_rn(norm) : norm;
        }
        __syncthreads();
        const double norm = sh_norm;
        if (norm > 0.0)
            for (int r = threadIdx.x; r < nw; r += blockDim.x) bow_vals[r] = __ddiv_rn(vals[r], norm);
    } else {
        for (int r = threadIdx.x; r < nw; r += blockDim.x) bow_vals[r] = vals[r];
    }
    if (threadIdx.x == 0) A.n_bow[set] = nw;
    __syncthreads();

    // ---- FeatureVector (FeatureVector.cpp:31-45) ----
    for (int p = threadIdx.x; p < P; p += blockDim.x)
        keys[p] = (p < n && wts[p] > 0) ? ((uint64_t)(uint32_t)node[p] << 32) | (uint32_t)p : kNoKey;
    __syncthreads();
    bitonic_sort_lds(keys, P);
    const int nn = list_runs(keys, P, fv_nodes, fv_start, fv_feat, &nk, scan_lds);
    if (threadIdx.x == 0) {
        fv_start[nn] = nk;
        A.n_fv[set] = nn;
    }
}

int launch_bow_transform(hipStream_t st, const BowArgs &A)
{
    if (A.nsets < 1) return ORBGPU_OK;
    if (A.capacity < 1 || A.capacity > kBowMaxFeatures || A.max_children > kVocMaxChildren) {
        set_error("vocabulary transform: capacity %d / %d children per node outside the supported range (%d features, %d children)",
                  A.capacity, A.max_children, kBowMaxFeatures, kVocMaxChildren);
        return ORBGPU_E_UNSUPPORTED;
    }
    if (A.max_children <= 16)
        hipLaunchKernelGGL(k_voc_descend<16>, dim3((A.capacity + 15) / 16, A.nsets), dim3(256), 0, st, A);
    else
        hipLaunchKernelGGL(k_voc_descend<32>, dim3((A.capacity + 7) / 8, A.nsets), dim3(256), 0, st, A);
    ORBGPU_HIP_CHECK(hipGetLastError());
    int P = 2;
    while (P < A.capacity) P <<= 1;
    const size_t lds = (size_t)P * sizeof(uint64_t);
    if (lds > 48 * 1024)  // per device and cheap, so simply repeated
        ORBGPU_HIP_CHECK(hipFuncSetAttribute((const void *)k_bow_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int threads = P / 2 >= 1024 ? 1024 : (P / 2 < 64 ? 64 : P / 2);
    hipLaunchKernelGGL(k_bow_build, dim3(A.nsets), dim3(threads), lds, st, A, P);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

}  // namespace orbgpu
