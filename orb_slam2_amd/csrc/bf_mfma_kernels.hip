// Brute-force Hamming scan on the matrix cores of gfx950 (v_mfma_i32_32x32x32_i8).
//
// The scan of src/ORBmatcher.cc:205-226 (under /root/reference/src/ORB_SLAM2/) asks, for every query, for the
// train descriptor with the smallest Hamming distance (lowest index among equals), that distance and the second
// smallest.  With the descriptors' bits spread to one byte per bit, b in {0, -128},
//     sum_k a_k * b_k = 2^14 * popcount(a & b),      d(a, b) = |a| + |b| - 2 * popcount(a & b),
// so for a fixed query the train descriptor t with the smallest d is the one with the largest
//     key(t) = 2^13 * (2 * popcount(q & t) - |t|) + (8191 - index of t within its chunk of <= 8192).
// The second summand breaks ties towards the lower index, the first is the matrix product plus a per-train
// constant that enters as the accumulator's initial value, so the matrix instruction delivers finished keys:
// the vector unit only keeps the two largest per query (one v_max_i32 + one v_med3_i32 per pair).  Everything is
// integer arithmetic, so the result is the scan's, bit for bit.
//
// Layout.  k_bf_expand writes the byte-per-bit images: queries as [query][256 B]; train descriptors in tiles of
// 32, each tile already in the order the main kernel keeps it in LDS (16-byte pieces, piece q of row r at position
// r * 16 + (q ^ (r & 15)): a half-wave's 128-bit LDS reads of one piece index then fall into 16 different bank
// groups) plus the tile's 32 accumulator constants.  k_bf_mfma: a workgroup of four waves owns 4 * NQ * 32 queries
// (a wave holds NQ tiles of 32 queries as B operands in registers for the whole launch) and walks one chunk of the
// train set tile by tile through a double-buffered LDS image; the train tile is the A operand, so a lane's 16
// results of a tile belong to one query (its column) and 16 train rows, and the running maxima stay in two registers
// per query tile.  k_bf_mfma_merge folds the chunks' pairs of keys into (index, distance, second distance).
#include "orbgpu_internal.h"
#include "match_internal.h"

namespace orbgpu {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kBfTile = 32;                 // train descriptors per tile (rows of the matrix instruction)
constexpr int kBfTileBytes = kBfTile * 256; // byte-per-bit image of a tile
constexpr int kBfChunkMax = 8192;           // train descriptors per chunk: the key keeps 13 bits for the index
constexpr int kBfInvalid = -(1 << 30);      // accumulator constant of the rows that pad the last tile

__device__ __forceinline__ int med3_i32(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }

// 16 bits -> 16 bytes of 0x00 / 0x80.  A nibble x times 0x204081 puts bit i of x at bit 8 i (the partial products
// do not overlap), the mask keeps those, the shift makes them the byte's sign bit.
__device__ __forceinline__ uint32_t spread4(uint32_t x) { return ((__umul24(x, 0x204081u)) & 0x01010101u) << 7; }

// One thread per 16-byte piece: descriptor d = id / 16, piece q = id % 16 (bits 16 q .. 16 q + 15).
// swizzle == 0: the piece goes to out + d * 256 + q * 16 (queries).  swizzle != 0: train layout, see above; thread
// q == 0 of a descriptor also writes its accumulator constant, chunk = the number of train descriptors per chunk.
__global__ __launch_bounds__(256) void k_bf_expand(const uint8_t *__restrict__ desc, int n, int n_pad, int swizzle, int chunk,
                                                   uint8_t *__restrict__ out, int32_t *__restrict__ cinit)
{
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int d = id >> 4, q = id & 15;
    if (d >= n_pad) return;
    uint32_t bits = 0;
    if (d < n) bits = reinterpret_cast<const uint16_t *>(desc)[(size_t)d * 16 + q];
    uint4 v;
    v.x = spread4(bits & 15u);
    v.y = spread4((bits >> 4) & 15u);
    v.z = spread4((bits >> 8) & 15u);
    v.w = spread4(bits >> 12);
    if (!swizzle) {
        *reinterpret_cast<uint4 *>(out + (size_t)d * 256 + q * 16) = v;
        return;
    }
    const int tile = d >> 5, r = d & 31;
    *reinterpret_cast<uint4 *>(out + (size_t)tile * kBfTileBytes + (size_t)((r << 4) + (q ^ (r & 15))) * 16) = v;
    if (q == 0) {
        int c = kBfInvalid;
        if (d < n) {
            int pc = 0;
            const uint32_t *w = reinterpret_cast<const uint32_t *>(desc) + (size_t)d * 8;
#pragma unroll
            for (int k = 0; k < 8; k++) pc += __popc(w[k]);
            c = -(pc << 13) + (kBfChunkMax - 1 - (d % chunk));
        }
        cinit[d] = c;
    }
}

#ifndef BF_TB
#define BF_TB 1
#endif
constexpr int kBfStage = BF_TB;  // tiles per LDS stage (one workgroup barrier per stage); the train set is padded to whole stages

template <int NQ, int TB>
__global__ __launch_bounds__(256) void k_bf_mfma(const uint8_t *__restrict__ Qx, int nq_pad, const uint8_t *__restrict__ Tx,
                                                 const int32_t *__restrict__ cinit, int tiles_per_chunk, int ntiles,
                                                 int2 *__restrict__ part)
{
    constexpr int kStageBytes = TB * kBfTileBytes;
    __shared__ __attribute__((aligned(16))) uint8_t lds[2][kStageBytes + TB * 128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int q0 = (blockIdx.x * 4 + wave) * (NQ * 32);
    const int tile0 = blockIdx.y * tiles_per_chunk, tile1 = min(ntiles, tile0 + tiles_per_chunk);  // multiples of TB

    // B operands: lane (r, h) holds bytes h * 128 + s * 16 .. + 15 of query r for step s
    v4i qf[NQ][8];
#pragma unroll
    for (int n = 0; n < NQ; n++) {
        const uint8_t *src = Qx + (size_t)(q0 + n * 32 + r) * 256 + h * 128;
#pragma unroll
        for (int s = 0; s < 8; s++) qf[n][s] = *reinterpret_cast<const v4i *>(src + s * 16);
    }
    int m1[NQ], m2[NQ];
#pragma unroll
    for (int n = 0; n < NQ; n++) m1[n] = m2[n] = kBfInvalid * 2 + 1;  // below every key

    v4i g[2 * TB], gc = {0, 0, 0, 0};
    auto fetch = [&](int t) {
        const uint8_t *src = Tx + (size_t)t * kBfTileBytes + tid * 16;
#pragma unroll
        for (int k = 0; k < 2 * TB; k++) g[k] = *reinterpret_cast<const v4i *>(src + k * 4096);
        if (tid < 8 * TB) gc = *reinterpret_cast<const v4i *>(cinit + (size_t)t * kBfTile + tid * 4);
    };
    auto stash = [&](int b) {
#pragma unroll
        for (int k = 0; k < 2 * TB; k++) *reinterpret_cast<v4i *>(&lds[b][k * 4096 + tid * 16]) = g[k];
        if (tid < 8 * TB) *reinterpret_cast<v4i *>(&lds[b][kStageBytes + tid * 16]) = gc;
    };
    if (tile0 < tile1) {
        fetch(tile0);
        stash(0);
    }
    __syncthreads();
    for (int t = tile0; t < tile1; t += TB) {
        const int b = ((t - tile0) / TB) & 1;
        const bool more = t + TB < tile1;
        if (more) fetch(t + TB);
#pragma unroll
        for (int u = 0; u < TB; u++) {
            const uint8_t *img = &lds[b][u * kBfTileBytes];
            // A operand of step s: piece h * 8 + s of row r
            v4i a[8];
#pragma unroll
            for (int s = 0; s < 8; s++) a[s] = *reinterpret_cast<const v4i *>(img + ((r << 4) + ((h * 8 + s) ^ (r & 15))) * 16);
            // accumulator constants: register i of a lane is row (i & 3) + 8 * (i >> 2) + 4 * h
            v16i c;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const v4i cc = *reinterpret_cast<const v4i *>(&lds[b][kStageBytes + u * 128 + (8 * k + 4 * h) * 4]);
                c[4 * k + 0] = cc[0]; c[4 * k + 1] = cc[1]; c[4 * k + 2] = cc[2]; c[4 * k + 3] = cc[3];
            }
#pragma unroll
            for (int n = 0; n < NQ; n++) {
                v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], qf[n][0], c, 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 8; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], qf[n][s], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    m2[n] = med3_i32(m1[n], m2[n], acc[i]);
                    m1[n] = max(m1[n], acc[i]);
                }
            }
        }
        if (more) stash(b ^ 1);
        __syncthreads();
    }
    // the two halves of the wave hold different rows of the same queries
#pragma unroll
    for (int n = 0; n < NQ; n++) {
        const int o1 = __shfl_xor(m1[n], 32), o2 = __shfl_xor(m2[n], 32);
        const int hi = max(m1[n], o1), lo = max(min(m1[n], o1), max(m2[n], o2));
        if (h == 0) part[(size_t)blockIdx.y * nq_pad + q0 + n * 32 + r] = make_int2(hi, lo);
    }
}

__global__ __launch_bounds__(256) void k_bf_mfma_merge(const int2 *__restrict__ part, const uint32_t *__restrict__ Q, int nq, int nq_pad,
                                                       int nchunks, int chunk, int32_t *__restrict__ best_idx,
                                                       int32_t *__restrict__ best_d, int32_t *__restrict__ second_d)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    int pc = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) pc += __popc(Q[(size_t)q * 8 + k]);
    int b1 = 256, b2 = 256, bi = -1;
    for (int s = 0; s < nchunks; s++) {
        const int2 p = part[(size_t)s * nq_pad + q];
        if (p.x > kBfInvalid / 2) {
            const int d = pc - (p.x >> 13);
            if (d < b1) { b2 = b1; b1 = d; bi = s * chunk + (kBfChunkMax - 1 - (p.x & (kBfChunkMax - 1))); }
            else if (d < b2) b2 = d;
        }
        if (p.y > kBfInvalid / 2) {
            const int d = pc - (p.y >> 13);
            if (d < b2) b2 = d;
        }
    }
    best_idx[q] = bi;
    best_d[q] = b1;
    second_d[q] = b2;
}

BfMfmaPlan bf_mfma_plan(int nq, int nt)
{
    BfMfmaPlan P;
    P.nq_tile = 2;
    const int qb = 4 * P.nq_tile * 32;
    P.nq_pad = (nq + qb - 1) / qb * qb;
    P.ntiles = ((nt + kBfTile - 1) / kBfTile + kBfStage - 1) / kBfStage * kBfStage;
    P.nt_pad = P.ntiles * kBfTile;
    // enough workgroups to fill 256 CUs several times over, chunks of at most 8192 train descriptors
    const int qblocks = P.nq_pad / qb;
    int want = (4096 + qblocks - 1) / qblocks;
    if (want < 1) want = 1;
    int tpc = (P.ntiles + want - 1) / want;
    if (tpc < 4) tpc = 4;  // a workgroup's query fragments are worth at least a few tiles
    tpc = (tpc + kBfStage - 1) / kBfStage * kBfStage;
    if (tpc > kBfChunkMax / kBfTile) tpc = kBfChunkMax / kBfTile;
    P.tiles_per_chunk = tpc;
    P.chunk = tpc * kBfTile;
    P.nchunks = (P.ntiles + tpc - 1) / tpc;
    size_t off = 0;
    auto part = [&](size_t b) { const size_t o = off; off += (b + 255) & ~(size_t)255; return o; };
    P.off_qx = part((size_t)P.nq_pad * 256);
    P.off_tx = part((size_t)P.ntiles * kBfTileBytes);
    P.off_cinit = part((size_t)P.nt_pad * 4);
    P.off_part = part((size_t)P.nchunks * P.nq_pad * sizeof(int2));
    P.bytes = off;
    return P;
}

int launch_bruteforce_mfma(hipStream_t st, const BfMfmaPlan &P, const uint8_t *d_Q, int nq, const uint8_t *d_T, int nt, uint8_t *ws,
                           int32_t *bi, int32_t *b1, int32_t *b2)
{
    uint8_t *Qx = ws + P.off_qx, *Tx = ws + P.off_tx;
    int32_t *cinit = (int32_t *)(ws + P.off_cinit);
    int2 *part = (int2 *)(ws + P.off_part);
    hipLaunchKernelGGL(k_bf_expand, dim3((P.nq_pad * 16 + 255) / 256), dim3(256), 0, st, d_Q, nq, P.nq_pad, 0, P.chunk, Qx, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_bf_expand, dim3((P.nt_pad * 16 + 255) / 256), dim3(256), 0, st, d_T, nt, P.nt_pad, 1, P.chunk, Tx, cinit);
    const dim3 grid(P.nq_pad / (4 * P.nq_tile * 32), P.nchunks);
    hipLaunchKernelGGL((k_bf_mfma<2, kBfStage>), grid, dim3(256), 0, st, Qx, P.nq_pad, Tx, cinit, P.tiles_per_chunk, P.ntiles, part);
    hipLaunchKernelGGL(k_bf_mfma_merge, dim3((nq + 255) / 256), dim3(256), 0, st, part, (const uint32_t *)d_Q, nq, P.nq_pad, P.nchunks, P.chunk,
                       bi, b1, b2);
    ORBGPU_HIP_CHECK(hipGetLastError());
    return ORBGPU_OK;
}

}  // namespace orbgpu
