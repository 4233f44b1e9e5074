// Vector-issue microbenchmark for gfx950: how many cycles a SIMD needs per wave64 instruction of the integer opcodes
// the ORB kernels are made of, at 1 / 2 / 4 / 8 resident waves per SIMD on every CU.  Each lane runs eight
// independent dependency chains of one opcode, so a wave always has an instruction ready; cycles come from the
// shader clock counter (s_memtime) read by the wave itself, the clock rate from the same count over the HIP event time.
// Prints one JSON object.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue.hip -o tools/valu_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kChains = 8, kUnroll = 4, kIters = 4096;  // 8 * 4 * 4096 = 131072 instructions per wave

#define ISSUE_KERNEL(NAME, ASM)                                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, long long *cycles)                           \
    {                                                                                                        \
        uint32_t a[kChains];                                                                                 \
        for (int k = 0; k < kChains; k++) a[k] = threadIdx.x * 2654435761u + k;                              \
        uint32_t b = threadIdx.x * 40503u + 977u, c = 0x03020100u + blockIdx.x;                              \
        __syncthreads();                                                                                     \
        const long long t0 = __builtin_amdgcn_s_memtime();                                                   \
        for (int i = 0; i < kIters; i++) {                                                                   \
            _Pragma("unroll") for (int u = 0; u < kUnroll; u++)                                              \
                _Pragma("unroll") for (int k = 0; k < kChains; k++) asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); \
        }                                                                                                    \
        __builtin_amdgcn_s_waitcnt(0);                                                                       \
        const long long t1 = __builtin_amdgcn_s_memtime();                                                   \
        uint32_t s = 0;                                                                                      \
        for (int k = 0; k < kChains; k++) s ^= a[k];                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                             \
        if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                  \
    }

ISSUE_KERNEL(k_add, "v_add_u32 %0, %0, %1")
ISSUE_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
ISSUE_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
ISSUE_KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
ISSUE_KERNEL(k_pk_max_u16, "v_pk_max_u16 %0, %0, %1")
ISSUE_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
ISSUE_KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 1")
ISSUE_KERNEL(k_dot4_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
ISSUE_KERNEL(k_dot2_u16, "v_dot2_u32_u16 %0, %1, %2, %0")
ISSUE_KERNEL(k_med3, "v_med3_u32 %0, %0, %1, %2")
ISSUE_KERNEL(k_min_u32, "v_min_u32 %0, %0, %1")
ISSUE_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
ISSUE_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
ISSUE_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")

typedef void (*kern_t)(uint32_t *, long long *);
struct Op { const char *name; kern_t fn; };

int main()
{
    const Op ops[] = {{"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_bcnt_u32_b32", k_bcnt}, {"v_pk_min_i16", k_pk_min_i16},
                      {"v_pk_max_u16", k_pk_max_u16}, {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_alignbyte},
                      {"v_dot4_u32_u8", k_dot4_u8}, {"v_dot2_u32_u16", k_dot2_u16}, {"v_med3_u32", k_med3}, {"v_min_u32", k_min_u32},
                      {"v_mul_lo_u32", k_mul_lo}, {"v_lshl_or_b32", k_lshl_or}, {"v_cndmask_b32", k_cndmask}};
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const int max_blocks = ncu * 8;
    uint32_t *d_out;
    long long *d_cyc;
    CHECK(hipMalloc(&d_out, (size_t)max_blocks * 256 * 4));
    CHECK(hipMalloc(&d_cyc, (size_t)max_blocks * 4 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const double instr_per_wave = (double)kChains * kUnroll * kIters;
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"instructions_per_wave\": %.0f, \"ops\": {\n",
           prop.name, prop.gcnArchName, ncu, ncu * 4, instr_per_wave);
    for (size_t o = 0; o < sizeof(ops) / sizeof(ops[0]); o++) {
        printf("  \"%s\": {", ops[o].name);
        for (int wps = 1; wps <= 8; wps *= 2) {  // a 256-thread block puts one wave on each SIMD of a CU
            const int blocks = ncu * wps;
            hipLaunchKernelGGL(ops[o].fn, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);  // warm-up
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(ops[o].fn, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> cyc((size_t)blocks * 4);
            CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
            double mean = 0;
            for (long long c : cyc) mean += (double)c;
            mean /= (double)cyc.size();
            // a wave's own elapsed cycles cover the instructions of the wps waves sharing its SIMD
            const double cycles_per_instr = mean / (instr_per_wave * wps);
            const double ginstr = instr_per_wave * blocks * 4 / (ms * 1e-3) / 1e9;  // whole chip, wave-instructions per second
            printf("%s\"waves_per_simd_%d\": {\"cycles_per_wave_instr_per_simd\": %.3f, \"chip_Gwave_instr_per_s\": %.1f, \"launch_ms\": %.4f, \"counter_GHz\": %.3f}",
                   wps == 1 ? "" : ", ", wps, cycles_per_instr, ginstr, ms, mean / (ms * 1e6));
        }
        printf("}%s\n", o + 1 < sizeof(ops) / sizeof(ops[0]) ? "," : "");
    }
    printf("}}\n");
    return 0;
}
