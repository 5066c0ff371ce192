// How many cycles does a wave64 vector instruction hold its SIMD on MI355X, by instruction class and waves per SIMD?
// (profiles/r05_experiments.md: the denominator of the "vector issue" roofline of the cache-resident sweep kernels.)
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue_probe valu_issue_probe.hip && ./valu_issue_probe
// Each kernel runs ITER x 64 independent-enough instructions of one class per wave; grid = CUs x 4 SIMDs x k waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
constexpr int ITER = 4096;

template <int KIND>
__global__ void __launch_bounds__(256) probe(unsigned *out, unsigned seed) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 8 + i;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)a[i];
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if constexpr (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if constexpr (KIND == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if constexpr (KIND == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if constexpr (KIND == 4) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if constexpr (KIND == 5) asm volatile("v_dot4_i32_i8 %0, %1, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if constexpr (KIND == 6) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
                if constexpr (KIND == 7) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(a[i]));
                if constexpr (KIND == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
                if constexpr (KIND == 9) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
                if constexpr (KIND == 10) asm volatile("s_nop 0");
                if constexpr (KIND == 11) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
                if constexpr (KIND == 12) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
            }
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + (unsigned)f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char *name, int cus, unsigned *out) {
    for (int k : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 256 threads = 4 waves = one per SIMD
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(probe<KIND>, dim3(cus * k), dim3(256), 0, 0, out, 1u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<KIND>, dim3(cus * k), dim3(256), 0, 0, out, 2u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)k * ITER * 64;
        std::printf("%-22s waves/SIMD=%d  %.3f ms  %.2f ns per instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", name, k, ms,
                    ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned *out;
    CHECK(hipMalloc(&out, sizeof(unsigned) * 256 * cus * 8));
    std::printf("CUs %d\n", cus);
    run<0>("v_add_u32", cus, out);
    run<1>("v_fma_f32", cus, out);
    run<2>("v_and_b32", cus, out);
    run<3>("v_cndmask_b32", cus, out);
    run<4>("v_add_u32_dpp row_shr", cus, out);
    run<5>("v_dot4_i32_i8", cus, out);
    run<6>("v_cmp_lt_u32", cus, out);
    run<7>("v_bfe_u32", cus, out);
    run<8>("v_mad_u32_u24", cus, out);
    run<9>("v_lshlrev_b32", cus, out);
    run<10>("s_nop 0", cus, out);
    run<11>("v_readlane_b32", cus, out);
    run<12>("s_add_u32", cus, out);
    return 0;
}
