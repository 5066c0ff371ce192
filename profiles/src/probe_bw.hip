// Round 4: which plain streaming-read pattern reaches the practical HBM rate of an MI355X?  (The round-3 probe --
// grid-stride, 4 loads in flight -- read 5.8 TB/s where the dense sweep kernel sustains 6.65 TB/s beyond the caches.)
// hipcc --offload-arch=gfx950 -O3 profiles/src/probe_bw.hip -o build/probe_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int U>
__global__ void __launch_bounds__(1024) seg_kernel(const float4 *__restrict__ x, long long n4, long long seg4, float *sink) {
    // workgroup b streams its own contiguous segment [b seg4, (b + 1) seg4): U independent 16-byte loads per lane in flight
    const long long base = (long long)blockIdx.x * seg4;
    const long long end = base + seg4 < n4 ? base + seg4 : n4;
    float acc = 0.0f;
    long long i = base + threadIdx.x;
    const long long step = blockDim.x;
    for (; i + (U - 1) * step < end; i += U * step) {
        float4 v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(&x[i + q * step]));
            v[q] = make_float4(t.x, t.y, t.z, t.w);
        }
#pragma unroll
        for (int q = 0; q < U; ++q) acc += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
    for (; i < end; i += step) {
        const float4 a = x[i];
        acc += a.x + a.y + a.z + a.w;
    }
    if (acc == 123456.789f) *sink = acc;
}
template <int U>
__global__ void __launch_bounds__(1024) seg_kernel_t(const float4 *__restrict__ x, long long n4, long long seg4, float *sink) {
    const long long base = (long long)blockIdx.x * seg4;
    const long long end = base + seg4 < n4 ? base + seg4 : n4;
    float acc = 0.0f;
    long long i = base + threadIdx.x;
    const long long step = blockDim.x;
    for (; i + (U - 1) * step < end; i += U * step) {
        float4 v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = x[i + q * step];
#pragma unroll
        for (int q = 0; q < U; ++q) acc += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    }
    for (; i < end; i += step) {
        const float4 a = x[i];
        acc += a.x + a.y + a.z + a.w;
    }
    if (acc == 123456.789f) *sink = acc;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const long long bytes = 4ll << 30, n4 = bytes / 16;
    float4 *buf; float *sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMemset(buf, 0, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto kern, int blocks, int threads) {
        const long long seg4 = (n4 + blocks - 1) / blocks;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, n4, seg4, sink);
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, buf, n4, seg4, sink);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s blocks %6d x %4d threads: %8.1f GB/s\n", name, blocks, threads, 4.0 * bytes / (ms * 1e-3) / 1e9);
        fflush(stdout);
    };
    for (int threads : {256, 512, 1024})
        for (int mult : {2, 4, 8, 32, 128}) {
            const int blocks = 256 * mult * 256 / threads;
            if (blocks < 256) continue;
            run("segments, 4 in flight", seg_kernel_t<4>, blocks, threads);
            run("segments, 8 in flight", seg_kernel_t<8>, blocks, threads);
            run("segments, 8 in flight, nt", seg_kernel<8>, blocks, threads);
            run("segments, 16 in flight", seg_kernel_t<16>, blocks, threads);
        }
    return 0;
}
