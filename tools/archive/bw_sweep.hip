// bw_sweep.hip -- which launch shape gives the highest streaming read / fill / copy rate on MI355X (1 GiB per array)?
//   hipcc --offload-arch=gfx950 -O3 tools/bw_sweep.hip -o tools/_tmp/bw_sweep && tools/_tmp/bw_sweep
// U independent 16-byte accesses in flight per lane and iteration; a block walks a contiguous chunk (CHUNKED) or the whole
// grid strides over the array; non-temporal variants of the stores / loads.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(1024) void k_read(const v4* __restrict__ a, long long n4, float* sink) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    v4 acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 1.2345678e-30f) sink[0] = acc.x;
}
template <int U, int NT>
__global__ __launch_bounds__(1024) void k_fill(v4* __restrict__ a, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const v4 v = {1.0f, 2.0f, 3.0f, 4.0f};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v, a + i + u * stride); else a[i + u * stride] = v;
        }
    }
    for (; i < n4; i += stride) a[i] = v;
}
template <int U, int NT>
__global__ __launch_bounds__(1024) void k_copy(const v4* __restrict__ a, v4* __restrict__ b, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (NT & 1) ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT & 2) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) b[i] = a[i];
}

int main() {
    const long long n = 1ll << 28, n4 = n / 4;
    float *a, *b;
    (void)hipMalloc(&a, n * 4);
    (void)hipMalloc(&b, n * 4);
    (void)hipMemset(a, 0, n * 4);
    (void)hipMemset(b, 0, n * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto bench = [&](const char* what, double bytes, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 7; ++r) {
            (void)hipEventRecord(e0);
            launch();
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-44s %.4f ms (min %.4f)  %.2f TB/s (best %.2f)\n", what, sum / 7, best, bytes / (sum / 7 * 1e-3) / 1e12, bytes / (best * 1e-3) / 1e12);
    };
    char nm[128];
#define SWEEP(U, NT)                                                                                                       \
    for (int blocks : {1024, 2048, 4096, 8192})                                                                            \
        for (int threads : {256, 512}) {                                                                                   \
            snprintf(nm, sizeof nm, "read  U=%d nt=%d grid %d x %d", U, NT, blocks, threads);                             \
            bench(nm, 4.0 * n, [&] { hipLaunchKernelGGL((k_read<U, NT>), dim3(blocks), dim3(threads), 0, 0, (const v4*)a, n4, b); }); \
            snprintf(nm, sizeof nm, "fill  U=%d nt=%d grid %d x %d", U, NT, blocks, threads);                             \
            bench(nm, 4.0 * n, [&] { hipLaunchKernelGGL((k_fill<U, NT>), dim3(blocks), dim3(threads), 0, 0, (v4*)a, n4); });          \
            snprintf(nm, sizeof nm, "copy  U=%d nt=%d grid %d x %d", U, NT, blocks, threads);                             \
            bench(nm, 8.0 * n, [&] { hipLaunchKernelGGL((k_copy<U, (NT ? 3 : 0)>), dim3(blocks), dim3(threads), 0, 0, (const v4*)a, (v4*)b, n4); }); \
        }
    SWEEP(1, 0) SWEEP(4, 0) SWEEP(8, 0) SWEEP(4, 1)
    bench("copy U=4 nt loads only 4096x256", 8.0 * n, [&] { hipLaunchKernelGGL((k_copy<4, 1>), dim3(4096), dim3(256), 0, 0, (const v4*)a, (v4*)b, n4); });
    bench("copy U=4 nt stores only 4096x256", 8.0 * n, [&] { hipLaunchKernelGGL((k_copy<4, 2>), dim3(4096), dim3(256), 0, 0, (const v4*)a, (v4*)b, n4); });
    bench("hipMemcpyDtoD", 8.0 * n, [&] { (void)hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
