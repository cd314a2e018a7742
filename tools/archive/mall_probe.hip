// Infinity-Cache probe (not part of the product): how fast is a buffer of S MB read right after it was written,
// compared with reading it cold (after 2 GB of unrelated traffic)?  Tells whether analysis -> synthesis in sub-batches
// whose features fit the 256 MB Infinity Cache would read them from the cache.
// hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void k_write(float4* a, long long n4, float v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        a[i] = make_float4(v, v + 1, v + 2, v + 3);
}
__global__ __launch_bounds__(512) void k_read(const float4* a, long long n4, float* out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) out[0] = acc;
}

int main() {
    const long long big = 2048LL << 20;
    float *buf, *junk, *out;
    CK(hipMalloc(&buf, 1024LL << 20)); CK(hipMalloc(&junk, big)); CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    for (int mb : {32, 64, 128, 175, 256, 350, 512, 1024}) {
        const long long n4 = ((long long)mb << 20) / 16;
        float tw = 0, tr_hot = 0, tr_cold = 0, tw_cold = 0;
        const int reps = 5;
        for (int r = 0; r < reps + 1; ++r) {
            hipLaunchKernelGGL(k_read, dim3(2048), dim3(512), 0, 0, (const float4*)junk, big / 16, out);   // evict
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_write, dim3(2048), dim3(512), 0, 0, (float4*)buf, n4, 1.0f);
            hipEventRecord(e1);
            hipLaunchKernelGGL(k_read, dim3(2048), dim3(512), 0, 0, (const float4*)buf, n4, out);
            hipEventRecord(e2);
            hipEventSynchronize(e2);
            float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
            if (r) { tw += a; tr_hot += b; }
            hipLaunchKernelGGL(k_read, dim3(2048), dim3(512), 0, 0, (const float4*)junk, big / 16, out);   // evict
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_read, dim3(2048), dim3(512), 0, 0, (const float4*)buf, n4, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&a, e0, e1);
            if (r) tr_cold += a;
        }
        const double gb = (double)mb * 1.048576e-3;
        printf("%5d MB: write %.1f us (%.2f TB/s)   read right after the write %.1f us (%.2f TB/s)   read cold %.1f us (%.2f TB/s)\n", mb,
               tw / reps * 1e3, gb / (tw / reps), tr_hot / reps * 1e3, gb / (tr_hot / reps), tr_cold / reps * 1e3, gb / (tr_cold / reps));
    }
    return 0;
}
