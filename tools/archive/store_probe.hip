// Write-bandwidth probe for the analysis epilogue's store pattern on MI355X (not part of the product).
// hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int H = 2049, NQ = 32;
typedef float v4f __attribute__((ext_vector_type(4)));

// mode 0: dword stores, lane -> bin lane+64q, 3 arrays (what k_analysis does), pitch in floats
// mode 1: same with nontemporal stores
// mode 2: float4 stores: lane (4a+l') writes bins 4a..4a+3 of segment q0+l'  (quad-transposed layout), pitch must be %4==0
// mode 3: same as 2, nontemporal
template <int MODE>
__global__ __launch_bounds__(1024) void k_rows(float* a0, float* a1, float* a2, long long nrows, int pitch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long f = (long long)blockIdx.x * 16 + wave; f < nrows; f += (long long)gridDim.x * 16) {
        float* r0 = a0 + f * pitch; float* r1 = a1 + f * pitch; float* r2 = a2 + f * pitch;
        const float v = (float)(f & 1023) + lane;
        if (MODE <= 1) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int k = lane + 64 * q;
                if (MODE == 0) { r0[k] = v; r1[k] = v + 1; r2[k] = v + 2; }
                else { __builtin_nontemporal_store(v, &r0[k]); __builtin_nontemporal_store(v + 1, &r1[k]); __builtin_nontemporal_store(v + 2, &r2[k]); }
            }
            if (lane == 0) { r0[2048] = v; r1[2048] = v; r2[2048] = v; }
        } else {
            const int a = lane >> 2, lp = lane & 3;
#pragma unroll
            for (int q0 = 0; q0 < NQ; q0 += 4) {
                const int k = 4 * a + 64 * (q0 + lp);
                float4 x = make_float4(v, v + 1, v + 2, v + 3);
                if (MODE == 2) { *(float4*)(r0 + k) = x; *(float4*)(r1 + k) = x; *(float4*)(r2 + k) = x; }
                else { v4f y = {v, v + 1, v + 2, v + 3}; __builtin_nontemporal_store(y, (v4f*)(r0 + k)); __builtin_nontemporal_store(y, (v4f*)(r1 + k)); __builtin_nontemporal_store(y, (v4f*)(r2 + k)); }
            }
            if (lane == 0) { r0[2048] = v; r1[2048] = v; r2[2048] = v; }
        }
    }
}

__global__ void k_fill(float4* p, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        p[i] = make_float4(1, 2, 3, 4);
}
__global__ void k_copy(const float4* __restrict__ s, float4* __restrict__ d, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        d[i] = s[i];
}

int main() {
    const long long nrows = 56985;
    const int pitchA = 2049, pitchB = 2052;
    float *a0, *a1, *a2;
    const size_t bytes = (size_t)nrows * pitchB * 4;
    CK(hipMalloc(&a0, bytes)); CK(hipMalloc(&a1, bytes)); CK(hipMalloc(&a2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double gbytes) {
        float best = 1e9, sum = 0; const int reps = 12;
        for (int r = 0; r < reps + 2; ++r) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
        }
        printf("%-44s avg %.4f ms  min %.4f ms  -> %.0f GB/s (min)\n", name, sum / reps, best, gbytes / (best * 1e-3) / 1e9);
        return 0;
    };
    const double gA = 3.0 * nrows * 2049 * 4;
    run("dword stores, pitch 2049 (analysis today)", [&] { hipLaunchKernelGGL(k_rows<0>, dim3(256), dim3(1024), 0, 0, a0, a1, a2, nrows, pitchA); }, gA);
    run("dword stores NT, pitch 2049", [&] { hipLaunchKernelGGL(k_rows<1>, dim3(256), dim3(1024), 0, 0, a0, a1, a2, nrows, pitchA); }, gA);
    run("dword stores, pitch 2052", [&] { hipLaunchKernelGGL(k_rows<0>, dim3(256), dim3(1024), 0, 0, a0, a1, a2, nrows, pitchB); }, gA);
    run("float4 stores, pitch 2052", [&] { hipLaunchKernelGGL(k_rows<2>, dim3(256), dim3(1024), 0, 0, a0, a1, a2, nrows, pitchB); }, gA);
    run("float4 stores NT, pitch 2052", [&] { hipLaunchKernelGGL(k_rows<3>, dim3(256), dim3(1024), 0, 0, a0, a1, a2, nrows, pitchB); }, gA);
    run("float4 stores, pitch 2049 (unaligned)", [&] { hipLaunchKernelGGL(k_rows<2>, dim3(256), dim3(1024), 0, 0, a0, a1, a2, nrows, pitchA); }, gA);
    run("dword stores, pitch 2049, grid 512", [&] { hipLaunchKernelGGL(k_rows<0>, dim3(512), dim3(1024), 0, 0, a0, a1, a2, nrows, pitchA); }, gA);
    const long long n4 = (long long)(bytes / 16);
    run("linear float4 fill (1 array)", [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (float4*)a0, n4); }, (double)bytes);
    run("linear float4 copy a0->a1 (R+W bytes)", [&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const float4*)a0, (float4*)a1, n4); }, 2.0 * bytes);
    return 0;
}
