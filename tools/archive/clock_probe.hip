// clock_probe.hip -- effective VALU issue rate under a pure fp32-FMA load, and what s_memtime counts.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int PK>
__global__ __launch_bounds__(1024) void k_fma(float* out, long long* ticks, long long* real, int iters) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = v2f{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    const v2f b = {1.0000001f, 1.0000002f}, c = {1e-7f, 2e-7f};
    const long long r0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
            }
    }
    const long long t1 = clock64();
    const long long r1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[blockIdx.x] = t1 - t0; real[blockIdx.x] = r1 - r0; }
}

int main() {
    const int iters = 100000;
    float* out;
    long long *ticks, *real;
    (void)hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    (void)hipMalloc(&ticks, 1024 * sizeof(long long));
    (void)hipMalloc(&real, 1024 * sizeof(long long));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int pk = 0; pk < 2; ++pk)
        for (int wps = 1; wps <= 4; ++wps)
            for (int rep = 0; rep < 2; ++rep) {
                const int blocks = 256;   // one block of 4*wps waves per CU = wps waves per SIMD
                (void)hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k_fma<1>, dim3(blocks), dim3(256 * wps), 0, 0, out, ticks, real, iters);
                else hipLaunchKernelGGL(k_fma<0>, dim3(blocks), dim3(256 * wps), 0, 0, out, ticks, real, iters);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                long long t, r;
                (void)hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost);
                (void)hipMemcpy(&r, real, sizeof(r), hipMemcpyDeviceToHost);
                const double instrs = (double)wps * iters * 32.0;   // per SIMD
                printf("pk %d waves/SIMD %d: %.3f ms, s_memtime %.3f ticks/instr, s_memrealtime %lld ticks (%.1f MHz), "
                       "instr rate/SIMD %.3f G/s\n", pk, wps, ms, t / instrs, r, r / (ms * 1e3), instrs / (ms * 1e6));
            }
    return 0;
}
