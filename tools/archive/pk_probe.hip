// Issue-rate probe for gfx950: scalar v_fma_f32 vs packed v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 and v_fma_f64,
// at 1, 2 and 4 waves per SIMD (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 tools/pk_probe.hip -o /tmp/pk_probe && /tmp/pk_probe
// Every variant runs ITER iterations of 16 independent instructions on 16 (pairs of) registers; reported: cycles per
// instruction per wave (s_memtime) and per SIMD (wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc) {
    float a[16]; f2 p[16]; double d[16];
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 0.5f}; d[i] = a[i]; }
    const float c = 0.999f, e = 1e-3f; const f2 pc = {c, c}, pe = {e, e}; const double dc = c, de = e;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pe));
            if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pe));
            if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dc), "v"(de));
            if (MODE == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(e));
            if (MODE == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(de));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
int run(const char* name, float* out, long long* cyc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4}) {   // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD; 256 CUs x wps blocks
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        const double n = 16.0 * ITER;
        printf("%-14s %d waves/SIMD: %6.2f memtime-ticks/instr/wave   wall %.3f ms = %5.2f ns/instr/SIMD (%.2f cyc @2.4GHz)\n", name, wps,
               (double)c / n, ms, ms * 1e6 / (n * wps), ms * 1e6 / (n * wps) * 2.4);
    }
    return 0;
}

int main() {
    float* out; long long* cyc;
    CK(hipMalloc(&out, 4 * 256 * 1024 * 4)); CK(hipMalloc(&cyc, 8));
    run<0>("v_fma_f32", out, cyc); run<5>("v_add_f32", out, cyc); run<1>("v_pk_fma_f32", out, cyc); run<2>("v_pk_add_f32", out, cyc);
    run<3>("v_pk_mul_f32", out, cyc); run<4>("v_fma_f64", out, cyc); run<6>("v_add_f64", out, cyc);
    return 0;
}
