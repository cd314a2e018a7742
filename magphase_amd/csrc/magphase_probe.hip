// magphase_probe.hip -- device memory-rate probes behind the C ABI (mpx_bw_probe).  Not on the MagPhase path: bench.py
// launches them in the same process as the timed kernels so that the roofline object can quote, next to the 8 TB/s
// spec peak, what THIS device sustains for a plain streaming read, a plain streaming write and a copy (SURVEY.md 8d:
// "also report against a measured device copy-kernel ceiling").  Grid-stride float4 kernels, 2048 x 256 threads.
#include "mpx_common.hpp"

namespace mpx {

__global__ __launch_bounds__(256) void k_probe_read(const float4* __restrict__ a, long long n4, float* __restrict__ sink) {
    float acc = 0.0f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = a[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345678e-30f) sink[0] = acc;   // keeps the loads alive; practically never taken
}

__global__ __launch_bounds__(256) void k_probe_fill(float4* __restrict__ a, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float4 v = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) a[i] = v;
}

__global__ __launch_bounds__(256) void k_probe_copy(const float4* __restrict__ a, float4* __restrict__ b, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}

}  // namespace mpx

using namespace mpx;

extern "C" int mpx_bw_probe(void* stream, int32_t mode, float* a, float* b, int64_t n_floats) {
    if (mode < 0 || mode > 2) return fail(MPX_ERR_ARG, "mpx_bw_probe: mode must be 0 (read), 1 (fill) or 2 (copy)%s");
    if (n_floats < 0 || (n_floats & 3)) return fail(MPX_ERR_ARG, "mpx_bw_probe: n_floats must be a non-negative multiple of 4%s");
    if (n_floats == 0) return MPX_OK;
    if (!a || !b) return fail(MPX_ERR_ARG, "mpx_bw_probe: null pointer%s");
    const long long n4 = n_floats / 4;
    const dim3 grid(2048), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(k_probe_read, grid, block, 0, s, (const float4*)a, n4, b);
    else if (mode == 1) hipLaunchKernelGGL(k_probe_fill, grid, block, 0, s, (float4*)a, n4);
    else hipLaunchKernelGGL(k_probe_copy, grid, block, 0, s, (const float4*)a, (float4*)b, n4);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}
