// magphase_probe.hip -- device memory-rate probes behind the C ABI (mpx_bw_probe).  Not on the MagPhase path: bench.py
// launches them in the same process as the timed kernels so that the roofline object can quote, next to the 8 TB/s
// spec peak, what THIS device sustains for a plain streaming read, a plain streaming write and a copy (SURVEY.md 8d:
// "also report against a measured device copy-kernel ceiling").
// Round 4: one launch shape is not a ceiling (tools/archive/bw_sweep.hip: the same 1 GiB reads at 6.1-6.6 TB/s, fills at 3.8-5.4
// and copies at 4.6-5.5 TB/s depending on grid / block size, accesses in flight per lane and the non-temporal bit; the
// round-3 shape, 2048 x 256 with one access in flight, was among the slowest for fill and copy).  mode = kind + 16 * shape
// selects one of kShapes; the caller times them all and quotes the best per kind.
#include "mpx_common.hpp"

namespace mpx {

typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(512) void k_probe_read(const v4f* __restrict__ a, long long n4, float* __restrict__ sink) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += a[i];
    if (acc.x + acc.y + acc.z + acc.w == 1.2345678e-30f) sink[0] = acc.x;   // keeps the loads alive; practically never taken
}

template <int U, int NT>
__global__ __launch_bounds__(512) void k_probe_fill(v4f* __restrict__ a, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const v4f v = {1.0f, 2.0f, 3.0f, 4.0f};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v, a + i + u * stride);
            else a[i + u * stride] = v;
        }
    }
    for (; i < n4; i += stride) a[i] = v;
}

template <int U, int NT>
__global__ __launch_bounds__(512) void k_probe_copy(const v4f* __restrict__ a, v4f* __restrict__ b, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride);
            else b[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) b[i] = a[i];
}

struct ProbeShape {
    int blocks, threads, unroll, nt;
};
// (the winners of tools/archive/bw_sweep.hip per kind, plus the round-3 shape as index 0)
constexpr ProbeShape kShapes[] = {{2048, 256, 1, 0}, {1024, 256, 1, 0}, {1024, 512, 1, 0}, {8192, 512, 1, 0},
                                  {8192, 512, 4, 0}, {8192, 512, 4, 1}, {4096, 512, 8, 0}};
constexpr int kNumShapes = (int)(sizeof(kShapes) / sizeof(kShapes[0]));

template <int U, int NT>
static void launch_probe(int kind, const ProbeShape& sh, hipStream_t s, float* a, float* b, long long n4) {
    const dim3 grid(sh.blocks), block(sh.threads);
    if (kind == 0) hipLaunchKernelGGL((k_probe_read<U, NT>), grid, block, 0, s, (const v4f*)a, n4, b);
    else if (kind == 1) hipLaunchKernelGGL((k_probe_fill<U, NT>), grid, block, 0, s, (v4f*)a, n4);
    else hipLaunchKernelGGL((k_probe_copy<U, NT>), grid, block, 0, s, (const v4f*)a, (v4f*)b, n4);
}

}  // namespace mpx

using namespace mpx;

extern "C" int mpx_bw_probe_shapes(void) { return kNumShapes; }

extern "C" int mpx_bw_probe(void* stream, int32_t mode, float* a, float* b, int64_t n_floats) {
    const int kind = mode & 15, shape = mode >> 4;
    if (mode < 0 || kind > 2) return fail(MPX_ERR_ARG, "mpx_bw_probe: mode & 15 must be 0 (read), 1 (fill) or 2 (copy)%s");
    if (shape >= kNumShapes) return fail(MPX_ERR_ARG, "mpx_bw_probe: mode >> 4 must be < mpx_bw_probe_shapes()%s");
    if (n_floats < 0 || (n_floats & 3)) return fail(MPX_ERR_ARG, "mpx_bw_probe: n_floats must be a non-negative multiple of 4%s");
    if (n_floats == 0) return MPX_OK;
    if (!a || !b) return fail(MPX_ERR_ARG, "mpx_bw_probe: null pointer%s");
    const long long n4 = n_floats / 4;
    const ProbeShape sh = kShapes[shape];
    hipStream_t s = (hipStream_t)stream;
    if (sh.unroll == 1) launch_probe<1, 0>(kind, sh, s, a, b, n4);
    else if (sh.unroll == 4 && !sh.nt) launch_probe<4, 0>(kind, sh, s, a, b, n4);
    else if (sh.unroll == 4) launch_probe<4, 1>(kind, sh, s, a, b, n4);
    else launch_probe<8, 0>(kind, sh, s, a, b, n4);
    MPX_HIP_CHECK(hipGetLastError());
    return MPX_OK;
}
