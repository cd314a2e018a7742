// Register layout of v_mfma_f32_4x4x1_16b_f32 on gfx950, found by experiment: lane l supplies a = 100 + l, b = one-hot
// probes; prints, for every (lane, register) of D, which (a lane, b lane) product it received.
//   hipcc --offload-arch=gfx950 -O2 tools/archive/mfma4x4_layout_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    // a = 1 + l (distinct), b = 1000^(position): use two passes: first b = 1 for all lanes -> D tells which a lanes are summed
    f32x4 c = {0, 0, 0, 0};
    f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(1 + l), 1.0f, c, 0, 0, 0);
    f32x4 d2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(1 + l), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {
        out[(l * 4 + r) * 2 + 0] = d1[r];
        out[(l * 4 + r) * 2 + 1] = d2[r];
    }
}
int main() {
    float* d;
    hipMalloc(&d, 64 * 4 * 2 * sizeof(float));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[512];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) printf("  r%d a-lane %2d b-lane %2d", r, (int)h[(l * 4 + r) * 2] - 1, (int)h[(l * 4 + r) * 2 + 1] - 1);
        printf("\n");
    }
    return 0;
}
