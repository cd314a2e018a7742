// Read-bandwidth probe for the lossless synthesis kernel's feature-read pattern on MI355X (not part of the product).
// hipcc --offload-arch=gfx950 -O3 tools/read_probe.hip -o /tmp/read_probe && /tmp/read_probe
//
// What it answers: which read rate does the chip give for 3 x [F x 2049] float32 matrices (1.4 GB at F = 56 985)
//   mode 0  linear float4 stream over the three arrays (grid-stride): the plain read ceiling
//   mode 1  one wave per row triple, dword loads, lane -> bin lane + 64 q (k_synth_lossless' pattern), rows dealt
//           round-robin to persistent waves
//   mode 2  as 1, but every wave owns a CONTIGUOUS range of rows (the run-based pattern of k_synth_ola_pair)
//   mode 3  as 2 with 16-byte loads (4 consecutive bins per lane)
//   mode 4  as 2 with non-temporal loads
//   mode 5  as 2, rows padded to 2112 floats (128-byte aligned rows)
//   mode 6  as 2 with k_synth_ola_pair's paired order: bins lane + 64 j ascending and 2048 - lane - 64 j descending, j < 16,
//           six streams interleaved
//   mode 7  as 6 but stream by stream
//   mode 8  as 6, two waves alternate over the rows of one contiguous range (the pair pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>
#include <string>
#include <dirent.h>
#include <unistd.h>
#include <climits>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int H = 2049;

__global__ __launch_bounds__(512) void k_linear(const float4* a, long long n4, float* out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = a[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) out[0] = acc;
}

template <int MODE>
__global__ __launch_bounds__(512) void k_rows(const float* a0, const float* a1, const float* a2, long long nrows, int pitch,
                                              float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nwaves = (long long)gridDim.x * 8, w = (long long)blockIdx.x * 8 + wave;
    long long f0, f1, fs;
    if (MODE == 1) { f0 = w; f1 = nrows; fs = nwaves; }
    else if (MODE == 8) { const long long np = nwaves / 2, per = (nrows + np - 1) / np; f0 = (w / 2) * per + (w & 1); f1 = min(nrows, (w / 2) * per + per); fs = 2; }
    else { const long long per = (nrows + nwaves - 1) / nwaves; f0 = w * per; f1 = min(nrows, f0 + per); fs = 1; }
    float acc = 0.f;
    for (long long f = f0; f < f1; f += fs) {
        const float* r0 = a0 + f * pitch; const float* r1 = a1 + f * pitch; const float* r2 = a2 + f * pitch;
        if (MODE == 3) {
            float4 v[24];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_memcpy(&v[q], r0 + 256 * q + 4 * lane, 16);
                __builtin_memcpy(&v[8 + q], r1 + 256 * q + 4 * lane, 16);
                __builtin_memcpy(&v[16 + q], r2 + 256 * q + 4 * lane, 16);
            }
#pragma unroll
            for (int q = 0; q < 24; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
        } else if (MODE >= 6) {
            float v[96];
            const float* lo[3] = {r0 + lane, r1 + lane, r2 + lane};
            const float* hi[3] = {r0 + 2048 - lane, r1 + 2048 - lane, r2 + 2048 - lane};
            if (MODE == 7) {
#pragma unroll
                for (int s = 0; s < 3; ++s) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[32 * s + j] = lo[s][64 * j];
#pragma unroll
                    for (int j = 0; j < 16; ++j) v[32 * s + 16 + j] = hi[s][-64 * j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
#pragma unroll
                    for (int s = 0; s < 3; ++s) v[32 * s + j] = lo[s][64 * j];
#pragma unroll
                    for (int s = 0; s < 3; ++s) v[32 * s + 16 + j] = hi[s][-64 * j];
                }
            }
#pragma unroll
            for (int q = 0; q < 96; ++q) acc += v[q];
            acc += lo[0][1024] + lo[1][1024] + lo[2][1024];
        } else {
            float v[96];
#pragma unroll
            for (int q = 0; q < 32; ++q) {
                const int k = lane + 64 * q;
                if (MODE == 4) { v[q] = __builtin_nontemporal_load(&r0[k]); v[32 + q] = __builtin_nontemporal_load(&r1[k]); v[64 + q] = __builtin_nontemporal_load(&r2[k]); }
                else { v[q] = r0[k]; v[32 + q] = r1[k]; v[64 + q] = r2[k]; }
            }
#pragma unroll
            for (int q = 0; q < 96; ++q) acc += v[q];
        }
        acc += r0[2048 - lane] + r1[2048 - lane] + r2[2048 - lane];
    }
    if (acc == 1.2345f) out[0] = acc;
}



// LDS read-modify-write energy: every wave does `iters` x 16 x (ds_read_bW + v_add + ds_write_bW) on its own conflict-free
// region; W = 32 / 64 / 128 bits.  VALU-only control: the same loop with the LDS ops replaced by FMAs.
template <int W>
__global__ __launch_bounds__(768) void k_lds_rmw(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[12 * 64 * 4 * 8];    // per wave: 8 rows of 64 lanes x 4 floats (96 KB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* base = buf + wave * (64 * 4 * 8);
    for (int i = lane; i < 64 * 4 * 8; i += 64) base[i] = (float)i;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (W == 32) {
                float* p = base + (r & 7) * 64 + lane;         // consecutive lanes: conflict-free
                float v = *(volatile float*)p; v += 1.0f; *(volatile float*)p = v;
            } else if (W == 64) {
                typedef float v2 __attribute__((ext_vector_type(2)));
                volatile v2* p = reinterpret_cast<volatile v2*>(base + (r & 7) * 128) + lane;
                v2 v = *p; v += 1.0f; *p = v;
            } else if (W == 128) {
                typedef float v4 __attribute__((ext_vector_type(4)));
                volatile v4* p = reinterpret_cast<volatile v4*>(base + (r & 7) * 256) + lane;
                v4 v = *p; v += 1.0f; *p = v;
            } else {
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %0, %0, %0, %0" : "+v"(acc));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + base[lane];
}
// board power of THIS device: /sys/class/drm/cardN/device/hwmon/hwmonM/power1_input of the card whose PCI address is ours
static std::string find_power_file() {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, 0) != hipSuccess) return "";
    for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
    DIR* d = opendir("/sys/class/drm");
    if (!d) return "";
    std::string found;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "card", 4) != 0 || strchr(e->d_name, '-')) continue;
        std::string dev = std::string("/sys/class/drm/") + e->d_name + "/device";
        char real[PATH_MAX];
        if (!realpath(dev.c_str(), real) || !strstr(real, bus)) continue;
        std::string hw = dev + "/hwmon";
        DIR* h = opendir(hw.c_str());
        if (!h) continue;
        while (dirent* g = readdir(h))
            if (strncmp(g->d_name, "hwmon", 5) == 0) found = hw + "/" + g->d_name + "/power1_input";
        closedir(h);
    }
    closedir(d);
    return found;
}
static double read_watts(const std::string& f) {
    FILE* fp = fopen(f.c_str(), "r");
    if (!fp) return 0;
    double v = 0;
    if (fscanf(fp, "%lf", &v) != 1) v = 0;
    fclose(fp);
    return v * 1e-6;
}

int main() {
    const long long F = 56985;
    const int pitch_pad = 2112;
    float *a[3], *out;
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&a[i], sizeof(float) * F * pitch_pad)); CK(hipMemset(a[i], 0, sizeof(float) * F * pitch_pad)); }
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 3.0 * F * H * 4;
    const std::string pf = find_power_file();
    usleep(1500000);
    const double idle_w = read_watts(pf);
    printf("power file %s, idle %.0f W\n", pf.c_str(), idle_w);
    auto time = [&](auto launch, const char* name, double b) {
        for (int i = 0; i < 3; ++i) launch();
        std::atomic<bool> stop{false};
        double wsum = 0; long wn = 0;
        const auto t0 = std::chrono::steady_clock::now();
        std::thread smp([&] {
            while (!stop) {
                const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                const double w = read_watts(pf);
                if (el > 1.0) { wsum += w; ++wn; }
                usleep(4000);
            }
        });
        float sum = 0; int n = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.2) {
            hipEventRecord(e0);
            for (int r = 0; r < 10; ++r) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); sum += ms / 10; ++n;
        }
        stop = true; smp.join();
        const double ms = sum / n, w = wn ? wsum / wn : 0;
        printf("%-58s %.4f ms  %.2f TB/s  %5.0f W  %.4f J above idle  %.0f pJ/B\n", name, ms, b / (ms * 1e-3) / 1e12, w,
               (w - idle_w) * ms * 1e-3, (w - idle_w) * ms * 1e-3 / b * 1e12);
        usleep(300000);
    };
    const long long n4 = F * H / 4;
    time([&] { for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_linear, dim3(2048), dim3(512), 0, 0, (const float4*)a[i], n4, out); }, "0 linear float4, 3 launches", bytes);
    for (int blocks : {256}) {
        char nm[128];
        snprintf(nm, sizeof nm, "1 row per wave round-robin, dword, %d blocks x 8 waves", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<1>, dim3(blocks), dim3(512), 0, 0, a[0], a[1], a[2], F, H, out); }, nm, bytes);
        snprintf(nm, sizeof nm, "2 contiguous rows per wave, dword, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<2>, dim3(blocks), dim3(512), 0, 0, a[0], a[1], a[2], F, H, out); }, nm, bytes);
        snprintf(nm, sizeof nm, "3 contiguous rows per wave, dwordx4, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<3>, dim3(blocks), dim3(512), 0, 0, a[0], a[1], a[2], F, H, out); }, nm, bytes);
        snprintf(nm, sizeof nm, "4 contiguous rows per wave, dword nontemporal, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<4>, dim3(blocks), dim3(512), 0, 0, a[0], a[1], a[2], F, H, out); }, nm, bytes);
        snprintf(nm, sizeof nm, "6 contiguous rows, paired asc/desc interleaved, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<6>, dim3(blocks), dim3(512), 0, 0, a[0], a[1], a[2], F, H, out); }, nm, bytes);
        snprintf(nm, sizeof nm, "7 contiguous rows, paired, stream by stream, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<7>, dim3(blocks), dim3(512), 0, 0, a[0], a[1], a[2], F, H, out); }, nm, bytes);
        snprintf(nm, sizeof nm, "8 paired order, two waves alternate rows of a range, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<8>, dim3(blocks), dim3(512), 0, 0, a[0], a[1], a[2], F, H, out); }, nm, bytes);
        snprintf(nm, sizeof nm, "5 contiguous rows, dword, pitch 2112, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(k_rows<2>, dim3(blocks), dim3(512), 0, 0, a[0], a[1], a[2], F, pitch_pad, out); }, nm, bytes);
    }
    {
        float* o2; CK(hipMalloc(&o2, 256 * 768 * 4));
        const int iters = 4000;
        const double ops = 256.0 * 12 * iters * 32.0;     // wave-level LDS instructions per launch (16 reads + 16 writes per iteration)
        auto lds = [&](auto launch, const char* name) {
            for (int i = 0; i < 2; ++i) launch();
            std::atomic<bool> stop{false}; double wsum = 0; long wn = 0;
            const auto t0 = std::chrono::steady_clock::now();
            std::thread smp([&] { while (!stop) { const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); const double w = read_watts(pf); if (el > 1.0) { wsum += w; ++wn; } usleep(4000); } });
            float sum = 0; int n = 0;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.2) {
                hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); sum += ms; ++n;
            }
            stop = true; smp.join();
            const double ms = sum / n, w = wn ? wsum / wn : 0;
            printf("%-40s %.3f ms  %5.0f W  %.2f G wave-ops/s  %.2f nJ per wave-op above idle\n", name, ms, w, ops / (ms * 1e6), (w - idle_w) * ms * 1e-3 / ops * 1e9);
            usleep(300000);
        };
        lds([&] { hipLaunchKernelGGL((k_lds_rmw<32>), dim3(256), dim3(768), 0, 0, o2, iters); }, "LDS rmw b32 (12 waves/CU)");
        lds([&] { hipLaunchKernelGGL((k_lds_rmw<64>), dim3(256), dim3(768), 0, 0, o2, iters); }, "LDS rmw b64");
        lds([&] { hipLaunchKernelGGL((k_lds_rmw<128>), dim3(256), dim3(768), 0, 0, o2, iters); }, "LDS rmw b128");
        lds([&] { hipLaunchKernelGGL((k_lds_rmw<0>), dim3(256), dim3(768), 0, 0, o2, iters); }, "control: 2 v_fma per slot (32 per iter)");
    }
    return 0;
}
