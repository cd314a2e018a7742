// Where would a fused analysis -> synthesis launch read its features from, and what would that cost?  (not part of the
// product)   hipcc --offload-arch=gfx950 -O3 tools/energy_mall_probe.hip -o /tmp/emp && /tmp/emp
//   A  read a buffer of 16 MB ... 1.4 GB over and over (float4 grid-stride): rate, board power, pJ per byte by level
//   B  1.4 GB written by one launch and read by the next (today's analysis -> synthesis hand-over through HBM)
//   C  one launch: every workgroup writes a chunk of S bytes and reads it straight back, chunk after chunk (the fused
//      form: the read finds the lines in L2 / Infinity Cache while the write-back drains behind it)
#include <hip/hip_runtime.h>
#include <cstdio>
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

__global__ __launch_bounds__(512) void k_read(const float4* a, long long n4, int reps, float* out) {
    float acc = 0.f;
    for (int r = 0; r < reps; ++r)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
            const float4 v = a[i];
            acc += v.x + v.y + v.z + v.w;
        }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ __launch_bounds__(512) void k_fill(float4* a, long long n4, float s) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        a[i] = make_float4(s, s + 1.f, s + 2.f, s + 3.f);
}
__global__ __launch_bounds__(512) void k_chunks(float4* a, long long n4, long long chunk4, float s, float* out) {
    float acc = 0.f;
    const long long nchunks = (n4 + chunk4 - 1) / chunk4;
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long long b = c * chunk4, e = (b + chunk4 < n4) ? b + chunk4 : n4;
        for (long long i = b + threadIdx.x; i < e; i += blockDim.x) a[i] = make_float4(s, s + 1.f, s + 2.f, s + 3.f);
        __syncthreads();
        for (long long i = b + threadIdx.x; i < e; i += blockDim.x) {
            const float4 v = a[i];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 1.2345f) out[0] = acc;
}

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
    const long long total = 56985ll * 2049 * 4 * 3;   // the lossless feature matrices of configs[1]
    const long long n4 = total / 16;
    float4* a; float* out;
    CK(hipMalloc(&a, n4 * 16)); CK(hipMemset(a, 0, n4 * 16)); CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
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
        printf("%-64s %.4f ms  %.2f TB/s  %5.0f W  %.4f J above idle  %.0f pJ/B\n", name, ms, b / (ms * 1e-3) / 1e12, w,
               (w - idle_w) * ms * 1e-3, (w - idle_w) * ms * 1e-3 / b * 1e12);
        usleep(300000);
    };
    char nm[160];
    for (long long mb : {16ll, 64ll, 128ll, 192ll, 256ll, 512ll, 1401ll}) {
        const long long m4 = (mb << 20) / 16 < n4 ? (mb << 20) / 16 : n4;
        const int reps = (int)(n4 / m4);
        snprintf(nm, sizeof nm, "A read %lld MB x %d", mb, reps);
        time([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(512), 0, 0, a, m4, reps, out); }, nm, (double)m4 * 16 * reps);
    }
    time([&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(512), 0, 0, a, n4, 1.f); }, "B1 fill 1.4 GB", (double)total);
    time([&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(512), 0, 0, a, n4, 1.f);
               hipLaunchKernelGGL(k_read, dim3(2048), dim3(512), 0, 0, a, n4, 1, out); }, "B fill 1.4 GB, then read it (bytes: both)", 2.0 * total);
    for (long long kb : {64ll, 256ll, 1024ll, 4096ll})
        for (int blocks : {256, 512, 1024}) {
            snprintf(nm, sizeof nm, "C %d workgroups, write %lld KB + read it back, chunk by chunk", blocks, kb);
            time([&] { hipLaunchKernelGGL(k_chunks, dim3(blocks), dim3(512), 0, 0, a, n4, (kb << 10) / 16, 1.f, out); }, nm, 2.0 * total);
        }
    return 0;
}
