// Measured HBM bandwidth of this GPU (quote the measured peak beside the 8 TB/s spec).
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_peak.hip -o /tmp/hbm_peak && /tmp/hbm_peak
// read: sum of a 4 GiB buffer (16-byte loads); copy: b = a; triad: a = b + s * c (f64).  Best of 10, bytes moved / time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, size_t n, double* out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 123.456) *out = s;       // keeps the loads alive
}
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_triad(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c,
                                               double s, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 x = b[i], y = c[i];
        a[i] = double2{x.x + s * y.x, x.y + s * y.y};
    }
}

int main() {
    const size_t bytes = (size_t)4 << 30, n = bytes / sizeof(double2);
    double2 *a, *b, *c; double* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const unsigned grid = (unsigned)p.multiProcessorCount * 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto best = [&](auto launch, double moved) {
        float bms = 1e30f;
        for (int r = 0; r < 10; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < bms) bms = ms;
        }
        return moved / (bms * 1e-3) / 1e9;
    };
    const double rd = best([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, out); }, (double)bytes);
    const double cp = best([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    const double tr = best([&] { hipLaunchKernelGGL(k_triad, dim3(grid), dim3(256), 0, 0, a, b, c, 3.0, n); }, 3.0 * bytes);
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"read_GBps\": %.0f, \"copy_GBps\": %.0f, \"triad_GBps\": %.0f}\n",
                p.name, p.multiProcessorCount, rd, cp, tr);
    return 0;
}
