// Dev tool (GPU box): what does the staged kernel's window loop cost on this chip?  Persistent workgroups run the loop's
// instruction mix — CH ds_read_b64 off one address + immediate offsets, then CH v_add_f64 — on a 128 x 131 f64 LDS region
// with random window corners, for several wave counts / pipelining depths.  Prints clocks per window per CU.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

constexpr int CH = 7, NCH = 3, LS = 131;

template <int OFF> __device__ __forceinline__ void rd(double& d, unsigned a) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(d) : "v"(a), "n"(OFF)); }
__device__ __forceinline__ void gather(double (&v)[CH], unsigned a) {
    rd<0>(v[0], a); rd<24>(v[1], a); rd<48>(v[2], a); rd<72>(v[3], a); rd<96>(v[4], a); rd<120>(v[5], a); rd<144>(v[6], a);
}
template <int N> __device__ __forceinline__ void wait_but() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void pin(double (&v)[CH]) {
#pragma unroll
    for (int i = 0; i < CH; ++i) asm volatile("" : "+v"(v[i]));
}

// MODE 0: two windows gathered, wait all, both added (round 2).  1: pipelined, one window ahead.  2: four gathered, wait, added.
// 3: reads only (no adds).  4: adds only (no reads).
template <int NW, int MODE, int RS = 128>
__global__ __launch_bounds__(64 * NW, RS == 64 ? 2 : 1) void probe(const unsigned short* __restrict__ win, int per_wave, double* out, long long* clk) {
    __shared__ double tile[RS * LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < RS * LS; t += 64 * NW) tile[t] = (double)(t % 97);
    __syncthreads();
    const int p = (lane / NCH) < 21 ? lane / NCH : 20, k = lane % NCH;
    const unsigned lane_off = (unsigned)(uintptr_t)tile + 8u * (unsigned)(p * LS + k);
    double sum[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) sum[i] = 0.0;
    const unsigned short* w = win + ((size_t)blockIdx.x * NW + wave) * per_wave;
    const long long t0 = clock64();
    for (int b = 0; b < per_wave; b += 64) {
        const int wf = w[b + lane];
        const int offv = 8 * ((wf & 127) * LS + ((wf >> 7) & 127));
        if (MODE == 0) {
            for (int j = 0; j < 64; j += 2) {
                double va[CH], vb[CH];
                gather(va, lane_off + __builtin_amdgcn_readlane(offv, j));
                gather(vb, lane_off + __builtin_amdgcn_readlane(offv, j + 1));
                wait_but<0>(); pin(va); pin(vb);
#pragma unroll
                for (int i = 0; i < CH; ++i) { sum[i] += va[i]; }
#pragma unroll
                for (int i = 0; i < CH; ++i) { sum[i] += vb[i]; }
            }
        } else if (MODE == 1) {
            double va[CH], vb[CH];
            gather(va, lane_off + __builtin_amdgcn_readlane(offv, 0));
            for (int j = 0; j < 64; j += 2) {
                gather(vb, lane_off + __builtin_amdgcn_readlane(offv, j + 1));
                wait_but<CH>(); pin(va);
#pragma unroll
                for (int i = 0; i < CH; ++i) sum[i] += va[i];
                gather(va, lane_off + __builtin_amdgcn_readlane(offv, (j + 2) & 63));
                wait_but<CH>(); pin(vb);
#pragma unroll
                for (int i = 0; i < CH; ++i) sum[i] += vb[i];
            }
            wait_but<0>(); pin(va);
        } else if (MODE == 2) {
            for (int j = 0; j < 64; j += 4) {
                double va[CH], vb[CH], vc[CH], vd[CH];
                gather(va, lane_off + __builtin_amdgcn_readlane(offv, j));
                gather(vb, lane_off + __builtin_amdgcn_readlane(offv, j + 1));
                gather(vc, lane_off + __builtin_amdgcn_readlane(offv, j + 2));
                gather(vd, lane_off + __builtin_amdgcn_readlane(offv, j + 3));
                wait_but<0>(); pin(va); pin(vb); pin(vc); pin(vd);
#pragma unroll
                for (int i = 0; i < CH; ++i) { sum[i] += va[i]; sum[i] += vb[i]; sum[i] += vc[i]; sum[i] += vd[i]; }
            }
        } else if (MODE == 3) {
            for (int j = 0; j < 64; j += 2) {
                double va[CH], vb[CH];
                gather(va, lane_off + __builtin_amdgcn_readlane(offv, j));
                gather(vb, lane_off + __builtin_amdgcn_readlane(offv, j + 1));
                wait_but<0>(); pin(va); pin(vb);
                sum[0] += va[0] + vb[6];
            }
        } else if (MODE == 5) {
            // the staged kernel's shape: a slice of 43 windows per wave, pipelined one window ahead with clamped lane
            // numbers, then a workgroup barrier (the end of a block), a burst of LDS stores (the next region), a barrier
            const int je = 43;
            double va[CH], vb[CH];
            int jj = 0;
            gather(va, lane_off + __builtin_amdgcn_readlane(offv, jj));
            for (;;) {
                gather(vb, lane_off + __builtin_amdgcn_readlane(offv, jj + 1 < je ? jj + 1 : jj));
                wait_but<CH>(); pin(va);
#pragma unroll
                for (int i = 0; i < CH; ++i) sum[i] += va[i];
                if (jj + 1 >= je) break;
                gather(va, lane_off + __builtin_amdgcn_readlane(offv, jj + 2 < je ? jj + 2 : jj + 1));
                wait_but<CH>(); pin(vb);
#pragma unroll
                for (int i = 0; i < CH; ++i) sum[i] += vb[i];
                if (jj + 2 >= je) break;
                jj += 2;
            }
            wait_but<0>(); pin(va); pin(vb);
            __syncthreads();
            for (int r = wave; r < RS; r += NW) { tile[r * LS + lane] = sum[0]; tile[r * LS + 64 + lane] = sum[1]; }
            __syncthreads();
        } else {
            for (int j = 0; j < 64; ++j) {
                const double x = (double)__builtin_amdgcn_readlane(offv, j);
#pragma unroll
                for (int i = 0; i < CH; ++i) sum[i] += x;
            }
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) s += sum[i];
    out[(size_t)blockIdx.x * 64 * NW + tid] = s;
    if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NW, int MODE, int RS = 128>
void run(const unsigned short* d_win, int total_per_cu, double* d_out, long long* d_clk, int ncu_in) {
    const int ncu = RS == 64 ? 2 * ncu_in : ncu_in;      // 64-row regions: two workgroups per CU
    total_per_cu = RS == 64 ? total_per_cu / 2 : total_per_cu;
    const int per_wave = total_per_cu / NW / 64 * 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NW, MODE, RS>), dim3(ncu), dim3(64 * NW), 0, 0, d_win, per_wave, d_out, d_clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NW, MODE, RS>), dim3(ncu), dim3(64 * NW), 0, 0, d_win, per_wave, d_out, d_clk);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(ncu); hipMemcpy(h.data(), d_clk, ncu * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : h) mean += c; mean /= ncu;
    const double windows = (MODE == 5 ? (double)(per_wave / 64) * 43 : (double)per_wave) * NW * (RS == 64 ? 2 : 1);
    if (RS == 64) printf("[2 WGs/CU, 64-row regions] ");
    printf("waves %2d mode %d: %.3f ms  %.1f s_memtime ticks/window/CU  (%.2f ns/window/CU)  windows/CU %d\n", NW, MODE, ms, mean / windows,
           ms * 1e6 / windows, (int)windows);
}

int main() {
    int ncu = 256;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); ncu = prop.multiProcessorCount;
    const int total_per_cu = 65536;
    std::vector<unsigned short> h((size_t)ncu * total_per_cu + 64);
    srand(1);
    for (auto& x : h) x = (unsigned short)((rand() % 108) | ((rand() % 108) << 7));
    unsigned short* d_win; double* d_out; long long* d_clk;
    hipMalloc(&d_win, h.size() * 2); hipMalloc(&d_out, (size_t)2 * ncu * 1024 * 8); hipMalloc(&d_clk, 2 * ncu * 8);
    hipMemcpy(d_win, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("CUs %d, clock %d kHz; LDS-bound ideal = 14 clk/window/CU = %.2f ns at 2.4 GHz\n", ncu, prop.clockRate, 14 / 2.4);
#define ALL(NW) run<NW, 0>(d_win, total_per_cu, d_out, d_clk, ncu); run<NW, 1>(d_win, total_per_cu, d_out, d_clk, ncu); \
                run<NW, 2>(d_win, total_per_cu, d_out, d_clk, ncu); run<NW, 3>(d_win, total_per_cu, d_out, d_clk, ncu); \
                run<NW, 4>(d_win, total_per_cu, d_out, d_clk, ncu); run<NW, 5>(d_win, total_per_cu, d_out, d_clk, ncu);
    ALL(4) ALL(8) ALL(16)
    run<8, 5, 64>(d_win, total_per_cu, d_out, d_clk, ncu); run<8, 1, 64>(d_win, total_per_cu, d_out, d_clk, ncu);
    return 0;
}
