// stall_probe -- does a YOUNG process on this box see 25-40 ms pauses of its GPU queue, independent of libpgicp?
// A chain of eight tiny kernels, the last writing a stamp into coherent pinned host memory the host polls (the ICP loop's
// pattern), 600 us of host idling between chains.  Usage: stall_probe [GB to allocate, touch and free first] [chains] [GB to hold while running]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void k_touch(int *a, int n, int v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] += v; }
__global__ void k_stamp(volatile int *flag, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) { __threadfence_system(); *flag = v; } }
int main(int argc, char **argv)
{
    const double gb = argc > 1 ? std::atof(argv[1]) : 0.0;
    const int chains = argc > 2 ? std::atoi(argv[2]) : 1500;
    const double hold_gb = argc > 3 ? std::atof(argv[3]) : 0.0;
    using clk = std::chrono::steady_clock;
    const auto t_start = clk::now();
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int *flag = nullptr, *buf = nullptr;
    hipHostMalloc((void **)&flag, 256, hipHostMallocCoherent | hipHostMallocMapped);
    *flag = 0;
    const int n = 100000;
    hipMalloc((void **)&buf, sizeof(int) * n);
    hipMemset(buf, 0, sizeof(int) * n);
    void *held = nullptr;
    if (hold_gb > 0) { hipMalloc(&held, (size_t)(hold_gb * 1e9)); }
    if (gb > 0) {
        void *big = nullptr;
        if (hipMalloc(&big, (size_t)(gb * 1e9)) == hipSuccess) { hipMemset(big, 1, (size_t)(gb * 1e9)); hipDeviceSynchronize(); hipFree(big); }
    }
    std::vector<double> slow_at, slow_ms;
    double worst = 0, sum = 0;
    for (int c = 1; c <= chains; c++) {
        const auto t0 = clk::now();
        for (int k = 0; k < 7; k++) hipLaunchKernelGGL(k_touch, dim3((n + 255) / 256), dim3(256), 0, st, buf, n, k);
        hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, st, flag, c);
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != c) { }
        const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        sum += ms;
        if (ms > worst) worst = ms;
        if (ms > 5.0) { slow_at.push_back(std::chrono::duration<double, std::milli>(t0 - t_start).count()); slow_ms.push_back(ms); }
        std::this_thread::sleep_for(std::chrono::microseconds(600));
    }
    std::printf("{\"freed_gb_first\": %.1f, \"held_gb\": %.1f, \"chains\": %d, \"mean_chain_ms\": %.4f, \"worst_chain_ms\": %.2f, \"chains_over_5ms\": %zu, \"stall_ms_total\": %.1f, \"first_stalls_at_ms\": [",
                gb, hold_gb, chains, sum / chains, worst, slow_ms.size(), [&] { double s = 0; for (double v : slow_ms) s += v; return s; }());
    for (size_t i = 0; i < slow_at.size() && i < 8; i++) std::printf("%s[%.0f, %.1f]", i ? ", " : "", slow_at[i], slow_ms[i]);
    std::printf("]}\n");
    if (held) hipFree(held);
    return 0;
}
