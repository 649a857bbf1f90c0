// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the fast matcher (MI355X_MICROARCH.md, HBM:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel touches each cache line of a
// buffer far larger than the 256 MiB Infinity Cache exactly once, so the bytes that must come from memory are known:
//   k_stream        every lane 16 B, coalesced: 1 KiB per wave instruction            -> buffer bytes
//   k_sparse<S>     every lane 16 B at stride S bytes (S = 64, 128, 256): one 16-byte gather per S-byte block
// Run under:  rocprofv3 --pmc FETCH_SIZE -d out -- ./fetch_calib      (and a second pass with the raw TCC_EA0_RDREQ* counters)
// and divide each kernel's FETCH_SIZE (KiB) by the printed byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ p, size_t n16, float *__restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.w; }
    if (acc == 123.456f) out[0] = acc;
}

template <int S>
__global__ __launch_bounds__(256) void k_sparse(const char *__restrict__ p, size_t nblk, float *__restrict__ out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nblk; i += (size_t)gridDim.x * 256) {
        // a lane-dependent 16-byte slot inside the block, so the lanes of a wave do not line up on one sector column
        const float4 v = *(const float4 *)(p + i * S + 16 * ((i * 7) % (S / 16)));
        acc += v.x + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;              // 2 GiB: 8x the Infinity Cache
    char *buf; float *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, bytes);
    (void)hipDeviceSynchronize();
    const int grid = 256 * 8;
    hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_sparse<64>, dim3(grid), dim3(256), 0, 0, buf, bytes / 64, out);
    hipLaunchKernelGGL(k_sparse<128>, dim3(grid), dim3(256), 0, 0, buf, bytes / 128, out);
    hipLaunchKernelGGL(k_sparse<256>, dim3(grid), dim3(256), 0, 0, buf, bytes / 256, out);
    (void)hipDeviceSynchronize();
    std::printf("buffer %zu bytes; k_stream reads all of it; k_sparse<S> reads 16 B of every S-byte block: %zu / %zu / %zu gathers of 16 B\n",
                bytes, bytes / 64, bytes / 128, bytes / 256);
    std::printf("useful bytes: stream %zu, sparse64 %zu, sparse128 %zu, sparse256 %zu\n", bytes, bytes / 4, bytes / 8, bytes / 16);
    return 0;
}
