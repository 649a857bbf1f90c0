// Micro-benchmark: how fast does a CU gather 16-byte records when the 64 lanes of a wave touch 64, 32, 16 or 8
// distinct cache lines per load instruction?  (Design question for the fast matcher: G lanes per query reading
// adjacent records would touch 64/G lines per instruction.)  Build: hipcc --offload-arch=gfx950 -O3 gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

template <int G>
__global__ __launch_bounds__(64) void k_gather(const float4 *__restrict__ recs, const int *__restrict__ starts, int n_rec, int trips,
                                                float *__restrict__ out, int window)
{
    const int lane = threadIdx.x, wave = blockIdx.x;
    // lane group g = lane / G walks one pseudo-random chain of record positions; the G lanes read adjacent records
    // every wave stays inside a window of `window` records (its neighbourhood of the map: L1 hits after the first touch)
    const int base = starts[wave * 64] % (n_rec - window - G);
    int pos = starts[wave * 64 + (lane / G) * G] % window;
    float acc = 0.f;
    for (int t = 0; t < trips; t += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = (pos + u * 37 * G) % window;
            v[u] = recs[base + p + (lane % G)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x * v[u].y + v[u].z;
        pos = (pos + 4 * 37 * G + (int)(acc > 1e30f)) % window;
    }
    out[wave * 64 + lane] = acc;
}

template <int G>
static double run(const float4 *d_recs, const int *d_starts, int n_rec, float *d_out, int waves, int trips, int window)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_gather<G>, dim3(waves), dim3(64), 0, 0, d_recs, d_starts, n_rec, trips, d_out, window);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_gather<G>, dim3(waves), dim3(64), 0, 0, d_recs, d_starts, n_rec, trips, d_out, window);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    const int n_rec = 1 << 20;                         // 16 MB of records: lives in L2 / MALL like the map points
    const int waves = 256 * 4 * 8 * 8, trips = 256;
    std::vector<int> starts((size_t)waves * 64);
    std::mt19937 rng(1);
    for (auto &s : starts) s = (int)(rng() % (n_rec - 64));
    float4 *d_recs; int *d_starts; float *d_out;
    hipMalloc(&d_recs, sizeof(float4) * n_rec); hipMemset(d_recs, 0, sizeof(float4) * n_rec);
    hipMalloc(&d_starts, sizeof(int) * starts.size());
    hipMemcpy(d_starts, starts.data(), sizeof(int) * starts.size(), hipMemcpyHostToDevice);
    hipMalloc(&d_out, sizeof(float) * starts.size());
    const double loads = (double)waves * 64 * trips;
    for (int window : {512, 4096, 1 << 19}) {
        const double t1 = run<1>(d_recs, d_starts, n_rec, d_out, waves, trips, window);
        const double t2 = run<2>(d_recs, d_starts, n_rec, d_out, waves, trips, window);
        const double t4 = run<4>(d_recs, d_starts, n_rec, d_out, waves, trips, window);
        const double t8 = run<8>(d_recs, d_starts, n_rec, d_out, waves, trips, window);
        std::printf("window %7d records: lane-records/s, G lanes on adjacent 16-byte records:  G=1 %.0f  G=2 %.0f  G=4 %.0f  G=8 %.0f G/s"
                    "   (per CU-cycle at 2.4 GHz: %.2f %.2f %.2f %.2f)\n", window, loads / t1 / 1e6, loads / t2 / 1e6, loads / t4 / 1e6,
                    loads / t8 / 1e6, loads / t1 / 1e6 / 614.4, loads / t2 / 1e6 / 614.4, loads / t4 / 1e6 / 614.4, loads / t8 / 1e6 / 614.4);
    }
    return 0;
}
