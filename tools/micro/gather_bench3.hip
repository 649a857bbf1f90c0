// Micro-benchmark 3: what does a PARTLY ACTIVE gather instruction cost?  K of the 64 lanes of a wave read a random 16-byte
// record of the wave's window, the others have nothing to read.  Three ways of keeping the idle lanes quiet:
//   addr0   the idle lanes load record 0 (straight-line code, what k_knn_grid did in round 1)
//   exec    the idle lanes are switched off by an if (EXEC mask); the four loads of a trip stay in flight together
//   oob     buffer loads; idle lanes carry an out-of-range offset (the range check drops them before the cache)
// Reported: CU-cycles per wave instruction (four loads in flight per lane and trip).
// Build: hipcc --offload-arch=gfx950 -O3 gather_bench3.hip -o gather_bench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

__device__ __forceinline__ unsigned next_pos(unsigned s) { return s * 1664525u + 1013904223u; }
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void k_part(const float4 *__restrict__ recs, const unsigned *__restrict__ seeds, unsigned n_mask,
                                              unsigned w_mask, int trips, int k_active, float *__restrict__ out)
{
    const int lane = threadIdx.x, wave = blockIdx.x;
    const unsigned base = seeds[wave * 64] & n_mask & ~w_mask;
    unsigned s = seeds[wave * 64 + lane];
    // the active lanes are spread over the wave (lane * 37 mod 64 is a permutation)
    const bool on = ((lane * 37) & 63) < k_active;
    float acc = 0.f;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(recs + base), 0, (int)((w_mask + 1) * 16), 0x00020000);
    for (int t = 0; t < trips; t += 4) {
        float4 v[4];
        unsigned p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s = next_pos(s); p[u] = (s >> 8) & w_mask; }
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = recs[base + (on ? p[u] : 0)];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += on ? v[u].x * v[u].y + v[u].z : 0.f;
        }
        if (MODE == 1) {
            if (on) {                            // ONE region for the four loads: they stay in flight together
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = recs[base + p[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += v[u].x * v[u].y + v[u].z;
            }
        }
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, on ? p[u] * 16u : 0xFFFFFFF0u, 0, 0);
                v[u] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += on ? v[u].x * v[u].y + v[u].z : 0.f;
        }
        s += (acc > 1e30f) ? 1u : 0u;            // the next addresses depend on the data: one round trip per trip
    }
    out[wave * 64 + lane] = acc;
}

template <typename F>
static double timed(F launch)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipEventRecord(a);
    launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms;
}

int main()
{
    const unsigned n_rec = 1u << 20;
    const int waves = 256 * 32 * 8, trips = 256;
    std::vector<unsigned> seeds((size_t)waves * 64);
    std::mt19937 rng(1);
    for (auto &s : seeds) s = rng();
    float4 *d_recs; unsigned *d_seeds; float *d_out;
    (void)hipMalloc(&d_recs, sizeof(float4) * n_rec); (void)hipMemset(d_recs, 0, sizeof(float4) * n_rec);
    (void)hipMalloc(&d_seeds, sizeof(unsigned) * seeds.size());
    (void)hipMemcpy(d_seeds, seeds.data(), sizeof(unsigned) * seeds.size(), hipMemcpyHostToDevice);
    (void)hipMalloc(&d_out, sizeof(float) * seeds.size());
    const double instr = (double)waves * trips;
    const double cu_cycles_per_ms = 256.0 * 2.4e6;
    for (unsigned window : {64u, 512u}) {
        const unsigned wm = window - 1;
        std::printf("window %u records per wave (%s): CU-cycles per wave load instruction\n   K active lanes:     addr0    exec     oob\n", window,
                    window == 64 ? "L1-resident" : "L2-resident");
        for (int k : {4, 8, 16, 32, 48, 64}) {
            const double t0 = timed([&] { hipLaunchKernelGGL(k_part<0>, dim3(waves), dim3(64), 0, 0, d_recs, d_seeds, n_rec - 1, wm, trips, k, d_out); });
            const double t1 = timed([&] { hipLaunchKernelGGL(k_part<1>, dim3(waves), dim3(64), 0, 0, d_recs, d_seeds, n_rec - 1, wm, trips, k, d_out); });
            const double t2 = timed([&] { hipLaunchKernelGGL(k_part<2>, dim3(waves), dim3(64), 0, 0, d_recs, d_seeds, n_rec - 1, wm, trips, k, d_out); });
            std::printf("   %2d              %8.1f %8.1f %8.1f\n", k, t0 * cu_cycles_per_ms / instr, t1 * cu_cycles_per_ms / instr, t2 * cu_cycles_per_ms / instr);
        }
    }
    return 0;
}
