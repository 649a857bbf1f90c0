// Micro-benchmark 2: what does the vector-memory path of a CU charge for a wave load instruction, as a function of how
// many distinct cache lines its 64 lanes touch?  Design input for the fast matcher:
//   * 16-byte records, G adjacent lanes reading G adjacent records (G = 1 .. 16), group start aligned to G records or not;
//   * 4-byte table entries, random per lane, or lane pairs reading two entries of one line;
//   * the same candidate reads served from LDS (ds_read_b128 at random positions of a wave-private tile).
// Windows: 8 KB (L1-resident), 256 KB (L2), 16 MB (L2 / Infinity Cache, like the map).
// Build: hipcc --offload-arch=gfx950 -O3 gather_bench2.hip -o gather_bench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

__device__ __forceinline__ unsigned next_pos(unsigned s) { return s * 1664525u + 1013904223u; }

// G lanes of a group read G adjacent 16-byte records at a pseudo-random position of the wave's window; four loads in flight per lane
template <int G, int ALIGNED>
__global__ __launch_bounds__(64) void k_gather16(const float4 *__restrict__ recs, const unsigned *__restrict__ seeds, unsigned n_mask,
                                                  unsigned w_mask, int trips, float *__restrict__ out)
{
    const int lane = threadIdx.x, wave = blockIdx.x;
    const unsigned base = seeds[wave * 64] & n_mask & ~w_mask;
    unsigned s = seeds[wave * 64 + (lane / G) * G];
    float acc = 0.f;
    for (int t = 0; t < trips; t += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = next_pos(s);
            unsigned p = (s >> 8) & w_mask;
            if (ALIGNED) p &= ~(unsigned)(G - 1);
            p = min(p, w_mask - (G - 1));
            v[u] = recs[base + p + (lane % G)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x * v[u].y + v[u].z;
        s += (acc > 1e30f) ? 1u : 0u;            // the next addresses depend on the data: one round trip per trip
    }
    out[wave * 64 + lane] = acc;
}

// 4-byte entries: PAIR = 0 random per lane; PAIR = 1 lanes 2k / 2k+1 read entries e and e + 5 (one line, as cs[xlo] / cs[xhi + 1])
template <int PAIR>
__global__ __launch_bounds__(64) void k_gather4(const int *__restrict__ tab, const unsigned *__restrict__ seeds, unsigned n_mask,
                                                 unsigned w_mask, int trips, float *__restrict__ out)
{
    const int lane = threadIdx.x, wave = blockIdx.x;
    const unsigned base = seeds[wave * 64] & n_mask & ~w_mask;
    unsigned s = seeds[wave * 64 + (PAIR ? (lane & ~1) : lane)];
    int acc = 0;
    for (int t = 0; t < trips; t += 4) {
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = next_pos(s);
            unsigned p = (s >> 8) & w_mask;
            if (PAIR) p = (p & ~15u) + ((lane & 1) ? 5u : 0u) + (p & 7u);
            p = min(p, w_mask);
            v[u] = tab[base + p];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
        s += (acc == 0x7fffffff) ? 1u : 0u;
    }
    out[wave * 64 + lane] = (float)acc;
}

// candidates served from LDS: the wave stages a tile of TILE records with coalesced loads, then every lane reads random records of it
template <int TILE>
__global__ __launch_bounds__(64) void k_lds16(const float4 *__restrict__ recs, const unsigned *__restrict__ seeds, unsigned n_mask,
                                               int trips, float *__restrict__ out)
{
    __shared__ float4 tile[TILE];
    const int lane = threadIdx.x, wave = blockIdx.x;
    const unsigned base = seeds[wave * 64] & n_mask & ~(unsigned)(TILE - 1);
    for (int j = lane; j < TILE; j += 64) tile[j] = recs[base + j];
    __syncthreads();
    unsigned s = seeds[wave * 64 + lane];
    float acc = 0.f;
    for (int t = 0; t < trips; t += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = next_pos(s);
            v[u] = tile[(s >> 8) & (TILE - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x * v[u].y + v[u].z;
        s += (acc > 1e30f) ? 1u : 0u;
    }
    out[wave * 64 + lane] = acc;
}

template <typename F>
static double timed(F launch)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return ms;
}

int main()
{
    const unsigned n_rec = 1u << 20;                   // 16 MB of records
    const int waves = 256 * 32 * 8, trips = 256;
    std::vector<unsigned> seeds((size_t)waves * 64);
    std::mt19937 rng(1);
    for (auto &s : seeds) s = rng();
    float4 *d_recs; unsigned *d_seeds; float *d_out; int *d_tab;
    hipMalloc(&d_recs, sizeof(float4) * n_rec); hipMemset(d_recs, 0, sizeof(float4) * n_rec);
    hipMalloc(&d_tab, sizeof(int) * 4 * n_rec); hipMemset(d_tab, 0, sizeof(int) * 4 * n_rec);
    hipMalloc(&d_seeds, sizeof(unsigned) * seeds.size());
    hipMemcpy(d_seeds, seeds.data(), sizeof(unsigned) * seeds.size(), hipMemcpyHostToDevice);
    hipMalloc(&d_out, sizeof(float) * seeds.size());
    const double loads = (double)waves * 64 * trips;
    const double cu_cycles_per_ms = 256.0 * 2.4e6;
    auto rep = [&](const char *name, double ms) {
        std::printf("  %-34s %8.1f G lane-loads/s  %6.2f per CU-cycle  (%5.1f CU-cycles per wave instruction)\n", name, loads / ms / 1e6,
                    loads / ms / cu_cycles_per_ms, 64.0 * ms * cu_cycles_per_ms / loads);
    };
#define RUN16(G, A) rep(A ? "16 B, G=" #G " aligned" : "16 B, G=" #G " any start", timed([&] { hipLaunchKernelGGL((k_gather16<G, A>), dim3(waves), dim3(64), 0, 0, d_recs, d_seeds, n_rec - 1, wm, trips, d_out); }))
    for (unsigned window : {64u, 512u, 16384u, 1u << 20}) {
        const unsigned wm = window - 1;
        std::printf("window %u records (%u KB):\n", window, window * 16 / 1024);
        RUN16(1, 1); RUN16(2, 1); RUN16(2, 0); RUN16(4, 1); RUN16(4, 0); RUN16(8, 1); RUN16(8, 0); RUN16(16, 1); RUN16(16, 0);
        const unsigned wm4 = window * 4 - 1;
        rep("4 B, random per lane", timed([&] { hipLaunchKernelGGL((k_gather4<0>), dim3(waves), dim3(64), 0, 0, d_tab, d_seeds, 4 * n_rec - 1, wm4, trips, d_out); }));
        rep("4 B, lane pairs in one line", timed([&] { hipLaunchKernelGGL((k_gather4<1>), dim3(waves), dim3(64), 0, 0, d_tab, d_seeds, 4 * n_rec - 1, wm4, trips, d_out); }));
    }
    std::printf("LDS tiles (wave-private), random ds_read_b128:\n");
    rep("tile 256 records (4 KB)", timed([&] { hipLaunchKernelGGL((k_lds16<256>), dim3(waves), dim3(64), 0, 0, d_recs, d_seeds, n_rec - 1, trips, d_out); }));
    rep("tile 512 records (8 KB)", timed([&] { hipLaunchKernelGGL((k_lds16<512>), dim3(waves), dim3(64), 0, 0, d_recs, d_seeds, n_rec - 1, trips, d_out); }));
    rep("tile 1024 records (16 KB)", timed([&] { hipLaunchKernelGGL((k_lds16<1024>), dim3(waves), dim3(64), 0, 0, d_recs, d_seeds, n_rec - 1, trips, d_out); }));
    return 0;
}
