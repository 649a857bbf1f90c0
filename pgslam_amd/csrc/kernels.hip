// kernels.hip -- hand-written gfx950 kernels of the ICP hot path.
//
// Arithmetic contract (shared with oracle/icp_oracle.c, compiled with
// -ffp-contract=off so hipcc never fuses a multiply-add):
//   transform   x' = ((r00*x + r01*y) + r02*z) + tx            in T
//   distance    d2 = ((dx*dx + dy*dy) + dz*dz), dx = q - m     in T
//   neighbour   argmin over (d2, original index), accepted iff d2 <= maxDist^2
//   sums        double, fixed reduction tree (lane -> wave -> block -> block order)
//
// wave = 64 lanes everywhere; no warp-size-32 idiom is used.
#include "kernels.hpp"
#ifndef PGICP_FAST_BLOCK
#define PGICP_FAST_BLOCK 64
#endif
// The fast matcher kernel keeps ~100 scalars live; capping its SGPR allocation at 80 (the rest spill into
// lanes of one VGPR) admits 7 waves per SIMD instead of 6 (800 SGPRs per SIMD: MI355X_MICROARCH.md).
#ifndef PGICP_FAST_ATTR
#define PGICP_FAST_ATTR __attribute__((amdgpu_num_sgpr(80)))
#endif

namespace pgicp {

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
template <typename T> struct Bits;
template <> struct Bits<float> {
    using U = unsigned int;
    static constexpr int kBits = 32;
    __device__ static U key(float v) { return __float_as_uint(v); }
    __device__ static float val(U k) { return __uint_as_float(k); }
    __device__ static float pack_idx(int i) { return __int_as_float(i); }
    __device__ static int unpack_idx(float w) { return __float_as_int(w); }
    __device__ static float inf() { return __uint_as_float(0x7F800000u); }
};
template <> struct Bits<double> {
    using U = unsigned long long;
    static constexpr int kBits = 64;
    __device__ static U key(double v) { return (U)__double_as_longlong(v); }
    __device__ static double val(U k) { return __longlong_as_double((long long)k); }
    __device__ static double pack_idx(int i) { return __longlong_as_double((long long)i); }
    __device__ static int unpack_idx(double w) { return (int)__double_as_longlong(w); }
    __device__ static double inf() { return __longlong_as_double(0x7FF0000000000000LL); }
};

__device__ __forceinline__ float4 make_v4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }
__device__ __forceinline__ double4 make_v4(double x, double y, double z, double w) { return make_double4(x, y, z, w); }

// Blocks b and b+8 share an XCD (round-robin dispatch, observed; used for L2
// locality only).  Give every XCD one contiguous span of tiles so that the map
// cells a span touches stay in that XCD's 4 MiB L2.  Bijective for any nb.
__device__ __forceinline__ int xcd_tile(int b, int nb)
{
    const int q = nb >> 3, r = nb & 7;
    const int xcd = b & 7, j = b >> 3;
    return xcd * q + (xcd < r ? xcd : r) + j;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

template <typename T>
__device__ __forceinline__ void apply_T(const double *Tc, T x, T y, T z, T &ox, T &oy, T &oz)
{
    const T r00 = (T)Tc[0], r01 = (T)Tc[1], r02 = (T)Tc[2], tx = (T)Tc[3];
    const T r10 = (T)Tc[4], r11 = (T)Tc[5], r12 = (T)Tc[6], ty = (T)Tc[7];
    const T r20 = (T)Tc[8], r21 = (T)Tc[9], r22 = (T)Tc[10], tz = (T)Tc[11];
    ox = ((r00 * x + r01 * y) + r02 * z) + tx;
    oy = ((r10 * x + r11 * y) + r12 * z) + ty;
    oz = ((r20 * x + r21 * y) + r22 * z) + tz;
}

// ---------------------------------------------------------------------------
// map build: centroid (order-independent fixed point), bbox, cell sort
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long ordered_key(double v)
{
    const long long i = __double_as_longlong(v);
    return (unsigned long long)(i >= 0 ? (i ^ (long long)0x8000000000000000LL) : ~i);
}

// cell index of coordinate offset u on a grid of n cells of width 1 / inv_h (clamped; shared by the build
// and every query so that both sides round identically)
template <typename T>
__device__ __forceinline__ int clamp_cell(T u, T inv_h, int n)
{
    const T lim = (T)16777216.0;
    return min(max((int)fmin(fmax(floor(u * inv_h), -lim), lim), 0), n - 1);
}

template <typename T>
__device__ __forceinline__ int build_cell(const GridDesc<T> &g, T x, T y, T z)
{
    int cx = (int)floor((x - g.ox) * g.inv_h);
    int cy = (int)floor((y - g.oy) * g.inv_h);
    int cz = (int)floor((z - g.oz) * g.inv_h);
    cx = min(max(cx, 0), g.nx - 1);
    cy = min(max(cy, 0), g.ny - 1);
    cz = min(max(cz, 0), g.nz - 1);
    return cx + g.nx * (cy + g.ny * cz);
}

__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// One atomic per distinct key among the lanes of a wave (clouds arrive in scan order, so neighbouring
// lanes mostly share a cell); falls back to per-lane atomics after 8 leader rounds.  Returns the
// lane's arrival position inside its key's bucket.
__device__ __forceinline__ int wave_bucket_add(int *__restrict__ counts, int key, bool live)
{
    const int lane = threadIdx.x & 63;
    int pos = 0;
    bool todo = live;
    for (int round = 0; round < 8; ++round) {
        const unsigned long long pending = __ballot(todo);
        if (pending == 0ULL) break;
        const int leader = __ffsll((long long)pending) - 1;
        const int lkey = __shfl(key, leader, 64);
        const bool mine = todo && key == lkey;
        const unsigned long long grp = __ballot(mine);
        int base = 0;
        if (lane == leader) base = atomicAdd(&counts[lkey], __popcll(grp));
        base = __shfl(base, leader, 64);
        if (mine) { pos = base + __popcll(grp & ((1ULL << lane) - 1ULL)); todo = false; }
    }
    if (todo) pos = atomicAdd(&counts[key], 1);
    return pos;
}

// The same grouping without the arrival positions: the atomics return nothing, so the wave does not wait for them.
__device__ __forceinline__ void wave_bucket_count(int *__restrict__ counts, int key, bool live)
{
    const int lane = threadIdx.x & 63;
    bool todo = live;
    for (int round = 0; round < 4; ++round) {
        const unsigned long long pending = __ballot(todo);
        if (pending == 0ULL) break;
        const int leader = __ffsll((long long)pending) - 1;
        const int lkey = __builtin_amdgcn_readlane(key, leader);
        const bool mine = todo && key == lkey;
        const unsigned long long grp = __ballot(mine);
        if (lane == leader) atomicAdd(&counts[lkey], __popcll(grp));
        if (mine) todo = false;
    }
    if (todo) atomicAdd(&counts[key], 1);
}

// MapDev::sc_dist -- Chebyshev distance, in super-cells, from every super-cell to the nearest occupied
// one (separable: distance along x, then min over y of max(|dy|, .), then the same over z), capped at
// kScReach + 1.  A query whose super-cell is d super-cells from anything has no point closer than
// (d - 1) * 8h: the matcher answers "no neighbour within maxDist" from this one look-up.
__device__ __forceinline__ void scdist_cell(const int *__restrict__ in, int nsx, int nsy, int nsz, int axis, int first,
                                            int *__restrict__ out, int c)
{
    const int x = c % nsx, y = (c / nsx) % nsy, z = c / (nsx * nsy);
    const int pos = axis == 0 ? x : (axis == 1 ? y : z), len = axis == 0 ? nsx : (axis == 1 ? nsy : nsz);
    const int stride = axis == 0 ? 1 : (axis == 1 ? nsx : nsx * nsy);
    int best = kScReach + 1;
    for (int d = -kScReach; d <= kScReach; ++d) {
        const int p = pos + d;
        if (p < 0 || p >= len) continue;
        const int v = in[c + d * stride];
        // first pass reads occupancy flags (occupied = distance 0), the others read distances
        const int dv = first ? (v > 0 ? 0 : kScReach + 1) : v;
        best = min(best, max(abs(d), dv));
    }
    out[c] = best;
}

// three-phase exclusive scan over `n` ints (n up to 2^27)
#ifndef PGICP_CHUNK_COPIES
#define PGICP_CHUNK_COPIES 32
#endif
constexpr int kChunkCopies = PGICP_CHUNK_COPIES;   // partial copies of the scan's chunk sums (a multiple of the 8 XCDs)
constexpr int kQueueCounterStride = 32;            // ints between the queue counters of two segments: one 128-byte line each
constexpr int kScanChunk = 4096;   // elements per block (1024 threads x 4)

__device__ __forceinline__ int block_exclusive_scan_1024(int v, int *lds /*>=17 ints*/, int &total)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        int w = (lane < (int)(blockDim.x >> 6)) ? lds[lane] : 0;
        int winc = w;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int t = __shfl_up(winc, o, 64);
            if (lane >= o) winc += t;
        }
        if (lane < 16) lds[lane] = winc - w;      // exclusive wave offsets
        if (lane == 15) lds[16] = winc;
    }
    __syncthreads();
    total = lds[16];
    const int res = lds[wid] + inc - v;
    __syncthreads();
    return res;
}

__global__ __launch_bounds__(1024) void k_scan_block_sums(const int *__restrict__ in, int n, int *__restrict__ block_sums)
{
    __shared__ int lds[32];
    const long long base = (long long)blockIdx.x * kScanChunk + threadIdx.x * 4;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) if (base + k < n) s += in[base + k];
    int total;
    block_exclusive_scan_1024(s, lds, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_sum_copies(int *__restrict__ sums, int nb, int ncopies)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    int v = 0;
    for (int r = 0; r < ncopies; ++r) v += sums[(long long)r * nb + i];
    sums[i] = v;
}

__global__ __launch_bounds__(1024) void k_scan_sums_inplace(int *__restrict__ block_sums, int nb)
{
    __shared__ int lds[32];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? block_sums[i] : 0;
        int total;
        const int ex = block_exclusive_scan_1024(v, lds, total);
        if (i < nb) block_sums[i] = ex + carry;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_scan_final(const int *__restrict__ in, int n, const int *__restrict__ block_offs,
                                                      int *__restrict__ out /* n+1 */, int *__restrict__ cursor, int skip_empty)
{
    __shared__ int lds[32];
    const long long base = (long long)blockIdx.x * kScanChunk + threadIdx.x * 4;
    if (skip_empty && blockIdx.x + 1 < gridDim.x && block_offs[blockIdx.x + 1] == block_offs[blockIdx.x]) {
        // nothing counted in this chunk: every entry is the running offset (block-uniform branch)
        const int ex0 = block_offs[blockIdx.x];
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (base + k < n) { out[base + k] = ex0; if (cursor) cursor[base + k] = ex0; }
        return;
    }
    int v[4];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    int total;
    int ex = block_exclusive_scan_1024(s, lds, total) + block_offs[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (base + k < n) { out[base + k] = ex; if (cursor) cursor[base + k] = ex; }
        ex += v[k];
    }
    if (base <= n - 1 && n - 1 < base + 4) out[n] = ex;    // the thread holding the last element
}

// Cells are filled through an atomic cursor (arbitrary order inside a cell); the
// rank kernel then places every point at cell_start + (number of points of the
// same cell with a smaller original index), so the resident layout -- and with
// it every later sum order -- is reproducible run to run.
__global__ __launch_bounds__(256) void k_scatter_idx(int m, const int *__restrict__ cell_of, const int *__restrict__ cell_start,
                                                      const int *__restrict__ arrival, int *__restrict__ order_tmp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    order_tmp[cell_start[cell_of[i]] + arrival[i]] = i;      // arrival position from k_cell_count: no second atomic
}

// ---------------------------------------------------------------------------
// Batched index build: n clouds in one set of launches (blockIdx.y = cloud).  Counting, the prefix scan
// and the scatter run on the concatenation of all clouds' points and cells, so a batch of loop-closure
// candidate maps costs a dozen launches instead of a dozen per map.
// ---------------------------------------------------------------------------
// per cloud: stats[0..2] = fixed-point coordinate sums (int64), stats[3..5] = min keys, stats[6..8] = max keys
template <typename T>
__global__ __launch_bounds__(256) void k_centroid_bbox_b(const BuildDesc<T> *__restrict__ descs, unsigned long long *__restrict__ stats)
{
    const BuildDesc<T> &d = descs[blockIdx.y];
    const T *xyz = d.xyz;
    const int stride = d.xstride, m = d.m;
    unsigned long long *st = stats + 9 * (long long)blockIdx.y;
    long long s[3] = {0, 0, 0};
    double mn[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, mx[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double v = (double)xyz[i * stride + a];
            s[a] += __double2ll_rn(v * 16777216.0);
            mn[a] = fmin(mn[a], v);
            mx[a] = v == v ? fmax(mx[a], v) : HUGE_VAL;
        }
    }
    if ((long long)blockIdx.x * blockDim.x >= m) return;
    // wave results meet in LDS; ONE wave per block then issues the nine atomics (every wave doing so cost
    // 370 us on a 2M-point cloud: ~4000 contended 64-bit atomics per address)
    __shared__ long long w_sum[4][3];
    __shared__ double w_lo[4][3], w_hi[4][3];
    const int wid = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        long long sv = s[a];
        double lo = mn[a], hi = mx[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sv += __shfl_down(sv, o, 64);
            lo = fmin(lo, __shfl_down(lo, o, 64));
            hi = fmax(hi, __shfl_down(hi, o, 64));
        }
        if ((threadIdx.x & 63) == 0) { w_sum[wid][a] = sv; w_lo[wid][a] = lo; w_hi[wid][a] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        long long sv = 0;
        double lo = HUGE_VAL, hi = -HUGE_VAL;
        for (int w = 0; w < 4; w++) { sv += w_sum[w][a]; lo = fmin(lo, w_lo[w][a]); hi = fmax(hi, w_hi[w][a]); }
        atomicAdd(&st[a], (unsigned long long)sv);
        atomicMin(&st[3 + a], ordered_key(lo));
        atomicMax(&st[6 + a], ordered_key(hi));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_cell_count_b(const BuildDesc<T> *__restrict__ descs, int *__restrict__ cell_of,
                                                       int *__restrict__ counts, int *__restrict__ sc_count, int *__restrict__ arrival,
                                                       int *__restrict__ chunk_sums, int n_chunks)
{
    const BuildDesc<T> &d = descs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= d.m) return;
    const bool live = i < d.m;
    const GridDesc<T> g = d.g;
    int c = 0, fx = 0, cy = 0, cz = 0;
    const int kx = d.kx, nxf = g.nx * kx;
    if (live) {
        const T x = d.xyz[(long long)i * d.xstride] - d.mean[0], y = d.xyz[(long long)i * d.xstride + 1] - d.mean[1],
                z = d.xyz[(long long)i * d.xstride + 2] - d.mean[2];
        // the fine x cell with exactly the expression the matcher narrows with (clamp_cell on inv_h * kx)
        fx = clamp_cell<T>(x - g.ox, g.inv_h * (T)kx, nxf);
        cy = min(max((int)floor((y - g.oy) * g.inv_h), 0), g.ny - 1);
        cz = min(max((int)floor((z - g.oz) * g.inv_h), 0), g.nz - 1);
        c = (int)(d.fbase + fx + (long long)nxf * (cy + g.ny * cz));
        cell_of[d.pbase + i] = c;
    }
    const int pos = wave_bucket_add(counts, c, live);
    // the per-chunk sums of the prefix scan are taken here (a wave's points fall into one or two chunks):
    // the cell table of a range scan is 99 % zeros, which the scan then neither sums nor re-reads
    // (one copy of the sums per XCD -- blocks b and b+8 share one: neighbouring chunks share cache lines, and
    // atomics from eight L2s on one line were the whole kernel)
    wave_bucket_count(chunk_sums + (long long)((blockIdx.x + gridDim.x * blockIdx.y) & (kChunkCopies - 1)) * n_chunks, c / kScanChunk, live);
    if (!live) return;
    arrival[d.pbase + i] = pos;
    if (pos == 0) {
        const int nsx = (g.nx + 7) >> 3, nsy = (g.ny + 7) >> 3;
        sc_count[d.sbase + ((fx / kx) >> 3) + nsx * ((cy >> 3) + nsy * (cz >> 3))] = 1;
    }
}

// rank inside the cell by original index, placement, slot map; then the cell table of this cloud is made
// local (offsets into ITS points) by k_localise_b
template <typename T>
__global__ __launch_bounds__(256) void k_rank_place_b(const BuildDesc<T> *__restrict__ descs, const int *__restrict__ cell_of,
                                                       const int *__restrict__ cell_start, const int *__restrict__ order_tmp,
                                                       typename Vec4<T>::type *__restrict__ pts,
                                                       typename Vec4<T>::type *__restrict__ nrm_out, int *__restrict__ slot_of)
{
    const BuildDesc<T> &d = descs[blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d.m) return;
    const int i = order_tmp[d.pbase + j];                    // global point index
    const int c = cell_of[i];
    const int a = cell_start[c], b = cell_start[c + 1];
    int rank = 0;
    for (int k = a; k < b; k += 8) {
        int w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = order_tmp[min(k + u, b - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += (k + u < b && w[u] < i) ? 1 : 0;
    }
    const int pos = a + rank;                                // global slot
    const int li = (int)(i - d.pbase);
    const T x = d.xyz[(long long)li * d.xstride] - d.mean[0], y = d.xyz[(long long)li * d.xstride + 1] - d.mean[1],
            z = d.xyz[(long long)li * d.xstride + 2] - d.mean[2];
    pts[pos] = make_v4(x, y, z, Bits<T>::pack_idx(li));
    if (d.nrm) nrm_out[pos] = make_v4(d.nrm[(long long)li * d.nstride], d.nrm[(long long)li * d.nstride + 1],
                                      d.nrm[(long long)li * d.nstride + 2], (T)0);
    slot_of[i] = (int)pos;                                   // a slot of the batch's shared point array (MapDev::first)
}

// the table of the search cells is every kx-th entry of the fine table (both hold slots of the batch's shared
// point array: no per-cloud rebasing pass -- it was 3 % of a loop-closure batch)
template <typename T>
__global__ __launch_bounds__(256) void k_coarse_table_b(const BuildDesc<T> *__restrict__ descs, const int *__restrict__ cell_start_f,
                                                         int *__restrict__ cell_start)
{
    const BuildDesc<T> &d = descs[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > d.ncells) return;
    const int nx = d.g.nx;
    // cell (cx, row) starts at fine cell (cx * kx, row); the sentinel maps to the fine sentinel
    const long long f = c == d.ncells ? (long long)d.ncells_f : (long long)(c % nx) * d.kx + (long long)nx * d.kx * (c / nx);
    cell_start[d.cbase + c] = cell_start_f[d.fbase + f];
}

// MapDev::near -- for every 2x2x2 BLOCK of cells a nearby OCCUPIED cell.  Pass -1 picks an occupied cell of
// the block itself; three separable sweeps over blocks follow (nearest answer along x, then the best of the
// neighbouring answers in y, then in z).  Answers travel as 6-bit cell offsets from the block's first cell.
// A heuristic, not a nearest-cell guarantee: the matcher only uses it to give a query that starts in empty
// space a first candidate, so that its exact ring search prunes from the first row on.  (Per cell this table
// cost three sweeps over a grid that is 99 % empty: 1.7 ms of a 4.2 ms map build.)
__device__ __forceinline__ int near_pack(int dx, int dy, int dz) { return (dx + 32) | ((dy + 32) << 6) | ((dz + 32) << 12); }
__device__ __forceinline__ int near_dist(int v, int ax, int ay, int az)
{
    // squared distance (in half cells) from the block's centre to the centre of the cell at offset v + (ax, ay, az)
    const int dx = 2 * ((v & 63) - 32 + ax) - 1, dy = 2 * (((v >> 6) & 63) - 32 + ay) - 1, dz = 2 * (((v >> 12) & 63) - 32 + az) - 1;
    return dx * dx + dy * dy + dz * dz;
}

template <typename T>
__global__ __launch_bounds__(256) void k_near_b(const BuildDesc<T> *__restrict__ descs, int pass, const int *__restrict__ cell_start,
                                                 int *__restrict__ tmp_a, int *__restrict__ tmp_b, int *__restrict__ near)
{
    const BuildDesc<T> &d = descs[blockIdx.y];
    const int nx = d.g.nx, ny = d.g.ny, nz = d.g.nz;
    const int nbx = (nx + 1) >> 1, nby = (ny + 1) >> 1, nbz = (nz + 1) >> 1;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;         // blocks <= cells <= 2^26: 32-bit index arithmetic
    if (b >= nbx * nby * nbz) return;
    const int reach_c = d.near_reach < 1 ? 1 : (d.near_reach > kNearReach ? kNearReach : d.near_reach);
    const int reach = (reach_c + 1) >> 1;                        // in blocks
    const int brow = b / nbx;
    const int bx = b - brow * nbx, bz = brow / nby, by = brow - bz * nby;
    if (pass < 0) {
        const int *cs = cell_start + d.cbase;
        int found = -1;
        for (int k = 0; k < 2 && found < 0; ++k)
            for (int j = 0; j < 2 && found < 0; ++j) {
                const int y = 2 * by + j, z = 2 * bz + k;
                if (y >= ny || z >= nz) continue;
                const int c0 = 2 * bx + nx * (y + ny * z);
                const int s0 = cs[c0], s1 = cs[c0 + 1];
                if (s1 > s0) found = near_pack(0, j, k);
                else if (2 * bx + 1 < nx && cs[c0 + 2] > s1) found = near_pack(1, j, k);
            }
        near[d.cbase + b] = found;                                // (sweep input; overwritten by the last pass)
    } else if (pass == 0) {
        const int *in = near + d.cbase;
        int found = -1;
        for (int t = 0; t <= reach && found < 0; ++t) {
            if (bx - t >= 0 && in[b - t] >= 0) found = in[b - t] - 2 * t;             // the x offset sits in the low bits
            else if (bx + t < nbx && in[b + t] >= 0) found = in[b + t] + 2 * t;
        }
        tmp_a[d.cbase + b] = found;
    } else if (pass == 1) {
        const int *in = tmp_a + d.cbase;
        int best = -1, bd = 0x7FFFFFFF;
        for (int t = -reach; t <= reach; ++t) {
            if (by + t < 0 || by + t >= nby) continue;
            const int v = in[b + t * nbx];
            if (v < 0) continue;
            const int dist = near_dist(v, 0, 2 * t, 0);
            if (dist < bd) { bd = dist; best = v + 2 * t * 64; }
        }
        tmp_b[d.cbase + b] = best;
    } else {
        const int *in = tmp_b + d.cbase;
        const int plane = nbx * nby;
        int best = -1, bd = 0x7FFFFFFF;
        for (int t = -reach; t <= reach; ++t) {
            if (bz + t < 0 || bz + t >= nbz) continue;
            const int v = in[b + t * plane];
            if (v < 0) continue;
            const int dist = near_dist(v, 0, 0, 2 * t);
            if (dist < bd) { bd = dist; best = v + 2 * t * 4096; }
        }
        int out = -1;
        if (best >= 0)
            out = (2 * bx + (best & 63) - 32) + nx * ((2 * by + ((best >> 6) & 63) - 32) + ny * (2 * bz + ((best >> 12) & 63) - 32));
        near[d.cbase + b] = out;
    }
}

// MapDev::sc_wit -- for every super-cell the slot of one point of a nearest occupied super-cell (Chebyshev, as
// sc_dist; the first point of that super-cell in layout order), or -1 beyond kWitReach.  A query the matcher
// could give no candidate at all -- a scan point over ground the map does not hold yet -- tests this one point:
// if it lies within maxDist, "a neighbour exists" is settled and the point is the seed of the next iteration;
// without it every such query walked super-cell rings in the wave-per-query path (70 % of that kernel's time
// in a fleet's first iterations).
template <typename T>
__global__ __launch_bounds__(256) void k_sc_first_b(const BuildDesc<T> *__restrict__ descs, const int *__restrict__ sc_count,
                                                     const int *__restrict__ cell_start, int *__restrict__ sc_first)
{
    const BuildDesc<T> &d = descs[blockIdx.y];
    const int C = blockIdx.x * blockDim.x + threadIdx.x;
    if (C >= d.nsc) return;
    const int nx = d.g.nx, ny = d.g.ny, nz = d.g.nz;
    const int nsx = (nx + 7) >> 3, nsy = (ny + 7) >> 3;
    int first = -1;
    if (sc_count[d.sbase + C] > 0) {
        const int X = C % nsx, Y = (C / nsx) % nsy, Z = C / (nsx * nsy);
        const int *cs = cell_start + d.cbase;
        const int xa = 8 * X, xb = min(8 * X + 8, nx);
        for (int r = 0; r < 64 && first < 0; ++r) {
            const int y = 8 * Y + (r & 7), z = 8 * Z + (r >> 3);
            if (y >= ny || z >= nz) continue;
            const int row = nx * (y + ny * z);
            const int a = cs[row + xa], b = cs[row + xb];
            if (b > a) first = a;
        }
    }
    sc_first[d.sbase + C] = first;
}

template <typename T>
__global__ __launch_bounds__(256) void k_sc_wit_b(const BuildDesc<T> *__restrict__ descs, const int *__restrict__ sc_dist,
                                                   const int *__restrict__ sc_first, int *__restrict__ sc_wit)
{
    const BuildDesc<T> &d = descs[blockIdx.y];
    const int C = blockIdx.x * blockDim.x + threadIdx.x;
    if (C >= d.nsc) return;
    const int nsx = (d.g.nx + 7) >> 3, nsy = (d.g.ny + 7) >> 3, nsz = (d.g.nz + 7) >> 3;
    const int X = C % nsx, Y = (C / nsx) % nsy, Z = C / (nsx * nsy);
    const int r = sc_dist[d.sbase + C];
    int wit = -1;
    if (r == 0) wit = sc_first[d.sbase + C];
    else if (r <= kWitReach) {
        // some super-cell of the shell at Chebyshev distance r is occupied: the first in (z, y, x) order wins
        for (int dz = -r; dz <= r && wit < 0; ++dz) {
            const int z = Z + dz;
            if (z < 0 || z >= nsz) continue;
            for (int dy = -r; dy <= r && wit < 0; ++dy) {
                const int y = Y + dy;
                if (y < 0 || y >= nsy) continue;
                const bool face = dz == r || dz == -r || dy == r || dy == -r;
                for (int dx = -r; dx <= r && wit < 0; dx += face ? 1 : 2 * r) {
                    const int x = X + dx;
                    if (x < 0 || x >= nsx) continue;
                    wit = sc_first[d.sbase + x + nsx * (y + nsy * z)];
                }
            }
        }
    }
    sc_wit[d.sbase + C] = wit;
}

// super-cell distance maps: one block per cloud, three sweeps separated by block barriers
template <typename T>
__global__ __launch_bounds__(1024) void k_scdist_b(const BuildDesc<T> *__restrict__ descs, const int *__restrict__ sc_count,
                                                    int *__restrict__ tmp_a, int *__restrict__ tmp_b, int *__restrict__ sc_dist)
{
    const BuildDesc<T> &d = descs[blockIdx.x];
    const int nsx = (d.g.nx + 7) >> 3, nsy = (d.g.ny + 7) >> 3, nsz = (d.g.nz + 7) >> 3;
    const int n = d.nsc;
    int *ta = tmp_a + d.cbase, *tb = tmp_b + d.cbase;         // ncells >= nsc: the cell scratch is large enough
    for (int c = threadIdx.x; c < n; c += blockDim.x) scdist_cell(sc_count + d.sbase, nsx, nsy, nsz, 0, 1, ta, c);
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += blockDim.x) scdist_cell(ta, nsx, nsy, nsz, 1, 0, tb, c);
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += blockDim.x) scdist_cell(tb, nsx, nsy, nsz, 2, 0, sc_dist + d.sbase, c);
}

// the same sweeps one launch each, for clouds whose super-cell grid is too large for one block
template <typename T>
__global__ __launch_bounds__(256) void k_scdist_pass_b(const BuildDesc<T> *__restrict__ descs, int pass, const int *__restrict__ sc_count,
                                                        int *__restrict__ tmp_a, int *__restrict__ tmp_b, int *__restrict__ sc_dist)
{
    const BuildDesc<T> &d = descs[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nsc) return;
    const int nsx = (d.g.nx + 7) >> 3, nsy = (d.g.ny + 7) >> 3, nsz = (d.g.nz + 7) >> 3;
    int *ta = tmp_a + d.cbase, *tb = tmp_b + d.cbase;
    if (pass == 0) scdist_cell(sc_count + d.sbase, nsx, nsy, nsz, 0, 1, ta, c);
    else if (pass == 1) scdist_cell(ta, nsx, nsy, nsz, 1, 0, tb, c);
    else scdist_cell(tb, nsx, nsy, nsz, 2, 0, sc_dist + d.sbase, c);
}

// ---------------------------------------------------------------------------
// rigid transform (a9) -- also the per-scan pre-transform of the ICP prologue
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_transform(const T *__restrict__ in, int in_stride, T *__restrict__ out,
                                                    int out_stride, int n, Mat34 M, int rotate_only)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const T x = in[(long long)i * in_stride], y = in[(long long)i * in_stride + 1], z = in[(long long)i * in_stride + 2];
    const T r00 = (T)M.v[0], r01 = (T)M.v[1], r02 = (T)M.v[2], r10 = (T)M.v[4], r11 = (T)M.v[5], r12 = (T)M.v[6],
            r20 = (T)M.v[8], r21 = (T)M.v[9], r22 = (T)M.v[10];
    T ox = (r00 * x + r01 * y) + r02 * z;
    T oy = (r10 * x + r11 * y) + r12 * z;
    T oz = (r20 * x + r21 * y) + r22 * z;
    if (!rotate_only) { ox = ox + (T)M.v[3]; oy = oy + (T)M.v[7]; oz = oz + (T)M.v[11]; }
    out[(long long)i * out_stride] = ox;
    out[(long long)i * out_stride + 1] = oy;
    out[(long long)i * out_stride + 2] = oz;
}

template <typename T>
__global__ __launch_bounds__(256) void k_pretransform(const ProblemDev *__restrict__ probs, const SrcDesc *__restrict__ src,
                                                       T *__restrict__ rd_pre)
{
    const ProblemDev &P = probs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const SrcDesc s = src[blockIdx.y];
    const T *in = (const T *)s.ptr;
    const T x = in[(long long)i * s.stride], y = in[(long long)i * s.stride + 1], z = in[(long long)i * s.stride + 2];
    T ox, oy, oz;
    apply_T<T>(P.Tpre, x, y, z, ox, oy, oz);
    T *o = rd_pre + 3 * (P.off + i);
    o[0] = ox; o[1] = oy; o[2] = oz;
}


// Readings arrive in scan order, so the lanes of a wave fall into a handful of bins: the lanes of
// each distinct bin are counted with ONE atomic (up to 8 leader rounds, then plain per-lane atomics
// for incoherent input).  The value the atomic returns is the point's arrival position inside its
// bin, so the scatter pass needs no second round of atomics.
// ---------------------------------------------------------------------------
// reading sort: once per scan the (pre-transformed) reading is put in a 3-D-compact order: by block of
// map cells (blocks x-fastest), then by cell inside the block, then by original index.  A rigid
// correction keeps neighbours neighbours, so for every later iteration the 64 queries of a wave share one
// small neighbourhood of the cell-sorted map (L1 locality).
// ---------------------------------------------------------------------------
// hardware square root (1 ulp): only where the result feeds a conservative bound with a margin
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }

template <typename T>
__global__ __launch_bounds__(256) void k_qbin(const ProblemDev *__restrict__ probs, const MapDev<T> *__restrict__ maps,
                                               const T *__restrict__ rd_pre, int max_bins, int bin_shift,
                                               int *__restrict__ qbin, int *__restrict__ counts, int *__restrict__ qpos)
{
    const ProblemDev &P = probs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < P.n;
    int bin = -1;
    if (live) {
        const GridDesc<T> g = maps[P.map].g;
        const T *q = rd_pre + 3 * (P.off + i);
        const int cx = clamp_cell<T>(q[0] - g.ox, g.inv_h, g.nx), cy = clamp_cell<T>(q[1] - g.oy, g.inv_h, g.ny),
                  cz = clamp_cell<T>(q[2] - g.oz, g.inv_h, g.nz);
        const int s = bin_shift, r = (1 << s) - 1;
        const int nbx = (g.nx + r) >> s, nby = (g.ny + r) >> s;
        bin = (cx >> s) + nbx * ((cy >> s) + nby * (cz >> s));
        if (bin >= max_bins) bin = max_bins - 1;
        const int local = (cx & r) | ((cy & r) << s) | ((cz & r) << (2 * s));
        qbin[P.off + i] = (bin << 6) | local;              // bin < 2^25 (<= 2^26 cells / 64 + slack)
    }
    const int pos = wave_bucket_add(counts + (long long)blockIdx.y * max_bins, bin, live);
    if (live) qpos[P.off + i] = pos;
}

__global__ __launch_bounds__(256) void k_qscatter(const ProblemDev *__restrict__ probs, int max_bins,
                                                   const int *__restrict__ qbin, const int *__restrict__ qstart,
                                                   const int *__restrict__ qpos, unsigned long long *__restrict__ qtmp)
{
    const ProblemDev &P = probs[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.n) return;
    const int key = qbin[P.off + i];
    // (bin, cell-in-block key, index) packed: the rank kernel compares these words straight from the bin
    // segment and needs no second look-up of the bin (bin < 2^25)
    qtmp[qstart[(long long)blockIdx.y * max_bins + (key >> 6)] + qpos[P.off + i]] =
        ((unsigned long long)(unsigned int)key << 32) | (unsigned int)i;
}

template <typename T>
__global__ __launch_bounds__(256) void k_qrank(const ProblemDev *__restrict__ probs, int max_bins,
                                                const int *__restrict__ qbin, const int *__restrict__ qstart,
                                                const unsigned long long *__restrict__ qtmp, const T *__restrict__ rd_pre,
                                                T *__restrict__ rd_sorted, int *__restrict__ order)
{
    (void)qbin;
    const ProblemDev &P = probs[blockIdx.y];
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= P.n) return;
    const bool live = j < P.n;
    const int lane = threadIdx.x & 63;
    // The 64 words of a wave are consecutive in bin order, so the segments of their bins form one contiguous
    // span [lo, hi).  The wave reads that span once, 64 words per trip, and every lane counts the words below
    // its own through lane broadcasts -- no lane walks its bin through memory (that loop was the cost of the
    // reading sort: O(bin population) cached reads per point).  All words of earlier bins are smaller and all
    // words of later bins larger, so (words below mine in the span) - (words before my bin) is the rank.
    unsigned long long me = ~0ULL;
    int a = 0x7FFFFFFF, b = 0;
    if (live) {
        me = qtmp[P.off + j];
        const long long bin = (long long)blockIdx.y * max_bins + (long long)(me >> 38);
        a = qstart[bin]; b = qstart[bin + 1];
    }
    const int lo = wave_min_i(a), hi = wave_max_i(b);
    int below = 0;
    for (int t = lo; t < hi; t += 64) {
        const unsigned long long w = t + lane < hi ? qtmp[t + lane] : ~0ULL;
        const unsigned int wl = (unsigned int)w, wh = (unsigned int)(w >> 32);
#pragma unroll 16
        for (int u = 0; u < 64; ++u) {
            const unsigned long long wu = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)wh, u) << 32) |
                                          (unsigned int)__builtin_amdgcn_readlane((int)wl, u);
            below += wu < me ? 1 : 0;
        }
    }
    if (!live) return;
    const int i = (int)(unsigned int)(me & 0xFFFFFFFFu);
    const int f = a + (below - (a - lo));                   // global position (the scan runs over all problems)
    order[f] = i;
    const T *src = rd_pre + 3 * (P.off + i);
    T *dst = rd_sorted + 3 * (long long)f;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// results in reading order for the public matcher output
template <typename T>
__global__ __launch_bounds__(256) void k_unpermute(const MapDev<T> *__restrict__ maps, int map, const int *__restrict__ order,
                                                    const int *__restrict__ slot, const T *__restrict__ d2, int n,
                                                    int *__restrict__ ids_out, T *__restrict__ d2_out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int i = order[j];
    const int s = slot[j];
    ids_out[i] = s < 0 ? s : Bits<T>::unpack_idx(maps[map].pts[s].w);     // -1 none, -2 exists but not located (lazy)
    d2_out[i] = d2[j];
}

// ---------------------------------------------------------------------------
// matcher: exact nearest neighbour on the cell-sorted map
// ---------------------------------------------------------------------------
template <typename T>
struct Best {
    T d2;
    int idx;
    int slot;
#ifdef PGICP_KNN_STATS
    int cnt = 0;      // diagnostics build only: candidates evaluated
#endif
};

#ifdef PGICP_KNN_STATS
__device__ int g_trace_i = -1;          // diagnostics: sorted index of one query of problem 0 to narrate
#define KNN_TRACE(prob_, i_, ...) do { if ((prob_) == 0 && (i_) == g_trace_i) printf(__VA_ARGS__); } while (0)
__device__ unsigned long long g_knn_stats[56];   // [16..31] histogram of own-row, [32..47] of flat-walk candidates per lane (log2 bins)
#define KNN_STAT_WAVE_ADD(slot_, v_)                                                              \
    do {                                                                                          \
        long long s_ = (v_);                                                                      \
        for (int o_ = 32; o_ > 0; o_ >>= 1) s_ += __shfl_xor(s_, o_, 64);                          \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_knn_stats[slot_], (unsigned long long)s_);      \
    } while (0)
#define KNN_STAT_WAVE_MAX(slot_, v_)                                                              \
    do {                                                                                          \
        int s_ = (v_);                                                                            \
        for (int o_ = 32; o_ > 0; o_ >>= 1) s_ = max(s_, __shfl_xor(s_, o_, 64));                  \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_knn_stats[slot_], (unsigned long long)s_);      \
    } while (0)
#else
#define KNN_TRACE(prob_, i_, ...) do { } while (0)
#endif

template <typename T>
__device__ __forceinline__ void eval_point(const typename Vec4<T>::type v, int s, T qx, T qy, T qz, Best<T> &best)
{
    const T dx = qx - v.x, dy = qy - v.y, dz = qz - v.z;
    const T d = (dx * dx + dy * dy) + dz * dz;
    const int idx = Bits<T>::unpack_idx(v.w);
    // selects, not branches: the update is rare and the loop bodies must stay straight-line
    const bool better = (d < best.d2) | ((d == best.d2) & (idx < best.idx));
    best.d2 = better ? d : best.d2;
    best.idx = better ? idx : best.idx;
    best.slot = better ? s : best.slot;
#ifdef PGICP_KNN_STATS
    best.cnt++;
#endif
}

// predicated form for straight-line loops (field-wise selects: a struct-typed ?: goes through scratch)
template <typename T>
__device__ __forceinline__ void eval_point_if(bool on, const typename Vec4<T>::type v, int s, T qx, T qy, T qz, Best<T> &best)
{
    const T dx = qx - v.x, dy = qy - v.y, dz = qz - v.z;
    const T d = (dx * dx + dy * dy) + dz * dz;
    const int idx = Bits<T>::unpack_idx(v.w);
    const bool better = on & ((d < best.d2) | ((d == best.d2) & (idx < best.idx)));
    best.d2 = better ? d : best.d2;
    best.idx = better ? idx : best.idx;
    best.slot = better ? s : best.slot;
#ifdef PGICP_KNN_STATS
    best.cnt += on ? 1 : 0;
#endif
}

// map arrays are reached through pointers stored in a struct, which the compiler can only treat as
// generic (flat) addresses; they are always device-global memory, and global loads do not occupy
// the LDS counter
template <typename V>
__device__ __forceinline__ const __attribute__((address_space(1))) V *as_global(const V *p)
{
    return (const __attribute__((address_space(1))) V *)p;
}
template <typename T> struct Raw4;
template <> struct Raw4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Raw4<double> { typedef double type __attribute__((ext_vector_type(4))); };
// one aligned 16/32-byte global load of a map record
template <typename T>
__device__ __forceinline__ typename Vec4<T>::type load_rec(const typename Vec4<T>::type *p, int j)
{
    const typename Raw4<T>::type r = as_global(reinterpret_cast<const typename Raw4<T>::type *>(p))[j];
    return make_v4(r.x, r.y, r.z, r.w);
}

template <typename T>
__device__ __forceinline__ void scan_range(const typename Vec4<T>::type *__restrict__ pts_generic, int a, int b, T qx, T qy,
                                           T qz, Best<T> &best)
{
    const auto *pts = pts_generic;
    int s = a;
    for (; s + 4 <= b; s += 4) {                 // four independent loads in flight
        const auto v0 = load_rec<T>(pts, s), v1 = load_rec<T>(pts, s + 1), v2 = load_rec<T>(pts, s + 2), v3 = load_rec<T>(pts, s + 3);
        eval_point<T>(v0, s, qx, qy, qz, best);
        eval_point<T>(v1, s + 1, qx, qy, qz, best);
        eval_point<T>(v2, s + 2, qx, qy, qz, best);
        eval_point<T>(v3, s + 3, qx, qy, qz, best);
    }
    if (s < b) {                                 // the 1..3 left over: one round trip, not one each
        const auto v0 = load_rec<T>(pts, s), v1 = load_rec<T>(pts, min(s + 1, b - 1)), v2 = load_rec<T>(pts, min(s + 2, b - 1));
        eval_point<T>(v0, s, qx, qy, qz, best);
        eval_point_if<T>(s + 1 < b, v1, s + 1, qx, qy, qz, best);
        eval_point_if<T>(s + 2 < b, v2, s + 2, qx, qy, qz, best);
    }
}

// distance from coordinate offset u (= x - origin) to the slab of cell c
template <typename T>
__device__ __forceinline__ T slab_dist(T u, int c, T h)
{
    const T lo = (T)c * h - u;          // > 0 when the query is below the slab
    const T hi = u - (T)(c + 1) * h;    // > 0 when above
    return fmax((T)0, fmax(lo, hi));
}

// One (y,z) row of cells, restricted to cells [xa,xb]: the cells of a row are
// contiguous in memory, and only those within sqrt(best - rowdist^2) of the
// query in x can hold a better point.
template <typename T>
__device__ __forceinline__ void scan_row(const MapDev<T> &M, int row_id, int xa, int xb, T ux, T lb2, T qx, T qy, T qz,
                                         Best<T> &best)
{
    // [xa, xb] are cells of the search structure; the range actually read is cut down on the kx-times
    // finer x grid the points are ordered by (every candidate is a 16-byte gather: fewer is faster)
    const int kx = M.kx, nxf = M.g.nx * kx;
    int fa = xa * kx, fb = xb * kx + kx - 1;
    if (best.d2 < Bits<T>::inf()) {
        const T rad = sqrt(fmax(best.d2 - lb2, (T)0)) + M.g.margin;
        const T inv_hx = M.g.inv_h * (T)kx;
        fa = max(fa, clamp_cell<T>(ux - rad, inv_hx, nxf));
        fb = min(fb, clamp_cell<T>(ux + rad, inv_hx, nxf));
        if (fa > fb) return;
    }
    const auto *cs = as_global(M.cell_start_f);
    const int row = nxf * row_id;
    scan_range<T>(M.pts, cs[row + fa], cs[row + fb + 1], qx, qy, qz, best);
}

// guaranteed radius after ring r: every cell outside Chebyshev ring r around c0 is at
// least this far from the query (+inf when the grid is exhausted); margin NOT yet removed
template <typename T>
__device__ __forceinline__ T ring_guarantee(const GridDesc<T> &g, T ux, T uy, T uz, int c0x, int c0y, int c0z, int r)
{
    T gr = Bits<T>::inf();
    if (c0x - r - 1 >= 0) gr = fmin(gr, slab_dist(ux, c0x - r - 1, g.h));
    if (c0x + r + 1 <= g.nx - 1) gr = fmin(gr, slab_dist(ux, c0x + r + 1, g.h));
    if (c0y - r - 1 >= 0) gr = fmin(gr, slab_dist(uy, c0y - r - 1, g.h));
    if (c0y + r + 1 <= g.ny - 1) gr = fmin(gr, slab_dist(uy, c0y + r + 1, g.h));
    if (c0z - r - 1 >= 0) gr = fmin(gr, slab_dist(uz, c0z - r - 1, g.h));
    if (c0z + r + 1 <= g.nz - 1) gr = fmin(gr, slab_dist(uz, c0z + r + 1, g.h));
    return gr;
}

// Exact NN by expanding Chebyshev rings of cells around the query's (clamped)
// cell, starting at ring r_start (rings below it were already examined).  A row
// is skipped when its slab distance already exceeds the best candidate; the
// search is RESOLVED when every unexamined cell is provably farther than the
// best candidate or than maxDist.  At most ring max_rings is walked here (the
// per-lane path); returns false if unresolved, with the last guaranteed radius.
template <typename T>
__device__ __forceinline__ bool grid_nn(const MapDev<T> &M, T qx, T qy, T qz, T max_dist, int r_start, int max_rings,
                                        T stop_d2, Best<T> &best, T &gr_out, int &r_next)
{
    r_next = r_start;
    const GridDesc<T> g = M.g;
    const T ux = qx - g.ox, uy = qy - g.oy, uz = qz - g.oz;
    const int c0x = clamp_cell<T>(ux, g.inv_h, g.nx), c0y = clamp_cell<T>(uy, g.inv_h, g.ny),
              c0z = clamp_cell<T>(uz, g.inv_h, g.nz);
    gr_out = (T)0;
    if (r_start > 0) {
        T gr = ring_guarantee<T>(g, ux, uy, uz, c0x, c0y, c0z, r_start - 1);
        if (!(gr < Bits<T>::inf())) return true;
        gr = gr - g.margin;
        gr_out = gr;
        if (gr > (T)0 && (best.d2 < gr * gr || gr > max_dist)) return true;
        if (gr > (T)0 && gr * gr > stop_d2) return false;               // far enough for the caller: leave it unresolved
    }
    for (int r = r_start; r <= max_rings; ++r) {
        const int z0 = max(c0z - r, 0), z1 = min(c0z + r, g.nz - 1);
        const int y0 = max(c0y - r, 0), y1 = min(c0y + r, g.ny - 1);
        const int xa = max(c0x - r, 0), xb = min(c0x + r, g.nx - 1);
        for (int z = z0; z <= z1; ++z) {
            const T lz = fmax(slab_dist(uz, z, g.h) - g.margin, (T)0);
            const bool zo = (z - c0z == r) || (c0z - z == r);
            for (int y = y0; y <= y1; ++y) {
                const T ly = fmax(slab_dist(uy, y, g.h) - g.margin, (T)0);
                const T lb2 = ly * ly + lz * lz;
                if (lb2 > best.d2) continue;
                const int row = y + g.ny * z;
                if (zo || (y - c0y == r) || (c0y - y == r)) {
                    scan_row<T>(M, row, xa, xb, ux, lb2, qx, qy, qz, best);
                } else {                         // inner row of the shell: only its two end cells are new
                    if (c0x - r >= 0) scan_row<T>(M, row, c0x - r, c0x - r, ux, lb2, qx, qy, qz, best);
                    if (c0x + r <= g.nx - 1) scan_row<T>(M, row, c0x + r, c0x + r, ux, lb2, qx, qy, qz, best);
                }
            }
        }
        T gr = ring_guarantee<T>(g, ux, uy, uz, c0x, c0y, c0z, r);
        if (!(gr < Bits<T>::inf())) return true;                        // grid exhausted
        gr = gr - g.margin;
        gr_out = gr;
        if (gr > (T)0 && (best.d2 < gr * gr || gr > max_dist)) return true;
        r_next = r + 1;
        if (gr > (T)0 && gr * gr > stop_d2) return false;
    }
    return false;
}

// farthest-corner distance from coordinate offset u to the slab [lo, hi]
template <typename T>
__device__ __forceinline__ T slab_far(T u, T lo, T hi)
{
    return fmax(fabs(u - lo), fabs(hi - u));
}

// Phase C of the fast path (shared by both fast kernels): store a resolved result, or queue the
// query with a lower bound while d2 keeps an upper bound (see k_knn_grid's header).
template <typename T>
__device__ __forceinline__ void finish_query(const MapDev<T> &M, const GridDesc<T> &g, const ChainDev<T> &ch, bool resolved, T gr,
                                             Best<T> best, T qx, T qy, T qz, T ux, T uy, T uz, int cx, int cy, int cz, int prob,
                                             int i, long long pos, int r_next, T lb_override, int *__restrict__ slot_io,
                                             T *__restrict__ d2_out, T *__restrict__ none_r,
                                             int *__restrict__ slow_count, int2 *__restrict__ slow_list, T *__restrict__ slow_lb,
                                             int *__restrict__ slow_ring, int q_cap)
{
    // ---- phase C: bookkeeping for the lazy slow path ----
    if (resolved) {
        // (the search pruned with maxDist: it proves nothing beyond it, so no radius is cached here)
        if (best.slot < 0) { best.d2 = Bits<T>::inf(); none_r[pos] = (T)0; }
    } else {
        T lb = lb_override >= (T)0 ? lb_override : (gr > (T)0 ? gr * gr : (T)0);
        if (best.slot < 0) {
            // no candidate yet: one point of a nearest occupied super-cell (MapDev::sc_wit) -- within maxDist it
            // settles that a neighbour exists, and it is a real candidate: the seed of the next iteration
            const int Cx = cx >> 3, Cy = cy >> 3, Cz = cz >> 3;
            const int wslot = as_global(M.sc_wit)[Cx + M.nsx * (Cy + M.nsy * Cz)];
            if (wslot >= 0) {
                Best<T> wb;
                wb.d2 = ch.max_dist2; wb.idx = 0x7FFFFFFF; wb.slot = -1;
                eval_point<T>(M.pts[wslot], wslot, qx, qy, qz, wb);
                if (wb.slot >= 0) best = wb;
            }
        }
        if (best.slot < 0) {
            // still none: look for a certificate that a neighbour within maxDist exists
            const T H = g.h * (T)8;
            const int Cx = cx >> 3, Cy = cy >> 3, Cz = cz >> 3;
            T ub = Bits<T>::inf();
            for (int t = 0; t < 27 && !(ub < Bits<T>::inf()); ++t) {
                const int order = (t + 13) % 27;               // own super-cell first
                const int X = Cx + order % 3 - 1, Y = Cy + (order / 3) % 3 - 1, Z = Cz + order / 9 - 1;
                if (X < 0 || X >= M.nsx || Y < 0 || Y >= M.nsy || Z < 0 || Z >= M.nsz) continue;
                if (M.sc_count[X + M.nsx * (Y + M.nsy * Z)] <= 0) continue;
                const T fx = slab_far(ux, (T)X * H, (T)(X + 1) * H) + g.margin, fy = slab_far(uy, (T)Y * H, (T)(Y + 1) * H) + g.margin,
                        fz = slab_far(uz, (T)Z * H, (T)(Z + 1) * H) + g.margin;
                const T far2 = (fx * fx + fy * fy) + fz * fz;
                if (far2 <= ch.max_dist2) ub = far2;
            }
            if (!(ub < Bits<T>::inf()) && M.m > 0 && !(ch.max_dist2 < Bits<T>::inf())) {
                // maxDist = inf: the whole (non-empty) grid is a certificate
                const T fx = slab_far(ux, (T)0, (T)g.nx * g.h) + g.margin, fy = slab_far(uy, (T)0, (T)g.ny * g.h) + g.margin,
                        fz = slab_far(uz, (T)0, (T)g.nz * g.h) + g.margin;
                ub = (fx * fx + fy * fy) + fz * fz;
            }
            if (ub < Bits<T>::inf()) { best.d2 = ub; best.slot = -2; }   // exists, not located
            else { best.d2 = ch.max_dist2; best.slot = -2; lb = (T)-1; } // existence unknown: always resolved later
        }
        // One atomic per wave: the unresolved lanes of the wave reserve consecutive queue slots -- in the queue
        // SEGMENT of (problem, XCD), whose counter has a cache line of its own.  Nearly every wave queues
        // something (the ~10 % of a scan beyond the search cap), and with one counter for the whole launch
        // those returning atomics, from eight L2s on one line, serialised the kernel: half of its time.
        const unsigned long long m = __ballot(1);
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long)m) - 1;
        const int seg = prob * 8 + (blockIdx.x & 7);
        int base = 0;
        if (lane == leader) base = atomicAdd(slow_count + kQueueCounterStride * seg, __popcll(m));
        base = __shfl(base, leader, 64);
        const long long k = (long long)seg * q_cap + base + __popcll(m & ((1ULL << lane) - 1ULL));
        slow_list[k] = make_int2(prob, i);
        slow_lb[k] = lb;
        slow_ring[k] = r_next;
    }
    slot_io[pos] = best.slot;
    d2_out[pos] = best.d2;
}


// Fast path: one query per lane, 64 consecutive queries of the 3-D-compact sorted
// reading per wave (workgroup = one wave; LDS is wave private).
//
// Start -- previous match as seed (or, without one, the points of MapDev::near's cell); "no neighbour"
// settled without a search where the super-cell distance map or the empty radius of the previous pass
// allows it.
// Phase A -- the 3x3x3 cells around the query's own cell: own row first (it usually holds the neighbour
// and shrinks the bound), then the lane COLLECTS the point ranges of the other eight rows it cannot
// prune (straight-line code, all cell-table look-ups in flight together) and walks the concatenation
// of its non-empty ranges in ONE flat loop.  A wave therefore iterates max-over-lanes(total
// candidates) times, not sum-over-rows(max-over-lanes), which is what lock-step row loops cost on
// divergent data.
// Phase B -- lanes not resolved within that block continue ring by ring (1 ring more when seeded, 3 on
// the first iteration), never beyond 1.21x the previous trim threshold.
// Phase C -- what is still unresolved is queued with a lower bound LB on its true squared distance
// while d2 keeps an UPPER bound (partial best, or a certificate that a non-empty super-cell lies
// wholly within maxDist).  The trimmed-distance filter only needs exact values up to its threshold,
// so k_knn_med / k_knn_slow resolve just the queued queries with LB <= threshold: kept pairs,
// threshold and n_finite stay exact.
constexpr int kFastBlock = PGICP_FAST_BLOCK;

template <typename T, int R>
__global__ __launch_bounds__(kFastBlock) PGICP_FAST_ATTR void k_knn_grid(const ProblemDev *__restrict__ probs, const MapDev<T> *__restrict__ maps,
                                                  const T *__restrict__ rd, int *__restrict__ slot_io,
                                                  T *__restrict__ d2_out, ChainDev<T> ch, int use_seed, int fast_rings,
                                                  int *__restrict__ slow_count, int2 *__restrict__ slow_list,
                                                  T *__restrict__ slow_lb, int *__restrict__ slow_ring,
                                                  const int *__restrict__ active, T *__restrict__ none_r, int q_cap)
{
    using V4 = typename Vec4<T>::type;
    constexpr int NR = (2 * R + 1) * (2 * R + 1);
    __shared__ int rng_a[NR - 1][kFastBlock];      // column `lane` is private to that lane: no barrier needed
    __shared__ int rng_b[NR - 1][kFastBlock];
    const int prob = active[blockIdx.y];           // only problems still iterating are launched
    const ProblemDev &P = probs[prob];
    if (P.done) return;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x;
    const int i = tile * kFastBlock + lane;
    if (tile * kFastBlock >= P.n) return;
    bool live = i < P.n;
    const MapDev<T> M = maps[P.map];
    const GridDesc<T> g = M.g;
    T qx = 0, qy = 0, qz = 0;
    // The outlier filter discards everything beyond its threshold, so the search itself is bounded by
    // 1.21x the PREVIOUS iteration's threshold (cap2): rows and cells farther than that are never
    // visited, which spares the ~15 % of queries that will be trimmed the full 27-cell scan.  A query
    // with nothing inside the cap is queued with LB = cap2 and its seed as upper bound; should this
    // iteration's threshold exceed the cap, the lazy medium/slow path resolves it exactly.
    const T cap2 = use_seed ? (T)(1.21 * P.limit) : Bits<T>::inf();
    const bool capped = cap2 < ch.max_dist2;
    bool still_none = false;
    Best<T> best, seed;
    best.d2 = capped ? cap2 : ch.max_dist2;            // anything farther is useless
    best.idx = 0x7FFFFFFF;
    best.slot = -1;
    seed.d2 = Bits<T>::inf(); seed.idx = 0x7FFFFFFF; seed.slot = -1;
    int dsc = 0;
    if (live) {
        const T *q = rd + 3 * (P.off + i);
        apply_T<T>(P.Tcur, q[0], q[1], q[2], qx, qy, qz);
        {   // the super-cell distance look-up (used below) is requested together with the seed
            const int sx = clamp_cell<T>(qx - g.ox, g.inv_h, g.nx) >> 3, sy8 = clamp_cell<T>(qy - g.oy, g.inv_h, g.ny) >> 3,
                      sz8 = clamp_cell<T>(qz - g.oz, g.inv_h, g.nz) >> 3;
            dsc = as_global(M.sc_dist)[sx + M.nsx * (sy8 + M.nsy * sz8)];
        }
        if (use_seed) {
            const int prev = slot_io[P.off + i];
            if (prev >= 0) {
                eval_point<T>(M.pts[prev], prev, qx, qy, qz, seed);
                if (seed.d2 <= best.d2) best = seed;
            } else if (prev == -1) {
                // Last pass proved that no map point lies within none_r of where the query was; it has
                // moved by |dq| since, so none lies within none_r - |dq| now.  While that still exceeds
                // maxDist the answer stays "no neighbour" without any search (points outside the map's
                // reach would otherwise walk the queue and the slow path in every iteration).
                T px, py, pz;
                apply_T<T>(P.Tcur_prev, q[0], q[1], q[2], px, py, pz);
                const T mx = qx - px, my = qy - py, mz = qz - pz;
                const T left = none_r[P.off + i] - sqrt((mx * mx + my * my) + mz * mz) - g.margin;
                if (left > ch.max_dist) {
                    none_r[P.off + i] = left;
                    d2_out[P.off + i] = Bits<T>::inf();        // slot_io already holds -1
                    still_none = true;
                }
            }
        }
    }
    if (still_none) live = false;
    const T ux = qx - g.ox, uy = qy - g.oy, uz = qz - g.oz;
    const int cx = clamp_cell<T>(ux, g.inv_h, g.nx), cy = clamp_cell<T>(uy, g.inv_h, g.ny), cz = clamp_cell<T>(uz, g.inv_h, g.nz);

    // Far from everything: the nearest occupied super-cell is d super-cells away, so no point is closer than
    // (d - 1) * 8h (measured from the query's own super-cell, in which it lies or to which it was clamped
    // from farther out).  Beyond maxDist that settles "no neighbour" with one look-up -- scan points ahead
    // of a streaming map would otherwise each walk the queue, the medium and the slow path every iteration.
    if (live) {
        const T empty_r = (T)(dsc - 1) * (g.h * (T)8) - g.margin;
        if (empty_r > ch.max_dist) {
            slot_io[P.off + i] = -1;
            d2_out[P.off + i] = Bits<T>::inf();
            none_r[P.off + i] = empty_r;
            live = false;
        }
    }

    // no previous match (first iteration, or nothing located last time): a query that starts in an empty
    // cell takes the points of a nearby occupied cell as first candidates, so every later row test prunes
    // against a finite bound and "a neighbour exists within maxDist" is settled by a real candidate
    if (live && seed.slot < 0) {
        const int c = cx + g.nx * (cy + g.ny * cz);
        const int nc = as_global(M.near)[(cx >> 1) + ((g.nx + 1) >> 1) * ((cy >> 1) + ((g.ny + 1) >> 1) * (cz >> 1))];
        // (the own cell is scanned by phase A.1 anyway -- but under the cap, which hides what lies beyond it)
        if (nc >= 0 && (nc != c || capped)) {
            Best<T> ns;
            ns.d2 = ch.max_dist2; ns.idx = 0x7FFFFFFF; ns.slot = -1;
            scan_range<T>(M.pts, as_global(M.cell_start)[nc], as_global(M.cell_start)[nc + 1], qx, qy, qz, ns);
            if (ns.slot >= 0) { seed = ns; if (seed.d2 <= best.d2) best = seed; }
        }
    }
    // ---- phase A.1: own row first -- it usually holds the neighbour and shrinks the bound ----
    if (live) scan_row<T>(M, cy + g.ny * cz, max(cx - R, 0), min(cx + R, g.nx - 1), ux, (T)0, qx, qy, qz, best);
#ifdef PGICP_KNN_STATS
    KNN_STAT_WAVE_ADD(0, (threadIdx.x & 63) == 0 ? 1 : 0);
    KNN_STAT_WAVE_MAX(1, best.cnt);
    KNN_STAT_WAVE_ADD(2, best.cnt);
    const int cnt_a1 = best.cnt;
    int flat_iters = 0;
#endif
#if defined(PGICP_ABLATE_A2)
    if (false)
#endif
    {
    // ---- phase A.2: collect the ranges of the other rows of the (2R+1)^3 block with that bound ----
    // Straight-line code: the slab terms of the 2R+1 offsets per axis are computed once, every row then
    // costs a handful of selects (a row that is out of the grid, pruned or empty in x loads entry 0 and
    // yields an empty range).  The x-range uses the fast square root: its rounding is far below the margin.
    constexpr int W = 2 * R + 1;
    T ly2[W], lz2[W];
    bool oky[W], okz[W];
#pragma unroll
    for (int d = 0; d < W; ++d) {
        const int y = cy + d - R, z = cz + d - R;
        oky[d] = y >= 0 && y < g.ny;
        okz[d] = z >= 0 && z < g.nz;
        const T l = fmax(slab_dist(uy, y, g.h) - g.margin, (T)0), m = fmax(slab_dist(uz, z, g.h) - g.margin, (T)0);
        ly2[d] = l * l;
        lz2[d] = m * m;
    }
    const int kx = M.kx, nxf = g.nx * kx;
    const T inv_hx = g.inv_h * (T)kx;
    const int row0 = nxf * (cy + g.ny * cz), sy = nxf, sz = nxf * g.ny;
    const int xlo0 = max(cx - R, 0) * kx, xhi0 = min(cx + R, g.nx - 1) * kx + kx - 1;
    const T bound = best.d2;
    int nr = 0;
    // two groups of rows: all look-ups of a group are in flight together, and only half of them are live
    // in registers at a time (the kernel sits at the edge of a register-allocation step)
#pragma unroll
    for (int grp = 0; grp < 2; ++grp) {
        constexpr int H = (NR - 1) / 2;
        int ra[H], rb[H];
#pragma unroll
        for (int u = 0; u < H; ++u) {
            const int t = grp * H + u;
            const int tt = (t < R * W + R) ? t : t + 1;            // skip the centre (own) row
            const int dy = tt % W, dz = tt / W;
            const T lb2 = ly2[dy] + lz2[dz];
            const T rad = fast_sqrt(fmax(bound - lb2, (T)0)) * (T)1.000001 + g.margin;
            const int xlo = max(xlo0, clamp_cell<T>(ux - rad, inv_hx, nxf)), xhi = min(xhi0, clamp_cell<T>(ux + rad, inv_hx, nxf));
            const bool need = live && oky[dy] && okz[dz] && !(lb2 > bound) && xlo <= xhi;
            const int row = row0 + (dy - R) * sy + (dz - R) * sz;
            ra[u] = as_global(M.cell_start_f)[need ? row + xlo : 0];
            rb[u] = as_global(M.cell_start_f)[need ? row + xhi + 1 : 0];
        }
#pragma unroll
        for (int u = 0; u < H; ++u)
            if (ra[u] < rb[u]) { rng_a[nr][lane] = ra[u]; rng_b[nr][lane] = rb[u]; ++nr; }
    }
    // ---- flat walk over the concatenated ranges (LDS is only read by the lane that wrote it) ----
    // Four candidates per trip, as two pairs; the second pair may already belong to the lane's next range, so
    // four gathers per lane are in flight per L1 round trip whatever the range lengths are.  (Carrying a
    // requested pair across the loop edge instead does not work: the register copy on the back edge waits for
    // the data.)
    {
        int k = 0, j = 0, e = 0;
        bool valid = nr > 0;
        if (valid) { j = rng_a[0][lane]; e = rng_b[0][lane]; }
        auto advance = [&]() {
            j += 2;
            if (valid & (j >= e)) {                                   // next non-empty range of this lane
                ++k;
                valid = k < nr;
                if (valid) { j = rng_a[k][lane]; e = rng_b[k][lane]; }
            }
        };
        while (__any(valid)) {
#ifndef PGICP_FLAT_PAIRS
#define PGICP_FLAT_PAIRS 2
#endif
            constexpr int NP = PGICP_FLAT_PAIRS;
            bool pv[NP], pt[NP];
            int pj[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) { pv[u] = valid; pt[u] = valid & (j + 1 < e); pj[u] = j; advance(); }
            V4 c0[NP], c1[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) { c0[u] = load_rec<T>(M.pts, pv[u] ? pj[u] : 0); c1[u] = load_rec<T>(M.pts, pt[u] ? pj[u] + 1 : 0); }
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                eval_point_if<T>(pv[u], c0[u], pj[u], qx, qy, qz, best);
                eval_point_if<T>(pt[u], c1[u], pj[u] + 1, qx, qy, qz, best);
            }
#ifdef PGICP_KNN_STATS
            flat_iters += 2;
#endif
        }
    }
    }
#ifdef PGICP_KNN_STATS
    KNN_STAT_WAVE_MAX(3, flat_iters);
    KNN_STAT_WAVE_ADD(4, best.cnt - cnt_a1);
    if (live) {
        atomicAdd(&g_knn_stats[16 + (cnt_a1 <= 0 ? 0 : min(15, 32 - __clz(cnt_a1)))], 1ULL);
        const int c2 = best.cnt - cnt_a1;
        atomicAdd(&g_knn_stats[32 + (c2 <= 0 ? 0 : min(15, 32 - __clz(c2)))], 1ULL);
    }
    const int cnt_a = best.cnt;
#endif
    if (!live) return;

    // ---- phase B: per-lane continuation from ring R+1 ----
    T gr;
    // Everything beyond the trim threshold is discarded by the outlier filter, so a lane stops as soon as
    // its guaranteed radius passes (1.1x) the PREVIOUS iteration's threshold and leaves the query queued
    // with that lower bound (lazy resolution keeps the result exact).  Lanes that need more than
    // `fast_rings` rings are queued too: k_knn_med continues them in waves made only of such queries.
    int r_next;
    bool resolved = grid_nn<T>(M, qx, qy, qz, ch.max_dist, R + 1, fast_rings, cap2, best, gr, r_next);
#ifdef PGICP_KNN_STATS
    atomicAdd(&g_knn_stats[5], (unsigned long long)(best.cnt - cnt_a));
    atomicAdd(&g_knn_stats[6], (unsigned long long)(resolved ? 0 : 1));
    atomicAdd(&g_knn_stats[7], (unsigned long long)(best.cnt > cnt_a ? 1 : 0));
    KNN_STAT_WAVE_MAX(8, best.cnt - cnt_a);
    KNN_STAT_WAVE_MAX(9, best.cnt);
#endif
    KNN_TRACE(prob, i, "[fast] i=%d use_seed=%d capped=%d cap2=%g best=(%g,%d) seed=(%g,%d) resolved=%d gr=%g r_next=%d\n", i, use_seed, (int)capped, (double)cap2, (double)best.d2, best.slot, (double)seed.d2, seed.slot, (int)resolved, (double)gr, r_next);
    T lb_override = (T)-1;
    if (capped && best.slot < 0) {
        // nothing within the cap: that is not "no neighbour", only "farther than the cap"
        if (resolved) lb_override = cap2;              // every cell within sqrt(cap2) was examined
        resolved = false;
        // the seed is an upper bound (and the next seed) only while it is itself within maxDist
        if (seed.slot >= 0 && seed.d2 <= ch.max_dist2) best = seed;
        else best.d2 = ch.max_dist2;
    }
    finish_query<T>(M, g, ch, resolved, gr, best, qx, qy, qz, ux, uy, uz, cx, cy, cz, prob, i, P.off + i, r_next, lb_override,
                    slot_io, d2_out, none_r, slow_count, slow_list, slow_lb, slow_ring, q_cap);
}

// The fast pass queues per (problem, XCD) segment; the medium and slow paths want one dense list.
// k_queue_offsets: exclusive scan of the segment counts (total -> counters[0], survivors counter cleared);
// k_queue_compact: every segment's entries are copied behind those of the segments before it.
__global__ __launch_bounds__(1024) void k_queue_offsets(const int *__restrict__ seg_count, int nseg, int *__restrict__ seg_start,
                                                         int *__restrict__ counters)
{
    __shared__ int lds[32];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nseg; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nseg ? seg_count[kQueueCounterStride * i] : 0;
        int total;
        const int ex = block_exclusive_scan_1024(v, lds, total);
        if (i < nseg) seg_start[i] = ex + carry;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { counters[0] = carry; counters[1] = 0; counters[2] = 0; counters[3] = 0; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_queue_compact(const int *__restrict__ seg_count, const int *__restrict__ seg_start, int q_cap,
                                                        const int2 *__restrict__ seg_list, const T *__restrict__ seg_lb,
                                                        const int *__restrict__ seg_ring, int2 *__restrict__ list,
                                                        T *__restrict__ lb, int *__restrict__ ring, const int *__restrict__ active)
{
    // (segments along x: a batch may have more than 8191 problems; only the launched problems' segments hold anything)
    const int seg = active[blockIdx.x >> 3] * 8 + (blockIdx.x & 7);
    const int cnt = seg_count[kQueueCounterStride * seg];
    const int j = blockIdx.y * blockDim.x + threadIdx.x;
    if (j >= cnt) return;
    const long long src = (long long)seg * q_cap + j;
    const int dst = seg_start[seg] + j;
    list[dst] = seg_list[src];
    lb[dst] = seg_lb[src];
    ring[dst] = seg_ring[src];
}

// Medium path: the queued queries that can still matter (lower bound within the threshold just
// selected, or existence unknown) continue their ring search one query per LANE -- but in waves made
// only of such queries, so easy queries no longer wait for hard ones.  A lane stops when it is
// resolved, when its guaranteed radius passes 1.1x the threshold, or after `med_rings` rings (the
// wave-cooperative slow path takes what is left).  Entries it finishes are marked with LB = +inf.

constexpr int kMedShortQueue = 65536;

template <typename T>
__global__ __launch_bounds__(64) void k_knn_med(ProblemDev *__restrict__ probs, const MapDev<T> *__restrict__ maps,
                                                 const T *__restrict__ rd, int *__restrict__ slot_io, T *__restrict__ d2_out,
                                                 ChainDev<T> ch, int *__restrict__ slow_count,
                                                 const int2 *__restrict__ slow_list, T *__restrict__ slow_lb,
                                                 int *__restrict__ slow_ring, int *__restrict__ slow2_idx, int med_rings,
                                                 int use_seed, T *__restrict__ none_r)
{
    const int count = *slow_count;
    const int lane = threadIdx.x;
    // A short queue (one scan, not a batch) cannot hide a lane's serial ring walk behind other waves: beyond
    // two rings the wave-per-query path is the quicker one (single streamed scan: 238 us -> 30 us per pass).
    if (count < kMedShortQueue) med_rings = min(med_rings, 2);
    for (int base = blockIdx.x * 64; base < count; base += gridDim.x * 64) {
      bool survivor = false;                                  // still matters after this pass -> wave-cooperative path
      const int k = base + lane;
#ifdef PGICP_KNN_STATS
      const long long med_t0 = wall_clock64();
#endif
      if (k < count) do {
        const T lb = slow_lb[k];
        const int2 e = slow_list[k];
        ProblemDev &P = probs[e.x];
        const T limit = (T)P.limit;
        if (lb >= (T)0 && lb > limit) break;                  // cannot influence the result: stays lazy
        const int i = e.y;
        const T *q = rd + 3 * (P.off + i);
        T qx, qy, qz;
        apply_T<T>(P.Tcur, q[0], q[1], q[2], qx, qy, qz);
        const MapDev<T> M = maps[P.map];
        const T cap2 = (T)1.21 * limit;
        const bool capped = cap2 < ch.max_dist2;
        Best<T> best, seed;
        best.d2 = capped ? cap2 : ch.max_dist2; best.idx = 0x7FFFFFFF; best.slot = -1;
        seed.d2 = Bits<T>::inf(); seed.idx = 0x7FFFFFFF; seed.slot = -1;
        const int prev = slot_io[P.off + i];
        if (prev >= 0) {
            eval_point<T>(M.pts[prev], prev, qx, qy, qz, seed);
            if (seed.d2 <= best.d2) best = seed;
        }
        // The rings below slow_ring[k] were examined by the fast pass under ITS bound (the cap derived from the
        // previous threshold, or maxDist); rows beyond that bound were skipped.  If this pass has to look
        // farther than that -- the threshold grew, or only a far seed is known -- it starts over from ring 0.
        const T fast_cap2 = use_seed ? (T)(1.21 * P.prev_limit) : Bits<T>::inf();
        const T fast_bound = fast_cap2 < ch.max_dist2 ? fast_cap2 : ch.max_dist2;
        const int r_begin = best.d2 > fast_bound ? 0 : slow_ring[k];
        T gr;
        int r_next;
        bool resolved = grid_nn<T>(M, qx, qy, qz, ch.max_dist, r_begin, med_rings, cap2, best, gr, r_next);
#ifdef PGICP_KNN_STATS
        atomicAdd(&g_knn_stats[44], 1ULL);
        if (lb < (T)0) atomicAdd(&g_knn_stats[45], 1ULL);
        if (resolved) atomicAdd(&g_knn_stats[46], 1ULL);
        atomicAdd(&g_knn_stats[47], (unsigned long long)best.cnt);
#endif
        KNN_TRACE(e.x, i, "[med] i=%d lb=%g limit=%g cap2=%g ring=%d -> resolved=%d best=(%g,%d) gr=%g r_next=%d prev=%d\n", i, (double)lb, (double)limit, (double)cap2, r_begin, (int)resolved, (double)best.d2, best.slot, (double)gr, r_next, prev);
        if (capped && best.slot < 0) {
            // nothing within 1.1x the threshold: irrelevant for the filter, keep it queued beyond reach
            const bool seed_ok = seed.slot >= 0 && seed.d2 <= ch.max_dist2;
            if (resolved && (lb >= (T)0 || seed_ok)) { slow_lb[k] = cap2; slow_ring[k] = r_next; break; }
            resolved = false;
            if (seed_ok) best = seed;
        }
        if (resolved) {
            if (best.slot < 0) { best.d2 = Bits<T>::inf(); none_r[P.off + i] = (T)0; }
            slot_io[P.off + i] = best.slot;
            d2_out[P.off + i] = best.d2;
            slow_lb[k] = Bits<T>::inf();                      // finished: later passes skip it
            if (P.n_refined == 0) P.n_refined = 1;            // a flag; stored only while it reads 0: thousands of
                                                              // stores to one address serialise at the memory side
        } else {
            // still open: keep the better upper bound (if a real candidate exists) and the larger lower bound
            if (best.slot >= 0) { slot_io[P.off + i] = best.slot; d2_out[P.off + i] = best.d2; if (P.n_refined == 0) P.n_refined = 1; }
            T nlb = lb;
            const T reached = gr > (T)0 ? gr * gr : (T)0;
            if (lb >= (T)0 || best.slot >= 0) { nlb = reached; slow_lb[k] = nlb; }
            else slow_lb[k] = -((T)1 + reached);              // existence still unknown; -(1 + proven lower bound)
            slow_ring[k] = r_next;
            survivor = nlb < (T)0 || !(nlb > limit);
        }
      } while (false);
#ifdef PGICP_KNN_STATS
      if (lane == 0) {
          const unsigned long long dt = (unsigned long long)(wall_clock64() - med_t0);
          atomicMax(&g_knn_stats[48], dt); atomicAdd(&g_knn_stats[49], dt); atomicAdd(&g_knn_stats[50], 1ULL);
      }
#endif
      // survivors of this wave reserve consecutive slots of the second-stage list
      const unsigned long long m = __ballot(survivor);
      if (m) {
          int b0 = 0;
          if (lane == 0) b0 = atomicAdd(slow_count + 3, __popcll(m));
          b0 = __shfl(b0, 0, 64);
          if (survivor) slow2_idx[b0 + __popcll(m & ((1ULL << lane) - 1ULL))] = k;
      }
    }
}

// Slow path: one WAVE per queued query (no neighbour proven within kFastRings
// cells).  Rings of 8x8x8 SUPER-cells are walked around the query: the lanes
// first test one super-cell of the shell each (occupancy count, box distance
// against the wave's bound), then every surviving super-cell is scanned by the
// whole wave, one lane per (y,z) row of it.  Empty space costs one table look-up
// per super-cell, so proving "nothing within maxDist" stays cheap.  A
// lexicographic (d2, index) wave reduction picks the winner.
#ifndef PGICP_SLOW_PRIVATE
#define PGICP_SLOW_PRIVATE 4
#endif
#ifndef PGICP_SLOW_GROUP
#define PGICP_SLOW_GROUP 1
#endif
constexpr int kSlowPrivate = PGICP_SLOW_PRIVATE;   // rows of at most this many points are scanned by their own lane
constexpr int kSlowGroup = PGICP_SLOW_GROUP;       // surviving super-cells looked up together

template <typename T>
__global__ __launch_bounds__(256) void k_knn_slow(ProblemDev *__restrict__ probs, const MapDev<T> *__restrict__ maps,
                                                   const T *__restrict__ rd, int *__restrict__ slot_io,
                                                   T *__restrict__ d2_out, ChainDev<T> ch,
                                                   const int *__restrict__ slow_count, const int2 *__restrict__ slow_list,
                                                   const T *__restrict__ slow_lb, const int *__restrict__ slow2_idx,
                                                   int exact_all, T *__restrict__ none_r)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    // lazy mode walks the list k_knn_med left behind (entries that still matter); exact_all (public
    // matcher output) walks every queued query
    const int count = exact_all ? slow_count[0] : slow_count[3];
    for (int t = wave; t < count; t += nwaves) {
        const int k = exact_all ? t : slow2_idx[t];
        const int2 e = slow_list[k];
        ProblemDev &P = probs[e.x];
        const int i = e.y;
        // (a plain flag store: contended atomics on one address serialise the waves, and on gfx9 every
        // later load of the wave waits behind them)
        if (!exact_all && lane == 0) {
            if (P.n_refined == 0) P.n_refined = 1;
#ifdef PGICP_KNN_STATS
            atomicAdd(const_cast<int *>(slow_count) + (slow_lb[k] < (T)0 ? 1 : 2), 1);     // diagnostics: forced / bound-hit
#endif
        }
        const T *q = rd + 3 * (P.off + i);
        T qx, qy, qz;
        apply_T<T>(P.Tcur, q[0], q[1], q[2], qx, qy, qz);
        const MapDev<T> M = maps[P.map];
        const GridDesc<T> g = M.g;
        const T H = g.h * (T)8;
        // The search prunes with a radius 15 % beyond maxDist (candidates out there are found but not accepted):
        // when the answer is "none" it then knows how far the nearest point really is, or that nothing lies
        // within that larger radius -- the empty radius the fast pass subtracts the query's movement from in
        // later iterations instead of searching again.
        const T prune_dist = ch.max_dist * (T)1.15;
        const T prune2 = prune_dist * prune_dist;
        Best<T> best;
        best.d2 = prune2; best.idx = 0x7FFFFFFF; best.slot = -1;
        const int prev = slot_io[P.off + i];
        if (prev >= 0) eval_point<T>(M.pts[prev], prev, qx, qy, qz, best);
        // An entry whose neighbour -- if it has one -- is already proven farther than the threshold only
        // needs the answer to "is there any point within maxDist" (n_finite): the first candidate ends it,
        // and its distance stays in d2 as an upper bound like any other unresolved entry's.
        const T lbk = slow_lb[k];
        const bool exist_only = !exact_all && lbk < (T)0 && (-lbk - (T)1) > (T)P.limit;
        if (lane == 0) KNN_TRACE(e.x, i, "[slow] i=%d lbk=%g limit=%g exist_only=%d prev=%d\n", i, (double)lbk, (double)P.limit, (int)exist_only, prev);
#ifdef PGICP_KNN_STATS
        const long long t_begin = wall_clock64();
        int st_sc = 0, st_rows = 0, st_trips = 0, st_R = 0;
#endif
        bool found = exist_only && prev >= 0 && best.slot >= 0 && best.d2 <= ch.max_dist2;
        const T ux = qx - g.ox, uy = qy - g.oy, uz = qz - g.oz;
        const int Cx = clamp_cell<T>(ux, g.inv_h, g.nx) >> 3, Cy = clamp_cell<T>(uy, g.inv_h, g.ny) >> 3,
                  Cz = clamp_cell<T>(uz, g.inv_h, g.nz) >> 3;
        T empty_r = (T)0;                                     // radius proven free of points when the answer is "none"
        for (int R = 0; !found; ++R) {
            const int side = 2 * R + 1, total = side * side * side;
            for (int base = 0; base < total && !found; base += 64) {
                const int t = base + lane;
                const int dx = t % side - R, dy = (t / side) % side - R, dz = t / (side * side) - R;
                const int X = Cx + dx, Y = Cy + dy, Z = Cz + dz;
                bool need = t < total && (abs(dx) == R || abs(dy) == R || abs(dz) == R) && X >= 0 && X < M.nsx &&
                            Y >= 0 && Y < M.nsy && Z >= 0 && Z < M.nsz;
                if (need) need = M.sc_count[X + M.nsx * (Y + M.nsy * Z)] > 0;
                if (need) {
                    const T bx = fmax(slab_dist(ux, X, H) - g.margin, (T)0), by = fmax(slab_dist(uy, Y, H) - g.margin, (T)0),
                            bz = fmax(slab_dist(uz, Z, H) - g.margin, (T)0);
                    need = !((bx * bx + by * by) + bz * bz > best.d2);
                }
                unsigned long long mask = __ballot(need);
                while (mask) {
                    // lane r looks up row r of the super-cell (bound, x-range on the fine table, point range);
                    // kSlowGroup super-cells per trip (looking two up together measured no gain)
                    int srcs[kSlowGroup];
#pragma unroll
                    for (int u = 0; u < kSlowGroup; ++u) {
                        srcs[u] = -1;
                        if (mask) { srcs[u] = __ffsll((long long)mask) - 1; mask &= mask - 1; }
                    }
                    int ra[kSlowGroup], rb[kSlowGroup];
                    T rlb[kSlowGroup];
#pragma unroll
                    for (int u = 0; u < kSlowGroup; ++u) { ra[u] = 0; rb[u] = 0; rlb[u] = (T)0; }
#pragma unroll
                    for (int u = 0; u < kSlowGroup; ++u) {
                        if (srcs[u] < 0) continue;                         // wave-uniform
                        const int SX = __shfl(X, srcs[u], 64), SY = __shfl(Y, srcs[u], 64), SZ = __shfl(Z, srcs[u], 64);
                        const int y = 8 * SY + (lane & 7), z = 8 * SZ + (lane >> 3);
                        if (y < g.ny && z < g.nz) {
                            const T ly = fmax(slab_dist(uy, y, g.h) - g.margin, (T)0);
                            const T lz = fmax(slab_dist(uz, z, g.h) - g.margin, (T)0);
                            const T lb2 = ly * ly + lz * lz;
                            if (!(lb2 > best.d2)) {
                                const int kx = M.kx, nxf = g.nx * kx;
                                int fa = 8 * SX * kx, fb = min(8 * SX + 7, g.nx - 1) * kx + kx - 1;
                                if (best.d2 < Bits<T>::inf()) {
                                    const T rad = sqrt(fmax(best.d2 - lb2, (T)0)) + g.margin;
                                    const T inv_hx = g.inv_h * (T)kx;
                                    fa = max(fa, clamp_cell<T>(ux - rad, inv_hx, nxf));
                                    fb = min(fb, clamp_cell<T>(ux + rad, inv_hx, nxf));
                                }
                                if (fa <= fb) {
                                    const int row = nxf * (y + g.ny * z);
                                    ra[u] = as_global(M.cell_start_f)[row + fa];
                                    rb[u] = as_global(M.cell_start_f)[row + fb + 1];
                                    rlb[u] = lb2;
                                }
                            }
                        }
                    }
#ifdef PGICP_KNN_STATS
                    for (int u = 0; u < kSlowGroup; ++u) { st_sc += srcs[u] >= 0 ? 1 : 0; st_rows += __popcll(__ballot(ra[u] < rb[u])); }
#endif
                    // short rows (the sparse fringe of a map: a few points per row) are read by their own lane,
                    // all lanes at once; ...
                    if constexpr (kSlowPrivate > 0) {
#pragma unroll
                        for (int u = 0; u < kSlowGroup; ++u) {
                            const int len = rb[u] - ra[u];
                            if (len > 0 && len <= kSlowPrivate) {
                                typename Vec4<T>::type v[kSlowPrivate > 0 ? kSlowPrivate : 1];
#pragma unroll
                                for (int w = 0; w < kSlowPrivate; ++w) v[w] = load_rec<T>(M.pts, ra[u] + min(w, len - 1));
#pragma unroll
                                for (int w = 0; w < kSlowPrivate; ++w)
                                    if (w < len) eval_point<T>(v[w], ra[u] + w, qx, qy, qz, best);
                            }
                        }
                        T wmin = best.d2;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) wmin = fmin(wmin, __shfl_xor(wmin, o, 64));
                        if (wmin < best.d2) { best.d2 = wmin; best.idx = 0x7FFFFFFF; best.slot = -1; }
                    }
                    // ... long rows by the WHOLE wave, 64 consecutive points per trip: the rows of a dense
                    // super-cell hold thousands of points in a few rows (ground near the sensor), which one
                    // lane per row would walk serially
#pragma unroll
                    for (int u = 0; u < kSlowGroup; ++u) {
                        unsigned long long rows = __ballot(rb[u] - ra[u] > kSlowPrivate);
                        while (rows) {
                            const int r = __ffsll((long long)rows) - 1;
                            rows &= rows - 1;
                            const int pa = __shfl(ra[u], r, 64), pb = __shfl(rb[u], r, 64);
                            const T l2 = __shfl(rlb[u], r, 64);
                            if (l2 > best.d2) continue;                     // best.d2 is wave-uniform here
                            for (int q = pa + lane; q < pb; q += 64) eval_point<T>(load_rec<T>(M.pts, q), q, qx, qy, qz, best);
#ifdef PGICP_KNN_STATS
                            st_trips += (pb - pa + 63) / 64;
#endif
                            // share the tightest bound (pruning only; the winner is reduced at the end)
                            T wmin = best.d2;
#pragma unroll
                            for (int o = 32; o > 0; o >>= 1) wmin = fmin(wmin, __shfl_xor(wmin, o, 64));
                            if (wmin < best.d2) { best.d2 = wmin; best.idx = 0x7FFFFFFF; best.slot = -1; }
                        }
                    }
                    if (exist_only && best.d2 <= ch.max_dist2) { found = true; break; }   // wave-uniform
                }
            }
            // every super-cell outside ring R is at least this far away (wave-uniform)
            T gr = Bits<T>::inf();
            if (Cx - R - 1 >= 0) gr = fmin(gr, slab_dist(ux, Cx - R - 1, H));
            if (Cx + R + 1 <= M.nsx - 1) gr = fmin(gr, slab_dist(ux, Cx + R + 1, H));
            if (Cy - R - 1 >= 0) gr = fmin(gr, slab_dist(uy, Cy - R - 1, H));
            if (Cy + R + 1 <= M.nsy - 1) gr = fmin(gr, slab_dist(uy, Cy + R + 1, H));
            if (Cz - R - 1 >= 0) gr = fmin(gr, slab_dist(uz, Cz - R - 1, H));
            if (Cz + R + 1 <= M.nsz - 1) gr = fmin(gr, slab_dist(uz, Cz + R + 1, H));
            if (!(gr < Bits<T>::inf())) { empty_r = gr; break; }           // the whole grid was covered
            gr = gr - g.margin;
            empty_r = gr;
            T wbest = best.d2;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) wbest = fmin(wbest, __shfl_xor(wbest, o, 64));
            if (gr > (T)0 && (wbest < gr * gr || gr > prune_dist)) break;
        }
        // lexicographic (d2, idx) minimum over the wave; lanes whose bound was only borrowed hold slot -1
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const T od = __shfl_xor(best.d2, o, 64);
            const int oi = __shfl_xor(best.idx, o, 64);
            const int os = __shfl_xor(best.slot, o, 64);
            if (od < best.d2 || (od == best.d2 && oi < best.idx)) { best.d2 = od; best.idx = oi; best.slot = os; }
        }
        if (lane == 0) {
            if (best.slot >= 0 && best.d2 > ch.max_dist2) {
                // the nearest point lies beyond maxDist: no neighbour, and nothing closer than that point
                none_r[P.off + i] = sqrt(best.d2) * (T)0.9999;
                best.slot = -1;
            } else if (best.slot < 0) {
                none_r[P.off + i] = fmin(empty_r, prune_dist);
            }
            if (best.slot < 0) best.d2 = Bits<T>::inf();
            slot_io[P.off + i] = best.slot;
            d2_out[P.off + i] = best.d2;
#ifdef PGICP_KNN_STATS
            const unsigned long long dt = (unsigned long long)(wall_clock64() - t_begin);   // 100 MHz ticks
            atomicAdd(&g_knn_stats[10], 1ULL);
            atomicAdd(&g_knn_stats[11], dt);
            const unsigned long long old = atomicMax(&g_knn_stats[12], dt);
            if (dt > old) { g_knn_stats[13] = ((unsigned long long)st_sc << 40) | ((unsigned long long)st_rows << 20) | (unsigned long long)st_trips;
                            g_knn_stats[14] = ((unsigned long long)(exist_only ? 1 : 0) << 32) | (unsigned long long)(best.slot >= 0 ? 1 : 0); }
            atomicAdd(&g_knn_stats[15], (unsigned long long)st_trips);
#endif
        }
    }
}

// ---------------------------------------------------------------------------
// SurfaceNormalDataPointsFilter ([EXT] libpointmatcher): per point
// of a cloud its knn nearest neighbours IN the cloud, the scatter matrix of those neighbours and the
// eigenvector of its smallest eigenvalue.  The cloud is indexed like any map; lane i owns the point
// in slot i, so a wave is 64 points of one run of cells and shares its neighbourhood in L1.
// The K-entry candidate list lives in registers (every loop over it is unrolled, no dynamic index);
// K - knn dummy entries with distance -1 sit at its front and are never displaced, so the last entry
// is always the current knn-th best: the pruning bound.  Order = lexicographic (d2, original index),
// the order the oracle's k-d tree produces.
// ---------------------------------------------------------------------------
template <typename T, int K>
struct TopK {
    T d[K];
    int idx[K];
    int slot[K];
};

template <typename T, int K>
__device__ __forceinline__ void topk_offer(TopK<T, K> &L, T d, int idx, int slot)
{
    if (!(d < L.d[K - 1] || (d == L.d[K - 1] && idx < L.idx[K - 1]))) return;
    L.d[K - 1] = d; L.idx[K - 1] = idx; L.slot[K - 1] = slot;
#pragma unroll
    for (int p = K - 1; p > 0; --p) {
        const bool sw = L.d[p] < L.d[p - 1] || (L.d[p] == L.d[p - 1] && L.idx[p] < L.idx[p - 1]);
        const T td = L.d[p]; const int ti = L.idx[p], ts = L.slot[p];
        L.d[p] = sw ? L.d[p - 1] : td;       L.idx[p] = sw ? L.idx[p - 1] : ti;       L.slot[p] = sw ? L.slot[p - 1] : ts;
        L.d[p - 1] = sw ? td : L.d[p - 1];   L.idx[p - 1] = sw ? ti : L.idx[p - 1];   L.slot[p - 1] = sw ? ts : L.slot[p - 1];
    }
}

template <typename T, int K>
__device__ __forceinline__ void topk_row(const MapDev<T> &M, int row_base, int xa, int xb, T ux, T lb2, T qx, T qy, T qz,
                                         TopK<T, K> &L)
{
    const T bound = L.d[K - 1];
    if (bound < Bits<T>::inf()) {
        const T rad = sqrt(fmax(bound - lb2, (T)0)) + M.g.margin;
        xa = max(xa, clamp_cell<T>(ux - rad, M.g.inv_h, M.g.nx));
        xb = min(xb, clamp_cell<T>(ux + rad, M.g.inv_h, M.g.nx));
        if (xa > xb) return;
    }
    const auto *cs = as_global(M.cell_start);
    const int a = cs[row_base + xa], b = cs[row_base + xb + 1];
    for (int s = a; s < b; ++s) {
        const auto v = load_rec<T>(M.pts, s);
        const T dx = qx - v.x, dy = qy - v.y, dz = qz - v.z;
        topk_offer<T, K>(L, (dx * dx + dy * dy) + dz * dz, Bits<T>::unpack_idx(v.w), s);
    }
}

// cyclic Jacobi on a symmetric 3x3 in double -- the same sequence of operations as the oracle's jacobi3
__device__ __forceinline__ void jacobi3(double a[3][3], double v[3][3])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 16; sweep++) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const double arp = a[r][p], arq = a[r][q];
                    a[r][p] = c * arp - s * arq; a[r][q] = s * arp + c * arq;
                }
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const double apr = a[p][r], aqr = a[q][r];
                    a[p][r] = c * apr - s * aqr; a[q][r] = s * apr + c * aqr;
                }
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const double vrp = v[r][p], vrq = v[r][q];
                    v[r][p] = c * vrp - s * vrq; v[r][q] = s * vrp + c * vrq;
                }
            }
    }
}

template <typename T, int K>
__global__ __launch_bounds__(128) void k_surface_normals(const MapDev<T> *__restrict__ maps, int map, int knn, T max_dist,
                                                          T eps_rank, T *__restrict__ out_nrm, int out_stride,
                                                          T *__restrict__ out_eig, int *__restrict__ out_ids,
                                                          T *__restrict__ out_d2)
{
    const MapDev<T> M = maps[map];
    const GridDesc<T> g = M.g;
    const int s0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (s0 >= M.m) return;
    const auto me = load_rec<T>(M.pts, M.first + s0);
    const T qx = me.x, qy = me.y, qz = me.z;
    const int self = Bits<T>::unpack_idx(me.w);
    const T md2 = max_dist * max_dist;
    TopK<T, K> L;
#pragma unroll
    for (int j = 0; j < K; j++) {
        const bool real = j >= K - knn;
        L.d[j] = real ? md2 : (T)-1;              // anything beyond maxDist is useless
        L.idx[j] = real ? 0x7FFFFFFF : -1;
        L.slot[j] = -1;
    }
    const T ux = qx - g.ox, uy = qy - g.oy, uz = qz - g.oz;
    const int c0x = clamp_cell<T>(ux, g.inv_h, g.nx), c0y = clamp_cell<T>(uy, g.inv_h, g.ny), c0z = clamp_cell<T>(uz, g.inv_h, g.nz);
    for (int r = 0;; ++r) {
        const int z0 = max(c0z - r, 0), z1 = min(c0z + r, g.nz - 1);
        const int y0 = max(c0y - r, 0), y1 = min(c0y + r, g.ny - 1);
        const int xa = max(c0x - r, 0), xb = min(c0x + r, g.nx - 1);
        for (int z = z0; z <= z1; ++z) {
            const T lz = fmax(slab_dist(uz, z, g.h) - g.margin, (T)0);
            const bool zo = (z - c0z == r) || (c0z - z == r);
            for (int y = y0; y <= y1; ++y) {
                const T ly = fmax(slab_dist(uy, y, g.h) - g.margin, (T)0);
                const T lb2 = ly * ly + lz * lz;
                if (lb2 > L.d[K - 1]) continue;
                const int row = g.nx * (y + g.ny * z);
                if (zo || (y - c0y == r) || (c0y - y == r)) {
                    topk_row<T, K>(M, row, xa, xb, ux, lb2, qx, qy, qz, L);
                } else {                         // inner row of the shell: only its two end cells are new
                    if (c0x - r >= 0) topk_row<T, K>(M, row, c0x - r, c0x - r, ux, lb2, qx, qy, qz, L);
                    if (c0x + r <= g.nx - 1 && r > 0) topk_row<T, K>(M, row, c0x + r, c0x + r, ux, lb2, qx, qy, qz, L);
                }
            }
        }
        T gr = ring_guarantee<T>(g, ux, uy, uz, c0x, c0y, c0z, r);
        if (!(gr < Bits<T>::inf())) break;                         // grid exhausted
        gr = gr - g.margin;
        if (gr > (T)0 && (L.d[K - 1] < gr * gr || gr > max_dist)) break;
    }
    // ---- scatter matrix of the neighbours, in list order (the oracle's order), in T ----
    int cnt = 0;
    T sx = 0, sy = 0, sz = 0;
#pragma unroll
    for (int j = 0; j < K; j++)
        if (j >= K - knn && L.slot[j] >= 0) {
            const auto v = load_rec<T>(M.pts, L.slot[j]);
            sx += v.x; sy += v.y; sz += v.z; cnt++;
        }
    T c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    if (cnt > 0) {
        const T mx = sx / (T)cnt, my = sy / (T)cnt, mz = sz / (T)cnt;
#pragma unroll
        for (int j = 0; j < K; j++)
            if (j >= K - knn && L.slot[j] >= 0) {
                const auto v = load_rec<T>(M.pts, L.slot[j]);
                const T dx = v.x - mx, dy = v.y - my, dz = v.z - mz;
                c00 += dx * dx; c01 += dx * dy; c02 += dx * dz; c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;
            }
    }
    double A[3][3] = {{(double)c00, (double)c01, (double)c02}, {(double)c01, (double)c11, (double)c12}, {(double)c02, (double)c12, (double)c22}};
    double V[3][3];
    jacobi3(A, V);
    const double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
    int lo = 0, hi = 0;
    double elo = e0, ehi = e0;
    if (e1 < elo) { lo = 1; elo = e1; }
    if (e2 < elo) { lo = 2; elo = e2; }
    if (e1 > ehi) { hi = 1; ehi = e1; }
    if (e2 > ehi) { hi = 2; ehi = e2; }
    const int mid = 3 - lo - hi;
    const double emid = lo == hi ? e0 : (mid == 0 ? e0 : (mid == 1 ? e1 : e2));
    const bool degenerate = lo == hi || !(ehi > 0.0) || !(emid > 3.0 * (double)eps_rank * ehi);
    T nx = 0, ny = 1, nz = 0, w0 = 0, w1 = 0, w2 = 1;              // libpointmatcher's defaults for a rank < 2 scatter
    if (!degenerate) {
        nx = (T)(lo == 0 ? V[0][0] : (lo == 1 ? V[0][1] : V[0][2]));
        ny = (T)(lo == 0 ? V[1][0] : (lo == 1 ? V[1][1] : V[1][2]));
        nz = (T)(lo == 0 ? V[2][0] : (lo == 1 ? V[2][1] : V[2][2]));
        w0 = (T)elo; w1 = (T)emid; w2 = (T)ehi;
    }
    T *o = out_nrm + (long long)self * out_stride;
    o[0] = nx; o[1] = ny; o[2] = nz;
    if (out_eig) { out_eig[3LL * self] = w0; out_eig[3LL * self + 1] = w1; out_eig[3LL * self + 2] = w2; }
    if (out_ids || out_d2) {
#pragma unroll
        for (int j = 0; j < K; j++)
            if (j >= K - knn) {
                const int jj = j - (K - knn);
                const bool ok = L.slot[j] >= 0;
                if (out_ids) out_ids[(long long)self * knn + jj] = ok ? L.idx[j] : -1;
                if (out_d2) out_d2[(long long)self * knn + jj] = ok ? L.d[j] : Bits<T>::inf();
            }
    }
}

// Brute force (parity path): a block owns 256 queries; the map streams through
// LDS in tiles, every lane reads the same LDS address (broadcast, conflict free).
constexpr int kBruteTile = 1024;

template <typename T>
__global__ __launch_bounds__(kKnnBlock) void k_knn_brute(const ProblemDev *__restrict__ probs,
                                                          const MapDev<T> *__restrict__ maps, const T *__restrict__ rd_pre,
                                                          int *__restrict__ slot_out, T *__restrict__ d2_out,
                                                          ChainDev<T> ch, const int *__restrict__ active)
{
    using V4 = typename Vec4<T>::type;
    __shared__ V4 tile[kBruteTile];
    const ProblemDev &P = probs[active[blockIdx.y]];
    if (P.done) return;
    if (blockIdx.x * kKnnBlock >= P.n) return;
    const MapDev<T> M = maps[P.map];
    const int i = blockIdx.x * kKnnBlock + threadIdx.x;
    const bool live = i < P.n;
    T qx = 0, qy = 0, qz = 0;
    if (live) {
        const T *q = rd_pre + 3 * (P.off + i);
        apply_T<T>(P.Tcur, q[0], q[1], q[2], qx, qy, qz);
    }
    Best<T> best;
    best.d2 = ch.max_dist2; best.idx = 0x7FFFFFFF; best.slot = -1;
    for (int base = 0; base < M.m; base += kBruteTile) {
        const int cnt = min(kBruteTile, M.m - base);
        for (int k = threadIdx.x; k < cnt; k += kKnnBlock) tile[k] = M.pts[M.first + base + k];
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const V4 v = tile[k];
            const T dx = qx - v.x, dy = qy - v.y, dz = qz - v.z;
            const T d = (dx * dx + dy * dy) + dz * dz;
            const int idx = Bits<T>::unpack_idx(v.w);
            if (d < best.d2 || (d == best.d2 && idx < best.idx)) { best.d2 = d; best.idx = idx; best.slot = M.first + base + k; }
        }
        __syncthreads();
    }
    if (live) {
        if (best.slot < 0) best.d2 = Bits<T>::inf();
        slot_out[P.off + i] = best.slot;
        d2_out[P.off + i] = best.d2;
    }
}

// ---------------------------------------------------------------------------
// outlier filter: exact order statistic of the finite squared distances
// (radix select on the IEEE bit pattern, 11 bits per level; one block/problem)
// ---------------------------------------------------------------------------
template <typename T>
__device__ void trim_select_block(const T *__restrict__ d2, int n, T ratio, T &limit_out, int &nf_out)
{
    using U = typename Bits<T>::U;
    constexpr int KB = Bits<T>::kBits;
    __shared__ int hist[2048];
    __shared__ int lds_scan[32];
    __shared__ U s_prefix;
    __shared__ long long s_k;
    __shared__ int s_nf;
    U prefix = 0;
    int done_bits = 0;
    long long k = 0;
    const U inf_key = Bits<T>::key(Bits<T>::inf());
    while (done_bits < KB) {
        const int width = (KB - done_bits) >= 11 ? 11 : (KB - done_bits);
        const int shift = KB - done_bits - width;
        for (int b = threadIdx.x; b < 2048; b += blockDim.x) hist[b] = 0;
        __syncthreads();
        // 8 independent loads in flight per thread: with one load per iteration this loop is
        // latency bound (one block owns a whole problem, measured 68 us per selection)
        for (int base = threadIdx.x; base < n; base += 8 * blockDim.x) {
            U keys[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * blockDim.x;
                keys[u] = i < n ? Bits<T>::key(d2[i]) : inf_key;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const U key = keys[u];
                if (key >= inf_key) continue;                     // +inf (no neighbour) is not a value
                if (done_bits == 0 || (key >> (KB - done_bits)) == prefix)
                    atomicAdd(&hist[(int)((key >> shift) & (U)((1u << width) - 1u))], 1);
            }
        }
        __syncthreads();
        // block scan over 2048 bins: 2 bins per thread (blockDim = 1024)
        const int b0 = 2 * threadIdx.x;
        const int h0 = hist[b0], h1 = hist[b0 + 1];
        int total;
        const int ex = block_exclusive_scan_1024(h0 + h1, lds_scan, total);
        if (done_bits == 0) {
            if (threadIdx.x == 0) {
                s_nf = total;
                long long kk;
                if (ratio == (T)1) kk = (long long)total - 1;
                else {
                    kk = (long long)((T)total * ratio);           // `values.size() * quantile` evaluated in T
                    if (kk > (long long)total - 1) kk = (long long)total - 1;
                }
                s_k = kk < 0 ? 0 : kk;
            }
            __syncthreads();
            k = s_k;
            const int nf0 = s_nf;
            __syncthreads();                       // everyone has read s_k before the owner rewrites it
            if (nf0 == 0) { limit_out = Bits<T>::inf(); nf_out = 0; return; }
        }
        if (k >= ex && k < ex + h0) { s_prefix = (prefix << width) | (U)b0; s_k = k - ex; }
        else if (k >= ex + h0 && k < ex + h0 + h1) { s_prefix = (prefix << width) | (U)(b0 + 1); s_k = k - ex - h0; }
        __syncthreads();
        prefix = s_prefix;
        k = s_k;
        done_bits += width;
        __syncthreads();
    }
    limit_out = Bits<T>::val(prefix);
    nf_out = s_nf;
}

// ---------------------------------------------------------------------------
// The batched outlier filter spreads the selection over the chip (one block per problem took 45 us --
// three serial passes over the problem's distances -- and was half of a single scan's iteration):
//   k_sel_hist    every block histograms the top kSelBits key bits of its span of distances in LDS (plain
//                 LDS atomics) and adds the non-empty bins to the problem's table,
//   k_sel_filter  every block finds the bin holding the wanted rank (4096-bin scan, repeated per block: no
//                 cross-block hand-over inside a launch) and appends its distances of that bin -- a few per
//                 cent -- to the problem's compact key list, one global atomic per flush,
//   k_sel_final   one block per problem finishes the radix select on the compact list and clears the table.
// The result is the same exact order statistic as trim_select_block's.  Measured per selection: 17 us for one
// 100k scan, 28 us average for 128 of them.
// ---------------------------------------------------------------------------
constexpr int kSelBits = 12;
constexpr int kSelBins = 1 << kSelBits;
constexpr int kSelTile = 2048;                 // distances per block of the two wide passes
constexpr int kSelStride = kSelBins + 8;       // ints per problem: the table, then the compact list's cursor

template <typename T>
__device__ __forceinline__ long long select_rank(int total, T ratio)
{
    long long kk;
    if (ratio == (T)1) kk = (long long)total - 1;
    else {
        kk = (long long)((T)total * ratio);               // `values.size() * quantile` evaluated in T
        if (kk > (long long)total - 1) kk = (long long)total - 1;
    }
    return kk < 0 ? 0 : kk;
}

// bin of the wanted rank, the rank inside that bin and the number of finite values; NT threads, all return
// the same answer (total == 0: bin = -1)
template <typename T, int NT>
__device__ __forceinline__ void sel_pick_bin(const int *__restrict__ gh, T ratio, int *lds_scan, int *s_out /*3 ints*/,
                                             int &bin, int &krem, int &total)
{
    constexpr int PER = kSelBins / NT;
    int h[PER];
    int sum = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) { h[u] = gh[PER * threadIdx.x + u]; sum += h[u]; }
    const int ex = block_exclusive_scan_1024(sum, lds_scan, total);
    if (total > 0) {
        const long long k = select_rank<T>(total, ratio);
        if (k >= ex && k < (long long)ex + sum) {
            int acc = ex;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                if (k >= acc && k < (long long)acc + h[u]) { s_out[0] = PER * threadIdx.x + u; s_out[1] = (int)(k - acc); }
                acc += h[u];
            }
        }
    } else if (threadIdx.x == 0) { s_out[0] = -1; s_out[1] = 0; }
    __syncthreads();
    bin = s_out[0];
    krem = s_out[1];
    __syncthreads();
}

__device__ __forceinline__ bool sel_skip(const ProblemDev &P, int second)
{
    return P.done || (second && P.n_refined == 0);      // the re-selection only runs if the lazy path refined something
}

template <typename T>
__global__ __launch_bounds__(256) void k_sel_hist(const ProblemDev *__restrict__ probs, const T *__restrict__ d2, int second,
                                                   const int *__restrict__ active, int *__restrict__ tables, int span)
{
    using U = typename Bits<T>::U;
    constexpr int KB = Bits<T>::kBits;
    const int prob = active[blockIdx.y];
    const ProblemDev &P = probs[prob];
    if (sel_skip(P, second)) return;
    const int first = blockIdx.x * span;               // this block's distances: [first, first + span)
    if (first >= P.n) return;
    const int last = min(first + span, P.n);
    __shared__ int hist[kSelBins];
    for (int b = threadIdx.x; b < kSelBins; b += 256) hist[b] = 0;
    __syncthreads();
    const U inf_key = Bits<T>::key(Bits<T>::inf());
    for (int base = first; base < last; base += kSelTile) {
        U keys[kSelTile / 256];
#pragma unroll
        for (int u = 0; u < kSelTile / 256; ++u) {
            const int i = base + u * 256 + threadIdx.x;
            keys[u] = i < last ? Bits<T>::key(d2[P.off + i]) : inf_key;
        }
#pragma unroll
        for (int u = 0; u < kSelTile / 256; ++u) {
            const int bin = (int)(keys[u] >> (KB - kSelBits));
            // plain LDS atomics: aggregating the lanes of a wave per distinct bin first was measured 3.6x
            // SLOWER (37 us against 10 us per pass at 128 x 100k) -- the LDS unit takes same-address adds well
            if (keys[u] < inf_key) atomicAdd(&hist[bin], 1);      // +inf (no neighbour) is not a value
        }
    }
    __syncthreads();
    int *gh = tables + (long long)prob * kSelStride;
    for (int b = threadIdx.x; b < kSelBins; b += 256) {
        const int v = hist[b];
        if (v) atomicAdd(&gh[b], v);
    }
}

constexpr int kSelStage = 2 * kSelTile;

template <typename T>
__global__ __launch_bounds__(256) void k_sel_filter(const ProblemDev *__restrict__ probs, const T *__restrict__ d2,
                                                     ChainDev<T> ch, int second, const int *__restrict__ active,
                                                     int *__restrict__ tables, typename Bits<T>::U *__restrict__ keys_out,
                                                     int span)
{
    using U = typename Bits<T>::U;
    constexpr int KB = Bits<T>::kBits;
    const int prob = active[blockIdx.y];
    const ProblemDev &P = probs[prob];
    if (sel_skip(P, second)) return;
    const int first = blockIdx.x * span;
    if (first >= P.n) return;
    const int last = min(first + span, P.n);
    __shared__ int lds_scan[32];
    __shared__ int s_pick[3];
    __shared__ int s_cnt, s_base;
    __shared__ U stage[kSelStage];
    if (threadIdx.x == 0) s_cnt = 0;
    int *gh = tables + (long long)prob * kSelStride;
    // the first tile's loads are in flight while the bin is picked
    U keys[kSelTile / 256];
#pragma unroll
    for (int u = 0; u < kSelTile / 256; ++u) {
        const int i = first + u * 256 + threadIdx.x;
        keys[u] = i < last ? Bits<T>::key(d2[P.off + i]) : ~(U)0;
    }
    int bin, krem, total;
    sel_pick_bin<T, 256>(gh, ch.trim_ratio, lds_scan, s_pick, bin, krem, total);     // barriers inside
    if (total == 0) return;
    for (int base = first; base < last; base += kSelTile) {
        if (base != first) {
#pragma unroll
            for (int u = 0; u < kSelTile / 256; ++u) {
                const int i = base + u * 256 + threadIdx.x;
                keys[u] = i < last ? Bits<T>::key(d2[P.off + i]) : ~(U)0;
            }
        }
#pragma unroll
        for (int u = 0; u < kSelTile / 256; ++u) {
            // the padding key (all ones) is beyond +inf's bin; a few per cent of the lanes hit
            if ((int)(keys[u] >> (KB - kSelBits)) == bin) stage[atomicAdd(&s_cnt, 1)] = keys[u];
        }
        __syncthreads();
        const int cnt = s_cnt;
        // flush when another tile might not fit (and at the end)
        if (cnt > 0 && (cnt + kSelTile > kSelStage || base + kSelTile >= last)) {
            if (threadIdx.x == 0) s_base = atomicAdd(&gh[kSelBins], cnt);
            __syncthreads();                                // everyone has read s_cnt by now
            U *dst = keys_out + P.off + s_base;
            for (int j = threadIdx.x; j < cnt; j += 256) dst[j] = stage[j];
            if (threadIdx.x == 0) s_cnt = 0;
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(kSelectBlock) void k_sel_final(ProblemDev *__restrict__ probs, ChainDev<T> ch, int second,
                                                             const int *__restrict__ active, int *__restrict__ tables,
                                                             const typename Bits<T>::U *__restrict__ keys_in)
{
    using U = typename Bits<T>::U;
    constexpr int KB = Bits<T>::kBits;
    const int prob = active[blockIdx.x];
    ProblemDev &P = probs[prob];
    if (sel_skip(P, second)) return;
    __shared__ int hist[2048];
    __shared__ int lds_scan[32];
    __shared__ int s_pick[3];
    __shared__ U s_prefix;
    __shared__ int s_k;
    int *gh = tables + (long long)prob * kSelStride;
    int bin, krem, total;
    sel_pick_bin<T, kSelectBlock>(gh, ch.trim_ratio, lds_scan, s_pick, bin, krem, total);
    const int cnt = gh[kSelBins];
    __syncthreads();
    // leave the table clean for the next selection of this problem
    for (int b = threadIdx.x; b < kSelBins + 1; b += kSelectBlock) gh[b] = 0;
    T limit = Bits<T>::inf();
    if (total > 0) {
        const U *keys = keys_in + P.off;
        U prefix = (U)bin;
        int done_bits = kSelBits;
        int k = krem;
        while (done_bits < KB) {
            const int width = (KB - done_bits) >= 11 ? 11 : (KB - done_bits);
            const int shift = KB - done_bits - width;
            for (int b = threadIdx.x; b < 2048; b += kSelectBlock) hist[b] = 0;
            __syncthreads();
            for (int j = threadIdx.x; j < cnt; j += kSelectBlock) {
                const U key = keys[j];
                if ((key >> (KB - done_bits)) == prefix) atomicAdd(&hist[(int)((key >> shift) & (U)((1u << width) - 1u))], 1);
            }
            __syncthreads();
            const int b0 = 2 * threadIdx.x;
            const int h0 = hist[b0], h1 = hist[b0 + 1];
            int tot;
            const int ex = block_exclusive_scan_1024(h0 + h1, lds_scan, tot);
            if (k >= ex && k < ex + h0) { s_prefix = (prefix << width) | (U)b0; s_k = k - ex; }
            else if (k >= ex + h0 && k < ex + h0 + h1) { s_prefix = (prefix << width) | (U)(b0 + 1); s_k = k - ex - h0; }
            __syncthreads();
            prefix = s_prefix;
            k = s_k;
            done_bits += width;
            __syncthreads();
        }
        limit = Bits<T>::val(prefix);
    }
    if (threadIdx.x == 0) {
        if (second) P.n_refined = 0;
        else P.prev_limit = P.limit;              // the fast pass of this iteration capped its search with it
        P.limit = (double)limit;
        P.n_finite = total;
    }
}

// stand-alone outlier weights for the stage-level API (one problem)
template <typename T>
__global__ __launch_bounds__(kSelectBlock) void k_trim_select_raw(const T *__restrict__ d2, int n, T ratio,
                                                                   T *__restrict__ limit_nf /* [0]=limit, [1]=nf */)
{
    T limit;
    int nf;
    trim_select_block<T>(d2, n, ratio, limit, nf);
    if (threadIdx.x == 0) { limit_nf[0] = limit; limit_nf[1] = (T)nf; }
}

template <typename T>
__global__ __launch_bounds__(256) void k_weights(const T *__restrict__ d2, int n, const T *__restrict__ limit_nf,
                                                  T *__restrict__ w)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    w[i] = (d2[i] <= limit_nf[0]) ? (T)1 : (T)0;
}

// ---------------------------------------------------------------------------
// error minimiser: per-pair 6-DoF Jacobian, 30 sums, wave __shfl reduction
// ---------------------------------------------------------------------------
__device__ __forceinline__ void accumulate_pair(double *acc, double w, double px, double py, double pz, double qx,
                                                double qy, double qz, double nx, double ny, double nz)
{
    const double dx = px - qx, dy = py - qy, dz = pz - qz;
    const double e = (nx * dx + ny * dy) + nz * dz;
    double J[6];
    J[0] = py * nz - pz * ny;
    J[1] = pz * nx - px * nz;
    J[2] = px * ny - py * nx;
    J[3] = nx; J[4] = ny; J[5] = nz;
    int k = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) acc[k++] += w * (J[a] * J[b]);
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] -= w * (J[a] * e);
    acc[27] += w;
    acc[28] += 1.0;
    acc[29] += w * (e * e);
}

template <int NT>
__device__ __forceinline__ void block_reduce_store(double *acc, double *__restrict__ out)
{
    __shared__ double red[kReduceBlock / 64][NT];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NT; k++) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[wid][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NT) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kReduceBlock / 64; w++) s += red[w][threadIdx.x];
        out[threadIdx.x] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(kReduceBlock) void k_p2plane_reduce(const ProblemDev *__restrict__ probs,
                                                                  const MapDev<T> *__restrict__ maps,
                                                                  const T *__restrict__ rd_pre, const int *__restrict__ slot,
                                                                  const T *__restrict__ d2, double *__restrict__ partials,
                                                                  int max_blocks, const int *__restrict__ active)
{
    const int prob = active[blockIdx.y];
    const ProblemDev &P = probs[prob];
    if (P.done) return;
    const int tile = xcd_tile(blockIdx.x, gridDim.x);
    if (tile * kReduceSpan >= P.n) return;      // whole block past the end: partial stays unused
    const MapDev<T> M = maps[P.map];
    const T limit = (T)P.limit;
    double acc[kSys];
#pragma unroll
    for (int k = 0; k < kSys; k++) acc[k] = 0.0;
    // three load stages, each with kReduceItems independent loads in flight per lane:
    // (slot, d2) -> gathered (point, normal, reading) -> arithmetic
    using V4 = typename Vec4<T>::type;
    for (int rnd = 0; rnd < kReduceRounds; ++rnd) {
    const int base = tile * kReduceSpan + rnd * (kReduceBlock * kReduceItems);
    if (base >= P.n) break;
    int ss[kReduceItems];
    bool keep[kReduceItems];
#pragma unroll
    for (int it = 0; it < kReduceItems; it++) {
        const int i = base + it * kReduceBlock + threadIdx.x;
        ss[it] = -1;
        T dd = Bits<T>::inf();
        if (i < P.n) { dd = d2[P.off + i]; ss[it] = slot[P.off + i]; }
        keep[it] = ss[it] >= 0 && dd <= limit;
    }
    V4 mp[kReduceItems], mn[kReduceItems];
    T qv[kReduceItems][3];
#pragma unroll
    for (int it = 0; it < kReduceItems; it++) {
        const int i = base + it * kReduceBlock + threadIdx.x;
        const int s = keep[it] ? ss[it] : 0;
        mp[it] = M.pts[s];
        mn[it] = M.nrm[s];
        const T *q = rd_pre + 3 * (P.off + (keep[it] ? i : 0));
        qv[it][0] = q[0]; qv[it][1] = q[1]; qv[it][2] = q[2];
    }
#pragma unroll
    for (int it = 0; it < kReduceItems; it++) {
        if (keep[it]) {
            T px, py, pz;
            apply_T<T>(P.Tcur, qv[it][0], qv[it][1], qv[it][2], px, py, pz);
            accumulate_pair(acc, 1.0, (double)px, (double)py, (double)pz, (double)mp[it].x, (double)mp[it].y, (double)mp[it].z,
                            (double)mn[it].x, (double)mn[it].y, (double)mn[it].z);
        }
    }
    }
    block_reduce_store<kSys>(acc, partials + ((long long)prob * max_blocks + tile) * kSys);
}

// stage-level ErrorElements/residual with caller-provided ids (original
// indices) and weights; reading is already in the map frame.
template <typename T>
__global__ __launch_bounds__(kReduceBlock) void k_error_stats(const MapDev<T> *__restrict__ maps, int map,
                                                               const int *__restrict__ slot_of, const T *__restrict__ rd,
                                                               int stride, const int *__restrict__ ids,
                                                               const T *__restrict__ w, int n, T mx, T my, T mz,
                                                               double *__restrict__ partials)
{
    const MapDev<T> M = maps[map];
    double acc[kSys];
#pragma unroll
    for (int k = 0; k < kSys; k++) acc[k] = 0.0;
    for (int it = 0; it < kReduceItems * kReduceRounds; it++) {
        const int i = blockIdx.x * kReduceSpan + it * kReduceBlock + threadIdx.x;
        if (i < n) {
            const T wi = w[i];
            const int id = ids[i];
            if (id >= 0 && wi != (T)0) {
                // the map is stored centred; express the reading in the same frame
                const T px = rd[(long long)i * stride] - mx, py = rd[(long long)i * stride + 1] - my,
                        pz = rd[(long long)i * stride + 2] - mz;
                const int s = slot_of[id];
                const auto mp = M.pts[s];
                const auto mn = M.nrm[s];
                accumulate_pair(acc, (double)wi, (double)px, (double)py, (double)pz, (double)mp.x, (double)mp.y,
                                (double)mp.z, (double)mn.x, (double)mn.y, (double)mn.z);
            }
        }
    }
    block_reduce_store<kSys>(acc, partials + (long long)blockIdx.x * kSys);
}

// sums `nb` block partials of `nt` doubles each, per problem, in block order
__global__ __launch_bounds__(64) void k_sum_partials(const double *__restrict__ partials, int max_blocks, int nt,
                                                      const ProblemDev *__restrict__ probs /* or null */,
                                                      int nb_uniform, double *__restrict__ out)
{
    const int p = blockIdx.x;
    int nb = nb_uniform;
    if (probs) {
        nb = (probs[p].n + kReduceSpan - 1) / kReduceSpan;
        if (probs[p].status != PGICP_ST_OK) nb = 0;       // partials were never written
    }
    if ((int)threadIdx.x < nt) {
        double s = 0.0;
        const double *src = partials + (long long)p * max_blocks * nt + threadIdx.x;
        int b = 0;
        for (; b + 8 <= nb; b += 8) {                     // eight loads in flight, added in block order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long long)(b + u) * nt];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nb; b++) s += src[(long long)b * nt];
        out[(long long)p * nt + threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------
// solve + update + convergence check: one wave per problem, lane 0 does the
// (tiny, serial) double-precision algebra; no host round trip per iteration
// ---------------------------------------------------------------------------
// fixed-shape two-level sum of `nb` block partials (nt <= 32 doubles each): 8 groups
// of threads each add a strided subset of the blocks, then the 8 subtotals are
// added in group order.  Deterministic, and ~8x shorter than one serial chain.
__device__ __forceinline__ double sum_partials_256(const double *__restrict__ part, int nb, int nt, double (*lds)[32])
{
    const int v = threadIdx.x & 31, grp = threadIdx.x >> 5;      // 256 threads: 8 groups x 32 values
    double s = 0.0;
    if (v < nt)
        for (int b = grp; b < nb; b += 8) s += part[(long long)b * nt + v];
    lds[grp][v] = s;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x < 32) {
#pragma unroll
        for (int k = 0; k < 8; k++) tot += lds[k][threadIdx.x];
    }
    return tot;                                                   // valid for threadIdx.x < nt
}

template <typename T>
__global__ __launch_bounds__(256) void k_solve_update(ProblemDev *__restrict__ probs, const double *__restrict__ partials,
                                                       int max_blocks, ChainDev<T> ch, int *__restrict__ n_done,
                                                       const int *__restrict__ active)
{
    const int prob = active[blockIdx.x];
    ProblemDev &P = probs[prob];
    if (P.done) return;
    __shared__ double sys[kSys];
    __shared__ double red[8][32];
    const int nb = (P.n + kReduceSpan - 1) / kReduceSpan;
    const double tot = sum_partials_256(partials + (long long)prob * max_blocks * kSys, nb, kSys, red);
    if (threadIdx.x < kSys) {
        sys[threadIdx.x] = tot;
        P.sys[threadIdx.x] = tot;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    int status = PGICP_ST_OK;
    if (P.n_finite == 0 || !(sys[28] > 0.0)) status = PGICP_ST_NO_MATCH;
    if (status == PGICP_ST_OK) {
        double x[6], dT[16], Tn[16];
        P.rank = solve6(sys, ch.rank_rel_tol, x);
        delta_T(x, dT);
        for (int i = 0; i < 16; i++) { P.T_prev[i] = P.T_iter[i]; P.dT[i] = dT[i]; }
        mat4_mul(dT, P.T_iter, Tn);
        for (int i = 0; i < 16; i++) P.T_iter[i] = Tn[i];
        for (int i = 0; i < 12; i++) { P.Tcur_prev[i] = P.Tcur[i]; P.Tcur[i] = Tn[i]; }
        P.n_kept = (int)sys[28];
        P.iters += 1;
        const int f = checker_check(P.chk, Tn, ch.max_iters, ch.min_rot, ch.min_trans, ch.smooth);
        if (f & 8) status = PGICP_ST_NAN;
        else {
            if (f & 2) P.converged = 1;
            if (f & 4) P.max_iter_reached = 1;
            if (!(f & 1)) { P.done = 1; atomicAdd(n_done, 1); }
        }
    }
    if (status != PGICP_ST_OK) {
        P.status = status;
        P.done = 1;
        atomicAdd(n_done, 1);
    }
}

// active[] = ids of the problems still iterating, followed by the finished ones (so a stale, larger
// launch count only adds blocks that exit at once).  One block; P is at most a few thousand.
__global__ __launch_bounds__(1024) void k_compact_active(const ProblemDev *__restrict__ probs, int P, int *__restrict__ active,
                                                         int *host_flag, int stamp, int *__restrict__ queue_counters)
{
    // the matcher's queue counters are cleared here for the next iteration (their last values stay readable 8 ints on)
    if (queue_counters && threadIdx.x < 4) { queue_counters[8 + threadIdx.x] = queue_counters[threadIdx.x]; queue_counters[threadIdx.x] = 0; }
    __shared__ int lds[32];
    __shared__ int base_live, base_done;
    if (threadIdx.x == 0) { base_live = 0; base_done = 0; }
    __syncthreads();
    int n_live_total = 0;
    for (int b = 0; b < P; b += 1024) n_live_total += 0;     // (kept simple: two passes below)
    // pass 1: count live
    int live_cnt = 0;
    for (int p = threadIdx.x; p < P; p += 1024) live_cnt += probs[p].done ? 0 : 1;
    int tot;
    block_exclusive_scan_1024(live_cnt, lds, tot);
    const int n_live = tot;
    // pass 2: stable placement chunk by chunk
    for (int b = 0; b < P; b += 1024) {
        const int p = b + threadIdx.x;
        const int is_live = (p < P && !probs[p].done) ? 1 : 0;
        const int is_done = (p < P && probs[p].done) ? 1 : 0;
        int tl, td;
        const int el = block_exclusive_scan_1024(is_live, lds, tl);
        const int ed = block_exclusive_scan_1024(is_done, lds, td);
        if (is_live) active[base_live + el] = p;
        if (is_done) active[n_live + base_done + ed] = p;
        __syncthreads();
        if (threadIdx.x == 0) { base_live += tl; base_done += td; }
        __syncthreads();
    }
    (void)n_live_total;
    // the host polls this pair in pinned memory instead of waiting for a copy: {problems done, iteration stamp}
    if (host_flag && threadIdx.x == 0) {
        __hip_atomic_store(host_flag, P - n_live, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(host_flag + 1, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int knn_trace_set(int sorted_index)
{
#ifdef PGICP_KNN_STATS
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_i), &sorted_index, sizeof(int)) == hipSuccess ? 0 : -1;
#else
    (void)sorted_index;
    return -1;
#endif
}

int knn_stats_read(unsigned long long out[56], int reset)
{
#ifdef PGICP_KNN_STATS
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knn_stats), 56 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[56] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_knn_stats), z, sizeof z); }
    return 0;
#else
    (void)out; (void)reset;
    return -1;
#endif
}

void launch_compact_active(hipStream_t st, const ProblemDev *probs, int P, int *active, int *host_flag, int stamp,
                           int *queue_counters)
{
    hipLaunchKernelGGL(k_compact_active, dim3(1), dim3(1024), 0, st, probs, P, active, host_flag, stamp, queue_counters);
}

// ---------------------------------------------------------------------------
// Censi covariance sums over the last iteration's kept pairs:
// 21 terms of H (upper) + 21 terms of G (upper); the host inverts H.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kReduceBlock) void k_cov_reduce(const ProblemDev *__restrict__ probs,
                                                              const MapDev<T> *__restrict__ maps,
                                                              const T *__restrict__ rd_pre, const int *__restrict__ slot,
                                                              const T *__restrict__ d2, double *__restrict__ partials,
                                                              int max_blocks)
{
    const ProblemDev &P = probs[blockIdx.y];
    if (P.status != PGICP_ST_OK) return;
    if (blockIdx.x * kReduceSpan >= P.n) return;
    const MapDev<T> M = maps[P.map];
    const T limit = (T)P.limit;
    // small-angle parameters of the last increment
    const double beta = -asin(P.dT[8]);
    const double alpha = atan2(P.dT[9], P.dT[10]);
    const double gamma = atan2(P.dT[4] / cos(beta), P.dT[0] / cos(beta));
    const double t_x = P.dT[3], t_y = P.dT[7], t_z = P.dT[11];
    double Tp[12];
#pragma unroll
    for (int k = 0; k < 12; k++) Tp[k] = P.T_prev[k];
    double acc[kCovTerms];
#pragma unroll
    for (int k = 0; k < kCovTerms; k++) acc[k] = 0.0;
    for (int it = 0; it < kReduceItems * kReduceRounds; it++) {
        const int i = blockIdx.x * kReduceSpan + it * kReduceBlock + threadIdx.x;
        if (i >= P.n) continue;
        const T dd = d2[P.off + i];
        const int s = slot[P.off + i];
        if (s < 0 || !(dd <= limit)) continue;
        const T *q = rd_pre + 3 * (P.off + i);
        T pxt, pyt, pzt;
        apply_T<T>(Tp, q[0], q[1], q[2], pxt, pyt, pzt);
        const auto mp = M.pts[s];
        const auto mn = M.nrm[s];
        const double px = pxt, py = pyt, pz = pzt, qx = mp.x, qy = mp.y, qz = mp.z, nx = mn.x, ny = mn.y, nz = mn.z;
        const double rr = sqrt((px * px + py * py) + pz * pz);
        const double rdx = px / rr, rdy = py / rr, rdz = pz / rr;
        const double qr = sqrt((qx * qx + qy * qy) + qz * qz);
        const double qdx = qx / qr, qdy = qy / qr, qdz = qz / qr;
        const double n_alpha = nz * rdy - ny * rdz;
        const double n_beta = nx * rdz - nz * rdx;
        const double n_gamma = ny * rdx - nx * rdy;
        double E = nx * (px - gamma * py + beta * pz + t_x - qx);
        E += ny * (gamma * px + py - alpha * pz + t_y - qy);
        E += nz * (-beta * px + alpha * py + pz + t_z - qz);
        double Nr = nx * (rdx - gamma * rdy + beta * rdz);
        Nr += ny * (gamma * rdx + rdy - alpha * rdz);
        Nr += nz * (-beta * rdx + alpha * rdy + rdz);
        const double Nq = -((nx * qdx + ny * qdy) + nz * qdz);
        const double er = E + rr * Nr;
        const double h[6] = {nx, ny, nz, rr * n_alpha, rr * n_beta, rr * n_gamma};
        const double gr[6] = {nx * Nr, ny * Nr, nz * Nr, n_alpha * er, n_beta * er, n_gamma * er};
        const double gq[6] = {nx * Nq, ny * Nq, nz * Nq, qr * n_alpha * Nq, qr * n_beta * Nq, qr * n_gamma * Nq};
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) {
                acc[k] += h[a] * h[b];
                acc[21 + k] += gr[a] * gr[b] + gq[a] * gq[b];
                k++;
            }
    }
    block_reduce_store<kCovTerms>(acc, partials + ((long long)blockIdx.y * max_blocks + blockIdx.x) * kCovTerms);
}

// ---------------------------------------------------------------------------
// launchers (host side, called from pgicp_api.cpp)
// ---------------------------------------------------------------------------
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int round8(int x) { return (x + 7) & ~7; }

template <typename T>
void launch_centroid_bbox_batch(hipStream_t st, const BuildDesc<T> *descs, int n, int max_m, unsigned long long *stats)
{
    int nb = cdiv(max_m, 256 * 8);
    if (nb < 1) nb = 1;
    if (nb > 512) nb = 512;
    hipLaunchKernelGGL(k_centroid_bbox_b<T>, dim3(nb, n), dim3(256), 0, st, descs, stats);
}

// descs: device array; totals over the batch: points, cells (incl. one sentinel slot per cloud), super-cells
template <typename T>
void launch_grid_build_batch(hipStream_t st, const BuildDesc<T> *descs, int n, long long tot_m, long long tot_f, long long tot_s,
                             int max_m, int max_cells, int max_cells_f, int max_nsc, int max_blocks, int *cell_of, int *counts, int *block_sums,
                             int *cell_start, int *cell_start_f, int *cursor, int *order_tmp, typename Vec4<T>::type *pts, typename Vec4<T>::type *nrm_out,
                             int *slot_of, int *sc_count, int *near, int *sc_dist, int *sc_wit)
{
    (void)hipMemsetAsync(counts, 0, sizeof(int) * tot_f, st);
    (void)hipMemsetAsync(sc_count, 0, sizeof(int) * tot_s, st);
    const int nb = cdiv(tot_f, kScanChunk);
    (void)hipMemsetAsync(block_sums, 0, sizeof(int) * (size_t)nb * kChunkCopies, st);
    hipLaunchKernelGGL(k_cell_count_b<T>, dim3(cdiv(max_m, 256), n), dim3(256), 0, st, descs, cell_of, counts, sc_count, slot_of, block_sums, nb);
    hipLaunchKernelGGL(k_sum_copies, dim3(cdiv(nb, 256)), dim3(256), 0, st, block_sums, nb, kChunkCopies);
    hipLaunchKernelGGL(k_scan_sums_inplace, dim3(1), dim3(1024), 0, st, block_sums, nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(1024), 0, st, (const int *)counts, (int)tot_f, (const int *)block_sums,
                       cell_start_f, (int *)nullptr, 1);
    hipLaunchKernelGGL(k_scatter_idx, dim3(cdiv(tot_m, 256)), dim3(256), 0, st, (int)tot_m, (const int *)cell_of,
                       (const int *)cell_start_f, (const int *)slot_of, order_tmp);
    hipLaunchKernelGGL(k_rank_place_b<T>, dim3(cdiv(max_m, 256), n), dim3(256), 0, st, descs, (const int *)cell_of,
                       (const int *)cell_start_f, (const int *)order_tmp, pts, nrm_out, slot_of);
    if (cell_start != cell_start_f)
        hipLaunchKernelGGL(k_coarse_table_b<T>, dim3(cdiv(max_cells + 1, 256), n), dim3(256), 0, st, descs, (const int *)cell_start_f,
                           cell_start);
    for (int pass = -1; pass < 3; pass++)
        hipLaunchKernelGGL(k_near_b<T>, dim3(cdiv(max_blocks, 256), n), dim3(256), 0, st, descs, pass, (const int *)cell_start, counts,
                           cursor, near);
    if (max_nsc <= 4096) {
        hipLaunchKernelGGL(k_scdist_b<T>, dim3(n), dim3(1024), 0, st, descs, (const int *)sc_count, counts, cursor, sc_dist);
    } else {
        for (int pass = 0; pass < 3; pass++)
            hipLaunchKernelGGL(k_scdist_pass_b<T>, dim3(cdiv(max_nsc, 256), n), dim3(256), 0, st, descs, pass, (const int *)sc_count,
                               counts, cursor, sc_dist);
    }
    // (`counts` is scratch by now: the first point of every occupied super-cell, then the witness table)
    hipLaunchKernelGGL(k_sc_first_b<T>, dim3(cdiv(max_nsc, 256), n), dim3(256), 0, st, descs, (const int *)sc_count, (const int *)cell_start, counts);
    hipLaunchKernelGGL(k_sc_wit_b<T>, dim3(cdiv(max_nsc, 256), n), dim3(256), 0, st, descs, (const int *)sc_dist, (const int *)counts, sc_wit);
}

// once per scan: order every problem's pre-transformed reading by (map row, x)
template <typename T>
void launch_query_sort(hipStream_t st, const ProblemDev *probs, const MapDev<T> *maps, const T *rd_pre, T *rd_sorted,
                       int *qrow, unsigned long long *qtmp, int *order, int *counts, int *block_sums, int *qstart, int *cursor, int P,
                       int max_n, int max_rows, int bin_shift)
{
    const long long nbins = (long long)P * max_rows;
    (void)hipMemsetAsync(counts, 0, sizeof(int) * nbins, st);
    const dim3 grid(cdiv(max_n, 256), P);
    hipLaunchKernelGGL(k_qbin<T>, grid, dim3(256), 0, st, probs, maps, rd_pre, max_rows, bin_shift, qrow, counts, order);
    const int nb = cdiv(nbins, kScanChunk);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(1024), 0, st, (const int *)counts, (int)nbins, block_sums);
    hipLaunchKernelGGL(k_scan_sums_inplace, dim3(1), dim3(1024), 0, st, block_sums, nb);
    hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(1024), 0, st, (const int *)counts, (int)nbins, (const int *)block_sums,
                       qstart, (int *)nullptr, 0);
    // `order` carries the arrival positions until k_qrank overwrites it with the final permutation
    hipLaunchKernelGGL(k_qscatter, grid, dim3(256), 0, st, probs, max_rows, (const int *)qrow, (const int *)qstart, (const int *)order, qtmp);
    hipLaunchKernelGGL(k_qrank<T>, grid, dim3(256), 0, st, probs, max_rows, (const int *)qrow, (const int *)qstart,
                       (const unsigned long long *)qtmp, rd_pre, rd_sorted, order);
}

template <typename T>
void launch_transform(hipStream_t st, const T *in, int in_stride, T *out, int out_stride, int n, const double *T16,
                      int rotate_only)
{
    Mat34 M;
    for (int i = 0; i < 12; i++) M.v[i] = T16[i];
    if (n > 0)
        hipLaunchKernelGGL(k_transform<T>, dim3(cdiv(n, 256)), dim3(256), 0, st, in, in_stride, out, out_stride, n, M,
                           rotate_only);
}

template <typename T>
void launch_pretransform(hipStream_t st, const ProblemDev *probs, const SrcDesc *src, T *rd_pre, int P, int max_n)
{
    hipLaunchKernelGGL(k_pretransform<T>, dim3(cdiv(max_n, 256), P), dim3(256), 0, st, probs, src, rd_pre);
}

size_t knn_queue_bytes(int n_problems, int max_n, size_t elem)
{
    const size_t nseg = (size_t)n_problems * 8, q_cap = (size_t)(round8(cdiv(max_n, kFastBlock)) / 8) * kFastBlock;
    return nseg * kQueueCounterStride * sizeof(int) + nseg * sizeof(int) + nseg * q_cap * (sizeof(int2) + elem + sizeof(int)) + 1024;
}

template <typename T>
void launch_knn(hipStream_t st, int matcher, const ProblemDev *probs, const MapDev<T> *maps, const T *rd, int *slot,
                T *d2, const ChainDev<T> &ch, int P, int max_n, int use_seed, int *slow_count, int2 *slow_list, T *slow_lb,
                int *slow_ring, int fast_rings, const int *active, T *none_r, int n_problems, void *queue_buf)
{
    if (matcher == 1) {
        (void)hipMemsetAsync(slow_count, 0, 4 * sizeof(int), st);
        hipLaunchKernelGGL(k_knn_brute<T>, dim3(cdiv(max_n, kKnnBlock), P), dim3(kKnnBlock), 0, st, probs, maps, rd, slot, d2,
                           ch, active);
        return;
    }
    // segmented queue of this pass (see finish_query): [counters | starts | list | lb | ring]
    const int nseg = n_problems * 8, q_cap = (round8(cdiv(max_n, kFastBlock)) / 8) * kFastBlock;
    char *qb = (char *)queue_buf;
    int *seg_count = (int *)qb;                 qb += sizeof(int) * (size_t)nseg * kQueueCounterStride;
    int *seg_start = (int *)qb;                 qb += (sizeof(int) * (size_t)nseg + 255) & ~(size_t)255;
    int2 *seg_list = (int2 *)qb;                qb += sizeof(int2) * (size_t)nseg * q_cap;
    T *seg_lb = (T *)qb;                        qb += sizeof(T) * (size_t)nseg * q_cap;
    int *seg_ring = (int *)qb;
    (void)hipMemsetAsync(seg_count, 0, sizeof(int) * (size_t)nseg * kQueueCounterStride, st);
    // R = 1 in both cases: measured, a 5x5x5 collected block on the unseeded first iteration costs
    // 2.5x the ring-by-ring continuation (nothing prunes it until the own row has a hit)
    hipLaunchKernelGGL((k_knn_grid<T, 1>), dim3(round8(cdiv(max_n, kFastBlock)), P), dim3(kFastBlock), 0, st, probs, maps, rd, slot, d2, ch,
                       use_seed, fast_rings, seg_count, seg_list, seg_lb, seg_ring, active, none_r, q_cap);
    hipLaunchKernelGGL(k_queue_offsets, dim3(1), dim3(1024), 0, st, (const int *)seg_count, nseg, seg_start, slow_count);
    hipLaunchKernelGGL(k_queue_compact<T>, dim3(P * 8, cdiv(q_cap, 256)), dim3(256), 0, st, (const int *)seg_count, (const int *)seg_start,
                       q_cap, (const int2 *)seg_list, (const T *)seg_lb, (const int *)seg_ring, slow_list, slow_lb, slow_ring, active);
}

template <typename T>
void launch_knn_med(hipStream_t st, ProblemDev *probs, const MapDev<T> *maps, const T *rd, int *slot, T *d2,
                    const ChainDev<T> &ch, int *slow_count, const int2 *slow_list, T *slow_lb, int *slow_ring, int *slow2_idx,
                    int med_rings, int use_seed, T *none_r)
{
    hipLaunchKernelGGL(k_knn_med<T>, dim3(8192), dim3(64), 0, st, probs, maps, rd, slot, d2, ch, slow_count, slow_list, slow_lb,
                       slow_ring, slow2_idx, med_rings, use_seed, none_r);
}

// resolves the queries the fast path queued: all of them (exact_all, public matcher
// output) or only those that can still matter for the trimmed filter (lazy)
constexpr int kSlowBlocks = 2048;

template <typename T>
void launch_knn_slow(hipStream_t st, ProblemDev *probs, const MapDev<T> *maps, const T *rd, int *slot, T *d2,
                     const ChainDev<T> &ch, const int *slow_count, const int2 *slow_list, const T *slow_lb,
                     const int *slow2_idx, int exact_all, T *none_r)
{
    hipLaunchKernelGGL(k_knn_slow<T>, dim3(kSlowBlocks), dim3(256), 0, st, probs, maps, rd, slot, d2, ch, slow_count, slow_list,
                       slow_lb, slow2_idx, exact_all, none_r);
}

template <typename T>
void launch_trim_select(hipStream_t st, ProblemDev *probs, const T *d2, const ChainDev<T> &ch, int P, int max_n, int second,
                        const int *active, int *tables, void *keys)
{
    using U = typename Bits<T>::U;
    // about 2048 blocks in all: every block costs one table merge (global atomics on the problem's bins)
    const int tiles = cdiv(max_n, kSelTile);
    const int per_problem = std::min(tiles, std::max(1, 2048 / P));
    const int span = cdiv(tiles, per_problem) * kSelTile;
    const dim3 wide(cdiv(max_n, span), P);
    hipLaunchKernelGGL(k_sel_hist<T>, wide, dim3(256), 0, st, (const ProblemDev *)probs, d2, second, active, tables, span);
    hipLaunchKernelGGL(k_sel_filter<T>, wide, dim3(256), 0, st, (const ProblemDev *)probs, d2, ch, second, active, tables, (U *)keys, span);
    hipLaunchKernelGGL(k_sel_final<T>, dim3(P), dim3(kSelectBlock), 0, st, probs, ch, second, active, tables, (const U *)keys);
}

size_t trim_select_table_bytes(int P) { return sizeof(int) * (size_t)P * kSelStride; }

template <typename T>
int launch_surface_normals(hipStream_t st, const MapDev<T> *maps, int map, int m, int knn, T max_dist, T eps_rank, T *out_nrm,
                           int out_stride, T *out_eig, int *out_ids, T *out_d2)
{
    const dim3 grid(cdiv(m, 128)), block(128);
    if (knn <= 8) hipLaunchKernelGGL((k_surface_normals<T, 8>), grid, block, 0, st, maps, map, knn, max_dist, eps_rank, out_nrm, out_stride, out_eig, out_ids, out_d2);
    else if (knn <= 16) hipLaunchKernelGGL((k_surface_normals<T, 16>), grid, block, 0, st, maps, map, knn, max_dist, eps_rank, out_nrm, out_stride, out_eig, out_ids, out_d2);
    else if (knn <= 32) hipLaunchKernelGGL((k_surface_normals<T, 32>), grid, block, 0, st, maps, map, knn, max_dist, eps_rank, out_nrm, out_stride, out_eig, out_ids, out_d2);
    else return -1;
    return 0;
}

int reduce_blocks(int max_n) { return round8(cdiv(max_n, kReduceSpan)); }

template <typename T>
void launch_reduce(hipStream_t st, const ProblemDev *probs, const MapDev<T> *maps, const T *rd_pre, const int *slot,
                   const T *d2, double *partials, int P, int max_n, const int *active)
{
    const int nb = reduce_blocks(max_n);
    hipLaunchKernelGGL(k_p2plane_reduce<T>, dim3(nb, P), dim3(kReduceBlock), 0, st, probs, maps, rd_pre, slot, d2, partials,
                       nb, active);
}

template <typename T>
void launch_solve(hipStream_t st, ProblemDev *probs, const double *partials, const ChainDev<T> &ch, int *n_done, int P,
                  int max_n, const int *active)
{
    hipLaunchKernelGGL(k_solve_update<T>, dim3(P), dim3(256), 0, st, probs, partials, reduce_blocks(max_n), ch, n_done, active);
}

template <typename T>
void launch_cov(hipStream_t st, const ProblemDev *probs, const MapDev<T> *maps, const T *rd_pre, const int *slot,
                const T *d2, double *partials, double *out, int P, int max_n)
{
    const int nb = reduce_blocks(max_n);
    hipLaunchKernelGGL(k_cov_reduce<T>, dim3(nb, P), dim3(kReduceBlock), 0, st, probs, maps, rd_pre, slot, d2, partials, nb);
    hipLaunchKernelGGL(k_sum_partials, dim3(P), dim3(64), 0, st, (const double *)partials, nb, kCovTerms, probs, 0, out);
}

void launch_sum_partials(hipStream_t st, const double *partials, int max_blocks, int nt, const ProblemDev *probs,
                         int nb_uniform, double *out, int P)
{
    hipLaunchKernelGGL(k_sum_partials, dim3(P), dim3(64), 0, st, partials, max_blocks, nt, probs, nb_uniform, out);
}

template <typename T>
void launch_trim_raw(hipStream_t st, const T *d2, int n, T ratio, T *limit_nf, T *w)
{
    hipLaunchKernelGGL(k_trim_select_raw<T>, dim3(1), dim3(kSelectBlock), 0, st, d2, n, ratio, limit_nf);
    if (w) hipLaunchKernelGGL(k_weights<T>, dim3(cdiv(n, 256)), dim3(256), 0, st, d2, n, (const T *)limit_nf, w);
}

template <typename T>
void launch_error_stats(hipStream_t st, const MapDev<T> *maps, int map, const int *slot_of, const T *rd, int stride,
                        const int *ids, const T *w, int n, const T mean[3], double *partials, double *out)
{
    const int nb = cdiv(n, kReduceSpan);
    hipLaunchKernelGGL(k_error_stats<T>, dim3(nb), dim3(kReduceBlock), 0, st, maps, map, slot_of, rd, stride, ids, w, n,
                       mean[0], mean[1], mean[2], partials);
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, st, (const double *)partials, nb, kSys,
                       (const ProblemDev *)nullptr, nb, out);
}

template <typename T>
void launch_unpermute(hipStream_t st, const MapDev<T> *maps, int map, const int *order, const int *slot, const T *d2, int n,
                      int *ids_out, T *d2_out)
{
    hipLaunchKernelGGL(k_unpermute<T>, dim3(cdiv(n, 256)), dim3(256), 0, st, maps, map, order, slot, d2, n, ids_out, d2_out);
}

#define INSTANTIATE(T)                                                                                                   \
    template void launch_centroid_bbox_batch<T>(hipStream_t, const BuildDesc<T> *, int, int, unsigned long long *);       \
    template void launch_grid_build_batch<T>(hipStream_t, const BuildDesc<T> *, int, long long, long long, long long, int, \
                                             int, int, int, int, int *, int *, int *, int *, int *, int *, int *, typename Vec4<T>::type *,      \
                                             typename Vec4<T>::type *, int *, int *, int *, int *, int *);                \
    template void launch_query_sort<T>(hipStream_t, const ProblemDev *, const MapDev<T> *, const T *, T *, int *,         \
                                       unsigned long long *, int *, int *, int *, int *, int *, int, int, int, int);      \
    template void launch_transform<T>(hipStream_t, const T *, int, T *, int, int, const double *, int);                   \
    template void launch_pretransform<T>(hipStream_t, const ProblemDev *, const SrcDesc *, T *, int, int);                \
    template void launch_knn<T>(hipStream_t, int, const ProblemDev *, const MapDev<T> *, const T *, int *, T *,           \
                                const ChainDev<T> &, int, int, int, int *, int2 *, T *, int *, int, const int *, T *, int, void *); \
    template void launch_knn_med<T>(hipStream_t, ProblemDev *, const MapDev<T> *, const T *, int *, T *,                  \
                                    const ChainDev<T> &, int *, const int2 *, T *, int *, int *, int, int, T *);          \
    template void launch_knn_slow<T>(hipStream_t, ProblemDev *, const MapDev<T> *, const T *, int *, T *,                 \
                                     const ChainDev<T> &, const int *, const int2 *, const T *, const int *, int, T *);   \
    template void launch_trim_select<T>(hipStream_t, ProblemDev *, const T *, const ChainDev<T> &, int, int, int, const int *, int *, void *); \
    template void launch_reduce<T>(hipStream_t, const ProblemDev *, const MapDev<T> *, const T *, const int *, const T *, \
                                   double *, int, int, const int *);                                                      \
    template void launch_solve<T>(hipStream_t, ProblemDev *, const double *, const ChainDev<T> &, int *, int, int,        \
                                  const int *);                                                                           \
    template void launch_cov<T>(hipStream_t, const ProblemDev *, const MapDev<T> *, const T *, const int *, const T *,    \
                                double *, double *, int, int);                                                            \
    template void launch_trim_raw<T>(hipStream_t, const T *, int, T, T *, T *);                                           \
    template void launch_error_stats<T>(hipStream_t, const MapDev<T> *, int, const int *, const T *, int, const int *,    \
                                        const T *, int, const T[3], double *, double *);                                  \
    template int launch_surface_normals<T>(hipStream_t, const MapDev<T> *, int, int, int, T, T, T *, int, T *, int *, T *); \
    template void launch_unpermute<T>(hipStream_t, const MapDev<T> *, int, const int *, const int *, const T *, int,      \
                                      int *, T *);

INSTANTIATE(float)
INSTANTIATE(double)

}  // namespace pgicp
