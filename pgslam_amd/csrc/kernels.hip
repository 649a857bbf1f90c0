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
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#ifndef PGICP_FAST_BLOCK
#define PGICP_FAST_BLOCK 64
#endif
// The fast matcher kernel keeps ~100 scalars live; capping its SGPR allocation at 80 (the rest spill into
// lanes of one VGPR) admits 7 waves per SIMD instead of 6 (800 SGPRs per SIMD: MI355X_MICROARCH.md).
// ... and asking for 6 waves per SIMD keeps it within 80 VGPRs without a spill (left alone the allocator takes 91: 5 waves).
#ifndef PGICP_FAST_WAVES
#define PGICP_FAST_WAVES 6
#endif
#ifndef PGICP_FAST_ATTR
// (the double instantiation needs ~105 registers: 4 waves)
#ifndef PGICP_FAST_WAVES_D
#define PGICP_FAST_WAVES_D 4
#endif
#define PGICP_FAST_ATTR __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(sizeof(T) == 4 ? PGICP_FAST_WAVES : PGICP_FAST_WAVES_D, sizeof(T) == 4 ? PGICP_FAST_WAVES : PGICP_FAST_WAVES_D)))
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

template <typename T, typename S>
__device__ __forceinline__ void apply_T(const S *Tc, T x, T y, T z, T &ox, T &oy, T &oz)
{
    const T r00 = (T)Tc[0], r01 = (T)Tc[1], r02 = (T)Tc[2], tx = (T)Tc[3];
    const T r10 = (T)Tc[4], r11 = (T)Tc[5], r12 = (T)Tc[6], ty = (T)Tc[7];
    const T r20 = (T)Tc[8], r21 = (T)Tc[9], r22 = (T)Tc[10], tz = (T)Tc[11];
    ox = ((r00 * x + r01 * y) + r02 * z) + tx;
    oy = ((r10 * x + r11 * y) + r12 * z) + ty;
    oz = ((r20 * x + r21 * y) + r22 * z) + tz;
}

// The current / previous iteration transform of a problem in the kernel's arithmetic type (ProblemDev keeps a float copy)
template <typename T> __device__ __forceinline__ auto tcur_of(const ProblemDev &P)
{
    if constexpr (sizeof(T) == 4) return (const float *)P.Tcur_f; else return (const double *)P.Tcur;
}
template <typename T> __device__ __forceinline__ auto tcur_prev_of(const ProblemDev &P)
{
    if constexpr (sizeof(T) == 4) return (const float *)P.Tcur_prev_f; else return (const double *)P.Tcur_prev;
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
// One aligned 16/32-byte global load of a map record.  (A uniform base in SGPRs + a 32-bit byte offset per lane --
// `global_load_dwordx4 v, v_off, s[base:base+1]`, two VALU instructions fewer per load -- measured neutral, 8.53-8.63 against
// 8.57-8.64 ms per step, and it would cap the shared point array of a batched build at 2^32 bytes:
// tools/experiments/r03_e4_*.)
template <typename T, int K = 0>
__device__ __forceinline__ typename Vec4<T>::type load_rec(const typename Vec4<T>::type *p, int j)
{
    using R4 = typename Raw4<T>::type;
    const R4 r = as_global(reinterpret_cast<const R4 *>(p))[j + K];
    return make_v4(r.x, r.y, r.z, r.w);
}
// the same for the int tables (the cell tables of ONE map: fewer than 2^30 entries by construction)
__device__ __forceinline__ int load_tab(const int *p, int i)
{
    return as_global(p)[i];
}

// ---- the cell tables of a map, dense or succinct (MapDev::sw) ----------------------------------------------------------
// occupied fine cells before cell k (0 .. 63) of a group, batch-wide: the word's rank + the mask bits below k
__device__ __forceinline__ int succ_rank(const uint4 w, int k)
{
    const unsigned long long m = ((unsigned long long)w.y << 32) | w.x;
    return (int)w.z + __popcll(m & ((1ULL << k) - 1ULL));
}
__device__ __forceinline__ uint4 load_word(const uint4 *sw, int g)
{
    typedef unsigned U4 __attribute__((ext_vector_type(4)));
    const U4 r = as_global(reinterpret_cast<const U4 *>(sw))[g];
    return make_uint4(r.x, r.y, r.z, r.w);
}
// slot of the first point whose fine cell is >= f  (f = 0 .. ncells_f, the sentinel included)
template <bool SUCC, typename T>
__device__ __forceinline__ int tab_fine(const MapDev<T> &M, int f)
{
    if constexpr (SUCC) return load_tab(M.ostart, succ_rank(load_word(M.sw, f >> 6), f & 63));
    else return load_tab(M.cell_start_f, f);
}
// [a, b) = the points of fine cells [ia, ib): both table entries requested together; an EMPTY range costs the succinct
// table no second round trip (equal ranks) and comes back as a = b = 0
template <bool SUCC, typename T>
__device__ __forceinline__ void tab_range(const MapDev<T> &M, int ia, int ib, int &a, int &b)
{
    if constexpr (SUCC) {
        const uint4 wa = load_word(M.sw, ia >> 6), wb = load_word(M.sw, ib >> 6);
        const int ra = succ_rank(wa, ia & 63), rb = succ_rank(wb, ib & 63);
        a = 0; b = 0;
        if (ra != rb) { a = load_tab(M.ostart, ra); b = load_tab(M.ostart, rb); }
    } else { a = load_tab(M.cell_start_f, ia); b = load_tab(M.cell_start_f, ib); }
}
// the kernels that are not instantiated per table kind ask the map
template <typename T> __device__ __forceinline__ int tab_fine_any(const MapDev<T> &M, int f) { return M.sw ? tab_fine<true, T>(M, f) : tab_fine<false, T>(M, f); }
template <typename T> __device__ __forceinline__ void tab_range_any(const MapDev<T> &M, int ia, int ib, int &a, int &b)
{
    if (M.sw) tab_range<true, T>(M, ia, ib, a, b); else tab_range<false, T>(M, ia, ib, a, b);
}
// TK: 0 dense, 1 succinct (the kernel is instantiated per table kind), 2 ask the map
template <int TK, typename T> __device__ __forceinline__ void tab_range_k(const MapDev<T> &M, int ia, int ib, int &a, int &b)
{
    if constexpr (TK == 2) tab_range_any<T>(M, ia, ib, a, b); else tab_range<TK == 1, T>(M, ia, ib, a, b);
}
// the points of SEARCH cell c (coarse: kx fine cells): [tab_coarse(c), tab_coarse(c + 1))
template <typename T> __device__ __forceinline__ int tab_coarse_any(const MapDev<T> &M, int c) { return M.sw ? tab_fine<true, T>(M, c * M.kx) : load_tab(M.cell_start, c); }

// The per-PAIR arrays (slot, d2, the selection's keys): knn entries per reading point, [point][neighbour]
__device__ __forceinline__ int pairs_n(const ProblemDev &P) { return P.n * P.knn; }
__device__ __forceinline__ long long pairs_off(const ProblemDev &P) { return P.off * P.knn; }

// The kernels, by stage (all part of this translation unit):
#include "k_build.inc"
#include "k_sort.inc"
#include "k_mapbuild.inc"
#include "k_match.inc"
#include "k_normals.inc"
#include "k_select.inc"
#include "k_minimise.inc"
#include "k_filter.inc"
#include "k_launch.inc"

}  // namespace pgicp
