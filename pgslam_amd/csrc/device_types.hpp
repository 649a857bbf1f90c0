// device_types.hpp -- structures shared by the HIP kernels and the host side of
// libpgicp.  Everything that lives in HBM is described here.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "icp_math.hpp"

namespace pgicp {

template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

// Uniform grid over the (centred) reference cloud.  Cell id = x + nx*(y + ny*z),
// x fastest, so the cells of one x-row are contiguous in the cell-sorted point
// array and a whole row segment is one contiguous range.
template <typename T>
struct GridDesc {
    T ox, oy, oz;        // lower corner
    T h, inv_h;          // cell edge and its reciprocal
    T margin;            // conservative slack for rounding in cell assignment (0.02 h)
    int nx, ny, nz;
};

// A resident reference cloud ("map"): cell-sorted AoS records, 16/32 B each so
// a neighbour candidate is ONE aligned vector load.
template <typename T>
struct MapDev {
    const typename Vec4<T>::type *pts;   // (x, y, z, bit-cast original index) cell-sorted, centred
    const typename Vec4<T>::type *nrm;   // the minimiser's records, same order: entry 2 s = the point again, 2 s + 1 = (nx, ny, nz, 0) -- one
                                         // 32-byte gather per pair instead of two 16-byte ones from two arrays; may be null
    const int *cell_start;               // ncells + 1 exclusive prefix sums                       } the DENSE tables; both null when
    const int *cell_start_f;             // the same prefix sums on cells kx times finer in x      } the map carries the succinct one
                                         // (== cell_start when kx == 1)
    // The SUCCINCT fine table (round 6; a map carries the dense tables or this one).  A range scan fills a fraction of a per
    // cent of its fine cells (a 100 k-pt keyframe map: 2.8 M fine cells, 19 k of them occupied; the dense table is 11 MB a
    // map -- seven times the points -- and 5.7 GB written per 512-pair loop-closure batch).  Per 64 fine cells one 16-byte word
    // {occupancy mask (64 bits), rank = occupied fine cells of the BATCH before the group, 0}; per occupied fine cell the slot
    // of its first point (`ostart`, ranks run over the maps of one batched build like the slots do; one sentinel entry).
    //   start(f) = ostart[word[f >> 6].rank + popcount(word[f >> 6].mask & below(f & 63))]       (tab_fine, kernels.hip)
    // Two dependent loads where the dense table has one -- but 20x fewer bytes, an empty range is recognised from the words
    // alone, and the build writes 16 bytes where it wrote 256.
    const uint4 *sw;                     // the words of this map (group g of its fine cells = sw[g]); null: dense tables
    const int *ostart;                   // shared by the maps of one batched build (indexed by rank)
    int kx;                              // points of a cell are ordered by fine x cell, so fine ranges are contiguous too
    const int *sc_count;                 // occupancy flag per 8x8x8 super-cell
    const int *sc_dist;                  // Chebyshev distance (in super-cells, capped at kScReach + 1) to the nearest occupied one
    const int *sc_wit;                   // slot of a point in a nearest occupied super-cell (within kWitReach), else -1
    const float *sc_ext;                 // per super-cell: the box of its POINTS, (min x, y, z, max x, y, z) as offsets from the grid
                                         // origin, float, min > max when empty -- the slices of a street scene fill little of a 0.7 m cube
    const float *ptsf;                   // double maps only (else null): the points again as (float x, y, z, index bits) records of 16 bytes, same
                                         // order -- the fast matcher's prefilter reads these and the double record only of a candidate that may win
    const int *slot_of;                  // original index -> position in pts / nrm
    const int *near;                     // per 2x2x2 block of cells: a nearby occupied cell, -1 if none within kNearReach
    const unsigned *occ;                 // one bit per cell (cell id = bit index): occupied.  64x smaller than the tables, so it
                                         // stays in L2 where they do not: the searches that walk mostly EMPTY rows test it first
    GridDesc<T> g;
    int m;
    int first;                           // slot of this map's first point: the maps of one batched build share `pts`,
                                         // and all cell tables hold slots into that shared array (no per-map rebasing pass)
    int nsx, nsy, nsz;                   // super-cell grid dims
};

// One cloud of a batched index build.  All clouds of the batch share concatenated arrays: points of
// cloud k live at [pbase, pbase + m), its cells (plus one sentinel slot) at [cbase, cbase + ncells],
// its super-cells at [sbase, sbase + nsc).
template <typename T>
struct BuildDesc {
    const T *xyz; const T *nrm;
    int xstride, nstride, m, near_reach;
    T mean[3];
    GridDesc<T> g;
    long long pbase, cbase, sbase, fbase;   // fbase: offset of this cloud's FINE cells (+1 sentinel) in the fine table
    long long obase;                        // offset (32-bit words) of this cloud's occupancy bits
    long long wbase;                        // offset (16-byte words) of this cloud's succinct table (MapDev::sw)
    long long bbase;                        // offset of this cloud's BINS (512 fine cells each) in the build's counting sort
    int ncells, nsc, kx, ncells_f, nbins;
};

// Chain parameters as the kernels need them.
// [EXT] RobustOutlierFilter (round 5): the weight of a pair is a function of its squared distance over the squared scale
// (ProblemDev::robust_s2: the median absolute deviation's rounded root, squared -- or 1); fct 0: not in the chain
template <typename T>
struct RobustDev {
    int fct;             // 1 cauchy 2 welsch 3 sc 4 gm 5 tukey 6 huber 7 L1
    int mad;             // scaleEstimator: 1 mad, 0 none
    int cut;             // approximation given: e2 >= a2 -> weight 0
    T k, k2, a2;
};

template <typename T>
struct ChainDev {
    T max_dist;          // may be +inf
    T max_dist2;         // max_dist * max_dist in T
    T trim_ratio;
    T trim_scale;        // the quantile filter's limit = trim_scale * order statistic (1: TrimmedDist; MedianDist: its factor)
    T outlier_max_d2;    // MaxDistOutlierFilter.maxDist squared in T; +inf when the chain has none
    int max_iters;
    int smooth;
    double min_rot, min_trans;
    double rank_rel_tol; // 6 * eps(T)
    // the other modules the chain's slots may hold (ABI 4)
    int knn;             // KDTreeMatcher.knn: neighbours per reading point; > 1: the top-K matcher, pairs laid out [point][neighbour]
    int minimizer;       // 0 PointToPlane(WithCov), 1 PointToPoint
    int force4dof;       // PointToPlane{force4DOF}: the increment is a rotation about z and a translation
    double bound_rot, bound_trans;   // BoundTransformationChecker limits; <= 0: not in the chain
    T normal_cos;        // SurfaceNormalOutlierFilter: cos(maxAngle) evaluated in T on the host
    int use_normals;     // ... and whether it is in the chain
    RobustDev<T> robust; // RobustOutlierFilter (the chain's distance filter then: no quantile filter beside it)
    // (ABI 6, per call rather than per chain) PGICP_SUM_ORDER_SCAN: sorted position of every reading point of the batch -- the
    // inverse of the reading sort, indexed like the per-point arrays -- so that the reduce kernels can walk the pairs in the
    // caller's order; null: they walk the sorted order
    const int *scan_pos;
};

// One ICP problem of a batch.  Lives in device memory; written by the solve
// kernel, read by every other kernel; copied back to the host once at the end.
struct ProblemDev {
    int map;                 // slot in the MapDev table
    int n;                   // reading points
    int knn;                 // neighbours per reading point (ChainDev::knn): the per-PAIR arrays (slot, d2, selection keys) hold
    int pad_knn_;            // n * knn entries from offset off * knn on (pairs_n / pairs_off)
    long long off;           // offset (in points) of this problem in the packed per-point arrays
    double Tpre[16];         // T_refMean^-1 * T_init (applied once per scan, cast to T)
    double T_iter[16];       // accumulated correction in the centred map frame
    double T_prev[16];       // T_iter used for the last matching
    double dT[16];           // last increment
    double Tcur[12];         // T_iter as 3x4 for the kernels (cast to T when applied)
    double Tcur_prev[12];    // the Tcur the previous matcher pass ran with (how far each query moved since)
    float Tcur_f[12], Tcur_prev_f[12];   // the same two rounded to float once (by whoever writes Tcur): the float kernels read these
                             // instead of converting twelve uniform doubles per wave and pass
    int done, status, iters, converged, max_iter_reached;
    int n_finite, n_kept, rank;
    int n_refined;           // queued queries the slow path resolved since the last threshold selection
    int far_mode;            // the last matcher pass queued more than an eighth of the queries (a scan that has left its map behind):
                             // the fast kernel then looks at every far query's proven empty radius before it searches (k_sel_final sets it)
    double limit;            // last threshold the pairs are KEPT with: min(scale * quantile, MaxDist filter) (squared distance)
    double rlimit;           // what the matcher must be exact up to for that: max(quantile, limit) -- equal to `limit` for the
                             // TrimmedDist filter; larger with a MedianDist factor below 1 or a MaxDist filter below the quantile
    double prev_limit;       // the `rlimit` the last fast matcher pass derived its search cap from
    double qraw;             // the order statistic the last selection found (before the filter's scale): the next selection's
                             // guess -- it compacts only a band of distances around it (k_sel_band); 0: no guess yet
    double qraw1;            // the same of the last FIRST selection of an iteration: the next first selection's guess.  (A first
                             // selection sees upper bounds where queries are still queued -- a scan that runs ahead of its map has
                             // thousands -- so its result sits well above the iteration's final one, iteration after iteration.)
    // (round 5) hints from the context's PREVIOUS call of the same kind: the raw quantiles that call's problem of the same
    // index found in its iterations 0..3 -- [0]: first selections, [1]: second selections; 0: none.  A stream of scans (SLAM, the
    // streaming mapper) runs two or three iterations per scan: without them every selection of a scan is an "early" one (no
    // guess in iteration 0, a band ten octaves wide after it).  A hint only shapes the band: a rank outside it falls back to the
    // full select, the result is exact either way.  qrec: what THIS call found (read back by the host for the next call).
    double qhint[2][4];
    double qrec[2][4];
    double robust_s2;        // RobustOutlierFilter: this iteration's squared scale (k_robust_finish)
    double sys[kSys];        // final sums of the last iteration
    Checker chk;
};

// status codes inside ProblemDev (mirror PGICP_OK / PGICP_ERR_NO_MATCH / PGICP_ERR_NAN)
#define PGICP_ST_OK 0
#define PGICP_ST_NO_MATCH 1
#define PGICP_ST_NAN 2
#define PGICP_ST_BOUND 3     // BoundTransformationChecker's limit exceeded (PGICP_ERR_BOUND at the ABI)

struct Mat34 { double v[12]; };                    // row-major 3x4, kernel argument
// where a problem's reading lives (device); nptr: its `normals` descriptor, or null
struct SrcDesc { const void *ptr; int stride; int nstride; const void *nptr; };

#ifndef PGICP_REDUCE_ROUNDS
#define PGICP_REDUCE_ROUNDS 2
#endif
constexpr int kKnnBlock = 256;
constexpr int kReduceBlock = 256;
constexpr int kReduceItems = 4;      // queries per thread and round in the reduce kernels (loads of one round in flight together)
constexpr int kReduceRounds = PGICP_REDUCE_ROUNDS;   // rounds per block: one 30-double block reduction per kReduceSpan queries
constexpr int kReduceSpan = kReduceBlock * kReduceItems * kReduceRounds;
constexpr int kSelectBlock = 1024;
constexpr int kCovTerms = 42;
constexpr int kNearReach = 8;
constexpr int kWitReach = 6;      // super-cells searched for MapDev::sc_wit
constexpr int kScReach = 15;      // super-cells searched per axis for MapDev::sc_dist     // cells searched per axis for MapDev::near        // 21 (H upper) + 21 (G upper)

}  // namespace pgicp
