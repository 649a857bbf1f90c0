/*
 * pgicp.h -- C ABI of the MI355X-native ICP hot path of pgslam.
 *
 * This is the drop-in boundary: every entry point replaces a
 * call pgslam makes into libpointmatcher's ICP chain.  Paths below are relative
 * to /root/reference/src/pgslam/.  All entry points are `extern "C"`, take plain
 * pointers and sizes, return an int status (0 = PGICP_OK) and never throw.
 *
 * Conventions
 *   - Point buffers are `float` (PointMatcher<float>) or `double`
 *     (PointMatcher<double>): `_f32` / `_f64` suffix.
 *   - A point buffer is (pointer, stride, count): element a of point i is
 *     ptr[i*stride + a], a in {0,1,2}.  libpointmatcher's `features` matrix
 *     (4 x N, column-major, homogeneous) is (features.data(), 4, N); a `normals`
 *     descriptor view inside a D-row descriptor matrix is (&desc(row0,0), D, N);
 *     a packed xyz array is (ptr, 3, N).
 *   - `mem` says where a buffer lives: PGICP_HOST (caller-owned host memory,
 *     read only during the call) or PGICP_DEVICE (HIP device memory on the
 *     context's GPU, e.g. a torch tensor's data_ptr()).
 *   - 4x4 transforms cross the ABI as 16 doubles, ROW-major.  (The C++ shim
 *     converts from the column-major PM::Matrix.)
 *   - A context is thread-compatible (one thread at a time); different contexts
 *     are fully concurrent, each owns its HIP stream (SURVEY.md §8(b) threading).
 *   - There is NO CPU fallback: without a usable gfx950 device
 *     pgicp_ctx_create() fails with PGICP_ERR_NO_DEVICE.
 */
#ifndef PGICP_H
#define PGICP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGICP_ABI_VERSION 6

/* status codes (pgslam sees PM::ConvergenceError for 1, 2 and 7 through the C++ shim) */
#define PGICP_OK 0
#define PGICP_ERR_NO_MATCH 1    /* "no outlier to filter" / "no point to minimize" */
#define PGICP_ERR_NAN 2         /* NaN in the transformation checkers */
#define PGICP_ERR_ARG 3
#define PGICP_ERR_HIP 4
#define PGICP_ERR_NO_DEVICE 5
#define PGICP_ERR_NOT_RIGID 6   /* RigidTransformation::checkParameters failed */
#define PGICP_ERR_BOUND 7       /* BoundTransformationChecker: limit exceeded (ConvergenceError) */

#define PGICP_HOST 0
#define PGICP_DEVICE 1
#define PGICP_HOST_PINNED 2     /* host memory the caller pinned (pgicp_host_alloc / hipHostMalloc): only for pgicp_upload_* */

#define PGICP_MATCHER_GRID 0    /* grid-hashed exact kNN (performance path) */
#define PGICP_MATCHER_BRUTE 1   /* LDS-tiled brute force (parity path) */

#define PGICP_MINIMIZER_POINT_TO_PLANE 0   /* PointToPlane(WithCov)ErrorMinimizer */
#define PGICP_MINIMIZER_POINT_TO_POINT 1   /* PointToPointErrorMinimizer (no covariance: the base class's zeros) */
#define PGICP_MINIMIZER_POINT_TO_PLANE_4DOF 2   /* PointToPlaneErrorMinimizer{force4DOF: 1}: rotation about z and translation only */
#define PGICP_MINIMIZER_POINT_TO_POINT_WITH_COV 3   /* (ABI 5) PointToPointWithCovErrorMinimizer{sensorStdDev}: the point-to-point solve with the
                                                     * covariance estimate pgslam hands to its optimiser (Localizer.hpp:238, LoopCloser.hpp:108);
                                                     * the estimate reads the reference's `normals`: the map must have them */
#define PGICP_MAX_KNN 16
#define PGICP_ROBUST_NONE 0
#define PGICP_ROBUST_CAUCHY 1
#define PGICP_ROBUST_WELSCH 2
#define PGICP_ROBUST_SC 3
#define PGICP_ROBUST_GM 4
#define PGICP_ROBUST_TUKEY 5
#define PGICP_ROBUST_HUBER 6
#define PGICP_ROBUST_L1 7
#define PGICP_ROBUST_SCALE_NONE 0
#define PGICP_ROBUST_SCALE_MAD 1

typedef struct pgicp_ctx pgicp_ctx;

/* The ICP chain configuration: what pgslam loads from YAML through
 * icp_sequence_.loadFromYaml (Localizer.hpp:70) / icp_.loadFromYaml
 * (LoopCloser.hpp:73).  Field <- libpointmatcher module.parameter. */
typedef struct pgicp_params {
    int knn;                 /* KDTreeMatcher.knn, 1 .. PGICP_MAX_KNN.  With knn > 1 the matches are knn x N ([point][neighbour], in
                              * (distance, index) order), the quantile filter runs over all knn * N distances, every pair is a
                              * constraint of the minimiser and the ratios are over knn * N */
    double epsilon;          /* KDTreeMatcher.epsilon >= 0: libnabo's allowance for an approximate neighbour ((1 + epsilon) x the nearest
                              * distance).  Accepted; the search is exact, which meets every allowance: results are those of epsilon = 0 */
    double max_dist;         /* KDTreeMatcher.maxDist, metres; +inf allowed */
    double trim_ratio;       /* TrimmedDistOutlierFilter.ratio */
    int max_iters;           /* CounterTransformationChecker.maxIterationCount */
    double min_diff_rot;     /* DifferentialTransformationChecker.minDiffRotErr */
    double min_diff_trans;   /* DifferentialTransformationChecker.minDiffTransErr */
    int smooth_length;       /* DifferentialTransformationChecker.smoothLength (<= 15) */
    double sensor_std_dev;   /* PointToPlaneWithCovErrorMinimizer.sensorStdDev */
    int matcher;             /* PGICP_MATCHER_GRID | PGICP_MATCHER_BRUTE */
    double grid_cell;        /* grid cell edge in metres; 0 = automatic */
    int check_every;         /* host polls the device-side "all done" flag every this many iterations (>=1) */
    double outlier_max_dist; /* MaxDistOutlierFilter.maxDist: a second outlier filter whose weights multiply the trimmed
                              * filter's (pairs farther than this get weight 0); 0 or +inf = not in the chain */
    double quantile_scale;   /* (ABI 3) MedianDistOutlierFilter.factor: the chain's quantile filter keeps pairs with
                              * dist <= quantile_scale * getDistsQuantile(trim_ratio) (squared distances).  TrimmedDist is
                              * (trim_ratio, 1); MedianDist is (0.5, factor).  Default 1 */
    /* (ABI 4) the other modules the chain's slots may hold */
    int error_minimizer;     /* PGICP_MINIMIZER_POINT_TO_PLANE | _POINT_TO_POINT | _POINT_TO_PLANE_4DOF | _POINT_TO_POINT_WITH_COV */
    double bound_max_rot;    /* BoundTransformationChecker.maxRotationNorm (rad): 0 or +inf = not in the chain.  The ICP fails with */
    double bound_max_trans;  /* PGICP_ERR_BOUND when the accumulated correction exceeds either (.maxTranslationNorm) */
    double normal_max_angle; /* SurfaceNormalOutlierFilter.maxAngle (rad): a pair whose reading and reference normals differ by more
                              * gets weight 0 (the weights multiply in); needs pgicp_problem.normals; 0 = not in the chain */
    /* (ABI 5) RobustOutlierFilter{robustFct, tuning, scaleEstimator, approximation} (distanceType point2point, nbIterationForScale 0):
     * the chain's DISTANCE filter then -- no quantile filter beside it (trim_ratio must be 1, quantile_scale 1); MaxDist and
     * SurfaceNormal filters may multiply in.  Every pair with a neighbour carries the weight f(dist / scale^2): nothing is trimmed,
     * the matcher resolves every query exactly. */
    int robust_fct;          /* PGICP_ROBUST_NONE (not in the chain) | _CAUCHY | _WELSCH | _SC | _GM | _TUKEY | _HUBER | _L1 */
    double robust_tuning;    /* tuning (default 1) */
    int robust_scale;        /* scaleEstimator: PGICP_ROBUST_SCALE_NONE | PGICP_ROBUST_SCALE_MAD (default: mad) */
    double robust_approx;    /* approximation: pairs with dist / scale^2 >= approximation^2 get weight 0; 0 or +inf = none */
    /* (ABI 6) The order the pairs enter the reduction tree in.  Every sum over pairs that feeds a result -- the normal
     * equations, the residual, the covariance sums -- is added in ONE stated tree (DESIGN.md section 2 "RT-1", restated in
     * oracle/icp_oracle.c): blocks of 2048 pair positions, 256 accumulators a block, a shuffle-down fold, eight block chains.
     * Double addition is not associative, so WHICH pair sits at which position decides the last bits of T, cov and residual:
     *   PGICP_SUM_ORDER_SORTED  (default) positions follow the order the library sorts a reading in for its matcher (by map
     *                           cell): the reduce kernels read everything coalesced.  The order of a call can be read back
     *                           (pgicp_debug_reading_order); results are reproducible run to run, but a function of the map's grid.
     *   PGICP_SUM_ORDER_SCAN    positions follow the caller's reading: results are a function of the inputs alone, bit for bit
     *                           what a scan-order walk of the same tree gives on any implementation; the reduce kernels then
     *                           gather (costs: DESIGN.md section 5). */
    int sum_order;
} pgicp_params;
#define PGICP_SUM_ORDER_SORTED 0
#define PGICP_SUM_ORDER_SCAN 1

/* What pgslam reads back after an ICP: errorMinimizer->getOverlap()
 * (Localizer.hpp:278, LoopCloser.hpp:331), getCovariance() (Localizer.hpp:238,
 * LoopCloser.hpp:108), icp_.getMaxNumIterationsReached() (LoopCloser.hpp:317). */
typedef struct pgicp_stats {
    int status;              /* PGICP_OK or PGICP_ERR_NO_MATCH / PGICP_ERR_NAN for this problem */
    int iterations;
    int converged;           /* DifferentialTransformationChecker stopped the loop */
    int max_iter_reached;    /* CounterTransformationChecker stopped the loop */
    double overlap;          /* weightedPointUsedRatio of the last ErrorElements */
    double residual;         /* sum w (n.(p-q))^2 of the last ErrorElements */
    double trim_limit;       /* last TrimmedDist threshold (squared distance) */
    int n_kept;              /* pairs with non-zero weight in the last iteration */
    int n_finite;            /* pairs with a neighbour within maxDist */
    double cov[36];          /* 6x6 row-major, order [x y z rx ry rz] (Optimizer.hpp:32-42) */
} pgicp_stats;

/* One ICP problem of a batch (scan-to-map: many readings, one map;
 * loop closure: one map per pair -- LoopCloser.hpp:98). */
typedef struct pgicp_problem {
    int map_id;
    const void *reading;     /* float* or double* according to the entry point */
    int stride;
    int n;
    int mem;
    double T_init[16];
    const void *normals;     /* (ABI 4) the reading's `normals` descriptor (same type and `mem` as `reading`), or NULL: only a */
    int nstride;             /* SurfaceNormalOutlierFilter in the chain looks at it */
} pgicp_problem;

/* Loop-closure edge record exchanged between GPUs: the payload of
 * Optimizer<T>::InputData = tuple<Vertex,Vertex,Matrix,CovMatrix>
 * (Optimizer.h:22) plus the acceptance evidence of LoopCloser::CheckIcpResult
 * (LoopCloser.hpp:308-340).  Exactly 512 bytes so a rank's edges all-gather as
 * one contiguous buffer. */
typedef struct pgicp_edge {
    int64_t from_id;
    int64_t to_id;
    int32_t accepted;        /* CheckIcpResult() */
    int32_t status;
    int32_t iterations;
    int32_t max_iter_reached;
    double overlap;
    double residual;
    double T_from_to[16];
    double cov[36];
    double reserved[6];
} pgicp_edge;

/* ---- context --------------------------------------------------------- */
int pgicp_abi_version(void);
int pgicp_device_count(void);
int pgicp_ctx_create(int device, pgicp_ctx **out);
/* (ABI 5) the same with a stream of the device's highest priority (high_priority != 0): the context's launches are scheduled ahead of
 * what other contexts have queued -- for a short, latency-critical stage that runs next to a long one (the MT flavour's input stage,
 * LocalizerMT.hpp:27-40, next to the localizer's ICP) */
int pgicp_ctx_create_priority(int device, int high_priority, pgicp_ctx **out);
void pgicp_ctx_destroy(pgicp_ctx *ctx);
const char *pgicp_last_error(const pgicp_ctx *ctx);
/* The HIP stream (hipStream_t) every kernel of this context is launched on. */
void *pgicp_ctx_stream(pgicp_ctx *ctx);
int pgicp_ctx_device(const pgicp_ctx *ctx, int *device);
int pgicp_ctx_synchronize(pgicp_ctx *ctx);

/* Readable text of a status code (never NULL). */
const char *pgicp_status_string(int status);

/* ---- host-input pipeline ---------------------------------------------------
 * pgslam hands the localizer caller-owned HOST clouds (Localizer.hpp:103-126), and in the MT flavour the
 * next scan is already queued while the current one is being aligned (LocalizerMT.hpp:27-40).
 * pgicp_upload_* starts the transfer of n host readings to the device on the context's COPY stream and
 * returns at once; dev_ptrs[k] receives the device address of reading k (same stride as the source).  Any
 * later call on this context that is given such a pointer (mem = PGICP_DEVICE: pgicp_align*, pgicp_match,
 * pgicp_partial_chain*, pgicp_map_create*) first makes its own stream wait, ON THE DEVICE, for the
 * transfer -- the host never blocks on it -- so the upload of scan k+1 overlaps the ICP of scan k.
 * Two uploads are kept: upload u+2 reuses the buffers of upload u (it waits, on the device, until the
 * calls that read upload u have consumed it) -- the pointers of upload u are dead once upload u+2 has been started, so
 * readings that one call will use together travel in ONE upload.  `mem` says what the host pointers are: PGICP_HOST
 * (pageable: copied through a pinned staging buffer of the context first) or PGICP_HOST_PINNED (DMA
 * straight from the caller's buffer, which must stay untouched until a call that uses it has returned).
 * pgicp_host_alloc / pgicp_host_free: pinned host memory for such clouds. */
int pgicp_upload_f32(pgicp_ctx *ctx, int n_readings, const float *const *host, const int *stride, const int *n, int mem,
                     const float **dev_ptrs);
int pgicp_upload_f64(pgicp_ctx *ctx, int n_readings, const double *const *host, const int *stride, const int *n, int mem,
                     const double **dev_ptrs);
int pgicp_host_alloc(pgicp_ctx *ctx, size_t bytes, void **out);
int pgicp_host_free(pgicp_ctx *ctx, void *p);
/* Device memory a caller owns between calls: the keyframe clouds of a local map (LocalMap.hpp:209-224 walks them at every
 * rebuild; Keyframe, types.h:31-44) stay in HBM, pgicp_build_local_map(mem = PGICP_DEVICE) assembles the next map from
 * them into a buffer of this kind and pgicp_map_create(mem = PGICP_DEVICE) indexes it -- no cloud crosses PCIe at a rebuild.
 * Plain buffers on the context's device, usable by every context of that device.  pgicp_device_copy returns when the copy
 * is complete (it is ordered after the context's queued work).  pgicp_device_free waits for the device; its context may be
 * NULL (an owner that outlives its context). */
#define PGICP_COPY_TO_DEVICE 0
#define PGICP_COPY_FROM_DEVICE 1
#define PGICP_COPY_ON_DEVICE 2
int pgicp_device_alloc(pgicp_ctx *ctx, size_t bytes, void **out);
int pgicp_device_free(pgicp_ctx *ctx, void *p);
int pgicp_device_copy(pgicp_ctx *ctx, void *dst, const void *src, size_t bytes, int kind);

void pgicp_default_params(pgicp_params *p);
/* replaces ICP::loadFromYaml / setDefault for the supported chain */
int pgicp_set_params(pgicp_ctx *ctx, const pgicp_params *p);
int pgicp_get_params(const pgicp_ctx *ctx, pgicp_params *p);

/* ---- reference cloud / matcher index ---------------------------------
 * pgicp_map_create = ICPSequence::setMap (Localizer.hpp:148,168,254) when
 * center != 0 (the reference is mean-centred) and
 * matcher->init(reference) (Localizer.hpp:317, LoopCloser.hpp:356) when
 * center == 0.  Copies the cloud to the device, builds the grid index; the map
 * stays resident until pgicp_map_destroy. `nrm` may be NULL for match-only maps. */
int pgicp_map_create_f32(pgicp_ctx *ctx, const float *xyz, int xyz_stride, const float *nrm, int nrm_stride,
                         int m, int mem, int center, int *map_id);
int pgicp_map_create_f64(pgicp_ctx *ctx, const double *xyz, int xyz_stride, const double *nrm, int nrm_stride,
                         int m, int mem, int center, int *map_id);
/* n reference clouds in one call (one per loop-closure pair, LoopCloser.hpp:98 builds the matcher of
 * `reference` inside icp_): one host round trip for all of them.  Arrays are indexed by cloud;
 * nrm[k] (and nrm itself) may be NULL. */
int pgicp_map_create_batch_f32(pgicp_ctx *ctx, int n_maps, const float *const *xyz, const int *xyz_stride,
                               const float *const *nrm, const int *nrm_stride, const int *m, int mem, int center,
                               int *map_ids);
int pgicp_map_create_batch_f64(pgicp_ctx *ctx, int n_maps, const double *const *xyz, const int *xyz_stride,
                               const double *const *nrm, const int *nrm_stride, const int *m, int mem, int center,
                               int *map_ids);
int pgicp_map_destroy(pgicp_ctx *ctx, int map_id);
int pgicp_map_size(pgicp_ctx *ctx, int map_id, int *m);
/* Hands a map from one context to another of the same device without copying it.  A background
 * context (its own stream and scratch) can so assemble and index the next local map --
 * LocalMap::UpdateToNewComposition + setMap, Localizer.hpp:262-266 -- while `to` keeps aligning
 * scans against the current one.  `to`'s stream waits, on the device, for the work queued on
 * `from`'s stream so far; the host does not block.  The id is invalid in `from` afterwards. */
int pgicp_map_transfer(pgicp_ctx *from, int map_id, pgicp_ctx *to, int *new_id);

/* ---- full ICP --------------------------------------------------------
 * pgicp_align = ICPSequence::operator()(reading, T_init) (Localizer.hpp:126).
 * pgicp_align_batch runs independent problems concurrently on the device
 * (LoopCloserMT's queue, LoopCloserMT.hpp:26-67, is where batches form).
 * pgicp_icp_pair = ICP::operator()(reading, reference, T_init)
 * (LoopCloser.hpp:98): builds the centred index, aligns, frees the index. */
int pgicp_align_f32(pgicp_ctx *ctx, int map_id, const float *reading, int stride, int n, int mem,
                    const double T_init[16], double T_out[16], pgicp_stats *stats);
int pgicp_align_f64(pgicp_ctx *ctx, int map_id, const double *reading, int stride, int n, int mem,
                    const double T_init[16], double T_out[16], pgicp_stats *stats);
int pgicp_align_batch_f32(pgicp_ctx *ctx, int n_problems, const pgicp_problem *problems, double *T_out,
                          pgicp_stats *stats);
int pgicp_align_batch_f64(pgicp_ctx *ctx, int n_problems, const pgicp_problem *problems, double *T_out,
                          pgicp_stats *stats);
/* pgicp_align_residual_batch = the numeric part of LoopCloser::ProcessVertex for a batch of candidates
 * (LoopCloser.hpp:83-110): ICP::operator() per pair (:98), then -- for CheckIcpResult (:308-340) -- ComputeResidualError's
 * chain on the result (:343-365: the reading moved by the result, matched, weighted, getResidualError), FUSED: the residual
 * pass starts from the last iteration's correspondences instead of searching from scratch (they are candidates only; the
 * matches are the exact ones).  residual[p] = getResidualError (+inf when the ICP of p failed or nothing is kept), ratio[p]
 * (optional) = weightedPointUsedRatio of that pass, status[p] (optional) = PGICP_OK / PGICP_ERR_NO_MATCH.  The maps are the
 * ICP's (centred) indices: against pgicp_partial_chain_batch on an un-centred index the figures agree to float rounding. */
int pgicp_align_residual_batch_f32(pgicp_ctx *ctx, int n_problems, const pgicp_problem *problems, double *T_out, pgicp_stats *stats,
                                   double *residual, double *ratio, int *status);
int pgicp_align_residual_batch_f64(pgicp_ctx *ctx, int n_problems, const pgicp_problem *problems, double *T_out, pgicp_stats *stats,
                                   double *residual, double *ratio, int *status);
int pgicp_icp_pair_f32(pgicp_ctx *ctx, const float *reading, int rd_stride, int n, const float *ref_xyz,
                       int ref_stride, const float *ref_nrm, int nrm_stride, int m, int mem,
                       const double T_init[16], double T_out[16], pgicp_stats *stats);
int pgicp_icp_pair_f64(pgicp_ctx *ctx, const double *reading, int rd_stride, int n, const double *ref_xyz,
                       int ref_stride, const double *ref_nrm, int nrm_stride, int m, int mem,
                       const double T_init[16], double T_out[16], pgicp_stats *stats);

/* ---- chain stages (the partial-chain callers) ------------------------
 * pgicp_match = matcher->findClosests (Localizer.hpp:328, LoopCloser.hpp:358):
 * the reading is first moved by T (NULL = identity), then matched.  ids are
 * indices into the cloud given to pgicp_map_create (-1 = no neighbour within
 * maxDist), dist2 are SQUARED distances (+inf when id == -1); both hold knn entries
 * per reading point ([point][neighbour]: a knn x N column-major Matches matrix).
 * pgicp_outlier_weights = outlierFilters.compute (Localizer.hpp:330,
 * LoopCloser.hpp:360) for TrimmedDistOutlierFilter.
 * pgicp_error_stats = ErrorElements(...) + weightedPointUsedRatio
 * (Localizer.hpp:332,347) and getResidualError (LoopCloser.hpp:362);
 * `reading` must already be in the map's frame (as pgslam passes it).
 * pgicp_partial_chain fuses the three for Localizer::ComputeOverlapWith
 * (Localizer.hpp:282-348) and LoopCloser::ComputeResidualError
 * (LoopCloser.hpp:343-365): one device pass, no host round trip. */
int pgicp_match_f32(pgicp_ctx *ctx, int map_id, const float *reading, int stride, int n, int mem,
                    const double *T, int32_t *ids, float *dist2);
int pgicp_match_f64(pgicp_ctx *ctx, int map_id, const double *reading, int stride, int n, int mem,
                    const double *T, int32_t *ids, double *dist2);
int pgicp_outlier_weights_f32(pgicp_ctx *ctx, const float *dist2, int n, int mem, float *weights,
                              float *limit, int *n_finite);
int pgicp_outlier_weights_f64(pgicp_ctx *ctx, const double *dist2, int n, int mem, double *weights,
                              double *limit, int *n_finite);
int pgicp_error_stats_f32(pgicp_ctx *ctx, int map_id, const float *reading, int stride, int n, int mem,
                          const int32_t *ids, const float *weights, double *weighted_point_used_ratio,
                          double *residual, double sys[30]);
int pgicp_error_stats_f64(pgicp_ctx *ctx, int map_id, const double *reading, int stride, int n, int mem,
                          const int32_t *ids, const double *weights, double *weighted_point_used_ratio,
                          double *residual, double sys[30]);
int pgicp_partial_chain_f32(pgicp_ctx *ctx, int map_id, const float *reading, int stride, int n, int mem,
                            const double *T, double *weighted_point_used_ratio, double *residual);
int pgicp_partial_chain_f64(pgicp_ctx *ctx, int map_id, const double *reading, int stride, int n, int mem,
                            const double *T, double *weighted_point_used_ratio, double *residual);
/* (ABI 6) The same chain, its matcher SEEDED from the correspondences another context's last align call ended with -- the
 * overlap probe of a scan that has just been aligned (Localizer.hpp:210,231 -> 282-348: up to once per scan): the candidate
 * composition shares keyframes with the map the ICP matched against, so most of the scan's neighbours exist in the candidate
 * map too, at known index offsets.  `src`: the context that aligned THIS reading (same points, same order) as problem 0 of its
 * last pgicp_align* call, idle now, on the same device; the two maps as concatenations of keyframe clouds: segment k of
 * the ICP's reference holds original indices [src_start[k], src_start[k + 1]) and sits at dst_start[k] in this call's map
 * (-1: not part of it).  A seed is a candidate only: the matches are exact, the ratio is pgicp_partial_chain's double.  The reading
 * is taken in the order the SOURCE context sorted it in (one set-up kernel instead of a sort of its own), so with
 * PGICP_SUM_ORDER_SORTED the residual is the same sum in another order (equal to ~1e-16 relative; identical with
 * PGICP_SUM_ORDER_SCAN) -- tests/test_gpu_parity.py.  From this context's second probe on the search is also CAPPED at 1.21 x 3 x the
 * threshold its previous probe ended with (PGICP_PROBE_CAP_SCALE), as an ICP's later iterations are by their previous one's (a query with nothing inside
 * the cap is resolved lazily, only if the new threshold turns out to need it: a cap that is wrong costs time, never a result --
 * same test).  A reading the source context did not align (another size), a chain with
 * knn > 1, the brute matcher or a SurfaceNormalOutlierFilter: searched unseeded, as pgicp_partial_chain. */
int pgicp_partial_chain_seeded_f32(pgicp_ctx *ctx, int map_id, const float *reading, int stride, int n, int mem, const double *T,
                                   pgicp_ctx *src, int n_seg, const int32_t *src_start, const int32_t *dst_start,
                                   double *weighted_point_used_ratio, double *residual);
int pgicp_partial_chain_seeded_f64(pgicp_ctx *ctx, int map_id, const double *reading, int stride, int n, int mem, const double *T,
                                   pgicp_ctx *src, int n_seg, const int32_t *src_start, const int32_t *dst_start,
                                   double *weighted_point_used_ratio, double *residual);
/* the same chain for a batch of (map, reading, T = problems[p].T_init): the residual check of every
 * loop-closure candidate of a batch (LoopCloser.hpp:340-363) in one device pass.  status[p] is
 * PGICP_OK or PGICP_ERR_NO_MATCH; the call returns the worst status.  Output arrays may be NULL. */
int pgicp_partial_chain_batch_f32(pgicp_ctx *ctx, int n_problems, const pgicp_problem *problems,
                                  double *weighted_point_used_ratio, double *residual, int *status);
int pgicp_partial_chain_batch_f64(pgicp_ctx *ctx, int n_problems, const pgicp_problem *problems,
                                  double *weighted_point_used_ratio, double *residual, int *status);

/* ---- rigid transform and local-map assembly ---------------------------
 * pgicp_transform = rigid_transformation_->compute (Localizer.hpp:106,323,
 * LocalMap.hpp:97,222) / transformations.apply (LoopCloser.hpp:352): points get
 * R*p + t, normals (rotate_only) get R*n.  In-place (out == in) is allowed.
 * pgicp_build_local_map = LocalMap::BuildCloudFromData (LocalMap.hpp:209-224):
 * cloud 0 is copied, cloud k>0 is moved by T_ref_kf[k] and appended. */
int pgicp_transform_f32(pgicp_ctx *ctx, const double T[16], const float *in, int in_stride, float *out,
                        int out_stride, int n, int rotate_only, int mem);
int pgicp_transform_f64(pgicp_ctx *ctx, const double T[16], const double *in, int in_stride, double *out,
                        int out_stride, int n, int rotate_only, int mem);
int pgicp_build_local_map_f32(pgicp_ctx *ctx, int n_kf, const float *const *xyz, const float *const *nrm,
                              const int *strides_xyz, const int *strides_nrm, const int *counts,
                              const double *T_ref_kf, float *out_xyz, int out_xyz_stride, float *out_nrm,
                              int out_nrm_stride, int mem);
int pgicp_build_local_map_f64(pgicp_ctx *ctx, int n_kf, const double *const *xyz, const double *const *nrm,
                              const int *strides_xyz, const int *strides_nrm, const int *counts,
                              const double *T_ref_kf, double *out_xyz, int out_xyz_stride, double *out_nrm,
                              int out_nrm_stride, int mem);

/* ---- input filters ---------------------------------------------------------
 * pgicp_surface_normals = [EXT] libpointmatcher SurfaceNormalDataPointsFilter{knn, maxDist,
 * epsilon = 0, keepNormals = 1, keepEigenValues, keepMatchedIds} as a user's input-filter or
 * reference-filter YAML applies it (input_filters_.apply, Localizer.hpp:103; referenceDataPointsFilters
 * .apply, Localizer.hpp:315): for every point the knn nearest points of the SAME cloud (itself
 * included, within maxDist), their scatter matrix, and the unit eigenvector of its smallest
 * eigenvalue.  out_nrm: 3 values per point at out_stride; out_eig (optional): 3 eigenvalues per
 * point, ascending; out_ids / out_d2 (optional): knn neighbour indices / squared distances per
 * point in (distance, index) order, -1 / +inf where fewer than knn lie within maxDist.  A scatter
 * of rank < 2 yields libpointmatcher's defaults: normal (0,1,0), eigenvalues (0,0,1).  knn <= 32. */
int pgicp_surface_normals_f32(pgicp_ctx *ctx, const float *xyz, int stride, int n, int mem, int knn, double max_dist,
                              float *out_nrm, int out_stride, float *out_eig, int32_t *out_ids, float *out_d2);
int pgicp_surface_normals_f64(pgicp_ctx *ctx, const double *xyz, int stride, int n, int mem, int knn, double max_dist,
                              double *out_nrm, int out_stride, double *out_eig, int32_t *out_ids, double *out_d2);

/* pgicp_filter_cloud = the localizer's input stage on the device: input_filters_.apply(cloud) (Localizer.hpp:103) for the
 * filters that only drop points, then rigid_transformation_->compute(cloud, T_robot_sensor) (Localizer.hpp:106), in one pass
 * over the uploaded scan.  `features`: frows x n column-major (a point = frows contiguous values, xyz first); `descriptors`
 * (may be NULL): drows x n; both HOST.  The filters run in list order, each on what the one before it kept (FixStep /
 * RandomSampling see the points' indices in THAT cloud); the kept points are those libpointmatcher's filters keep, in order.
 * T (16 doubles, row major; NULL: none) moves the kept points (R p + t) and rotates the descriptor rows [rotate_row0, +3) and
 * [rotate_row1, +3) (`normals`, `observationDirections`; -1: none).  out_features / out_descriptors (HOST, room for n points;
 * may alias the inputs: the call is synchronous) receive the n_out kept points, kept_idx (optional) their input indices.
 * dev_features (optional) receives the DEVICE address of the filtered features (stride frows): a reading for pgicp_align_*
 * with mem = PGICP_DEVICE that needs no second upload; valid for the next three pgicp_filter_cloud calls on the context. */
#define PGICP_FILTER_IDENTITY 0
#define PGICP_FILTER_MAX_DIST 1         /* p[0] = maxDist, p[1] = dim + 1 (0: dim = -1, the radius): keeps |p| < |maxDist| -- the norm in T, strict --
                                         * or features(dim) < maxDist ([EXT] MaxDistDataPointsFilter) */
#define PGICP_FILTER_MIN_DIST 2         /* p[0] = minDist, p[1] = dim + 1: keeps |p| > |minDist| or features(dim) > minDist (MinDistDataPointsFilter);
                                         * a NaN coordinate fails both filters' comparisons: dropped */
#define PGICP_FILTER_BOUNDING_BOX 3     /* p[0..2] = min xyz, p[3..5] = max xyz, p[6] = removeInside */
#define PGICP_FILTER_REMOVE_NAN 4
#define PGICP_FILTER_FIX_STEP 5         /* p[0] = step: keeps points 0, step, 2 step, ... (upstream starts at rand() % step: the phase is not
                                         * reproducible there; here it is 0.  A step that changes from call to call -- stepMult -- is the caller's state) */
#define PGICP_FILTER_RANDOM_SAMPLING 6  /* p[0] = prob, p[1] = seed: the counter-based sampler of pointmatcher.hpp */
#define PGICP_FILTER_MAX_POINT_COUNT 7  /* (ABI 5) p[0] = maxCount, p[1] = seed: when more than maxCount points arrive, the same sampler with
                                         * prob = T(maxCount) / T(N) ([EXT] MaxPointCountDataPointsFilter; like RandomSampling not rand()-parity) */
#define PGICP_MAX_FILTERS 8
typedef struct pgicp_filter {
    int type;
    double p[8];
} pgicp_filter;
int pgicp_filter_cloud_f32(pgicp_ctx *ctx, int n_filters, const pgicp_filter *filters, const float *features, int frows,
                           const float *descriptors, int drows, int n, const double *T, int rotate_row0, int rotate_row1,
                           float *out_features, float *out_descriptors, int32_t *kept_idx, int *n_out, const float **dev_features);
int pgicp_filter_cloud_f64(pgicp_ctx *ctx, int n_filters, const pgicp_filter *filters, const double *features, int frows,
                           const double *descriptors, int drows, int n, const double *T, int rotate_row0, int rotate_row1,
                           double *out_features, double *out_descriptors, int32_t *kept_idx, int *n_out, const double **dev_features);

/* (ABI 5) pgicp_filter_cloud_dev: the same device pass for a caller that does the host side itself -- WITHOUT a transformation (a
 * sensor at the robot's origin, or a caller that transforms later): the caller's arrays are not written.  n_out = points kept,
 * dev_features = their device copy (an ICP reading, valid for the next three filter calls on the context), dropped_idx = the
 * ASCENDING input indices of the points dropped, n_dropped of them -- when n_dropped > dropped_cap the list is incomplete and
 * the caller takes pgicp_filter_cloud instead.  A range sensor's input filters drop a handful of a scan's points: the caller
 * closes those gaps in its own arrays, on another thread while the ICP already runs on the device copy if it likes
 * (pgslam::GraphLocalizer does). */
int pgicp_filter_cloud_dev_f32(pgicp_ctx *ctx, int n_filters, const pgicp_filter *filters, const float *features, int frows, int n,
                               int32_t *dropped_idx, int dropped_cap, int *n_dropped, int *n_out, const float **dev_features);
int pgicp_filter_cloud_dev_f64(pgicp_ctx *ctx, int n_filters, const pgicp_filter *filters, const double *features, int frows, int n,
                               int32_t *dropped_idx, int dropped_cap, int *n_dropped, int *n_out, const double **dev_features);

/* ---- loop-closure dispatcher helpers (host logic, no GPU needed) -------
 * pgicp_shard_pairs: deterministic longest-processing-time split of n_pairs
 * candidate ICPs (cost[i] ~ N_i + M_i) over world_size ranks; writes the pair
 * indices of `rank` (at most cap) and returns their count through n_out.
 * pgicp_check_icp_result = LoopCloser::CheckIcpResult (LoopCloser.hpp:308-340). */
int pgicp_shard_pairs(int n_pairs, const int64_t *cost, int world_size, int rank, int *out_idx, int cap,
                      int *n_out);
int pgicp_check_icp_result(const pgicp_stats *icp, double residual_error, double overlap_threshold,
                           double residual_error_threshold);

/* ---- the collective of the path: all-gather of loop-closure edges over RCCL / xGMI ----------
 * One process per GPU.  Rank 0 makes a unique id (pgicp_comm_unique_id), the caller passes its bytes to the
 * other ranks by any channel it has, every rank builds its communicator on its context's device
 * (pgicp_comm_create = ncclCommInitRank).  RCCL is opened at run time on first use.
 * pgicp_allgather_edges: every rank contributes the n_local edges it aligned (pair_index[k] = position of
 * edge k in the job's candidate list) and receives ALL n_total edges in candidate order -- what
 * OptimizerMT::Main drains into one solve (OptimizerMT.hpp:59-65; payload Optimizer.h:22).  It is ONE
 * ncclAllGather of fixed-size blocks of slots_per_rank 512-byte records: slots_per_rank is the largest shard,
 * which every rank computes itself from the deterministic split (pgicp_shard_slots) -- no size exchange.
 * Empty slots and candidates nobody reported come back with from_id = to_id = status = -1.  The context's
 * stream carries the collective; the call returns when `out` is complete (reserved[] of every record is zero
 * again: the pair index travels in reserved[0] and is removed).  Errors of these entry points:
 * pgicp_comm_last_error() (per thread).
 * pgicp_comm_create_host makes a communicator with the HOST transport instead (ABI 3): the same pack / unpack, the
 * blocks travel through a shared-memory file `shm_path` (rank 0 creates it, the others wait for it; the path must
 * not be reused by two live communicators; the last rank to be destroyed removes it) holding at most
 * max_slots_per_rank records per rank.  No device is needed: it is how the slot / reorder logic is tested at world
 * sizes > 1 on a CPU box, and a single-node job without RCCL can gather through it. */
typedef struct pgicp_comm pgicp_comm;
#define PGICP_UNIQUE_ID_BYTES 128
int pgicp_comm_unique_id(char id[PGICP_UNIQUE_ID_BYTES]);
int pgicp_comm_create(pgicp_ctx *ctx, int world_size, int rank, const char id[PGICP_UNIQUE_ID_BYTES], pgicp_comm **out);
int pgicp_comm_create_host(int world_size, int rank, const char *shm_path, int max_slots_per_rank, pgicp_comm **out);
void pgicp_comm_destroy(pgicp_comm *comm);
int pgicp_comm_info(const pgicp_comm *comm, int *world_size, int *rank);
const char *pgicp_comm_last_error(void);
int pgicp_shard_slots(int n_pairs, const int64_t *cost, int world_size, int *slots);
int pgicp_allgather_edges(pgicp_comm *comm, const pgicp_edge *local, const int *pair_index, int n_local, int slots_per_rank,
                          int n_total, pgicp_edge *out);

/* ---- measurement -------------------------------------------------------
 * With profiling on, every launch of the named kernels is bracketed by HIP
 * events on the context stream; pgicp_profile_get returns launch count and
 * summed device milliseconds.  Kernel ids: 0 knn_grid, 1 knn_brute, 2 trim_select,
 * 3 p2plane_reduce, 4 solve_update, 5 pretransform (+ reading sort), 6 covariance, 7 grid_build,
 * 8 knn_slow (wave-cooperative resolution of queued queries). */
#define PGICP_PROF_KNN_GRID 0
#define PGICP_PROF_KNN_BRUTE 1
#define PGICP_PROF_TRIM 2
#define PGICP_PROF_REDUCE 3
#define PGICP_PROF_SOLVE 4
#define PGICP_PROF_PRETRANSFORM 5
#define PGICP_PROF_COV 6
#define PGICP_PROF_GRID_BUILD 7
#define PGICP_PROF_KNN_SLOW 8
#define PGICP_PROF_NORMALS 9
#define PGICP_PROF_KNN_GRID_UNSEEDED 10   /* the launches of id 0 that had no previous correspondences to start from (an ICP's first pass) */
#define PGICP_PROF_COUNT 11
/* diagnostics of the last kNN launch: [0] queries queued by the fast path, [1] queued queries
 * resolved because their existence was unknown, [2] resolved because their lower bound was
 * within the trim threshold, [3] outlier-filter selections since the last call (all contexts of the device) whose
 * guess -- the last selection's result -- was off, so that one block selected over all of a problem's distances. */
int pgicp_debug_counters(pgicp_ctx *ctx, int out[4]);
/* (ABI 5) Process-wide count and summed host nanoseconds of the library's hipMalloc [0,1], hipFree [2,3], hipHostMalloc [4,5] and
 * hipHostFree [6,7] calls: their cost is the HOST's (1 ms on some boxes, 45 ms on others; hipFree waits for the whole device),
 * so a caller's slow pass can be told apart from a slow GPU.  No context needed. */
int pgicp_debug_alloc_stats(long long out[8]);
/* The correspondences the LAST iteration of problem `problem` of the last align call ended with (libpointmatcher's
 * lastErrorElements, before the compaction), in reading order (host buffers of n x knn entries, point-major).  Read by the
 * tests, and by the C++ shim for getOverlap()'s sensor-noise branch (ErrorMinimizers/PointToPlane.cpp: a reading that
 * carries `simpleSensorNoise`; Localizer.hpp:278, LoopCloser.hpp:331) -- part of the supported surface despite its name.  ids: reference index, -1 = no neighbour
 * within maxDist, -2 = a neighbour exists but was not located (lazy resolution: its distance is an
 * upper bound and lies beyond the trim threshold).  Kept pairs carry exact ids and distances. */
int pgicp_debug_last_matches_f32(pgicp_ctx *ctx, int problem, int32_t *ids, float *dist2);
int pgicp_debug_last_matches_f64(pgicp_ctx *ctx, int problem, int32_t *ids, double *dist2);
/* (ABI 6) Diagnostics: the permutation problem `problem` of the last align / partial-chain call sorted its reading by --
 * order[j] = index (in the caller's reading) of the point at sorted position j; n entries.  With PGICP_SUM_ORDER_SORTED this
 * is the order the pairs entered the reduction tree in: hand it to the oracle (orc_params.pair_order) and the whole ICP
 * compares bit for bit (tests/test_gpu_bit_exact.py). */
int pgicp_debug_reading_order(pgicp_ctx *ctx, int problem, int32_t *order);
int pgicp_profile_enable(pgicp_ctx *ctx, int on);
int pgicp_profile_reset(pgicp_ctx *ctx);
int pgicp_profile_get(pgicp_ctx *ctx, int kernel_id, long long *launches, double *total_ms,
                      long long *units /* reading points of the ACTIVE problems of those launches */,
                      long long *problems /* ACTIVE (not yet converged) problems of those launches */);

/* Process-wide profile (ABI 3): the same sums over EVERY context of the process, destroyed or alive (alive ones must be
 * idle), plus the reference points of the active problems of the matcher launches -- what the roofline's algorithmic
 * bytes need (20 N + 12 M per active problem).  With PGICP_PROFILE_ALL=1 in the environment every context profiles
 * from its creation: for callers that never see the contexts (the C++ facade's ICP objects own them). */
int pgicp_profile_process(int kernel_id, long long *launches, double *total_ms, long long *units, long long *problems,
                          long long *map_points);

#ifdef __cplusplus
}
#endif
#endif /* PGICP_H */
