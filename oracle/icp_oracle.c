/*
 * icp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the libpointmatcher ICP chain that pgslam's hot path
 * resolves to.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (libpgicp.so) never
 * links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (Ellon/pgslam) holds no golden vectors and its
 * arithmetic lives in un-vendored, un-pinned libpointmatcher / libnabo / Eigen
 * (reference CMakeLists.txt:17-18), none of which exist in the build container
 * (SURVEY.md F4/F5, §8(c)).  This file restates the *published* algorithm of
 * those libraries and is pinned instead by
 *   (i)  analytic known-answer tests and
 *   (ii) an independent scipy.cKDTree / numpy-float64 cross-check
 * (tests/test_oracle_*.py, fixtures under tests/golden/).
 *
 * Reference call sites each function stands in for (paths relative to
 * /root/reference/src/pgslam/):
 *   orc_icp              ICPSequence::operator() Localizer.hpp:126,
 *                        ICP::operator()         LoopCloser.hpp:98      [A.2]
 *   orc_knn_brute/_kd    matcher->init / findClosests
 *                        Localizer.hpp:317,328  LoopCloser.hpp:356,358  [A.3]
 *   orc_trim_weights     outlierFilters.compute Localizer.hpp:330,
 *                        LoopCloser.hpp:360                             [A.4]
 *   orc_p2plane_system   ErrorElements ctor + weightedPointUsedRatio
 *                        Localizer.hpp:332,347 (sys[27]/N)              [A.5]
 *                        errorMinimizer->compute (inside ICP)           [A.6]
 *                        getResidualError LoopCloser.hpp:362 (sys[29])  [A.7]
 *   orc_partial_chain    Localizer::ComputeOverlapWith Localizer.hpp:282-348,
 *                        LoopCloser::ComputeResidualError LoopCloser.hpp:343-365
 *   orc_covariance       getCovariance          Localizer.hpp:238,
 *                        LoopCloser.hpp:108                             [A.8]
 *   checkers             getMaxNumIterationsReached LoopCloser.hpp:317  [A.9]
 *                        (Counter, Differential, Bound)
 *   orc_p2point_system / orc_solve_p2point, orc_normal_weights, knn > 1
 *                        the other modules the same chain slots may name: the chain is whatever the user's YAML says
 *                        (loadFromYaml at Localizer.hpp:70, LoopCloser.hpp:73)  [A.3, A.4, A.6]
 *   orc_transform        rigid_transformation_->compute Localizer.hpp:106,323
 *                        LocalMap.hpp:97,222  LoopCloser.hpp:352        [a9]
 *   orc_build_local_map  LocalMap::BuildCloudFromData LocalMap.hpp:209-224 [a12]
 *
 * Arithmetic contract shared with the HIP path (DESIGN.md "Arithmetic contract"):
 *   - points/normals are `real` (float or double, -DORC_DOUBLE); compiled with
 *     -ffp-contract=off so no FMA is formed anywhere;
 *   - rigid transform: x' = ((r00*x + r01*y) + r02*z) + tx in `real`;
 *   - squared distance: d2 = ((dx*dx + dy*dy) + dz*dz) in `real`, dx = q - m;
 *   - nearest neighbour = argmin over (d2, index) lexicographically, accepted
 *     iff d2 <= maxDist*maxDist; otherwise id = -1, d2 = +inf;
 *   - centroid: fixed-point (2^-24) integer sum => order independent;
 *   - normal equations, residual, covariance: double accumulation;
 *   - 6x6 solve, SE(3) composition and convergence checks: double.
 *
 * SENSITIVITY VARIANTS (round 5; oracle/Makefile builds one extra library per flag, tests/test_sensitivity.py and
 * tools/sensitivity_envelope.py compare them with the default build -- they answer "how far does the result move under the
 * assumptions nobody can check here"; none of them is the contract the HIP path is held to):
 *   -DORC_ACCUM_T        everything libpointmatcher's PointMatcher<T> does in T is done in `real`: the normal equations
 *                        A = wF F^T, b = -(wF dot^T) and the residual accumulated in `real` in pair (column) order, the
 *                        Cholesky solve in `real`, the AngleAxis increment in `real`, T_iter = dT * T_iter, the pre- and
 *                        post-multiplication with T_refIn_refMean in `real` ([EXT] ICP.cpp computeWithTransformedReference
 *                        keeps TransformationParameters = Matrix of T), the reference mean as 8 running `real` lane sums
 *                        (an Eigen packet reduction).  The checkers see the `real`-rounded T_iter (their own quaternion
 *                        arithmetic stays double).
 *   -DORC_TIE_HIGH       among candidates at exactly the same squared distance the HIGHEST index wins (libnabo's order is
 *                        traversal dependent; lowest / highest index are its two extremes).
 *   -DORC_FMA_TRANSFORM  the rigid transform contracts to fused multiply-adds, as an Eigen 4x4 * 4xN product compiled with
 *                        FMA enabled would: fma(r02, z, fma(r01, y, r00 * x)) + tx.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#ifdef ORC_DOUBLE
typedef double real;
#define REAL_EPS DBL_EPSILON
#define FN(name) name##_f64
#else
typedef float real;
#define REAL_EPS FLT_EPSILON
#define FN(name) name##_f32
#endif

#ifdef ORC_DOUBLE
#define FMA_R(a, b, c) fma((a), (b), (c))
#define SQRT_R(a) sqrt(a)
#define COS_R(a) cos(a)
#define SIN_R(a) sin(a)
#define EXP_R(a) exp(a)
#else
#define FMA_R(a, b, c) fmaf((a), (b), (c))
#define SQRT_R(a) sqrtf(a)
#define COS_R(a) cosf(a)
#define SIN_R(a) sinf(a)
#define EXP_R(a) expf(a)
#endif
/* which index wins among candidates at exactly the same squared distance (see ORC_TIE_HIGH above) */
#ifdef ORC_TIE_HIGH
#define TIE_WINS(j, bi) ((j) > (bi))
#define TIE_NONE (-1)                 /* "no candidate yet": loses every tie */
#else
#define TIE_WINS(j, bi) ((j) < (bi))
#define TIE_NONE INT32_MAX
#endif

#define ORC_OK 0
#define ORC_ERR_NO_MATCH 1      /* "no outlier to filter" / "no point to minimize" (ConvergenceError) */
#define ORC_ERR_NAN 2
#define ORC_ERR_ARG 3
#define ORC_ERR_BOUND 7         /* BoundTransformationChecker: limit exceeded (ConvergenceError) */

typedef struct {
    real max_dist;          /* KDTreeMatcher.maxDist (may be +inf) */
    real trim_ratio;        /* TrimmedDistOutlierFilter.ratio */
    int max_iters;          /* CounterTransformationChecker.maxIterationCount */
    double min_diff_rot;    /* DifferentialTransformationChecker.minDiffRotErr */
    double min_diff_trans;  /* ....minDiffTransErr */
    int smooth_length;      /* ....smoothLength */
    double sensor_std_dev;  /* PointToPlaneWithCovErrorMinimizer.sensorStdDev */
    int use_kdtree;         /* 0: brute force (ground truth), 1: kd-tree (same results) */
    int center_reference;   /* 1: subtract centroid as ICP::operator()/setMap do [A.2] */
    real outlier_max_dist;  /* MaxDistOutlierFilter.maxDist, a second filter of the chain [A.4]; <= 0 or +inf: absent */
    real quantile_scale;    /* [A.4] MedianDistOutlierFilter.factor: with trim_ratio = 0.5 the chain's quantile filter is the
                             * median filter -- limit = factor * getDistsQuantile(0.5), weight = (dist <= limit), all on
                             * SQUARED distances, as [EXT] OutlierFiltersImpl.cpp writes it; 1 (or <= 0) = TrimmedDist */
    /* ---- round 4: the rest of the chain slots a pgslam user's YAML may fill (Localizer.hpp:70, LoopCloser.hpp:73) ---- */
    int knn;                /* [A.3] KDTreeMatcher.knn: neighbours per reading point (<= 1: one).  Matches are knn x N; every
                             * later stage sees knn * N pairs */
    int minimizer;          /* 0: PointToPlane(WithCov)ErrorMinimizer [A.6]; 1: PointToPointErrorMinimizer (Kabsch / SVD);
                             * 2: PointToPlane with force4DOF (orc_solve_4dof); 3: PointToPointWithCovErrorMinimizer -- the solve of 1 and
                             * [EXT] ErrorMinimizers/PointToPointWithCov.cpp's estimateCovariance, which is PointToPlaneWithCov's body: the
                             * same Censi sums over the kept pairs, read from the reference's `normals` (A.8) */
    double bound_max_rot;   /* [A.9] BoundTransformationChecker.maxRotationNorm (rad); <= 0 or inf: not in the chain */
    double bound_max_trans; /* [A.9] BoundTransformationChecker.maxTranslationNorm; <= 0 or inf: not in the chain */
    real normal_max_angle;  /* [A.4] SurfaceNormalOutlierFilter.maxAngle (rad); <= 0: not in the chain */
    /* ---- round 5: [A.4] RobustOutlierFilter{robustFct, tuning, scaleEstimator, approximation} (distanceType point2point,
     * nbIterationForScale 0); orc_robust_weights below ---- */
    int robust_fct;         /* 0: not in the chain; 1 cauchy 2 welsch 3 sc 4 gm 5 tukey 6 huber 7 L1 */
    real robust_tuning;     /* tuning */
    int robust_scale;       /* scaleEstimator: 0 none, 1 mad */
    real robust_approx;     /* approximation (a distance; its square cuts e2); <= 0 or +inf: none */
    /* ---- round 6: the order the pairs enter the reduction tree in (FN(tree_sum) below).  NULL: scan order.  Otherwise a
     * permutation of the reading's points -- position j of the tree holds point pair_order[j] -- e.g. the one the HIP path
     * sorts a reading by (pgicp_debug_reading_order): a sum does not depend on it mathematically, its last bits do ---- */
    const int *pair_order;
} FN(orc_params);

/* [A.4] the chain multiplies the weights of its outlier filters.  MaxDistOutlierFilter: weight 1 while the SQUARED
 * distance is <= maxDist * maxDist, 0 beyond; the trimmed filter's threshold is not affected by it (each filter sees
 * all the matches). */
static void FN(orc_maxdist_weights)(const real *d2, int n, real max_dist, real *w)
{
    if (!(max_dist > (real)0) || isinf(max_dist)) return;
    const real lim = max_dist * max_dist;
    for (int i = 0; i < n; i++) if (!(d2[i] <= lim)) w[i] = (real)0;
}

typedef struct {
    int status;
    int iterations;
    int converged;            /* Differential checker stopped the loop */
    int max_iter_reached;     /* Counter checker stopped the loop */
    double overlap;           /* weightedPointUsedRatio of the last iteration */
    double residual;          /* sum w (n.(p-q))^2 of the last iteration (pre-update) */
    double trim_limit;        /* last trim threshold (squared distance) */
    int n_kept;
    int n_finite;
    double cov[36];           /* Censi covariance, order [x y z rx ry rz] */
} orc_result;

/* --------------------------------------------------------------------------
 * small dense helpers (double, row-major 4x4 / 6x6)
 * ------------------------------------------------------------------------ */
__attribute__((unused)) static void mat4_mul(const double *a, const double *b, double *c)
{
    double t[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
            t[i * 4 + j] = s;
        }
    memcpy(c, t, sizeof t);
}

/* a product of two TransformationParameters of the ICP loop: double by contract; Matrix of T under ORC_ACCUM_T */
static void mat4_mul_chain(const double *a, const double *b, double *c)
{
#ifdef ORC_ACCUM_T
    double t[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            real s = (real)0;
            for (int k = 0; k < 4; k++) s += (real)a[i * 4 + k] * (real)b[k * 4 + j];
            t[i * 4 + j] = (double)s;
        }
    memcpy(c, t, sizeof t);
#else
    mat4_mul(a, b, c);
#endif
}

static void mat4_identity(double *a)
{
    memset(a, 0, 16 * sizeof(double));
    a[0] = a[5] = a[10] = a[15] = 1.0;
}

static void mat4_rigid_inverse(const double *t, double *o)
{
    double r[16];
    mat4_identity(r);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r[i * 4 + j] = t[j * 4 + i];
    for (int i = 0; i < 3; i++)
        r[i * 4 + 3] = -(r[i * 4 + 0] * t[3] + r[i * 4 + 1] * t[7] + r[i * 4 + 2] * t[11]);
    memcpy(o, r, sizeof r);
}

/* [a9] RigidTransformation::compute: features' = T*features; normals' = R*normals. */
void FN(orc_transform)(const double *T, const real *in, real *out, int n, int rotate_only)
{
    const real r00 = (real)T[0], r01 = (real)T[1], r02 = (real)T[2], tx = (real)T[3];
    const real r10 = (real)T[4], r11 = (real)T[5], r12 = (real)T[6], ty = (real)T[7];
    const real r20 = (real)T[8], r21 = (real)T[9], r22 = (real)T[10], tz = (real)T[11];
    for (int i = 0; i < n; i++) {
        const real x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
#ifdef ORC_FMA_TRANSFORM
        real ox = FMA_R(r02, z, FMA_R(r01, y, r00 * x));
        real oy = FMA_R(r12, z, FMA_R(r11, y, r10 * x));
        real oz = FMA_R(r22, z, FMA_R(r21, y, r20 * x));
#else
        real ox = (r00 * x + r01 * y) + r02 * z;
        real oy = (r10 * x + r11 * y) + r12 * z;
        real oz = (r20 * x + r21 * y) + r22 * z;
#endif
        if (!rotate_only) { ox = ox + tx; oy = oy + ty; oz = oz + tz; }
        out[3 * i] = ox; out[3 * i + 1] = oy; out[3 * i + 2] = oz;
    }
}

/* [A.2] centroid of the reference, order-independent fixed-point definition. */
void FN(orc_centroid)(const real *xyz, int n, real *mean)
{
#ifdef ORC_ACCUM_T
    /* reference.features.rowwise().sum() / N in T: eight running lane sums per row, folded at the end */
    for (int a = 0; a < 3; a++) {
        real lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; i++) lane[i & 7] += xyz[3 * i + a];
        const real s = ((lane[0] + lane[4]) + (lane[2] + lane[6])) + ((lane[1] + lane[5]) + (lane[3] + lane[7]));
        mean[a] = s / (real)n;
    }
    return;
#endif
    for (int a = 0; a < 3; a++) {
        int64_t s = 0;
        for (int i = 0; i < n; i++) s += llrint((double)xyz[3 * i + a] * 16777216.0);
        mean[a] = (real)(((double)s / 16777216.0) / (double)n);
    }
}

/* [a12] LocalMap::BuildCloudFromData: out = ref_kf cloud ++ T_k * cloud_k ... */
void FN(orc_build_local_map)(int n_kf, const real *const *xyz, const real *const *nrm, const int *counts,
                             const double *T_ref_kf /* n_kf x 16, entry 0 ignored (identity) */,
                             real *out_xyz, real *out_nrm)
{
    int off = 0;
    for (int k = 0; k < n_kf; k++) {
        if (k == 0) {
            memcpy(out_xyz + 3 * off, xyz[k], sizeof(real) * 3 * counts[k]);
            memcpy(out_nrm + 3 * off, nrm[k], sizeof(real) * 3 * counts[k]);
        } else {
            FN(orc_transform)(T_ref_kf + 16 * k, xyz[k], out_xyz + 3 * off, counts[k], 0);
            FN(orc_transform)(T_ref_kf + 16 * k, nrm[k], out_nrm + 3 * off, counts[k], 1);
        }
        off += counts[k];
    }
}

/* --------------------------------------------------------------------------
 * Input filters that only drop points -- input_filters_.apply(cloud) at Localizer.hpp:103, the reading / reference filters at
 * Localizer.hpp:314-326 -- restated from [EXT] libpointmatcher DataPointsFilters/{MaxDist,MinDist,BoundingBox,RemoveNaN,
 * FixStepSampling,RandomSampling,MaxPointCount}.cpp.  `feat`: n points of `frows` values (xyz first, then the homogeneous row).
 * The filters run in list order, each on what the one before it kept; kept_idx receives the surviving input indices.
 *   type 1 MaxDist {maxDist = p[0], dim = p[1] - 1}: dim -1: keep when features.col(i).head(3).norm() < |maxDist| (the norm in T:
 *          sqrt((x x + y y) + z z)); dim 0..2: keep when features(dim, i) < maxDist
 *   type 2 MinDist {minDist, dim}: norm > |minDist|; features(dim, i) > minDist       (strict both; a NaN fails either)
 *   type 3 BoundingBox {xMin yMin zMin xMax yMax zMax removeInside}: inside = every coordinate strictly between its bounds
 *   type 4 RemoveNaN: drops a point when any FEATURE value (the padding row too) is NaN
 *   type 5 FixStepSampling {step = p[0]}: points phase, phase + step, ...; upstream draws phase = rand() % step (the C library's
 *          global generator: nothing to be bit-exact against) -- phase 0 here and in the build; the step's evolution over calls
 *          (startStep, endStep, stepMult) is the caller's state, see orc_fixstep_next
 *   type 6 RandomSampling {prob = p[0]}: upstream keeps when rand() / RAND_MAX < prob; here the build's counter-based draw
 *          (SplitMix64 of seed p[1] and the point's index) -- same distribution, documented as NOT rand()-parity
 *   type 7 MaxPointCount {maxCount = p[0], seed = p[1]}: only when maxCount < N: the same draw with prob = T(maxCount) / T(N)
 * ------------------------------------------------------------------------ */
/* [EXT] ShadowDataPointsFilter{eps} (DataPointsFilters/Shadow.cpp, as recalled): a point is KEPT when
 *   | normals.col(i).normalized() . features.col(i).head(3).normalized() | > sin(eps)
 * -- a surface seen at a grazing angle (the normal nearly perpendicular to the ray from the sensor's origin, where the cloud
 * still is when the input filters run) is a "shadow" and goes.  normalized() = v / sqrt(v.v) when v.v > 0, v itself otherwise
 * (Eigen); the dot product and the squared norms are summed in order, in T; sin() in T from the parameter read as T.
 * keep[i] = 1 / 0.  Host side in the build (pointmatcher.hpp: ShadowDataPointsFilter): it needs the normals descriptor. */
void FN(orc_shadow_keep)(const real *xyz, const real *nrm, int n, real eps_angle, int *keep)
{
    const real eps = SIN_R(eps_angle);
    for (int i = 0; i < n; i++) {
        real a[3] = {nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]}, b[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        const real za = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2], zb = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2];
        if (za > (real)0) { const real sa = SQRT_R(za); a[0] = a[0] / sa; a[1] = a[1] / sa; a[2] = a[2] / sa; }
        if (zb > (real)0) { const real sb = SQRT_R(zb); b[0] = b[0] / sb; b[1] = b[1] / sb; b[2] = b[2] / sb; }
        const real d = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
        keep[i] = (d < (real)0 ? -d : d) > eps ? 1 : 0;
    }
}

static uint64_t orc_splitmix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
int FN(orc_filter_chain)(int n_filters, const int *types, const double *params /* 8 per filter */, const real *feat, int frows, int n,
                         int *kept_idx, int *n_out)
{
    int cur = n;
    for (int i = 0; i < n; i++) kept_idx[i] = i;
    for (int k = 0; k < n_filters; k++) {
        const double *p = params + 8 * k;
        const int t = types[k];
        int out = 0;
        for (int j = 0; j < cur; j++) {
            const real *q = feat + (size_t)kept_idx[j] * frows;
            int keep = 1;
            if (t == 1 || t == 2) {
                const int dim = (int)p[1] - 1;
                const real lim = (real)p[0];
                if (dim < 0) {
                    const real r = SQRT_R((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]), al = lim < (real)0 ? -lim : lim;
                    keep = t == 1 ? r < al : r > al;
                } else keep = t == 1 ? q[dim] < lim : q[dim] > lim;
            } else if (t == 3) {
                const int in = (real)p[0] < q[0] && q[0] < (real)p[3] && (real)p[1] < q[1] && q[1] < (real)p[4] && (real)p[2] < q[2] && q[2] < (real)p[5];
                keep = in != (p[6] != 0.0);
            } else if (t == 4) {
                for (int r = 0; r < frows; r++) if (q[r] != q[r]) keep = 0;
            } else if (t == 5) {
                keep = j % (int)p[0] == 0;
            } else if (t == 6 || t == 7) {
                double prob = p[0];
                if (t == 7) prob = (double)cur > p[0] ? (double)((real)p[0] / (real)cur) : 2.0;
                const double u = (double)(orc_splitmix((uint64_t)p[1] * 0x100000001B3ULL + (uint64_t)j) >> 11) / 9007199254740992.0;
                keep = u < prob;
            } else if (t != 0) return ORC_ERR_ARG;
            if (keep) kept_idx[out++] = kept_idx[j];
        }
        cur = out;
    }
    *n_out = cur;
    return ORC_OK;
}
/* [EXT] FixStepSampling.cpp, the end of inPlaceFilter: step *= stepMult, clamped at endStep in the direction of travel */
#ifndef ORC_DOUBLE
double orc_fixstep_next(double step, double start_step, double end_step, double step_mult)
{
    const double delta = start_step * step_mult - start_step;
    step *= step_mult;
    if (delta < 0 && step < end_step) step = end_step;
    if (delta > 0 && step > end_step) step = end_step;
    return step;
}
#endif

/* --------------------------------------------------------------------------
 * [A.3] matcher: brute force (ground truth) and kd-tree (identical results)
 * ------------------------------------------------------------------------ */
static inline real dist2(const real *q, const real *m)
{
    const real dx = q[0] - m[0], dy = q[1] - m[1], dz = q[2] - m[2];
    return (dx * dx + dy * dy) + dz * dz;
}

void FN(orc_knn_brute)(const real *q, int nq, const real *m, int nm, real max_dist, int *ids, real *d2)
{
    const real md2 = max_dist * max_dist;
    for (int i = 0; i < nq; i++) {
        real best = INFINITY;
        int bi = -1;
        for (int j = 0; j < nm; j++) {
            const real d = dist2(q + 3 * i, m + 3 * j);
#ifdef ORC_TIE_HIGH
            if (d <= best) { best = d; bi = j; }     /* the other extreme: highest index wins ties */
#else
            if (d < best) { best = d; bi = j; }      /* strict <: lowest index wins ties */
#endif
        }
        if (bi >= 0 && best <= md2) { ids[i] = bi; d2[i] = best; }
        else { ids[i] = -1; d2[i] = INFINITY; }
    }
}

typedef struct {
    int n;
    const real *pts;       /* borrowed */
    int *perm;             /* point order inside the tree */
    int n_nodes;
    int *node_lo, *node_hi;    /* leaf: [lo,hi) into perm */
    int *node_axis;            /* -1 for leaves */
    real *node_split;
    int *node_left, *node_right;
} kdtree;

#define KD_LEAF 12

static int kd_build_rec(kdtree *t, int lo, int hi)
{
    const int id = t->n_nodes++;
    t->node_lo[id] = lo; t->node_hi[id] = hi;
    t->node_left[id] = t->node_right[id] = -1;
    if (hi - lo <= KD_LEAF) { t->node_axis[id] = -1; t->node_split[id] = 0; return id; }
    real mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = lo; i < hi; i++)
        for (int a = 0; a < 3; a++) {
            const real v = t->pts[3 * t->perm[i] + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    int ax = 0;
    if (mx[1] - mn[1] > mx[ax] - mn[ax]) ax = 1;
    if (mx[2] - mn[2] > mx[ax] - mn[ax]) ax = 2;
    if (!(mx[ax] > mn[ax])) { t->node_axis[id] = -1; t->node_split[id] = 0; return id; }
    /* quickselect median on axis */
    int l = lo, r = hi - 1;
    const int k = lo + (hi - lo) / 2;
    while (l < r) {
        const real pv = t->pts[3 * t->perm[(l + r) / 2] + ax];
        int i = l, j = r;
        while (i <= j) {
            while (t->pts[3 * t->perm[i] + ax] < pv) i++;
            while (t->pts[3 * t->perm[j] + ax] > pv) j--;
            if (i <= j) { int tmp = t->perm[i]; t->perm[i] = t->perm[j]; t->perm[j] = tmp; i++; j--; }
        }
        if (k <= j) r = j; else if (k >= i) l = i; else break;
    }
    /* split value: everything in [lo,k) <= split <= everything in [k,hi) */
    real split = t->pts[3 * t->perm[k] + ax];
    t->node_axis[id] = ax; t->node_split[id] = split;
    const int left = kd_build_rec(t, lo, k);
    const int right = kd_build_rec(t, k, hi);
    t->node_left[id] = left; t->node_right[id] = right;
    return id;
}

void *FN(orc_kdtree_build)(const real *m, int nm)
{
    kdtree *t = (kdtree *)calloc(1, sizeof *t);
    t->n = nm; t->pts = m;
    t->perm = (int *)malloc(sizeof(int) * (nm > 0 ? nm : 1));
    for (int i = 0; i < nm; i++) t->perm[i] = i;
    const int cap = 2 * (nm / (KD_LEAF / 2) + 2) + 64;   /* leaves hold >= KD_LEAF/2 points */
    t->node_lo = (int *)malloc(sizeof(int) * cap); t->node_hi = (int *)malloc(sizeof(int) * cap);
    t->node_axis = (int *)malloc(sizeof(int) * cap); t->node_split = (real *)malloc(sizeof(real) * cap);
    t->node_left = (int *)malloc(sizeof(int) * cap); t->node_right = (int *)malloc(sizeof(int) * cap);
    t->n_nodes = 0;
    if (nm > 0) kd_build_rec(t, 0, nm);
    return t;
}

void FN(orc_kdtree_free)(void *h)
{
    kdtree *t = (kdtree *)h;
    if (!t) return;
    free(t->perm); free(t->node_lo); free(t->node_hi); free(t->node_axis);
    free(t->node_split); free(t->node_left); free(t->node_right); free(t);
}

static void kd_search(const kdtree *t, int node, const real *q, real *best, int *bi)
{
    const int ax = t->node_axis[node];
    if (ax < 0) {
        for (int i = t->node_lo[node]; i < t->node_hi[node]; i++) {
            const int j = t->perm[i];
            const real d = dist2(q, t->pts + 3 * j);
            if (d < *best || (d == *best && TIE_WINS(j, *bi))) { *best = d; *bi = j; }
        }
        return;
    }
    const real diff = q[ax] - t->node_split[node];
    const int near = diff < 0 ? t->node_left[node] : t->node_right[node];
    const int far = diff < 0 ? t->node_right[node] : t->node_left[node];
    kd_search(t, near, q, best, bi);
    /* conservative prune: rounding is monotone, so every point beyond the
     * plane has computed d2 >= fl(diff*diff); descend on equality (ties). */
    if (!(diff * diff > *best)) kd_search(t, far, q, best, bi);
}

void FN(orc_kdtree_knn)(const void *h, const real *q, int nq, real max_dist, int *ids, real *d2)
{
    const kdtree *t = (const kdtree *)h;
    const real md2 = max_dist * max_dist;
    for (int i = 0; i < nq; i++) {
        real best = INFINITY;
        int bi = -1;
        if (t->n > 0) {
            /* seed with maxDist so far subtrees are pruned early; keep exactness:
             * anything with d2 > md2 is rejected below anyway. */
            best = md2; bi = TIE_NONE;
            kd_search(t, 0, q + 3 * i, &best, &bi);
            if (bi == TIE_NONE) bi = -1;
        }
        if (bi >= 0 && best <= md2) { ids[i] = bi; d2[i] = best; }
        else { ids[i] = -1; d2[i] = INFINITY; }
    }
}

/* --------------------------------------------------------------------------
 * k nearest neighbours (k > 1) -- libnabo's KDTreeMatcher with knn = k as
 * SurfaceNormalDataPointsFilter uses it on the cloud itself.  Per query the k
 * best (d2, index) pairs in lexicographic order; missing ones are -1 / +inf.
 * ------------------------------------------------------------------------ */
typedef struct { real *d; int *i; int k; } kbest;

static inline void kbest_push(kbest *b, real d, int j)
{
    const int k = b->k;
    if (!(d < b->d[k - 1] || (d == b->d[k - 1] && TIE_WINS(j, b->i[k - 1])))) return;
    int p = k - 1;
    while (p > 0 && (d < b->d[p - 1] || (d == b->d[p - 1] && TIE_WINS(j, b->i[p - 1])))) { b->d[p] = b->d[p - 1]; b->i[p] = b->i[p - 1]; --p; }
    b->d[p] = d; b->i[p] = j;
}

static void kd_search_k(const kdtree *t, int node, const real *q, kbest *b)
{
    const int ax = t->node_axis[node];
    if (ax < 0) {
        for (int i = t->node_lo[node]; i < t->node_hi[node]; i++) {
            const int j = t->perm[i];
            kbest_push(b, dist2(q, t->pts + 3 * j), j);
        }
        return;
    }
    const real diff = q[ax] - t->node_split[node];
    const int near = diff < 0 ? t->node_left[node] : t->node_right[node];
    const int far = diff < 0 ? t->node_right[node] : t->node_left[node];
    kd_search_k(t, near, q, b);
    if (!(diff * diff > b->d[b->k - 1])) kd_search_k(t, far, q, b);     /* same conservative prune as kd_search */
}

void FN(orc_kdtree_knn_k)(const void *h, const real *q, int nq, int k, real max_dist, int *ids, real *d2)
{
    const kdtree *t = (const kdtree *)h;
    const real md2 = max_dist * max_dist;
    for (int i = 0; i < nq; i++) {
        kbest b = { d2 + (size_t)i * k, ids + (size_t)i * k, k };
        for (int j = 0; j < k; j++) { b.d[j] = md2; b.i[j] = TIE_NONE; }   /* anything beyond maxDist is useless */
        if (t->n > 0) kd_search_k(t, 0, q + 3 * i, &b);
        for (int j = 0; j < k; j++)
            if (b.i[j] == TIE_NONE || !(b.d[j] <= md2)) { b.i[j] = -1; b.d[j] = INFINITY; }
    }
}

/* brute force with knn > 1: the ground truth of [A.3] for the k-d tree version above (same order, same sentinels) */
void FN(orc_knn_brute_k)(const real *q, int nq, const real *m, int nm, int k, real max_dist, int *ids, real *d2)
{
    const real md2 = max_dist * max_dist;
    for (int i = 0; i < nq; i++) {
        kbest b = { d2 + (size_t)i * k, ids + (size_t)i * k, k };
        for (int j = 0; j < k; j++) { b.d[j] = md2; b.i[j] = TIE_NONE; }
        for (int j = 0; j < nm; j++) kbest_push(&b, dist2(q + 3 * i, m + 3 * j), j);
        for (int j = 0; j < k; j++)
            if (b.i[j] == TIE_NONE || !(b.d[j] <= md2)) { b.i[j] = -1; b.d[j] = INFINITY; }
    }
}

/* cyclic Jacobi eigen-decomposition of a symmetric 3x3 (double): a -> diag(ev), columns of v */
static void jacobi3(double a[3][3], double v[3][3], double ev[3])
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 16; sweep++) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off == 0.0) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int r = 0; r < 3; r++) {                 /* A <- A J */
                    const double arp = a[r][p], arq = a[r][q];
                    a[r][p] = c * arp - s * arq; a[r][q] = s * arp + c * arq;
                }
                for (int r = 0; r < 3; r++) {                 /* A <- J^T A */
                    const double apr = a[p][r], aqr = a[q][r];
                    a[p][r] = c * apr - s * aqr; a[q][r] = s * apr + c * aqr;
                }
                for (int r = 0; r < 3; r++) {
                    const double vrp = v[r][p], vrq = v[r][q];
                    v[r][p] = c * vrp - s * vrq; v[r][q] = s * vrp + c * vrq;
                }
            }
    }
    for (int i = 0; i < 3; i++) ev[i] = a[i][i];
}

/* --------------------------------------------------------------------------
 * [EXT] SurfaceNormalDataPointsFilter{knn, maxDist, epsilon = 0, keepNormals, keepEigenValues}
 * (libpointmatcher, version unpinned; applied by pgslam at Localizer.hpp:103 / 314-326 when the
 * user's YAML asks for it).  Per point: its knn nearest neighbours in the cloud itself (the point
 * included), mean and scatter matrix C = sum (p - mean)(p - mean)^T accumulated in T in
 * neighbour order, eigen-decomposition, normal = unit eigenvector of the smallest eigenvalue
 * (the sign is the solver's; point-to-plane ICP is invariant to it).  A neighbourhood whose
 * scatter has rank < 2 keeps libpointmatcher's defaults (eigenvalues (1,0,0), eigenvectors I),
 * which makes the normal (0,1,0) -- [EXT, unverified here].  Eigenvalues are returned ascending
 * (sortEigen = 1).  The eigen solver is cyclic Jacobi in double, not Eigen's EigenSolver.
 * ------------------------------------------------------------------------ */
int FN(orc_surface_normals)(const real *xyz, int n, int knn, real max_dist, real *nrm, real *eigval, int *ids_out, real *d2_out)
{
    if (n <= 0 || knn < 1) return ORC_ERR_ARG;
    void *tree = FN(orc_kdtree_build)(xyz, n);
    int *ids = ids_out ? ids_out : (int *)malloc(sizeof(int) * (size_t)n * knn);
    real *d2 = d2_out ? d2_out : (real *)malloc(sizeof(real) * (size_t)n * knn);
    FN(orc_kdtree_knn_k)(tree, xyz, n, knn, max_dist, ids, d2);
    for (int i = 0; i < n; i++) {
        const int *nb = ids + (size_t)i * knn;
        int cnt = 0;
        real sx = 0, sy = 0, sz = 0;
        for (int j = 0; j < knn; j++)
            if (nb[j] >= 0) { sx += xyz[3 * nb[j]]; sy += xyz[3 * nb[j] + 1]; sz += xyz[3 * nb[j] + 2]; cnt++; }
        real c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
        if (cnt > 0) {
            const real mx = sx / (real)cnt, my = sy / (real)cnt, mz = sz / (real)cnt;
            for (int j = 0; j < knn; j++)
                if (nb[j] >= 0) {
                    const real dx = xyz[3 * nb[j]] - mx, dy = xyz[3 * nb[j] + 1] - my, dz = xyz[3 * nb[j] + 2] - mz;
                    c00 += dx * dx; c01 += dx * dy; c02 += dx * dz; c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;
                }
        }
        double A[3][3] = {{c00, c01, c02}, {c01, c11, c12}, {c02, c12, c22}}, V[3][3], ev[3];
        jacobi3(A, V, ev);
        int lo = 0, hi = 0;
        for (int k = 1; k < 3; k++) { if (ev[k] < ev[lo]) lo = k; if (ev[k] > ev[hi]) hi = k; }
        const int mid = 3 - lo - hi;
        /* rank >= 2  <=>  the middle eigenvalue is not negligible against the largest */
        const double tol = 3.0 * (double)REAL_EPS;
        const int degenerate = lo == hi || !(ev[hi] > 0.0) || !(ev[mid] > tol * ev[hi]);
        if (degenerate) {
            nrm[3 * i] = 0; nrm[3 * i + 1] = 1; nrm[3 * i + 2] = 0;
            if (eigval) { eigval[3 * i] = 0; eigval[3 * i + 1] = 0; eigval[3 * i + 2] = 1; }
        } else {
            nrm[3 * i] = (real)V[0][lo]; nrm[3 * i + 1] = (real)V[1][lo]; nrm[3 * i + 2] = (real)V[2][lo];
            if (eigval) { eigval[3 * i] = (real)ev[lo]; eigval[3 * i + 1] = (real)ev[mid]; eigval[3 * i + 2] = (real)ev[hi]; }
        }
    }
    FN(orc_kdtree_free)(tree);
    if (!ids_out) free(ids);
    if (!d2_out) free(d2);
    return ORC_OK;
}

/* --------------------------------------------------------------------------
 * [EXT] SamplingSurfaceNormalDataPointsFilter{ratio, knn, samplingMethod, maxBoxDim, averageExistingDescriptors,
 * keepNormals, ...} (libpointmatcher DataPointsFilters/SamplingSurfaceNormal.cpp, version unpinned, restated from its
 * published source; upstream's ICP::setDefault() installs it with its defaults -- ratio 0.5, knn 7 -- as THE reference filter,
 * and it is how most point-to-plane configurations get their normals: Localizer.hpp:73-78, 314-315 apply whatever the YAML names).
 *   buildNew(first, last, min, max): a range of at most knn points is a BOX (fuseRange); otherwise it is cut at the median of
 *     its widest dimension -- argmax(max - min) over the FEATURE rows, first maximum, the bounds being the cloud's bounding box
 *     narrowed by the cut values on the way down --: rightCount = count / 2, leftCount = count - rightCount,
 *     std::nth_element at first + leftCount, cutVal = that element's coordinate; recursion left then right.
 *   fuseRange: box = max - min of the points; dropped whole when its largest side exceeds maxBoxDim; mean = sum / count and
 *     C = NN NN^T of the mean-subtracted points, in T; dropped whole when rank(C) + 1 < 3 (a box of collinear or identical
 *     points: "no noise in data"); normal = eigenvector of the smallest eigenvalue (the solver's sign).
 *     samplingMethod 0: every point of the box is kept with probability ratio (upstream: rand() / RAND_MAX < ratio) and gets the
 *       box's normal; samplingMethod 1: ONE point per box -- the first index of the range -- moved to the box's mean.
 *   The kept indices are sorted and the cloud compacted in that order.
 * What is NOT upstream's, and cannot be (nothing to be bit-exact against):
 *   - std::nth_element leaves the order inside the two halves, and which of several points AT the cut value fall left, to the
 *     C++ library: here a range is cut by a full sort on (coordinate, index) -- one of the arrangements nth_element may produce --
 *     so a box's members are determined except among equal coordinates, and "the first index of the range" (method 1) is the
 *     point smallest along the last cut;
 *   - rand(): the build's counter-based draw (SplitMix64 of the seed and the point's ORIGINAL index), as in RandomSampling;
 *   - Eigen's EigenSolver: cyclic Jacobi in double on the T-accumulated scatter, rank test as in orc_surface_normals.
 * keep[i] = 1 / 0 per input point; nrm (3 per input point) is written for kept points; xyz_out (3 per input point, or NULL):
 * the kept point's coordinates (its own, or the box mean with method 1).  Returns the number of boxes that were fused.
 * ------------------------------------------------------------------------ */
typedef struct { const real *xyz; int cut; } FN(ssn_cmp_ctx);
static _Thread_local FN(ssn_cmp_ctx) FN(ssn_cmp);
static int FN(ssn_compare)(const void *a, const void *b)
{
    const int i = *(const int *)a, j = *(const int *)b;
    const real x = FN(ssn_cmp).xyz[3 * i + FN(ssn_cmp).cut], y = FN(ssn_cmp).xyz[3 * j + FN(ssn_cmp).cut];
    if (x < y) return -1;
    if (x > y) return 1;
    return (i > j) - (i < j);
}
typedef struct {
    const real *xyz; int *idx; int knn, method; real ratio, max_box; uint64_t seed;
    int *keep; real *nrm, *xyz_out; int boxes;
} FN(ssn_state);
static void FN(ssn_fuse)(FN(ssn_state) *S, int first, int last)
{
    const int cnt = last - first;
    const real *X = S->xyz;
    real lo[3], hi[3], sum[3] = {0, 0, 0};
    for (int a = 0; a < 3; a++) { lo[a] = X[3 * S->idx[first] + a]; hi[a] = lo[a]; }
    for (int k = first; k < last; k++)
        for (int a = 0; a < 3; a++) {
            const real v = X[3 * S->idx[k] + a];
            if (v < lo[a]) lo[a] = v;
            if (v > hi[a]) hi[a] = v;
            sum[a] += v;
        }
    real box = hi[0] - lo[0];
    if (hi[1] - lo[1] > box) box = hi[1] - lo[1];
    if (hi[2] - lo[2] > box) box = hi[2] - lo[2];
    if (box > S->max_box) return;
    const real mx = sum[0] / (real)cnt, my = sum[1] / (real)cnt, mz = sum[2] / (real)cnt;
    real c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    for (int k = first; k < last; k++) {
        const int i = S->idx[k];
        const real dx = X[3 * i] - mx, dy = X[3 * i + 1] - my, dz = X[3 * i + 2] - mz;
        c00 += dx * dx; c01 += dx * dy; c02 += dx * dz; c11 += dy * dy; c12 += dy * dz; c22 += dz * dz;
    }
    double A[3][3] = {{c00, c01, c02}, {c01, c11, c12}, {c02, c12, c22}}, V[3][3], ev[3];
    jacobi3(A, V, ev);
    int l = 0, h = 0;
    for (int k = 1; k < 3; k++) { if (ev[k] < ev[l]) l = k; if (ev[k] > ev[h]) h = k; }
    const int mid = 3 - l - h;
    const double tol = 3.0 * (double)REAL_EPS;
    if (l == h || !(ev[h] > 0.0) || !(ev[mid] > tol * ev[h])) return;      /* rank < 2: the box is dropped */
    S->boxes++;
    const real n0 = (real)V[0][l], n1 = (real)V[1][l], n2 = (real)V[2][l];
    if (S->method == 0) {
        for (int k = first; k < last; k++) {
            const int i = S->idx[k];
            const double u = (double)(orc_splitmix(S->seed * 0x100000001B3ULL + (uint64_t)i) >> 11) / 9007199254740992.0;
            if (!(u < (double)S->ratio)) continue;
            S->keep[i] = 1;
            S->nrm[3 * i] = n0; S->nrm[3 * i + 1] = n1; S->nrm[3 * i + 2] = n2;
            if (S->xyz_out) { S->xyz_out[3 * i] = X[3 * i]; S->xyz_out[3 * i + 1] = X[3 * i + 1]; S->xyz_out[3 * i + 2] = X[3 * i + 2]; }
        }
    } else {
        const int i = S->idx[first];
        S->keep[i] = 1;
        S->nrm[3 * i] = n0; S->nrm[3 * i + 1] = n1; S->nrm[3 * i + 2] = n2;
        if (S->xyz_out) { S->xyz_out[3 * i] = mx; S->xyz_out[3 * i + 1] = my; S->xyz_out[3 * i + 2] = mz; }
    }
}
static void FN(ssn_build)(FN(ssn_state) *S, int first, int last, const real *lo, const real *hi)
{
    const int count = last - first;
    if (count <= S->knn) { FN(ssn_fuse)(S, first, last); return; }
    int cut = 0;                                            /* argMax(max - min), first maximum */
    for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[cut] - lo[cut]) cut = a;
    const int right = count / 2, left = count - right;
    FN(ssn_cmp).xyz = S->xyz; FN(ssn_cmp).cut = cut;
    qsort(S->idx + first, (size_t)count, sizeof(int), FN(ssn_compare));
    const real cut_val = S->xyz[3 * S->idx[first + left] + cut];
    real lhi[3] = {hi[0], hi[1], hi[2]}, rlo[3] = {lo[0], lo[1], lo[2]};
    lhi[cut] = cut_val; rlo[cut] = cut_val;
    FN(ssn_build)(S, first, first + left, lo, lhi);
    FN(ssn_build)(S, first + left, last, rlo, hi);
}
int FN(orc_sampling_surface_normal)(const real *xyz, int n, int knn, real ratio, int sampling_method, real max_box_dim, double seed,
                                    int *keep, real *nrm, real *xyz_out)
{
    if (n <= 0 || knn < 3) return -1;
    FN(ssn_state) S;
    S.xyz = xyz; S.knn = knn; S.method = sampling_method; S.ratio = ratio; S.max_box = max_box_dim; S.seed = (uint64_t)seed;
    S.keep = keep; S.nrm = nrm; S.xyz_out = xyz_out; S.boxes = 0;
    S.idx = (int *)malloc(sizeof(int) * (size_t)n);
    real lo[3] = {xyz[0], xyz[1], xyz[2]}, hi[3] = {xyz[0], xyz[1], xyz[2]};
    for (int i = 0; i < n; i++) {
        S.idx[i] = i; keep[i] = 0;
        for (int a = 0; a < 3; a++) { const real v = xyz[3 * i + a]; if (v < lo[a]) lo[a] = v; if (v > hi[a]) hi[a] = v; }
    }
    FN(ssn_build)(&S, 0, n, lo, hi);
    free(S.idx);
    return S.boxes;
}

/* --------------------------------------------------------------------------
 * [EXT] `densities` of SurfaceNormalDataPointsFilter{keepDensities: 1} (computeDensity in DataPointsFilters/utils) and
 * MaxDensityDataPointsFilter{maxDensity} (DataPointsFilters/MaxDensity.cpp), restated from the published source:
 *   density(i) = k / ((4 / 3) pi r^3), k = neighbours found (the point itself among them), r = the largest distance of a
 *     neighbour from the neighbourhood's MEAN -- NN.colwise().norm().maxCoeff() on the mean-subtracted columns -- in T;
 *   MaxDensity keeps a point whose density is <= maxDensity; a denser one with probability maxDensity / density (upstream:
 *     rand() / RAND_MAX < acceptRatio -- the build's counter-based draw here); for points AT the cloud's largest density the
 *     ratio is multiplied by (1 - nbSaturatedPts / nbPointsIn) in INTEGER arithmetic, as upstream writes it: 1 unless every
 *     point is saturated, then 0.
 * ------------------------------------------------------------------------ */
void FN(orc_densities)(const real *xyz, int n, int knn, const int *ids /* n x knn, -1 = none */, real *dens)
{
    for (int i = 0; i < n; i++) {
        const int *nb = ids + (size_t)i * knn;
        int cnt = 0;
        real sx = 0, sy = 0, sz = 0;
        for (int j = 0; j < knn; j++)
            if (nb[j] >= 0) { sx += xyz[3 * nb[j]]; sy += xyz[3 * nb[j] + 1]; sz += xyz[3 * nb[j] + 2]; cnt++; }
        real r2 = 0;
        if (cnt > 0) {
            const real mx = sx / (real)cnt, my = sy / (real)cnt, mz = sz / (real)cnt;
            for (int j = 0; j < knn; j++)
                if (nb[j] >= 0) {
                    const real dx = xyz[3 * nb[j]] - mx, dy = xyz[3 * nb[j] + 1] - my, dz = xyz[3 * nb[j] + 2] - mz;
                    const real q = (dx * dx + dy * dy) + dz * dz;
                    if (q > r2) r2 = q;
                }
        }
        const real r = SQRT_R(r2);
        const real volume = (real)((4.0 / 3.0) * 3.14159265358979323846) * ((r * r) * r);
        dens[i] = (real)cnt / volume;
    }
}
void FN(orc_max_density_keep)(const real *dens, int n, real max_density, double seed, int *keep)
{
    real last = dens[0];
    for (int i = 1; i < n; i++) if (dens[i] > last) last = dens[i];
    int saturated = 0;
    for (int i = 0; i < n; i++) saturated += dens[i] == last;
    for (int i = 0; i < n; i++) {
        keep[i] = 1;
        if (dens[i] > max_density) {
            float accept = (float)(max_density / dens[i]);
            if (dens[i] == last) accept = accept * (float)(1 - saturated / n);
            const double u = (double)(orc_splitmix((uint64_t)seed * 0x100000001B3ULL + (uint64_t)i) >> 11) / 9007199254740992.0;
            keep[i] = u < (double)accept;
        }
    }
}

/* --------------------------------------------------------------------------
 * SimpleSensorNoiseDataPointsFilter and the sensor-noise branch of getOverlap() (libpointmatcher 1.3.1,
 * DataPointsFilters/SimpleSensorNoise.cpp, ErrorMinimizers/PointToPlane.cpp / PointToPoint.cpp: restated from the published
 * source as recalled; parity unpinned like the rest of this file).  pgslam reads getOverlap() at Localizer.hpp:278 and
 * LoopCloser.hpp:331; its own clouds carry no such descriptor, a user's input-filter YAML may add it.
 *   noise(i) = max(minRadius, beamAngle * |p_i| + beamConst), |p_i| the norm of the point's coordinates in T, per sensor
 *     type: 0 Sick LMS-1xx (0.012, 0.0068, 0.0008), 1 Hokuyo URG-04LX (0.028, 0.0013, 0.0001), 2 Hokuyo UTM-30LX
 *     (0.018, 0.0006, 0.0015), 4 Sick Tim3xx (0.004, 0.0053, -0.0092); 3 Kinect / Xtion: noise(i) = |p_i|^2 * (0.5 * 0.00285);
 *     the descriptor holds gain * noise.
 *   getOverlap(): over the LAST error elements (the pairs the outlier filters kept in the last iteration, nbPoints of them):
 *     dists(j) = |reading_j - reference_j| in T, mean = sum(dists) / nbPoints, overlap = #{ dists(j) < mean + noise(j) } / nbPoints
 *     (both in T); without the descriptor -- for the point-to-plane minimizer also without `normals` on the reading --
 *     weightedPointUsedRatio.
 * ------------------------------------------------------------------------ */
int FN(orc_simple_sensor_noise)(const real *xyz, int n, int sensor_type, real gain, real *noise)
{
    real min_r = 0, angle = 0, cst = 0;
    switch (sensor_type) {
    case 0: min_r = (real)0.012; angle = (real)0.0068; cst = (real)0.0008; break;
    case 1: min_r = (real)0.028; angle = (real)0.0013; cst = (real)0.0001; break;
    case 2: min_r = (real)0.018; angle = (real)0.0006; cst = (real)0.0015; break;
    case 3: break;
    case 4: min_r = (real)0.004; angle = (real)0.0053; cst = (real)-0.0092; break;
    default: return -1;
    }
    for (int i = 0; i < n; i++) {
        const real x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const real r2 = (x * x + y * y) + z * z;
        real v;
        if (sensor_type == 3) { const real r = SQRT_R(r2); v = (r * r) * (real)(0.5 * 0.00285); }
        else { v = angle * SQRT_R(r2) + cst; if (v < min_r) v = min_r; }
        noise[i] = gain * v;
    }
    return 0;
}
/* d2 / w: the last iteration's squared distances and outlier weights, knn entries per reading point (point-major); noise per point */
double FN(orc_sensor_noise_overlap)(const real *d2, const real *w, const real *noise, int n, int knn)
{
    int nb = 0;
    real sum = 0;
    for (int k = 0; k < knn; k++)
        for (int i = 0; i < n; i++)
            if (w[(size_t)i * knn + k] != (real)0) { sum += SQRT_R(d2[(size_t)i * knn + k]); nb++; }
    if (nb == 0) return -1.0;
    const real mean = sum / (real)nb;
    int count = 0;
    for (int k = 0; k < knn; k++)
        for (int i = 0; i < n; i++)
            if (w[(size_t)i * knn + k] != (real)0 && SQRT_R(d2[(size_t)i * knn + k]) < mean + noise[i]) count++;
    return (double)((real)count / (real)nb);
}

/* --------------------------------------------------------------------------
 * [A.4] TrimmedDistOutlierFilter
 * ------------------------------------------------------------------------ */
static int cmp_real(const void *a, const void *b)
{
    const real x = *(const real *)a, y = *(const real *)b;
    return (x > y) - (x < y);
}

static int FN(orc_quantile_weights)(const real *d2, int n, real ratio, real scale, real *w, real *limit_out, int *n_finite_out);
int FN(orc_trim_weights)(const real *d2, int n, real ratio, real *w, real *limit_out, int *n_finite_out)
{
    return FN(orc_quantile_weights)(d2, n, ratio, (real)1, w, limit_out, n_finite_out);
}
/* TrimmedDist (scale 1) and MedianDist (ratio 0.5, scale = factor): limit = scale * getDistsQuantile(ratio) in T */
int FN(orc_median_weights)(const real *d2, int n, real factor, real *w, real *limit_out, int *n_finite_out)
{
    return FN(orc_quantile_weights)(d2, n, (real)0.5, factor, w, limit_out, n_finite_out);
}
static int FN(orc_quantile_weights)(const real *d2, int n, real ratio, real scale, real *w, real *limit_out, int *n_finite_out)
{
    real *vals = (real *)malloc(sizeof(real) * (n > 0 ? n : 1));
    int nf = 0;
    for (int i = 0; i < n; i++)
        if (d2[i] != INFINITY) vals[nf++] = d2[i];
    if (n_finite_out) *n_finite_out = nf;
    if (nf == 0) { free(vals); return ORC_ERR_NO_MATCH; }
    if (ratio < 0 || ratio > 1) { free(vals); return ORC_ERR_ARG; }
    qsort(vals, nf, sizeof(real), cmp_real);          /* nth_element == sorted[k] */
    real limit;
    if (ratio == (real)1) limit = vals[nf - 1];
    else {
        /* `values.size() * quantile` is evaluated in T (size_t * T -> T) */
        size_t k = (size_t)((real)nf * ratio);
        if (k >= (size_t)nf) k = nf - 1;
        limit = vals[k];
    }
    free(vals);
    if (scale > (real)0 && scale != (real)1) limit = scale * limit;
    for (int i = 0; i < n; i++) w[i] = (d2[i] <= limit) ? (real)1 : (real)0;
    if (limit_out) *limit_out = limit;
    return ORC_OK;
}

/* --------------------------------------------------------------------------
 * [A.4] RobustOutlierFilter ([EXT] OutlierFiltersImpl.cpp RobustOutlierFilter::robustFiltering, distanceType point2point,
 * nbIterationForScale 0 = the scale is estimated at every call):
 *   scale: scaleEstimator "mad": scale = sqrt(matches.getMedianAbsDeviation()), with [EXT] Matches::getMedianAbsDeviation =
 *          median of |d - median(d)| over the finite (squared) distances, both medians the element at index size / 2 after
 *          nth_element; "none": scale = 1
 *   e2 = dists / (scale * scale)   (scale * scale in T: the square of the rounded root, not the deviation itself)
 *   k = tuning, k2 = k * k:  cauchy 1 / (1 + e2 / k2) | welsch exp(-e2 / k2) | sc e2 >= k ? 4 k2 / (k + e2)^2 : 1 |
 *          gm k2 / (k + e2)^2 | tukey e2 >= k2 ? 0 : (1 - e2 / k2)^2 | huber e2 >= k2 ? k / sqrt(e2) : 1 | L1 1 / sqrt(e2)
 *   w = (w <= 1e-50) ? 1e-50 : w   -- the constant is a double compared with an array of T: in float it is 0, the clause a no-op
 *   approximation != inf: w = e2 >= approximation^2 ? 0 : w
 * An invalid match (distance +inf) comes out with weight 0 in every function (1e-50 in double: the pair is still dropped, its
 * id is invalid).  Returns the squared scale through scale2_out.
 * ------------------------------------------------------------------------ */
int FN(orc_robust_weights)(const real *d2, int n, int fct, real tuning, int scale_est, real approx, real *w, real *scale2_out)
{
    real s2 = (real)1;
    if (scale_est == 1) {
        real *vals = (real *)malloc(sizeof(real) * (n > 0 ? n : 1));
        int nf = 0;
        for (int i = 0; i < n; i++) if (d2[i] != INFINITY) vals[nf++] = d2[i];
        if (nf == 0) { free(vals); return ORC_ERR_NO_MATCH; }
        qsort(vals, nf, sizeof(real), cmp_real);
        const real med = vals[nf / 2];
        for (int i = 0; i < nf; i++) vals[i] = vals[i] > med ? vals[i] - med : med - vals[i];      /* fabs(v - median) */
        qsort(vals, nf, sizeof(real), cmp_real);
        const real scale = SQRT_R(vals[nf / 2]);
        free(vals);
        s2 = scale * scale;
    }
    if (scale2_out) *scale2_out = s2;
    const real k = tuning, k2 = k * k;
    const int cut = approx > (real)0 && !isinf(approx);
    const real a2 = approx * approx;
    for (int i = 0; i < n; i++) {
        const real e2 = d2[i] / s2;
        real wi = (real)0;
        switch (fct) {
        case 1: wi = (real)1 / ((real)1 + e2 / k2); break;
        case 2: wi = EXP_R(-(e2 / k2)); break;
        case 3: { const real t = k + e2; wi = e2 >= k ? ((real)4 * k2) / (t * t) : (real)1; break; }
        case 4: { const real t = k + e2; wi = k2 / (t * t); break; }
        case 5: { const real t = (real)1 - e2 / k2; wi = e2 >= k2 ? (real)0 : t * t; break; }
        case 6: wi = e2 >= k2 ? k / SQRT_R(e2) : (real)1; break;
        case 7: wi = (real)1 / SQRT_R(e2); break;
        default: free(NULL); return ORC_ERR_ARG;
        }
#ifdef ORC_DOUBLE
        if (wi <= 1e-50) wi = 1e-50;
#endif
        if (cut && e2 >= a2) wi = (real)0;
        if (d2[i] == INFINITY) wi = (real)0;                 /* (no neighbour: never an error element) */
        w[i] = wi;
    }
    return ORC_OK;
}

/* --------------------------------------------------------------------------
 * THE REDUCTION TREE ("RT-1") -- one agreed order and one agreed shape for every sum over pairs that
 * feeds a result (normal equations, Kabsch sums, residual, Censi sums), stated here and walked by the HIP kernels
 * (k_p2plane_reduce, k_cov_reduce, k_error_stats + block_reduce_store, sum_partials_256), so that a whole ICP -- T_out, cov,
 * residual -- can be compared BIT FOR BIT instead of "to 1e-12": double addition is not associative, and until round 6 the
 * oracle added in scan order in one chain while the device added in its own sorting order in a tree of its own.
 *   T1  positions: pair position pos = j * K + k holds (point order[j], its k-th neighbour); order = identity (scan order)
 *       unless the caller names another permutation of the points (orc_params.pair_order).
 *   T2  blocks of 2048 positions; in block b "thread" t (0..255) adds the kept pairs at positions b*2048 + u*256 + t,
 *       u = 0..7 in that order, each into its own accumulators, which start at +0.0.
 *   T3  the 256 threads are four groups of 64 ("waves"); a group is folded by v[l] += v[l + o] for l < o, o = 32, 16, 8, 4,
 *       2, 1 (the shuffle-down tree), the group's value is v[0].
 *   T4  block value = (((+0.0 + g0) + g1) + g2) + g3.
 *   T5  blocks -> problem: eight chains c = 0..7, chain c = ((+0.0 + B[c]) + B[c + 8]) + B[c + 16] ...; total =
 *       (((+0.0 + chain0) + chain1) + ...) + chain7.
 * What a pair adds is the callback's business (its own arithmetic, pair by pair, is the same on both sides already).
 * This is NOT what libpointmatcher does (Eigen products over columns, in T: the ORC_ACCUM_T variant); double accumulation
 * was the oracle's choice from round 1 and the envelope of round 5 says what it is worth.
 * ------------------------------------------------------------------------ */
#define ORC_RT_THREADS 256
#define ORC_RT_PER_THREAD 8
#define ORC_RT_SPAN (ORC_RT_THREADS * ORC_RT_PER_THREAD)
#define ORC_RT_MAX_TERMS 48
typedef void (*FN(pair_fn))(const void *ctx, size_t e, double *acc);      /* adds pair e (ORIGINAL numbering) if it is kept */
static void FN(tree_sum)(size_t np, int K, int nt, const int *order, FN(pair_fn) f, const void *ctx, double *out)
{
    double (*thr)[ORC_RT_MAX_TERMS] = (double (*)[ORC_RT_MAX_TERMS])malloc(sizeof(double) * ORC_RT_MAX_TERMS * ORC_RT_THREADS);
    const size_t nb = (np + ORC_RT_SPAN - 1) / ORC_RT_SPAN;
    double chain[8][ORC_RT_MAX_TERMS];
    for (int c = 0; c < 8; c++) for (int k = 0; k < nt; k++) chain[c][k] = 0.0;
    for (size_t b = 0; b < nb; b++) {
        for (int t = 0; t < ORC_RT_THREADS; t++) {                                  /* T2 */
            for (int k = 0; k < nt; k++) thr[t][k] = 0.0;
            for (int u = 0; u < ORC_RT_PER_THREAD; u++) {
                const size_t pos = b * ORC_RT_SPAN + (size_t)u * ORC_RT_THREADS + t;
                if (pos >= np) continue;
                const size_t e = order ? (size_t)order[pos / K] * K + pos % K : pos;
                f(ctx, e, thr[t]);
            }
        }
        for (int k = 0; k < nt; k++) {
            double blk = 0.0;
            for (int g = 0; g < ORC_RT_THREADS / 64; g++) {                         /* T3 */
                double v[64];
                for (int l = 0; l < 64; l++) v[l] = thr[g * 64 + l][k];
                for (int o = 32; o > 0; o >>= 1)
                    for (int l = 0; l < o; l++) v[l] = v[l] + v[l + o];
                blk += v[0];                                                        /* T4 */
            }
            chain[b & 7][k] += blk;                                                 /* T5: chain c takes blocks c, c + 8, ... in order */
        }
    }
    for (int k = 0; k < nt; k++) {
        double tot = 0.0;
        for (int c = 0; c < 8; c++) tot += chain[c][k];
        out[k] = tot;
    }
    free(thr);
}

/* --------------------------------------------------------------------------
 * [A.5]/[A.6]/[A.7] error elements -> point-to-plane normal equations
 *   sys[0..20]  upper triangle of A (row-major: 00 01 .. 05 11 12 .. 55)
 *   sys[21..26] b
 *   sys[27] sum w, sys[28] kept count, sys[29] residual sum w e^2
 * ------------------------------------------------------------------------ */
/* knn = K neighbours per reading point: pair e = i * K + k is (reading point i, its k-th neighbour).  The default build adds
 * the pairs through the reduction tree above, in pair order e (point major: the Matches matrix's column order); the ORC_ACCUM_T
 * variant keeps its own chain, k-major (all first neighbours, then all second ones), the order ErrorElements compacts them in
 * [A.5] */
typedef struct { const real *p, *ref_xyz, *ref_nrm, *w; const int *ids; int K; } FN(p2plane_ctx);
static void FN(p2plane_pair)(const void *vc, size_t pe, double *sys)
{
    const FN(p2plane_ctx) *c = (const FN(p2plane_ctx) *)vc;
    if (c->w[pe] == (real)0 || c->ids[pe] < 0) return;
    const size_t i = pe / (size_t)c->K;
    const int j = c->ids[pe];
    const real *p = c->p, *ref_xyz = c->ref_xyz, *ref_nrm = c->ref_nrm;
    const double wi = (double)c->w[pe];
    const double px = p[3 * i], py = p[3 * i + 1], pz = p[3 * i + 2];
    const double nx = ref_nrm[3 * j], ny = ref_nrm[3 * j + 1], nz = ref_nrm[3 * j + 2];
    const double dx = px - (double)ref_xyz[3 * j], dy = py - (double)ref_xyz[3 * j + 1],
                 dz = pz - (double)ref_xyz[3 * j + 2];
    const double e = (nx * dx + ny * dy) + nz * dz;
    double J[6];
    J[0] = py * nz - pz * ny;           /* c = p x n */
    J[1] = pz * nx - px * nz;
    J[2] = px * ny - py * nx;
    J[3] = nx; J[4] = ny; J[5] = nz;
    int k = 0;
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) sys[k++] += wi * (J[a] * J[b]);
    for (int a = 0; a < 6; a++) sys[21 + a] -= wi * (J[a] * e);
    sys[27] += wi;
    sys[28] += 1.0;
    sys[29] += wi * (e * e);
}
static int FN(p2plane_system_ko)(const real *p, int n, int K, const real *ref_xyz, const real *ref_nrm,
                                 const int *ids, const real *w, const int *order, double *sys)
{
    for (int i = 0; i < 30; i++) sys[i] = 0.0;
#ifdef ORC_ACCUM_T
    /* PointMatcher<T>: cross = p x n, wF = w .* F, A = wF F^T, dot = sum_rows(deltas .* normals), b = -(wF dot^T), all Matrix
     * of T; the products accumulated here pair by pair (column order) in `real` */
    {
        real acc[30];
        for (int i = 0; i < 30; i++) acc[i] = (real)0;
        for (int k = 0; k < K; k++)
        for (int i = 0; i < n; i++) {
            const size_t pe = (size_t)i * K + k;
            if (w[pe] == (real)0 || ids[pe] < 0) continue;
            const int j = ids[pe];
            const real wi = w[pe];
            const real px = p[3 * i], py = p[3 * i + 1], pz = p[3 * i + 2];
            const real nx = ref_nrm[3 * j], ny = ref_nrm[3 * j + 1], nz = ref_nrm[3 * j + 2];
            const real dx = px - ref_xyz[3 * j], dy = py - ref_xyz[3 * j + 1], dz = pz - ref_xyz[3 * j + 2];
            const real e = (dx * nx + dy * ny) + dz * nz;
            real J[6], wJ[6];
            J[0] = py * nz - pz * ny;
            J[1] = pz * nx - px * nz;
            J[2] = px * ny - py * nx;
            J[3] = nx; J[4] = ny; J[5] = nz;
            for (int a = 0; a < 6; a++) wJ[a] = wi * J[a];
            int q = 0;
            for (int a = 0; a < 6; a++)
                for (int b = a; b < 6; b++) acc[q++] += wJ[a] * J[b];
            for (int a = 0; a < 6; a++) acc[21 + a] -= wJ[a] * e;
            acc[27] += wi;
            acc[28] += (real)1;
            acc[29] += wi * (e * e);
        }
        for (int i = 0; i < 30; i++) sys[i] = (double)acc[i];
        /* (the kept count must stay exact: a float counter saturates at 2^24) */
        double kept = 0.0;
        for (size_t pe = 0; pe < (size_t)n * K; pe++) if (!(w[pe] == (real)0 || ids[pe] < 0)) kept += 1.0;
        sys[28] = kept;
        return sys[28] > 0 ? ORC_OK : ORC_ERR_NO_MATCH;
    }
#endif
    {
        const FN(p2plane_ctx) c = {p, ref_xyz, ref_nrm, w, ids, K};
        FN(tree_sum)((size_t)n * K, K, 30, order, FN(p2plane_pair), &c, sys);
    }
    return sys[28] > 0 ? ORC_OK : ORC_ERR_NO_MATCH;
}
__attribute__((unused)) static int FN(p2plane_system_k)(const real *p, int n, int K, const real *ref_xyz, const real *ref_nrm,
                                const int *ids, const real *w, double *sys)
{
    return FN(p2plane_system_ko)(p, n, K, ref_xyz, ref_nrm, ids, w, NULL, sys);
}
int FN(orc_p2plane_system)(const real *p, int n, const real *ref_xyz, const real *ref_nrm,
                           const int *ids, const real *w, double *sys)
{
    return FN(p2plane_system_ko)(p, n, 1, ref_xyz, ref_nrm, ids, w, NULL, sys);
}
/* the same with knn and the tree's order named (NULL: scan order) */
int FN(orc_p2plane_system_o)(const real *p, int n, int K, const real *ref_xyz, const real *ref_nrm,
                             const int *ids, const real *w, const int *order, double *sys)
{
    return FN(p2plane_system_ko)(p, n, K > 1 ? K : 1, ref_xyz, ref_nrm, ids, w, order, sys);
}

/* --------------------------------------------------------------------------
 * [A.4] SurfaceNormalOutlierFilter{maxAngle}: weight 0 when the angle between the reading point's normal (rotated with the
 * reading) and the matched reference point's normal exceeds maxAngle -- [EXT] OutlierFiltersImpl.cpp: both normals
 * .normalized() (v / sqrt(v.v), left alone when v.v == 0), value = dot, weight = value < cos(maxAngle) ? 0 : 1; an invalid
 * match gets 0.  Without normals on the reading the filter leaves every weight at 1.  Like every filter of the chain it
 * sees all matches and its weights multiply in.
 * ------------------------------------------------------------------------ */
static inline void FN(normalized3)(const real *v, real *o)
{
    const real z = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    if (z > (real)0) {
#ifdef ORC_DOUBLE
        const real s = sqrt(z);
#else
        const real s = sqrtf(z);
#endif
        o[0] = v[0] / s; o[1] = v[1] / s; o[2] = v[2] / s;
    } else { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }
}
void FN(orc_normal_weights)(const real *rd_nrm, const real *ref_nrm, const int *ids, int n, int K, real max_angle, real *w)
{
    if (!(max_angle > (real)0) || !rd_nrm) return;
#ifdef ORC_DOUBLE
    const real eps = cos(max_angle);
#else
    const real eps = cosf(max_angle);
#endif
    for (int i = 0; i < n; i++) {
        real a[3];
        FN(normalized3)(rd_nrm + 3 * i, a);
        for (int k = 0; k < K; k++) {
            const size_t e = (size_t)i * K + k;
            if (ids[e] < 0) { w[e] = (real)0; continue; }
            real b[3];
            FN(normalized3)(ref_nrm + 3 * (size_t)ids[e], b);
            const real v = (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
            if (v < eps) w[e] = (real)0;
        }
    }
}

/* --------------------------------------------------------------------------
 * PointToPointErrorMinimizer ([EXT] ErrorMinimizersImpl.cpp / PointToPoint.cpp): weighted means of the kept reading and
 * reference points, M = sum w (q - mq)(p - mp)^T, SVD M = U S V^T, R = U V^T -- with the last row of V^T negated when that is
 * a reflection --, t = mq - R mp.  getResidualError = sum over the kept pairs of |p - q| (the norm, unweighted: "return sum
 * of the norm of each delta").  getCovariance is the base class's: zeros.
 *   sys[0..2] sum w p   sys[3..5] sum w q   sys[6..14] sum w q_a p_b (row major)   sys[27] sum w  sys[28] kept  sys[29] residual
 * ------------------------------------------------------------------------ */
typedef struct { const real *p, *ref_xyz, *w; const int *ids; int K; } FN(p2point_ctx);
static void FN(p2point_pair)(const void *vc, size_t e, double *sys)
{
    const FN(p2point_ctx) *c = (const FN(p2point_ctx) *)vc;
    if (c->w[e] == (real)0 || c->ids[e] < 0) return;
    const real *pp = c->p + 3 * (e / (size_t)c->K), *qq = c->ref_xyz + 3 * (size_t)c->ids[e];
    const double wi = (double)c->w[e];
    for (int a = 0; a < 3; a++) { sys[a] += wi * (double)pp[a]; sys[3 + a] += wi * (double)qq[a]; }
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) sys[6 + 3 * a + b] += wi * ((double)qq[a] * (double)pp[b]);
    const double dx = (double)(real)(pp[0] - qq[0]), dy = (double)(real)(pp[1] - qq[1]), dz = (double)(real)(pp[2] - qq[2]);
    sys[27] += wi;
    sys[28] += 1.0;
    sys[29] += sqrt((dx * dx + dy * dy) + dz * dz);
}
int FN(orc_p2point_system_o)(const real *p, int n, int K, const real *ref_xyz, const int *ids, const real *w, const int *order, double *sys)
{
    if (K < 1) K = 1;
    const FN(p2point_ctx) c = {p, ref_xyz, w, ids, K};
    FN(tree_sum)((size_t)n * K, K, 30, order, FN(p2point_pair), &c, sys);      /* (slots 15..26 stay +0.0) */
    return sys[28] > 0 ? ORC_OK : ORC_ERR_NO_MATCH;
}
int FN(orc_p2point_system)(const real *p, int n, int K, const real *ref_xyz, const int *ids, const real *w, double *sys)
{
    return FN(orc_p2point_system_o)(p, n, K, ref_xyz, ids, w, NULL, sys);
}

/* cyclic Jacobi eigen-decomposition of a symmetric 6x6 (double) */
static void jacobi6(double *a /* 36, destroyed */, double *v /* 36 */, double *ev)
{
    for (int i = 0; i < 36; i++) v[i] = 0.0;
    for (int i = 0; i < 6; i++) v[i * 6 + i] = 1.0;
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int i = 0; i < 6; i++)
            for (int j = i + 1; j < 6; j++) off += a[i * 6 + j] * a[i * 6 + j];
        if (off == 0.0) break;
        for (int p = 0; p < 5; p++)
            for (int q = p + 1; q < 6; q++) {
                const double apq = a[p * 6 + q];
                if (apq == 0.0) continue;
                const double theta = (a[q * 6 + q] - a[p * 6 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; k++) {
                    const double akp = a[k * 6 + p], akq = a[k * 6 + q];
                    a[k * 6 + p] = c * akp - s * akq;
                    a[k * 6 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; k++) {
                    const double apk = a[p * 6 + k], aqk = a[q * 6 + k];
                    a[p * 6 + k] = c * apk - s * aqk;
                    a[q * 6 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; k++) {
                    const double vkp = v[k * 6 + p], vkq = v[k * 6 + q];
                    v[k * 6 + p] = c * vkp - s * vkq;
                    v[k * 6 + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 6; i++) ev[i] = a[i * 6 + i];
}

/* [A.6] solvePossiblyUnderdeterminedLinearSystem.  libpointmatcher tests
 * invertibility with a rank-revealing QR and then uses A.llt(); rank-deficient
 * systems get the minimal-norm solution.  Restated as: Cholesky whose pivots
 * must all exceed 6*eps(T)*max|A_ii| (numerically full rank), otherwise the
 * pseudo-inverse solution over the eigen-directions with
 * lambda > 6*eps(T)*lambda_max.  (Assumption-ledger item: rank threshold.) */
int FN(orc_solve6)(const double *sys, double *x, int *rank_out)
{
#ifdef ORC_ACCUM_T
    /* A.llt().solve(b) on Matrix / Vector of T: the factorisation and both substitutions in `real` (the rank test is the
     * default build's, on the same pivots) */
    {
        real Ar[36], Lr[36], yr[6], xr[6];
        int q = 0;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) { Ar[a * 6 + b] = (real)sys[q]; Ar[b * 6 + a] = (real)sys[q]; q++; }
        real dmax_r = (real)0;
        for (int i = 0; i < 6; i++) if (fabs((double)Ar[i * 6 + i]) > (double)dmax_r) dmax_r = (real)fabs((double)Ar[i * 6 + i]);
        memset(Lr, 0, sizeof Lr);
        int ok_r = 1;
        for (int j = 0; j < 6 && ok_r; j++) {
            real d = Ar[j * 6 + j];
            for (int m = 0; m < j; m++) d -= Lr[j * 6 + m] * Lr[j * 6 + m];
            if (!(d > dmax_r * (real)6 * REAL_EPS)) { ok_r = 0; break; }
            Lr[j * 6 + j] = SQRT_R(d);
            for (int i = j + 1; i < 6; i++) {
                real t = Ar[i * 6 + j];
                for (int m = 0; m < j; m++) t -= Lr[i * 6 + m] * Lr[j * 6 + m];
                Lr[i * 6 + j] = t / Lr[j * 6 + j];
            }
        }
        if (ok_r) {
            for (int i = 0; i < 6; i++) {
                real t = (real)sys[21 + i];
                for (int m = 0; m < i; m++) t -= Lr[i * 6 + m] * yr[m];
                yr[i] = t / Lr[i * 6 + i];
            }
            for (int i = 5; i >= 0; i--) {
                real t = yr[i];
                for (int m = i + 1; m < 6; m++) t -= Lr[m * 6 + i] * xr[m];
                xr[i] = t / Lr[i * 6 + i];
            }
            for (int i = 0; i < 6; i++) x[i] = (double)xr[i];
            if (rank_out) *rank_out = 6;
            return ORC_OK;
        }
        /* rank deficient: the default build's minimal-norm branch below */
    }
#endif
    double A[36], L[36];
    int k = 0;
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) { A[a * 6 + b] = sys[k]; A[b * 6 + a] = sys[k]; k++; }
    const double *bv = sys + 21;
#ifdef ORC_DOUBLE
    const double rel = 6.0 * DBL_EPSILON;
#else
    const double rel = 6.0 * (double)FLT_EPSILON;
#endif
    double dmax = 0.0;
    for (int i = 0; i < 6; i++) if (fabs(A[i * 6 + i]) > dmax) dmax = fabs(A[i * 6 + i]);
    memset(L, 0, sizeof L);
    int ok = 1;
    for (int j = 0; j < 6 && ok; j++) {
        double d = A[j * 6 + j];
        for (int m = 0; m < j; m++) d -= L[j * 6 + m] * L[j * 6 + m];
        if (!(d > dmax * rel)) { ok = 0; break; }
        L[j * 6 + j] = sqrt(d);
        for (int i = j + 1; i < 6; i++) {
            double s = A[i * 6 + j];
            for (int m = 0; m < j; m++) s -= L[i * 6 + m] * L[j * 6 + m];
            L[i * 6 + j] = s / L[j * 6 + j];
        }
    }
    if (ok) {
        double y[6];
        for (int i = 0; i < 6; i++) {
            double s = bv[i];
            for (int m = 0; m < i; m++) s -= L[i * 6 + m] * y[m];
            y[i] = s / L[i * 6 + i];
        }
        for (int i = 5; i >= 0; i--) {
            double s = y[i];
            for (int m = i + 1; m < 6; m++) s -= L[m * 6 + i] * x[m];
            x[i] = s / L[i * 6 + i];
        }
        if (rank_out) *rank_out = 6;
        return ORC_OK;
    }
    /* minimal-norm solution over the numerically non-null eigen-space */
    double Aw[36], V[36], ev[6];
    memcpy(Aw, A, sizeof A);
    jacobi6(Aw, V, ev);
    double emax = 0.0;
    for (int i = 0; i < 6; i++) if (fabs(ev[i]) > emax) emax = fabs(ev[i]);
    const double tol = emax * rel;
    int rank = 0;
    for (int i = 0; i < 6; i++) x[i] = 0.0;
    for (int e = 0; e < 6; e++) {
        if (!(ev[e] > tol)) continue;
        rank++;
        double dot = 0.0;
        for (int i = 0; i < 6; i++) dot += V[i * 6 + e] * bv[i];
        const double c = dot / ev[e];
        for (int i = 0; i < 6; i++) x[i] += c * V[i * 6 + e];
    }
    if (rank_out) *rank_out = rank;
    return ORC_OK;
}

/* [EXT] PointToPlaneErrorMinimizer{force4DOF: 1} (libpointmatcher ErrorMinimizers/PointToPlane.cpp): only the z component
 * of p x n enters F, the unknowns are [rz tx ty tz], the system is 4 x 4 -- the [2..5] block of the 6 x 6 normal equations,
 * since dropping rows of F drops the matching rows and columns of F F^T -- and the increment is AngleAxis(x0, unitZ) plus the
 * translation.  Solved like [A.6]: Cholesky with the relative pivot test on the block, else the minimal-norm solution.
 * x comes back as [0 0 rz tx ty tz]. */
int FN(orc_solve_4dof)(const double *sys, double *x, int *rank_out)
{
    double A[16], L[16], b[4];
    static const int pk[4][4] = {{11, 12, 13, 14}, {12, 15, 16, 17}, {13, 16, 18, 19}, {14, 17, 19, 20}};   /* packed upper triangle */
    for (int i = 0; i < 4; i++) { b[i] = sys[21 + 2 + i]; for (int j = 0; j < 4; j++) A[i * 4 + j] = sys[pk[i][j]]; }
#ifdef ORC_DOUBLE
    const double rel = 6.0 * DBL_EPSILON;
#else
    const double rel = 6.0 * (double)FLT_EPSILON;
#endif
    double dmax = 0.0;
    for (int i = 0; i < 4; i++) if (fabs(A[i * 4 + i]) > dmax) dmax = fabs(A[i * 4 + i]);
    memset(L, 0, sizeof L);
    int ok = 1;
    for (int j = 0; j < 4 && ok; j++) {
        double d = A[j * 4 + j];
        for (int m = 0; m < j; m++) d -= L[j * 4 + m] * L[j * 4 + m];
        if (!(d > dmax * rel)) { ok = 0; break; }
        L[j * 4 + j] = sqrt(d);
        for (int i = j + 1; i < 4; i++) {
            double t = A[i * 4 + j];
            for (int m = 0; m < j; m++) t -= L[i * 4 + m] * L[j * 4 + m];
            L[i * 4 + j] = t / L[j * 4 + j];
        }
    }
    double y[4], z[4] = {0, 0, 0, 0};
    int rank = 4;
    if (ok) {
        for (int i = 0; i < 4; i++) {
            double t = b[i];
            for (int m = 0; m < i; m++) t -= L[i * 4 + m] * y[m];
            y[i] = t / L[i * 4 + i];
        }
        for (int i = 3; i >= 0; i--) {
            double t = y[i];
            for (int m = i + 1; m < 4; m++) t -= L[m * 4 + i] * z[m];
            z[i] = t / L[i * 4 + i];
        }
    } else {
        /* minimal-norm solution: the block embedded in a 6 x 6 of zeros has the block's eigen-pairs and two null ones */
        double A6[36], V[36], ev[6];
        memset(A6, 0, sizeof A6);
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) A6[(i + 2) * 6 + (j + 2)] = A[i * 4 + j];
        jacobi6(A6, V, ev);
        double emax = 0.0;
        for (int i = 0; i < 6; i++) if (fabs(ev[i]) > emax) emax = fabs(ev[i]);
        rank = 0;
        for (int e = 0; e < 6; e++) {
            if (!(ev[e] > emax * rel)) continue;
            rank++;
            double dot = 0.0;
            for (int i = 0; i < 4; i++) dot += V[(i + 2) * 6 + e] * b[i];
            for (int i = 0; i < 4; i++) z[i] += dot / ev[e] * V[(i + 2) * 6 + e];
        }
    }
    x[0] = 0.0; x[1] = 0.0;
    for (int i = 0; i < 4; i++) x[2 + i] = z[i];
    if (rank_out) *rank_out = rank;
    return ORC_OK;
}

/* [A.6] x = [rx ry rz tx ty tz] -> 4x4; AngleAxis(|r|, r/|r|); NaN => R = I */
void FN(orc_delta_T)(const double *x, double *T)
{
    mat4_identity(T);
#ifdef ORC_ACCUM_T
    /* Eigen::AngleAxis<T>(x.head(3).norm(), x.head(3).normalized()).toRotationMatrix(), in T */
    {
        const real x0 = (real)x[0], x1 = (real)x[1], x2 = (real)x[2];
        const real th = SQRT_R((x0 * x0 + x1 * x1) + x2 * x2);
        if (th > (real)0 && isfinite(th)) {
            const real ux = x0 / th, uy = x1 / th, uz = x2 / th;
            const real c = COS_R(th), sn = SIN_R(th), Cc = (real)1 - c;
            T[0] = (double)(real)(c + ux * ux * Cc);       T[1] = (double)(real)(ux * uy * Cc - uz * sn); T[2] = (double)(real)(ux * uz * Cc + uy * sn);
            T[4] = (double)(real)(uy * ux * Cc + uz * sn); T[5] = (double)(real)(c + uy * uy * Cc);       T[6] = (double)(real)(uy * uz * Cc - ux * sn);
            T[8] = (double)(real)(uz * ux * Cc - uy * sn); T[9] = (double)(real)(uz * uy * Cc + ux * sn); T[10] = (double)(real)(c + uz * uz * Cc);
        }
        T[3] = (double)(real)x[3]; T[7] = (double)(real)x[4]; T[11] = (double)(real)x[5];
        return;
    }
#endif
    const double th = sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]);
    if (th > 0.0 && isfinite(th)) {
        const double ux = x[0] / th, uy = x[1] / th, uz = x[2] / th;
        const double c = cos(th), s = sin(th), C = 1.0 - c;
        T[0] = c + ux * ux * C;      T[1] = ux * uy * C - uz * s; T[2] = ux * uz * C + uy * s;
        T[4] = uy * ux * C + uz * s; T[5] = c + uy * uy * C;      T[6] = uy * uz * C - ux * s;
        T[8] = uz * ux * C - uy * s; T[9] = uz * uy * C + ux * s; T[10] = c + uz * uz * C;
    }
    T[3] = x[3]; T[7] = x[4]; T[11] = x[5];
}

/* SVD of a 3x3 through the eigen-decomposition of M^T M (cyclic Jacobi, double): singular values descending, V's columns
 * the right singular vectors, U's columns M v / s -- completed to an orthonormal frame where M is rank deficient. */
static void svd3(const double M[3][3], double U[3][3], double S[3], double V[3][3])
{
    double B[3][3], W[3][3], ev[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) B[i][j] = (M[0][i] * M[0][j] + M[1][i] * M[1][j]) + M[2][i] * M[2][j];
    jacobi3(B, W, ev);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; a++)
        for (int b = a + 1; b < 3; b++)
            if (ev[ord[b]] > ev[ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    for (int c = 0; c < 3; c++) {
        S[c] = ev[ord[c]] > 0.0 ? sqrt(ev[ord[c]]) : 0.0;
        for (int r = 0; r < 3; r++) V[r][c] = W[r][ord[c]];
    }
    const double tol = 1e-12 * S[0];
    int have[3] = {0, 0, 0};
    for (int c = 0; c < 3; c++) {
        if (!(S[c] > tol)) continue;
        double u[3], nn = 0.0;
        for (int r = 0; r < 3; r++) { u[r] = ((M[r][0] * V[0][c] + M[r][1] * V[1][c]) + M[r][2] * V[2][c]) / S[c]; }
        /* Gram-Schmidt against the columns before it (they are orthogonal up to rounding already) */
        for (int b = 0; b < c; b++)
            if (have[b]) {
                const double d = (u[0] * U[0][b] + u[1] * U[1][b]) + u[2] * U[2][b];
                for (int r = 0; r < 3; r++) u[r] -= d * U[r][b];
            }
        nn = sqrt((u[0] * u[0] + u[1] * u[1]) + u[2] * u[2]);
        if (!(nn > 0.0)) continue;
        for (int r = 0; r < 3; r++) U[r][c] = u[r] / nn;
        have[c] = 1;
    }
    if (have[0] && have[1] && !have[2]) {                       /* rank 2: the third direction is the cross product */
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
        have[2] = 1;
    }
    if (!(have[0] && have[1] && have[2]))                       /* rank < 2: no rotation is determined; leave it out */
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { U[r][c] = r == c ? 1.0 : 0.0; V[r][c] = r == c ? 1.0 : 0.0; }
}

/* PointToPoint: sums -> increment (4x4).  Returns the rank of M (3, 2, or less: identity rotation). */
int FN(orc_solve_p2point)(const double *sys, double *T)
{
    mat4_identity(T);
    const double sw = sys[27];
    if (!(sw > 0.0)) return 0;
    double mp[3], mq[3], M[3][3], U[3][3], S[3], V[3][3];
    for (int a = 0; a < 3; a++) { mp[a] = sys[a] / sw; mq[a] = sys[3 + a] / sw; }
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) M[a][b] = sys[6 + 3 * a + b] - sw * (mq[a] * mp[b]);
    svd3(M, U, S, V);
    double R[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) R[a][b] = (U[a][0] * V[b][0] + U[a][1] * V[b][1]) + U[a][2] * V[b][2];
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (det < 0.0)                                              /* a reflection: the second best solution, a rotation */
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) R[a][b] = (U[a][0] * V[b][0] + U[a][1] * V[b][1]) - U[a][2] * V[b][2];
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) T[4 * a + b] = R[a][b];
        T[4 * a + 3] = mq[a] - ((R[a][0] * mp[0] + R[a][1] * mp[1]) + R[a][2] * mp[2]);
    }
    return S[0] > 0.0 ? (S[2] > 1e-12 * S[0] ? 3 : (S[1] > 1e-12 * S[0] ? 2 : 1)) : 0;
}

/* rotation block -> unit quaternion (w,x,y,z), Shepperd's method */
__attribute__((unused)) static void rot_to_quat(const double *T, double *q)
{
    const double m00 = T[0], m01 = T[1], m02 = T[2], m10 = T[4], m11 = T[5], m12 = T[6],
                 m20 = T[8], m21 = T[9], m22 = T[10];
    const double tr = m00 + m11 + m22;
    if (tr > 0.0) {
        double s = sqrt(tr + 1.0) * 2.0;
        q[0] = 0.25 * s; q[1] = (m21 - m12) / s; q[2] = (m02 - m20) / s; q[3] = (m10 - m01) / s;
    } else if (m00 > m11 && m00 > m22) {
        double s = sqrt(1.0 + m00 - m11 - m22) * 2.0;
        q[0] = (m21 - m12) / s; q[1] = 0.25 * s; q[2] = (m01 + m10) / s; q[3] = (m02 + m20) / s;
    } else if (m11 > m22) {
        double s = sqrt(1.0 + m11 - m00 - m22) * 2.0;
        q[0] = (m02 - m20) / s; q[1] = (m01 + m10) / s; q[2] = 0.25 * s; q[3] = (m12 + m21) / s;
    } else {
        double s = sqrt(1.0 + m22 - m00 - m11) * 2.0;
        q[0] = (m10 - m01) / s; q[1] = (m02 + m20) / s; q[2] = (m12 + m21) / s; q[3] = 0.25 * s;
    }
    const double nn = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= nn;
}

/* Quaternion::angularDistance: d = a * conj(b); 2*atan2(|d.vec|, |d.w|) */
__attribute__((unused)) static double quat_angdist(const double *a, const double *b)
{
    const double w = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    const double x = -a[0] * b[1] + a[1] * b[0] - a[2] * b[3] + a[3] * b[2];
    const double y = -a[0] * b[2] + a[1] * b[3] + a[2] * b[0] - a[3] * b[1];
    const double z = -a[0] * b[3] - a[1] * b[2] + a[2] * b[1] + a[3] * b[0];
    return 2.0 * atan2(sqrt(x * x + y * y + z * z), fabs(w));
}

/* [A.9] checker state: Counter + Differential.  Exposed so that the scripted
 * known-answer test can drive it directly. */
#define ORC_HIST 64
typedef struct {
    int count, max_iters, smooth;
    double min_rot, min_trans;
    int n_hist;
    double quat[ORC_HIST][4];
    double trans[ORC_HIST][3];
    double bound_rot, bound_trans;    /* BoundTransformationChecker limits; <= 0: not in the chain */
} orc_checker;

void orc_checker_init(orc_checker *c, int max_iters, double min_rot, double min_trans, int smooth);
void orc_checker_set_bound(orc_checker *c, double max_rot, double max_trans);
int orc_checker_check(orc_checker *c, const double *T);
#ifndef ORC_DOUBLE   /* type-independent: defined once, in the f32 object */
void orc_checker_init(orc_checker *c, int max_iters, double min_rot, double min_trans, int smooth)
{
    memset(c, 0, sizeof *c);
    c->max_iters = max_iters; c->min_rot = min_rot; c->min_trans = min_trans;
    c->smooth = smooth < 1 ? 1 : (smooth > ORC_HIST - 2 ? ORC_HIST - 2 : smooth);
    /* checkers.init(T_iter = I): history starts with the identity */
    c->quat[0][0] = 1.0; c->n_hist = 1;
}

/* [A.9] BoundTransformationChecker{maxRotationNorm, maxTranslationNorm}: the checkers are initialised with T_iter = I
 * (A.2), so the bound is on the accumulated correction itself: angular distance of its rotation from the identity and norm
 * of its translation; exceeding either raises ConvergenceError.  The checkers run in the order Counter, Differential,
 * Bound: the Counter's max-iterations condition leaves the check before the Bound is looked at. */
void orc_checker_set_bound(orc_checker *c, double max_rot, double max_trans)
{
    c->bound_rot = (max_rot > 0.0 && isfinite(max_rot)) ? max_rot : 0.0;
    c->bound_trans = (max_trans > 0.0 && isfinite(max_trans)) ? max_trans : 0.0;
}

/* returns: bit0 = keep iterating, bit1 = differential stop, bit2 = counter stop, bit3 = NaN, bit4 = bound exceeded */
int orc_checker_check(orc_checker *c, const double *T)
{
    int iterate = 1, flags = 0;
    /* Counter */
    c->count++;
    if (c->count >= c->max_iters) { iterate = 0; flags |= 4; }
    /* Differential */
    if (c->n_hist == ORC_HIST) {
        memmove(c->quat[0], c->quat[1], sizeof(double) * 4 * (ORC_HIST - 1));
        memmove(c->trans[0], c->trans[1], sizeof(double) * 3 * (ORC_HIST - 1));
        c->n_hist--;
    }
    rot_to_quat(T, c->quat[c->n_hist]);
    c->trans[c->n_hist][0] = T[3]; c->trans[c->n_hist][1] = T[7]; c->trans[c->n_hist][2] = T[11];
    c->n_hist++;
    if (c->n_hist > c->smooth) {
        double rsum = 0.0, tsum = 0.0;
        for (int i = c->n_hist - 1; i >= c->n_hist - c->smooth; i--) {
            rsum += fabs(quat_angdist(c->quat[i], c->quat[i - 1]));
            const double dx = c->trans[i][0] - c->trans[i - 1][0], dy = c->trans[i][1] - c->trans[i - 1][1],
                         dz = c->trans[i][2] - c->trans[i - 1][2];
            tsum += fabs(sqrt(dx * dx + dy * dy + dz * dz));
        }
        rsum /= (double)c->smooth; tsum /= (double)c->smooth;
        if (isnan(rsum) || isnan(tsum)) return 8;
        if (rsum < c->min_rot && tsum < c->min_trans) { iterate = 0; flags |= 2; }
    }
    if (!(flags & 4) && (c->bound_rot > 0.0 || c->bound_trans > 0.0)) {
        const double ident[4] = {1.0, 0.0, 0.0, 0.0};
        const double *q = c->quat[c->n_hist - 1], *t = c->trans[c->n_hist - 1];
        const double rot = quat_angdist(q, ident), tr = sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]);
        if ((c->bound_rot > 0.0 && rot > c->bound_rot) || (c->bound_trans > 0.0 && tr > c->bound_trans)) return 16;
    }
    return flags | iterate;
}
#endif

/* [A.8] PointToPlaneWithCov: Censi closed form on the kept pairs of the last
 * iteration (p = reading moved by the previous T_iter), dT = last increment. */
static void FN(covariance_ko)(const real *p, int n, int K, const real *ref_xyz, const real *ref_nrm, const int *ids,
                              const real *w, const double *dT, double sensor_std_dev, const int *order, double *cov);
void FN(orc_covariance)(const real *p, int n, const real *ref_xyz, const real *ref_nrm, const int *ids,
                        const real *w, const double *dT, double sensor_std_dev, double *cov)
{
    FN(covariance_ko)(p, n, 1, ref_xyz, ref_nrm, ids, w, dT, sensor_std_dev, NULL, cov);
}
void FN(orc_covariance_o)(const real *p, int n, int K, const real *ref_xyz, const real *ref_nrm, const int *ids,
                          const real *w, const double *dT, double sensor_std_dev, const int *order, double *cov)
{
    FN(covariance_ko)(p, n, K > 1 ? K : 1, ref_xyz, ref_nrm, ids, w, dT, sensor_std_dev, order, cov);
}
/* one pair's terms: the upper triangles of H (21) and G (21) -- both are symmetric sums of symmetric terms, h[a] h[b] and
 * gr[a] gr[b] + gq[a] gq[b] are the same doubles either way round -- added through the reduction tree (tree_sum) */
typedef struct { const real *p, *ref_xyz, *ref_nrm, *w; const int *ids; int K; double alpha, beta, gamma, t_x, t_y, t_z; } FN(cov_ctx);
static void FN(cov_pair)(const void *vc, size_t pe, double *acc)
{
    const FN(cov_ctx) *c = (const FN(cov_ctx) *)vc;
    if (c->w[pe] == (real)0 || c->ids[pe] < 0) return;
    const size_t i = pe / (size_t)c->K;
    const int j = c->ids[pe];
    const real *p = c->p, *ref_xyz = c->ref_xyz, *ref_nrm = c->ref_nrm;
    const double alpha = c->alpha, beta = c->beta, gamma = c->gamma, t_x = c->t_x, t_y = c->t_y, t_z = c->t_z;
    const double px = p[3 * i], py = p[3 * i + 1], pz = p[3 * i + 2];
    const double qx = ref_xyz[3 * j], qy = ref_xyz[3 * j + 1], qz = ref_xyz[3 * j + 2];
    const double nx = ref_nrm[3 * j], ny = ref_nrm[3 * j + 1], nz = ref_nrm[3 * j + 2];
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
    const double h[6] = {nx, ny, nz, rr * n_alpha, rr * n_beta, rr * n_gamma};
    const double er = E + rr * Nr;
    const double gr[6] = {nx * Nr, ny * Nr, nz * Nr, n_alpha * er, n_beta * er, n_gamma * er};
    const double gq[6] = {nx * Nq, ny * Nq, nz * Nq, qr * n_alpha * Nq, qr * n_beta * Nq, qr * n_gamma * Nq};
    int k = 0;
    for (int a = 0; a < 6; a++)
        for (int b = a; b < 6; b++) {
            acc[k] += h[a] * h[b];
            acc[21 + k] += gr[a] * gr[b] + gq[a] * gq[b];
            k++;
        }
}
static void FN(covariance_ko)(const real *p, int n, int K, const real *ref_xyz, const real *ref_nrm, const int *ids,
                              const real *w, const double *dT, double sensor_std_dev, const int *order, double *cov)
{
    double H[36], G[36], acc[42];
    FN(cov_ctx) c = {p, ref_xyz, ref_nrm, w, ids, K, 0, 0, 0, dT[3], dT[7], dT[11]};
    c.beta = -asin(dT[8]);
    c.alpha = atan2(dT[9], dT[10]);
    c.gamma = atan2(dT[4] / cos(c.beta), dT[0] / cos(c.beta));
    FN(tree_sum)((size_t)n * K, K, 42, order, FN(cov_pair), &c, acc);
    {
        int k = 0;
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) { H[a * 6 + b] = H[b * 6 + a] = acc[k]; G[a * 6 + b] = G[b * 6 + a] = acc[21 + k]; k++; }
    }
    /* inv(H) by Gauss-Jordan with partial pivoting */
    double M[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { M[i][j] = H[i * 6 + j]; M[i][6 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++) if (fabs(M[r][c]) > fabs(M[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 12; j++) { double t = M[c][j]; M[c][j] = M[piv][j]; M[piv][j] = t; }
        const double d = M[c][c];
        for (int j = 0; j < 12; j++) M[c][j] /= d;
        for (int r = 0; r < 6; r++) {
            if (r == c) continue;
            const double f = M[r][c];
            for (int j = 0; j < 12; j++) M[r][j] -= f * M[c][j];
        }
    }
    double Hi[36], tmp[36];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Hi[i * 6 + j] = M[i][6 + j];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += Hi[i * 6 + k] * G[k * 6 + j];
            tmp[i * 6 + j] = s;
        }
    const double s2 = sensor_std_dev * sensor_std_dev;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            double s = 0.0;
            for (int k = 0; k < 6; k++) s += tmp[i * 6 + k] * Hi[k * 6 + j];
            cov[i * 6 + j] = s2 * s;
        }
}

/* one matcher call of the chain: K neighbours per point of `q`, ids / d2 laid out [point][neighbour] */
static void FN(match_k)(const FN(orc_params) *prm, void *tree, const real *q, int n, const real *ref, int m, int K, int *ids, real *d2)
{
    if (K <= 1) {
        if (tree) FN(orc_kdtree_knn)(tree, q, n, prm->max_dist, ids, d2);
        else FN(orc_knn_brute)(q, n, ref, m, prm->max_dist, ids, d2);
    } else {
        if (tree) FN(orc_kdtree_knn_k)(tree, q, n, K, prm->max_dist, ids, d2);
        else FN(orc_knn_brute_k)(q, n, ref, m, K, prm->max_dist, ids, d2);
    }
}

/* partial chain used by Localizer::ComputeOverlapWith (Localizer.hpp:309-347)
 * and LoopCloser::ComputeResidualError (LoopCloser.hpp:346-364): reading is
 * moved by T, matched against the RAW (un-centred) reference, weighted, and
 * both weightedPointUsedRatio and the residual are returned. */
int FN(orc_partial_chain)(const FN(orc_params) *prm, const real *reading, int n, const real *ref_xyz,
                          const real *ref_nrm, int m, const double *T, double *overlap, double *residual,
                          int *ids_out, real *d2_out)
{
    const int K = prm->knn > 1 ? prm->knn : 1;
    const size_t np = (size_t)(n > 0 ? n : 1) * K;              /* ids_out / d2_out hold knn entries per point */
    real *p = (real *)malloc(sizeof(real) * 3 * (n > 0 ? n : 1));
    int *ids = ids_out ? ids_out : (int *)malloc(sizeof(int) * np);
    real *d2 = d2_out ? d2_out : (real *)malloc(sizeof(real) * np);
    real *w = (real *)malloc(sizeof(real) * np);
    FN(orc_transform)(T, reading, p, n, 0);
    {
        void *t = prm->use_kdtree ? FN(orc_kdtree_build)(ref_xyz, m) : NULL;
        FN(match_k)(prm, t, p, n, ref_xyz, m, K, ids, d2);
        if (t) FN(orc_kdtree_free)(t);
    }
    real limit; int nf;
    int st = prm->robust_fct > 0 ? FN(orc_robust_weights)(d2, n * K, prm->robust_fct, prm->robust_tuning, prm->robust_scale, prm->robust_approx, w, &limit)
                                 : FN(orc_quantile_weights)(d2, n * K, prm->trim_ratio, prm->quantile_scale, w, &limit, &nf);
    if (st == ORC_OK) FN(orc_maxdist_weights)(d2, n * K, prm->outlier_max_dist, w);
    if (st == ORC_OK) {
        double sys[30];
        st = (prm->minimizer == 1 || prm->minimizer == 3) ? FN(orc_p2point_system_o)(p, n, K, ref_xyz, ids, w, prm->pair_order, sys)
                                 : FN(p2plane_system_ko)(p, n, K, ref_xyz, ref_nrm, ids, w, prm->pair_order, sys);
        if (overlap) *overlap = sys[27] / ((double)n * K);
        if (residual) *residual = sys[29];
    }
    free(p); free(w);
    if (!ids_out) free(ids);
    if (!d2_out) free(d2);
    return st;
}

/* --------------------------------------------------------------------------
 * [A.2] the ICP loop.  The reference side (copy, centre, index) is what
 * ICPSequence::setMap / ICP::operator() do once per map; it is split off so the
 * CPU baseline can time the per-scan loop separately from the index build
 * (mirrors setMap's amortisation, Localizer.hpp:148,168,254).
 * `trace` (optional) receives per iteration 16 doubles of T_iter.
 * ------------------------------------------------------------------------ */
typedef struct {
    real *ref;              /* centred copy of the reference xyz */
    const real *nrm;        /* borrowed */
    real mean[3];
    int m;
    void *tree;             /* kd-tree over ref, or NULL (brute force) */
} FN(orc_map);

void *FN(orc_map_create)(const real *ref_xyz, const real *ref_nrm, int m, int center, int use_kdtree)
{
    FN(orc_map) *M = (FN(orc_map) *)calloc(1, sizeof *M);
    M->m = m; M->nrm = ref_nrm;
    M->ref = (real *)malloc(sizeof(real) * 3 * (m > 0 ? m : 1));
    if (center) {
        FN(orc_centroid)(ref_xyz, m, M->mean);
        for (int i = 0; i < m; i++)
            for (int a = 0; a < 3; a++) M->ref[3 * i + a] = ref_xyz[3 * i + a] - M->mean[a];
    } else memcpy(M->ref, ref_xyz, sizeof(real) * 3 * m);
    M->tree = use_kdtree ? FN(orc_kdtree_build)(M->ref, m) : NULL;
    return M;
}

void FN(orc_map_free)(void *h)
{
    FN(orc_map) *M = (FN(orc_map) *)h;
    if (!M) return;
    if (M->tree) FN(orc_kdtree_free)(M->tree);
    free(M->ref); free(M);
}

int FN(orc_icp_map_ex)(const FN(orc_params) *prm, const void *map, const real *reading, const real *reading_nrm, int n,
                       const double *T_init, double *T_out, orc_result *res, double *trace, int trace_cap, int *last_ids, real *last_d2);
int FN(orc_icp_map)(const FN(orc_params) *prm, const void *map, const real *reading, int n, const double *T_init,
                    double *T_out, orc_result *res, double *trace, int trace_cap, int *last_ids, real *last_d2)
{
    return FN(orc_icp_map_ex)(prm, map, reading, NULL, n, T_init, T_out, res, trace, trace_cap, last_ids, last_d2);
}
/* `reading_nrm` (optional): the reading's `normals` descriptor, which RigidTransformation rotates with the points (a9) and the
 * SurfaceNormalOutlierFilter compares with the reference's; last_ids / last_d2 hold knn entries per point */
int FN(orc_icp_map_ex)(const FN(orc_params) *prm, const void *map, const real *reading, const real *reading_nrm, int n,
                       const double *T_init, double *T_out, orc_result *res, double *trace, int trace_cap, int *last_ids, real *last_d2)
{
    const FN(orc_map) *M = (const FN(orc_map) *)map;
    memset(res, 0, sizeof *res);
    if (n <= 0 || M->m <= 0) { res->status = ORC_ERR_ARG; return ORC_ERR_ARG; }
    const real *ref = M->ref, *ref_nrm = M->nrm;
    const int m = M->m;
    double T_ref_mean[16], T_ref_mean_inv[16], T_pre[16];
    mat4_identity(T_ref_mean);
    T_ref_mean[3] = M->mean[0]; T_ref_mean[7] = M->mean[1]; T_ref_mean[11] = M->mean[2];
    mat4_rigid_inverse(T_ref_mean, T_ref_mean_inv);
    mat4_mul_chain(T_ref_mean_inv, T_init, T_pre);

    const int K = prm->knn > 1 ? prm->knn : 1;
    const size_t np = (size_t)n * K;
    real *rd = (real *)malloc(sizeof(real) * 3 * n);       /* reading in centred-map frame */
    real *step = (real *)malloc(sizeof(real) * 3 * n);
    int *ids = (int *)malloc(sizeof(int) * np);
    real *d2 = (real *)malloc(sizeof(real) * np);
    real *w = (real *)malloc(sizeof(real) * np);
    FN(orc_transform)(T_pre, reading, rd, n, 0);
    const int use_nrm = reading_nrm && prm->normal_max_angle > (real)0;
    real *rd_n = use_nrm ? (real *)malloc(sizeof(real) * 3 * n) : NULL, *step_n = use_nrm ? (real *)malloc(sizeof(real) * 3 * n) : NULL;
    if (use_nrm) FN(orc_transform)(T_pre, reading_nrm, rd_n, n, 1);

    void *tree = M->tree;
    double T_iter[16], dT[16], T_prev[16];
    mat4_identity(T_iter); mat4_identity(dT); mat4_identity(T_prev);
    orc_checker chk;
    orc_checker_init(&chk, prm->max_iters, prm->min_diff_rot, prm->min_diff_trans, prm->smooth_length);
    orc_checker_set_bound(&chk, prm->bound_max_rot, prm->bound_max_trans);
    int status = ORC_OK, iterate = 1, it = 0;
    double sys[30];
    while (iterate) {
        FN(orc_transform)(T_iter, rd, step, n, 0);
        if (use_nrm) FN(orc_transform)(T_iter, rd_n, step_n, n, 1);
        FN(match_k)(prm, tree, step, n, ref, m, K, ids, d2);
        real limit; int nf;
        if (prm->robust_fct > 0) {
            /* the chain's distance filter is the robust one (no quantile filter beside it): every finite pair carries a weight */
            real s2;
            status = FN(orc_robust_weights)(d2, n * K, prm->robust_fct, prm->robust_tuning, prm->robust_scale, prm->robust_approx, w, &s2);
            limit = INFINITY; nf = 0;
            for (int i = 0; i < n * K; i++) nf += d2[i] != INFINITY;
        } else
            status = FN(orc_quantile_weights)(d2, n * K, prm->trim_ratio, prm->quantile_scale, w, &limit, &nf);
        if (status != ORC_OK) break;
        FN(orc_maxdist_weights)(d2, n * K, prm->outlier_max_dist, w);
        if (use_nrm) FN(orc_normal_weights)(step_n, ref_nrm, ids, n, K, prm->normal_max_angle, w);
        if (prm->minimizer == 1 || prm->minimizer == 3) {
            status = FN(orc_p2point_system_o)(step, n, K, ref, ids, w, prm->pair_order, sys);
            if (status != ORC_OK) break;
            FN(orc_solve_p2point)(sys, dT);
        } else {
            status = FN(p2plane_system_ko)(step, n, K, ref, ref_nrm, ids, w, prm->pair_order, sys);
            if (status != ORC_OK) break;
            double x[6]; int rank;
            if (prm->minimizer == 2) FN(orc_solve_4dof)(sys, x, &rank);
            else FN(orc_solve6)(sys, x, &rank);
            FN(orc_delta_T)(x, dT);
        }
        memcpy(T_prev, T_iter, sizeof T_iter);
        mat4_mul_chain(dT, T_iter, T_iter);
        res->overlap = sys[27] / ((double)n * K);
        res->residual = sys[29];
        res->trim_limit = (double)limit;
        if (prm->outlier_max_dist > (real)0 && !isinf(prm->outlier_max_dist) && prm->outlier_max_dist * prm->outlier_max_dist < limit)
            res->trim_limit = (double)(prm->outlier_max_dist * prm->outlier_max_dist);   /* the threshold the pairs were kept with */
        res->n_kept = (int)sys[28];
        res->n_finite = nf;
        if (trace && it < trace_cap) memcpy(trace + 16 * it, T_iter, sizeof T_iter);
        it++;
        const int f = orc_checker_check(&chk, T_iter);
        if (f & 8) { status = ORC_ERR_NAN; break; }
        if (f & 16) { status = ORC_ERR_BOUND; break; }
        iterate = f & 1;
        if (f & 2) res->converged = 1;
        if (f & 4) res->max_iter_reached = 1;
    }
    res->iterations = it;
    res->status = status;
    if (status == ORC_OK) {
        /* covariance on the last iteration's error elements (step = T_prev*rd); the point-to-point minimiser has the base
         * class's getCovariance: zeros */
        if (prm->minimizer == 1) memset(res->cov, 0, sizeof res->cov);
        else FN(covariance_ko)(step, n, K, ref, ref_nrm, ids, w, dT, prm->sensor_std_dev, prm->pair_order, res->cov);
        double t1[16];
        mat4_mul_chain(T_iter, T_pre, t1);
        mat4_mul_chain(T_ref_mean, t1, T_out);
    } else mat4_identity(T_out);
    if (last_ids) memcpy(last_ids, ids, sizeof(int) * np);
    if (last_d2) memcpy(last_d2, d2, sizeof(real) * np);
    free(rd); free(step); free(ids); free(d2); free(w); free(rd_n); free(step_n);
    return status;
}

int FN(orc_icp_ex)(const FN(orc_params) *prm, const real *reading, const real *reading_nrm, int n, const real *ref_xyz_in,
                   const real *ref_nrm, int m, const double *T_init, double *T_out, orc_result *res,
                   double *trace, int trace_cap, int *last_ids, real *last_d2)
{
    if (n <= 0 || m <= 0) { memset(res, 0, sizeof *res); res->status = ORC_ERR_ARG; return ORC_ERR_ARG; }
    void *M = FN(orc_map_create)(ref_xyz_in, ref_nrm, m, prm->center_reference, prm->use_kdtree);
    const int st = FN(orc_icp_map_ex)(prm, M, reading, reading_nrm, n, T_init, T_out, res, trace, trace_cap, last_ids, last_d2);
    FN(orc_map_free)(M);
    return st;
}
int FN(orc_icp)(const FN(orc_params) *prm, const real *reading, int n, const real *ref_xyz_in,
                const real *ref_nrm, int m, const double *T_init, double *T_out, orc_result *res,
                double *trace, int trace_cap, int *last_ids, real *last_d2)
{
    return FN(orc_icp_ex)(prm, reading, NULL, n, ref_xyz_in, ref_nrm, m, T_init, T_out, res, trace, trace_cap, last_ids, last_d2);
}
