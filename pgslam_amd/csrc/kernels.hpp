// kernels.hpp -- launch entry points of kernels.hip (host callable).
#pragma once
#include "device_types.hpp"

namespace pgicp {

template <typename T>
void launch_centroid_bbox_batch(hipStream_t st, const BuildDesc<T> *descs, int n, int max_m, unsigned long long *stats);
template <typename T>
void launch_grid_build_batch(hipStream_t st, const BuildDesc<T> *descs, int n, long long tot_m, long long tot_b, long long tot_s,
                             int max_m, int max_cells, int max_bins, int max_nsc, int max_blocks, int kx_all4, int *fkey, int *bins_a, int *block_sums,
                             int *cell_start, int *cell_start_f, int *bins_b, int *arrival, unsigned long long *words,
                             typename Vec4<T>::type *cpts, typename Vec4<T>::type *cnrm,
                             typename Vec4<T>::type *pts, typename Vec4<T>::type *nrm_out,
                             int *slot_of, int *sc_count, int *near, int *sc_dist, int *sc_wit, unsigned *occ, long long tot_o, float *sc_ext, float *ptsf,
                             uint4 *sw, int *ostart, int *flag_scratch, int *rank_scratch);
template <typename T>
void launch_query_sort(hipStream_t st, const ProblemDev *probs, const SrcDesc *src, const MapDev<T> *maps, typename Vec4<T>::type *rd_pre, T *rd_sorted,
                       int *qkey, unsigned long long *qtmp, int *order, int *counts, int *block_sums, int *qstart, int P,
                       int max_n, int max_rows, int bin_shift, typename Vec4<T>::type *nrm_pre, T *nrm_sorted);
template <typename T>
int launch_knn_topk(hipStream_t st, const ProblemDev *probs, const MapDev<T> *maps, const T *rd, int *slot, T *d2,
                    const ChainDev<T> &ch, int P, int max_n, const int *active);
template <typename T>
void launch_transform(hipStream_t st, const T *in, int in_stride, T *out, int out_stride, int n, const double *T16,
                      int rotate_only);
template <typename T>
void launch_knn(hipStream_t st, int matcher, const ProblemDev *probs, const MapDev<T> *maps, const T *rd, int *slot,
                T *d2, const ChainDev<T> &ch, int P, int max_n, int use_seed, int *slow_count, int2 *slow_list, T *slow_lb,
                int *slow_ring, int fast_rings, const int *active, T *none_r, int n_problems, void *queue_buf, int clear_queue_counters,
                int table_kinds);
size_t knn_queue_bytes(int n_problems, int max_n, size_t elem);
template <typename T>
void launch_knn_med(hipStream_t st, ProblemDev *probs, const MapDev<T> *maps, const T *rd, int *slot, T *d2,
                    const ChainDev<T> &ch, int *slow_count, const int2 *slow_list, T *slow_lb, int *slow_ring, int *slow2_idx,
                    int med_rings, int use_seed, T *none_r);
template <typename T>
void launch_knn_slow(hipStream_t st, ProblemDev *probs, const MapDev<T> *maps, const T *rd, int *slot, T *d2,
                     const ChainDev<T> &ch, const int *slow_count, const int2 *slow_list, const T *slow_lb,
                     const int *slow2_idx, int exact_all, T *none_r);
template <typename T>
void launch_trim_select(hipStream_t st, ProblemDev *probs, const T *d2, const ChainDev<T> &ch, int P, int max_n, int second,
                        const int *active, int *tables, void *keys, int *seg_count, int guess);
size_t trim_select_table_bytes(int P);
int sel_fallbacks_read(int reset);
int knn_stats_read(unsigned long long out[56], int reset);
int knn_phase_read(unsigned long long out[48], int reset);   // diagnostics build only: wave cycles per phase of the fast kernel
int knn_dump_setup(long long total, int passes);              // diagnostics build only: per-query candidate dump
int knn_dump_read(unsigned *cnt, float *d2, long long n);
int knn_trace_set(int sorted_index);                       // diagnostics build only   // diagnostics build (-DPGICP_KNN_STATS) only
void launch_reopen(hipStream_t st, ProblemDev *probs, int P);
void launch_compact_active(hipStream_t st, const ProblemDev *probs, int P, int *active, int *host_flag, int *stamp_counter,
                           int *queue_counters);
template <typename T>
int launch_surface_normals(hipStream_t st, const MapDev<T> *maps, int map, int m, int knn, T max_dist, T eps_rank, T *out_nrm,
                           int out_stride, T *out_eig, int *out_ids, T *out_d2);
int reduce_blocks(int max_n);
struct SeedSegs { int n; int src[17]; int dst[16]; };      // pgicp_partial_chain_seeded: the two maps as concatenations of keyframe clouds
template <typename T>
void launch_borrow_order(hipStream_t st, const ProblemDev *probs, const SrcDesc *src, const int *order_a, const int *slot_a, const typename Vec4<T>::type *pts_a,
                         int n, const SeedSegs &sg, const int *slot_of_b, int m_b, T *rd_sorted, int *order_b, int *slot_b);
void launch_batch_setup(hipStream_t st, int *active, int P, int *z0, long long n0, int *z1, long long n1, int *z2, long long n2, int *z3, long long n3);
size_t knn_queue_counter_words(int n_problems);      // the segmented queue counters at the head of the matcher's queue buffer
void launch_invert_order(hipStream_t st, const ProblemDev *probs, const int *order, int *scan_pos, int P, int max_n);
template <typename T>
void launch_reduce(hipStream_t st, const ProblemDev *probs, const MapDev<T> *maps, const T *rd_pre, const T *rd_nrm, const int *slot,
                   const T *d2, double *partials, int P, int max_n, const int *active, const ChainDev<T> &ch);
template <typename T>
void launch_solve(hipStream_t st, ProblemDev *probs, const double *partials, const ChainDev<T> &ch, int *n_done, int P,
                  int max_n, const int *active, int *single_host_flag, int *single_stamp, int *single_queue_counters);
template <typename T>
void launch_cov(hipStream_t st, const ProblemDev *probs, const MapDev<T> *maps, const T *rd_pre, const T *rd_nrm, const int *slot,
                const T *d2, double *partials, double *out, int P, int max_n, const ChainDev<T> &ch);
void launch_sum_partials(hipStream_t st, const double *partials, int max_blocks, int nt, const ProblemDev *probs,
                         int nb_uniform, double *out, int P);
void launch_robust_open(hipStream_t st, ProblemDev *probs, const int *active, int n_active);
template <typename T>
void launch_robust_raw(hipStream_t st, const T *d2, int n, const RobustDev<T> &rb, T *dev, T *stat, T *w);
template <typename T>
void launch_robust_scale(hipStream_t st, ProblemDev *probs, const T *d2, T *dev, const ChainDev<T> &ch, int n_active, int max_pairs,
                         const int *active, int *tables, void *keys);
template <typename T>
void launch_trim_raw(hipStream_t st, const T *d2, int n, T ratio, T scale, T *limit_nf, T *w);
template <typename T>
void launch_filter_cloud(hipStream_t st, const T *feat, int fstride, int frows, const T *desc, int drows, int n, int n_filters,
                         const int *types, const double *params, const double *T16, int rot0, int rot1, int *keep, int *pos,
                         int *block_sums, T *out_feat, T *out_desc, int *kept_idx, int *dropped = nullptr, int dropped_cap = 0);
template <typename T>
void launch_slot_of(hipStream_t st, const typename Vec4<T>::type *pts, int first, int m, int *slot_of);
template <typename T>
void launch_error_stats(hipStream_t st, const MapDev<T> *maps, int map, const int *slot_of, const T *rd, int stride,
                        const int *ids, const T *w, int n, int knn, const T mean[3], double *partials, double *out, int minimizer);
template <typename T>
void launch_unpermute(hipStream_t st, const MapDev<T> *maps, int map, const int *order, const int *slot, const T *d2, int n, int knn,
                      int *ids_out, T *d2_out);

constexpr int kScanChunkHost = 4096;

}  // namespace pgicp
