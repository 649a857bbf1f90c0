// pgslam.hpp -- pgslam's templated class API for the ICP hot path, on top of the
// drop-in PointMatcher (pointmatcher.hpp).  Only the hot-path methods of
// SURVEY.md §8(a) are provided; map management, candidate search and the
// pose-graph solve stay CPU-side concerns of the caller (north_star).
//
//   pgslam::Types<T>            reference src/pgslam/types.h:13-61
//   pgslam::BuildLocalMapCloud  LocalMap<T>::BuildCloudFromData   LocalMap.hpp:209-224
//   pgslam::CloudInFrame        LocalMap<T>::CloudInWorldFrame    LocalMap.hpp:95-98
//   pgslam::ScanLocalizer<T>    ProcessData / ComputeCurrentOverlap / ComputeOverlapWith
//                               Localizer.hpp:91-135, 276-348
//   pgslam::PairLoopCloser<T>   ProcessVertex core / CheckIcpResult / ComputeResidualError
//                               LoopCloser.hpp:83-110, 308-365
//   pgslam::LoopClosureBatch<T> LoopCloserMT queue (LoopCloserMT.hpp:26-67) processed as a batch
#pragma once
#include <chrono>
#include <deque>
#include <functional>
#include <memory>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "pointmatcher.hpp"

namespace pgslam {

template <typename T>
struct Types {
    using Time = std::chrono::time_point<std::chrono::high_resolution_clock>;
    using PM = PointMatcher<T>;
    using DP = typename PM::DataPoints;
    using DPPtr = std::shared_ptr<DP>;
    using Matrix = typename PM::Matrix;
    using CovMatrix = typename PM::Matrix;          // 6x6 (the reference uses Eigen::Matrix<T,6,6>)
    using ICP = typename PM::ICP;
    using ICPSequence = typename PM::ICPSequence;
    using TransformationPtr = std::shared_ptr<typename PM::Transformation>;
    using DataPointsFilters = typename PM::DataPointsFilters;
    using Label = typename DP::Label;
    using Labels = typename DP::Labels;
    using Vertex = size_t;                          // the reference uses a Boost.Graph descriptor

    struct Keyframe {
        size_t id;
        DPPtr cloud_ptr;
        Matrix T_world_kf;
        Matrix optimized_T_world_kf;
        Time update_time;
        //! (not in the reference) the keyframe's cloud in device memory, made when a local map is first assembled from it on
        //! the device (EnsureKeyframeOnDevice): keyframe clouds are immutable, so it is uploaded once and walked from HBM at
        //! every rebuild.  Shared by the copies of the keyframe made after that.
        std::shared_ptr<pgslam_amd::DeviceCloud<T>> device_cloud;
    };
    struct Constraint {
        enum Type { kOdomConstraint, kLoopConstraint };
        Type type;
        Matrix T_from_to;
        CovMatrix cov_from_to;
        T weight;
    };
};

#define IMPORT_PGSLAM_TYPES(TYPE)                                            \
    using Time = typename pgslam::Types<TYPE>::Time;                         \
    using PM = typename pgslam::Types<TYPE>::PM;                             \
    using DP = typename pgslam::Types<TYPE>::DP;                             \
    using DPPtr = typename pgslam::Types<TYPE>::DPPtr;                       \
    using Matrix = typename pgslam::Types<TYPE>::Matrix;                     \
    using CovMatrix = typename pgslam::Types<TYPE>::CovMatrix;               \
    using ICP = typename pgslam::Types<TYPE>::ICP;                           \
    using ICPSequence = typename pgslam::Types<TYPE>::ICPSequence;           \
    using TransformationPtr = typename pgslam::Types<TYPE>::TransformationPtr; \
    using DataPointsFilters = typename pgslam::Types<TYPE>::DataPointsFilters; \
    using Keyframe = typename pgslam::Types<TYPE>::Keyframe;                 \
    using Label = typename pgslam::Types<TYPE>::Label;                       \
    using Labels = typename pgslam::Types<TYPE>::Labels;                     \
    using Constraint = typename pgslam::Types<TYPE>::Constraint;             \
    using Vertex = typename pgslam::Types<TYPE>::Vertex;

//! LocalMap<T>::BuildCloudFromData: keyframes[0] is the reference keyframe (copied as is), every
//! other keyframe cloud is moved by T_refkf_world * optimized_T_world_kf and appended, in the given order
//! (the reference walks its buffer newest -> oldest, LocalMap.hpp:213-223).  One device pass.
template <typename T>
typename Types<T>::DP BuildLocalMapCloud(const std::vector<typename Types<T>::Keyframe> &keyframes)
{
    using PMT = PointMatcher<T>;
    using DP = typename Types<T>::DP;
    if (keyframes.empty()) return DP();
    const auto T_refkf_world = keyframes[0].optimized_T_world_kf.inverse();
    const int k = (int)keyframes.size();
    bool all_normals = true;
    long long total = 0;
    for (auto &kf : keyframes) { all_normals = all_normals && kf.cloud_ptr->descriptorExists("normals"); total += kf.cloud_ptr->getNbPoints(); }
    DP out;
    out.features = typename PMT::Matrix(4, (int)total);
    out.featureLabels = keyframes[0].cloud_ptr->featureLabels;
    for (long long j = 0; j < total; j++) out.features(3, (int)j) = T(1);
    if (all_normals) { out.descriptors = typename PMT::Matrix(3, (int)total); out.descriptorLabels.push_back(typename DP::Label("normals", 3)); }
    std::vector<const T *> xs(k), ns(k);
    std::vector<int> sx(k), sn(k), cnt(k);
    std::vector<double> Ts((size_t)16 * k, 0.0);
    for (int i = 0; i < k; i++) {
        const DP &c = *keyframes[i].cloud_ptr;
        xs[i] = c.xyzPtr(); sx[i] = c.xyzStride(); ns[i] = all_normals ? c.normalsPtr() : nullptr; sn[i] = all_normals ? c.normalsStride() : 3;
        cnt[i] = (int)c.getNbPoints();
        const auto Tk = (i == 0) ? PMT::Matrix::Identity(4, 4) : T_refkf_world * keyframes[i].optimized_T_world_kf;
        pgslam_amd::to_row_major16(Tk, Ts.data() + 16 * i);
    }
    pgicp_ctx *ctx = pgslam_amd::default_context();
    PMT::check(ctx, pgslam_amd::Abi<T>::local_map(ctx, k, xs.data(), all_normals ? ns.data() : nullptr, sx.data(), sn.data(), cnt.data(), Ts.data(),
                                                   out.features.data(), 4, all_normals ? out.descriptors.data() : nullptr, 3));
    return out;
}

//! the keyframe's cloud goes to device memory once (points and normals, host strides kept)
//! `ctx`: a context of the device the copy is to live on (the chain that will assemble maps from it); the buffers are plain
//! device memory, usable by every context of that device
template <typename T>
void EnsureKeyframeOnDevice(typename Types<T>::Keyframe &kf, pgicp_ctx *ctx = nullptr)
{
    if (kf.device_cloud) return;
    auto dc = std::make_shared<pgslam_amd::DeviceCloud<T>>();
    const auto &c = *kf.cloud_ptr;
    dc->upload(ctx ? ctx : pgslam_amd::default_context(), c.xyzPtr(), c.xyzStride(), c.descriptorExists("normals") ? c.normalsPtr() : nullptr, c.normalsStride(), (int)c.getNbPoints());
    kf.device_cloud = dc;
}
//! bytes a keyframe's device copy holds
template <typename T>
size_t DeviceKeyframeBytes(const typename Types<T>::Keyframe &kf)
{
    return kf.device_cloud ? sizeof(T) * (kf.device_cloud->cap_x + kf.device_cloud->cap_n) : 0;
}

//! BuildLocalMapCloud over keyframes whose clouds are resident (EnsureKeyframeOnDevice), INTO device memory: the same
//! device pass (pgicp_build_local_map), inputs and output in HBM; `out` is an assembly buffer the caller reuses.  Optionally
//! the assembled cloud is then moved by `T_then` (LocalMap::CloudInWorldFrame, LocalMap.hpp:95-98) in a second pass, as the
//! host flow does -- two roundings, the same two.  Same values as the host flow, bit for bit.
template <typename T>
void BuildLocalMapOnDevice(pgicp_ctx *ctx, const std::vector<typename Types<T>::Keyframe> &keyframes, pgslam_amd::DeviceCloud<T> &out,
                           const typename Types<T>::Matrix *T_then = nullptr)
{
    using PMT = PointMatcher<T>;
    if (keyframes.empty()) throw std::logic_error("BuildLocalMapOnDevice: no keyframes");
    const auto T_refkf_world = keyframes[0].optimized_T_world_kf.inverse();
    const int k = (int)keyframes.size();
    bool all_normals = true;
    long long total = 0;
    for (auto &kf : keyframes) {
        if (!kf.device_cloud) throw std::logic_error("BuildLocalMapOnDevice: a keyframe is not on the device");
        all_normals = all_normals && kf.device_cloud->hasNormals();
        total += kf.device_cloud->n;
    }
    std::vector<const T *> xs(k), ns(k);
    std::vector<int> sx(k), sn(k), cnt(k);
    std::vector<double> Ts((size_t)16 * k, 0.0);
    for (int i = 0; i < k; i++) {
        const auto &c = *keyframes[i].device_cloud;
        xs[i] = c.xyz; sx[i] = c.xs; ns[i] = all_normals ? c.nrm : nullptr; sn[i] = all_normals ? c.ns : 3;
        cnt[i] = c.n;
        const auto Tk = (i == 0) ? PMT::Matrix::Identity(4, 4) : T_refkf_world * keyframes[i].optimized_T_world_kf;
        pgslam_amd::to_row_major16(Tk, Ts.data() + 16 * i);
    }
    out.reserve(ctx, (int)total, 3, all_normals, 3);
    PMT::check(ctx, pgslam_amd::Abi<T>::local_map_dev(ctx, k, xs.data(), all_normals ? ns.data() : nullptr, sx.data(), sn.data(), cnt.data(), Ts.data(),
                                                       out.xyz, 3, all_normals ? out.nrm : nullptr, 3));
    if (T_then) {
        double T16[16];
        pgslam_amd::to_row_major16(*T_then, T16);
        PMT::check(ctx, pgslam_amd::Abi<T>::transform_dev(ctx, T16, out.xyz, 3, out.xyz, 3, (int)total, 0));
        if (all_normals) PMT::check(ctx, pgslam_amd::Abi<T>::transform_dev(ctx, T16, out.nrm, 3, out.nrm, 3, (int)total, 1));
    }
}

template <typename T>
class ScanLocalizer {
public:
    IMPORT_PGSLAM_TYPES(T)
    ScanLocalizer() : rigid_transformation_(PM::get().REG(Transformation).create("RigidTransformation")),
                  T_refkf_robot_(Matrix::Identity(4, 4)), T_world_robot_(Matrix::Identity(4, 4)),
                  last_input_T_world_robot_(Matrix::Identity(4, 4)), T_world_refkf_(Matrix::Identity(4, 4)),
                  overlap_threshold_(T(0.8)), minimal_overlap_(T(0.5)), has_map_(false) {}
    void SetOverlapThreshold(T v) { overlap_threshold_ = v; }
    void SetMinimalOverlapThreshold(T v) { minimal_overlap_ = v; }
    //! Localizer.hpp:54-71 (the YAML text is kept to re-create temporary ICP objects)
    void SetIcpConfigFromString(const std::string &yaml)
    {
        icp_config_buffer_ = yaml;
        temp_icp_.reset();
        std::istringstream iss(icp_config_buffer_);
        icp_sequence_.loadFromYaml(iss);
    }
    //! Localizer.hpp:148,168,254: the local map (already in the reference keyframe's frame) goes to the device
    void SetLocalMap(const DP &local_map_cloud, const Matrix &optimized_T_world_refkf)
    {
        icp_sequence_.setMap(local_map_cloud);
        T_world_refkf_ = optimized_T_world_refkf;
        has_map_ = true;
    }
    //! Localizer.hpp:91-135 without the graph bookkeeping: filters, sensor->robot, odometry guess, ICP, world pose
    Matrix ProcessData(const Matrix &input_T_world_robot, const Matrix &input_T_robot_sensor, DPPtr input_cloud_ptr)
    {
        input_cloud_ptr_ = input_cloud_ptr;
        input_filters_.apply(*input_cloud_ptr_);
        (*input_cloud_ptr_) = rigid_transformation_->compute(*input_cloud_ptr_, input_T_robot_sensor);
        if (!has_map_) throw std::logic_error("[Localizer] no local map set");
        const Matrix input_dT_robot = last_input_T_world_robot_.inverse() * input_T_world_robot;
        const Matrix input_T_refkf_robot = T_refkf_robot_ * input_dT_robot;
        T_refkf_robot_ = icp_sequence_(*input_cloud_ptr_, input_T_refkf_robot);
        T_world_robot_ = T_world_refkf_ * T_refkf_robot_;
        last_input_T_world_robot_ = input_T_world_robot;
        return T_refkf_robot_;
    }
    //! Localizer.hpp:276-279
    T ComputeCurrentOverlap() { return icp_sequence_.errorMinimizer->getOverlap(); }
    Matrix CurrentCovariance() { return icp_sequence_.errorMinimizer->getCovariance(); }
    bool IsOverlapEnough(T overlap) const { return overlap >= overlap_threshold_; }
    //! Localizer.hpp:282-348 with the candidate local map given in WORLD frame: index build + one
    //! findClosests + outlier weights + error elements -> weightedPointUsedRatio, as ONE device pass
    T ComputeOverlapWith(const DP &candidate_map_in_world_frame)
    {
        return ComputeOverlapOf(*input_cloud_ptr_, T_world_robot_, candidate_map_in_world_frame);
    }
    //! the same partial chain for an explicit (reading in robot frame, robot pose, map in world frame) triple
    T ComputeOverlapOf(const DP &reading_in, const Matrix &T_world_robot, const DP &candidate_map_in_world_frame)
    {
        // The reference builds a fresh ICP object from the YAML for every call (Localizer.hpp:309-311); here an ICP object
        // owns a device context (stream, pinned buffers, scratch), so the one temporary is kept and re-made only when the
        // configuration changes -- same chain, same result, without a context creation per scan.
        PrepareOverlapReference(candidate_map_in_world_frame);
        return ComputeOverlapAgainstPrepared(reading_in, T_world_robot);
    }
    //! the reference half of ComputeOverlapOf (Localizer.hpp:313-317): reference filters + matcher->init.  A caller that
    //! asks about the SAME candidate map scan after scan (Localizer's neighbour composition, unchanged while the
    //! graph is) prepares it once and calls ComputeOverlapAgainstPrepared per scan: same index, same result, without
    //! assembling, uploading and indexing the map every time.
    void PrepareOverlapReference(const DP &candidate_map_in_world_frame)
    {
        if (!temp_icp_) {
            temp_icp_.reset(new typename PM::ICP());
            std::istringstream iss(icp_config_buffer_);
            temp_icp_->loadFromYaml(iss);
        }
        typename PM::ICP &temp_icp = *temp_icp_;
        DP reference(candidate_map_in_world_frame);
        temp_icp.referenceDataPointsFilters.init();
        temp_icp.referenceDataPointsFilters.apply(reference);
        temp_icp.matcher->init(reference);
        // a SurfaceNormalOutlierFilter compares descriptors of both clouds: the stage-by-stage chain below needs the reference
        if (temp_icp.hasNormalFilter()) prepared_reference_ = std::move(reference); else prepared_reference_ = DP();
    }
    //! the same for a candidate map that is in device memory already (assembled there from resident keyframe clouds): no
    //! upload; only when the chain's reference filters change nothing (false: the caller takes the host flow)
    bool PrepareOverlapReference(const pgslam_amd::DeviceCloud<T> &candidate_map_in_world_frame)
    {
        if (!temp_icp_) {
            temp_icp_.reset(new typename PM::ICP());
            std::istringstream iss(icp_config_buffer_);
            temp_icp_->loadFromYaml(iss);
        }
        if (!temp_icp_->referenceDataPointsFilters.allIdentity() || temp_icp_->hasNormalFilter()) return false;
        temp_icp_->matcher->initDevice(candidate_map_in_world_frame, 0);
        return true;
    }
    //! the context the probe's chain computes on (device-side assembly of its reference runs on the same stream)
    pgicp_ctx *OverlapContext()
    {
        if (!temp_icp_) {
            temp_icp_.reset(new typename PM::ICP());
            std::istringstream iss(icp_config_buffer_);
            temp_icp_->loadFromYaml(iss);
        }
        return temp_icp_->ctx;
    }
    //! the reading half (Localizer.hpp:319-334) against the reference prepared last
    T ComputeOverlapAgainstPrepared(const DP &reading_in, const Matrix &T_world_robot)
    {
        if (!temp_icp_) throw std::logic_error("ComputeOverlapAgainstPrepared: no reference prepared");
        typename PM::ICP &temp_icp = *temp_icp_;
        DP reading(reading_in);
        temp_icp.readingDataPointsFilters.init();
        temp_icp.readingDataPointsFilters.apply(reading);
        // Localizer.hpp:325-326: the step filters act on the filtered reading in its own frame, before it is moved (they
        // are pure functions of the cloud here: pointmatcher.hpp alignOnMap)
        temp_icp.readingStepDataPointsFilters.init();
        temp_icp.readingStepDataPointsFilters.apply(reading);
        if (temp_icp.hasNormalFilter() && reading.normalsPtr() != nullptr && prepared_reference_.normalsPtr() != nullptr) {
            // the fused device pass knows no descriptor filter: stage by stage, exactly as Localizer.hpp:323-347 spells it --
            // move the reading, findClosests, outlierFilters.compute (the normal filter on the host), weightedPointUsedRatio
            const DP moved = rigid_transformation_->compute(reading, T_world_robot);
            const typename PM::Matches matches = temp_icp.matcher->findClosests(moved);
            const typename PM::OutlierWeights w = temp_icp.outlierFilters.compute(moved, prepared_reference_, matches);
            double sum = 0;
            for (int j = 0; j < w.cols(); j++) for (int i = 0; i < w.rows(); i++) sum += (double)w(i, j);
            return (T)(sum / ((double)w.rows() * (double)w.cols()));
        }
        double Tm[16], ratio = 0, residual = 0;
        pgslam_amd::to_row_major16(T_world_robot, Tm);
        temp_icp.pushParams();
        PM::check(temp_icp.ctx, pgslam_amd::Abi<T>::partial(temp_icp.ctx, temp_icp.matcher->mapId, reading.xyzPtr(), reading.xyzStride(),
                                                           (int)reading.getNbPoints(), Tm, &ratio, &residual));
        return (T)ratio;
    }
    //! the same for a reading that is on the device already (the localizer's input stage left it there): no copy of the
    //! cloud, no upload -- when the chain's reading filters change nothing; else the host cloud goes the usual way
    T ComputeOverlapAgainstPrepared(const typename PM::ICPChainBase::DeviceReading &reading, const Matrix &T_world_robot)
    {
        if (!temp_icp_) throw std::logic_error("ComputeOverlapAgainstPrepared: no reference prepared");
        typename PM::ICP &temp_icp = *temp_icp_;
        if (!reading || !temp_icp.deviceReadingEquivalent()) return ComputeOverlapAgainstPrepared(*reading.filtered, T_world_robot);
        double Tm[16], ratio = 0, residual = 0;
        pgslam_amd::to_row_major16(T_world_robot, Tm);
        temp_icp.pushParams();
        PM::check(temp_icp.ctx, pgslam_amd::Abi<T>::partial_dev(temp_icp.ctx, temp_icp.matcher->mapId, reading.dev, reading.xyzStride(), reading.points(), Tm,
                                                               &ratio, &residual));
        return (T)ratio;
    }
    //! ... with the matcher seeded from the correspondences `icp_ctx`'s last align of this very reading ended with (the two maps as
    //! concatenations of keyframe clouds: pgicp_partial_chain_seeded); the same result, bit for bit
    T ComputeOverlapAgainstPrepared(const typename PM::ICPChainBase::DeviceReading &reading, const Matrix &T_world_robot, pgicp_ctx *icp_ctx,
                                    const std::vector<int32_t> &src_start, const std::vector<int32_t> &dst_start)
    {
        if (!temp_icp_) throw std::logic_error("ComputeOverlapAgainstPrepared: no reference prepared");
        typename PM::ICP &temp_icp = *temp_icp_;
        if (!reading || !temp_icp.deviceReadingEquivalent() || !icp_ctx || src_start.size() != dst_start.size() + 1)
            return ComputeOverlapAgainstPrepared(reading, T_world_robot);
        double Tm[16], ratio = 0, residual = 0;
        pgslam_amd::to_row_major16(T_world_robot, Tm);
        temp_icp.pushParams();
        PM::check(temp_icp.ctx, pgslam_amd::Abi<T>::partial_seeded_dev(temp_icp.ctx, temp_icp.matcher->mapId, reading.dev, reading.xyzStride(), reading.points(), Tm,
                                                                      icp_ctx, (int)dst_start.size(), src_start.data(), dst_start.data(), &ratio, &residual));
        return (T)ratio;
    }
    const Matrix &T_refkf_robot() const { return T_refkf_robot_; }
    const Matrix &T_world_robot() const { return T_world_robot_; }
    ICPSequence &icp() { return icp_sequence_; }
    DataPointsFilters &input_filters() { return input_filters_; }

private:
    DPPtr input_cloud_ptr_;
    TransformationPtr rigid_transformation_;
    DataPointsFilters input_filters_;
    ICPSequence icp_sequence_;
    std::string icp_config_buffer_;
    std::unique_ptr<typename PM::ICP> temp_icp_;    // ComputeOverlapOf's temporary (kept: see there)
    DP prepared_reference_;                         // the probe's filtered reference, kept only for a chain with a descriptor filter
    Matrix T_refkf_robot_, T_world_robot_, last_input_T_world_robot_, T_world_refkf_;
    T overlap_threshold_, minimal_overlap_;
    bool has_map_;
};

//! LocalMap<T> without the graph: a circular window of keyframes whose BACK is the reference keyframe
//! (LocalMap.hpp:20-31, 116-127); the cloud is rebuilt by BuildCloudFromData (LocalMap.hpp:209-224).
template <typename T>
class LocalMap {
public:
    IMPORT_PGSLAM_TYPES(T)
    explicit LocalMap(size_t capacity) : capacity_(capacity) {}
    size_t Capacity() const { return capacity_; }
    bool HasCloud() const { return cloud_.features.cols() != 0; }
    const DP &Cloud() const { return cloud_; }
    const std::deque<Keyframe> &Data() const { return data_; }
    const Keyframe &ReferenceKeyframe() const { return data_.back(); }
    //! push_back on the circular buffer: the oldest keyframe drops out once the window is full
    void PushKeyframe(const Keyframe &kf) { data_.push_back(kf); if (data_.size() > capacity_) data_.pop_front(); }
    //! LocalMap.hpp:185-203: the FIRST keyframe at minimal distance (Metrics::Distance = translation norm)
    size_t FindClosest(const Matrix &T_world_x) const
    {
        size_t best = 0;
        double bd = dist(data_[0].optimized_T_world_kf, T_world_x);
        for (size_t i = 1; i < data_.size(); i++) { const double d = dist(data_[i].optimized_T_world_kf, T_world_x); if (d < bd) { bd = d; best = i; } }
        return best;
    }
    void MakeReference(size_t i) { std::swap(data_[i], data_.back()); }      // std::iter_swap, Localizer.hpp:220
    //! reference keyframe first, then newest -> oldest (LocalMap.hpp:213-223), one device pass
    void BuildCloudFromData()
    {
        std::vector<Keyframe> order;
        order.push_back(data_.back());
        for (size_t i = data_.size() - 1; i-- > 0;) order.push_back(data_[i]);
        cloud_ = BuildLocalMapCloud<T>(order);
    }
    //! the same order, for the device-side assembly (BuildLocalMapOnDevice)
    std::vector<Keyframe> AssemblyOrder() const
    {
        std::vector<Keyframe> order;
        order.push_back(data_.back());
        for (size_t i = data_.size() - 1; i-- > 0;) order.push_back(data_[i]);
        return order;
    }

private:
    static double dist(const Matrix &A, const Matrix &B)
    {
        double s = 0;
        for (int k = 0; k < 3; k++) { const double d = (double)B(k, 3) - (double)A(k, 3); s += d * d; }
        return std::sqrt(s);
    }
    size_t capacity_;
    std::deque<Keyframe> data_;
    DP cloud_;
};

//! Localizer::ProcessData + the graph-free part of UpdateAfterIcp (Localizer.hpp:91-135, 178-268) on a
//! sliding LocalMap: first cloud = first keyframe; overlap >= threshold keeps the keyframe set and makes the
//! keyframe closest to the robot the reference; otherwise the scan becomes a new keyframe.  A changed
//! composition rebuilds the cloud and calls setMap; a changed reference re-expresses the robot pose
//! (UpdateRefkfRobotPose).  `pgslam_amd/local_mapper.py` is the same policy with device-resident keyframes
//! and a background rebuild.
template <typename T>
class StreamingLocalizer {
public:
    IMPORT_PGSLAM_TYPES(T)
    explicit StreamingLocalizer(size_t capacity = 3)
        : local_map_(capacity), rigid_transformation_(PM::get().REG(Transformation).create("RigidTransformation")),
          T_refkf_robot_(Matrix::Identity(4, 4)), T_world_robot_(Matrix::Identity(4, 4)),
          last_input_T_world_robot_(Matrix::Identity(4, 4)), overlap_threshold_(T(0.8)), next_id_(0), rebuilds_(0) {}
    void SetOverlapThreshold(T v) { overlap_threshold_ = v; }
    void SetIcpConfigFromString(const std::string &yaml) { std::istringstream iss(yaml); icp_sequence_.loadFromYaml(iss); }
    DataPointsFilters &input_filters() { return input_filters_; }
    const LocalMap<T> &local_map() const { return local_map_; }
    const Matrix &T_world_robot() const { return T_world_robot_; }
    int rebuilds() const { return rebuilds_; }
    T last_overlap() const { return last_overlap_; }

    Matrix ProcessData(const Matrix &input_T_world_robot, const Matrix &input_T_robot_sensor, DPPtr cloud)
    {
        input_filters_.apply(*cloud);
        (*cloud) = rigid_transformation_->compute(*cloud, input_T_robot_sensor);
        if (!local_map_.HasCloud()) {                                       // ProcessFirstCloud, Localizer.hpp:137-152
            AddKeyframe(cloud, input_T_world_robot);
            Rebuild();
            T_refkf_robot_ = Matrix::Identity(4, 4);
            T_world_robot_ = input_T_world_robot;
            last_input_T_world_robot_ = input_T_world_robot;
            return T_world_robot_;
        }
        const Matrix input_dT_robot = last_input_T_world_robot_.inverse() * input_T_world_robot;
        T_refkf_robot_ = icp_sequence_(*cloud, T_refkf_robot_ * input_dT_robot);
        T_world_robot_ = local_map_.ReferenceKeyframe().optimized_T_world_kf * T_refkf_robot_;
        last_overlap_ = icp_sequence_.errorMinimizer->getOverlap();
        const size_t old_ref = local_map_.ReferenceKeyframe().id;
        bool changed = false;
        if (last_overlap_ >= overlap_threshold_) {
            const size_t closest = local_map_.FindClosest(T_world_robot_);
            if (local_map_.Data()[closest].id != old_ref) { local_map_.MakeReference(closest); changed = true; }
        } else {
            AddKeyframe(cloud, T_world_robot_);
            changed = true;
        }
        if (changed) {
            Rebuild();
            if (local_map_.ReferenceKeyframe().id != old_ref)
                T_refkf_robot_ = local_map_.ReferenceKeyframe().optimized_T_world_kf.inverse() * T_world_robot_;   // Localizer.hpp:273
        }
        last_input_T_world_robot_ = input_T_world_robot;
        return T_world_robot_;
    }

private:
    void AddKeyframe(DPPtr cloud, const Matrix &T_world_kf)
    {
        Keyframe kf;
        kf.id = next_id_++; kf.cloud_ptr = cloud; kf.T_world_kf = T_world_kf; kf.optimized_T_world_kf = T_world_kf;
        local_map_.PushKeyframe(kf);
    }
    void Rebuild() { local_map_.BuildCloudFromData(); icp_sequence_.setMap(local_map_.Cloud()); rebuilds_++; }
    LocalMap<T> local_map_;
    TransformationPtr rigid_transformation_;
    DataPointsFilters input_filters_;
    ICPSequence icp_sequence_;
    Matrix T_refkf_robot_, T_world_robot_, last_input_T_world_robot_;
    T overlap_threshold_, last_overlap_ = T(0);
    size_t next_id_;
    int rebuilds_;
};

template <typename T>
class PairLoopCloser {
public:
    IMPORT_PGSLAM_TYPES(T)
    struct Result {
        bool accepted;
        Matrix T_refkf_kf;
        CovMatrix cov;
        T overlap, residual;
        bool max_iterations_reached;
    };
    PairLoopCloser() : overlap_threshold_(T(0.8)), residual_error_threshold_(T(5000)) {}
    void SetOverlapThreshold(T v) { overlap_threshold_ = v; }
    void SetResidualErrorThreshold(T v) { residual_error_threshold_ = v; }
    void SetIcpConfigFromString(const std::string &yaml)
    {
        icp_config_buffer_ = yaml;
        temp_icp_.reset();
        std::istringstream iss(icp_config_buffer_);
        icp_.loadFromYaml(iss);
    }
    //! LoopCloser.hpp:95-109 for one candidate: ICP, CheckIcpResult, covariance for the optimizer
    Result ProcessCandidate(const DP &input_cloud, const DP &candidate_local_map_cloud, const Matrix &input_T_refkf_kf)
    {
        Result r;
        r.T_refkf_kf = icp_(input_cloud, candidate_local_map_cloud, input_T_refkf_kf);
        r.max_iterations_reached = icp_.getMaxNumIterationsReached();
        r.overlap = icp_.errorMinimizer->getOverlap();
        r.cov = icp_.errorMinimizer->getCovariance();
        r.residual = ComputeResidualError(input_cloud, candidate_local_map_cloud, r.T_refkf_kf);
        r.accepted = CheckIcpResult(r);
        return r;
    }
    //! The same for a candidate whose clouds are in device memory already (keyframe clouds resident, the candidate map
    //! assembled from them there): ICP::operator() and ComputeResidualError on device memory -- the same index builds, the same
    //! chain, no cloud crosses PCIe (LoopCloser.hpp:282-296 assembles the candidate map, :98 aligns, :343-365 checks).  Only for a
    //! chain whose data-point filters change nothing and that looks at no reading descriptor; `host_reference` makes the host copy
    //! of the candidate map if an observer (onAlign) asks for it.
    bool DeviceCandidateEquivalent() const
    {
        return icp_.referenceDataPointsFilters.allIdentity() && icp_.deviceReadingEquivalent();
    }
    Result ProcessCandidateOnDevice(const DPPtr &input_cloud, const pgslam_amd::DeviceCloud<T> &input_dev, const pgslam_amd::DeviceCloud<T> &candidate_dev,
                                    const Matrix &input_T_refkf_kf, std::function<DP()> host_reference)
    {
        Result r;
        typename PM::ICPChainBase::DeviceReading rd;
        rd.dev = input_dev.xyz; rd.filtered = input_cloud;
        r.T_refkf_kf = icp_.computeOnDevice(rd, candidate_dev, input_T_refkf_kf, host_reference);
        r.max_iterations_reached = icp_.getMaxNumIterationsReached();
        r.overlap = icp_.errorMinimizer->getOverlap();
        r.cov = icp_.errorMinimizer->getCovariance();
        // ComputeResidualError (LoopCloser.hpp:343-365): a second index on the RAW candidate cloud, the reading moved by the result
        if (!temp_icp_) {
            temp_icp_.reset(new typename PM::ICP());
            std::istringstream iss(icp_config_buffer_);
            temp_icp_->loadFromYaml(iss);
        }
        temp_icp_->matcher->initDevice(candidate_dev, 0);
        double Tm[16], ratio = 0, residual = 0;
        pgslam_amd::to_row_major16(r.T_refkf_kf, Tm);
        temp_icp_->pushParams();
        PM::check(temp_icp_->ctx, pgslam_amd::Abi<T>::partial_dev(temp_icp_->ctx, temp_icp_->matcher->mapId, input_dev.xyz, input_dev.xs, input_dev.n, Tm, &ratio, &residual));
        r.residual = (T)residual;
        r.accepted = CheckIcpResult(r);
        return r;
    }
    //! LoopCloser.hpp:308-340
    bool CheckIcpResult(const Result &r) const
    {
        if (r.max_iterations_reached) return false;
        if (r.overlap < overlap_threshold_) return false;
        if (r.residual > residual_error_threshold_) return false;
        return true;
    }
    //! LoopCloser.hpp:343-365: temp ICP, reading moved by the result, new index on the RAW candidate cloud,
    //! findClosests, outlier weights, getResidualError -- one device pass
    T ComputeResidualError(const DP &input_cloud, const DP &candidate_cloud, const Matrix &T_refkf_kf) const
    {
        if (!temp_icp_) {                               // LoopCloser.hpp:346-348 builds it from the YAML every time
            temp_icp_.reset(new typename PM::ICP());
            std::istringstream iss(icp_config_buffer_);
            temp_icp_->loadFromYaml(iss);
        }
        typename PM::ICP &temp_icp = *temp_icp_;
        temp_icp.matcher->init(candidate_cloud);
        if (temp_icp.hasNormalFilter() && input_cloud.normalsPtr() != nullptr && candidate_cloud.normalsPtr() != nullptr) {
            // with a descriptor filter in the chain: LoopCloser.hpp:352-362 stage by stage (transformations.apply, findClosests,
            // outlierFilters.compute -- the normal filter on the host --, getResidualError)
            DP reading(input_cloud);
            temp_icp.transformations.apply(reading, T_refkf_kf);
            const typename PM::Matches matches = temp_icp.matcher->findClosests(reading);
            const typename PM::OutlierWeights w = temp_icp.outlierFilters.compute(reading, candidate_cloud, matches);
            return temp_icp.errorMinimizer->getResidualError(reading, candidate_cloud, w, matches);
        }
        double Tm[16], ratio = 0, residual = 0;
        pgslam_amd::to_row_major16(T_refkf_kf, Tm);
        temp_icp.pushParams();
        PM::check(temp_icp.ctx, pgslam_amd::Abi<T>::partial(temp_icp.ctx, temp_icp.matcher->mapId, input_cloud.xyzPtr(), input_cloud.xyzStride(),
                                                           (int)input_cloud.getNbPoints(), Tm, &ratio, &residual));
        return (T)residual;
    }
    ICP &icp() { return icp_; }

private:
    ICP icp_;
    std::string icp_config_buffer_;
    mutable std::unique_ptr<typename PM::ICP> temp_icp_;   // ComputeResidualError's temporary (kept: an ICP object owns a device context)
    T overlap_threshold_, residual_error_threshold_;
};

//! The LoopCloserMT queue (LoopCloserMT.hpp:26-67) handled as a batch: all queued candidates of this
//! rank run concurrently on the GPU, results come back as the 512-byte edge records that ranks all-gather.
template <typename T>
class LoopClosureBatch {
public:
    IMPORT_PGSLAM_TYPES(T)
    //! reading_dev / reference_dev (optional): the same clouds in device memory (keyframe clouds resident, the candidate map
    //! assembled there) -- used, and nothing uploaded, when every candidate of a batch has both and the chain's data-point
    //! filters change nothing; `reference` may then be null (reference_points says how many points the device copy holds)
    struct Candidate {
        long long from_id, to_id; DPPtr reading, reference; Matrix T_init;
        std::shared_ptr<pgslam_amd::DeviceCloud<T>> reading_dev, reference_dev;
    };
    pgicp_ctx *Context() { return chain_.ctx; }
    bool DeviceCandidateEquivalent() const { return chain_.referenceDataPointsFilters.allIdentity() && chain_.deviceReadingEquivalent(); }
    size_t device_batches() const { return device_batches_; }
    void SetIcpConfigFromString(const std::string &yaml) { std::istringstream iss(yaml); chain_.loadFromYaml(iss); yaml_ = yaml; }
    void Add(const Candidate &c) { queue_.push_back(c); }
    void Clear() { queue_.clear(); }
    size_t Size() const { return queue_.size(); }
    //! indices of the queue this rank owns (deterministic LPT split, identical on every rank)
    std::vector<int> Shard(int world_size, int rank) const
    {
        std::vector<int64_t> cost(queue_.size());
        for (size_t i = 0; i < queue_.size(); i++) cost[i] = CostOf(queue_[i]);
        std::vector<int> idx(queue_.size());
        int n = 0;
        if (pgicp_shard_pairs((int)queue_.size(), cost.data(), world_size, rank, idx.data(), (int)idx.size(), &n) != PGICP_OK)
            throw std::runtime_error("pgicp_shard_pairs failed");
        idx.resize(n);
        return idx;
    }
    //! align the given queue entries as one device batch and evaluate CheckIcpResult
    std::vector<pgicp_edge> Run(const std::vector<int> &mine, T overlap_threshold = T(0.8), T residual_threshold = T(5000))
    {
        pgicp_ctx *ctx = chain_.ctx;
        chain_.pushParams();
        const int P = (int)mine.size();
        std::vector<pgicp_edge> edges(P);
        if (P == 0) return edges;
        std::vector<pgicp_problem> pr(P);
        std::vector<int> maps(P, -1), xs(P), ns(P), ms(P);
        std::vector<const T *> xyz(P), nrm(P);
        for (int k = 0; k < P; k++)
            if (queue_[mine[k]].reading && chain_.sensorNoiseApplies(*queue_[mine[k]].reading))
                throw std::runtime_error("LoopClosureBatch: a reading carries simpleSensorNoise (getOverlap's sensor-noise branch): process the pair with PairLoopCloser");
        bool on_device = DeviceCandidateEquivalent();
        for (int k = 0; k < P && on_device; k++) on_device = queue_[mine[k]].reading_dev && queue_[mine[k]].reference_dev && queue_[mine[k]].reference_dev->hasNormals();
        if (on_device) return RunOnDevice(mine, overlap_threshold, residual_threshold);
        // the chain's data-point filters act on copies, as ICP::operator() applies them (LoopCloser.hpp:98 -> reference
        // filters, then reading filters); a chain without filters reads the candidates' clouds where they lie
        std::vector<DP> ref_f, rd_f;
        const bool filt_ref = !chain_.referenceDataPointsFilters.empty(), filt_rd = !chain_.readingDataPointsFilters.empty();
        if (filt_ref) { ref_f.reserve(P); chain_.referenceDataPointsFilters.init(); }
        if (filt_rd) { rd_f.reserve(P); chain_.readingDataPointsFilters.init(); }
        auto reference_of = [&](int k) -> const DP & { return filt_ref ? ref_f[k] : *queue_[mine[k]].reference; };
        auto reading_of = [&](int k) -> const DP & { return filt_rd ? rd_f[k] : *queue_[mine[k]].reading; };
        for (int k = 0; k < P; k++) {
            const Candidate &c = queue_[mine[k]];
            if (filt_ref) { ref_f.push_back(*c.reference); chain_.referenceDataPointsFilters.apply(ref_f.back()); }
            if (filt_rd) { rd_f.push_back(*c.reading); chain_.readingDataPointsFilters.apply(rd_f.back()); }
            const DP &ref = reference_of(k);
            if (!ref.descriptorExists("normals")) throw std::runtime_error("LoopClosureBatch: a candidate reference has no 'normals' descriptor");
            xyz[k] = ref.xyzPtr(); xs[k] = ref.xyzStride();
            nrm[k] = ref.normalsPtr(); ns[k] = ref.normalsStride();
            ms[k] = (int)ref.getNbPoints();
        }
        // every candidate's reference is indexed in one call (ICP::operator() builds it inside, LoopCloser.hpp:98)
        PM::check(ctx, pgslam_amd::Abi<T>::map_create_batch(ctx, P, xyz.data(), xs.data(), nrm.data(), ns.data(), ms.data(), 1, maps.data()));
        // a SurfaceNormalOutlierFilter acts when the readings carry normals (without them its weights are ones):
        // all of the batch's readings or none -- one device chain per call
        int with_nrm = 0;
        if (chain_.hasNormalFilter()) for (int k = 0; k < P; k++) with_nrm += reading_of(k).normalsPtr() != nullptr ? 1 : 0;
        if (with_nrm != 0 && with_nrm != P)
            throw std::runtime_error("LoopClosureBatch: a SurfaceNormalOutlierFilter is configured and only some readings carry normals");
        // getOverlap() of a reading that carries `simpleSensorNoise` is computed from the ICP's LAST error elements; this call's
        // fused residual pass replaces them: such a candidate goes through PairLoopCloser::ProcessCandidate (the caller's choice --
        // LoopCloserMT does) instead of getting another overlap silently
        for (int k = 0; k < P; k++)
            if (chain_.sensorNoiseApplies(reading_of(k)))
                throw std::runtime_error("LoopClosureBatch: a reading carries simpleSensorNoise (getOverlap's sensor-noise branch): process the pair with PairLoopCloser");
        chain_.pushParams(with_nrm == P && P > 0 && chain_.hasNormalFilter());
        for (int k = 0; k < P; k++) {
            const Candidate &c = queue_[mine[k]];
            const DP &rd = reading_of(k);
            std::memset(&pr[k], 0, sizeof pr[k]);
            pr[k].map_id = maps[k]; pr[k].reading = rd.xyzPtr(); pr[k].stride = rd.xyzStride();
            pr[k].n = (int)rd.getNbPoints(); pr[k].mem = PGICP_HOST;
            if (with_nrm == P && chain_.hasNormalFilter()) { pr[k].normals = rd.normalsPtr(); pr[k].nstride = rd.normalsStride(); }
            pgslam_amd::to_row_major16(c.T_init, pr[k].T_init);
        }
        std::vector<double> Tout((size_t)16 * P);
        std::vector<pgicp_stats> st(P);
        // ICP::operator() of every pair (LoopCloser.hpp:98) and ComputeResidualError (LoopCloser.hpp:343-365) of every result
        // in one device call: the residual pass starts from the last iteration's correspondences
        std::vector<double> residual(P, 1.0 / 0.0);
        const int rc = sizeof(T) == 4 ? pgicp_align_residual_batch_f32(ctx, P, pr.data(), Tout.data(), st.data(), residual.data(), nullptr, nullptr)
                                      : pgicp_align_residual_batch_f64(ctx, P, pr.data(), Tout.data(), st.data(), residual.data(), nullptr, nullptr);
        if (rc != PGICP_OK && rc != PGICP_ERR_NO_MATCH && rc != PGICP_ERR_NAN && rc != PGICP_ERR_BOUND) PM::check(ctx, rc);
        for (int k = 0; k < P; k++) {
            const Candidate &c = queue_[mine[k]];
            pgicp_edge &e = edges[k];
            std::memset(&e, 0, sizeof e);
            e.from_id = c.from_id; e.to_id = c.to_id; e.status = st[k].status; e.iterations = st[k].iterations;
            e.max_iter_reached = st[k].max_iter_reached; e.overlap = st[k].overlap;
            std::memcpy(e.T_from_to, Tout.data() + 16 * k, sizeof e.T_from_to);
            std::memcpy(e.cov, st[k].cov, sizeof e.cov);
            e.residual = residual[k];
            e.accepted = pgicp_check_icp_result(&st[k], residual[k], (double)overlap_threshold, (double)residual_threshold);
            pgicp_map_destroy(ctx, maps[k]);
        }
        return edges;
    }
    //! Every rank's edges in queue order (OptimizerMT::Main drains them into one solve, OptimizerMT.hpp:59-65): ONE
    //! ncclAllGather over RCCL (pgicp_allgather_edges).  `mine` / `local` as given to / returned by Run; the block size
    //! is derived from the same deterministic split on every rank.  Entries nobody reported have from_id -1.
    std::vector<pgicp_edge> Gather(pgicp_comm *comm, const std::vector<int> &mine, const std::vector<pgicp_edge> &local) const
    {
        int world = 1, rank = 0, slots = 0;
        if (pgicp_comm_info(comm, &world, &rank) != PGICP_OK) throw std::runtime_error("LoopClosureBatch::Gather: no communicator");
        std::vector<int64_t> cost(queue_.size());
        for (size_t i = 0; i < queue_.size(); i++) cost[i] = CostOf(queue_[i]);
        if (pgicp_shard_slots((int)queue_.size(), cost.data(), world, &slots) != PGICP_OK) throw std::runtime_error("pgicp_shard_slots failed");
        std::vector<pgicp_edge> all(queue_.size());
        if (pgicp_allgather_edges(comm, local.data(), mine.data(), (int)local.size(), slots, (int)queue_.size(), all.data()) != PGICP_OK)
            throw std::runtime_error(std::string("pgicp_allgather_edges: ") + pgicp_comm_last_error());
        return all;
    }

private:
    static int64_t CostOf(const Candidate &c)
    {
        return (int64_t)(c.reading ? c.reading->getNbPoints() : c.reading_dev ? (size_t)c.reading_dev->n : 0) +
               (int64_t)(c.reference ? c.reference->getNbPoints() : c.reference_dev ? (size_t)c.reference_dev->n : 0);
    }
    //! Run for candidates that live in device memory: the same two calls (pgicp_map_create_batch, pgicp_align_residual_batch)
    //! with mem = PGICP_DEVICE
    std::vector<pgicp_edge> RunOnDevice(const std::vector<int> &mine, T overlap_threshold, T residual_threshold)
    {
        pgicp_ctx *ctx = chain_.ctx;
        const int P = (int)mine.size();
        std::vector<pgicp_edge> edges(P);
        std::vector<pgicp_problem> pr(P);
        std::vector<int> maps(P, -1), xs(P), ns(P), ms(P);
        std::vector<const T *> xyz(P), nrm(P);
        for (int k = 0; k < P; k++) {
            const auto &ref = *queue_[mine[k]].reference_dev;
            xyz[k] = ref.xyz; xs[k] = ref.xs; nrm[k] = ref.nrm; ns[k] = ref.ns; ms[k] = ref.n;
        }
        // whatever throws below (a partially failed map_create_batch, pushParams, PM::check): the maps made so far go back to the pool
        struct MapGuard {
            pgicp_ctx *ctx; std::vector<int> &ids;
            ~MapGuard() { for (int &m : ids) if (m >= 0) { pgicp_map_destroy(ctx, m); m = -1; } }
        } guard{ctx, maps};
        PM::check(ctx, sizeof(T) == 4 ? pgicp_map_create_batch_f32(ctx, P, (const float *const *)xyz.data(), xs.data(), (const float *const *)nrm.data(), ns.data(), ms.data(), PGICP_DEVICE, 1, maps.data())
                                      : pgicp_map_create_batch_f64(ctx, P, (const double *const *)xyz.data(), xs.data(), (const double *const *)nrm.data(), ns.data(), ms.data(), PGICP_DEVICE, 1, maps.data()));
        chain_.pushParams();
        for (int k = 0; k < P; k++) {
            const Candidate &c = queue_[mine[k]];
            std::memset(&pr[k], 0, sizeof pr[k]);
            pr[k].map_id = maps[k]; pr[k].reading = c.reading_dev->xyz; pr[k].stride = c.reading_dev->xs; pr[k].n = c.reading_dev->n; pr[k].mem = PGICP_DEVICE;
            pgslam_amd::to_row_major16(c.T_init, pr[k].T_init);
        }
        std::vector<double> Tout((size_t)16 * P), residual(P, 1.0 / 0.0);
        std::vector<pgicp_stats> st(P);
        const int rc = sizeof(T) == 4 ? pgicp_align_residual_batch_f32(ctx, P, pr.data(), Tout.data(), st.data(), residual.data(), nullptr, nullptr)
                                      : pgicp_align_residual_batch_f64(ctx, P, pr.data(), Tout.data(), st.data(), residual.data(), nullptr, nullptr);
        if (rc != PGICP_OK && rc != PGICP_ERR_NO_MATCH && rc != PGICP_ERR_NAN && rc != PGICP_ERR_BOUND) PM::check(ctx, rc);
        for (int k = 0; k < P; k++) {
            const Candidate &c = queue_[mine[k]];
            pgicp_edge &e = edges[k];
            std::memset(&e, 0, sizeof e);
            e.from_id = c.from_id; e.to_id = c.to_id; e.status = st[k].status; e.iterations = st[k].iterations;
            e.max_iter_reached = st[k].max_iter_reached; e.overlap = st[k].overlap;
            std::memcpy(e.T_from_to, Tout.data() + 16 * k, sizeof e.T_from_to);
            std::memcpy(e.cov, st[k].cov, sizeof e.cov);
            e.residual = residual[k];
            e.accepted = pgicp_check_icp_result(&st[k], residual[k], (double)overlap_threshold, (double)residual_threshold);
        }
        device_batches_++;
        return edges;
    }
    typename PM::ICP chain_;
    std::string yaml_;
    std::vector<Candidate> queue_;
    size_t device_batches_ = 0;
};

}  // namespace pgslam
