// PointCloudOdometry.cc -- control flow of point_cloud_odometry/src/PointCloudOdometry.cc:136-322 on the HIP path.
#include "PointCloudOdometry.hpp"

#include <cstdlib>
#include <cstring>

namespace locus_hip {

PointCloudOdometry::PointCloudOdometry(lh_ctx* ctx) : ctx_(ctx) {
  query_.reset(new PointCloudF);
  reference_.reset(new PointCloudF);
  query_trans_.reset(new PointCloudF);
  for (int i = 0; i < 9; i++) imu_delta_[i] = (i % 4 == 0);
  for (int i = 0; i < 16; i++) odometry_delta_[i] = (i % 5 == 0);
}
PointCloudOdometry::~PointCloudOdometry() {}

bool PointCloudOdometry::Initialize(const Config& cfg) {  // Initialize -> LoadParameters + SetupICP (:46-134)
  params_ = cfg;
  transform_thresholding_ = cfg.transform_thresholding;
  max_translation_ = cfg.max_translation;
  max_rotation_ = cfg.max_rotation;
  b_is_flat_ground_assumption_ = cfg.b_is_flat_ground_assumption;
  integrated_estimate_ = cfg.initial_pose;
  if (b_is_flat_ground_assumption_) integrated_estimate_.rotation = gu::Rot3(0, 0, integrated_estimate_.rotation.Yaw());
  return SetupICP();
}

bool PointCloudOdometry::SetupICP() {  // :136-204
  if (params_.registration_method == "ndt") {  // RegistrationMethod::NDT branch
    std::shared_ptr<NdtHip> ndt(new NdtHip(ctx_));
    ndt->setTransformationEpsilon(params_.icp_tf_epsilon);
    ndt->setMaxCorrespondenceDistance(params_.icp_corr_dist);
    ndt->setMaximumIterations((int)params_.icp_iterations);
    ndt->setRANSACIterations(0);
    ndt->setNumThreads(params_.num_threads);
    ndt->enableTimingOutput(params_.enable_timing_output);
    icp_ = ndt;
    return true;
  }
  if (params_.registration_method != "gicp" && params_.registration_method != "gicp_hip")
    throw std::runtime_error("No such Registration mode or not implemented yet " + params_.registration_method);
  std::shared_ptr<MultithreadedGicpHip> gicp(new MultithreadedGicpHip(ctx_));
  icp_ = gicp;
  icp_->setTransformationEpsilon(params_.icp_tf_epsilon);
  icp_->setMaxCorrespondenceDistance(params_.icp_corr_dist);
  icp_->setMaximumIterations((int)params_.icp_iterations);
  icp_->setRANSACIterations(0);
  icp_->setNumThreads(params_.num_threads);
  icp_->enableTimingOutput(params_.enable_timing_output);
  gicp->RecomputeTargetCovariance(params_.recompute_covariances);
  gicp->RecomputeSourceCovariance(params_.recompute_covariances);
  icp_->setEuclideanFitnessEpsilon(0.005);
  return true;
}

void PointCloudOdometry::EnableOdometryIntegration() { b_use_odometry_integration_ = true; b_use_imu_integration_ = false; }
void PointCloudOdometry::EnableImuIntegration() { b_use_imu_integration_ = true; b_use_odometry_integration_ = false; }
void PointCloudOdometry::DisableSensorIntegration() { b_use_imu_integration_ = false; b_use_odometry_integration_ = false; }

bool PointCloudOdometry::SetLidar(const PointCloudF& points) { points_ = points; return true; }                   // :221-225
bool PointCloudOdometry::SetImuDelta(const double d[9]) { memcpy(imu_delta_, d, sizeof(imu_delta_)); return true; }  // :227-230
bool PointCloudOdometry::SetOdometryDelta(const double d[16]) { memcpy(odometry_delta_, d, sizeof(odometry_delta_)); return true; }

bool PointCloudOdometry::UpdateEstimate() {  // :237-247
  if (!initialized_) {
    *query_ = points_;
    initialized_ = true;
    return false;
  } else {
    *reference_ = *query_;
    *query_ = points_;
    return UpdateICP();
  }
}

bool PointCloudOdometry::UpdateICP() {  // :249-322
  query_trans_->clear();
  double prior[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  bool have_prior = false;
  if (b_use_imu_integration_) {  // imu_prior_ = [imu_delta_ 0; 0 1]
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) prior[r * 4 + c] = imu_delta_[r * 3 + c];
    have_prior = true;
  } else if (b_use_odometry_integration_) {  // transformAsMatrix -> float -> double (:256-259)
    for (int i = 0; i < 16; i++) prior[i] = (double)(float)odometry_delta_[i];
    have_prior = true;
  }
  *query_trans_ = *query_;
  if (have_prior) {
    // pcl::transformPointCloud(*query_, *query_trans_, prior) with an Eigen::Matrix4d (:255, :260): xyz only; PCL promotes the
    // float point to the matrix' scalar, evaluates the row in DOUBLE left to right and rounds once to float
    for (auto& p : query_trans_->points) {
      const double x = p.x, y = p.y, z = p.z;
      p.x = static_cast<float>(prior[0] * x + prior[1] * y + prior[2] * z + prior[3]);
      p.y = static_cast<float>(prior[4] * x + prior[5] * y + prior[6] * z + prior[7]);
      p.z = static_cast<float>(prior[8] * x + prior[9] * y + prior[10] * z + prior[11]);
    }
  }
  // The reference of this update is the query of the last one (copyPointCloud(*query_, *reference_), :243).  When that query went
  // to the registration object unchanged (no sensor prior was applied to it), it is already on the GPU as the previous SOURCE: it
  // is promoted to target there (lh_gicp_promote_source_to_target: same data, same index build inside align, identical result)
  // instead of being uploaded a second time.  LOCUS_HIP_NO_PROMOTE=1 restores the two uploads per scan (host_check compares both).
  MultithreadedGicpHip* gicp = dynamic_cast<MultithreadedGicpHip*>(icp_.get());
  static const bool no_promote = []() { const char* e = getenv("LOCUS_HIP_NO_PROMOTE"); return e && atoi(e) != 0; }();
  if (gicp && device_source_is_last_query_ && device_promotion_ && !no_promote) gicp->promoteSourceToTarget(reference_);
  else icp_->setInputTarget(reference_);
  icp_->setInputSource(query_trans_);
  device_source_is_last_query_ = !have_prior;   // (with a prior the uploaded source is the MOVED query, not what the next update's reference is)
  icp_->align(icpAlignedPointsOdometry_);
  double T[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) T[r * 4 + c] = (double)icp_->T(r, c);
  if (have_prior) {  // T = T * prior (:271-275)
    double R[16];
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += T[r * 4 + k] * prior[k * 4 + c];
        R[r * 4 + c] = s;
      }
    memcpy(T, R, sizeof(T));
  }
  if (b_is_flat_ground_assumption_) {  // :277-292 (tf::Matrix3x3::getRPY yaw)
    double yaw = gu::Rot3(T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]).Yaw();
    incremental_estimate_.translation = gu::Vec3(T[3], T[7], 0);
    incremental_estimate_.rotation = gu::Rot3(cos(yaw), -sin(yaw), 0, sin(yaw), cos(yaw), 0, 0, 0, 1);
  } else {
    incremental_estimate_.translation = gu::Vec3(T[3], T[7], T[11]);
    incremental_estimate_.rotation = gu::Rot3(T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]);
  }
  if (!transform_thresholding_ || (incremental_estimate_.translation.Norm() <= max_translation_ &&
                                   incremental_estimate_.rotation.ToEulerZYX().Norm() <= max_rotation_)) {  // :305-316
    integrated_estimate_ = gu::PoseUpdate(integrated_estimate_, incremental_estimate_);
  }
  is_healthy_ = true;
  return true;
}

void PointCloudOdometry::SetFlatGroundAssumptionValue(const bool& value) {  // :324-331
  b_is_flat_ground_assumption_ = value;
  if (value) integrated_estimate_.rotation = gu::Rot3(0, 0, integrated_estimate_.rotation.Yaw());
}

bool PointCloudOdometry::GetLastPointCloud(PointCloudF::Ptr& out) const {  // :341-352
  if (!out || query_->empty()) return false;
  *out = *query_;
  return true;
}

PointCloudOdometry::Diagnostics PointCloudOdometry::GetDiagnostics() const {  // :367-380
  return is_healthy_ ? Diagnostics{0, "Healthy"} : Diagnostics{2, "Non healthy - Null-pointer error."};
}

}  // namespace locus_hip
