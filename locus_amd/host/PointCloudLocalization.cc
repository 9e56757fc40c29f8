// PointCloudLocalization.cc -- control flow of point_cloud_localization/src/PointCloudLocalization.cc:174-541 on the HIP path.
#include "PointCloudLocalization.hpp"

#include <cmath>
#include <cstring>

namespace locus_hip {

PointCloudLocalization::PointCloudLocalization(lh_ctx* ctx) : ctx_(ctx) {
  memset(icp_covariance_, 0, sizeof(icp_covariance_));
  memset(observability_matrix_, 0, sizeof(observability_matrix_));
}
PointCloudLocalization::~PointCloudLocalization() {}

bool PointCloudLocalization::Initialize(const Config& cfg) {
  params_ = cfg;
  b_is_flat_ground_assumption_ = cfg.b_is_flat_ground_assumption;
  integrated_estimate_ = cfg.initial_pose;
  if (b_is_flat_ground_assumption_) integrated_estimate_.rotation = gu::Rot3(0, 0, integrated_estimate_.rotation.Yaw());
  return SetupICP();
}

bool PointCloudLocalization::SetupICP() {  // :223-289
  if (params_.registration_method == "ndt") {  // RegistrationMethod::NDT branch
    std::shared_ptr<NdtHip> ndt(new NdtHip(ctx_));
    ndt->setTransformationEpsilon(params_.tf_epsilon);
    ndt->setMaxCorrespondenceDistance(params_.corr_dist);
    ndt->setMaximumIterations(params_.iterations);
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
  icp_->setTransformationEpsilon(params_.tf_epsilon);
  icp_->setMaxCorrespondenceDistance(params_.corr_dist);
  icp_->setMaximumIterations(params_.iterations);
  gicp->setMaximumOptimizerIterations(50);  // :238
  icp_->setRANSACIterations(0);
  icp_->setNumThreads(params_.num_threads);
  icp_->enableTimingOutput(params_.enable_timing_output);
  gicp->RecomputeTargetCovariance(params_.recompute_covariance_local_map);
  gicp->RecomputeSourceCovariance(params_.recompute_covariance_scan);
  return true;
}

bool PointCloudLocalization::MotionUpdate(const gu::Transform3& incremental_odom) {  // :174-179
  incremental_estimate_ = incremental_odom;
  return true;
}

// pcl::transformPointCloudWithNormals with a double 4x4 cast to float (PCL casts the matrix to the point scalar)
static void TransformWithNormals(const PointCloudF& in, PointCloudF* out, const gu::Transform3& e) {
  float R[9], t[3] = {(float)e.translation.x, (float)e.translation.y, (float)e.translation.z};
  for (int i = 0; i < 9; i++) R[i] = (float)e.rotation.m[i];
  if (out != &in) *out = in;
  for (auto& p : out->points) {
    float x = p.x, y = p.y, z = p.z, nx = p.normal_x, ny = p.normal_y, nz = p.normal_z;
    p.x = ((R[0] * x + R[1] * y) + R[2] * z) + t[0];
    p.y = ((R[3] * x + R[4] * y) + R[5] * z) + t[1];
    p.z = ((R[6] * x + R[7] * y) + R[8] * z) + t[2];
    p.normal_x = (R[0] * nx + R[1] * ny) + R[2] * nz;
    p.normal_y = (R[3] * nx + R[4] * ny) + R[5] * nz;
    p.normal_z = (R[6] * nx + R[7] * ny) + R[8] * nz;
  }
}

bool PointCloudLocalization::TransformPointsToFixedFrame(const PointCloudF& points, PointCloudF* out) const {  // :181-200
  if (out == NULL) return false;
  TransformWithNormals(points, out, gu::PoseUpdate(integrated_estimate_, incremental_estimate_));
  return true;
}
bool PointCloudLocalization::TransformPointsToSensorFrame(const PointCloudF& points, PointCloudF* out) const {  // :202-221
  if (out == NULL) return false;
  TransformWithNormals(points, out, gu::PoseInverse(gu::PoseUpdate(integrated_estimate_, incremental_estimate_)));
  return true;
}

bool PointCloudLocalization::MeasurementUpdate(const PointCloudF::Ptr& query, const PointCloudF::Ptr& reference,
                                               PointCloudF* aligned_query) {  // :291-427
  if (aligned_query == NULL) {
    is_healthy_ = false;
    return false;
  }
  icp_->setInputSource(query);
  icp_->setInputTarget(reference);
  // align (:309), transformPointCloudWithNormals (:325), the 1-NN loop (:327-336) and ComputeAp + the covariance conditioning (:398-421) are ONE
  // call on the two clouds the setters uploaded (lh_gicp_measurement_update): no host point loop, no second upload, one wait.  A registration
  // object without that entry point (NDT) gets the steps one by one.
  const bool want_info = params_.compute_icp_observability || (params_.compute_icp_covariance && params_.icp_covariance_method == 1);
  RegistrationHip::Measurement meas;
  std::vector<size_t> correspondences;
  const bool fused = icp_->measurementUpdate(aligned_query, &correspondences, want_info, params_.icp_max_covariance, &meas);
  if (!fused) {
    PointCloudF icpAlignedPointsLocalization_;
    icp_->align(icpAlignedPointsLocalization_);
  }
  const float* T = icp_->getFinalTransformation();  // column-major
  auto Tm = [&](int r, int c) { return (double)T[c * 4 + r]; };
  gu::Transform3 Tt;
  Tt.translation = gu::Vec3(Tm(0, 3), Tm(1, 3), Tm(2, 3));
  Tt.rotation = gu::Rot3(Tm(0, 0), Tm(0, 1), Tm(0, 2), Tm(1, 0), Tm(1, 1), Tm(1, 2), Tm(2, 0), Tm(2, 1), Tm(2, 2));
  if (!fused) {
    TransformWithNormals(*query, aligned_query, Tt);   // transformPointCloudWithNormals(*query, *aligned_query, T) (:325)
    icp_->nearestTargetIndices(*aligned_query, &correspondences);   // (:327-336)
  }

  UpdatePoses();
  if (params_.compute_icp_observability) {  // :384-396
    double evec[36], eval[6];
    if (fused && meas.have_information) {
      memcpy(observability_matrix_, meas.Ap, sizeof(meas.Ap));
      EigenDecomp6x6(meas.Ap, evec, eval);
    } else
      ComputeIcpObservability(*query, *reference, correspondences, T, evec, eval, observability_matrix_);
  }
  {
    std::lock_guard<std::mutex> lock(icp_covariance_mutex_);  // :398-421
    memset(icp_covariance_, 0, sizeof(icp_covariance_));
    if (params_.compute_icp_covariance && params_.icp_covariance_method == 1) {
      if (fused && meas.have_information) {
        memcpy(icp_covariance_, meas.covariance, sizeof(meas.covariance));
        condition_number_ = meas.condition_number;
      } else
        ComputePoint2PlaneICPCovariance(*query, *reference, correspondences, T, icp_covariance_);
    }
  }
  is_healthy_ = true;
  return true;
}

// the pose chain of MeasurementUpdate (:338-382) from the registration object's final transformation
void PointCloudLocalization::UpdatePoses() {
  const float* T = icp_->getFinalTransformation();  // column-major
  auto Tm = [&](int r, int c) { return (double)T[c * 4 + r]; };
  gu::Transform3 Tt;
  Tt.translation = gu::Vec3(Tm(0, 3), Tm(1, 3), Tm(2, 3));
  Tt.rotation = gu::Rot3(Tm(0, 0), Tm(0, 1), Tm(0, 2), Tm(1, 0), Tm(1, 1), Tm(1, 2), Tm(2, 0), Tm(2, 1), Tm(2, 2));
  gu::Transform3 pose_update;
  if (b_is_flat_ground_assumption_) {  // :340-353
    double yaw = Tt.rotation.Yaw();
    pose_update.translation = gu::Vec3(Tm(0, 3), Tm(1, 3), 0);
    pose_update.rotation = gu::Rot3(cos(yaw), -sin(yaw), 0, sin(yaw), cos(yaw), 0, 0, 0, 1);
  } else {
    pose_update = Tt;
  }
  if (!params_.transform_thresholding || (pose_update.translation.Norm() <= params_.max_translation &&
                                          pose_update.rotation.ToEulerZYX().Norm() <= params_.max_rotation)) {  // :369-379
    incremental_estimate_ = gu::PoseUpdate(incremental_estimate_, pose_update);
  }
  integrated_estimate_ = gu::PoseUpdate(integrated_estimate_, incremental_estimate_);  // :381-382

}

// ---- the device-resident surface ----------------------------------------------------------------------------------------------------------
static void Pose16(const gu::Transform3& e, float T16[16]) {   // column-major float 4x4, the cast TransformWithNormals makes
  for (int i = 0; i < 16; i++) T16[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T16[c * 4 + r] = (float)e.rotation.m[r * 3 + c];
  T16[12] = (float)e.translation.x; T16[13] = (float)e.translation.y; T16[14] = (float)e.translation.z;
}
bool PointCloudLocalization::TransformPointsToFixedFrame(const lh_cloud* points, lh_cloud** out) const {  // :181-200
  if (!points || !out) return false;
  float T16[16];
  Pose16(gu::PoseUpdate(integrated_estimate_, incremental_estimate_), T16);
  return lh_cloud_transform(points, T16, 1, out) == LH_OK;
}
bool PointCloudLocalization::TransformPointsToSensorFrame(const lh_cloud* points, lh_cloud** out) const {  // :202-221
  if (!points || !out) return false;
  float T16[16];
  Pose16(gu::PoseInverse(gu::PoseUpdate(integrated_estimate_, incremental_estimate_)), T16);
  return lh_cloud_transform(points, T16, 1, out) == LH_OK;
}
bool PointCloudLocalization::MeasurementUpdate(lh_cloud* query, lh_cloud* reference, lh_cloud** aligned_query) {  // :291-427
  MultithreadedGicpHip* gicp = dynamic_cast<MultithreadedGicpHip*>(icp_.get());
  if (aligned_query == NULL || !gicp || !query || !reference) {
    is_healthy_ = false;
    return false;
  }
  gicp->setInputSourceCloud(query);
  gicp->setInputTargetCloud(reference);
  const bool want_info = params_.compute_icp_observability || (params_.compute_icp_covariance && params_.icp_covariance_method == 1);
  RegistrationHip::Measurement meas;
  gicp->measurementUpdateCloud(aligned_query, want_info, params_.icp_max_covariance, &meas);
  UpdatePoses();
  if (params_.compute_icp_observability && meas.have_information) {
    double evec[36], eval[6];
    memcpy(observability_matrix_, meas.Ap, sizeof(meas.Ap));
    EigenDecomp6x6(meas.Ap, evec, eval);
  }
  {
    std::lock_guard<std::mutex> lock(icp_covariance_mutex_);
    memset(icp_covariance_, 0, sizeof(icp_covariance_));
    if (params_.compute_icp_covariance && params_.icp_covariance_method == 1 && meas.have_information) {
      memcpy(icp_covariance_, meas.covariance, sizeof(meas.covariance));
      condition_number_ = meas.condition_number;
    }
  }
  is_healthy_ = true;
  return true;
}

bool PointCloudLocalization::ComputeAp(const PointCloudF& query_cloud, const PointCloudF& reference_cloud,
                                       const std::vector<size_t>& corr, double Ap[36]) {
  // normalizePCloud + ComputeAp_ForPoint2PlaneICP (utils.cc:106-128, PointCloudLocalization.cc:723-750) on the GPU
  lh_cloud *q = nullptr, *r = nullptr;
  lh_cloud_view vq = ViewOf(query_cloud), vr = ViewOf(reference_cloud);
  if (lh_cloud_create(ctx_, &vq, &q) != LH_OK) return false;
  if (lh_cloud_create(ctx_, &vr, &r) != LH_OK) { lh_cloud_destroy(q); return false; }
  std::vector<int64_t> c64(corr.begin(), corr.end());
  lh_status st = lh_p2plane_information(ctx_, q, r, c64.data(), Ap);
  lh_cloud_destroy(q);
  lh_cloud_destroy(r);
  return st == LH_OK;
}

bool PointCloudLocalization::ComputePoint2PlaneICPCovariance(const PointCloudF& query_cloud, const PointCloudF& reference_cloud,
                                                             const std::vector<size_t>& correspondences, const float*,
                                                             double covariance[36]) {  // :469-541
  double Ap[36];
  if (!ComputeAp(query_cloud, reference_cloud, correspondences, Ap)) return false;
  return lh_icp_covariance(Ap, params_.icp_max_covariance, covariance, &condition_number_) == LH_OK;
}

void PointCloudLocalization::ComputeIcpObservability(const PointCloudF& query_cloud, const PointCloudF& reference_cloud,
                                                     const std::vector<size_t>& correspondences, const float*, double eigenvectors[36],
                                                     double eigenvalues[6], double A[36]) {  // :439-467
  double Ap[36];
  if (!ComputeAp(query_cloud, reference_cloud, correspondences, Ap)) return;
  memcpy(A, Ap, sizeof(Ap));
  EigenDecomp6x6(Ap, eigenvectors, eigenvalues);
}

// doEigenDecomp6x6 (utils.cc:130-142): cyclic Jacobi, ascending eigenvalues, eigenvectors in columns
void PointCloudLocalization::EigenDecomp6x6(const double Ap[36], double eigenvectors[36], double eigenvalues[6]) {
  const int n = 6;
  double M[36], V[36];
  memcpy(M, Ap, sizeof(M));
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) off += M[i * n + j] * M[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = M[p * n + q];
        if (std::fabs(apq) < 1e-300) continue;
        double theta = (M[q * n + q] - M[p * n + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) { double a = M[k * n + p], b = M[k * n + q]; M[k * n + p] = c * a - s * b; M[k * n + q] = s * a + c * b; }
        for (int k = 0; k < n; k++) { double a = M[p * n + k], b = M[q * n + k]; M[p * n + k] = c * a - s * b; M[q * n + k] = s * a + c * b; }
        for (int k = 0; k < n; k++) { double a = V[k * n + p], b = V[k * n + q]; V[k * n + p] = c * a - s * b; V[k * n + q] = s * a + c * b; }
      }
  }
  int ord[6] = {0, 1, 2, 3, 4, 5};
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (M[ord[j] * n + ord[j]] < M[ord[i] * n + ord[i]]) std::swap(ord[i], ord[j]);
  for (int i = 0; i < n; i++) {
    eigenvalues[i] = M[ord[i] * n + ord[i]];
    for (int k = 0; k < n; k++) eigenvectors[k * n + i] = V[k * n + ord[i]];
  }
}

void PointCloudLocalization::SetIntegratedEstimate(const gu::Transform3& e) {  // :543-553
  integrated_estimate_ = e;
  incremental_estimate_ = gu::Transform3();
}
void PointCloudLocalization::GetLatestDeltaCovariance(double out[36]) {  // :771-774
  std::lock_guard<std::mutex> lock(icp_covariance_mutex_);
  memcpy(out, icp_covariance_, sizeof(icp_covariance_));
}
void PointCloudLocalization::SetFlatGroundAssumptionValue(const bool& value) {  // :429-437
  b_is_flat_ground_assumption_ = value;
  if (value) integrated_estimate_.rotation = gu::Rot3(0, 0, integrated_estimate_.rotation.Yaw());
}
PointCloudLocalization::Diagnostics PointCloudLocalization::GetDiagnostics() const {  // :752-769
  return is_healthy_ ? Diagnostics{0, "Healthy"} : Diagnostics{2, "Non healthy - Null-pointer error."};
}

}  // namespace locus_hip
