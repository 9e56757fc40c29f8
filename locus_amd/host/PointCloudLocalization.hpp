// PointCloudLocalization.hpp -- scan-to-map wrapper, same class surface as the reference
// (point_cloud_localization/include/point_cloud_localization/PointCloudLocalization.h:64-148) without ROS/PCL.
#pragma once
#include <mutex>
#include <string>
#include <vector>

#include "MultithreadedGicpHip.hpp"
#include "NdtHip.hpp"
#include "geometry_utils.hpp"

namespace locus_hip {

class PointCloudLocalization {
public:
  struct Config {                          // rosparam key (point_cloud_localization/config/parameters.yaml)
    std::string registration_method = "gicp";
    double tf_epsilon = 1e-5;              // localization/tf_epsilon
    double corr_dist = 0.2;                // localization/corr_dist
    int iterations = 20;                   // localization/iterations
    int num_threads = 1;
    bool enable_timing_output = false;
    bool recompute_covariance_local_map = false, recompute_covariance_scan = false;
    bool transform_thresholding = false;
    double max_translation = 0.5, max_rotation = 0.3;
    bool compute_icp_covariance = true;
    int icp_covariance_method = 1;
    double icp_max_covariance = 0.01;
    bool compute_icp_observability = false;
    bool b_is_flat_ground_assumption = false;
    gu::Transform3 initial_pose;
  };

  explicit PointCloudLocalization(lh_ctx* ctx);
  ~PointCloudLocalization();

  bool Initialize(const Config& cfg);
  bool TransformPointsToFixedFrame(const PointCloudF& points, PointCloudF* points_transformed) const;
  bool TransformPointsToSensorFrame(const PointCloudF& points, PointCloudF* points_transformed) const;
  bool MotionUpdate(const gu::Transform3& incremental_odom);
  bool MeasurementUpdate(const PointCloudF::Ptr& query, const PointCloudF::Ptr& reference, PointCloudF* aligned_query);
  // The same three steps on clouds that live in HBM (lh_cloud: the scan uploaded once, the map's neighbours found there): the MI355X-first form
  // of Locus.cc:474-489.  *out / *aligned_query are new device clouds the caller destroys.  Poses, covariance and observability are the host
  // surface's bit for bit (tests: host_check `device_flow`).
  bool TransformPointsToFixedFrame(const lh_cloud* points, lh_cloud** out) const;
  bool TransformPointsToSensorFrame(const lh_cloud* points, lh_cloud** out) const;
  bool MeasurementUpdate(lh_cloud* query, lh_cloud* reference, lh_cloud** aligned_query);
  bool ComputePoint2PlaneICPCovariance(const PointCloudF& query_cloud, const PointCloudF& reference_cloud,
                                       const std::vector<size_t>& correspondences, const float* T_colmajor, double covariance[36]);
  void ComputeIcpObservability(const PointCloudF& query_cloud, const PointCloudF& reference_cloud,
                               const std::vector<size_t>& correspondences, const float* T_colmajor, double eigenvectors[36],
                               double eigenvalues[6], double A[36]);

  const gu::Transform3& GetIncrementalEstimate() const { return incremental_estimate_; }
  const gu::Transform3& GetIntegratedEstimate() const { return integrated_estimate_; }
  void SetIntegratedEstimate(const gu::Transform3& integrated_estimate);
  gu::Transform3 incremental_estimate_;
  gu::Transform3 integrated_estimate_;

  void GetLatestDeltaCovariance(double out[36]);
  double condition_number() const { return condition_number_; }
  void SetFlatGroundAssumptionValue(const bool& value);
  struct Diagnostics { int level; std::string message; };
  Diagnostics GetDiagnostics() const;

  RegistrationHip::Ptr icp_;

private:
  bool SetupICP();
  void UpdatePoses();
  static void EigenDecomp6x6(const double Ap[36], double eigenvectors[36], double eigenvalues[6]);
  bool ComputeAp(const PointCloudF& query_cloud, const PointCloudF& reference_cloud, const std::vector<size_t>& corr, double Ap[36]);

  lh_ctx* ctx_;
  Config params_;
  bool is_healthy_ = false, b_is_flat_ground_assumption_ = false;
  double icp_covariance_[36];
  double observability_matrix_[36];
  double condition_number_ = 0;
  std::mutex icp_covariance_mutex_;
};

}  // namespace locus_hip
