// PointCloudOdometry.hpp -- scan-to-scan wrapper, same class surface as the reference
// (point_cloud_odometry/include/point_cloud_odometry/PointCloudOdometry.h:62-94) without ROS/PCL: Initialize() takes the
// rosparam values as a struct (keys of point_cloud_odometry/config/parameters.yaml), tf::Transform becomes a 4x4 matrix.
#pragma once
#include <string>

#include "MultithreadedGicpHip.hpp"
#include "NdtHip.hpp"
#include "geometry_utils.hpp"

namespace locus_hip {

class PointCloudOdometry {
public:
  struct Config {                      // rosparam key (PointCloudOdometry.cc:72-96)
    std::string registration_method = "gicp";  // icp/registration_method   ("gicp" | "ndt", registration_settings.h:13-20)
    double icp_tf_epsilon = 0.001;     // icp/tf_epsilon
    double icp_corr_dist = 1.0;        // icp/corr_dist
    unsigned int icp_iterations = 20;  // icp/iterations
    bool transform_thresholding = true;// icp/transform_thresholding
    double max_translation = 1.0;      // icp/max_translation
    double max_rotation = 1.0;         // icp/max_rotation
    int num_threads = 1;               // icp/num_threads
    bool enable_timing_output = false; // icp/enable_timing_output
    bool recompute_covariances = false;// icp/recompute_covariances
    bool b_is_flat_ground_assumption = false;
    gu::Transform3 initial_pose;       // fiducial_calibration/*
  };

  explicit PointCloudOdometry(lh_ctx* ctx);
  ~PointCloudOdometry();

  bool Initialize(const Config& cfg);

  bool SetLidar(const PointCloudF& points);
  bool SetImuDelta(const double imu_delta_rowmajor[9]);
  bool SetOdometryDelta(const double odometry_delta_rowmajor[16]);

  bool UpdateEstimate();

  const gu::Transform3& GetIncrementalEstimate() const { return incremental_estimate_; }
  const gu::Transform3& GetIntegratedEstimate() const { return integrated_estimate_; }
  gu::Transform3 incremental_estimate_;
  gu::Transform3 integrated_estimate_;

  bool GetLastPointCloud(PointCloudF::Ptr& out) const;
  // the source cloud the registration was given: the query moved by the sensor prior (query_trans_, PointCloudOdometry.cc:255-262)
  PointCloudF::Ptr GetQueryTransformed() const { return query_trans_; }

  PointCloudF icpAlignedPointsOdometry_;  // aligned point cloud returned by ICP

  void EnableImuIntegration();
  void EnableOdometryIntegration();
  void DisableSensorIntegration();
  void SetFlatGroundAssumptionValue(const bool& value);
  // (not in the reference) false: upload the reference cloud every update instead of promoting the previous query on the GPU; results are identical
  void EnableDevicePromotion(bool on) { device_promotion_ = on; }

  // diagnostic_msgs::DiagnosticStatus analogue: level 0 = OK, 2 = ERROR (PointCloudOdometry.cc:367-380)
  struct Diagnostics { int level; std::string message; };
  Diagnostics GetDiagnostics() const;

  RegistrationHip::Ptr icp_;  // the reference keeps this private; tests reach it through a friend accessor

private:
  bool SetupICP();
  bool UpdateICP();

  lh_ctx* ctx_;
  std::string name_ = "PointCloudOdometry";
  bool initialized_ = false, is_healthy_ = false;
  PointCloudF points_;
  PointCloudF::Ptr query_, reference_, query_trans_;
  bool transform_thresholding_ = true;
  double max_translation_ = 1.0, max_rotation_ = 1.0;
  Config params_;
  bool b_use_imu_integration_ = false, b_use_odometry_integration_ = false;
  double imu_delta_[9];
  double odometry_delta_[16];
  bool b_is_flat_ground_assumption_ = false;
  bool device_promotion_ = true;
  bool device_source_is_last_query_ = false;   // the registration object's device-resident source is *query_ as it was set (UpdateICP)
};

}  // namespace locus_hip
