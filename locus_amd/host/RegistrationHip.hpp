// RegistrationHip.hpp -- what PointCloudOdometry / PointCloudLocalization hold as `icp_`: the slice of
// pcl::Registration<PointF, PointF> they call (PointCloudOdometry.h:154, PointCloudLocalization.h:228), so that the
// registration_method string selects the object behind it exactly as in the reference (registration_settings.h:3-20):
// "gicp" -> MultithreadedGicpHip, "ndt" -> NdtHip.
#pragma once
#include <memory>
#include <vector>

#include "point_types.hpp"

namespace locus_hip {

class RegistrationHip {
public:
  typedef std::shared_ptr<RegistrationHip> Ptr;
  virtual ~RegistrationHip() {}
  // pcl::Registration setters used by both SetupICP functions
  virtual void setTransformationEpsilon(double e) = 0;
  virtual void setMaxCorrespondenceDistance(double d) = 0;
  virtual void setMaximumIterations(int n) = 0;
  virtual void setRANSACIterations(int) {}
  virtual void setEuclideanFitnessEpsilon(double) {}
  virtual void setNumThreads(int) {}
  virtual void enableTimingOutput(bool) {}
  // the run-time surface
  virtual void setInputSource(const PointCloudF::Ptr& cloud) = 0;
  virtual void setInputTarget(const PointCloudF::Ptr& cloud) = 0;
  virtual void align(PointCloudF& output, const float* guess = nullptr) = 0;
  virtual const float* getFinalTransformation() const = 0;  // column-major 4x4 (Eigen::Matrix4f memory order)
  float T(int r, int c) const { return getFinalTransformation()[c * 4 + r]; }
  virtual bool hasConverged() const = 0;
  virtual double getFitnessScore() = 0;
  // getSearchMethodTarget()->nearestKSearch(pt, 1, ...) for a whole cloud (PointCloudLocalization.cc:327-336)
  virtual void nearestTargetIndices(const PointCloudF& q, std::vector<size_t>* out) = 0;
  // PointCloudLocalization::MeasurementUpdate's device work in one call on the clouds setInputSource / setInputTarget uploaded
  // (lh_gicp_measurement_update: align, aligned query with normals, correspondences, Ap, covariance).  false: this registration
  // object has no such entry point (NDT) and the caller stitches the steps itself.
  struct Measurement { double Ap[36]; double covariance[36]; double condition_number; bool have_information, covariance_ok; };
  virtual bool measurementUpdate(PointCloudF*, std::vector<size_t>*, bool, double, Measurement*) { return false; }
};

}  // namespace locus_hip
