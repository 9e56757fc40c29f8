// MultithreadedGicpHip.hpp -- the registration object the wrappers hold as `icp_`.
// Same public surface as pcl::MultithreadedGeneralizedIterativeClosestPoint<PointF,PointF> (gicp.h:134-298 + the
// pcl::Registration methods LOCUS calls: setInputSource/Target, align, getFinalTransformation, hasConverged,
// getFitnessScore), implemented on the C ABI.  In a PCL environment the adapter in INTEGRATION.md derives from
// pcl::Registration instead and forwards to the same calls.
#pragma once
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "RegistrationHip.hpp"

namespace locus_hip {

class MultithreadedGicpHip : public RegistrationHip {
public:
  typedef std::shared_ptr<MultithreadedGicpHip> Ptr;

  explicit MultithreadedGicpHip(lh_ctx* ctx) : ctx_(ctx) {
    lh_default_gicp_params(&p_);  // gicp.h:111-132
    if (lh_gicp_create(ctx_, &p_, &g_) != LH_OK) throw std::runtime_error("lh_gicp_create failed (no HIP device?)");
    for (int i = 0; i < 16; i++) final_[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  }
  ~MultithreadedGicpHip() override { lh_gicp_destroy(g_); }
  MultithreadedGicpHip(const MultithreadedGicpHip&) = delete;
  MultithreadedGicpHip& operator=(const MultithreadedGicpHip&) = delete;

  // gicp.h setters (names kept)
  void setNumThreads(int n) override { p_.num_threads = n; push(); }                                  // gicp.h:134-141 (ignored on GPU)
  void enableTimingOutput(bool on) override { p_.enable_timing = on ? 1 : 0; push(); }                // gicp.h:143
  void setRotationEpsilon(double e) { p_.rotation_epsilon = e; push(); }                     // gicp.h:236
  double getRotationEpsilon() const { return p_.rotation_epsilon; }
  void setCorrespondenceRandomness(int k) { p_.k_correspondences = k; push(); }              // gicp.h:250
  int getCorrespondenceRandomness() const { return p_.k_correspondences; }
  void setMaximumOptimizerIterations(int n) { p_.max_inner_iterations = n; push(); }         // gicp.h:263
  int getMaximumOptimizerIterations() const { return p_.max_inner_iterations; }
  void RecomputeTargetCovariance(bool r) { p_.recompute_target_cov = r ? 1 : 0; push(); }    // gicp.h:277
  void RecomputeSourceCovariance(bool r) { p_.recompute_source_cov = r ? 1 : 0; push(); }    // gicp.h:285
  // pcl::Registration setters used by SetupICP (PointCloudOdometry.cc:147-155)
  void setTransformationEpsilon(double e) override { p_.transformation_epsilon = e; push(); }
  double getTransformationEpsilon() const { return p_.transformation_epsilon; }
  void setMaxCorrespondenceDistance(double d) override { p_.corr_dist = d; push(); }
  double getMaxCorrespondenceDistance() const { return p_.corr_dist; }
  void setMaximumIterations(int n) override { p_.max_iterations = n; push(); }
  int getMaximumIterations() const { return p_.max_iterations; }
  void setRANSACIterations(int) override {}            // accepted, unused by this computeTransformation
  void setEuclideanFitnessEpsilon(double) override {}  // set by SetupICP but never consulted
  void setCostMode(int mode) { p_.cost_mode = mode; push(); }

  void setInputSource(const PointCloudF::Ptr& cloud) override {  // gicp.h:162-179
    src_ = cloud;
    lh_cloud_view v = ViewOf(*cloud);
    check(lh_gicp_set_source(g_, &v), "setInputSource");
  }
  void setInputTarget(const PointCloudF::Ptr& cloud) override {  // gicp.h:196-200
    tgt_ = cloud;
    lh_cloud_view v = ViewOf(*cloud);
    check(lh_gicp_set_target(g_, &v), "setInputTarget");
  }
  // odometry fast path: the previous query becomes the reference without leaving the GPU
  // (reference_cloud: the host copy the caller holds of that same data -- what getInputTarget() would return)
  void promoteSourceToTarget(const PointCloudF::Ptr& reference_cloud = PointCloudF::Ptr()) {
    tgt_ = reference_cloud ? reference_cloud : src_;
    check(lh_gicp_promote_source_to_target(g_), "promoteSourceToTarget");
  }

  // pcl::Registration::align(output) / align(output, guess): column-major 4x4 like Eigen::Matrix4f
  void align(PointCloudF& output, const float* guess = nullptr) override {
    output.points = src_->points;  // PCL copies the input (all fields) and overwrites xyz with the aligned positions
    output.stamp = src_->stamp;
    lh_gicp_result r;
    lh_status st = lh_gicp_align(g_, guess, &r, nullptr, output.points.data(), sizeof(PointF), offsetof(PointF, x));
    if (st != LH_OK && st != LH_ETOO_FEW_CORR && st != LH_ESOLVER && st != LH_ENO_NN) check(st, "align");   // (the reference catches / returns on these: gicp.hpp:504-506, 542-547)
    for (int i = 0; i < 16; i++) final_[i] = r.T[i];
    converged_ = r.converged != 0;
    iterations_ = r.iterations;
    last_status_ = r.status;
  }
  const float* getFinalTransformation() const override { return final_; }  // column-major
  bool hasConverged() const override { return converged_; }
  int getNumIterations() const { return iterations_; }
  int getLastStatus() const { return last_status_; }
  double getFitnessScore() override {
    double f = 0;
    check(lh_gicp_fitness(g_, &f), "getFitnessScore");
    return f;
  }
  // getSearchMethodTarget()->nearestKSearch(pt, 1, ...) for a whole cloud (PointCloudLocalization.cc:327-336)
  void nearestTargetIndices(const PointCloudF& q, std::vector<size_t>* out) override {
    std::vector<int32_t> idx(q.size());
    lh_cloud_view v = ViewOf(q);
    check(lh_nn1(g_, &v, idx.data(), nullptr), "nearestTargetIndices");
    out->resize(q.size());
    for (size_t i = 0; i < idx.size(); i++) (*out)[i] = (size_t)idx[i];
  }
  // align + transformPointCloudWithNormals + the 1-NN loop + ComputeAp + the covariance conditioning of MeasurementUpdate
  // (PointCloudLocalization.cc:305-336, 398-421) without a second upload and with ONE wait (include/locus_hip.h)
  bool measurementUpdate(PointCloudF* aligned, std::vector<size_t>* correspondences, bool want_information, double icp_max_covariance,
                         Measurement* m) override {
    lh_measurement r;
    std::vector<int32_t> idx(src_->size());
    if (aligned) {   // PCL copies every field of the input and overwrites the points and normals
      aligned->points = src_->points;
      aligned->stamp = src_->stamp;
    }
    lh_status st = lh_gicp_measurement_update(g_, nullptr, want_information ? 1 : 0, icp_max_covariance, &r, idx.data(), aligned ? aligned->points.data() : nullptr,
                                              sizeof(PointF), offsetof(PointF, x), offsetof(PointF, normal_x));
    if (st != LH_OK && st != LH_ETOO_FEW_CORR && st != LH_ESOLVER && st != LH_ENO_NN) check(st, "measurementUpdate");
    for (int i = 0; i < 16; i++) final_[i] = r.result.T[i];
    converged_ = r.result.converged != 0;
    iterations_ = r.result.iterations;
    last_status_ = r.result.status;
    if (correspondences) {
      correspondences->resize(idx.size());
      for (size_t i = 0; i < idx.size(); i++) (*correspondences)[i] = (size_t)idx[i];
    }
    if (m) {
      memcpy(m->Ap, r.Ap, sizeof(r.Ap));
      memcpy(m->covariance, r.covariance, sizeof(r.covariance));
      m->condition_number = r.condition_number;
      m->have_information = r.have_information != 0;
      m->covariance_ok = r.covariance_ok != 0;
    }
    return true;
  }
  // the device-resident form: source and target are clouds that already live in HBM (the caller keeps ownership), the aligned query stays there
  void setInputSourceCloud(lh_cloud* c) { src_.reset(); check(lh_gicp_set_source_cloud(g_, c), "setInputSourceCloud"); }
  void setInputTargetCloud(lh_cloud* c) { tgt_.reset(); check(lh_gicp_set_target_cloud(g_, c), "setInputTargetCloud"); }
  bool measurementUpdateCloud(lh_cloud** aligned, bool want_information, double icp_max_covariance, Measurement* m) {
    lh_measurement r;
    lh_status st = lh_gicp_measurement_update_cloud(g_, nullptr, want_information ? 1 : 0, icp_max_covariance, &r, nullptr, aligned);
    if (st != LH_OK && st != LH_ETOO_FEW_CORR && st != LH_ESOLVER && st != LH_ENO_NN) check(st, "measurementUpdateCloud");
    for (int i = 0; i < 16; i++) final_[i] = r.result.T[i];
    converged_ = r.result.converged != 0;
    iterations_ = r.result.iterations;
    last_status_ = r.result.status;
    if (m) {
      memcpy(m->Ap, r.Ap, sizeof(r.Ap));
      memcpy(m->covariance, r.covariance, sizeof(r.covariance));
      m->condition_number = r.condition_number;
      m->have_information = r.have_information != 0;
      m->covariance_ok = r.covariance_ok != 0;
    }
    return true;
  }
  lh_gicp* handle() { return g_; }
  lh_ctx* context() { return ctx_; }

private:
  void push() { lh_gicp_set_params(g_, &p_); }
  static void check(lh_status st, const char* what) {
    if (st != LH_OK) throw std::runtime_error(std::string("locus_hip: ") + what + ": " + lh_status_string(st));
  }
  lh_ctx* ctx_;
  lh_gicp* g_ = nullptr;
  lh_gicp_params p_;
  PointCloudF::Ptr src_, tgt_;
  float final_[16];
  bool converged_ = false;
  int iterations_ = 0, last_status_ = 0;
};

}  // namespace locus_hip
