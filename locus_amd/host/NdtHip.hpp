// NdtHip.hpp -- registration_method "ndt": the surface of pclomp::NormalDistributionsTransform<PointF, PointF>
// (multithreaded_gicp/include/multithreaded_ndt/ndt_omp.h:116-246) on the C ABI's lh_ndt; configured by the NDT branch of
// SetupICP (PointCloudOdometry.cc:182-196, PointCloudLocalization.cc:268-282).
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>

#include "RegistrationHip.hpp"

namespace locus_hip {

class NdtHip : public RegistrationHip {
public:
  explicit NdtHip(lh_ctx* ctx) : ctx_(ctx) {
    lh_default_ndt_params(&p_);  // ndt_omp_impl.hpp:50-52, 93-94
    if (lh_ndt_create(ctx_, &p_, &g_) != LH_OK) throw std::runtime_error("lh_ndt_create failed (no HIP device?)");
    for (int i = 0; i < 16; i++) final_[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  }
  ~NdtHip() override {
    lh_ndt_destroy(g_);
    lh_cloud_destroy(src_dev_);
    lh_cloud_destroy(tgt_dev_);
  }
  NdtHip(const NdtHip&) = delete;
  NdtHip& operator=(const NdtHip&) = delete;

  void setResolution(float r) { p_.resolution = r; push(); }                 // ndt_omp.h:124
  float getResolution() const { return p_.resolution; }
  void setStepSize(double s) { p_.step_size = s; push(); }                   // :150
  double getStepSize() const { return p_.step_size; }
  void setOulierRatio(double r) { p_.outlier_ratio = r; push(); }            // :164 (the reference's spelling)
  double getOulierRatio() const { return p_.outlier_ratio; }
  void setTransformationEpsilon(double e) override { p_.transformation_epsilon = e; push(); }
  void setMaxCorrespondenceDistance(double) override {}                      // set by SetupICP, never consulted by NDT
  void setMaximumIterations(int n) override { p_.max_iterations = n; push(); }
  double getTransformationProbability() const { return trans_probability_; } // :176
  int getFinalNumIteration() const { return iterations_; }                   // :183

  void setInputSource(const PointCloudF::Ptr& cloud) override { upload(cloud, &src_, &src_dev_); check(lh_ndt_set_source_cloud(g_, src_dev_), "setInputSource"); }
  void setInputTarget(const PointCloudF::Ptr& cloud) override { upload(cloud, &tgt_, &tgt_dev_); check(lh_ndt_set_target_cloud(g_, tgt_dev_), "setInputTarget"); }
  void align(PointCloudF& output, const float* guess = nullptr) override {
    output.points = src_->points;
    output.stamp = src_->stamp;
    lh_gicp_result r;
    check(lh_ndt_align(g_, guess, &r, output.points.data(), sizeof(PointF), offsetof(PointF, x)), "align");
    for (int i = 0; i < 16; i++) final_[i] = r.T[i];
    converged_ = r.converged != 0;
    iterations_ = r.iterations;
    trans_probability_ = r.fitness;
  }
  const float* getFinalTransformation() const override { return final_; }
  bool hasConverged() const override { return converged_; }
  // pcl::Registration::getFitnessScore(): mean squared distance of the aligned source to its nearest target point
  double getFitnessScore() override {
    lh_cloud* moved = nullptr;
    check(lh_cloud_transform(src_dev_, final_, 0, &moved), "getFitnessScore");
    std::vector<float> d2(lh_cloud_size(moved));
    lh_status st = lh_nn1_cloud(tgt_dev_, moved, nullptr, d2.data());
    lh_cloud_destroy(moved);
    check(st, "getFitnessScore");
    double s = 0;
    for (float v : d2) s += (double)v;
    return d2.empty() ? 0.0 : s / (double)d2.size();
  }
  void nearestTargetIndices(const PointCloudF& q, std::vector<size_t>* out) override {
    lh_cloud_view v = ViewOf(q);
    lh_cloud* qc = nullptr;
    check(lh_cloud_create(ctx_, &v, &qc), "nearestTargetIndices");
    std::vector<int32_t> idx(q.size());
    lh_status st = lh_nn1_cloud(tgt_dev_, qc, idx.data(), nullptr);
    lh_cloud_destroy(qc);
    check(st, "nearestTargetIndices");
    out->assign(idx.begin(), idx.end());
  }

private:
  void push() { lh_ndt_set_params(g_, &p_); }
  void upload(const PointCloudF::Ptr& cloud, PointCloudF::Ptr* keep, lh_cloud** dev) {
    *keep = cloud;
    lh_cloud_destroy(*dev);
    *dev = nullptr;
    lh_cloud_view v = ViewOf(*cloud);
    check(lh_cloud_create(ctx_, &v, dev), "upload");
  }
  static void check(lh_status st, const char* what) {
    if (st != LH_OK) throw std::runtime_error(std::string("locus_hip: ") + what + ": " + lh_status_string(st));
  }
  lh_ctx* ctx_;
  lh_ndt* g_ = nullptr;
  lh_ndt_params p_;
  PointCloudF::Ptr src_, tgt_;
  lh_cloud *src_dev_ = nullptr, *tgt_dev_ = nullptr;
  float final_[16];
  bool converged_ = false;
  int iterations_ = 0;
  double trans_probability_ = 0.0;
};

}  // namespace locus_hip
