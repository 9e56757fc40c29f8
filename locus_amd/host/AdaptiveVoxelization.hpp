// AdaptiveVoxelization.hpp -- the input-voxelisation feedback controller of the LOCUS node (Locus::ApplyAdaptiveInputVoxelization,
// Locus.cc:780-810): after every scan the voxel leaf is scaled by (points that arrived) / (points the callback can afford),
// clamped to [0.01, 5.0]; the CustomVoxelGrid nodelet is told (change_leaf_size topic, custom_voxel_grid.cc:89-96) when the
// value moved by more than 0.01 or every 20th scan.  Plain arithmetic, no device work: it closes the loop around lh_voxel_grid.
#pragma once
#include <cmath>
#include <cstddef>

namespace locus_hip {

class AdaptiveVoxelization {
public:
  AdaptiveVoxelization(double initial_leaf, size_t points_to_process_in_callback)
      : value_(initial_leaf), points_to_process_(points_to_process_in_callback) {}

  // msg_points = size of the (already voxelised) scan the callback received.  Returns true when the filter must be
  // reconfigured; *leaf_out (nullable) always receives the computed value (what Locus publishes on dchange_voxel).
  bool Update(size_t msg_points, double* leaf_out) {
    bool change = false;
    double dchange_voxel = value_ * (static_cast<double>(msg_points) / static_cast<double>(points_to_process_));  // Locus.cc:782-784
    if (dchange_voxel < 0.01) dchange_voxel = 0.01;                                                                // :785-788
    if (dchange_voxel > 5.0) dchange_voxel = 5.0;
    if (std::abs(value_ - dchange_voxel) > 0.01 || counter_voxel_ % 20 == 0) {                                      // :790-797
      value_ = dchange_voxel;
      change = true;
      counter_voxel_ = 0;
    }
    counter_voxel_++;                                                                                               // :798
    if (leaf_out) *leaf_out = dchange_voxel;
    return change;
  }
  double leaf_size() const { return value_; }

private:
  double value_;             // double_param.value: the leaf size last sent to the filter
  size_t points_to_process_; // points_to_process_in_callback_
  int counter_voxel_ = 0;
};

}  // namespace locus_hip
