// point_types.hpp -- memory-layout twins of the PCL types the reference passes around, so the wrappers below compile
// without PCL and a real pcl::PointCloud<pcl::PointXYZINormal>::points array can be handed to the C ABI unchanged.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#include "../../include/locus_hip.h"

namespace locus_hip {

// pcl::PointXYZINormal = frontend_utils PointF (48 bytes: xyz1 | normal xyz0 | intensity curvature pad pad)
struct alignas(16) PointF {
  float x = 0, y = 0, z = 0, data3 = 1.0f;
  float normal_x = 0, normal_y = 0, normal_z = 0, data_n3 = 0;
  float intensity = 0, curvature = 0, pad0 = 0, pad1 = 0;
};
static_assert(sizeof(PointF) == 48, "PointF must match pcl::PointXYZINormal");

// pcl::PointXYZI (32 bytes)
struct alignas(16) PointXYZI {
  float x = 0, y = 0, z = 0, data3 = 1.0f;
  float intensity = 0, pad0 = 0, pad1 = 0, pad2 = 0;
};
static_assert(sizeof(PointXYZI) == 32, "PointXYZI must match pcl::PointXYZI");

struct PointCloudF {
  typedef std::shared_ptr<PointCloudF> Ptr;
  std::vector<PointF> points;
  uint64_t stamp = 0;  // header.stamp (microseconds, PCL convention)
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); }
};

inline lh_cloud_view ViewOf(const PointCloudF& c) {
  lh_cloud_view v;
  v.base = c.points.data();
  v.count = (uint32_t)c.points.size();
  v.stride = sizeof(PointF);
  v.off_xyz = offsetof(PointF, x);
  v.off_normal = offsetof(PointF, normal_x);
  v.off_intensity = offsetof(PointF, intensity);
  v.off_curvature = offsetof(PointF, curvature);
  return v;
}

}  // namespace locus_hip
