// PointCloudMapperHip.hpp -- the local-map object LOCUS holds as `mapper_` (IPointCloudMapper: Locus.cc:134, 268-269, 464-465,
// 479-483, 531-538), on the C ABI's device-resident lh_map.  point_cloud_mapper itself is un-vendored in the reference tree
// ("parity unpinned"); the surface and call order follow Locus.cc, the semantics its BLAM lineage:
//   InsertPoints            a point enters the map iff its octree voxel (edge = octree_resolution) is still empty
//   ApproxNearestNeighbors  the nearest map point of every query point (exact here, so never farther than the reference's)
//   Refresh                 box crop of half-extent box_filter_size around the current pose (multi-threaded sliding window)
#pragma once
#include <stdexcept>

#include "geometry_utils.hpp"
#include "point_types.hpp"

namespace locus_hip {

class PointCloudMapperHip {
public:
  explicit PointCloudMapperHip(lh_ctx* ctx) : ctx_(ctx) {}
  ~PointCloudMapperHip() { lh_map_destroy(map_); }
  PointCloudMapperHip(const PointCloudMapperHip&) = delete;
  PointCloudMapperHip& operator=(const PointCloudMapperHip&) = delete;

  bool Initialize(double octree_resolution) {   // IPointCloudMapper::Initialize(n): map/octree_resolution
    resolution_ = octree_resolution;
    Reset();
    return map_ != nullptr;
  }
  void Reset() {
    lh_map_destroy(map_);
    map_ = nullptr;
    if (lh_map_create(ctx_, resolution_, &map_) != LH_OK) map_ = nullptr;
  }
  void SetBoxFilterSize(int box_filter_size) { box_filter_size_ = box_filter_size; }   // Locus.cc:268
  void SetupNumberThreads(int) {}                                                     // Locus.cc:269: CPU knob, ignored
  void UpdateCurrentPose(const gu::Transform3& pose) { current_pose_ = pose; }            // Locus.cc:464, 531
  size_t Size() const { return lh_map_size(map_); }
  lh_cloud* DeviceCloud() { return lh_map_cloud(map_); }   // for lh_gicp_set_target_cloud: scan-to-map without a host round trip

  // Locus.cc:465 -- points are already in the fixed frame; incremental_points (nullable) receives what was actually added
  bool InsertPoints(const PointCloudF& points, PointCloudF* incremental_points) {
    if (!map_ || points.empty()) return false;
    lh_cloud_view v = ViewOf(points);
    lh_cloud* c = nullptr;
    if (lh_cloud_create(ctx_, &v, &c) != LH_OK) return false;
    uint32_t before = lh_map_size(map_), added = 0;
    lh_status st = lh_map_insert(map_, c, &added);
    lh_cloud_destroy(c);
    if (st != LH_OK) return false;
    if (incremental_points) {
      incremental_points->clear();
      if (added > 0) {
        lh_cloud* tail = nullptr;
        if (lh_cloud_slice(lh_map_cloud(map_), before, added, &tail) != LH_OK) return false;
        bool ok = Download(tail, incremental_points);
        lh_cloud_destroy(tail);
        if (!ok) return false;
      }
    }
    return true;
  }

  // Locus.cc:479-480
  bool ApproxNearestNeighbors(const PointCloudF& points, PointCloudF* neighbors) {
    if (!map_ || !neighbors || points.empty() || lh_map_size(map_) == 0) return false;
    lh_cloud_view v = ViewOf(points);
    lh_cloud *q = nullptr, *nb = nullptr;
    if (lh_cloud_create(ctx_, &v, &q) != LH_OK) return false;
    lh_status st = lh_cloud_nearest_neighbors(lh_map_cloud(map_), q, &nb);
    lh_cloud_destroy(q);
    if (st != LH_OK) return false;
    bool ok = Download(nb, neighbors);
    lh_cloud_destroy(nb);
    return ok;
  }

  // the same two calls on clouds that already live in HBM (the device-resident LOCUS flow): no upload, no download
  bool InsertPoints(const lh_cloud* points, uint32_t* added = nullptr) {
    if (!map_ || !points || lh_cloud_size(points) == 0) return false;
    uint32_t a = 0;
    lh_status st = lh_map_insert(map_, points, &a);
    if (added) *added = a;
    return st == LH_OK;
  }
  bool ApproxNearestNeighbors(const lh_cloud* points, lh_cloud** neighbors) {   // *neighbors: a new device cloud the caller destroys
    if (!map_ || !neighbors || !points || lh_cloud_size(points) == 0 || lh_map_size(map_) == 0) return false;
    return lh_cloud_nearest_neighbors(lh_map_cloud(map_), points, neighbors) == LH_OK;
  }

  // Locus.cc:537
  void Refresh(const gu::Transform3& current_pose) {
    if (!map_ || box_filter_size_ <= 0) return;
    float c[3] = {(float)current_pose.translation.x, (float)current_pose.translation.y, (float)current_pose.translation.z};
    (void)lh_map_refresh(map_, c, (float)box_filter_size_);
  }

private:
  static bool Download(const lh_cloud* c, PointCloudF* out) {
    out->points.assign(lh_cloud_size(c), PointF());
    if (out->points.empty()) return true;
    return lh_cloud_download(c, out->points.data(), sizeof(PointF), offsetof(PointF, x), offsetof(PointF, normal_x),
                             offsetof(PointF, intensity), offsetof(PointF, curvature)) == LH_OK;
  }
  lh_ctx* ctx_;
  lh_map* map_ = nullptr;
  double resolution_ = 0.05;
  int box_filter_size_ = 20;   // lo_settings.yaml:58
  gu::Transform3 current_pose_;
};

}  // namespace locus_hip
