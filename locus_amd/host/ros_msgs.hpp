// ros_msgs.hpp -- layout twin of sensor_msgs/PointCloud2 and its mapping onto the C ABI's lh_cloud_view, i.e. what
// pcl::fromROSMsg / pcl::toROSMsg do at the filter nodelets' boundary (normal_computation.cc:30,58; custom_voxel_grid.cc:81).
// A PointCloud2 is already "base pointer + point_step + per-field byte offsets": the conversion is a table lookup, the point
// data itself is handed to lh_cloud_create / lh_voxel_grid / lh_normals_* without a host-side repack.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "point_types.hpp"

namespace locus_hip {

struct PointField {  // sensor_msgs/PointField
  enum : uint8_t { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
  std::string name;
  uint32_t offset = 0;
  uint8_t datatype = FLOAT32;
  uint32_t count = 1;
};
struct PointCloud2 {  // sensor_msgs/PointCloud2 (header reduced to the stamp)
  uint64_t stamp = 0;
  uint32_t height = 1, width = 0;
  std::vector<PointField> fields;
  bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0;
  std::vector<uint8_t> data;
  bool is_dense = true;
};

// the view pcl::fromROSMsg would decode: x/y/z must be consecutive FLOAT32 (any point_step); normal_x/y/z consecutive FLOAT32
// if present; intensity / curvature FLOAT32 if present.  false = a layout the C ABI cannot read in place.
inline bool ViewFromPointCloud2(const PointCloud2& m, lh_cloud_view* v) {
  if (!v || m.is_bigendian || m.point_step < 12) return false;
  const uint64_t n = (uint64_t)m.width * m.height;
  if (m.data.size() < n * m.point_step) return false;
  auto find = [&](const char* name) -> const PointField* {
    for (const PointField& f : m.fields)
      if (f.name == name) return &f;
    return nullptr;
  };
  auto f32 = [](const PointField* f) { return f && f->datatype == PointField::FLOAT32 && f->count == 1; };
  const PointField *x = find("x"), *y = find("y"), *z = find("z");
  if (!f32(x) || !f32(y) || !f32(z) || y->offset != x->offset + 4 || z->offset != x->offset + 8 || z->offset + 4 > m.point_step) return false;
  v->base = m.data.data();
  v->count = (uint32_t)n;
  v->stride = m.point_step;
  v->off_xyz = x->offset;
  v->off_normal = UINT32_MAX;
  v->off_intensity = UINT32_MAX;
  v->off_curvature = UINT32_MAX;
  const PointField *nx = find("normal_x"), *ny = find("normal_y"), *nz = find("normal_z");
  if (nx || ny || nz) {
    if (!f32(nx) || !f32(ny) || !f32(nz) || ny->offset != nx->offset + 4 || nz->offset != nx->offset + 8) return false;
    v->off_normal = nx->offset;
  }
  if (const PointField* i = find("intensity")) { if (!f32(i)) return false; v->off_intensity = i->offset; }
  if (const PointField* c = find("curvature")) { if (!f32(c)) return false; v->off_curvature = c->offset; }
  return true;
}

// the message pcl::toROSMsg builds for a PointCloud<PointXYZINormal> (point_step 48: the PCL struct incl. its padding) or
// PointCloud<PointXYZI> (point_step 32); data sized for n points, ready to be filled by lh_cloud_download with the same offsets
inline void LayoutPointCloud2(PointCloud2* m, uint32_t n, bool with_normals) {
  m->height = 1; m->width = n; m->is_bigendian = false; m->is_dense = true;
  m->fields.clear();
  auto add = [&](const char* name, uint32_t off) { PointField f; f.name = name; f.offset = off; m->fields.push_back(f); };
  add("x", 0); add("y", 4); add("z", 8);
  if (with_normals) {
    add("normal_x", 16); add("normal_y", 20); add("normal_z", 24); add("intensity", 32); add("curvature", 36);
    m->point_step = sizeof(PointF);
  } else {
    add("intensity", 16);
    m->point_step = sizeof(PointXYZI);
  }
  m->row_step = m->point_step * n;
  m->data.assign((size_t)m->row_step, 0);
}

}  // namespace locus_hip
