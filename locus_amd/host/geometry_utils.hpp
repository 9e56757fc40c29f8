// geometry_utils.hpp -- the handful of common_nebula_slam/geometry_utils types the hot-path wrappers use
// (Transform3, PoseUpdate, PoseInverse, PoseDelta, Rot3::ToEulerZYX; call sites PointCloudOdometry.cc:99-112,289-309,
// PointCloudLocalization.cc:190-191,211-212,369-382).  That package is NOT vendored in the reference; semantics restated
// from its BLAM lineage: PoseUpdate(a,b) = (a.R*b.t + a.t, a.R*b.R).  Row-major 3x3 doubles.
#pragma once
#include <cmath>

namespace locus_hip {
namespace gu {

struct Vec3 {
  double x = 0, y = 0, z = 0;
  Vec3() {}
  Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  double Norm() const { return std::sqrt(x * x + y * y + z * z); }
};

struct Rot3 {
  double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  Rot3() {}
  Rot3(double r00, double r01, double r02, double r10, double r11, double r12, double r20, double r21, double r22) {
    m[0] = r00; m[1] = r01; m[2] = r02; m[3] = r10; m[4] = r11; m[5] = r12; m[6] = r20; m[7] = r21; m[8] = r22;
  }
  Rot3(double roll, double pitch, double yaw) {  // ZYX Euler
    double cr = std::cos(roll), sr = std::sin(roll), cp = std::cos(pitch), sp = std::sin(pitch), cy = std::cos(yaw), sy = std::sin(yaw);
    m[0] = cy * cp; m[1] = cy * sp * sr - sy * cr; m[2] = cy * sp * cr + sy * sr;
    m[3] = sy * cp; m[4] = sy * sp * sr + cy * cr; m[5] = sy * sp * cr - cy * sr;
    m[6] = -sp;     m[7] = cp * sr;                m[8] = cp * cr;
  }
  double operator()(int r, int c) const { return m[r * 3 + c]; }
  Rot3 operator*(const Rot3& o) const {
    Rot3 r;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r.m[i * 3 + j] = m[i * 3] * o.m[j] + m[i * 3 + 1] * o.m[3 + j] + m[i * 3 + 2] * o.m[6 + j];
    return r;
  }
  Vec3 operator*(const Vec3& v) const {
    return Vec3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z);
  }
  Rot3 Trans() const { return Rot3(m[0], m[3], m[6], m[1], m[4], m[7], m[2], m[5], m[8]); }
  double Roll() const { return ToEulerZYX().x; }
  double Pitch() const { return ToEulerZYX().y; }
  double Yaw() const { return ToEulerZYX().z; }
  Vec3 ToEulerZYX() const {  // (roll, pitch, yaw)
    double theta = -std::asin(m[6]);
    double c = std::cos(theta);
    if (std::fabs(c) < 1e-6) return Vec3(0.0, theta, 0.0);
    return Vec3(std::atan2(m[7] / c, m[8] / c), theta, std::atan2(m[3] / c, m[0] / c));
  }
};

struct Transform3 {
  Vec3 translation;
  Rot3 rotation;
};

inline Transform3 PoseUpdate(const Transform3& a, const Transform3& b) {
  Transform3 o;
  Vec3 rt = a.rotation * b.translation;
  o.translation = Vec3(rt.x + a.translation.x, rt.y + a.translation.y, rt.z + a.translation.z);
  o.rotation = a.rotation * b.rotation;
  return o;
}
inline Transform3 PoseInverse(const Transform3& a) {
  Transform3 o;
  o.rotation = a.rotation.Trans();
  Vec3 t = o.rotation * a.translation;
  o.translation = Vec3(-t.x, -t.y, -t.z);
  return o;
}
inline Transform3 PoseDelta(const Transform3& a, const Transform3& b) { return PoseUpdate(PoseInverse(a), b); }

}  // namespace gu
}  // namespace locus_hip
