// CPU check of locus_amd/host/pcd_io.hpp and ros_msgs.hpp: the reference's own PCD fixtures round-trip, and the
// PointCloud2 <-> lh_cloud_view mapping decodes the layouts pcl::toROSMsg produces.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../locus_amd/host/pcd_io.hpp"
#include "../../locus_amd/host/ros_msgs.hpp"

using namespace locus_hip;
static int fails = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main(int argc, char** argv) {
  if (argc < 3) { printf("usage: io_check <fixture.pcd> <tmp.pcd>\n"); return 2; }
  PointCloudF a, b;
  EXPECT(ReadPCD(argv[1], &a));
  double sx = 0, sy = 0, sz = 0, si = 0;
  for (auto& p : a.points) { sx += p.x; sy += p.y; sz += p.z; si += p.intensity; }
  printf("POINTS %zu SUMS %.9g %.9g %.9g %.9g\n", a.size(), sx, sy, sz, si);
  for (size_t i = 0; i < a.size(); i++) { a.points[i].normal_z = 1.0f; a.points[i].curvature = 0.25f * (float)(i % 7); }
  EXPECT(WritePCDBinary(argv[2], a));
  EXPECT(ReadPCD(argv[2], &b));
  EXPECT(b.size() == a.size());
  bool same = b.size() == a.size();
  for (size_t i = 0; same && i < a.size(); i++) same = memcmp(&a.points[i], &b.points[i], sizeof(PointF)) == 0;
  EXPECT(same);
  // ascii variant written by hand: fields in another order, one foreign field, a NaN
  {
    FILE* f = fopen(argv[2], "w");
    fprintf(f, "VERSION 0.7\nFIELDS intensity x y z ring\nSIZE 4 4 4 4 2\nTYPE F F F F U\nCOUNT 1 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA ascii\n"
               "7.5 1 2 3 11\n8.5 nan -2 -3 12\n");
    fclose(f);
    PointCloudF c;
    EXPECT(ReadPCD(argv[2], &c) && c.size() == 2);
    EXPECT(c.points[0].x == 1.f && c.points[0].z == 3.f && c.points[0].intensity == 7.5f && c.points[0].normal_x == 0.f);
    EXPECT(std::isnan(c.points[1].x) && c.points[1].y == -2.f && c.points[1].intensity == 8.5f);
  }
  // PointCloud2 layouts
  PointCloud2 m;
  LayoutPointCloud2(&m, 5, true);
  lh_cloud_view v;
  EXPECT(ViewFromPointCloud2(m, &v));
  EXPECT(v.count == 5 && v.stride == 48 && v.off_xyz == 0 && v.off_normal == 16 && v.off_intensity == 32 && v.off_curvature == 36 && v.base == m.data.data());
  LayoutPointCloud2(&m, 3, false);
  EXPECT(ViewFromPointCloud2(m, &v));
  EXPECT(v.count == 3 && v.stride == 32 && v.off_normal == UINT32_MAX && v.off_intensity == 16 && v.off_curvature == UINT32_MAX);
  m.fields[0].datatype = PointField::FLOAT64;   // a double x cannot be read in place
  EXPECT(!ViewFromPointCloud2(m, &v));
  m.fields[0].datatype = PointField::FLOAT32;
  m.is_bigendian = true;
  EXPECT(!ViewFromPointCloud2(m, &v));
  m.is_bigendian = false;
  m.data.resize(10);                            // truncated blob
  EXPECT(!ViewFromPointCloud2(m, &v));
  printf(fails ? "IO_CHECK_FAILED\n" : "IO_CHECK_OK\n");
  return fails ? 1 : 0;
}
