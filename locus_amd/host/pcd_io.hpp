// pcd_io.hpp -- PCD v0.7 reader / writer for PointCloudF (pcl::PointXYZINormal), the format LOCUS reads its ground-truth map
// from (pcl::PCDReader::read, Locus.cc:749-751) and the reference's GICP test loads its fixtures from
// (pcl::io::loadPCDFile<PointF>, test_same_output_different_num_threads.cpp:16-21).
// Supported: FIELDS any subset / order of x y z intensity normal_x normal_y normal_z curvature (other fields are skipped),
// SIZE 4 / TYPE F / COUNT 1 for the fields that are read, DATA ascii | binary.  Like pcl::PCDReader, fields the file does not
// have keep the point type's defaults.  binary_compressed is not supported (returns false).
#pragma once
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "point_types.hpp"

namespace locus_hip {

inline bool ReadPCD(const std::string& path, PointCloudF* out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f || !out) { if (f) fclose(f); return false; }
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  std::vector<char> types;
  long points = -1, width = -1, height = 1;
  int data_mode = -1;  // 0 ascii, 1 binary
  char line[4096];
  while (fgets(line, sizeof(line), f)) {
    if (line[0] == '#') continue;
    std::istringstream is(line);
    std::string key;
    is >> key;
    if (key == "FIELDS") { std::string s; while (is >> s) fields.push_back(s); }
    else if (key == "SIZE") { int v; while (is >> v) sizes.push_back(v); }
    else if (key == "TYPE") { char c; while (is >> c) types.push_back(c); }
    else if (key == "COUNT") { int v; while (is >> v) counts.push_back(v); }
    else if (key == "WIDTH") is >> width;
    else if (key == "HEIGHT") is >> height;
    else if (key == "POINTS") is >> points;
    else if (key == "DATA") {
      std::string m;
      is >> m;
      data_mode = m == "ascii" ? 0 : (m == "binary" ? 1 : -1);
      break;
    }
  }
  if (points < 0 && width >= 0) points = width * height;
  if (counts.empty()) counts.assign(fields.size(), 1);
  if (data_mode < 0 || points < 0 || fields.empty() || sizes.size() != fields.size() || types.size() != fields.size() ||
      counts.size() != fields.size()) { fclose(f); return false; }
  // where each file field lands in PointF (-1: skipped)
  static const char* names[8] = {"x", "y", "z", "intensity", "normal_x", "normal_y", "normal_z", "curvature"};
  static const size_t offs[8] = {offsetof(PointF, x), offsetof(PointF, y), offsetof(PointF, z), offsetof(PointF, intensity),
                                 offsetof(PointF, normal_x), offsetof(PointF, normal_y), offsetof(PointF, normal_z), offsetof(PointF, curvature)};
  std::vector<long> dst(fields.size(), -1), src_off(fields.size(), 0);
  long step = 0;
  for (size_t k = 0; k < fields.size(); k++) {
    src_off[k] = step;
    step += (long)sizes[k] * counts[k];
    for (int j = 0; j < 8; j++)
      if (fields[k] == names[j]) {
        if (sizes[k] != 4 || types[k] != 'F' || counts[k] != 1) { fclose(f); return false; }
        dst[k] = (long)offs[j];
      }
  }
  out->points.assign((size_t)points, PointF());
  bool ok = true;
  if (data_mode == 1) {
    std::vector<char> buf((size_t)step);
    for (long i = 0; i < points && ok; i++) {
      if (fread(buf.data(), 1, (size_t)step, f) != (size_t)step) { ok = false; break; }
      for (size_t k = 0; k < fields.size(); k++)
        if (dst[k] >= 0) memcpy(reinterpret_cast<char*>(&out->points[(size_t)i]) + dst[k], buf.data() + src_off[k], 4);
    }
  } else {
    for (long i = 0; i < points && ok; i++) {
      if (!fgets(line, sizeof(line), f)) { ok = false; break; }
      std::istringstream is(line);
      for (size_t k = 0; k < fields.size() && ok; k++)
        for (int c = 0; c < counts[k]; c++) {
          std::string tok;
          if (!(is >> tok)) { ok = false; break; }
          if (dst[k] >= 0) {
            float v = strtof(tok.c_str(), nullptr);  // handles "nan"
            memcpy(reinterpret_cast<char*>(&out->points[(size_t)i]) + dst[k], &v, 4);
          }
        }
    }
  }
  fclose(f);
  if (!ok) out->points.clear();
  return ok;
}

// pcl::io::savePCDFileBinary layout for PointXYZINormal: x y z normal_x normal_y normal_z intensity curvature
inline bool WritePCDBinary(const std::string& path, const PointCloudF& cloud) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z intensity curvature\n"
             "SIZE 4 4 4 4 4 4 4 4\nTYPE F F F F F F F F\nCOUNT 1 1 1 1 1 1 1 1\nWIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n",
          cloud.size(), cloud.size());
  bool ok = true;
  for (const PointF& p : cloud.points) {
    float rec[8] = {p.x, p.y, p.z, p.normal_x, p.normal_y, p.normal_z, p.intensity, p.curvature};
    if (fwrite(rec, sizeof(float), 8, f) != 8) { ok = false; break; }
  }
  return fclose(f) == 0 && ok;
}

}  // namespace locus_hip
