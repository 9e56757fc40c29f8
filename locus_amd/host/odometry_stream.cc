// odometry_stream.cc -- LOCUS's real operating point, end to end, through the drop-in mirror: a stream of filtered scans
// (~3 000 points after the adaptive voxel grid, lo_settings.yaml:84-85) goes through PointCloudOdometry::SetLidar /
// UpdateEstimate one scan at a time (lidar queue depth 1, Locus.cc:64-68, 451-453) with the shipped parameters
// (point_cloud_odometry/config/parameters.yaml: corr_dist 1.0, tf_epsilon 1e-3, 20 iterations, covariances from normals).
// Every scan arrives as a HOST PointCloudF (what the ROS callback holds) and the aligned cloud comes back to the host: the
// time per update includes the host cloud copies of UpdateEstimate, the PCIe transfers and every synchronisation.
//
//   odometry_stream <scans.bin> [warmup=3]      -> one JSON object on stdout
// scans.bin: int32 n_scans, then per scan int32 n_points + n_points * 48-byte PointF (pcl::PointXYZINormal layout); written by
// bench.py / tests from synthetic scans.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "PointCloudOdometry.hpp"

using namespace locus_hip;

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: odometry_stream scans.bin [warmup]\n"); return 2; }
  const int warm = argc > 2 ? atoi(argv[2]) : 3;
  (void)lh_runtime_init(0);   // what a nodelet manager does first: the hardware-queue setting, before the process's first HIP call
  FILE* f = fopen(argv[1], "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  int32_t n_scans = 0;
  if (fread(&n_scans, 4, 1, f) != 1 || n_scans < 2) return 2;
  std::vector<PointCloudF> scans(n_scans);
  double pts = 0;
  for (auto& s : scans) {
    int32_t n = 0;
    if (fread(&n, 4, 1, f) != 1 || n <= 0) return 2;
    s.points.resize(n);
    if (fread(s.points.data(), sizeof(PointF), n, f) != (size_t)n) return 2;
    pts += n;
  }
  fclose(f);
  lh_ctx* ctx = nullptr;
  if (lh_create(&ctx, 0) != LH_OK) { fprintf(stderr, "no HIP device: the host mirror has no CPU fallback\n"); return 3; }
  PointCloudOdometry::Config cfg;   // defaults = point_cloud_odometry/config/parameters.yaml
  std::vector<double> ms;
  int ok = 0;
  for (int pass = 0; pass < 2; pass++) {   // pass 0 warms the library up (buffers, streams); pass 1 is timed from a fresh wrapper
    PointCloudOdometry od(ctx);
    if (!od.Initialize(cfg)) return 4;
    for (int i = 0; i < n_scans; i++) {
      auto t0 = std::chrono::steady_clock::now();
      od.SetLidar(scans[i]);                       // Locus.cc:451
      bool updated = od.UpdateEstimate();          // Locus.cc:453 (false for the very first scan)
      auto t1 = std::chrono::steady_clock::now();
      if (pass == 1 && i >= 1) {
        ok += updated ? 1 : 0;
        if (i > warm) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
      }
      if (pass == 0 && i >= std::min(n_scans - 1, 8)) break;
    }
    if (pass == 1) {
      const gu::Transform3& T = od.GetIntegratedEstimate();
      std::sort(ms.begin(), ms.end());
      double sum = 0;
      for (double v : ms) sum += v;
      printf("{\"scans\": %d, \"points_per_scan_mean\": %.1f, \"updates_ok\": %d, \"timed_updates\": %zu, \"ms_per_update_median\": %.4f, "
             "\"ms_per_update_mean\": %.4f, \"ms_per_update_p90\": %.4f, \"ms_per_update_max\": %.4f, \"promote_source_to_target\": %s, "
             "\"integrated_translation\": [%.6f, %.6f, %.6f]}\n",
             n_scans, pts / n_scans, ok, ms.size(), ms.empty() ? 0.0 : ms[ms.size() / 2], ms.empty() ? 0.0 : sum / ms.size(),
             ms.empty() ? 0.0 : ms[(size_t)(0.9 * (ms.size() - 1))], ms.empty() ? 0.0 : ms.back(),
             getenv("LOCUS_HIP_NO_PROMOTE") && atoi(getenv("LOCUS_HIP_NO_PROMOTE")) ? "false" : "true", T.translation.x, T.translation.y, T.translation.z);
    }
  }
  lh_destroy(ctx);
  return 0;
}
