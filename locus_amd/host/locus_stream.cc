// locus_stream.cc -- LOCUS's per-scan registration work end to end (Locus.cc:450-520): for every filtered scan
//   odometry_.SetLidar + UpdateEstimate                 scan-to-scan GICP (PointCloudOdometry.cc:237-322)
//   localization_.MotionUpdate                          (PointCloudLocalization.cc:174-179)
//   TransformPointsToFixedFrame -> mapper_->ApproxNearestNeighbors -> TransformPointsToSensorFrame        (Locus.cc:474-486)
//   localization_.MeasurementUpdate                     scan-to-submap GICP + correspondences + Ap + covariance (:291-427)
//   keyframe: TransformPointsToFixedFrame + mapper_->InsertPoints when the pose moved > 1 m / 0.3 rad (Locus.cc:505-520, lo_settings.yaml:8-9)
// through the drop-in mirrors, two ways:
//   host    host PointCloudF in and out of every call, exactly the reference's call sequence (what a ROS node holds)
//   device  the scan is uploaded ONCE per consumer (odometry's setInputSource, localization's query cloud) and everything between the
//           two registrations -- frame transforms, map neighbours, aligned query, map insertion -- stays in HBM
// Both must produce the same poses bit for bit (checked here: the exit code says so).
//   locus_stream <scans.bin> [warmup=3]   -> one JSON object on stdout
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "PointCloudLocalization.hpp"
#include "PointCloudMapperHip.hpp"
#include "PointCloudOdometry.hpp"

using namespace locus_hip;
typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

struct Stage { std::vector<double> odom, neigh, meas, total; };
static double median(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static double vmax(const std::vector<double>& v) { double m = 0; for (double x : v) m = std::max(m, x); return m; }

struct Run { gu::Transform3 pose; double cov[36]; Stage st; int keyframes = 0, updates = 0; size_t map_points = 0; };

static bool keyframe_due(const gu::Transform3& last, const gu::Transform3& cur) {   // Locus.cc:505-512
  const gu::Transform3 d = gu::PoseDelta(last, cur);
  const double tr = d.rotation.m[0] + d.rotation.m[4] + d.rotation.m[8];
  const double ang = std::acos(std::min(1.0, std::max(-1.0, (tr - 1.0) / 2.0)));   // |2 acos(q.w)| of the delta rotation
  return d.translation.Norm() > 1.0 || std::fabs(ang) > 0.3;
}

static Run run(lh_ctx* ctx, const std::vector<PointCloudF>& scans, bool device, int warm, bool timed) {
  Run R;
  PointCloudOdometry od(ctx);
  PointCloudOdometry::Config oc;
  PointCloudLocalization loc(ctx);
  PointCloudLocalization::Config lc;   // point_cloud_localization/config/parameters.yaml defaults
  PointCloudMapperHip mapper(ctx);
  if (!od.Initialize(oc) || !loc.Initialize(lc) || !mapper.Initialize(0.05)) { fprintf(stderr, "initialisation failed\n"); exit(4); }
  gu::Transform3 last_kf;
  bool first = true;
  for (size_t i = 0; i < scans.size(); i++) {
    const auto t0 = Clock::now();
    PointCloudF::Ptr scan(new PointCloudF(scans[i]));
    od.SetLidar(*scan);                                  // Locus.cc:451
    const bool updated = od.UpdateEstimate();            // Locus.cc:453
    const double t_od = ms_since(t0);
    if (first || !updated) {                             // b_add_first_scan_to_key_ (Locus.cc:463-473)
      PointCloudF fixed;
      loc.TransformPointsToFixedFrame(*scan, &fixed);
      mapper.UpdateCurrentPose(loc.GetIntegratedEstimate());
      mapper.InsertPoints(fixed, nullptr);
      last_kf = loc.GetIntegratedEstimate();
      first = false;
      continue;
    }
    loc.MotionUpdate(od.GetIncrementalEstimate());       // Locus.cc:476
    const auto t1 = Clock::now();
    double t_nb = 0, t_mu = 0;
    if (!device) {
      PointCloudF transformed, base;
      PointCloudF::Ptr neighbors(new PointCloudF);
      loc.TransformPointsToFixedFrame(*scan, &transformed);
      if (!mapper.ApproxNearestNeighbors(transformed, neighbors.get())) { fprintf(stderr, "ApproxNearestNeighbors returned false\n"); exit(5); }
      loc.TransformPointsToSensorFrame(*neighbors, neighbors.get());
      t_nb = ms_since(t1);
      const auto t2 = Clock::now();
      loc.MeasurementUpdate(scan, neighbors, &base);
      t_mu = ms_since(t2);
    } else {
      lh_cloud_view v = ViewOf(*scan);
      lh_cloud *q = nullptr, *fixed = nullptr, *nb = nullptr, *nb_s = nullptr, *aligned = nullptr;
      if (lh_cloud_create(ctx, &v, &q) != LH_OK) exit(5);
      if (!loc.TransformPointsToFixedFrame(q, &fixed) || !mapper.ApproxNearestNeighbors(fixed, &nb) || !loc.TransformPointsToSensorFrame(nb, &nb_s)) exit(5);
      t_nb = ms_since(t1);
      const auto t2 = Clock::now();
      loc.MeasurementUpdate(q, nb_s, &aligned);
      t_mu = ms_since(t2);
      lh_cloud_destroy(aligned); lh_cloud_destroy(nb_s); lh_cloud_destroy(nb); lh_cloud_destroy(fixed);
      if (keyframe_due(last_kf, loc.GetIntegratedEstimate())) {   // (the keyframe's cloud goes to the map without leaving the device)
        loc.MotionUpdate(gu::Transform3());
        lh_cloud* kf = nullptr;
        if (loc.TransformPointsToFixedFrame(q, &kf)) { mapper.UpdateCurrentPose(loc.GetIntegratedEstimate()); mapper.InsertPoints(kf); lh_cloud_destroy(kf); }
        last_kf = loc.GetIntegratedEstimate();
        R.keyframes++;
      }
      lh_cloud_destroy(q);
    }
    if (!device && keyframe_due(last_kf, loc.GetIntegratedEstimate())) {   // Locus.cc:505-520
      loc.MotionUpdate(gu::Transform3());
      PointCloudF fixed;
      loc.TransformPointsToFixedFrame(*scan, &fixed);
      mapper.UpdateCurrentPose(loc.GetIntegratedEstimate());
      mapper.InsertPoints(fixed, nullptr);
      last_kf = loc.GetIntegratedEstimate();
      R.keyframes++;
    }
    R.updates++;
    if (timed && (int)i > warm) {
      R.st.odom.push_back(t_od); R.st.neigh.push_back(t_nb); R.st.meas.push_back(t_mu); R.st.total.push_back(ms_since(t0));
    }
  }
  R.pose = loc.GetIntegratedEstimate();
  loc.GetLatestDeltaCovariance(R.cov);
  R.map_points = mapper.Size();
  return R;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: locus_stream scans.bin [warmup]\n"); return 2; }
  const int warm = argc > 2 ? atoi(argv[2]) : 3;
  (void)lh_runtime_init(0);
  FILE* f = fopen(argv[1], "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  int32_t n_scans = 0;
  if (fread(&n_scans, 4, 1, f) != 1 || n_scans < 3) return 2;
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
  std::vector<PointCloudF> head(scans.begin(), scans.begin() + std::min<size_t>(scans.size(), 8));
  (void)run(ctx, head, false, 0, false);   // warms the library up (buffers, streams)
  (void)run(ctx, head, true, 0, false);
  const Run H = run(ctx, scans, false, warm, true), D = run(ctx, scans, true, warm, true);
  bool same = memcmp(&H.pose, &D.pose, sizeof(H.pose)) == 0 && memcmp(H.cov, D.cov, sizeof(H.cov)) == 0 && H.keyframes == D.keyframes && H.map_points == D.map_points;
  auto emit = [&](const char* name, const Run& r) {
    printf("\"%s\": {\"ms_per_scan_median\": %.4f, \"ms_per_scan_max\": %.4f, \"ms_odometry_update_median\": %.4f, \"ms_frames_and_map_neighbours_median\": %.4f, "
           "\"ms_measurement_update_median\": %.4f, \"updates\": %d, \"timed\": %zu, \"keyframes\": %d, \"map_points\": %zu, \"integrated_translation\": [%.6f, %.6f, %.6f]}",
           name, median(r.st.total), vmax(r.st.total), median(r.st.odom), median(r.st.neigh), median(r.st.meas), r.updates, r.st.total.size(), r.keyframes, r.map_points,
           r.pose.translation.x, r.pose.translation.y, r.pose.translation.z);
  };
  printf("{\"scans\": %d, \"points_per_scan_mean\": %.1f, ", n_scans, pts / n_scans);
  emit("host_surface", H);
  printf(", ");
  emit("device_resident", D);
  printf(", \"device_equals_host_bit_for_bit\": %s}\n", same ? "true" : "false");
  lh_destroy(ctx);
  return same ? 0 : 6;
}
