// host_check.cc -- the reference's own gtest cases for the hot path, replayed against the C++ host mirror on a GPU
// (test_point_cloud_odometry.cpp:280-305; test_point_cloud_localization.cpp:243-276, 288-507, 509-528).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>

#include "PointCloudLocalization.hpp"
#include "PointCloudMapperHip.hpp"
#include "PointCloudOdometry.hpp"
#include "pcd_io.hpp"
#include "ros_msgs.hpp"

using namespace locus_hip;

static int g_fail = 0;
#define EXPECT(cond)                                                     \
  do {                                                                   \
    if (!(cond)) { printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); g_fail++; } \
  } while (0)
#define EXPECT_NEAR(a, b, eps) EXPECT(std::fabs((double)(a) - (double)(b)) <= (eps))

static PointCloudF::Ptr GenerateHollowCubic(lh_ctx* ctx, size_t nx = 10, size_t ny = 10, size_t nz = 10, float step = 0.1f) {
  PointCloudF::Ptr pc(new PointCloudF);  // test_point_cloud_odometry.cpp:60-97
  for (size_t ix = 0; ix < nx; ix++)
    for (size_t iy = 0; iy < ny; iy++)
      for (size_t iz = 0; iz < nz; iz++)
        if (ix == 0 || iy == 0 || ix == nx - 1 || iy == ny - 1) {
          PointF p;
          p.x = ix * step; p.y = iy * step; p.z = iz * step;
          pc->points.push_back(p);
        }
  std::vector<float> nrm(pc->size() * 4);
  lh_cloud_view v = ViewOf(*pc);
  v.off_normal = UINT32_MAX;
  EXPECT(lh_normals_knn(ctx, &v, 5, nrm.data()) == LH_OK);  // ne.setKSearch(5)
  for (size_t i = 0; i < pc->size(); i++) {
    pc->points[i].normal_x = nrm[4 * i]; pc->points[i].normal_y = nrm[4 * i + 1]; pc->points[i].normal_z = nrm[4 * i + 2];
  }
  return pc;
}

static PointCloudF::Ptr GeneratePlane(size_t nx = 10, size_t ny = 10, float step = 0.1f) {
  PointCloudF::Ptr pc(new PointCloudF);  // test_point_cloud_localization.cpp:26-42, normal (0,0,1)
  for (size_t ix = 0; ix < nx; ix++)
    for (size_t iy = 0; iy < ny; iy++) {
      PointF p;
      p.x = ix * step; p.y = iy * step; p.z = 0; p.normal_z = 1.0f;
      pc->points.push_back(p);
    }
  return pc;
}

static void Inverse4(const float* Tc, double* inv /*row-major*/) {  // rigid inverse
  double R[9], t[3];
  for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[r * 3 + c] = Tc[c * 4 + r]; t[r] = Tc[12 + r]; }
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) inv[r * 4 + c] = R[c * 3 + r];
    inv[r * 4 + 3] = -(R[0 * 3 + r] * t[0] + R[1 * 3 + r] * t[1] + R[2 * 3 + r] * t[2]);
  }
}

int main() {
  (void)lh_runtime_init(0);   // what a nodelet manager does first: the hardware-queue setting, before the process's first HIP call
  lh_ctx* ctx = nullptr;
  if (lh_create(&ctx, 0) != LH_OK) { printf("no HIP device: the host mirror has no CPU fallback\n"); return 2; }
  const double epsiliond = 1e-2, epsilion = 1e-4;

  {  // TEST_F(PointCloudOdometryTest, UpdateEstimateUpdateICP)
    PointCloudOdometry pco(ctx);
    PointCloudOdometry::Config cfg;  // config/parameters.yaml + icp/num_threads 2
    cfg.num_threads = 2;
    auto pc_box = GenerateHollowCubic(ctx);
    PointCloudF translated = *pc_box;
    for (auto& p : translated.points) { p.x += 0.05f; p.y += 0.05f; p.normal_x = p.normal_y = p.normal_z = 0; }
    EXPECT(pco.Initialize(cfg));
    EXPECT(pco.GetDiagnostics().level == 2);
    EXPECT(pco.SetLidar(*pc_box));
    EXPECT(!pco.UpdateEstimate());
    EXPECT(pco.SetLidar(translated));
    EXPECT(pco.UpdateEstimate());
    EXPECT(pco.icp_->hasConverged());
    EXPECT(pco.icp_->getFitnessScore() < 0.1);
    double inv[12];
    Inverse4(pco.icp_->getFinalTransformation(), inv);
    EXPECT_NEAR(inv[3], 0.05, epsiliond);
    EXPECT_NEAR(inv[7], 0.05, epsiliond);
    EXPECT_NEAR(inv[11], 0.0, epsiliond);
    EXPECT(pco.GetDiagnostics().level == 0);
    EXPECT_NEAR(pco.GetIntegratedEstimate().translation.x, pco.GetIncrementalEstimate().translation.x, 1e-12);
    PointCloudF::Ptr last(new PointCloudF);
    EXPECT(pco.GetLastPointCloud(last) && last->size() == translated.size());
    // flat-ground + imu prior paths run
    pco.SetFlatGroundAssumptionValue(true);
    pco.EnableImuIntegration();
    double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    pco.SetImuDelta(I3);
    EXPECT(pco.SetLidar(*pc_box) && pco.UpdateEstimate());
    EXPECT_NEAR(pco.GetIncrementalEstimate().translation.z, 0.0, 1e-12);
  }

  {  // a stream of scans: from the third scan on the reference cloud is the previous query, already on the GPU -- promoted there
     // (lh_gicp_promote_source_to_target) instead of uploaded again.  Same data, same index build: the poses must be IDENTICAL to
     // the wrapper that uploads both clouds every update, with and without a sensor prior in between.
    auto pc_box = GenerateHollowCubic(ctx);
    std::vector<PointCloudF> stream;
    for (int k = 0; k < 6; k++) {
      PointCloudF c = *pc_box;
      for (auto& p : c.points) { p.x += 0.02f * k; p.y -= 0.015f * k; p.z += 0.004f * (k % 3); }
      stream.push_back(c);
    }
    double poses[2][6][3];
    for (int promote = 0; promote < 2; promote++) {
      PointCloudOdometry pco(ctx);
      PointCloudOdometry::Config cfg;
      EXPECT(pco.Initialize(cfg));
      pco.EnableDevicePromotion(promote != 0);
      for (int k = 0; k < 6; k++) {
        if (k == 4) {   // one update with an odometry prior in the middle: the moved query is NOT what the next reference is
          pco.EnableOdometryIntegration();
          double prior[16] = {1, 0, 0, 0.01, 0, 1, 0, -0.01, 0, 0, 1, 0, 0, 0, 0, 1};
          pco.SetOdometryDelta(prior);
        } else
          pco.DisableSensorIntegration();
        EXPECT(pco.SetLidar(stream[k]));
        EXPECT(pco.UpdateEstimate() == (k > 0));
        poses[promote][k][0] = pco.GetIntegratedEstimate().translation.x;
        poses[promote][k][1] = pco.GetIntegratedEstimate().translation.y;
        poses[promote][k][2] = pco.GetIntegratedEstimate().translation.z;
      }
    }
    for (int k = 0; k < 6; k++)
      for (int a = 0; a < 3; a++) EXPECT(poses[0][k][a] == poses[1][k][a]);
    EXPECT(std::fabs(poses[1][5][0]) > 0.05);   // (and the stream did move)
  }

  {  // a NON-identity odometry prior (PointCloudOdometry.cc:256-275): the query is moved by the prior in double arithmetic
     // (pcl::transformPointCloud with a Matrix4d), GICP only finds the remainder, and T * prior is the whole motion
    PointCloudOdometry pco(ctx);
    PointCloudOdometry::Config cfg;
    cfg.num_threads = 2;
    auto pc_box = GenerateHollowCubic(ctx);
    PointCloudF translated = *pc_box;
    for (auto& p : translated.points) { p.x += 0.05f; p.y += 0.05f; }
    EXPECT(pco.Initialize(cfg));
    pco.EnableOdometryIntegration();
    const double prior[16] = {1, 0, 0, -0.04, 0, 1, 0, -0.05, 0, 0, 1, 0, 0, 0, 0, 1};  // most of the motion, row-major
    pco.SetOdometryDelta(prior);
    EXPECT(pco.SetLidar(*pc_box));
    EXPECT(!pco.UpdateEstimate());
    EXPECT(pco.SetLidar(translated));
    EXPECT(pco.UpdateEstimate());
    const float* Tg = pco.icp_->getFinalTransformation();  // what GICP itself had to find: about (-0.01, 0, 0)
    EXPECT_NEAR(Tg[12], -0.01, epsiliond);
    EXPECT_NEAR(Tg[13], 0.0, epsiliond);
    EXPECT_NEAR(pco.GetIncrementalEstimate().translation.x, -0.05, epsiliond);
    EXPECT_NEAR(pco.GetIncrementalEstimate().translation.y, -0.05, epsiliond);
    EXPECT_NEAR(pco.GetIncrementalEstimate().translation.z, 0.0, epsiliond);
    // the prior is applied in double and rounded once: the source the registration saw is float(prior * double(p))
    PointCloudF::Ptr src = pco.GetQueryTransformed();
    EXPECT(src && src->size() == translated.size());
    const double px = (double)(float)-0.04, py = (double)(float)-0.05;  // pcl_ros::transformAsMatrix fills a Matrix4f first (:256-259)
    if (src && src->size() == translated.size())
      for (size_t i = 0; i < src->size(); i += 37) {
        EXPECT(src->points[i].x == static_cast<float>(1.0 * (double)translated.points[i].x + 0.0 * (double)translated.points[i].y +
                                                      0.0 * (double)translated.points[i].z + px));
        EXPECT(src->points[i].y == static_cast<float>(0.0 * (double)translated.points[i].x + 1.0 * (double)translated.points[i].y +
                                                      0.0 * (double)translated.points[i].z + py));
      }
  }

  {  // TransformPointsToFixedFrame / ToSensorFrame (z = +-3) and the covariance KATs
    PointCloudLocalization pcl_(ctx);
    PointCloudLocalization::Config cfg;
    cfg.num_threads = 2;
    EXPECT(pcl_.Initialize(cfg));
    gu::Transform3 e;
    e.translation = gu::Vec3(0, 0, 3);
    pcl_.SetIntegratedEstimate(e);
    auto plane = GeneratePlane();
    PointCloudF out, back;
    EXPECT(pcl_.TransformPointsToFixedFrame(*plane, &out));
    for (auto& p : out.points) { EXPECT_NEAR(p.z, 3.0, 1e-6); EXPECT_NEAR(p.normal_z, 1.0, 1e-6); }
    EXPECT(pcl_.TransformPointsToSensorFrame(out, &back));
    for (size_t i = 0; i < back.size(); i++) EXPECT_NEAR(back.points[i].z, 0.0, 1e-6);
    EXPECT(!pcl_.TransformPointsToFixedFrame(*plane, NULL));

    // ComputePoint2PlaneICPCovariance: single plane -> diag 0.01 (clamp)
    std::vector<size_t> corr(plane->size());
    std::iota(corr.begin(), corr.end(), 0);
    float I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    double cov[36];
    pcl_.ComputePoint2PlaneICPCovariance(*plane, *plane, corr, I16, cov);
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) EXPECT_NEAR(cov[i * 6 + j], i == j ? 0.01 : 0.0, epsilion);
    // observability eigenvalues of the plane: (0,0,0,56.7753,56.7753,100), Ap diag KAT
    double evec[36], eval[6], A[36];
    pcl_.ComputeIcpObservability(*plane, *plane, corr, I16, evec, eval, A);
    EXPECT_NEAR(A[0], 56.7753, epsilion); EXPECT_NEAR(A[7], 56.7753, epsilion); EXPECT_NEAR(A[35], 100.0, epsilion);
    EXPECT_NEAR(eval[0], 0, epsilion); EXPECT_NEAR(eval[2], 0, epsilion); EXPECT_NEAR(eval[3], 56.7753, epsilion); EXPECT_NEAR(eval[5], 100.0, epsilion);

    // three orthogonal planes + yaw-30deg offset -> ~1e-6 on the diagonal
    PointCloudF all = *plane, p1 = *plane, p2 = *plane;
    for (auto& p : p1.points) { float y = p.y, z = p.z, ny = p.normal_y, nz = p.normal_z; p.y = -z; p.z = y; p.normal_y = -nz; p.normal_z = ny; }
    for (auto& p : p2.points) { float x = p.x, z = p.z, nx = p.normal_x, nz = p.normal_z; p.x = z; p.z = -x; p.normal_x = nz; p.normal_z = -nx; }
    all.points.insert(all.points.end(), p1.points.begin(), p1.points.end());
    all.points.insert(all.points.end(), p2.points.begin(), p2.points.end());
    PointCloudF offset = all;
    for (auto& p : offset.points) {
      float x = p.x, y = p.y, nx = p.normal_x, ny = p.normal_y;
      p.x = 0.866f * x - 0.5f * y + 0.001f; p.y = 0.5f * x + 0.866f * y; p.z = 0;
      p.normal_x = 0.866f * nx - 0.5f * ny; p.normal_y = 0.5f * nx + 0.866f * ny; p.normal_z = 0;
    }
    std::vector<size_t> corr2(all.size());
    std::iota(corr2.begin(), corr2.end(), 0);
    EXPECT(pcl_.ComputePoint2PlaneICPCovariance(offset, all, corr2, I16, cov));
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) EXPECT_NEAR(cov[i * 6 + j], i == j ? 1e-6 : 0.0, epsilion);

    // MeasurementUpdate: null output -> level 2; plane vs plane -> level 0
    PointCloudF::Ptr q = GeneratePlane(), r = GeneratePlane();
    EXPECT(!pcl_.MeasurementUpdate(q, r, NULL));
    EXPECT(pcl_.GetDiagnostics().level == 2);
    PointCloudF aligned;
    EXPECT(pcl_.MeasurementUpdate(q, r, &aligned));
    EXPECT(pcl_.GetDiagnostics().level == 0);
    EXPECT(aligned.size() == q->size());
    double dc[36];
    pcl_.GetLatestDeltaCovariance(dc);
    EXPECT_NEAR(dc[0], 0.01, epsilion);
    EXPECT(pcl_.MotionUpdate(gu::Transform3()));

    // device_flow: the same three calls on clouds that live in HBM give the host surface's results bit for bit -- frame transforms (the
    // z +- 3 KAT of test_point_cloud_localization.cpp), MeasurementUpdate's aligned query, poses and covariance
    {
      PointCloudLocalization ha(ctx), da(ctx);
      PointCloudLocalization::Config lc;
      lc.initial_pose.translation = gu::Vec3(0.3, -0.2, 3.0);
      lc.initial_pose.rotation = gu::Rot3(0.01, -0.02, 0.3);
      lc.compute_icp_observability = true;
      EXPECT(ha.Initialize(lc) && da.Initialize(lc));
      PointCloudF::Ptr scan = GenerateHollowCubic(ctx), ref = GenerateHollowCubic(ctx);
      for (auto& p : scan->points) { p.x += 0.02f; p.y -= 0.01f; }
      gu::Transform3 inc;
      inc.translation = gu::Vec3(0.05, 0.0, -0.01);
      ha.MotionUpdate(inc); da.MotionUpdate(inc);
      lh_cloud_view vs = ViewOf(*scan), vr = ViewOf(*ref);
      lh_cloud *cs = nullptr, *cr = nullptr, *fixed = nullptr, *back = nullptr, *al = nullptr;
      EXPECT(lh_cloud_create(ctx, &vs, &cs) == LH_OK && lh_cloud_create(ctx, &vr, &cr) == LH_OK);
      PointCloudF h_fixed, h_back, d_dl;
      EXPECT(ha.TransformPointsToFixedFrame(*scan, &h_fixed) && da.TransformPointsToFixedFrame(cs, &fixed));
      auto same_cloud = [&](const lh_cloud* c, const PointCloudF& h) {
        PointCloudF dl;
        dl.points.assign(lh_cloud_size(c), PointF());
        if (dl.size() != h.size()) return false;
        if (lh_cloud_download(c, dl.points.data(), sizeof(PointF), offsetof(PointF, x), offsetof(PointF, normal_x), offsetof(PointF, intensity), offsetof(PointF, curvature)) != LH_OK) return false;
        for (size_t i = 0; i < h.size(); i++)
          if (memcmp(&dl.points[i].x, &h.points[i].x, 12) != 0 || memcmp(&dl.points[i].normal_x, &h.points[i].normal_x, 12) != 0) return false;
        return true;
      };
      EXPECT(same_cloud(fixed, h_fixed));
      EXPECT(ha.TransformPointsToSensorFrame(h_fixed, &h_back) && da.TransformPointsToSensorFrame(fixed, &back));
      EXPECT(same_cloud(back, h_back));
      PointCloudF h_al;
      EXPECT(ha.MeasurementUpdate(scan, ref, &h_al) && da.MeasurementUpdate(cs, cr, &al));
      EXPECT(al != nullptr && same_cloud(al, h_al));
      EXPECT(memcmp(&ha.GetIntegratedEstimate(), &da.GetIntegratedEstimate(), sizeof(gu::Transform3)) == 0);
      EXPECT(memcmp(&ha.GetIncrementalEstimate(), &da.GetIncrementalEstimate(), sizeof(gu::Transform3)) == 0);
      double hc[36], dcv[36];
      ha.GetLatestDeltaCovariance(hc); da.GetLatestDeltaCovariance(dcv);
      EXPECT(memcmp(hc, dcv, sizeof(hc)) == 0);
      EXPECT(ha.condition_number() == da.condition_number());
      lh_cloud_destroy(al); lh_cloud_destroy(back); lh_cloud_destroy(fixed); lh_cloud_destroy(cs); lh_cloud_destroy(cr);
    }
  }

  {  // the same odometry scenario with registration_method "ndt" (SetupICP NDT branch, PointCloudOdometry.cc:182-196): a dense
     // wall pattern (NDT needs >= 6 points per 1-m voxel), second scan translated by 5 cm
    PointCloudOdometry pco(ctx);
    PointCloudOdometry::Config cfg;
    cfg.registration_method = "ndt";
    cfg.icp_tf_epsilon = 1e-3;
    cfg.icp_iterations = 30;
    PointCloudF::Ptr room(new PointCloudF);
    for (int i = 0; i < 120; i++)
      for (int j = 0; j < 40; j++) {
        float u = 0.1f * i - 6.0f, v = 0.1f * j - 1.0f, w = 0.003f * (float)((i * 7 + j * 13) % 10);  // a little relief: non-degenerate voxels
        PointF a, b, c2, d;
        a.x = 6.0f + w; a.y = u; a.z = v;
        b.x = -6.0f - w; b.y = u; b.z = v;
        c2.x = u; c2.y = 6.0f + w; c2.z = v;
        d.x = u; d.y = -6.0f - w; d.z = v;
        room->points.push_back(a); room->points.push_back(b); room->points.push_back(c2); room->points.push_back(d);
      }
    for (int i = 0; i < 120; i++)
      for (int j = 0; j < 120; j++) { PointF f; f.x = 0.1f * i - 6.0f; f.y = 0.1f * j - 6.0f; f.z = -1.0f - 0.002f * (float)((i + j) % 7); room->points.push_back(f); }
    PointCloudF moved = *room;
    for (auto& p : moved.points) { p.x += 0.05f; p.y -= 0.03f; }
    EXPECT(pco.Initialize(cfg));
    EXPECT(pco.SetLidar(*room));
    EXPECT(!pco.UpdateEstimate());
    EXPECT(pco.SetLidar(moved));
    EXPECT(pco.UpdateEstimate());
    EXPECT(pco.icp_->hasConverged());
    double inv[12];
    Inverse4(pco.icp_->getFinalTransformation(), inv);
    EXPECT_NEAR(inv[3], 0.05, epsiliond);
    EXPECT_NEAR(inv[7], -0.03, epsiliond);
    EXPECT_NEAR(inv[11], 0.0, epsiliond);
    EXPECT(pco.icp_->getFitnessScore() < 0.01);
    bool threw = false;
    try { PointCloudOdometry bad(ctx); PointCloudOdometry::Config c3; c3.registration_method = "icp"; bad.Initialize(c3); } catch (const std::exception&) { threw = true; }
    EXPECT(threw);  // getRegistrationMethodFromString: unknown method (registration_settings.h:13-20)
  }

  {  // mapper_ call order of Locus.cc:462-489 / 531-538 on the device-resident map (no reference test exists: the mapper is un-vendored)
    PointCloudMapperHip mapper(ctx);
    EXPECT(mapper.Initialize(0.05));
    mapper.SetBoxFilterSize(1);
    auto box = GenerateHollowCubic(ctx);
    PointCloudF inc;
    EXPECT(mapper.InsertPoints(*box, &inc));
    EXPECT(inc.size() == mapper.Size() && inc.size() > 0 && inc.size() <= box->size());
    size_t first = mapper.Size();
    EXPECT(mapper.InsertPoints(*box, &inc));          // the same points again: every voxel is taken
    EXPECT(inc.size() == 0 && mapper.Size() == first);
    PointCloudF shifted = *box, nb;
    for (auto& p : shifted.points) p.x += 0.004f;     // stays inside the 5-cm voxels' neighbourhood
    EXPECT(mapper.ApproxNearestNeighbors(shifted, &nb));
    EXPECT(nb.size() == shifted.size());
    double worst = 0;
    for (size_t i = 0; i < nb.size(); i++) {
      double dx = nb.points[i].x - shifted.points[i].x, dy = nb.points[i].y - shifted.points[i].y, dz = nb.points[i].z - shifted.points[i].z;
      worst = std::max(worst, dx * dx + dy * dy + dz * dz);
    }
    EXPECT(worst < 0.11 * 0.11);                      // a map point within one grid step (0.1 m) of every query
    gu::Transform3 pose;                              // identity: the cube spans [0, 0.9]^3, so a 1-m box keeps all of it
    mapper.UpdateCurrentPose(pose);
    mapper.Refresh(pose);
    EXPECT(mapper.Size() == first);
    pose.translation = gu::Vec3(5.0, 0.0, 0.0);
    mapper.Refresh(pose);                             // window moved away: nothing left
    EXPECT(mapper.Size() == 0);
    EXPECT(!mapper.ApproxNearestNeighbors(shifted, &nb));
  }

  {  // wire formats THROUGH the device: the filter nodelets receive sensor_msgs/PointCloud2 and hand PCL clouds on
     // (pcl::fromROSMsg / toROSMsg, normal_computation.cc:30,58; pcl_conversions::toPCL, custom_voxel_grid.cc:81).  A velodyne
     // driver message (x y z @0, intensity @16, ring @20, time @24; point_step 32) is read IN PLACE by lh_cloud_create through
     // the view ros_msgs.hpp derives from its field table, runs K1 -> K3 on the GPU, and comes back in the layout pcl::toROSMsg
     // gives a PointCloud<PointXYZINormal>.
    const uint32_t n = 5000;
    PointCloud2 in;
    in.width = n; in.height = 1; in.point_step = 32; in.row_step = 32 * n;
    auto fld = [](const char* name, uint32_t off, uint8_t type) { PointField f; f.name = name; f.offset = off; f.datatype = type; return f; };
    in.fields = {fld("x", 0, PointField::FLOAT32), fld("y", 4, PointField::FLOAT32), fld("z", 8, PointField::FLOAT32),
                 fld("intensity", 16, PointField::FLOAT32), fld("ring", 20, PointField::UINT16), fld("time", 24, PointField::FLOAT32)};
    in.data.assign((size_t)in.row_step, 0xAB);   // the padding bytes are garbage on purpose
    PointCloudF same;                            // the same points as a PCL array, for the reference route
    for (uint32_t i = 0; i < n; i++) {
      float u = 0.013f * (float)(i % 97), v = 0.017f * (float)(i / 97), w = 0.002f * (float)((i * 31) % 11);
      float xyz[3] = {u, v, (i % 3 == 0) ? w : 1.5f + w}, inten = (float)(i % 255);
      uint16_t ring = (uint16_t)(i % 16);
      memcpy(&in.data[(size_t)i * 32 + 0], xyz, 12);
      memcpy(&in.data[(size_t)i * 32 + 16], &inten, 4);
      memcpy(&in.data[(size_t)i * 32 + 20], &ring, 2);
      PointF p; p.x = xyz[0]; p.y = xyz[1]; p.z = xyz[2]; p.intensity = inten;
      same.points.push_back(p);
    }
    lh_cloud_view v;
    EXPECT(ViewFromPointCloud2(in, &v));
    EXPECT(v.base == in.data.data() && v.stride == 32 && v.off_xyz == 0 && v.off_intensity == 16 && v.off_normal == UINT32_MAX);
    lh_cloud *c_msg = nullptr, *c_pcl = nullptr, *vox_msg = nullptr, *vox_pcl = nullptr;
    EXPECT(lh_cloud_create(ctx, &v, &c_msg) == LH_OK);
    lh_cloud_view vp = ViewOf(same);
    vp.off_normal = UINT32_MAX; vp.off_curvature = UINT32_MAX;
    EXPECT(lh_cloud_create(ctx, &vp, &c_pcl) == LH_OK);
    EXPECT(c_msg && c_pcl && lh_cloud_size(c_msg) == n && lh_cloud_size(c_pcl) == n);
    if (c_msg && c_pcl) {
      // upload from the message == upload from the PCL array, bit for bit
      PointCloud2 back;
      LayoutPointCloud2(&back, n, false);          // pcl::toROSMsg layout of PointCloud<PointXYZI>
      lh_cloud_view vb;
      EXPECT(ViewFromPointCloud2(back, &vb));
      EXPECT(lh_cloud_download(c_msg, back.data.data(), vb.stride, vb.off_xyz, vb.off_normal, vb.off_intensity, vb.off_curvature) == LH_OK);
      bool ok = true;
      for (uint32_t i = 0; i < n && ok; i++)
        ok = memcmp(&back.data[(size_t)i * vb.stride + vb.off_xyz], &in.data[(size_t)i * 32], 12) == 0 &&
             memcmp(&back.data[(size_t)i * vb.stride + vb.off_intensity], &in.data[(size_t)i * 32 + 16], 4) == 0;
      EXPECT(ok);
      // CustomVoxelGrid::filter on the message == on the PCL array (custom_voxel_grid.cc:76-87), then NormalComputation::filter
      EXPECT(lh_cloud_voxel_grid(c_msg, 0.1f, -1, -3e38, 3e38, &vox_msg) == LH_OK);
      EXPECT(lh_cloud_voxel_grid(c_pcl, 0.1f, -1, -3e38, 3e38, &vox_pcl) == LH_OK);
      EXPECT(vox_msg && vox_pcl && lh_cloud_size(vox_msg) == lh_cloud_size(vox_pcl) && lh_cloud_size(vox_msg) > 100);
      if (vox_msg && vox_pcl) {
        EXPECT(lh_normals_knn_cloud(vox_msg, 10) == LH_OK);
        EXPECT(lh_normals_knn_cloud(vox_pcl, 10) == LH_OK);
        const uint32_t m = lh_cloud_size(vox_msg);
        PointCloud2 out;                            // what NormalComputation publishes: toROSMsg(PointCloud<PointXYZINormal>)
        LayoutPointCloud2(&out, m, true);
        lh_cloud_view vo;
        EXPECT(ViewFromPointCloud2(out, &vo));
        EXPECT(vo.stride == 48 && vo.off_normal == 16 && vo.off_intensity == 32 && vo.off_curvature == 36);
        EXPECT(lh_cloud_download(vox_msg, out.data.data(), vo.stride, vo.off_xyz, vo.off_normal, vo.off_intensity, vo.off_curvature) == LH_OK);
        PointCloudF ref;
        ref.points.resize(m);
        lh_cloud_view vr = ViewOf(ref);
        EXPECT(lh_cloud_download(vox_pcl, ref.points.data(), vr.stride, vr.off_xyz, vr.off_normal, vr.off_intensity, vr.off_curvature) == LH_OK);
        bool same_out = true;
        int unit = 0;
        for (uint32_t i = 0; i < m; i++) {
          const uint8_t* p = &out.data[(size_t)i * 48];
          const PointF& r = ref.points[i];
          same_out = same_out && memcmp(p + 0, &r.x, 12) == 0 && memcmp(p + 16, &r.normal_x, 12) == 0 && memcmp(p + 32, &r.intensity, 4) == 0 &&
                     memcmp(p + 36, &r.curvature, 4) == 0;
          double l = std::sqrt((double)r.normal_x * r.normal_x + (double)r.normal_y * r.normal_y + (double)r.normal_z * r.normal_z);
          unit += std::fabs(l - 1.0) < 1e-4;
        }
        EXPECT(same_out);                          // message route == PCL route, every field of every point
        EXPECT(unit > (int)(0.95 * m));
        // and a map dump (Locus.cc:749-751): the published cloud written as PCD v0.7 binary and read back
        const char* tmp = "/tmp/locus_hip_host_check.pcd";
        EXPECT(WritePCDBinary(tmp, ref));
        PointCloudF rd;
        EXPECT(ReadPCD(tmp, &rd) && rd.size() == ref.size());
        bool pcd_ok = rd.size() == ref.size();
        for (size_t i = 0; pcd_ok && i < ref.size(); i++) pcd_ok = memcmp(&rd.points[i], &ref.points[i], sizeof(PointF)) == 0;
        EXPECT(pcd_ok);
        remove(tmp);
      }
    }
    lh_cloud_destroy(vox_msg); lh_cloud_destroy(vox_pcl); lh_cloud_destroy(c_msg); lh_cloud_destroy(c_pcl);
    // a layout the C ABI cannot read in place (double coordinates) is refused before anything is uploaded
    PointCloud2 bad = in;
    bad.fields[0].datatype = PointField::FLOAT64;
    EXPECT(!ViewFromPointCloud2(bad, &v));
  }

  lh_destroy(ctx);
  printf(g_fail ? "HOST_CHECK_FAILED (%d)\n" : "HOST_CHECK_OK\n", g_fail);
  return g_fail ? 1 : 0;
}
