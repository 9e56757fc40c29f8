// lh_math.hpp -- the handful of elementary functions the BFGS solve needs (sin/cos of the Euler angles, atan2/asin for
// the state extraction, gicp.hpp:160-214, 234-241, 619-634) as PLAIN IEEE double arithmetic, identical on the host and
// on the device.
//
// Why not libm: in cost_mode 1 the solve of an outer iteration runs on the GPU (k_solve) and, for the source-sharded pair
// (lh_set_allreduce), on the host.  glibc's and the device library's sin / atan2 differ in their last bits, and the BFGS
// line search compares cost values at their last bits, so the two would wander apart by the reference's noise floor.
// With one arithmetic definition both run the same trajectory bit for bit (tests/test_gpu_solver.py), and the host-side
// emulation of the device code in tests/host_emu is exact.  Only +, -, *, /, sqrt, rint and comparisons are used; the TUs
// are compiled with -ffp-contract=off, so no fused multiply-add sneaks in on either side.
//   pm_sincos: Cody-Waite reduction by pi/2 in three pieces (exact products for |k| < 2^20), then the classic degree-13 /
//              degree-14 minimax kernels on [-pi/4, pi/4] (Sun fdlibm's published coefficients); <= 1 ulp vs libm
//   pm_atan2 : octant reduction + a 5e-3 starting guess + three steps of theta += t (1 - t^2/3),
//              t = (y cos - x sin) / (x cos + y sin) (fifth-order: converged after two); <= 2 ulp
//   pm_asin  : atan2(v, sqrt((1 - v)(1 + v)))
// cost_mode 0 keeps libm on the host: it is the reference-arithmetic mode and must follow the oracle (which uses libm) bit
// for bit.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LH_FN __host__ __device__ inline
#else
#define LH_FN inline
#endif

namespace lh {

LH_FN double pm_ksin(double x) {  // |x| <= pi/4
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x, v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x + v * (S1 + z * r);
}
LH_FN double pm_kcos(double x) {  // |x| <= pi/4
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  return 1.0 - (0.5 * z - z * r);
}
LH_FN void pm_sincos(double x, double* s, double* c) {
  const double TWO_OVER_PI = 0x1.45f306dc9c883p-1;
  const double P1 = 0x1.921fb54400000p+0, P2 = 0x1.0b4611a600000p-34, P3 = 0x1.3198a2e037073p-69;  // pi/2 = P1 + P2 + P3
  if (!(fabs(x) < 1.0e6)) {  // NaN, infinity, or far outside anything an Euler angle of a scan-to-scan motion can be
    *s = x - x; *c = x - x;  // NaN (an unusable step: the line search rejects it like any other non-finite cost)
    return;
  }
  double fk = rint(x * TWO_OVER_PI);
  double r = ((x - fk * P1) - fk * P2) - fk * P3;
  double sn = pm_ksin(r), cs = pm_kcos(r);
  int q = (int)((long long)fk & 3LL);
  if (q == 0) { *s = sn; *c = cs; }
  else if (q == 1) { *s = cs; *c = -sn; }
  else if (q == 2) { *s = -sn; *c = -cs; }
  else { *s = -cs; *c = sn; }
}
LH_FN double pm_atan2(double y, double x) {
  const double PIO2 = 0x1.921fb54442d18p+0, PI = 0x1.921fb54442d18p+1;
  if (x != x || y != y) return x + y;
  double ax = fabs(x), ay = fabs(y);
  if (ax == 0.0 && ay == 0.0) return 0.0;
  double th;
  if (ay <= ax) {
    double a = ay / ax;
    th = a * (0.9724 - 0.1919 * a * a);
  } else {
    double a = ax / ay;
    th = PIO2 - a * (0.9724 - 0.1919 * a * a);
  }
  for (int it = 0; it < 3; it++) {
    double s, c;
    pm_sincos(th, &s, &c);
    double t = (ay * c - ax * s) / (ax * c + ay * s);
    th = th + t * (1.0 - t * t * (1.0 / 3.0));
  }
  if (x < 0.0) th = PI - th;
  return y < 0.0 ? -th : th;
}
LH_FN double pm_asin(double v) { return pm_atan2(v, sqrt((1.0 - v) * (1.0 + v))); }

// the two flavours of the solver's elementary functions
struct LibmMath {      // host only: glibc, like the reference build (and the oracle)
  static inline void sincos_d(double x, double* s, double* c) { *s = sin(x); *c = cos(x); }
  static inline void sincos_f(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
  static inline double atan2_d(double y, double x) { return atan2(y, x); }
  static inline double asin_d(double v) { return asin(v); }
};
struct PortableMath {  // host and device: identical bits on both
  LH_FN static void sincos_d(double x, double* s, double* c) { pm_sincos(x, s, c); }
  LH_FN static void sincos_f(float x, float* s, float* c) {  // float in, float out (Eigen::AngleAxisf -> Quaternionf, gicp.hpp:619-634)
    double sd, cd;
    pm_sincos((double)x, &sd, &cd);
    *s = (float)sd; *c = (float)cd;
  }
  LH_FN static double atan2_d(double y, double x) { return pm_atan2(y, x); }
  LH_FN static double asin_d(double v) { return pm_asin(v); }
};

}  // namespace lh
