// lh_bfgs.hpp -- host-side solver of the product path: estimateRigidTransformationBFGS (gicp.hpp:218-287)
// on top of pcl::BFGS (PCL 1.10 registration/bfgs.h: Eigen port of GSL vector_bfgs2 + Fletcher line search).
//
// The cost functor is abstract: every evaluation request (x -> f, g) is served by ONE fused device pass
// (k_cost computes the 13 sums that operator(), df and fdf of gicp.hpp:291-402 all share), so the wrapper
// caches the last evaluated x: the reference's "f(alpha) then df(alpha)" pair costs one pass instead of two
// and returns bit-identical numbers for both.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstring>

namespace lh {

enum { BFGS_RUNNING = -1, BFGS_SUCCESS = 0, BFGS_NOPROGRESS = 1 };

// ---- state <-> matrix -------------------------------------------------------------------------------
// applyState (gicp.hpp:619-634), float quaternion path of Eigen's AngleAxisf products; T = 16 floats column-major
inline void apply_state(const double* x, float* T) {
  struct Q { float w, x, y, z; };
  auto mul = [](Q a, Q b) {
    Q r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
  };
  float hz = 0.5f * (float)x[5], hy = 0.5f * (float)x[4], hx = 0.5f * (float)x[3];
  Q qz{cosf(hz), 0.f, 0.f, sinf(hz)}, qy{cosf(hy), 0.f, sinf(hy), 0.f}, qx{cosf(hx), sinf(hx), 0.f, 0.f};
  Q q = mul(mul(qz, qy), qx);
  float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  float R[9] = {1.0f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0f - (txx + tzz), tyz - twx,
                txz - twy, tyz + twx, 1.0f - (txx + tyy)};
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) T[c * 4 + r] = R[r * 3 + c];
  T[3] = T[7] = T[11] = 0.0f;
  T[12] = (float)x[0]; T[13] = (float)x[1]; T[14] = (float)x[2]; T[15] = 1.0f;
}

// computeRDerivative (gicp.hpp:160-214) + matricesInnerProd (gicp.h:361-370); R row-major
inline void compute_r_derivative(const double* x, const double* R, double* g) {
  double phi = x[3], theta = x[4], psi = x[5];
  double cphi = cos(phi), sphi = sin(phi), cth = cos(theta), sth = sin(theta), cpsi = cos(psi), spsi = sin(psi);
  double dPhi[9] = {0, sphi * spsi + cphi * cpsi * sth, cphi * spsi - cpsi * sphi * sth,
                    0, -cpsi * sphi + cphi * spsi * sth, -cphi * cpsi - sphi * spsi * sth,
                    0, cphi * cth, -cth * sphi};
  double dTheta[9] = {-cpsi * sth, cpsi * cth * sphi, cphi * cpsi * cth,
                      -spsi * sth, cth * sphi * spsi, cphi * cth * spsi,
                      -cth, -sphi * sth, -cphi * sth};
  double dPsi[9] = {-cth * spsi, -cphi * cpsi - sphi * spsi * sth, cpsi * sphi - cphi * spsi * sth,
                    cpsi * cth, -cphi * spsi + cpsi * sphi * sth, sphi * spsi + cphi * cpsi * sth,
                    0, 0, 0};
  double r3 = 0, r4 = 0, r5 = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      r3 += dPhi[j * 3 + i] * R[i * 3 + j];
      r4 += dTheta[j * 3 + i] * R[i * 3 + j];
      r5 += dPsi[j * 3 + i] * R[i * 3 + j];
    }
  g[3] = r3; g[4] = r4; g[5] = r5;
}

// f /= m, g_t *= 2/m, R *= 2/m, rotation gradient (gicp.hpp:398-401); S = f, g_t[3], R[9]
inline void cost_finish(const double* S, double m, const double* x, double* f, double* g) {
  *f = S[0] / m;
  double s = 2.0 / m;
  g[0] = S[1] * s; g[1] = S[2] * s; g[2] = S[3] * s;
  double R[9];
  for (int i = 0; i < 9; i++) R[i] = S[4 + i] * s;
  compute_r_derivative(x, R, g);
}

// ---- functor with a one-entry cache ------------------------------------------------------------------
struct CostFn {
  virtual ~CostFn() {}
  // one fused device pass: returns the 13 sums + correspondence count
  virtual void pass(const double x[6], double sums13[13], double* count) = 0;
  bool have = false;
  double cx[6], cf = 0, cg[6], cm = 0;
  int passes = 0;
  void eval(const double x[6], double* f, double* g) {
    if (!have || memcmp(x, cx, sizeof(cx)) != 0) {
      double S[13];
      pass(x, S, &cm);
      passes++;
      memcpy(cx, x, sizeof(cx));
      if (cm > 0) cost_finish(S, cm, x, &cf, cg);
      else { cf = 0; memset(cg, 0, sizeof(cg)); }
      have = true;
    }
    if (f) *f = cf;
    if (g) memcpy(g, cg, sizeof(cg));
  }
  double count() const { return cm; }
};

// ---- second-order moment model of the cost (cost_mode 1) ------------------------------------------------
// Device side: k_moments (lh_kernels.hip).  S = c0, B[3][4], H[6][10], count; T0 = transformation_ at sweep time.
struct MomentModel {
  double S[74];
  float T0[16];    // column-major
  double H12[144]; // expanded symmetric 12x12 form of H, row (r,c) = 4r+c, filled by prepare()
  static int sym3(int r, int s) { static const int m[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}}; return m[r][s]; }
  static int sym4(int c, int e) { static const int m[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}}; return m[c][e]; }
  double count() const { return S[73]; }
  void prepare() {  // once per sweep; the ~600 evaluations of a pair's BFGS solves then cost one 12x12 mat-vec each
    const double* H = S + 13;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++)
        for (int s = 0; s < 3; s++)
          for (int e = 0; e < 4; e++) H12[(4 * r + c) * 12 + 4 * s + e] = H[sym3(r, s) * 10 + sym4(c, e)];
  }
  // the 13 sums of gicp.hpp:388-396 at T (column-major float matrix from applyState)
  void sums(const float* T16, double* sums13) const {
    double d[12];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 4; c++) d[4 * r + c] = (double)T16[c * 4 + r] - (double)T0[c * 4 + r];
    const double* B = S + 1;
    double G[12], f = S[0];
    for (int i = 0; i < 12; i++) {
      const double* row = H12 + 12 * i;
      double g = 0.0;
      for (int k = 0; k < 12; k++) g += row[k] * d[k];  // same summation order (s outer, e inner) as the sym-indexed form
      G[i] = B[i] + g;
    }
    for (int k = 0; k < 12; k++) f += d[k] * (B[k] + G[k]);
    sums13[0] = f;
    sums13[1] = G[3]; sums13[2] = G[7]; sums13[3] = G[11];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) sums13[4 + 3 * a + b] = G[4 * b + a];  // R(a,b) = sum p_a (M res)_b
  }
};

// ---- pcl::BFGS ----------------------------------------------------------------------------------------
struct Bfgs {
  CostFn* fn = nullptr;
  double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 1;  // gicp.hpp:253-257
  int order = 3, bracket_iters = 100, section_iters = 100;
  double f = 0, delta_f = 0, fp0 = 0, pnorm = 0, g0norm = 0;
  double x0[6], g0[6], dx0[6], dg0[6], p[6], gradient[6];
  double x_alpha[6], g_alpha[6], f_alpha = 0, df_alpha = 0, f_key = 0, df_key = 0, x_key = 0, g_key = 0;

  static double dot(const double* a, const double* b) {
    double s = 0;
    for (int i = 0; i < 6; i++) s += a[i] * b[i];
    return s;
  }
  static double norm(const double* a) { return sqrt(dot(a, a)); }

  void moveto(double alpha) {
    if (alpha == x_key) return;
    for (int i = 0; i < 6; i++) x_alpha[i] = x0[i] + alpha * p[i];
    x_key = alpha;
  }
  double slope() const { return dot(g_alpha, p); }
  double apply_f(double alpha) {
    if (alpha == f_key) return f_alpha;
    moveto(alpha);
    fn->eval(x_alpha, &f_alpha, nullptr);
    f_key = alpha;
    return f_alpha;
  }
  double apply_df(double alpha) {
    if (alpha == df_key) return df_alpha;
    moveto(alpha);
    if (alpha != g_key) {
      fn->eval(x_alpha, nullptr, g_alpha);
      g_key = alpha;
    }
    df_alpha = slope();
    df_key = alpha;
    return df_alpha;
  }
  void apply_fdf(double alpha, double* fo, double* dfo) {
    if (alpha == f_key && alpha == df_key) { *fo = f_alpha; *dfo = df_alpha; return; }
    if (alpha == f_key || alpha == df_key) { *fo = apply_f(alpha); *dfo = apply_df(alpha); return; }
    moveto(alpha);
    fn->eval(x_alpha, &f_alpha, g_alpha);
    f_key = alpha; g_key = alpha;
    df_alpha = slope();
    df_key = alpha;
    *fo = f_alpha; *dfo = df_alpha;
  }

  static double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }

  static double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax, int order) {
    double y, ymin = (xmin - a) / (b - a), ymax = (xmax - a) / (b - a), fmin;
    if (ymin > ymax) { double t = ymin; ymin = ymax; ymax = t; }
    if (order > 2 && !(fpb != fpb) && fpb != INFINITY) {
      fpa = fpa * (b - a);
      fpb = fpb * (b - a);
      double eta = 3 * (fb - fa) - 2 * fpa - fpb, xi = fpa + fpb - 2 * (fb - fa);
      double c0 = fa, c1 = fpa, c2 = eta, c3 = xi;
      y = ymin;
      fmin = cubic(c0, c1, c2, c3, ymin);
      auto check = [&](double z) { double v = cubic(c0, c1, c2, c3, z); if (v < fmin) { y = z; fmin = v; } };
      check(ymax);
      double qa = 3 * c3, qb = 2 * c2, qc = c1;
      double disc = qb * qb - 4 * qc * qa;
      if (disc > 0) {
        double sd = sqrt(disc);
        double y0 = (-qb - sd) / (2 * qa), y1 = (-qb + sd) / (2 * qa);
        if (y0 > y1) { double t = y0; y0 = y1; y1 = t; }
        if (y0 > ymin && y0 < ymax) check(y0);
        if (y1 > ymin && y1 < ymax) check(y1);
      } else if (disc == 0) {
        double y0 = -qb / (2 * qa);
        if (y0 > ymin && y0 < ymax) check(y0);
      }
    } else {
      fpa = fpa * (b - a);
      double fl = fa + ymin * (fpa + ymin * (fb - fa - fpa));
      double fh = fa + ymax * (fpa + ymax * (fb - fa - fpa));
      double c = 2 * (fb - fa - fpa);
      y = ymin; fmin = fl;
      if (fh < fmin) { y = ymax; fmin = fh; }
      if (c > 0) {
        double z = -fpa / c;
        if (z > ymin && z < ymax) {
          double fz = fa + z * (fpa + z * (fb - fa - fpa));
          if (fz < fmin) { y = z; fmin = fz; }
        }
      }
    }
    return a + y * (b - a);
  }

  int line_search(double alpha1, double* alpha_new) {
    double f0, fp0l, falpha, falpha_prev, fpalpha = 0, fpalpha_prev, delta, alpha_next;
    double alpha = alpha1, alpha_prev = 0.0, a, b, fa, fb, fpa, fpb;
    int i = 0;
    apply_fdf(0.0, &f0, &fp0l);
    falpha_prev = f0; fpalpha_prev = fp0l;
    a = 0.0; b = alpha; fa = f0; fb = 0.0; fpa = fp0l; fpb = 0.0;
    while (i++ < bracket_iters) {
      falpha = apply_f(alpha);
      if (falpha > f0 + alpha * rho * fp0l || falpha >= falpha_prev) {
        a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
        b = alpha; fb = falpha; fpb = NAN;
        break;
      }
      fpalpha = apply_df(alpha);
      if (fabs(fpalpha) <= -sigma * fp0l) { *alpha_new = alpha; return BFGS_SUCCESS; }
      if (fpalpha >= 0) {
        a = alpha; fa = falpha; fpa = fpalpha;
        b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
        break;
      }
      delta = alpha - alpha_prev;
      alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta, order);
      alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha;
      alpha = alpha_next;
    }
    while (i++ < section_iters) {
      delta = b - a;
      alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order);
      falpha = apply_f(alpha);
      if ((a - alpha) * fpa <= DBL_EPSILON) return BFGS_NOPROGRESS;
      if (falpha > f0 + rho * alpha * fp0l || falpha >= fa) {
        b = alpha; fb = falpha; fpb = NAN;
      } else {
        fpalpha = apply_df(alpha);
        if (fabs(fpalpha) <= -sigma * fp0l) { *alpha_new = alpha; return BFGS_SUCCESS; }
        if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
          b = a; fb = fa; fpb = fpa;
          a = alpha; fa = falpha; fpa = fpalpha;
        } else {
          a = alpha; fa = falpha; fpa = fpalpha;
        }
      }
    }
    return BFGS_SUCCESS;
  }

  void init(CostFn* f_, const double* x) {  // minimizeInit
    fn = f_;
    delta_f = 0;
    fn->eval(x, &f, gradient);
    memcpy(x0, x, sizeof(x0));
    memcpy(g0, gradient, sizeof(g0));
    g0norm = norm(g0);
    for (int i = 0; i < 6; i++) p[i] = gradient[i] * (-1.0 / g0norm);
    pnorm = norm(p);
    fp0 = -g0norm;
    memcpy(x_alpha, x0, sizeof(x0)); x_key = 0;
    f_alpha = f; f_key = 0;
    memcpy(g_alpha, g0, sizeof(g0)); g_key = 0;
    df_alpha = slope(); df_key = 0;
  }

  int one_step(double* x) {  // minimizeOneStep
    double alpha = 0.0, alpha1, f0 = f;
    if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0) return BFGS_NOPROGRESS;
    if (delta_f < 0) {
      double del = fmax(-delta_f, 10 * DBL_EPSILON * fabs(f0));
      alpha1 = fmin(1.0, 2.0 * del / (-fp0));
    } else
      alpha1 = fabs(step_size);
    int status = line_search(alpha1, &alpha);
    if (status != BFGS_SUCCESS) return status;
    {  // updatePosition
      double fa, dfa;
      apply_fdf(alpha, &fa, &dfa);
      f = f_alpha;
      memcpy(x, x_alpha, sizeof(x_alpha));
      memcpy(gradient, g_alpha, sizeof(g_alpha));
    }
    delta_f = f - f0;
    for (int i = 0; i < 6; i++) { dx0[i] = x[i] - x0[i]; dg0[i] = gradient[i] - g0[i]; }
    double dxg = dot(dx0, gradient), dgg = dot(dg0, gradient), dxdg = dot(dx0, dg0), dgnorm = norm(dg0), A, B;
    if (dxdg != 0) {
      B = dxg / dxdg;
      A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
    } else {
      B = 0; A = 0;
    }
    for (int i = 0; i < 6; i++) p[i] = -A * dx0[i];
    for (int i = 0; i < 6; i++) p[i] += gradient[i];
    for (int i = 0; i < 6; i++) p[i] += -B * dg0[i];
    memcpy(g0, gradient, sizeof(g0));
    memcpy(x0, x, sizeof(x0));
    g0norm = norm(g0);
    pnorm = norm(p);
    double dir = (dot(p, gradient) > 0) ? -1.0 : 1.0;
    for (int i = 0; i < 6; i++) p[i] *= dir / pnorm;
    pnorm = norm(p);
    fp0 = dot(p, g0);
    // changeDirection
    memcpy(x_alpha, x0, sizeof(x0)); x_key = 0.0;
    f_key = 0.0;
    memcpy(g_alpha, g0, sizeof(g0)); g_key = 0.0;
    df_alpha = slope(); df_key = 0.0;
    return BFGS_SUCCESS;
  }
};

// estimateRigidTransformationBFGS (gicp.hpp:218-287).  T16 column-major in/out.
// returns 0 ok, -4 too few correspondences, -5 solver failure
inline int estimate_rigid_bfgs(CostFn* fn, int max_inner, float* T16, int* n_inner, double* f_end) {
  auto TM = [&](int r, int c) { return (double)T16[c * 4 + r]; };
  double x[6] = {TM(0, 3), TM(1, 3), TM(2, 3), atan2(TM(2, 1), TM(2, 2)), asin(-TM(2, 0)), atan2(TM(1, 0), TM(0, 0))};
  const double gradient_tol = 1e-2;
  Bfgs b;
  int inner = 0, result;
  b.init(fn, x);
  if (fn->count() < 4) return -4;  // gicp.hpp:225 (the count is known after the first fused pass)
  do {
    inner++;
    result = b.one_step(x);
    if (result) break;
    result = (Bfgs::norm(b.gradient) < gradient_tol) ? BFGS_SUCCESS : BFGS_RUNNING;  // testGradient
  } while (result == BFGS_RUNNING && inner < max_inner);
  *n_inner = inner;
  *f_end = b.f;
  if (result == BFGS_NOPROGRESS || result == BFGS_SUCCESS || inner == max_inner) {
    apply_state(x, T16);  // gicp.hpp:277-278
    return 0;
  }
  return -5;
}

}  // namespace lh
