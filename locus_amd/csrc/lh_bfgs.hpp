// lh_bfgs.hpp -- the solver of the product path: estimateRigidTransformationBFGS (gicp.hpp:218-287) on top of pcl::BFGS
// (PCL 1.10 registration/bfgs.h: Eigen port of GSL vector_bfgs2 + Fletcher line search), and the part of
// computeTransformation's loop body that follows the NN sweep (gicp.hpp:518-568).
//
// Everything here is __host__ __device__ (LH_FN) and templated on
//   Fn : the cost functor of one outer iteration -- eval(x, &f, g) / count() / passes
//   M  : the elementary functions (lh_math.hpp): LibmMath on the host for cost_mode 0 (reference arithmetic, follows the oracle
//        bit for bit), PortableMath for cost_mode 1, where the SAME code runs in k_solve on the GPU and -- for the
//        source-sharded pair, whose sums cross ranks through a host callback -- on the host, with identical bits.
// The functor is served by ONE fused pass per evaluation (the 13 sums that operator(), df and fdf of gicp.hpp:291-402 all
// share), so the wrapper caches the last evaluated x: the reference's "f(alpha) then df(alpha)" pair costs one pass
// instead of two and returns bit-identical numbers for both.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "lh_math.hpp"

namespace lh {

enum { BFGS_RUNNING = -1, BFGS_SUCCESS = 0, BFGS_NOPROGRESS = 1 };
// A source point without ANY nearest neighbour (non-finite query) adds this to its sweep's correspondence count instead of 1
// (lh_kernels.hip nn_index): 2^40, exact in a double next to any real count.  The reference gives the alignment up (gicp.hpp:471-478, 504-506).
constexpr double NO_NN_MARK = 1099511627776.0;

LH_FN bool same_bits6(const double* a, const double* b) {  // memcmp of six doubles
  bool eq = true;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    uint64_t u, v;
    __builtin_memcpy(&u, a + i, 8);
    __builtin_memcpy(&v, b + i, 8);
    eq = eq && (u == v);
  }
  return eq;
}

// ---- the six sine / cosine pairs one state needs ------------------------------------------------------------------------
// applyState (gicp.hpp:619-634) takes the float half angles (Eigen::AngleAxisf -> Quaternionf), computeRDerivative
// (gicp.hpp:160-214) the double angles.  One evaluation of the functor needs both, so they are computed once, together:
// on the host six calls; in k_solve (one wave per pair, every lane holds the same state) six LANES make one call each and the
// results are broadcast -- the same function on the same arguments either way, hence the same bits.
struct Trig {
  double sphi, cphi, sth, cth, spsi, cpsi;  // x[3], x[4], x[5]
  float shx, chx, shy, chy, shz, chz;       // 0.5f * float(x[3]), 0.5f * float(x[4]), 0.5f * float(x[5])
};
#if defined(__HIP_DEVICE_COMPILE__)
// value of lane `src` for the whole wave (v_readlane_b32 x 2; the wave's control flow is uniform in k_solve)
__device__ __forceinline__ double wave_bcast(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}
#endif
template <class M>
LH_FN void trig_all(const double* x, Trig* t) {
  const float hx = 0.5f * (float)x[3], hy = 0.5f * (float)x[4], hz = 0.5f * (float)x[5];
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (std::is_same<M, PortableMath>::value) {
    const int lane = (int)(threadIdx.x & 63u);
    // PortableMath::sincos_f(h) = float(pm_sincos(double(h))): all six are pm_sincos of a double argument, one per lane
    double arg = lane == 0 ? x[3] : (lane == 1 ? x[4] : (lane == 2 ? x[5] : (lane == 3 ? (double)hx : (lane == 4 ? (double)hy : (double)hz))));
    double sn, cs;
    pm_sincos(arg, &sn, &cs);
    t->sphi = wave_bcast(sn, 0); t->cphi = wave_bcast(cs, 0);
    t->sth = wave_bcast(sn, 1); t->cth = wave_bcast(cs, 1);
    t->spsi = wave_bcast(sn, 2); t->cpsi = wave_bcast(cs, 2);
    t->shx = (float)wave_bcast(sn, 3); t->chx = (float)wave_bcast(cs, 3);
    t->shy = (float)wave_bcast(sn, 4); t->chy = (float)wave_bcast(cs, 4);
    t->shz = (float)wave_bcast(sn, 5); t->chz = (float)wave_bcast(cs, 5);
    return;
  }
#endif
  M::sincos_d(x[3], &t->sphi, &t->cphi);
  M::sincos_d(x[4], &t->sth, &t->cth);
  M::sincos_d(x[5], &t->spsi, &t->cpsi);
  M::sincos_f(hx, &t->shx, &t->chx);
  M::sincos_f(hy, &t->shy, &t->chy);
  M::sincos_f(hz, &t->shz, &t->chz);
}

// ---- state <-> matrix -------------------------------------------------------------------------------
// applyState (gicp.hpp:619-634), float quaternion path of Eigen's AngleAxisf products; T = 16 floats column-major
LH_FN void apply_state_trig(const double* x, const Trig& tg, float* T) {
  struct Q { float w, x, y, z; };
  auto mul = [](Q a, Q b) {
    Q r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
  };
  Q qz{tg.chz, 0.f, 0.f, tg.shz}, qy{tg.chy, 0.f, tg.shy, 0.f}, qx{tg.chx, tg.shx, 0.f, 0.f};
  Q q = mul(mul(qz, qy), qx);
  float tx = 2.0f * q.x, ty = 2.0f * q.y, tz = 2.0f * q.z;
  float twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  float txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  float tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  float R[9] = {1.0f - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0f - (txx + tzz), tyz - twx,
                txz - twy, tyz + twx, 1.0f - (txx + tyy)};
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) T[c * 4 + r] = R[r * 3 + c];
  T[3] = T[7] = T[11] = 0.0f;
  T[12] = (float)x[0]; T[13] = (float)x[1]; T[14] = (float)x[2]; T[15] = 1.0f;
}
template <class M>
LH_FN void apply_state(const double* x, float* T) {
  Trig tg;
  trig_all<M>(x, &tg);
  apply_state_trig(x, tg, T);
}
inline void apply_state(const double* x, float* T) { apply_state<LibmMath>(x, T); }  // host, libm (reference build)

// computeRDerivative (gicp.hpp:160-214) + matricesInnerProd (gicp.h:361-370); R row-major
LH_FN void compute_r_derivative_trig(const Trig& tg, const double* R, double* g) {
  const double cphi = tg.cphi, sphi = tg.sphi, cth = tg.cth, sth = tg.sth, cpsi = tg.cpsi, spsi = tg.spsi;
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
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      r3 += dPhi[j * 3 + i] * R[i * 3 + j];
      r4 += dTheta[j * 3 + i] * R[i * 3 + j];
      r5 += dPsi[j * 3 + i] * R[i * 3 + j];
    }
  g[3] = r3; g[4] = r4; g[5] = r5;
}

// f /= m, g_t *= 2/m, R *= 2/m, rotation gradient (gicp.hpp:398-401); S = f, g_t[3], R[9]
LH_FN void cost_finish_trig(const double* S, double m, const Trig& tg, double* f, double* g) {
  *f = S[0] / m;
  double s = 2.0 / m;
  g[0] = S[1] * s; g[1] = S[2] * s; g[2] = S[3] * s;
  double R[9];
#pragma unroll
  for (int i = 0; i < 9; i++) R[i] = S[4 + i] * s;
  compute_r_derivative_trig(tg, R, g);
}
inline void cost_finish(const double* S, double m, const double* x, double* f, double* g) {  // host, libm
  Trig tg;
  trig_all<LibmMath>(x, &tg);
  cost_finish_trig(S, m, tg, f, g);
}

// ---- functor with a one-entry cache ------------------------------------------------------------------
// Pass: void operator()(const double x[6], const Trig& tg, double sums13[13], double* count) -- one fused pass: the 13 sums +
// correspondence count at the state x (tg = its sines / cosines, shared with the gradient's rotation part)
#if defined(__HIP_DEVICE_COMPILE__)
#define LH_EVAL_ATTR __attribute__((noinline))
#else
#define LH_EVAL_ATTR
#endif
struct EvalOut { double f, g[6], m; };
// One evaluation, everything by value.  In k_solve this is the ONE out-of-line function (the line search reaches it from six
// places): arguments and result travel in registers, so the solver's own state (the BFGS vectors, the functor cache) stays
// in registers too instead of living in scratch memory behind a `this` pointer.
template <class Pass, class M>
LH_FN LH_EVAL_ATTR EvalOut eval_state(Pass pass, double x0, double x1, double x2, double x3, double x4, double x5) {
  const double x[6] = {x0, x1, x2, x3, x4, x5};
  double S[13];
  Trig tg;
  trig_all<M>(x, &tg);
  EvalOut o;
  pass(x, tg, S, &o.m);
  if (o.m > 0) cost_finish_trig(S, o.m, tg, &o.f, o.g);
  else {
    o.f = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) o.g[i] = 0.0;
  }
  return o;
}
template <class Pass, class M>
struct CostEval {
  Pass pass;
  bool have = false;
  double cx[6], cf = 0, cg[6], cm = 0;
  int passes = 0;
  LH_FN void eval(const double x[6], double* f, double* g) {
    if (!have || !same_bits6(x, cx)) {
      EvalOut o = eval_state<Pass, M>(pass, x[0], x[1], x[2], x[3], x[4], x[5]);
      passes++;
      cf = o.f; cm = o.m;
#pragma unroll
      for (int i = 0; i < 6; i++) { cx[i] = x[i]; cg[i] = o.g[i]; }
      have = true;
    }
    if (f) *f = cf;
    if (g) {
#pragma unroll
      for (int i = 0; i < 6; i++) g[i] = cg[i];
    }
  }
  LH_FN double count() const { return cm; }
};

// ---- second-order moment model of the cost (cost_mode 1) ------------------------------------------------
// Device side: the fused sweep (lh_kernels.hip).  S = c0, B[3][4], H[6][10], count; T0 = transformation_ at sweep time.
struct MomentModel {
  double S[74];
  float T0[16];    // column-major
  double H12[144]; // expanded symmetric 12x12 form of H, row (r,c) = 4r+c, filled by prepare()
  LH_FN static int sym3(int r, int s) {  // index of (r, s) in (00 01 02 11 12 22)
    int a = r < s ? r : s, b = r < s ? s : r;
    return a == 0 ? b : (a == 1 ? 2 + b : 5);
  }
  LH_FN static int sym4(int c, int e) {  // index of (c, e) in (00 01 02 03 11 12 13 22 23 33)
    int a = c < e ? c : e, b = c < e ? e : c;
    return a == 0 ? b : (a == 1 ? 3 + b : (a == 2 ? 5 + b : 9));
  }
  LH_FN static int h_index(int i, int k) {  // entry of S's H block behind H12[i][k]
    return 13 + sym3(i >> 2, k >> 2) * 10 + sym4(i & 3, k & 3);
  }
  LH_FN double count() const { return S[73]; }
  LH_FN void prepare() {  // once per sweep; the ~600 evaluations of a pair's BFGS solves then cost one 12x12 mat-vec each
    for (int i = 0; i < 12; i++)
      for (int k = 0; k < 12; k++) H12[i * 12 + k] = S[h_index(i, k)];
  }
  // the 13 sums of gicp.hpp:388-396 at T (column-major float matrix from applyState)
  LH_FN void sums(const float* T16, double* sums13) const {
    double d[12];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) d[4 * r + c] = (double)T16[c * 4 + r] - (double)T0[c * 4 + r];
    const double* B = S + 1;
    double G[12], f = S[0];
    for (int i = 0; i < 12; i++) {
      const double* row = H12 + 12 * i;
      double g = 0.0;
#pragma unroll
      for (int k = 0; k < 12; k++) g += row[k] * d[k];  // same summation order (s outer, e inner) as the sym-indexed form
      G[i] = B[i] + g;
    }
#pragma unroll
    for (int k = 0; k < 12; k++) f += d[k] * (B[k] + G[k]);
    sums13[0] = f;
    sums13[1] = G[3]; sums13[2] = G[7]; sums13[3] = G[11];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) sums13[4 + 3 * a + b] = G[4 * b + a];  // R(a,b) = sum p_a (M res)_b
  }
};

// the Pass of cost_mode 1: every evaluation of an outer iteration comes from the 74 moments of its sweep.  base_transformation_
// is the identity here (gicp.hpp:435, 367-368), so T(x) = applyState(x).
// In k_solve (one wave per pair) lane i mod 12 keeps row i of the 12x12 form and B[i] in registers and computes G[i] -- the same
// twelve sequential multiply-adds the host does for that row -- and the twelve results are broadcast: 12 steps instead of 144.
template <class M>
struct MomentPass {
  const MomentModel* mom;
  LH_FN void operator()(const double x[6], const Trig& tg, double sums13[13], double* count) const {
    float T16[16];
    apply_state_trig(x, tg, T16);
#if defined(__HIP_DEVICE_COMPILE__)
    const int i_mine = (int)(threadIdx.x & 63u) % 12;
    const double* row = mom->H12 + 12 * i_mine;
    double d[12];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) d[4 * r + c] = (double)T16[c * 4 + r] - (double)mom->T0[c * 4 + r];
    double g = 0.0;
#pragma unroll
    for (int k = 0; k < 12; k++) g += row[k] * d[k];
    const double Gmine = mom->S[1 + i_mine] + g;
    double G[12], f = mom->S[0];
#pragma unroll
    for (int i = 0; i < 12; i++) G[i] = wave_bcast(Gmine, i);
#pragma unroll
    for (int k = 0; k < 12; k++) f += d[k] * (mom->S[1 + k] + G[k]);
    sums13[0] = f;
    sums13[1] = G[3]; sums13[2] = G[7]; sums13[3] = G[11];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) sums13[4 + 3 * a + b] = G[4 * b + a];
#else
    mom->sums(T16, sums13);
#endif
    *count = mom->count();
  }
};

// ---- pcl::BFGS ----------------------------------------------------------------------------------------
template <class Fn>
struct Bfgs {
  Fn* fn = nullptr;
  double rho = 0.01, sigma = 0.01, tau1 = 9, tau2 = 0.05, tau3 = 0.5, step_size = 1;  // gicp.hpp:253-257
  int order = 3, bracket_iters = 100, section_iters = 100;
  int quad_curv_gt_a = 0;   // lh_gicp_params::bfgs_quad_curv: 1 = `c > a` instead of GSL's `c > 0` in the quadratic interpolation (pcl::BFGS's reported reading)
  double f = 0, delta_f = 0, fp0 = 0, pnorm = 0, g0norm = 0;
  double x0[6], g0[6], dx0[6], dg0[6], p[6], gradient[6];
  double x_alpha[6], g_alpha[6], f_alpha = 0, df_alpha = 0, f_key = 0, df_key = 0, x_key = 0, g_key = 0;

  LH_FN static double dot(const double* a, const double* b) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) s += a[i] * b[i];
    return s;
  }
  LH_FN static double norm(const double* a) { return sqrt(dot(a, a)); }
  LH_FN static void copy6(double* d, const double* s) {
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = s[i];
  }

  LH_FN void moveto(double alpha) {
    if (alpha == x_key) return;
#pragma unroll
    for (int i = 0; i < 6; i++) x_alpha[i] = x0[i] + alpha * p[i];
    x_key = alpha;
  }
  LH_FN double slope() const { return dot(g_alpha, p); }
  LH_FN double apply_f(double alpha) {
    if (alpha == f_key) return f_alpha;
    moveto(alpha);
    fn->eval(x_alpha, &f_alpha, nullptr);
    f_key = alpha;
    return f_alpha;
  }
  LH_FN double apply_df(double alpha) {
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
  LH_FN void apply_fdf(double alpha, double* fo, double* dfo) {
    if (alpha == f_key && alpha == df_key) { *fo = f_alpha; *dfo = df_alpha; return; }
    if (alpha == f_key || alpha == df_key) { *fo = apply_f(alpha); *dfo = apply_df(alpha); return; }
    moveto(alpha);
    fn->eval(x_alpha, &f_alpha, g_alpha);
    f_key = alpha; g_key = alpha;
    df_alpha = slope();
    df_key = alpha;
    *fo = f_alpha; *dfo = df_alpha;
  }

  LH_FN static double cubic(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }

  LH_FN static double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax, int order, int c_gt_a) {
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
      if (c > (c_gt_a ? a : 0.0)) {
        double z = -fpa / c;
        if (z > ymin && z < ymax) {
          double fz = fa + z * (fpa + z * (fb - fa - fpa));
          if (fz < fmin) { y = z; fmin = fz; }
        }
      }
    }
    return a + y * (b - a);
  }

  LH_FN int line_search(double alpha1, double* alpha_new) {
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
      alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta, alpha + tau1 * delta, order, quad_curv_gt_a);
      alpha_prev = alpha; falpha_prev = falpha; fpalpha_prev = fpalpha;
      alpha = alpha_next;
    }
    while (i++ < section_iters) {
      delta = b - a;
      alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order, quad_curv_gt_a);
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

  LH_FN void init(Fn* f_, const double* x) {  // minimizeInit
    fn = f_;
    delta_f = 0;
    fn->eval(x, &f, gradient);
    copy6(x0, x);
    copy6(g0, gradient);
    g0norm = norm(g0);
#pragma unroll
    for (int i = 0; i < 6; i++) p[i] = gradient[i] * (-1.0 / g0norm);
    pnorm = norm(p);
    fp0 = -g0norm;
    copy6(x_alpha, x0); x_key = 0;
    f_alpha = f; f_key = 0;
    copy6(g_alpha, g0); g_key = 0;
    df_alpha = slope(); df_key = 0;
  }

  LH_FN int one_step(double* x) {  // minimizeOneStep
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
      copy6(x, x_alpha);
      copy6(gradient, g_alpha);
    }
    delta_f = f - f0;
#pragma unroll
    for (int i = 0; i < 6; i++) { dx0[i] = x[i] - x0[i]; dg0[i] = gradient[i] - g0[i]; }
    double dxg = dot(dx0, gradient), dgg = dot(dg0, gradient), dxdg = dot(dx0, dg0), dgnorm = norm(dg0), A, B;
    if (dxdg != 0) {
      B = dxg / dxdg;
      A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
    } else {
      B = 0; A = 0;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) p[i] = -A * dx0[i];
#pragma unroll
    for (int i = 0; i < 6; i++) p[i] += gradient[i];
#pragma unroll
    for (int i = 0; i < 6; i++) p[i] += -B * dg0[i];
    copy6(g0, gradient);
    copy6(x0, x);
    g0norm = norm(g0);
    pnorm = norm(p);
    double dir = (dot(p, gradient) > 0) ? -1.0 : 1.0;
#pragma unroll
    for (int i = 0; i < 6; i++) p[i] *= dir / pnorm;
    pnorm = norm(p);
    fp0 = dot(p, g0);
    // changeDirection
    copy6(x_alpha, x0); x_key = 0.0;
    f_key = 0.0;
    copy6(g_alpha, g0); g_key = 0.0;
    df_alpha = slope(); df_key = 0.0;
    return BFGS_SUCCESS;
  }
};

// estimateRigidTransformationBFGS (gicp.hpp:218-287).  T16 column-major in/out.
// returns 0 ok, -4 too few correspondences, -5 solver failure, -6 a source point without a nearest neighbour
template <class Fn, class M>
LH_FN int estimate_rigid_bfgs(Fn* fn, int max_inner, float* T16, int* n_inner, double* f_end, int bfgs_quad_curv = 0) {
  auto TM = [&](int r, int c) { return (double)T16[c * 4 + r]; };
  double x[6] = {TM(0, 3), TM(1, 3), TM(2, 3), M::atan2_d(TM(2, 1), TM(2, 2)), M::asin_d(-TM(2, 0)), M::atan2_d(TM(1, 0), TM(0, 0))};
  const double gradient_tol = 1e-2;
  Bfgs<Fn> b;
  b.quad_curv_gt_a = bfgs_quad_curv;
  int inner = 0, result;
  b.init(fn, x);
  if (fn->count() >= NO_NN_MARK) return -6;  // `failure` of the NN loop (gicp.hpp:471-478): computeTransformation returns before the solve (:504-506)
  if (fn->count() < 4) return -4;  // gicp.hpp:225 (the count is known after the first fused pass)
  do {
    inner++;
    double x_in[6], p_in[6];
    Bfgs<Fn>::copy6(x_in, x);
    Bfgs<Fn>::copy6(p_in, b.p);
    const double f_in = b.f, df_in = b.delta_f;
    const int passes_in = fn->passes;
    result = b.one_step(x);
    if (result) break;
    result = (Bfgs<Fn>::norm(b.gradient) < gradient_tol) ? BFGS_SUCCESS : BFGS_RUNNING;  // testGradient
    // A step that ends in the state it started from -- position, direction, f and delta_f bit for bit: the line search used up its 100
    // iterations on a cost that is flat to the last bit and left alpha = 0 (Fletcher's sectioning returns `success` then, as GSL's does) -- is
    // a fixed point of minimizeOneStep: every further inner iteration would replay it evaluation for evaluation, to the same end, until
    // max_inner_iterations.  (Pairs whose outer loop is forced on after they have converged do this: 20 x 101 evaluations per outer
    // iteration, 6 ms of one wave in k_solve with the 31 other pairs of its group waiting.)  The replays are skipped; their evaluations are
    // still counted, so `cost_passes` stays the number the algorithm specifies, and x, f, the inner count and every later decision are
    // exactly what the replays would have left.
    if (result == BFGS_RUNNING && inner < max_inner && b.f == f_in && b.delta_f == df_in) {
      bool same = true;
#pragma unroll
      for (int i = 0; i < 6; i++) same = same && x[i] == x_in[i] && b.p[i] == p_in[i];
      if (same) {
        fn->passes += (max_inner - inner) * (fn->passes - passes_in);
        inner = max_inner;
        break;
      }
    }
  } while (result == BFGS_RUNNING && inner < max_inner);
  *n_inner = inner;
  *f_end = b.f;
  if (result == BFGS_NOPROGRESS || result == BFGS_SUCCESS || inner == max_inner) {
    apply_state<M>(x, T16);  // gicp.hpp:277-278
    return 0;
  }
  return -5;
}

// ---- one outer iteration after its NN sweep (gicp.hpp:518-568) -------------------------------------------------------
struct OuterParams {
  int max_iterations, max_inner_iterations;
  double rotation_epsilon, transformation_epsilon;
  int bfgs_quad_curv = 0;   // lh_gicp_params::bfgs_quad_curv
};
struct OuterState {
  float T[16];      // transformation_ (column-major): where the next sweep transforms the source to
  float prev[16];   // previous_transformation_: what final_transformation_ is composed from (gicp.hpp:583)
  int iter;         // nr_iterations_
  int done;         // the loop has ended: converged_, or an exception was caught (gicp.hpp:542-547)
  int converged, status;       // status: 0, -4 (NotEnoughPointsException), -5 (SolverDidntConvergeException) or -6 (a query without a neighbour)
  int n_corr_last, passes, n_inner, pad;
  double f_end, delta;
  double corr_sum;  // correspondences summed over the iterations (instrumentation: the algorithmic bytes of SURVEY 8d scale with it)
  // T in the form the late sweeps (k_late) use it, made by whoever sets T (pose_of_transform below): eighteen doubles every wave would
  // otherwise re-derive from the floats with vector instructions, read there with scalar loads into SGPRs
  double poseT[12];  // double(T), row-major 3x4
  double poseG[6];   // R R^T + I of its rotation block: 00 01 02 11 12 22
};
// (explicit fused multiply-adds: the sweep kernels derive the same values from the same floats and must get the same bits)
LH_FN void pose_of_transform(const float* T12, double* poseT, double* poseG) {   // T12: row-major 3x4
#pragma unroll
  for (int k = 0; k < 12; k++) poseT[k] = (double)T12[k];
  int q = 0;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = r; c < 3; c++) {
      const double rr = __builtin_fma(poseT[r * 4 + 2], poseT[c * 4 + 2], __builtin_fma(poseT[r * 4 + 1], poseT[c * 4 + 1], poseT[r * 4 + 0] * poseT[c * 4 + 0]));
      poseG[q++] = r == c ? rr + 1.0 : rr;
    }
}
LH_FN void outer_state_init(OuterState* s) {
#pragma unroll
  for (int i = 0; i < 16; i++) { s->T[i] = (i % 5 == 0) ? 1.0f : 0.0f; s->prev[i] = s->T[i]; }  // align() resets transformation_ to identity
  s->iter = 0; s->done = 0; s->converged = 0; s->status = 0; s->n_corr_last = 0; s->passes = 0; s->n_inner = 0; s->pad = 0;
  s->f_end = 0.0; s->delta = 0.0; s->corr_sum = 0.0;
  const float I12[12] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
  pose_of_transform(I12, s->poseT, s->poseG);
}
// fn = the functor of THIS iteration's correspondences (a fresh cache)
template <class Fn, class M>
LH_FN void outer_step(Fn* fn, const OuterParams& P, OuterState* s) {
#pragma unroll
  for (int i = 0; i < 16; i++) s->prev[i] = s->T[i];  // previous_transformation_ = transformation_ (gicp.hpp:518)
  const int before = fn->passes;
  int n_inner = 0;
  double f_end = 0.0;
  int st = estimate_rigid_bfgs<Fn, M>(fn, P.max_inner_iterations, s->T, &n_inner, &f_end, P.bfgs_quad_curv);
  s->n_corr_last = st == -6 ? 0 : (int)fn->count();
  s->passes += fn->passes - before;
  s->n_inner = n_inner;
  s->f_end = f_end;
  if (st != 0) {  // exception caught -> break (gicp.hpp:542-547): final_transformation_ comes from previous_transformation_
                  // (st == -6: the plain `return` of gicp.hpp:504-506 -- final_transformation_ stays what align() reset it to)
    s->status = st;
    s->done = 1;
    return;
  }
  double delta = 0.0;  // gicp.hpp:526-541
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int l = 0; l < 4; l++) {
      double ratio = (k < 3 && l < 3) ? 1. / P.rotation_epsilon : 1. / P.transformation_epsilon;
      double c_delta = ratio * (double)fabsf(s->prev[l * 4 + k] - s->T[l * 4 + k]);  // two Matrix4f entries: a FLOAT subtraction (gicp.hpp:535-536)
      if (c_delta > delta) delta = c_delta;
    }
  s->delta = delta;
  s->iter++;
  if (s->iter >= P.max_iterations || delta < 1) {  // gicp.hpp:566
    s->converged = 1;
    s->done = 1;
#pragma unroll
    for (int i = 0; i < 16; i++) s->prev[i] = s->T[i];
  }
}

}  // namespace lh
