/* ccmp_solve.h — minimum-norm solve of the 2x14 Newton system, device side.
 * Replaces Eigen::JacobiSVD<MatrixXd>(j, ComputeThinU|ComputeThinV).solve(f) at
 * include/closed_chain_motion_planner/base/constraints/ConstraintFunction.h:71. */
#ifndef CCMP_SOLVE_H
#define CCMP_SOLVE_H
#include "ccmp_detmath.h"

namespace ccmp {

// Eigen JacobiSVD(2x14).solve(f) restated as a two-sweep one-sided Jacobi on the two rows
// (oracle/ccmp_oracle.c: orc_solve_minnorm — same operations, same order).
__device__ __forceinline__ void solve_minnorm(const double *J /*28, registers*/, double f0, double f1, double *dx)
{
  double r0[14], r1[14], g0 = f0, g1 = f1;
#pragma unroll
  for (int j = 0; j < 14; j++) { r0[j] = J[j]; r1[j] = J[14 + j]; }
  double a = 0, d = 0, b = 0;
#pragma unroll
  for (int sweep = 0; sweep < 2; sweep++) {
    a = 0; d = 0; b = 0;
#pragma unroll
    for (int j = 0; j < 14; j++) {
      a = CCMP_FMA(r0[j], r0[j], a);
      d = CCMP_FMA(r1[j], r1[j], d);
      b = CCMP_FMA(r0[j], r1[j], b);
    }
    if (b != 0.0) {
      double zeta = (d - a) / (2.0 * b);
      double t = 1.0 / (ccmp_abs(zeta) + ccmp_sqrt(CCMP_FMA(zeta, zeta, 1.0)));
      if (zeta < 0.0) t = -t;
      double c = 1.0 / ccmp_sqrt(CCMP_FMA(t, t, 1.0));
      double s = c * t;
#pragma unroll
      for (int j = 0; j < 14; j++) {
        double v0 = r0[j], v1 = r1[j];
        r0[j] = CCMP_FMA(c, v0, -(s * v1));
        r1[j] = CCMP_FMA(s, v0, c * v1);
      }
      double h0 = g0, h1 = g1;
      g0 = CCMP_FMA(c, h0, -(s * h1));
      g1 = CCMP_FMA(s, h0, c * h1);
    }
  }
  a = 0; d = 0;
#pragma unroll
  for (int j = 0; j < 14; j++) { a = CCMP_FMA(r0[j], r0[j], a); d = CCMP_FMA(r1[j], r1[j], d); }
  double s0 = ccmp_sqrt(a), s1 = ccmp_sqrt(d);
  double smax = s0 > s1 ? s0 : s1;
  double thr = smax * (2.0 * 2.220446049250313e-16);
  if (thr < 2.2250738585072014e-308) thr = 2.2250738585072014e-308;
  double k0 = s0 > thr ? g0 / a : 0.0;
  double k1 = s1 > thr ? g1 / d : 0.0;
#pragma unroll
  for (int j = 0; j < 14; j++) dx[j] = CCMP_FMA(k1, r1[j], k0 * r0[j]);
}

// The analytic mode's step (oracle/ccmp_oracle.c: orc_solve_gram — same operations, same order): dx = J^T (J J^T)^-1 f
// through the 2x2 Gram matrix [[a, b], [b, d]] in closed form: dx_j = y1 J1_j + y0 J0_j.  `false` where the two rows are
// nearly parallel (det / (a d) = sin^2 of their angle <= 2^-20) or something is NaN: solve_minnorm takes over there.
__device__ __forceinline__ bool gram_coeffs(double a, double d, double b, double f0, double f1, double &y0, double &y1)
{
  const double det = CCMP_FMA(a, d, -(b * b));
  const bool well = det > (a * d) * 9.5367431640625e-07; // 2^-20
  const double inv = 1.0 / det;
  y0 = CCMP_FMA(d, f0, -(b * f1)) * inv;
  y1 = CCMP_FMA(a, f1, -(b * f0)) * inv;
  return well;
}
// the three Gram sums of a full 2x14 Jacobian held by one lane: each arm's seven columns summed from zero, the two partial
// sums added (ccmp_kernels_fast.hip's lane-pair kernel holds one arm per lane)
__device__ __forceinline__ void gram_sums(const double *J /*28*/, double &a, double &d, double &b)
{
  double pa[2], pd[2], pb[2];
#pragma unroll
  for (int arm = 0; arm < 2; arm++) {
    double sa = 0.0, sd = 0.0, sb = 0.0;
#pragma unroll
    for (int j = 7 * arm; j < 7 * arm + 7; j++) {
      sa = CCMP_FMA(J[j], J[j], sa);
      sd = CCMP_FMA(J[14 + j], J[14 + j], sd);
      sb = CCMP_FMA(J[j], J[14 + j], sb);
    }
    pa[arm] = sa; pd[arm] = sd; pb[arm] = sb;
  }
  a = pa[0] + pa[1]; d = pd[0] + pd[1]; b = pb[0] + pb[1];
}

} /* namespace ccmp */
#endif
