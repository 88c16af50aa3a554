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


} /* namespace ccmp */
#endif
