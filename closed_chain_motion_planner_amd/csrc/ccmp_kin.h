/* ccmp_kin.h — dual-Panda kinematics and the closed-chain residual, host+device, in the project's
 * canonical rounding model (ccmp_detmath.h): every a*b+c that is fused is written CCMP_FMA, sums
 * run left to right, and the translation units that must agree bitwise are built with
 * -ffp-contract=off.  The same functions compute init_chain_ on the host at set-up and run inside
 * the gfx950 kernels.
 *
 * Reference behaviour restated here (paths relative to the reference checkout):
 *   PandaModel::getTransform / getRotation / getTranslation   src/kinematics/panda_rbdl.cpp:24-42
 *   PandaModel::initModel (modified-DH tables -> axes/offsets) src/kinematics/panda_rbdl.cpp:73-160
 *   KinematicChainConstraint::function                         include/.../constraints/ConstraintFunction.h:84-102
 * and the Eigen / RBDL arithmetic those call.
 */
#ifndef CCMP_KIN_H
#define CCMP_KIN_H
#include <stdint.h>

#include "ccmp_detmath.h"

/* Kernel-side constants derived from ccmp_problem (ccmp_problem.cpp: make_consts).  Passed by value
 * as a kernel argument: every lane reads the same entry at the same time, so the compiler keeps
 * them in SGPRs (scalar loads from the kernarg segment), never in VGPRs. */
struct ccmp_consts {
  double axis[2][7][3];
  double aprod[2][7][6]; /* a0a0 a0a1 a0a2 a1a1 a1a2 a2a2 of each axis */
  double offset[2][7][3];
  double ee[2][3];
  double R_tool[2][9];
  double base_R[2][9];
  double base_p[2][3];
  double init_p[3];
  double init_q[4]; /* Quaterniond(init_chain_.linear()) as (x,y,z,w) */
  double lbe[7];    /* lb + eps */
  double ube[7];    /* ub - eps */
  double lb[7];
  double ub[7];
  double span[7];   /* ub - lb (sampler) */
  double t_o7i_R[9]; /* inverse of arm 0's t_o7 (IKTask::compute_t_wo, src/base/constraints/ik_task.cpp:10-14) */
  double t_o7i_p[3];
  double tol_pos, tol_rot, step;
  int32_t max_iter;
  int32_t base_diag; /* bit a: base_R[a] is exactly diag(+-1, +-1, +-1) (every shipped t_wb, grasping_point.cpp:11-20) */
  int32_t stock;     /* both arms carry the stock Panda structure (see kStockZ below): launchers pick the STOCK kernels */
  int32_t twin_arms; /* stock, both arms have bit-identical chain constants (axis, offset, ee, R_tool) and both t_wb are
                        diag(+-1): the throughput kernel's STOCK instantiation reads arm 0's constants for either arm */
};

namespace ccmp {

/* a0*b0 + a1*b1 + a2*b2, accumulated left to right */
CCMP_HD double dot3(double a0, double b0, double a1, double b1, double a2, double b2)
{
  return CCMP_FMA(a2, b2, CCMP_FMA(a1, b1, a0 * b0));
}
/* c + a0*b0 + a1*b1 + a2*b2, accumulated left to right onto c */
CCMP_HD double dot3acc(double c, double a0, double b0, double a1, double b1, double a2, double b2)
{
  return CCMP_FMA(a2, b2, CCMP_FMA(a1, b1, CCMP_FMA(a0, b0, c)));
}

/* C = A B (row-major 3x3) */
CCMP_HD void mul33(const double *A, const double *B, double *C)
{
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = dot3(A[3 * i], B[j], A[3 * i + 1], B[3 + j], A[3 * i + 2], B[6 + j]);
}
/* C = A^T B */
CCMP_HD void mulT33(const double *A, const double *B, double *C)
{
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = dot3(A[i], B[j], A[3 + i], B[3 + j], A[6 + i], B[6 + j]);
}
/* r += A v */
CCMP_HD void mulvec_acc(const double *A, const double *v, double *r)
{
#pragma unroll
  for (int i = 0; i < 3; i++) r[i] = dot3acc(r[i], A[3 * i], v[0], A[3 * i + 1], v[1], A[3 * i + 2], v[2]);
}
/* r = A^T v */
CCMP_HD void mulTvec(const double *A, const double *v, double *r)
{
#pragma unroll
  for (int i = 0; i < 3; i++) r[i] = dot3(A[i], v[0], A[3 + i], v[1], A[6 + i], v[2]);
}

/* Rotation by angle with sine s, cosine c about unit axis a; ap = {a0a0,a0a1,a0a2,a1a1,a1a2,a2a2}.
 * RBDL Xrot(angle, axis) transposed (RBDL keeps world->body; panda_rbdl.cpp:32 transposes back). */
CCMP_HD void rot_sc(const double *a, const double *ap, double s, double c, double *R)
{
  double t = 1.0 - c;
  double a0s = a[0] * s, a1s = a[1] * s, a2s = a[2] * s;
  R[0] = CCMP_FMA(ap[0], t, c);
  R[1] = CCMP_FMA(ap[1], t, -a2s);
  R[2] = CCMP_FMA(ap[2], t, a1s);
  R[3] = CCMP_FMA(ap[1], t, a2s);
  R[4] = CCMP_FMA(ap[3], t, c);
  R[5] = CCMP_FMA(ap[4], t, -a0s);
  R[6] = CCMP_FMA(ap[2], t, -a1s);
  R[7] = CCMP_FMA(ap[4], t, a0s);
  R[8] = CCMP_FMA(ap[5], t, c);
}

/* ---- structure of the stock Panda constants -------------------------------------------------------------------
 * With zero calibration offsets (the shipped configuration: PandaModel::initModel(dh) is commented out at
 * ConstrainedPlanningCommon.cpp:97) the constants of panda_rbdl.cpp:97-147 contain exact zeros and ones: joints 1, 3
 * and 5 rotate about exactly (0, 0, 1); of the 21 joint-offset components only six are non-zero; the hand offset has
 * no x component.  A product with an exact zero is an exact zero, and adding it changes no bit (beyond the sign of a
 * sum that is itself exactly zero), so kernels instantiated with STOCK = true skip those operations and still
 * reproduce the general formulas bit for bit.  The host selects STOCK only after comparing the constants exactly
 * (ccmp_problem.cpp: is_stock_structure); anything else — calibrated arms — runs the general code. */
constexpr int kStockZ[7] = {1, 0, 1, 0, 1, 0, 0};   /* axis == (0, 0, 1) exactly */
constexpr int kStockOff[7] = {4, 0, 4, 1, 5, 0, 1}; /* bit k: offset component k may be non-zero */
constexpr int kStockEe = 6;                         /* ee = (0, y, z) */

/* r += A v where the components of v not flagged in NZ are exactly zero: the skipped terms of mulvec_acc are exact
 * zeros added to the running sum, the others keep their order. */
template <int NZ>
CCMP_HD void mulvec_acc_nz(const double *A, const double *v, double *r)
{
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double a = r[i];
    if (NZ & 1) a = CCMP_FMA(A[3 * i], v[0], a);
    if (NZ & 2) a = CCMP_FMA(A[3 * i + 1], v[1], a);
    if (NZ & 4) a = CCMP_FMA(A[3 * i + 2], v[2], a);
    r[i] = a;
  }
}

/* Rn = R * Rot((0, 0, 1), angle): rot_sc gives [[c, -s, 0], [s, c, 0], [0, 0, (1 - c) + c]] (the last entry is NOT 1:
 * it is rounded twice, as in the general formula) and mul33's remaining terms are products with those exact zeros. */
CCMP_HD void mul_zrot(const double *R, double s, double c, double *Rn)
{
  const double t = 1.0 - c;
  const double w = t + c;
  const double ns = -s;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    Rn[3 * r] = CCMP_FMA(R[3 * r + 1], s, R[3 * r] * c);
    Rn[3 * r + 1] = CCMP_FMA(R[3 * r + 1], c, R[3 * r] * ns);
    Rn[3 * r + 2] = R[3 * r + 2] * w;
  }
}

/* Rn = R * Rot(axis_I, q_I) with the joint index known at compile time; ax / ap are the joint's axis and axis
 * products (not read for a stock z joint) */
template <int I, bool STOCK>
CCMP_HD void chain_rot(const double *ax, const double *ap, double s, double c, const double *R, double *Rn)
{
  if (STOCK && kStockZ[I]) {
    mul_zrot(R, s, c, Rn);
  } else {
    double Rj[9];
    rot_sc(ax, ap, s, c, Rj);
    mul33(R, Rj, Rn);
  }
}
/* joint I of the chain: o += R*offset_I, then Rn = R*Rot(axis_I, q_I) */
template <int I, bool STOCK>
CCMP_HD void chain_step(const double *off, const double *ax, const double *ap, double s, double c, const double *R, double *Rn,
                        double *o)
{
  mulvec_acc_nz<STOCK ? kStockOff[I] : 7>(R, off, o);
  chain_rot<I, STOCK>(ax, ap, s, c, R, Rn);
}

/* One joint of the chain: o += R*offset_i (joint origin), then R = R*Rot(axis_i, q_i). */
CCMP_HD void joint_step(const ccmp_consts &K, int arm, int i, double s, double c, double *R, double *o)
{
  double Rj[9], Rn[9];
  mulvec_acc(R, K.offset[arm][i], o);
  rot_sc(K.axis[arm][i], K.aprod[arm][i], s, c, Rj);
  mul33(R, Rj, Rn);
#pragma unroll
  for (int k = 0; k < 9; k++) R[k] = Rn[k];
}

/* Body-7 frame (R,o) -> world pose of the hand frame: getTranslation/getRotation
 * (panda_rbdl.cpp:24-33) then t_wb * (ConstraintFunction.h:89-90). */
template <bool STOCK>
CCMP_HD void tool_pose_t(const ccmp_consts &K, int arm, const double *R, const double *o, double *Rw, double *pw)
{
  double pf[3] = {o[0], o[1], o[2]}, Rf[9];
  mulvec_acc_nz<STOCK ? kStockEe : 7>(R, K.ee[arm], pf);
  mul33(R, K.R_tool[arm], Rf);
#ifndef CCMP_NO_BASE_DIAG
  if ((K.base_diag >> arm) & 1) {
    /* t_wb.linear() = diag(+-1): the general product below adds exact zeros to d_r * Rf[r][c] and to
     * fma(d_r, pf[r], base_p[r]) — the same bits for a third of the operations */
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const double d = K.base_R[arm][4 * r];
#pragma unroll
      for (int c = 0; c < 3; c++) Rw[3 * r + c] = d * Rf[3 * r + c];
      pw[r] = CCMP_FMA(d, pf[r], K.base_p[arm][r]);
    }
    return;
  }
#endif
  mul33(K.base_R[arm], Rf, Rw);
  pw[0] = K.base_p[arm][0]; pw[1] = K.base_p[arm][1]; pw[2] = K.base_p[arm][2];
  mulvec_acc(K.base_R[arm], pf, pw);
}

CCMP_HD void tool_pose(const ccmp_consts &K, int arm, const double *R, const double *o, double *Rw, double *pw)
{
  tool_pose_t<false>(K, arm, R, o, Rw, pw);
}

/* Full FK of one arm (world pose of its hand frame). */
CCMP_HD void fk_arm(const ccmp_consts &K, int arm, const double *q, double *Rw, double *pw)
{
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
#pragma unroll
  for (int i = 0; i < 7; i++) { /* unrolled: q[] then stays in registers in the per-lane kernels (no scratch) */
    double s, c;
    ccmp_sincos(q[i], &s, &c);
    joint_step(K, arm, i, s, c, R, o);
  }
  tool_pose(K, arm, R, o, Rw, pw);
}

#ifndef CCMP_QUAT_BRANCHY
/* Eigen Quaterniond(Matrix3d) — trace / major-diagonal branches; out (x,y,z,w) — in branch-free form:
 * the four branches of Eigen's conversion differ only in WHICH sum feeds the one square root and which
 * differences/sums are scaled by 0.5/sqrt, so the operands are selected and sqrt/divide run once.
 * Every output is produced by the same operation on the same operands as in the branchy form below
 * (bit-identical; in-process A/B on MI355X, tools/ab.py: 1.4 % faster than the branchy form). */
CCMP_HD void quat_of(const double *m, double *q)
{
  const double tr = (m[0] + m[4]) + m[8];
  int i = (m[4] > m[0]) ? 1 : 0;
  const double mii = i ? m[4] : m[0];
  if (m[8] > mii) i = 2;
  const int kase = (tr > 0.0) ? 3 : i;
  const double a3 = tr + 1.0;
  const double a0 = ((m[0] - m[4]) - m[8]) + 1.0;
  const double a1 = ((m[4] - m[8]) - m[0]) + 1.0;
  const double a2 = ((m[8] - m[0]) - m[4]) + 1.0;
  const double arg = kase == 3 ? a3 : (kase == 0 ? a0 : (kase == 1 ? a1 : a2));
  const double t = ccmp_sqrt(arg);
  const double h = 0.5 * t;
  const double k = 0.5 / t;
  const double d1 = (m[7] - m[5]) * k, d2 = (m[2] - m[6]) * k, d3 = (m[3] - m[1]) * k;
  const double s1 = (m[3] + m[1]) * k, s2 = (m[6] + m[2]) * k, s3 = (m[7] + m[5]) * k;
  q[0] = kase == 3 ? d1 : (kase == 0 ? h : (kase == 1 ? s1 : s2));
  q[1] = kase == 3 ? d2 : (kase == 0 ? s1 : (kase == 1 ? h : s3));
  q[2] = kase == 3 ? d3 : (kase == 0 ? s2 : (kase == 1 ? s3 : h));
  q[3] = kase == 3 ? h : (kase == 0 ? d1 : (kase == 1 ? d2 : d3));
}
#else
/* Eigen Quaterniond(Matrix3d) — trace / major-diagonal branches; out (x,y,z,w). */
CCMP_HD void quat_of(const double *m, double *q)
{
  double t = (m[0] + m[4]) + m[8];
  if (t > 0.0) {
    t = ccmp_sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = (m[4] > m[0]) ? 1 : 0;
    double mii = i ? m[4] : m[0];
    if (m[8] > mii) i = 2;
    if (i == 0) { /* j=1, k=2 */
      t = ccmp_sqrt(((m[0] - m[4]) - m[8]) + 1.0);
      q[0] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[7] - m[5]) * t;
      q[1] = (m[3] + m[1]) * t;
      q[2] = (m[6] + m[2]) * t;
    } else if (i == 1) { /* j=2, k=0 */
      t = ccmp_sqrt(((m[4] - m[8]) - m[0]) + 1.0);
      q[1] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[2] - m[6]) * t;
      q[2] = (m[7] + m[5]) * t;
      q[0] = (m[1] + m[3]) * t;
    } else { /* i=2, j=0, k=1 */
      t = ccmp_sqrt(((m[8] - m[0]) - m[4]) + 1.0);
      q[2] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[3] - m[1]) * t;
      q[0] = (m[2] + m[6]) * t;
      q[1] = (m[5] + m[7]) * t;
    }
  }
}

#endif

/* current_chain = t_w72.inverse() * t_w71, then (|dp|, angularDistance) against init_chain_
 * (ConstraintFunction.h:92-101).  dq (nullable) receives q_c * conj(q_0) as (x,y,z,w) and pc
 * (nullable) the chain translation — the analytic Jacobian needs both. */
CCMP_HD void chain_residual(const ccmp_consts &K, const double *R1, const double *p1, const double *R2,
                            const double *p2, double *f, double *dq, double *pc_out)
{
  double Rc[9], ti[3], pc[3], qc[4];
  mulT33(R2, R1, Rc);
  mulTvec(R2, p2, ti);
  mulTvec(R2, p1, pc);
#pragma unroll
  for (int k = 0; k < 3; k++) pc[k] = pc[k] + (-ti[k]);
  quat_of(Rc, qc);
  double ax = qc[0], ay = qc[1], az = qc[2], aw = qc[3];
  double bx = -K.init_q[0], by = -K.init_q[1], bz = -K.init_q[2], bw = K.init_q[3];
  double dw = CCMP_FMA(-az, bz, CCMP_FMA(-ay, by, CCMP_FMA(-ax, bx, aw * bw)));
  double dx = CCMP_FMA(-az, by, CCMP_FMA(ay, bz, CCMP_FMA(ax, bw, aw * bx)));
  double dy = CCMP_FMA(-ax, bz, CCMP_FMA(az, bx, CCMP_FMA(ay, bw, aw * by)));
  double dz = CCMP_FMA(-ay, bx, CCMP_FMA(ax, by, CCMP_FMA(az, bw, aw * bz)));
  double vn = ccmp_sqrt(dot3(dx, dx, dy, dy, dz, dz));
  f[1] = 2.0 * ccmp_atan2_nn(vn, ccmp_abs(dw));
  double e0 = pc[0] - K.init_p[0], e1 = pc[1] - K.init_p[1], e2 = pc[2] - K.init_p[2];
  f[0] = ccmp_sqrt(dot3(e0, e0, e1, e1, e2, e2));
  if (dq) { dq[0] = dx; dq[1] = dy; dq[2] = dz; dq[3] = dw; }
  if (pc_out) { pc_out[0] = pc[0]; pc_out[1] = pc[1]; pc_out[2] = pc[2]; }
}

/* KinematicChainConstraint::function for one state. */
CCMP_HD void residual(const ccmp_consts &K, const double *x, double *f)
{
  double R1[9], p1[3], R2[9], p2[3];
  fk_arm(K, 0, x, R1, p1);
  fk_arm(K, 1, x + 7, R2, p2);
  chain_residual(K, R1, p1, R2, p2, f, nullptr, nullptr);
}

/* KinematicChainSpace::enforceBounds for one value (KinematicChain.h:118-130). */
CCMP_HD double wrap_pi(double q)
{
  const double pi = 3.14159265358979323846;
  double v = __builtin_fmod(q, 2.0 * pi);
  if (v < -pi) v += 2.0 * pi;
  else if (v >= pi) v -= 2.0 * pi;
  return v;
}

CCMP_HD uint64_t splitmix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
/* dimension j of ambient sample `index`: RNG::uniformReal(low, high) = (high-low)*u + low */
CCMP_HD double ambient_uniform_at(const ccmp_consts &K, uint64_t seed, uint64_t index, int j, int joint /* = j % 7 */)
{
  uint64_t r = splitmix64(seed ^ (index * 14ULL + (uint64_t)j));
  double u = (double)(r >> 11) * 1.1102230246251565e-16; /* 2^-53 */
  return CCMP_FMA(K.span[joint], u, K.lb[joint]);
}
CCMP_HD double ambient_uniform(const ccmp_consts &K, uint64_t seed, uint64_t index, int j)
{
  return ambient_uniform_at(K, seed, index, j, j % 7);
}

/* RealVectorStateSampler::sampleUniformNear: uniformReal(max(low, near-d), min(high, near+d)) */
CCMP_HD double ambient_near(const ccmp_consts &K, uint64_t seed, uint64_t index, int j, double near, double d)
{
  const double low = K.lb[j % 7], high = K.ub[j % 7];
  const double a = (near - d) > low ? (near - d) : low;
  const double b = (near + d) < high ? (near + d) : high;
  uint64_t r = splitmix64(seed ^ (index * 14ULL + (uint64_t)j));
  double u = (double)(r >> 11) * 1.1102230246251565e-16;
  return CCMP_FMA(b - a, u, a);
}
/* RealVectorStateSampler::sampleGaussian: mean + stdDev * N(0,1), clamped to the bounds.  OMPL's
 * normal deviate is library-defined; here it is Box-Muller on two counter-based uniforms. */
CCMP_HD double ambient_gaussian(const ccmp_consts &K, uint64_t seed, uint64_t index, int j, double mean, double stddev)
{
  const double low = K.lb[j % 7], high = K.ub[j % 7];
  uint64_t r1 = splitmix64(seed ^ (index * 28ULL + 2ULL * (uint64_t)j));
  uint64_t r2 = splitmix64(seed ^ (index * 28ULL + 2ULL * (uint64_t)j + 1ULL));
  double u1 = (double)((r1 >> 11) + 1ULL) * 1.1102230246251565e-16; /* (0,1] */
  double u2 = (double)(r2 >> 11) * 1.1102230246251565e-16;          /* [0,1) */
  double s, c;
  ccmp_sincos(6.283185307179586 * u2, &s, &c);
  double z = ccmp_sqrt(-2.0 * ccmp_log(u1)) * c;
  double v = CCMP_FMA(z, stddev, mean);
  if (v < low) v = low;
  else if (v > high) v = high;
  return v;
}

} /* namespace ccmp */
#endif /* CCMP_KIN_H */
