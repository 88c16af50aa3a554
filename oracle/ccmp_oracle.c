/* ccmp_oracle.c — CPU restatement of the reference projector.  TEST INFRASTRUCTURE ONLY.
 * See ccmp_oracle.h for scope, sources and pin status.  Plain C99, FP64 throughout,
 * built with -ffp-contract=off so that the arithmetic below is exactly what is written.
 *
 * Operation order matters: the HIP FD-faithful kernel performs the same IEEE operations in the
 * same order (sums left to right, products as parenthesised) so that, in the detmath build, the
 * two agree bit for bit.  Do not "simplify" expressions here without changing the kernel.
 */
#include "ccmp_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_DETMATH
/* elementary functions + rounding model of csrc/ccmp_detmath.h (the checker may read product
 * code; the product may not read the checker).  Built with -DCCMP_USE_FMA: FMA(a,b,c) is fused. */
#include "ccmp_detmath.h"
#define ORC_SINCOS(x, s, c) ccmp_sincos((x), (s), (c))
#define ORC_ATAN2_NN(y, x) ccmp_atan2_nn((y), (x))
#define ORC_LOG(x) ccmp_log(x)
#define FMA(a, b, c) CCMP_FMA(a, b, c)
#else
/* what a stock x86-64 build of the reference does: glibc sin/cos/atan2, no fused operations */
#define ORC_SINCOS(x, s, c) do { *(s) = sin(x); *(c) = cos(x); } while (0)
#define ORC_ATAN2_NN(y, x) atan2((y), (x))
#define ORC_LOG(x) log(x)
#define FMA(a, b, c) ((a) * (b) + (c))
#endif
/* a0*b0 + a1*b1 + a2*b2, accumulated left to right (optionally onto c) */
#define DOT3(a0, b0, a1, b1, a2, b2) FMA(a2, b2, FMA(a1, b1, (a0) * (b0)))
#define DOT3ACC(c, a0, b0, a1, b1, a2, b2) FMA(a2, b2, FMA(a1, b1, FMA(a0, b0, c)))

#define ORC_PI 3.14159265358979323846
#define ORC_PI_2 1.57079632679489661923

void orc_sincos(double x, double *s, double *c) { ORC_SINCOS(x, s, c); }
double orc_atan2_nn(double y, double x) { return ORC_ATAN2_NN(y, x); }
int orc_is_detmath(void)
{
#ifdef ORC_DETMATH
  return 1;
#else
  return 0;
#endif
}
size_t orc_problem_sizeof(void) { return sizeof(orc_problem); }

/* ---- deliberately WRONG variants (all zero = the restatement; never set outside tests/test_oracle_golden.py) --------
 * Used for one purpose: to measure which choices of the un-vendored third-party arithmetic (OMPL's stencil, Eigen's
 * solve and angularDistance) the reference's recorded artefacts can tell apart and which they cannot
 * (test_recorded_paths_resolving_power).  Process-global, not thread-safe: set, run single-threaded, reset. */
static int g_variant[ORC_VAR_COUNT];
void orc_set_variant(int which, int value)
{
  if (which >= 0 && which < ORC_VAR_COUNT) g_variant[which] = value;
}
int orc_get_variant(int which) { return which >= 0 && which < ORC_VAR_COUNT ? g_variant[which] : -1; }

/* ---- tiny fixed-size linear algebra, explicit evaluation order ---------------------------- */
static void m3_mul(const double A[9], const double B[9], double C[9])
{
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = DOT3(A[3 * i], B[j], A[3 * i + 1], B[3 + j], A[3 * i + 2], B[6 + j]);
}
static void m3t_mul(const double A[9], const double B[9], double C[9]) /* A^T B */
{
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      C[3 * i + j] = DOT3(A[i], B[j], A[3 + i], B[3 + j], A[6 + i], B[6 + j]);
}
static void m3_vec(const double A[9], const double v[3], double r[3])
{
  for (int i = 0; i < 3; i++) r[i] = DOT3(A[3 * i], v[0], A[3 * i + 1], v[1], A[3 * i + 2], v[2]);
}
static void m3t_vec(const double A[9], const double v[3], double r[3])
{
  for (int i = 0; i < 3; i++) r[i] = DOT3(A[i], v[0], A[3 + i], v[1], A[6 + i], v[2]);
}
/* r += A v, each component accumulated left to right onto r */
static void m3_vec_acc(const double A[9], const double v[3], double r[3])
{
  for (int i = 0; i < 3; i++) r[i] = DOT3ACC(r[i], A[3 * i], v[0], A[3 * i + 1], v[1], A[3 * i + 2], v[2]);
}
static void m3_identity(double R[9])
{
  for (int i = 0; i < 9; i++) R[i] = 0.0;
  R[0] = R[4] = R[8] = 1.0;
}

/* Rotation about a fixed axis; RBDL Xrot(angle, axis) transposed (RBDL stores world->body).
 * Entry formula order follows RBDL: axis_i*axis_j*(1-c) +- axis_k*s. */
static void rot_axis(const double a[3], double q, double R[9])
{
  double s, c;
  ORC_SINCOS(q, &s, &c);
  double t = 1.0 - c;
  double a0s = a[0] * s, a1s = a[1] * s, a2s = a[2] * s;
  R[0] = FMA(a[0] * a[0], t, c);
  R[1] = FMA(a[0] * a[1], t, -a2s);
  R[2] = FMA(a[0] * a[2], t, a1s);
  R[3] = FMA(a[1] * a[0], t, a2s);
  R[4] = FMA(a[1] * a[1], t, c);
  R[5] = FMA(a[1] * a[2], t, -a0s);
  R[6] = FMA(a[2] * a[0], t, -a1s);
  R[7] = FMA(a[2] * a[1], t, a0s);
  R[8] = FMA(a[2] * a[2], t, c);
}

/* Eigen Quaternion::toRotationMatrix (no normalisation); q = (x,y,z,w). */
static void quat_xyzw_to_R(const double q[4], double R[9])
{
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* Eigen Quaterniond(Matrix3d): trace / major-diagonal branches. out = (x,y,z,w). */
static void R_to_quat(const double m[9], double q[4])
{
  double t = (m[0] + m[4]) + m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(((m[4 * i] - m[4 * j]) - m[4 * k]) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
}

/* ---- setup ---------------------------------------------------------------------------------- */
/* PandaModel::transformDH, panda_rbdl.cpp:150-160 */
static void transform_dh(double a, double d, double alpha, double theta, double R[9], double p[3])
{
  double st, ct, sa, ca;
  ORC_SINCOS(theta, &st, &ct);
  ORC_SINCOS(alpha, &sa, &ca);
  R[0] = ct;      R[1] = -1 * st; R[2] = 0.0;
  R[3] = st * ca; R[4] = ct * ca; R[5] = -1 * sa;
  R[6] = st * sa; R[7] = ct * sa; R[8] = ca;
  p[0] = a; p[1] = -1 * sa * d; p[2] = ca * d;
}

/* PandaModel::initModel(dh), panda_rbdl.cpp:80-148 (kinematic part only) */
void orc_panda_constants(const double dh_off[7][4], double axis[7][3], double offset[7][3],
                         double ee[3], double R_tool[9])
{
  const double dh_al[7] = {0.0, -1.0 * ORC_PI_2, ORC_PI_2, ORC_PI_2, -1.0 * ORC_PI_2, ORC_PI_2, ORC_PI_2};
  const double dh_a[7] = {0.0, 0.0, 0.0, 0.0825, -0.0825, 0.0, 0.088};
  const double dh_d[7] = {0.333, 0.0, 0.316, 0.0, 0.384, 0.0, 0.0};
  double TR[9], Tp[3] = {0, 0, 0}, gpos[7][3];
  m3_identity(TR);
  for (int i = 0; i < 7; i++) {
    double a_off = dh_off ? dh_off[i][0] : 0.0, d_off = dh_off ? dh_off[i][1] : 0.0;
    double q_off = dh_off ? dh_off[i][2] : 0.0, al_off = dh_off ? dh_off[i][3] : 0.0;
    double R[9], p[3], NR[9], Rp[3];
    transform_dh(dh_a[i] + a_off, dh_d[i] + d_off, dh_al[i] + al_off, q_off, R, p);
    m3_mul(TR, R, NR);   /* T = T * DH : linear */
    m3_vec(TR, p, Rp);   /*             translation = TR*p + Tp */
    for (int k = 0; k < 3; k++) Tp[k] = Rp[k] + Tp[k];
    memcpy(TR, NR, sizeof NR);
    for (int k = 0; k < 3; k++) {
      axis[i][k] = TR[3 * k + 2]; /* column 2 */
      gpos[i][k] = Tp[k];
    }
  }
  const double ee0[3] = {0.0, 0.0, 0.107};
  m3_vec(TR, ee0, ee); /* ee_position_ = rot_ee_ * (0,0,0.107) */
  for (int k = 0; k < 3; k++) offset[0][k] = gpos[0][k];
  for (int i = 1; i < 7; i++)
    for (int k = 0; k < 3; k++) offset[i][k] = gpos[i][k] - gpos[i - 1][k];
  /* M = rot_ee_ * AngleAxisd(-pi/4, UnitZ), panda_rbdl.cpp:31 */
  double s, c, Rz[9];
  ORC_SINCOS(-ORC_PI / 4., &s, &c);
  Rz[0] = c; Rz[1] = -s; Rz[2] = 0.0; Rz[3] = s; Rz[4] = c; Rz[5] = 0.0; Rz[6] = 0.0; Rz[7] = 0.0; Rz[8] = 1.0;
  m3_mul(TR, Rz, R_tool);
}

/* grasping_point::grasping_point, grasping_point.cpp:5-20 */
void orc_base_frame(int arm_index, double R[9], double p[3])
{
  m3_identity(R);
  if (arm_index == 0) { p[0] = 0; p[1] = 0.3; p[2] = 1.006; }
  else if (arm_index == 1) { p[0] = 0; p[1] = -0.3; p[2] = 1.006; }
  else { p[0] = 1.35; p[1] = 0.3; p[2] = 1.006; R[0] = -1; R[4] = -1; }
}

/* PandaModel::getTransform (panda_rbdl.cpp:24-42) followed by t_wb * (ConstraintFunction.h:89-90) */
void orc_fk(const orc_problem *P, int arm, const double q[7], double Rw[9], double pw[3])
{
  double R[9], o[3] = {0, 0, 0};
  m3_identity(R);
  for (int i = 0; i < 7; i++) {
    double Rj[9], Rn[9];
    m3_vec_acc(R, P->offset[arm][i], o);           /* r_i = r_{i-1} + E_{i-1}^T * X_T.r      */
    rot_axis(P->axis[arm][i], q[i], Rj);           /* X_J = Xrot(q_i, axis_i)                */
    m3_mul(R, Rj, Rn);
    memcpy(R, Rn, sizeof Rn);
  }
  double Rf[9];
  m3_vec_acc(R, P->ee[arm], o);                    /* CalcBodyToBaseCoordinates(ee): o + R*ee */
  m3_mul(R, P->R_tool[arm], Rf);                   /* CalcBodyWorldOrientation^T * M         */
  m3_mul(P->base_R[arm], Rf, Rw);                  /* t_wb * T                               */
  for (int k = 0; k < 3; k++) pw[k] = P->base_p[arm][k];
  m3_vec_acc(P->base_R[arm], o, pw);               /* base_p + base_R * p                    */
}

/* current_chain = t_w72.inverse() * t_w71 (ConstraintFunction.h:92, Eigen Isometry ops) */
static void chain_of(const double R1[9], const double p1[3], const double R2[9], const double p2[3],
                     double Rc[9], double pc[3])
{
  double ti[3], tp[3];
  m3t_mul(R2, R1, Rc);
  m3t_vec(R2, p2, ti);
  for (int k = 0; k < 3; k++) ti[k] = -ti[k]; /* inverse().translation = -(R^T p) */
  m3t_vec(R2, p1, tp);
  for (int k = 0; k < 3; k++) pc[k] = tp[k] + ti[k];
}

/* residual of a chain against init_chain_; also returns d = q_c * conj(q_0) for the analytic J */
static void residual_of_chain(const orc_problem *P, const double Rc[9], const double pc[3], double f[2],
                              double dquat[4])
{
  double qc[4], q0[4];
  R_to_quat(Rc, qc);
  R_to_quat(P->init_R, q0);
  /* d = qc * conj(q0), Eigen quaternion product with b = (w0, -x0, -y0, -z0) */
  double ax = qc[0], ay = qc[1], az = qc[2], aw = qc[3];
  double bx = -q0[0], by = -q0[1], bz = -q0[2], bw = q0[3];
  double dw = FMA(-az, bz, FMA(-ay, by, FMA(-ax, bx, aw * bw)));
  double dx = FMA(-az, by, FMA(ay, bz, FMA(ax, bw, aw * bx)));
  double dy = FMA(-ax, bz, FMA(az, bx, FMA(ay, bw, aw * by)));
  double dz = FMA(-ay, bx, FMA(ax, by, FMA(az, bw, aw * bz)));
  double vn = sqrt(DOT3(dx, dx, dy, dy, dz, dz));
  f[1] = 2.0 * ORC_ATAN2_NN(vn, fabs(dw)); /* angularDistance, Eigen >= 3.3 */
  if (g_variant[ORC_VAR_ANGLE] == 1) { /* variant: Eigen 3.2's angularDistance, 2*acos(|a.b|) */
    double dd = fabs(FMA(qc[3], q0[3], DOT3(qc[0], q0[0], qc[1], q0[1], qc[2], q0[2])));
    f[1] = dd >= 1.0 ? 0.0 : 2.0 * acos(dd);
  }
  double e0 = pc[0] - P->init_p[0], e1 = pc[1] - P->init_p[1], e2 = pc[2] - P->init_p[2];
  f[0] = sqrt(DOT3(e0, e0, e1, e1, e2, e2));
  if (dquat) { dquat[0] = dx; dquat[1] = dy; dquat[2] = dz; dquat[3] = dw; }
}

/* KinematicChainConstraint::function, ConstraintFunction.h:84-102 */
void orc_function(const orc_problem *P, const double x[14], double f[2])
{
  double R1[9], p1[3], R2[9], p2[3], Rc[9], pc[3];
  orc_fk(P, 0, x, R1, p1);
  orc_fk(P, 1, x + 7, R2, p2);
  chain_of(R1, p1, R2, p2, Rc, pc);
  residual_of_chain(P, Rc, pc, f, NULL);
}

/* setInitialPosition (ConstraintFunction.h:31-40) + t_o7 (ConstrainedPlanningCommon.cpp:102-111) */
void orc_set_start(orc_problem *P, const double q0[14])
{
  double R1[9], p1[3], R2[9], p2[3];
  memcpy(P->start_joint, q0, 14 * sizeof(double));
  orc_fk(P, 0, q0, R1, p1);
  orc_fk(P, 1, q0 + 7, R2, p2);
  chain_of(R1, p1, R2, p2, P->init_R, P->init_p);
  /* t_o7 = t_wo_start.inverse() * t_w7a */
  chain_of(R1, p1, P->obj_start_R, P->obj_start_p, P->t_o7_R[0], P->t_o7_p[0]);
  chain_of(R2, p2, P->obj_start_R, P->obj_start_p, P->t_o7_R[1], P->t_o7_p[1]);
}

int orc_problem_init(orc_problem *P, const char *name1, int index1, const char *name2, int index2,
                     const double start_joint[14], const double obj_start_pos[3],
                     const double obj_start_quat_xyzw[4], const double obj_goal_pos[3],
                     const double obj_goal_quat_xyzw[4])
{
  static const double lb[7] = {-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973};
  static const double ub[7] = {2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973};
  memset(P, 0, sizeof *P);
  if (index1 < 0 || index1 > 2 || index2 < 0 || index2 > 2) return -1;
  /* std::map<std::string,int> iteration order = alphabetical (ConstrainedPlanningCommon.cpp:13-14,89-91) */
  int idx[2];
  if (strcmp(name1, name2) <= 0) { idx[0] = index1; idx[1] = index2; }
  else { idx[0] = index2; idx[1] = index1; }
  for (int a = 0; a < 2; a++) {
    P->arm_index[a] = idx[a];
    orc_panda_constants(NULL, P->axis[a], P->offset[a], P->ee[a], P->R_tool[a]);
    orc_base_frame(idx[a], P->base_R[a], P->base_p[a]);
  }
  memcpy(P->lb, lb, sizeof lb);
  memcpy(P->ub, ub, sizeof ub);
  P->joint_eps = 0.001;
  P->tol_pos = 0.001;
  P->tol_rot = 0.005;
  P->step = 0.30;
  P->delta = 0.25;
  P->lambda = 2.0;
  P->max_iter = 250;
  P->jacobian_mode = ORC_JAC_FD;
  m3_identity(P->obj_start_R);
  m3_identity(P->obj_goal_R);
  if (obj_start_quat_xyzw) quat_xyzw_to_R(obj_start_quat_xyzw, P->obj_start_R);
  if (obj_goal_quat_xyzw) quat_xyzw_to_R(obj_goal_quat_xyzw, P->obj_goal_R);
  if (obj_start_pos) memcpy(P->obj_start_p, obj_start_pos, 3 * sizeof(double));
  if (obj_goal_pos) memcpy(P->obj_goal_p, obj_goal_pos, 3 * sizeof(double));
  orc_set_start(P, start_joint);
  return 0;
}

/* ---- Jacobians ------------------------------------------------------------------------------ */
/* ompl::base::Constraint::jacobian (default; the reference does not override it): 7-point
 * central stencil, 6 residual evaluations per column, h = sqrt(eps)*max(1,|x_j|), divided by the
 * *stored* perturbed difference. */
void orc_jacobian_fd(const orc_problem *P, const double x[14], double J[28])
{
  double y1[14], y2[14], t1[2], t2[2];
  memcpy(y1, x, sizeof y1);
  memcpy(y2, x, sizeof y2);
  for (int j = 0; j < 14; j++) {
    const double ax = fabs(x[j]);
    double h = 1.4901161193847656e-08 /* sqrt(DBL_EPSILON) = 2^-26 */ * (ax >= 1 ? ax : 1);
    if (g_variant[ORC_VAR_H] == 1) h = 1e-6 * (ax >= 1 ? ax : 1); /* variant: a coarser step */
    double m[3][2];
    for (int s = 0; s < 3; s++) {
      y1[j] += h;
      y2[j] -= h;
      orc_function(P, y1, t1);
      orc_function(P, y2, t2);
      const double den = y1[j] - y2[j];
      m[s][0] = (t1[0] - t2[0]) / den;
      m[s][1] = (t1[1] - t2[1]) / den;
    }
    for (int r = 0; r < 2; r++) J[r * 14 + j] = FMA(0.1, m[2][r], FMA(-0.6, m[1][r], 1.5 * m[0][r]));
    if (g_variant[ORC_VAR_STENCIL] == 1) /* variant: plain 3-point central difference */
      for (int r = 0; r < 2; r++) J[r * 14 + j] = m[0][r];
    y1[j] = y2[j] = x[j];
  }
}

/* Per-joint world frames of one arm: z_i (rotation axis) and o_i (joint origin) in world coords. */
static void fk_frames(const orc_problem *P, int arm, const double q[7], double z[7][3], double ow[7][3],
                      double Rw[9], double pw[3])
{
  double R[9], o[3] = {0, 0, 0};
  m3_identity(R);
  for (int i = 0; i < 7; i++) {
    double Rj[9], Rn[9], zl[3];
    m3_vec_acc(R, P->offset[arm][i], o);
    m3_vec(R, P->axis[arm][i], zl);
    m3_vec(P->base_R[arm], zl, z[i]);
    for (int k = 0; k < 3; k++) ow[i][k] = P->base_p[arm][k];
    m3_vec_acc(P->base_R[arm], o, ow[i]);
    rot_axis(P->axis[arm][i], q[i], Rj);
    m3_mul(R, Rj, Rn);
    memcpy(R, Rn, sizeof Rn);
  }
  double Rf[9];
  m3_vec_acc(R, P->ee[arm], o);
  m3_mul(R, P->R_tool[arm], Rf);
  m3_mul(P->base_R[arm], Rf, Rw);
  for (int k = 0; k < 3; k++) pw[k] = P->base_p[arm][k];
  m3_vec_acc(P->base_R[arm], o, pw);
}

/* Exact derivative of the residual: row0 = u^T dp_c/dq, row1 = n^T dw_c/dq, formulated in WORLD
 * coordinates.  Not what the reference computes (it differentiates numerically); kept as the independent cross-check
 * of the FD stencil and of orc_jacobian_analytic below (tests compare the two to ~1e-13). */
void orc_jacobian_analytic_world(const orc_problem *P, const double x[14], double J[28])
{
  double z[2][7][3], o[2][7][3], R1[9], p1[3], R2[9], p2[3], Rc[9], pc[3], f[2], d[4];
  fk_frames(P, 0, x, z[0], o[0], R1, p1);
  fk_frames(P, 1, x + 7, z[1], o[1], R2, p2);
  chain_of(R1, p1, R2, p2, Rc, pc);
  residual_of_chain(P, Rc, pc, f, d);
  double u[3] = {0, 0, 0}, n[3] = {0, 0, 0}, a[3], b[3];
  if (f[0] > 0.0)
    for (int k = 0; k < 3; k++) u[k] = (pc[k] - P->init_p[k]) / f[0];
  double vn = sqrt(DOT3(d[0], d[0], d[1], d[1], d[2], d[2]));
  if (vn > 0.0) {
    double sg = d[3] < 0.0 ? -1.0 : 1.0;
    for (int k = 0; k < 3; k++) n[k] = sg * d[k] / vn;
  }
  m3_vec(R2, u, a); /* world direction of the position-error gradient */
  m3_vec(R2, n, b); /* world direction of the rotation-error gradient */
  for (int arm = 0; arm < 2; arm++) {
    double sgn = arm == 0 ? 1.0 : -1.0;
    for (int i = 0; i < 7; i++) {
      const double *zi = z[arm][i];
      double r[3] = {p1[0] - o[arm][i][0], p1[1] - o[arm][i][1], p1[2] - o[arm][i][2]};
      double cx = zi[1] * r[2] - zi[2] * r[1];
      double cy = zi[2] * r[0] - zi[0] * r[2];
      double cz = zi[0] * r[1] - zi[1] * r[0];
      J[0 * 14 + arm * 7 + i] = sgn * DOT3(a[0], cx, a[1], cy, a[2], cz);
      J[1 * 14 + arm * 7 + i] = sgn * DOT3(b[0], zi[0], b[1], zi[1], b[2], zi[2]);
    }
  }
}

/* The analytic mode of the product (jacobian_mode = CCMP_JAC_ANALYTIC, csrc/ccmp_kernels_fast.hip) — this library's own
 * extension, no counterpart in the reference.  The same derivative as above with every joint's axis and origin kept
 * in its ARM'S BASE frame and the probe vectors carried there (what a GPU lane can do without world-frame copies of
 * all fourteen joint frames), operation for operation in the order the kernel runs them, so that the det build can be
 * compared with the kernel bit for bit:
 *   u = (pc - p0) * (1 / f0),  n = dq_vec * (sign(dq_w) / |dq_vec|)          (chain frame)
 *   aw = R2 u, bw = R2 n                                                     (world)
 *   per arm: al = Rb^T aw, bl = Rb^T bw, pl = Rb^T (p1 - pb)                 (arm base frame)
 *   per joint: z_i = R_(i) axis_i, o_i = joint origin, both in the base frame;
 *              J0 = +-al . (z_i x (pl - o_i)),  J1 = +-bl . z_i             (+ arm 0, - arm 1) */
void orc_jacobian_analytic(const orc_problem *P, const double x[14], double J[28])
{
  double z[2][7][3], oj[2][7][3], Rw[2][9], pw[2][3];
  for (int arm = 0; arm < 2; arm++) {
    double R[9], o[3] = {0, 0, 0};
    m3_identity(R);
    for (int i = 0; i < 7; i++) {
      double Rj[9], Rn[9];
      m3_vec_acc(R, P->offset[arm][i], o);
      m3_vec(R, P->axis[arm][i], z[arm][i]);
      for (int k = 0; k < 3; k++) oj[arm][i][k] = o[k];
      rot_axis(P->axis[arm][i], x[7 * arm + i], Rj);
      m3_mul(R, Rj, Rn);
      memcpy(R, Rn, sizeof Rn);
    }
    double Rf[9];
    m3_vec_acc(R, P->ee[arm], o);
    m3_mul(R, P->R_tool[arm], Rf);
    m3_mul(P->base_R[arm], Rf, Rw[arm]);
    for (int k = 0; k < 3; k++) pw[arm][k] = P->base_p[arm][k];
    m3_vec_acc(P->base_R[arm], o, pw[arm]);
  }
  double Rc[9], pc[3], f[2], d[4];
  chain_of(Rw[0], pw[0], Rw[1], pw[1], Rc, pc);
  residual_of_chain(P, Rc, pc, f, d);
  double u[3] = {0, 0, 0}, n[3] = {0, 0, 0}, aw[3], bw[3];
  if (f[0] > 0.0) {
    const double inv = 1.0 / f[0];
    for (int k = 0; k < 3; k++) u[k] = (pc[k] - P->init_p[k]) * inv;
  }
  const double vn = sqrt(DOT3(d[0], d[0], d[1], d[1], d[2], d[2]));
  if (vn > 0.0) {
    const double sg = (d[3] < 0.0 ? -1.0 : 1.0) / vn;
    for (int k = 0; k < 3; k++) n[k] = d[k] * sg;
  }
  m3_vec(Rw[1], u, aw);
  m3_vec(Rw[1], n, bw);
  for (int arm = 0; arm < 2; arm++) {
    double al[3], bl[3], pl[3], dp[3];
    for (int k = 0; k < 3; k++) dp[k] = pw[0][k] - P->base_p[arm][k];
    m3t_vec(P->base_R[arm], aw, al);
    m3t_vec(P->base_R[arm], bw, bl);
    m3t_vec(P->base_R[arm], dp, pl);
    const double sgn = arm == 0 ? 1.0 : -1.0;
    for (int i = 0; i < 7; i++) {
      const double *zi = z[arm][i];
      const double r0 = pl[0] - oj[arm][i][0], r1 = pl[1] - oj[arm][i][1], r2 = pl[2] - oj[arm][i][2];
      const double cx = FMA(zi[1], r2, -(zi[2] * r1));
      const double cy = FMA(zi[2], r0, -(zi[0] * r2));
      const double cz = FMA(zi[0], r1, -(zi[1] * r0));
      J[0 * 14 + arm * 7 + i] = sgn * DOT3(al[0], cx, al[1], cy, al[2], cz);
      J[1 * 14 + arm * 7 + i] = sgn * DOT3(bl[0], zi[0], bl[1], zi[1], bl[2], zi[2]);
    }
  }
}

/* Eigen::JacobiSVD<MatrixXd>(j, ThinU|ThinV).solve(f) for the 2x14 j (ConstraintFunction.h:71):
 * minimum-norm least-squares solution, singular values <= 2*eps*sigma_max treated as zero.
 * Realised as a one-sided (Hestenes) Jacobi SVD on the two rows — two fixed sweeps, the second
 * is a polish — which keeps kappa*eps accuracy without forming (J J^T)^-1 explicitly. */
void orc_solve_minnorm(const double J[28], const double f[2], double dx[14])
{
  if (g_variant[ORC_VAR_SOLVE]) { /* variants: 1 = normal equations J^T (J J^T)^-1 f, 2 = the same damped by 1e-4 I */
    double a = 0, d = 0, b = 0;
    for (int j = 0; j < 14; j++) { a += J[j] * J[j]; d += J[14 + j] * J[14 + j]; b += J[j] * J[14 + j]; }
    if (g_variant[ORC_VAR_SOLVE] == 2) { a += 1e-4; d += 1e-4; }
    const double det = a * d - b * b;
    const double y0 = (d * f[0] - b * f[1]) / det, y1 = (a * f[1] - b * f[0]) / det;
    for (int j = 0; j < 14; j++) dx[j] = J[j] * y0 + J[14 + j] * y1;
    return;
  }
  double r0[14], r1[14], g0 = f[0], g1 = f[1];
  memcpy(r0, J, sizeof r0);
  memcpy(r1, J + 14, sizeof r1);
  double a = 0, d = 0, b = 0;
  for (int sweep = 0; sweep < 2; sweep++) {
    a = 0; d = 0; b = 0;
    for (int j = 0; j < 14; j++) {
      a = FMA(r0[j], r0[j], a);
      d = FMA(r1[j], r1[j], d);
      b = FMA(r0[j], r1[j], b);
    }
    if (b != 0.0) {
      double zeta = (d - a) / (2.0 * b);
      double t = 1.0 / (fabs(zeta) + sqrt(FMA(zeta, zeta, 1.0)));
      if (zeta < 0.0) t = -t;
      double c = 1.0 / sqrt(FMA(t, t, 1.0));
      double s = c * t;
      for (int j = 0; j < 14; j++) {
        double v0 = r0[j], v1 = r1[j];
        r0[j] = FMA(c, v0, -(s * v1));
        r1[j] = FMA(s, v0, c * v1);
      }
      double h0 = g0, h1 = g1;
      g0 = FMA(c, h0, -(s * h1));
      g1 = FMA(s, h0, c * h1);
    }
  }
  a = 0; d = 0;
  for (int j = 0; j < 14; j++) { a = FMA(r0[j], r0[j], a); d = FMA(r1[j], r1[j], d); }
  double s0 = sqrt(a), s1 = sqrt(d);
  double smax = s0 > s1 ? s0 : s1;
  double thr = smax * (2.0 * 2.220446049250313e-16); /* diagSize * epsilon */
  if (thr < 2.2250738585072014e-308) thr = 2.2250738585072014e-308;
  double k0 = s0 > thr ? g0 / a : 0.0;
  double k1 = s1 > thr ? g1 / d : 0.0;
  for (int j = 0; j < 14; j++) dx[j] = FMA(k1, r1[j], k0 * r0[j]);
}

/* The analytic mode's step (jacobian_mode = ORC_JAC_ANALYTIC; this library's own extension): the same
 * minimum-norm solution dx = J^T (J J^T)^-1 f through the 2x2 Gram matrix in closed form, operation for operation in the
 * order csrc/ccmp_kernels_fast.hip runs it — each arm's seven columns summed from zero (one lane per arm there), the two
 * partial sums added — wherever the two rows are not nearly parallel: det / (a d) = sin^2 of the angle between them, and
 * above 2^-20 the closed form's cancellation costs at most ~1e-10 relative.  Below that (or on a NaN) the SVD-equivalent
 * routine above takes over, so that rank handling stays what Eigen's JacobiSVD::solve gives the reference. */
void orc_solve_gram(const double J[28], const double f[2], double dx[14])
{
  double pa[2], pd[2], pb[2];
  for (int arm = 0; arm < 2; arm++) {
    double a = 0.0, d = 0.0, b = 0.0;
    for (int j = 7 * arm; j < 7 * arm + 7; j++) {
      a = FMA(J[j], J[j], a);
      d = FMA(J[14 + j], J[14 + j], d);
      b = FMA(J[j], J[14 + j], b);
    }
    pa[arm] = a; pd[arm] = d; pb[arm] = b;
  }
  const double a = pa[0] + pa[1], d = pd[0] + pd[1], b = pb[0] + pb[1];
  const double det = FMA(a, d, -(b * b));
  if (!(det > (a * d) * 9.5367431640625e-07)) { /* 2^-20 */
    orc_solve_minnorm(J, f, dx);
    return;
  }
  const double inv = 1.0 / det;
  const double y0 = FMA(d, f[0], -(b * f[1])) * inv;
  const double y1 = FMA(a, f[1], -(b * f[0])) * inv;
  for (int j = 0; j < 14; j++) dx[j] = FMA(y1, J[14 + j], y0 * J[j]);
}

/* ---- the projector --------------------------------------------------------------------------- */
/* KinematicChainConstraint::jointValid, ConstraintFunction.h:43-55 */
int orc_joint_valid(const orc_problem *P, const double q[14])
{
  double eps = P->joint_eps;
  for (int arm = 0; arm < 2; arm++)
    for (int i = 0; i < 7; i++) {
      if (q[arm * 7 + i] < P->lb[i] + eps) return 0;
      if (q[arm * 7 + i] > P->ub[i] - eps) return 0;
    }
  return 1;
}

/* KinematicChainConstraint::project, ConstraintFunction.h:57-82 — including the precedence
 * quirk at :68 (norm1 receives the boolean f[0] > tol1; norm2 receives f[1] only when that is
 * false), iter++ evaluated only when the residual test holds, x left at the last iterate. */
int orc_project(const orc_problem *P, double x[14], int32_t *iters)
{
  unsigned int iter = 0;
  int32_t updates = 0;
  double norm1 = 0, norm2 = 0;
  double f[2], J[28], dx[14];
  orc_function(P, x, f);
  while (((norm1 = (double)(f[0] > P->tol_pos)) != 0.0 || (norm2 = f[1]) > P->tol_rot) &&
         iter++ < (unsigned int)P->max_iter) {
    if (P->jacobian_mode == ORC_JAC_ANALYTIC) {
      orc_jacobian_analytic(P, x, J);
      orc_solve_gram(J, f, dx);
    } else {
      orc_jacobian_fd(P, x, J);
      orc_solve_minnorm(J, f, dx);
    }
    for (int i = 0; i < 14; i++) x[i] = FMA(-P->step, dx[i], x[i]); /* x -= 0.30*dx */
    orc_function(P, x, f);
    updates++;
  }
  if (iters) *iters = updates;
  if (g_variant[ORC_VAR_RETURN] == 1) /* variant: the test a reader expects — both residuals, non-strict */
    return orc_joint_valid(P, x) && f[0] <= P->tol_pos && f[1] <= P->tol_rot;
  return orc_joint_valid(P, x) && (norm1 < P->tol_pos) && (norm2 < P->tol_rot);
}

/* KinematicChainConstraint::isSatisfied, ConstraintFunction.h:114-120 */
int orc_is_satisfied(const orc_problem *P, const double x[14])
{
  double f[2];
  orc_function(P, x, f);
  return isfinite(f[0]) && isfinite(f[1]) && f[0] <= P->tol_pos && f[1] <= P->tol_rot;
}

/* ---- state space: KinematicChain.h:118-130,145-171; RealVectorStateSpace::distance ---------- */
void orc_enforce_bounds(double x[14])
{
  for (int i = 0; i < 14; i++) {
    double v = fmod(x[i], 2.0 * ORC_PI);
    if (v < -ORC_PI) v += 2.0 * ORC_PI;
    else if (v >= ORC_PI) v -= 2.0 * ORC_PI;
    x[i] = v;
  }
}

void orc_interpolate(const double from[14], const double to[14], double t, double out[14])
{
  for (int i = 0; i < 14; i++) {
    double diff = to[i] - from[i];
    if (fabs(diff) <= ORC_PI) out[i] = FMA(diff, t, from[i]);
    else {
      if (diff > 0.0) diff = 2.0 * ORC_PI - diff;
      else diff = -2.0 * ORC_PI - diff;
      double v = FMA(-diff, t, from[i]);
      if (v > ORC_PI) v -= 2.0 * ORC_PI;
      else if (v < -ORC_PI) v += 2.0 * ORC_PI;
      out[i] = v;
    }
  }
}

double orc_distance(const double a[14], const double b[14])
{
  double dist = 0.0;
  for (int i = 0; i < 14; i++) {
    double diff = a[i] - b[i];
    dist = FMA(diff, diff, dist);
  }
  return sqrt(dist);
}

/* ---- sampler: jy_ProjectedStateSpace.cpp:10-15 with the counter-based generator of
 * SURVEY.md §8d standing in for OMPL's (time-seeded, unreproducible) RNG ---------------------- */
uint64_t orc_splitmix64(uint64_t z)
{
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

void orc_ambient_uniform(const orc_problem *P, uint64_t seed, uint64_t index, double q[14])
{
  for (int j = 0; j < 14; j++) {
    uint64_t r = orc_splitmix64(seed ^ (index * 14ULL + (uint64_t)j));
    double u = (double)(r >> 11) * 1.1102230246251565e-16; /* 2^-53 */
    q[j] = FMA(P->ub[j % 7] - P->lb[j % 7], u, P->lb[j % 7]); /* RNG::uniformReal(low, high) */
  }
}

double orc_log(double x) { return ORC_LOG(x); }

/* RealVectorStateSampler::sampleUniformNear [3P OMPL]: per dimension
 * uniformReal(max(low, near - distance), min(high, near + distance)) */
void orc_ambient_near(const orc_problem *P, uint64_t seed, uint64_t index, const double near[14], double distance,
                      double q[14])
{
  for (int j = 0; j < 14; j++) {
    const double low = P->lb[j % 7], high = P->ub[j % 7];
    const double a = (near[j] - distance) > low ? (near[j] - distance) : low;
    const double b = (near[j] + distance) < high ? (near[j] + distance) : high;
    uint64_t r = orc_splitmix64(seed ^ (index * 14ULL + (uint64_t)j));
    double u = (double)(r >> 11) * 1.1102230246251565e-16;
    q[j] = FMA(b - a, u, a);
  }
}

/* RealVectorStateSampler::sampleGaussian [3P OMPL]: rng_.gaussian(mean, stdDev) clamped to the bounds.
 * OMPL draws from std::normal_distribution (library-defined, unreproducible); the deviate here is
 * Box-Muller on two counter-based uniforms — the same definition the kernel uses. */
void orc_ambient_gaussian(const orc_problem *P, uint64_t seed, uint64_t index, const double mean[14], double stddev,
                          double q[14])
{
  for (int j = 0; j < 14; j++) {
    const double low = P->lb[j % 7], high = P->ub[j % 7];
    uint64_t r1 = orc_splitmix64(seed ^ (index * 28ULL + 2ULL * (uint64_t)j));
    uint64_t r2 = orc_splitmix64(seed ^ (index * 28ULL + 2ULL * (uint64_t)j + 1ULL));
    double u1 = (double)((r1 >> 11) + 1ULL) * 1.1102230246251565e-16;
    double u2 = (double)(r2 >> 11) * 1.1102230246251565e-16;
    double s, c;
    ORC_SINCOS(6.283185307179586 * u2, &s, &c);
    double z = sqrt(-2.0 * ORC_LOG(u1)) * c;
    double v = FMA(z, stddev, mean[j]);
    if (v < low) v = low;
    else if (v > high) v = high;
    q[j] = v;
  }
}

int orc_sample_uniform(const orc_problem *P, uint64_t seed, uint64_t index, double q[14], int32_t *iters)
{
  orc_ambient_uniform(P, seed, index, q);
  int ok = orc_project(P, q, iters); /* return value ignored by the reference sampler */
  orc_enforce_bounds(q);
  return ok;
}

/* ---- jy_ProjectedStateSpace::discreteGeodesic, jy_ProjectedStateSpace.cpp:32-96 --------------
 * carry_in / carry_out (nullable) are this project's resumable form (include/ccmp.h: ccmp_geodesic_batch_ex): an edge
 * whose list is full stops, and a continuation re-enters the reference's do-while from the last stored state with the
 * running length it had before the step that did not fit and the bound lambda * dist(from, to) of the first call. */
int orc_discrete_geodesic_ex(const orc_problem *P, const double from[14], const double to[14],
                             int interpolate, orc_valid_fn valid, void *user, double *out,
                             int max_states, int *n_states, int64_t *newton_iters, const double carry_in[2],
                             double carry_out[2])
{
  /* when an accepted state finds the list full the traversal stops and reports max_states + 1 states (and false):
   * the caller continues the edge (or re-runs it with a larger buffer) — a creeping edge must not run on unbounded */
  int n = 1;
  int64_t its = 0;
  if (out && max_states > 0) memcpy(out, from, 14 * sizeof(double));
  const double tolerance = P->delta;
  double dist, step = 0, total = 0;
  dist = orc_distance(from, to);
  double max = dist * P->lambda;
  if (carry_in) { total = carry_in[0]; max = carry_in[1]; }
  if (carry_out) { carry_out[0] = total; carry_out[1] = max; }
  if (carry_in ? !(dist >= tolerance) : dist <= tolerance) {
    if (n_states) *n_states = n;
    if (newton_iters) *newton_iters = 0;
    return dist <= tolerance;
  }
  double previous[14], scratch[14];
  memcpy(previous, from, sizeof previous);
  do {
    orc_interpolate(previous, to, P->delta / dist, scratch);
    int32_t it = 0;
    int proj = orc_project(P, scratch, &it);
    its += it;
    if (!proj || !(interpolate || !valid || valid(scratch, user)) ||
        (step = orc_distance(previous, scratch)) > P->lambda * P->delta)
      break;
    const double total_before = total;
    total += step;
    if (total > max) break;
    const double newDist = orc_distance(scratch, to);
    if (newDist >= dist) break;
    if (out && n >= max_states) {
      if (n_states) *n_states = max_states + 1;
      if (newton_iters) *newton_iters = its - it; /* the state that did not fit is projected again by the continuation */
      if (carry_out) { carry_out[0] = total_before; carry_out[1] = max; }
      return 0;
    }
    dist = newDist;
    memcpy(previous, scratch, sizeof previous);
    if (out) memcpy(out + 14 * n, scratch, 14 * sizeof(double));
    n++;
  } while (dist >= tolerance);
  if (n_states) *n_states = n;
  if (newton_iters) *newton_iters = its;
  if (carry_out) { carry_out[0] = total; carry_out[1] = max; }
  return dist <= tolerance;
}

int orc_discrete_geodesic(const orc_problem *P, const double from[14], const double to[14],
                          int interpolate, orc_valid_fn valid, void *user, double *out,
                          int max_states, int *n_states, int64_t *newton_iters)
{
  return orc_discrete_geodesic_ex(P, from, to, interpolate, valid, user, out, max_states, n_states, newton_iters, NULL, NULL);
}

/* IKTask::compute_t_wo, ik_task.cpp:10-14: t_wb * FK(q_left) * t_o7.inverse() for panda_left
 * (always arm slot 0: "panda_left" sorts first). */
void orc_compute_t_wo(const orc_problem *P, const double q_left[7], double R[9], double p[3])
{
  double Rw[9], pw[3], Ri[9], pi_[3], tmp[3];
  orc_fk(P, 0, q_left, Rw, pw);
  /* inverse of t_o7 */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Ri[3 * i + j] = P->t_o7_R[0][3 * j + i];
  m3t_vec(P->t_o7_R[0], P->t_o7_p[0], pi_);
  for (int k = 0; k < 3; k++) pi_[k] = -pi_[k];
  m3_mul(Rw, Ri, R);
  (void)tmp;
  for (int k = 0; k < 3; k++) p[k] = pw[k];
  m3_vec_acc(Rw, pi_, p);
}

/* ---- proxy-geometry clearance (see ccmp_oracle.h) -------------------------------------------------------------- */
/* frames of one arm in the arm's base: bodies 0..6 (R after joint k's rotation, origin at joint k) and the hand frame */
static void arm_frames(const orc_problem *P, int arm, const double q[7], double Rk[8][9], double ok[8][3])
{
  double R[9], o[3] = {0, 0, 0};
  m3_identity(R);
  for (int i = 0; i < 7; i++) {
    double Rj[9], Rn[9];
    m3_vec_acc(R, P->offset[arm][i], o);
    rot_axis(P->axis[arm][i], q[i], Rj);
    m3_mul(R, Rj, Rn);
    memcpy(R, Rn, sizeof Rn);
    memcpy(Rk[i], R, sizeof R);
    memcpy(ok[i], o, sizeof o);
  }
  m3_vec_acc(R, P->ee[arm], o);
  m3_mul(R, P->R_tool[arm], Rk[7]);
  memcpy(ok[7], o, sizeof o);
}

void orc_proxy_centres(const orc_problem *P, const orc_sphere *sph, int ns, const double x[14], double *centres)
{
  double Rk[2][8][9], ok[2][8][3];
  arm_frames(P, 0, x, Rk[0], ok[0]);
  arm_frames(P, 1, x + 7, Rk[1], ok[1]);
  for (int s = 0; s < ns; s++) {
    double *w = centres + 3 * s;
    const int f = sph[s].frame;
    if (f < 0) { memcpy(w, sph[s].c, 3 * sizeof(double)); continue; }
    const int arm = f / 9, k = f % 9;
    double v[3];
    if (k == 8) memcpy(v, sph[s].c, sizeof v);                 /* arm base: only t_wb applies */
    else {
      memcpy(v, ok[arm][k], sizeof v);
      m3_vec_acc(Rk[arm][k], sph[s].c, v);                     /* origin + R * c, in the arm's base */
    }
    for (int c = 0; c < 3; c++) w[c] = P->base_p[arm][c];
    m3_vec_acc(P->base_R[arm], v, w);                          /* t_wb * */
  }
}

static int frame_is_static(int f) { return f < 0 || f % 9 == 8; }

int orc_clearance(const orc_problem *P, const orc_sphere *sph, int ns, const orc_box *box, int nb, const uint32_t allowed[32],
                  const double x[14], double *clearance, int32_t *pair)
{
  double cw[64][3];
  double best = INFINITY;
  int32_t best_pair = -1;
  int tested = 0, finite = 1;
  for (int k = 0; k < 14; k++) if (!(x[k] - x[k] == 0.0)) finite = 0;
  if (finite) orc_proxy_centres(P, sph, ns, x, &cw[0][0]);
  for (int i = 0; i < ns; i++)
    for (int j = i + 1; j < ns; j++) {
      if (sph[i].frame == sph[j].frame) continue;
      if (frame_is_static(sph[i].frame) && frame_is_static(sph[j].frame)) continue;
      if (allowed && (((allowed[sph[i].group] >> sph[j].group) & 1u) || ((allowed[sph[j].group] >> sph[i].group) & 1u))) continue;
      tested++;
      if (!finite) continue;
      const double d0 = cw[i][0] - cw[j][0], d1 = cw[i][1] - cw[j][1], d2 = cw[i][2] - cw[j][2];
      const double clr = sqrt(DOT3(d0, d0, d1, d1, d2, d2)) - (sph[i].r + sph[j].r);
      if (clr < best) { best = clr; best_pair = i | (j << 8); }
    }
  for (int i = 0; i < ns; i++)
    for (int b = 0; b < nb; b++) {
      if (frame_is_static(sph[i].frame)) continue;
      if (allowed && (((allowed[sph[i].group] >> box[b].group) & 1u) || ((allowed[box[b].group] >> sph[i].group) & 1u))) continue;
      tested++;
      if (!finite) continue;
      double d[3], l[3], e[3];
      for (int c = 0; c < 3; c++) d[c] = cw[i][c] - box[b].c[c];
      m3t_vec(box[b].R, d, l);                                  /* into the box's axes */
      for (int c = 0; c < 3; c++) {
        const double a = fabs(l[c]) - box[b].half[c];
        e[c] = a > 0.0 ? a : 0.0;
      }
      const double clr = sqrt(DOT3(e[0], e[0], e[1], e[1], e[2], e[2])) - sph[i].r;
      if (clr < best) { best = clr; best_pair = i | ((64 + b) << 8); }
    }
  *clearance = finite ? best : NAN;
  if (pair) *pair = finite ? best_pair : -1;
  return tested;
}

void orc_clearance_batch(const orc_problem *P, const orc_sphere *sph, int ns, const orc_box *box, int nb, const uint32_t allowed[32],
                         const double *q, size_t B, double *clearance, int32_t *pair)
{
  for (size_t i = 0; i < B; i++) (void)orc_clearance(P, sph, ns, box, nb, allowed, q + 14 * i, clearance + i, pair ? pair + i : NULL);
}

/* ---- batch drivers ---------------------------------------------------------------------------- */
/* Threads take samples in small dynamic chunks from a shared counter: iteration counts spread 15..250 per
 * sample, so a static B/n partition leaves most threads idle while the unluckiest one finishes. */
typedef struct {
  const orc_problem *P;
  const double *q_in, *q_to;
  double *q_out;
  double *f;
  uint8_t *ok;
  int32_t *iters, *n_states;
  uint64_t seed, first;
  size_t B, chunk;
  size_t *next; /* shared ticket counter */
  int max_states;
  int kind; /* 0 project, 1 function, 2 sample+project, 3 discreteGeodesic (interpolate == true) */
} orc_job;

static void orc_do(const orc_job *j, size_t i)
{
  if (j->kind == 1) { orc_function(j->P, j->q_in + 14 * i, j->f + 2 * i); return; }
  if (j->kind == 3) {
    int n = 0;
    int64_t its = 0;
    int ok = orc_discrete_geodesic(j->P, j->q_in + 14 * i, j->q_to + 14 * i, 1, NULL, NULL,
                                   j->q_out + 14 * (size_t)j->max_states * i, j->max_states, &n, &its);
    j->ok[i] = (uint8_t)ok;
    j->n_states[i] = n;
    if (j->iters) j->iters[i] = (int32_t)its;
    return;
  }
  double x[14];
  int32_t it = 0;
  int ok;
  if (j->kind == 0) { memcpy(x, j->q_in + 14 * i, sizeof x); ok = orc_project(j->P, x, &it); }
  else ok = orc_sample_uniform(j->P, j->seed, j->first + i, x, &it);
  memcpy(j->q_out + 14 * i, x, sizeof x);
  if (j->ok) j->ok[i] = (uint8_t)ok;
  if (j->iters) j->iters[i] = it;
}

static void *orc_worker(void *arg)
{
  const orc_job *j = (const orc_job *)arg;
  for (;;) {
    const size_t lo = __atomic_fetch_add(j->next, j->chunk, __ATOMIC_RELAXED);
    if (lo >= j->B) break;
    const size_t hi = lo + j->chunk < j->B ? lo + j->chunk : j->B;
    for (size_t i = lo; i < hi; i++) orc_do(j, i);
  }
  return NULL;
}

static void orc_run(orc_job job, size_t B, int nthreads)
{
  if (nthreads < 1) nthreads = 1;
  if ((size_t)nthreads > B) nthreads = B ? (int)B : 1;
  size_t next = 0;
  job.B = B;
  job.next = &next;
  job.chunk = job.kind == 1 ? 256 : (job.kind == 3 ? 1 : 4);
  if (nthreads == 1) { orc_worker(&job); return; }
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
  for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, orc_worker, &job);
  for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
  free(th);
}

void orc_function_batch(const orc_problem *P, const double *q, double *f, size_t B, int nthreads)
{
  orc_job j; memset(&j, 0, sizeof j);
  j.P = P; j.q_in = q; j.f = f; j.kind = 1;
  orc_run(j, B, nthreads);
}

void orc_project_batch(const orc_problem *P, const double *q_in, double *q_out, uint8_t *ok,
                       int32_t *iters, size_t B, int nthreads)
{
  orc_job j; memset(&j, 0, sizeof j);
  j.P = P; j.q_in = q_in; j.q_out = q_out; j.ok = ok; j.iters = iters; j.kind = 0;
  orc_run(j, B, nthreads);
}

void orc_sample_project_batch(const orc_problem *P, uint64_t seed, uint64_t first_index,
                              double *q_out, uint8_t *ok, int32_t *iters, size_t B, int nthreads)
{
  orc_job j; memset(&j, 0, sizeof j);
  j.P = P; j.q_out = q_out; j.ok = ok; j.iters = iters; j.seed = seed; j.first = first_index; j.kind = 2;
  orc_run(j, B, nthreads);
}

/* E edges through orc_discrete_geodesic (interpolate == true, no validity callback): states [E][max_states][14],
 * n_states[e] the true length (may exceed max_states), newton_iters[e] (nullable) the Newton updates spent */
void orc_discrete_geodesic_batch(const orc_problem *P, const double *from, const double *to, size_t E, int max_states,
                                 double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters, int nthreads)
{
  orc_job j; memset(&j, 0, sizeof j);
  j.P = P; j.q_in = from; j.q_to = to; j.q_out = states; j.n_states = n_states; j.ok = ok; j.iters = newton_iters;
  j.max_states = max_states; j.kind = 3;
  orc_run(j, E, nthreads);
}
