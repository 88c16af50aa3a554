// ccmp_problem.cpp — problem set-up of libccmp (host only): the reference's YAML keys, the Panda constants of
// PandaModel::initModel, base frames, init_chain_ and the kernels' constant block.  Compiled by hipcc with
// -ffp-contract=off -DCCMP_USE_FMA so that the set-up arithmetic (ccmp_kin.h on the host) follows the same rounding
// model as the kernels.  No HIP runtime calls here (the runtime header only supplies the __host__ __device__ macros).
#include <hip/hip_runtime.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>

#include "ccmp_host.h"

namespace {

const double kPi = 3.14159265358979323846;
const double kPi2 = 1.57079632679489661923;

void identity3(double *R)
{
  for (int i = 0; i < 9; i++) R[i] = 0.0;
  R[0] = R[4] = R[8] = 1.0;
}

// PandaModel::transformDH, src/kinematics/panda_rbdl.cpp:150-160
void transform_dh(double a, double d, double alpha, double theta, double *R, double *p)
{
  double st, ct, sa, ca;
  ccmp_sincos(theta, &st, &ct);
  ccmp_sincos(alpha, &sa, &ca);
  R[0] = ct;      R[1] = -1 * st; R[2] = 0.0;
  R[3] = st * ca; R[4] = ct * ca; R[5] = -1 * sa;
  R[6] = st * sa; R[7] = ct * sa; R[8] = ca;
  p[0] = a; p[1] = -1 * sa * d; p[2] = ca * d;
}

// PandaModel::initModel(dh), src/kinematics/panda_rbdl.cpp:80-148 (the kinematic part): walk the
// modified-DH table at q = 0, record joint axes (column 2) and origins, the hand offset and the
// tool rotation.
void panda_constants(const double (*dh)[4], double axis[7][3], double offset[7][3], double ee[3], double R_tool[9])
{
  const double dh_al[7] = {0.0, -1.0 * kPi2, kPi2, kPi2, -1.0 * kPi2, kPi2, kPi2};
  const double dh_a[7] = {0.0, 0.0, 0.0, 0.0825, -0.0825, 0.0, 0.088};
  const double dh_d[7] = {0.333, 0.0, 0.316, 0.0, 0.384, 0.0, 0.0};
  double TR[9], Tp[3] = {0, 0, 0}, gpos[7][3];
  identity3(TR);
  for (int i = 0; i < 7; i++) {
    const double a_off = dh ? dh[i][0] : 0.0, d_off = dh ? dh[i][1] : 0.0;
    const double q_off = dh ? dh[i][2] : 0.0, al_off = dh ? dh[i][3] : 0.0;
    double R[9], p[3], NR[9], Rp[3] = {0, 0, 0};
    transform_dh(dh_a[i] + a_off, dh_d[i] + d_off, dh_al[i] + al_off, q_off, R, p);
    ccmp::mul33(TR, R, NR);
    for (int k = 0; k < 3; k++) Rp[k] = ccmp::dot3(TR[3 * k], p[0], TR[3 * k + 1], p[1], TR[3 * k + 2], p[2]);
    for (int k = 0; k < 3; k++) Tp[k] = Rp[k] + Tp[k];
    memcpy(TR, NR, sizeof NR);
    for (int k = 0; k < 3; k++) {
      axis[i][k] = TR[3 * k + 2];
      gpos[i][k] = Tp[k];
    }
  }
  const double ee0[3] = {0.0, 0.0, 0.107};
  for (int k = 0; k < 3; k++) ee[k] = ccmp::dot3(TR[3 * k], ee0[0], TR[3 * k + 1], ee0[1], TR[3 * k + 2], ee0[2]);
  for (int k = 0; k < 3; k++) offset[0][k] = gpos[0][k];
  for (int i = 1; i < 7; i++)
    for (int k = 0; k < 3; k++) offset[i][k] = gpos[i][k] - gpos[i - 1][k];
  double s, c;
  ccmp_sincos(-kPi / 4., &s, &c);
  const double Rz[9] = {c, -s, 0.0, s, c, 0.0, 0.0, 0.0, 1.0};
  ccmp::mul33(TR, Rz, R_tool);
}

// grasping_point::grasping_point, src/kinematics/grasping_point.cpp:5-20
void base_frame(int index, double *R, double *p)
{
  identity3(R);
  if (index == 0) { p[0] = 0; p[1] = 0.3; p[2] = 1.006; }
  else if (index == 1) { p[0] = 0; p[1] = -0.3; p[2] = 1.006; }
  else { p[0] = 1.35; p[1] = 0.3; p[2] = 1.006; R[0] = -1; R[4] = -1; }
}

// Eigen Quaternion::toRotationMatrix on the YAML quaternion (x,y,z,w), grasping_point.cpp:40-43
void quat_to_R(const double *q, double *R)
{
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// Do both arms carry the structure of the uncalibrated Panda that the STOCK kernels assume (ccmp_kin.h: kStockZ,
// kStockOff, kStockEe)?  Exact comparisons: a component the kernels skip must be exactly zero, a z joint's axis exactly
// (0, 0, 1).  Calibration offsets (ccmp_set_calibration) fail the test and get the general kernels.
bool is_stock_structure(const ccmp_problem &P)
{
  for (int a = 0; a < 2; a++) {
    for (int i = 0; i < 7; i++) {
      const double *ax = P.axis[a][i];
      if (ccmp::kStockZ[i] && !(ax[0] == 0.0 && ax[1] == 0.0 && ax[2] == 1.0)) return false;
      for (int k = 0; k < 3; k++)
        if (!((ccmp::kStockOff[i] >> k) & 1) && !(P.offset[a][i][k] == 0.0)) return false;
    }
    for (int k = 0; k < 3; k++)
      if (!((ccmp::kStockEe >> k) & 1) && !(P.ee[a][k] == 0.0)) return false;
  }
  return true;
}

// Kernel constants from the problem.  Pure re-packing plus products of constants (each a single
// IEEE multiply/add, hence identical wherever it is evaluated).
void make_consts_impl(const ccmp_problem &P, ccmp_consts &K)
{
  memset(&K, 0, sizeof K);
  for (int a = 0; a < 2; a++) {
    for (int i = 0; i < 7; i++) {
      const double *ax = P.axis[a][i];
      for (int k = 0; k < 3; k++) { K.axis[a][i][k] = ax[k]; K.offset[a][i][k] = P.offset[a][i][k]; }
      K.aprod[a][i][0] = ax[0] * ax[0]; K.aprod[a][i][1] = ax[0] * ax[1]; K.aprod[a][i][2] = ax[0] * ax[2];
      K.aprod[a][i][3] = ax[1] * ax[1]; K.aprod[a][i][4] = ax[1] * ax[2]; K.aprod[a][i][5] = ax[2] * ax[2];
    }
    for (int k = 0; k < 3; k++) { K.ee[a][k] = P.ee[a][k]; K.base_p[a][k] = P.base_p[a][k]; }
    for (int k = 0; k < 9; k++) { K.R_tool[a][k] = P.R_tool[a][k]; K.base_R[a][k] = P.base_R[a][k]; }
    bool diag = true; // t_wb.linear() exactly diag(+-1): tool_pose skips the products with exact zeros
    for (int k = 0; k < 9; k++) diag = diag && (k % 4 == 0 ? (P.base_R[a][k] == 1.0 || P.base_R[a][k] == -1.0) : P.base_R[a][k] == 0.0);
    if (diag) K.base_diag |= 1 << a;
  }
  K.stock = is_stock_structure(P) ? 1 : 0;
  K.twin_arms = (K.stock && K.base_diag == 3 && memcmp(P.axis[0], P.axis[1], sizeof P.axis[0]) == 0 &&
                 memcmp(P.offset[0], P.offset[1], sizeof P.offset[0]) == 0 && memcmp(P.ee[0], P.ee[1], sizeof P.ee[0]) == 0 &&
                 memcmp(P.R_tool[0], P.R_tool[1], sizeof P.R_tool[0]) == 0)
                    ? 1 : 0;
  for (int k = 0; k < 3; k++) K.init_p[k] = P.init_p[k];
  ccmp::quat_of(P.init_R, K.init_q);
  for (int i = 0; i < 7; i++) {
    K.lbe[i] = P.lb[i] + P.joint_eps;
    K.ube[i] = P.ub[i] - P.joint_eps;
    K.lb[i] = P.lb[i];
    K.ub[i] = P.ub[i];
    K.span[i] = P.ub[i] - P.lb[i];
  }
  // t_o7.inverse() of arm 0 = (R^T, -(R^T p))
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) K.t_o7i_R[3 * i + j] = P.t_o7_R[0][3 * j + i];
  {
    double tmp[3];
    ccmp::mulTvec(P.t_o7_R[0], P.t_o7_p[0], tmp);
    for (int k = 0; k < 3; k++) K.t_o7i_p[k] = -tmp[k];
  }
  K.tol_pos = P.tol_pos;
  K.tol_rot = P.tol_rot;
  K.step = P.step;
  K.max_iter = P.max_iter;
}

// inverse(T2) * T1 for isometries (Eigen Isometry3d ops of ConstraintFunction.h:39)
void rel_pose(const double *R1, const double *p1, const double *R2, const double *p2, double *Rc, double *pc)
{
  double ti[3];
  ccmp::mulT33(R2, R1, Rc);
  ccmp::mulTvec(R2, p2, ti);
  ccmp::mulTvec(R2, p1, pc);
  for (int k = 0; k < 3; k++) pc[k] = pc[k] + (-ti[k]);
}

// ---- YAML subset reader ---------------------------------------------------------------------------
// Enough for config/*.yaml of the reference: top-level "key: scalar", "key: [a, b, ...]" (possibly
// spanning lines) and one level of nested maps ("arm1:" followed by indented "name: ..."), '#'
// comments.  Keys are flattened to "arm1.name".
struct YamlDoc {
  std::map<std::string, std::string> kv;
};

std::string trim(const std::string &s)
{
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) a++;
  while (b > a && isspace((unsigned char)s[b - 1])) b--;
  return s.substr(a, b - a);
}

int parse_yaml(const char *path, YamlDoc &doc)
{
  FILE *fp = fopen(path, "rb");
  if (!fp) return CCMP_EIO;
  std::string text;
  char buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
  fclose(fp);
  std::string parent;
  size_t pos = 0;
  while (pos < text.size()) {
    size_t eol = text.find('\n', pos);
    if (eol == std::string::npos) eol = text.size();
    std::string line = text.substr(pos, eol - pos);
    pos = eol + 1;
    size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    if (trim(line).empty()) continue;
    const bool indented = isspace((unsigned char)line[0]) != 0;
    size_t colon = line.find(':');
    if (colon == std::string::npos) continue;
    std::string key = trim(line.substr(0, colon));
    std::string val = trim(line.substr(colon + 1));
    if (!indented) parent.clear();
    if (val.empty()) { // start of a nested map
      if (!indented) parent = key;
      continue;
    }
    if (val[0] == '[') { // inline list, maybe continued on following lines
      while (val.find(']') == std::string::npos && pos < text.size()) {
        size_t e2 = text.find('\n', pos);
        if (e2 == std::string::npos) e2 = text.size();
        std::string more = text.substr(pos, e2 - pos);
        pos = e2 + 1;
        size_t h2 = more.find('#');
        if (h2 != std::string::npos) more = more.substr(0, h2);
        val += " " + trim(more);
      }
    }
    if (indented && !parent.empty()) key = parent + "." + key;
    doc.kv[key] = val;
  }
  return CCMP_OK;
}

int yaml_doubles(const YamlDoc &d, const char *key, double *out, int count)
{
  auto it = d.kv.find(key);
  if (it == d.kv.end()) return CCMP_EPARSE;
  std::string v = it->second;
  size_t a = v.find('['), b = v.find(']');
  if (a == std::string::npos || b == std::string::npos || b < a) return CCMP_EPARSE;
  v = v.substr(a + 1, b - a - 1);
  int got = 0;
  const char *s = v.c_str();
  while (*s) {
    while (*s && (isspace((unsigned char)*s) || *s == ',')) s++;
    if (!*s) break;
    char *end = nullptr;
    double x = strtod(s, &end);
    if (end == s) return CCMP_EPARSE;
    if (got < count) out[got] = x;
    got++;
    s = end;
  }
  return got == count ? CCMP_OK : CCMP_EPARSE;
}

int yaml_string(const YamlDoc &d, const char *key, std::string &out)
{
  auto it = d.kv.find(key);
  if (it == d.kv.end()) return CCMP_EPARSE;
  out = it->second;
  if (out.size() >= 2 && (out[0] == '"' || out[0] == '\'') && out.back() == out[0]) out = out.substr(1, out.size() - 2);
  return CCMP_OK;
}

int yaml_int(const YamlDoc &d, const char *key, int &out)
{
  std::string s;
  if (yaml_string(d, key, s) != CCMP_OK) return CCMP_EPARSE;
  char *end = nullptr;
  long v = strtol(s.c_str(), &end, 10);
  if (end == s.c_str()) return CCMP_EPARSE;
  out = (int)v;
  return CCMP_OK;
}

} // namespace

namespace ccmp_host {
void make_consts(const ccmp_problem &P, ccmp_consts &K) { ::make_consts_impl(P, K); }
}  // namespace ccmp_host

extern "C" {

int ccmp_set_start(ccmp_problem *p, const double q0[14])
{
  if (!p || !q0) return CCMP_EINVAL;
  ccmp_consts K;
  make_consts_impl(*p, K);
  double R1[9], p1[3], R2[9], p2[3];
  memcpy(p->start_joint, q0, 14 * sizeof(double));
  ccmp::fk_arm(K, 0, q0, R1, p1);
  ccmp::fk_arm(K, 1, q0 + 7, R2, p2);
  rel_pose(R1, p1, R2, p2, p->init_R, p->init_p);
  rel_pose(R1, p1, p->obj_start_R, p->obj_start_p, p->t_o7_R[0], p->t_o7_p[0]);
  rel_pose(R2, p2, p->obj_start_R, p->obj_start_p, p->t_o7_R[1], p->t_o7_p[1]);
  return CCMP_OK;
}

int ccmp_set_tolerance(ccmp_problem *p, double tolerance1, double tolerance2)
{
  if (!p) return CCMP_EINVAL;
  if (tolerance1 <= 0 || tolerance2 <= 0 || tolerance1 != tolerance1 || tolerance2 != tolerance2) return CCMP_EINVAL;
  p->tol_pos = tolerance1;
  p->tol_rot = tolerance2;
  return CCMP_OK;
}

int ccmp_set_calibration(ccmp_problem *p, int arm_slot, const double dh_offsets[7][4])
{
  if (!p || arm_slot < 0 || arm_slot > 1) return CCMP_EINVAL;
  panda_constants(dh_offsets, p->axis[arm_slot], p->offset[arm_slot], p->ee[arm_slot], p->R_tool[arm_slot]);
  return ccmp_set_start(p, p->start_joint);
}

int ccmp_problem_init(ccmp_problem *out, const char *arm1_name, int arm1_index, const char *arm2_name,
                      int arm2_index, const double start_joint[14], const double obj_start_pos[3],
                      const double obj_start_quat_xyzw[4], const double obj_goal_pos[3],
                      const double obj_goal_quat_xyzw[4])
{
  static const double lb[7] = {-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973};
  static const double ub[7] = {2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973};
  if (!out || !arm1_name || !arm2_name || !start_joint) return CCMP_EINVAL;
  if (arm1_index < 0 || arm1_index > 2 || arm2_index < 0 || arm2_index > 2) return CCMP_EINVAL;
  memset(out, 0, sizeof *out);
  // std::map<std::string,int> order: alphabetical by arm name (ConstrainedPlanningCommon.cpp:13-14,89-91)
  int idx[2];
  if (strcmp(arm1_name, arm2_name) <= 0) { idx[0] = arm1_index; idx[1] = arm2_index; }
  else { idx[0] = arm2_index; idx[1] = arm1_index; }
  for (int a = 0; a < 2; a++) {
    out->arm_index[a] = idx[a];
    panda_constants(nullptr, out->axis[a], out->offset[a], out->ee[a], out->R_tool[a]);
    base_frame(idx[a], out->base_R[a], out->base_p[a]);
  }
  memcpy(out->lb, lb, sizeof lb);
  memcpy(out->ub, ub, sizeof ub);
  out->joint_eps = 0.001;
  out->tol_pos = 0.001;
  out->tol_rot = 0.005;
  out->step = 0.30;
  out->delta = 0.25;
  out->lambda = 2.0;
  out->max_iter = 250;
  out->jacobian_mode = CCMP_JAC_FD;
  identity3(out->obj_start_R);
  identity3(out->obj_goal_R);
  if (obj_start_quat_xyzw) quat_to_R(obj_start_quat_xyzw, out->obj_start_R);
  if (obj_goal_quat_xyzw) quat_to_R(obj_goal_quat_xyzw, out->obj_goal_R);
  if (obj_start_pos) memcpy(out->obj_start_p, obj_start_pos, 3 * sizeof(double));
  if (obj_goal_pos) memcpy(out->obj_goal_p, obj_goal_pos, 3 * sizeof(double));
  return ccmp_set_start(out, start_joint);
}

// KinematicChainConstraint::setArmModels on a problem that is already set up (the reference calls it after loadConfig,
// ConstrainedPlanningCommon.cpp:126): only the arm selection and the base frames change; init_chain_ and t_o7 follow.
// The arms are stored IN THE ORDER GIVEN, as ConstraintFunction.h:122-126 does (arm_models_.push_back(arm1), then arm2);
// putting them in std::map order is the caller's business there (ConstrainedProblem::_setEnvironment, :89-91) and
// ccmp_problem_init's / ccmp_problem_from_yaml's here.  The names are accepted for symmetry with ccmp_problem_init only.
int ccmp_set_arms(ccmp_problem *p, const char *arm1_name, int arm1_index, const char *arm2_name, int arm2_index)
{
  if (!p || !arm1_name || !arm2_name) return CCMP_EINVAL;
  if (arm1_index < 0 || arm1_index > 2 || arm2_index < 0 || arm2_index > 2) return CCMP_EINVAL;
  const int idx[2] = {arm1_index, arm2_index};
  for (int a = 0; a < 2; a++) {
    p->arm_index[a] = idx[a];
    base_frame(idx[a], p->base_R[a], p->base_p[a]);
  }
  return ccmp_set_start(p, p->start_joint);
}

// ArmModel::t_wb of one arm (include/closed_chain_motion_planner/kinematics/panda_model.h:15, filled from
// config->t_wb[index] at ConstrainedPlanningCommon.cpp:98 and read by KinematicChainConstraint::function at
// ConstraintFunction.h:89-90): the adapter hands over the frame the ArmModel really carries instead of re-deriving it
// from the index.  R must be a rotation (orthonormal to 1e-9, determinant +1) and everything finite.
int ccmp_set_base_frame(ccmp_problem *p, int arm_slot, const double R[9], const double pos[3])
{
  if (!p || !R || !pos || arm_slot < 0 || arm_slot > 1) return CCMP_EINVAL;
  for (int k = 0; k < 9; k++)
    if (!(R[k] - R[k] == 0.0)) return CCMP_EINVAL;
  for (int k = 0; k < 3; k++)
    if (!(pos[k] - pos[k] == 0.0)) return CCMP_EINVAL;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const double d = R[3 * i] * R[3 * j] + R[3 * i + 1] * R[3 * j + 1] + R[3 * i + 2] * R[3 * j + 2] - (i == j ? 1.0 : 0.0);
      if (d > 1e-9 || d < -1e-9) return CCMP_EINVAL;
    }
  const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
  if (!(det > 0.0)) return CCMP_EINVAL;
  memcpy(p->base_R[arm_slot], R, 9 * sizeof(double));
  memcpy(p->base_p[arm_slot], pos, 3 * sizeof(double));
  return ccmp_set_start(p, p->start_joint);
}

static int problem_from_yaml_impl(const char *yaml_path, ccmp_problem *out);

int ccmp_problem_from_yaml(const char *yaml_path, ccmp_problem *out)
{
  if (!yaml_path || !out) return CCMP_EINVAL;
  try { // the reader uses std::string / std::map: nothing may propagate through the C boundary
    return problem_from_yaml_impl(yaml_path, out);
  } catch (const std::bad_alloc &) {
    return CCMP_ENOMEM;
  } catch (...) {
    return CCMP_EPARSE;
  }
}

static int problem_from_yaml_impl(const char *yaml_path, ccmp_problem *out)
{
  YamlDoc doc;
  int rc = parse_yaml(yaml_path, doc);
  if (rc != CCMP_OK) return rc;
  double start[14], sp[3], sq[4], gp[3], gq[4];
  std::string n1, n2;
  int i1 = 0, i2 = 0;
  if ((rc = yaml_doubles(doc, "start_joint", start, 14)) != CCMP_OK) return rc;
  if ((rc = yaml_doubles(doc, "t_wo_start_pos", sp, 3)) != CCMP_OK) return rc;
  if ((rc = yaml_doubles(doc, "t_wo_start_quat", sq, 4)) != CCMP_OK) return rc;
  if ((rc = yaml_doubles(doc, "t_wo_goal_pos", gp, 3)) != CCMP_OK) return rc;
  if ((rc = yaml_doubles(doc, "t_wo_goal_quat", gq, 4)) != CCMP_OK) return rc;
  if ((rc = yaml_string(doc, "arm1.name", n1)) != CCMP_OK) return rc;
  if ((rc = yaml_string(doc, "arm2.name", n2)) != CCMP_OK) return rc;
  if ((rc = yaml_int(doc, "arm1.index", i1)) != CCMP_OK) return rc;
  if ((rc = yaml_int(doc, "arm2.index", i2)) != CCMP_OK) return rc;
  return ccmp_problem_init(out, n1.c_str(), i1, n2.c_str(), i2, start, sp, sq, gp, gq);
}

} // extern "C"
