// ccmp_api.cpp — host side of libccmp: problem set-up (YAML subset reader, Panda constants,
// init_chain_), execution context, argument checking and kernel launches.  Compiled by hipcc with
// -ffp-contract=off -DCCMP_USE_FMA so that the set-up arithmetic (ccmp_kin.h on the host) follows
// the same rounding model as the kernels.
//
// There is deliberately no CPU implementation of the hot path in this library: project / function /
// isSatisfied batches run on the GPU or fail with CCMP_ENODEV / CCMP_EHIP.
#include <hip/hip_runtime.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/ccmp.h"
#include "ccmp_kin.h"

extern "C" {
hipError_t ccmp_launch_project_group(const ccmp_consts *K, int mode, const double *q_in, double *q_out, uint8_t *ok,
                                     uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue,
                                     unsigned long long seed, unsigned long long first, int nblocks, double *pool,
                                     int dump_threshold, const unsigned int *order, hipStream_t st);
hipError_t ccmp_launch_project_wave(const ccmp_consts *K, int src, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue_head,
                                    unsigned long long seed, unsigned long long first, const double *pool,
                                    const unsigned long long *pool_count, int wrap_output, int nblocks, hipStream_t st);
hipError_t ccmp_launch_project_flat(const ccmp_consts *K, int src, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue_head,
                                    unsigned long long seed, unsigned long long first, const double *pool,
                                    const unsigned long long *pool_count, int wrap_output, int nblocks, hipStream_t st);
hipError_t ccmp_launch_project_fast(const ccmp_consts *K, int mode, const double *q_in, double *q_out, uint8_t *ok,
                                    uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue,
                                    unsigned long long seed, unsigned long long first, int nblocks, hipStream_t st);
hipError_t ccmp_launch_scout_order(const ccmp_consts *K, int mode, const double *q_in, size_t B, uint16_t *pred,
                                   unsigned int *hist, unsigned int *order, unsigned long long *queue,
                                   unsigned long long seed, unsigned long long first, int nblocks, hipStream_t st);
hipError_t ccmp_launch_function(const ccmp_consts *K, const double *q, double *f, size_t B, hipStream_t st);
hipError_t ccmp_launch_is_satisfied(const ccmp_consts *K, const double *q, uint8_t *ok, size_t B, hipStream_t st);
hipError_t ccmp_launch_joint_valid(const ccmp_consts *K, const double *q, uint8_t *ok, size_t B, hipStream_t st);
hipError_t ccmp_launch_ambient_uniform(const ccmp_consts *K, unsigned long long seed, unsigned long long first,
                                       double *q, size_t B, hipStream_t st);
hipError_t ccmp_launch_enforce_bounds(double *q, size_t B, hipStream_t st);
hipError_t ccmp_launch_ambient_ref(const ccmp_consts *K, int kind, unsigned long long seed, unsigned long long first,
                                   const double *ref, int ref_stride, double param, double *q, size_t B, hipStream_t st);
hipError_t ccmp_launch_t_wo(const ccmp_consts *K, const double *q, int q_stride, double *out, size_t B, hipStream_t st);
hipError_t ccmp_launch_geodesic(const ccmp_consts *K, double delta, double lambda, const double *from, const double *to,
                                size_t E, int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters,
                                int nblocks, hipStream_t st);
hipError_t ccmp_launch_detmath_probe(const double *x, const double *y, double *out, size_t n, hipStream_t st);
hipError_t ccmp_launch_compact(const double *q, const uint8_t *ok, size_t B, double *out, unsigned int *block_counts,
                               unsigned long long *total, hipStream_t st);
}

namespace {

thread_local char g_hip_err[256] = "";

int hip_fail(hipError_t e, const char *what)
{
  snprintf(g_hip_err, sizeof g_hip_err, "%s: %s", what, hipGetErrorString(e));
  return CCMP_EHIP;
}
#define HIP_TRY(call)                                   \
  do {                                                  \
    hipError_t e_ = (call);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #call);   \
  } while (0)

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
void make_consts(const ccmp_problem &P, ccmp_consts &K)
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

// scheduling defaults (sweeps: tools/time_small.py, tools/time_mid.py, tools/time_lpt3.py)
constexpr size_t kDefaultSmallBatch = 24576;   // up to here the latency kernel alone is quickest
constexpr size_t kDefaultLptMinBatch = 28672;  // from about one fill of the throughput kernel (30720 samples) on, ordering pays for the scout

struct ccmp_ctx {
  int device = 0;
  int num_cus = 0;
  int waves_per_cu = 0;
  hipStream_t stream = nullptr;
  unsigned long long *queue = nullptr; // work-queue heads of the projector kernels (4 words)
  double *pool = nullptr;              // straggler hand-over records (group kernel -> wave kernel)
  size_t pool_cap = 0;                 // in records
  int wave_kernel = 1;                 // 0: group kernel only, 1: group + wave-per-sample (default), 2: wave only
  const unsigned int *order = nullptr; // experimental: externally supplied processing order
  int flat_kernel = 1;                 // latency work (small batches, hand-over): 1 = one-round 128-thread kernel, 0 = single-wave kernel
  int stock_kernels = 1;               // 0: always the general kernels, also for the stock Panda structure (tests, A/B)
  int lpt = 1;                         // 0: in-order; 1: FP32 scout + longest-predicted-first, hand-over kept; 2: same, no hand-over
  size_t lpt_min_batch = kDefaultLptMinBatch; // below this the scout costs more than the tail it removes
  void *lpt_buf = nullptr;             // pred (u16 x B) | hist (u32 x 1024) | order (u32 x B)
  size_t lpt_cap = 0;                  // in samples
  int dump_threshold = -1;             // hand a wave's samples over once the queue is dry and <= this many groups are busy; -1 = auto
  size_t small_batch = kDefaultSmallBatch;    // at or below: latency kernel on everything
  unsigned int *scan = nullptr;        // compaction block counts
  size_t scan_cap = 0;
  // staging for the *_host conveniences
  void *stage = nullptr;
  size_t stage_cap = 0;
  void *pin = nullptr;     // pinned, device-mapped host block for small *_host calls (single states of the reference signature)
  void *pin_dev = nullptr; // the same block as the kernels see it
};

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev)
  {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard()
  {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int ensure_stage(ccmp_ctx *ctx, size_t bytes)
{
  if (ctx->stage_cap >= bytes) return CCMP_OK;
  if (ctx->stage) (void)hipFree(ctx->stage);
  ctx->stage = nullptr;
  ctx->stage_cap = 0;
  HIP_TRY(hipMalloc(&ctx->stage, bytes));
  ctx->stage_cap = bytes;
  return CCMP_OK;
}

// Host buffers of the *_host entry points.  Up to kPinBytes the kernels work directly on a pinned, device-mapped host
// block (a single project(x) is then memcpy + launch + synchronize + memcpy: no staged pageable copies, ~3 driver
// calls fewer); larger batches go through device staging with asynchronous copies.
constexpr size_t kPinBytes = 64 * 1024;
struct HostIO {
  ccmp_ctx *ctx;
  char *dev = nullptr;  // what the kernels get
  char *host = nullptr; // pinned alias (small calls) or nullptr
  struct Out { void *dst; size_t off, n; } outs[4];
  int n_outs = 0;
  explicit HostIO(ccmp_ctx *c) : ctx(c) {}
  int begin(size_t bytes)
  {
    if (bytes <= kPinBytes) {
      if (!ctx->pin) {
        HIP_TRY(hipHostMalloc(&ctx->pin, kPinBytes, hipHostMallocMapped));
        hipError_t e = hipHostGetDevicePointer(&ctx->pin_dev, ctx->pin, 0);
        if (e != hipSuccess) { (void)hipHostFree(ctx->pin); ctx->pin = nullptr; return hip_fail(e, "hipHostGetDevicePointer"); }
      }
      host = (char *)ctx->pin;
      dev = (char *)ctx->pin_dev;
      return CCMP_OK;
    }
    int rc = ensure_stage(ctx, bytes);
    dev = (char *)ctx->stage;
    return rc;
  }
  int in(size_t off, const void *src, size_t n)
  {
    if (host) { memcpy(host + off, src, n); return CCMP_OK; }
    HIP_TRY(hipMemcpyAsync(dev + off, src, n, hipMemcpyHostToDevice, ctx->stream));
    return CCMP_OK;
  }
  int out(void *dst, size_t off, size_t n)
  {
    if (host) { outs[n_outs++] = Out{dst, off, n}; return CCMP_OK; }
    HIP_TRY(hipMemcpyAsync(dst, dev + off, n, hipMemcpyDeviceToHost, ctx->stream));
    return CCMP_OK;
  }
  int finish()
  {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_outs; i++) memcpy(outs[i].dst, host + outs[i].off, outs[i].n);
    return CCMP_OK;
  }
};

int projector_blocks(const ccmp_ctx *ctx, size_t B, int samples_per_wave, int default_wpc)
{
  const int wpc = ctx->waves_per_cu > 0 ? ctx->waves_per_cu : default_wpc;
  size_t want = (B + samples_per_wave - 1) / samples_per_wave;
  size_t cap = (size_t)ctx->num_cus * (size_t)wpc;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (int)want;
}

int check_problem(const ccmp_problem *p)
{
  if (!p) return CCMP_EINVAL;
  if (!(p->tol_pos > 0) || !(p->tol_rot > 0)) return CCMP_EINVAL;
  if (p->max_iter < 0 || p->max_iter > 65535) return CCMP_EINVAL;
  if (p->jacobian_mode != CCMP_JAC_FD && p->jacobian_mode != CCMP_JAC_ANALYTIC) return CCMP_EINVAL;
  return CCMP_OK;
}

} // namespace

extern "C" {

int ccmp_version(void) { return CCMP_VERSION; }
size_t ccmp_problem_sizeof(void) { return sizeof(ccmp_problem); }
const char *ccmp_last_hip_error(void) { return g_hip_err; }

const char *ccmp_strerror(int code)
{
  switch (code) {
    case CCMP_OK: return "ok";
    case CCMP_EINVAL: return "invalid argument";
    case CCMP_EHIP: return "HIP runtime error";
    case CCMP_EIO: return "cannot read file";
    case CCMP_EPARSE: return "YAML key missing or malformed";
    case CCMP_ENODEV: return "no usable HIP device";
    case CCMP_ENOMEM: return "out of memory";
    default: return "unknown error";
  }
}

int ccmp_set_start(ccmp_problem *p, const double q0[14])
{
  if (!p || !q0) return CCMP_EINVAL;
  ccmp_consts K;
  make_consts(*p, K);
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

// ---- context ---------------------------------------------------------------------------------------
int ccmp_ctx_create(int device, ccmp_ctx **out)
{
  if (!out) return CCMP_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    snprintf(g_hip_err, sizeof g_hip_err, "hipGetDeviceCount: no HIP device");
    return CCMP_ENODEV;
  }
  if (device < 0 || device >= ndev) return CCMP_ENODEV;
  DeviceGuard guard(device);
  if (!guard.ok) return CCMP_ENODEV;
  ccmp_ctx *ctx = new (std::nothrow) ccmp_ctx();
  if (!ctx) return CCMP_ENOMEM;
  ctx->device = device;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { delete ctx; return hip_fail(e, "hipGetDeviceProperties"); }
  ctx->num_cus = prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete ctx; return hip_fail(e, "hipStreamCreate"); }
  e = hipMalloc((void **)&ctx->queue, 64);
  if (e != hipSuccess) { (void)hipStreamDestroy(ctx->stream); delete ctx; return hip_fail(e, "hipMalloc(queue)"); }
  *out = ctx;
  return CCMP_OK;
}

void ccmp_ctx_destroy(ccmp_ctx *ctx)
{
  if (!ctx) return;
  DeviceGuard guard(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->queue) (void)hipFree(ctx->queue);
  if (ctx->pool) (void)hipFree(ctx->pool);
  if (ctx->lpt_buf) (void)hipFree(ctx->lpt_buf);
  if (ctx->scan) (void)hipFree(ctx->scan);
  if (ctx->stage) (void)hipFree(ctx->stage);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int ccmp_ctx_set_waves_per_cu(ccmp_ctx *ctx, int w)
{
  if (!ctx || w < 0 || w > 32) return CCMP_EINVAL;
  ctx->waves_per_cu = w;
  return CCMP_OK;
}
int ccmp_ctx_set_schedule(ccmp_ctx *ctx, int wave_kernel, size_t small_batch)
{
  if (!ctx || wave_kernel < 0 || wave_kernel > 2) return CCMP_EINVAL;
  ctx->wave_kernel = wave_kernel;
  ctx->small_batch = small_batch == CCMP_DEFAULT ? kDefaultSmallBatch : small_batch;
  return CCMP_OK;
}
int ccmp_ctx_set_option(ccmp_ctx *ctx, const char *name, long value)
{
  if (!ctx || !name) return CCMP_EINVAL;
  if (!strcmp(name, "flat_kernel")) { // latency work: 1 = one-round 128-thread kernel (default), 0 = single-wave kernel
    if (value != 0 && value != 1) return CCMP_EINVAL;
    ctx->flat_kernel = (int)value;
  } else if (!strcmp(name, "stock_kernels")) { // 1 = kernels specialised for the stock Panda structure when it applies (default)
    if (value != 0 && value != 1) return CCMP_EINVAL;
    ctx->stock_kernels = (int)value;
  } else if (!strcmp(name, "handover_threshold")) { // -1 = automatic, 0..10 = hand a wave over once <= this many groups are busy
    if (value < -1 || value > 10) return CCMP_EINVAL;
    ctx->dump_threshold = (int)value;
  } else {
    return CCMP_EINVAL;
  }
  return CCMP_OK;
}
int ccmp_ctx_set_lpt(ccmp_ctx *ctx, int mode, size_t min_batch)
{
  if (!ctx || mode < 0 || mode > 2) return CCMP_EINVAL;
  ctx->lpt = mode;
  ctx->lpt_min_batch = min_batch == CCMP_DEFAULT ? kDefaultLptMinBatch : min_batch;
  return CCMP_OK;
}
int ccmp_ctx_debug_lpt_pred(ccmp_ctx *ctx, uint16_t *host_out, size_t B)
{
  if (!ctx || !host_out || !ctx->lpt_buf || B > ctx->lpt_cap) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host_out, ctx->lpt_buf, B * sizeof(uint16_t), hipMemcpyDeviceToHost));
  return CCMP_OK;
}
int ccmp_ctx_set_order_experimental(ccmp_ctx *ctx, const unsigned int *order_dev)
{
  if (!ctx) return CCMP_EINVAL;
  ctx->order = order_dev;
  return CCMP_OK;
}
int ccmp_ctx_device(const ccmp_ctx *ctx) { return ctx ? ctx->device : -1; }
int ccmp_ctx_num_cus(const ccmp_ctx *ctx) { return ctx ? ctx->num_cus : 0; }

#define CCMP_PROLOGUE()                                        \
  if (!ctx) return CCMP_EINVAL;                                \
  { int rc_ = check_problem(p); if (rc_ != CCMP_OK) return rc_; } \
  DeviceGuard guard(ctx->device);                              \
  if (!guard.ok) return CCMP_ENODEV;                           \
  hipStream_t st = (hipStream_t)hip_stream; \
  ccmp_consts K;                                               \
  make_consts(*p, K);                                          \
  if (!ctx->stock_kernels) K.stock = 0

int ccmp_function_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, double *f, size_t B, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q || !f) return CCMP_EINVAL;
  HIP_TRY(ccmp_launch_function(&K, q, f, B, st));
  return CCMP_OK;
}

// ---- scheduling of a reference-arithmetic batch ---------------------------------------------------------------------
// Which kernels a batch of B samples runs on.  Pure function of the context's settings and B; never changes a result.
//   latency kernel alone            B <= small_batch (or schedule 2): one sample per block, lowest latency per sample
//   throughput kernel (+ hand-over) otherwise: 10 samples per wavefront from a queue; when the queue runs dry the samples
//                                   still in flight go to the latency kernel
//   scout + longest-first order     from lpt_min_batch on; from 120000 samples on without hand-over (see below)
struct FdPlan {
  int group_blocks = 0;    // persistent wavefronts of the throughput kernel; 0 = latency kernel alone
  bool handover = false;   // throughput kernel dumps its last samples to the pool, the latency kernel finishes them
  bool scout = false;      // FP32 scout pass + descending counting sort -> processing order
  int dump_threshold = 10; // a wave hands over once the queue is dry and at most this many of its 10 groups are busy
  int latency_blocks = 0;  // grid of the latency kernel (direct launch or hand-over)
  bool latency_static = false; // one block per sample, static striding: no queue word to reset
};

static FdPlan plan_fd_batch(const ccmp_ctx *ctx, size_t B, bool external_order)
{
  FdPlan pl;
  const int wpc = ctx->waves_per_cu > 0 ? ctx->waves_per_cu : 12;
  // latency kernels: the flat kernel runs 8 blocks of 128 threads per CU (16 waves), the one-wave kernel wpc waves
  const size_t lat_cap = ctx->flat_kernel ? (size_t)ctx->num_cus * 8 : (size_t)ctx->num_cus * (size_t)wpc;
  const bool latency_only = ctx->wave_kernel == 2 || (ctx->wave_kernel == 1 && B <= ctx->small_batch);
  if (latency_only) {
    pl.latency_blocks = (int)(B < lat_cap ? B : lat_cap);
    pl.latency_static = ctx->flat_kernel && B <= lat_cap;
    return pl;
  }
  pl.group_blocks = projector_blocks(ctx, B, 10, 12);
  pl.handover = ctx->wave_kernel == 1;
  pl.scout = !external_order && ctx->lpt > 0 && B >= ctx->lpt_min_batch && B < 0xffffffffull;
  // Large ordered batches end on their shortest samples, and the scout is accurate there (tools/scout_tail.py: in the
  // last fill of a 262144-sample batch it predicts <= 21 iterations and the truth is <= 23): nothing worth handing
  // over is left (-3 % at 262144 Wine_Bottle without it, tools/time_lpt3.py).  An explicit threshold keeps hand-over.
  if (pl.scout && (ctx->lpt == 2 || (ctx->dump_threshold < 0 && B >= 120000))) pl.handover = false;
  // Once the queue is dry the samples still in flight go to the latency kernel at once (it iterates ~20x faster than
  // a fully occupied throughput wave); sweeps: tools/time_mid.py, tools/time_lpt3.py.
  pl.dump_threshold = ctx->dump_threshold >= 0 ? ctx->dump_threshold : 10;
  if (pl.handover) {
    const size_t in_flight = (size_t)pl.group_blocks * 10;
    pl.latency_blocks = (int)(in_flight < lat_cap ? in_flight : lat_cap);
  }
  return pl;
}

// workspaces owned by the context; they grow outside any stream capture (the first call at a size is never captured)
static int ensure_pool(ccmp_ctx *ctx, size_t records)
{
  if (ctx->pool_cap >= records) return CCMP_OK;
  if (ctx->pool) (void)hipFree(ctx->pool);
  ctx->pool = nullptr;
  ctx->pool_cap = 0;
  HIP_TRY(hipMalloc((void **)&ctx->pool, records * 18 * sizeof(double)));
  ctx->pool_cap = records;
  return CCMP_OK;
}
static int ensure_lpt_buffers(ccmp_ctx *ctx, size_t B)
{
  if (ctx->lpt_cap >= B) return CCMP_OK;
  if (ctx->lpt_buf) (void)hipFree(ctx->lpt_buf);
  ctx->lpt_buf = nullptr;
  ctx->lpt_cap = 0;
  HIP_TRY(hipMalloc(&ctx->lpt_buf, ((B * 2 + 255) & ~(size_t)255) + 4096 + B * 4)); // pred u16 | hist 1024 x u32 | order u32
  ctx->lpt_cap = B;
  return CCMP_OK;
}

static int project_common(ccmp_ctx *ctx, const ccmp_problem *p, int mode, const double *q_in, double *q_out,
                          uint8_t *ok, uint16_t *iters, double *q_ambient, size_t B, uint64_t seed, uint64_t first,
                          void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q_out || !ok || (mode == 0 && !q_in)) return CCMP_EINVAL;
  if ((((uintptr_t)q_in) | ((uintptr_t)q_out)) & 15u) return CCMP_EINVAL; // rows are moved in 16-byte pieces
  if (p->jacobian_mode != CCMP_JAC_FD) {
    const int nblocks = projector_blocks(ctx, B, 64, 4);
    HIP_TRY(ccmp_launch_project_fast(&K, mode, q_in, q_out, ok, iters, q_ambient, B, ctx->queue, seed, first, nblocks, st));
    return CCMP_OK;
  }
  const FdPlan pl = plan_fd_batch(ctx, B, ctx->order != nullptr);
  // queue[0]: sample queue of the throughput kernel; queue[1]: pool fill count; queue[2]: read head of the latency kernel
  unsigned long long *const q_group = ctx->queue, *const q_pool_count = ctx->queue + 1, *const q_latency = ctx->queue + 2;
  if (!pl.latency_static) HIP_TRY(hipMemsetAsync(ctx->queue, 0, 4 * sizeof(unsigned long long), st));

  if (pl.group_blocks == 0) { // small batches and single states
    if (ctx->flat_kernel)
      HIP_TRY(ccmp_launch_project_flat(&K, mode, q_in, q_out, ok, iters, q_ambient, B, pl.latency_static ? nullptr : q_latency, seed, first,
                                       ctx->pool, q_pool_count, mode, pl.latency_blocks, st));
    else
      HIP_TRY(ccmp_launch_project_wave(&K, mode, q_in, q_out, ok, iters, q_ambient, B, q_latency, seed, first, ctx->pool, q_pool_count,
                                       mode, pl.latency_blocks, st));
    return CCMP_OK;
  }

  const unsigned int *order = ctx->order;
  if (pl.scout) { // FP32 scout pass -> predicted iteration counts -> descending counting sort -> processing order
    int rc = ensure_lpt_buffers(ctx, B);
    if (rc != CCMP_OK) return rc;
    char *base = (char *)ctx->lpt_buf;
    uint16_t *pred = (uint16_t *)base;
    unsigned int *hist = (unsigned int *)(base + ((ctx->lpt_cap * 2 + 255) & ~(size_t)255));
    unsigned int *ord = (unsigned int *)((char *)hist + 4096);
    // one 256-thread block per CU, 4 samples per lane at 262144: more lanes only lengthen the per-wave maximum
    HIP_TRY(ccmp_launch_scout_order(&K, mode, q_in, B, pred, hist, ord, ctx->queue + 5, seed, first, ctx->num_cus, st));
    order = ord;
  }
  if (pl.handover) {
    int rc = ensure_pool(ctx, (size_t)pl.group_blocks * 10);
    if (rc != CCMP_OK) return rc;
  }
  HIP_TRY(ccmp_launch_project_group(&K, mode, q_in, q_out, ok, iters, q_ambient, B, q_group, seed, first, pl.group_blocks,
                                    pl.handover ? ctx->pool : nullptr, pl.dump_threshold, order, st));
  if (pl.handover) { // the pool's fill count is read on the device: the latency kernel's surplus blocks exit at once
    if (ctx->flat_kernel)
      HIP_TRY(ccmp_launch_project_flat(&K, 2, q_in, q_out, ok, iters, q_ambient, B, q_latency, seed, first, ctx->pool, q_pool_count, mode,
                                       pl.latency_blocks, st));
    else
      HIP_TRY(ccmp_launch_project_wave(&K, 2, q_in, q_out, ok, iters, q_ambient, B, q_latency, seed, first, ctx->pool, q_pool_count, mode,
                                       pl.latency_blocks, st));
  }
  return CCMP_OK;
}

int ccmp_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q_in, double *q_out, uint8_t *ok,
                       uint16_t *iters, size_t B, void *hip_stream)
{
  return project_common(ctx, p, 0, q_in, q_out, ok, iters, nullptr, B, 0, 0, hip_stream);
}

int ccmp_sample_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index, double *q_out,
                              uint8_t *ok, uint16_t *iters, double *q_ambient, size_t B, void *hip_stream)
{
  return project_common(ctx, p, 1, nullptr, q_out, ok, iters, q_ambient, B, seed, first_index, hip_stream);
}

static int sample_ref_common(ccmp_ctx *ctx, const ccmp_problem *p, int kind, uint64_t seed, uint64_t first_index,
                             const double *ref, int ref_stride, double param, double *q_out, uint8_t *ok, uint16_t *iters,
                             double *q_ambient, size_t B, void *hip_stream)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!ref || !q_out || !ok || (ref_stride != 0 && ref_stride != 14) || !(param >= 0)) return CCMP_EINVAL;
  {
    CCMP_PROLOGUE();
    HIP_TRY(ccmp_launch_ambient_ref(&K, kind, seed, first_index, ref, ref_stride, param, q_out, B, st));
    if (q_ambient) HIP_TRY(hipMemcpyAsync(q_ambient, q_out, B * 14 * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  int rc = ccmp_project_batch(ctx, p, q_out, q_out, ok, iters, B, hip_stream);
  if (rc != CCMP_OK) return rc;
  return ccmp_enforce_bounds_batch(ctx, q_out, B, hip_stream);
}

int ccmp_sample_near_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                                   const double *near, int near_stride, double distance, double *q_out, uint8_t *ok,
                                   uint16_t *iters, double *q_ambient, size_t B, void *hip_stream)
{
  return sample_ref_common(ctx, p, 0, seed, first_index, near, near_stride, distance, q_out, ok, iters, q_ambient, B, hip_stream);
}

int ccmp_sample_gaussian_project_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                                       const double *mean, int mean_stride, double std_dev, double *q_out, uint8_t *ok,
                                       uint16_t *iters, double *q_ambient, size_t B, void *hip_stream)
{
  return sample_ref_common(ctx, p, 1, seed, first_index, mean, mean_stride, std_dev, q_out, ok, iters, q_ambient, B, hip_stream);
}

int ccmp_compute_t_wo_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, int q_stride, double *t_wo, size_t B,
                            void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q || !t_wo || q_stride < 7) return CCMP_EINVAL;
  HIP_TRY(ccmp_launch_t_wo(&K, q, q_stride, t_wo, B, st));
  return CCMP_OK;
}

int ccmp_geodesic_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                        double *states, int32_t *n_states, uint8_t *ok, int32_t *newton_iters, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (E == 0) return CCMP_OK;
  if (!from || !to || !states || !n_states || !ok || max_states < 1) return CCMP_EINVAL;
  if (!(p->delta > 0) || !(p->lambda > 0)) return CCMP_EINVAL;
  if (p->jacobian_mode != CCMP_JAC_FD) return CCMP_EINVAL; // the extend step exists in reference arithmetic only
  // one 128-thread block per edge; the hardware dispatcher balances edges of different length (the kernel strides
  // over the edges if the grid is capped)
  const size_t nb = E < ((size_t)1 << 20) ? E : ((size_t)1 << 20);
  HIP_TRY(ccmp_launch_geodesic(&K, p->delta, p->lambda, from, to, E, max_states, states, n_states, ok, newton_iters, (int)nb, st));
  return CCMP_OK;
}

int ccmp_is_satisfied_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q || !ok) return CCMP_EINVAL;
  HIP_TRY(ccmp_launch_is_satisfied(&K, q, ok, B, st));
  return CCMP_OK;
}

int ccmp_joint_valid_batch(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q || !ok) return CCMP_EINVAL;
  HIP_TRY(ccmp_launch_joint_valid(&K, q, ok, B, st));
  return CCMP_OK;
}

int ccmp_ambient_uniform_batch(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index, double *q_out,
                               size_t B, void *hip_stream)
{
  CCMP_PROLOGUE();
  if (B == 0) return CCMP_OK;
  if (!q_out) return CCMP_EINVAL;
  HIP_TRY(ccmp_launch_ambient_uniform(&K, seed, first_index, q_out, B, st));
  return CCMP_OK;
}

int ccmp_enforce_bounds_batch(ccmp_ctx *ctx, double *q, size_t B, void *hip_stream)
{
  if (!ctx) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(ccmp_launch_enforce_bounds(q, B, st));
  return CCMP_OK;
}

int ccmp_compact_valid(ccmp_ctx *ctx, const double *q, const uint8_t *ok, size_t B, double *q_valid, uint64_t *count_dev,
                       void *hip_stream)
{
  if (!ctx || !count_dev) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  hipStream_t st = (hipStream_t)hip_stream;
  if (B == 0) {
    HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(uint64_t), st));
    return CCMP_OK;
  }
  if (!q || !ok || !q_valid) return CCMP_EINVAL;
  const size_t nblocks = (B + 255) / 256;
  if (ctx->scan_cap < nblocks) {
    // growth happens outside any capture: callers that capture graphs call once un-captured first
    if (ctx->scan) (void)hipFree(ctx->scan);
    ctx->scan = nullptr;
    ctx->scan_cap = 0;
    HIP_TRY(hipMalloc((void **)&ctx->scan, nblocks * sizeof(unsigned int)));
    ctx->scan_cap = nblocks;
  }
  HIP_TRY(ccmp_launch_compact(q, ok, B, q_valid, ctx->scan, (unsigned long long *)count_dev, st));
  return CCMP_OK;
}

int ccmp_detmath_probe(ccmp_ctx *ctx, const double *x_dev, const double *y_dev, double *out_dev, size_t n, void *hip_stream)
{
  if (!ctx || !x_dev || !y_dev || !out_dev) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  hipStream_t st = (hipStream_t)hip_stream;
  if (n == 0) return CCMP_OK;
  HIP_TRY(ccmp_launch_detmath_probe(x_dev, y_dev, out_dev, n, st));
  return CCMP_OK;
}

// ---- host-pointer conveniences ----------------------------------------------------------------------
int ccmp_project_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q_in, double *q_out, uint8_t *ok,
                      uint16_t *iters, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q_in || !q_out || !ok) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_ok = (qb + 255) & ~(size_t)255;
  const size_t off_it = (off_ok + B + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_it + B * sizeof(uint16_t));
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, q_in, qb)) != CCMP_OK) return rc;
  rc = ccmp_project_batch(ctx, p, (const double *)io.dev, (double *)io.dev, (uint8_t *)(io.dev + off_ok), (uint16_t *)(io.dev + off_it), B,
                          ctx->stream);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.out(q_out, 0, qb)) != CCMP_OK) return rc;
  if ((rc = io.out(ok, off_ok, B)) != CCMP_OK) return rc;
  if (iters && (rc = io.out(iters, off_it, B * sizeof(uint16_t))) != CCMP_OK) return rc;
  return io.finish();
}

int ccmp_function_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, double *f, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q || !f) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_f = (qb + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_f + B * 2 * sizeof(double));
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, q, qb)) != CCMP_OK) return rc;
  rc = ccmp_function_batch(ctx, p, (const double *)io.dev, (double *)(io.dev + off_f), B, ctx->stream);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.out(f, off_f, B * 2 * sizeof(double))) != CCMP_OK) return rc;
  return io.finish();
}

int ccmp_is_satisfied_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q || !ok) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_ok = (qb + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_ok + B);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, q, qb)) != CCMP_OK) return rc;
  rc = ccmp_is_satisfied_batch(ctx, p, (const double *)io.dev, (uint8_t *)(io.dev + off_ok), B, ctx->stream);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.out(ok, off_ok, B)) != CCMP_OK) return rc;
  return io.finish();
}

int ccmp_joint_valid_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *q, uint8_t *ok, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q || !ok) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_ok = (qb + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_ok + B);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, q, qb)) != CCMP_OK) return rc;
  rc = ccmp_joint_valid_batch(ctx, p, (const double *)io.dev, (uint8_t *)(io.dev + off_ok), B, ctx->stream);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.out(ok, off_ok, B)) != CCMP_OK) return rc;
  return io.finish();
}

int ccmp_sample_project_host(ccmp_ctx *ctx, const ccmp_problem *p, uint64_t seed, uint64_t first_index, double *q_out,
                             uint8_t *ok, uint16_t *iters, size_t B)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q_out || !ok) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t qb = B * 14 * sizeof(double);
  const size_t off_ok = (qb + 255) & ~(size_t)255;
  const size_t off_it = (off_ok + B + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_it + B * sizeof(uint16_t));
  if (rc != CCMP_OK) return rc;
  rc = ccmp_sample_project_batch(ctx, p, seed, first_index, (double *)io.dev, (uint8_t *)(io.dev + off_ok), (uint16_t *)(io.dev + off_it),
                                 nullptr, B, ctx->stream);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.out(q_out, 0, qb)) != CCMP_OK) return rc;
  if ((rc = io.out(ok, off_ok, B)) != CCMP_OK) return rc;
  if (iters && (rc = io.out(iters, off_it, B * sizeof(uint16_t))) != CCMP_OK) return rc;
  return io.finish();
}

int ccmp_geodesic_host(ccmp_ctx *ctx, const ccmp_problem *p, const double *from, const double *to, size_t E, int max_states,
                       double *states, int32_t *n_states, uint8_t *ok)
{
  if (!ctx || !p) return CCMP_EINVAL;
  if (E == 0) return CCMP_OK;
  if (!from || !to || !states || !n_states || !ok || max_states < 1) return CCMP_EINVAL;
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return CCMP_ENODEV;
  const size_t eb = E * 14 * sizeof(double);
  const size_t sb = E * (size_t)max_states * 14 * sizeof(double);
  const size_t off_to = (eb + 255) & ~(size_t)255;
  const size_t off_st = (off_to + eb + 255) & ~(size_t)255;
  const size_t off_n = (off_st + sb + 255) & ~(size_t)255;
  const size_t off_ok = (off_n + E * sizeof(int32_t) + 255) & ~(size_t)255;
  HostIO io(ctx);
  int rc = io.begin(off_ok + E);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.in(0, from, eb)) != CCMP_OK) return rc;
  if ((rc = io.in(off_to, to, eb)) != CCMP_OK) return rc;
  rc = ccmp_geodesic_batch(ctx, p, (const double *)io.dev, (const double *)(io.dev + off_to), E, max_states, (double *)(io.dev + off_st),
                           (int32_t *)(io.dev + off_n), (uint8_t *)(io.dev + off_ok), nullptr, ctx->stream);
  if (rc != CCMP_OK) return rc;
  if ((rc = io.out(states, off_st, sb)) != CCMP_OK) return rc;
  if ((rc = io.out(n_states, off_n, E * sizeof(int32_t))) != CCMP_OK) return rc;
  if ((rc = io.out(ok, off_ok, E)) != CCMP_OK) return rc;
  return io.finish();
}

static int sharded_common(ccmp_ctx *const *ctxs, int n, const ccmp_problem *p, int mode, const double *q_in, double *q_out,
                          uint8_t *ok, uint16_t *iters, uint64_t seed, uint64_t first_index, size_t B)
{
  if (!ctxs || n < 1 || n > 64 || !p) return CCMP_EINVAL;
  for (int g = 0; g < n; g++)
    if (!ctxs[g]) return CCMP_EINVAL;
  if (B == 0) return CCMP_OK;
  if (!q_out || !ok || (mode == 0 && !q_in)) return CCMP_EINVAL;
  struct Shard { size_t lo, hi, off_ok, off_it; bool busy; } sh[64];
  for (int g = 0; g < n; g++) sh[g] = Shard{0, 0, 0, 0, false};
  int rc = CCMP_OK;
  // phase 1: upload and launch on every context's stream.  Copies from pageable host memory return once the data is
  // staged, so the next GPU starts while this one computes.
  for (int g = 0; g < n && rc == CCMP_OK; g++) {
    const size_t base = B / (size_t)n, rem = B % (size_t)n;
    sh[g].lo = (size_t)g * base + ((size_t)g < rem ? (size_t)g : rem);
    sh[g].hi = sh[g].lo + base + ((size_t)g < rem ? 1 : 0);
    const size_t nb = sh[g].hi - sh[g].lo;
    if (nb == 0) continue;
    ccmp_ctx *ctx = ctxs[g];
    DeviceGuard guard(ctx->device);
    if (!guard.ok) { rc = CCMP_ENODEV; break; }
    const size_t qb = nb * 14 * sizeof(double);
    sh[g].off_ok = (qb + 255) & ~(size_t)255;
    sh[g].off_it = (sh[g].off_ok + nb + 255) & ~(size_t)255;
    if ((rc = ensure_stage(ctx, sh[g].off_it + nb * sizeof(uint16_t))) != CCMP_OK) break;
    char *stage = (char *)ctx->stage;
    sh[g].busy = true;
    if (mode == 0) {
      hipError_t e = hipMemcpyAsync(stage, q_in + sh[g].lo * 14, qb, hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) { rc = hip_fail(e, "hipMemcpyAsync(H2D shard)"); break; }
      rc = ccmp_project_batch(ctx, p, (const double *)stage, (double *)stage, (uint8_t *)(stage + sh[g].off_ok),
                              (uint16_t *)(stage + sh[g].off_it), nb, ctx->stream);
    } else {
      rc = ccmp_sample_project_batch(ctx, p, seed, first_index + sh[g].lo, (double *)stage, (uint8_t *)(stage + sh[g].off_ok),
                                     (uint16_t *)(stage + sh[g].off_it), nullptr, nb, ctx->stream);
    }
  }
  // phase 2: downloads (a copy into pageable memory waits for its shard's kernels; the other GPUs keep computing)
  for (int g = 0; g < n && rc == CCMP_OK; g++) {
    if (!sh[g].busy) continue;
    ccmp_ctx *ctx = ctxs[g];
    DeviceGuard guard(ctx->device);
    const size_t nb = sh[g].hi - sh[g].lo;
    const char *stage = (const char *)ctx->stage;
    hipError_t e = hipMemcpyAsync(q_out + sh[g].lo * 14, stage, nb * 14 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ok + sh[g].lo, stage + sh[g].off_ok, nb, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && iters)
      e = hipMemcpyAsync(iters + sh[g].lo, stage + sh[g].off_it, nb * sizeof(uint16_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemcpyAsync(D2H shard)");
  }
  // phase 3: wait for every stream that was given work, also on the error path (the caller's buffers must be quiet)
  for (int g = 0; g < n; g++) {
    if (!sh[g].busy) continue;
    DeviceGuard guard(ctxs[g]->device);
    hipError_t e = hipStreamSynchronize(ctxs[g]->stream);
    if (e != hipSuccess && rc == CCMP_OK) rc = hip_fail(e, "hipStreamSynchronize(shard)");
  }
  return rc;
}

int ccmp_project_sharded_host(ccmp_ctx *const *ctxs, int n, const ccmp_problem *p, const double *q_in, double *q_out,
                              uint8_t *ok, uint16_t *iters, size_t B)
{
  return sharded_common(ctxs, n, p, 0, q_in, q_out, ok, iters, 0, 0, B);
}

int ccmp_sample_project_sharded_host(ccmp_ctx *const *ctxs, int n, const ccmp_problem *p, uint64_t seed, uint64_t first_index,
                                     double *q_out, uint8_t *ok, uint16_t *iters, size_t B)
{
  return sharded_common(ctxs, n, p, 1, nullptr, q_out, ok, iters, seed, first_index, B);
}

} // extern "C"
