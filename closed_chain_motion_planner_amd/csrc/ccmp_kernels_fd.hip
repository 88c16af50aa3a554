// ccmp_kernels_fd.hip — gfx950 kernels in the canonical (bit-reproducible) rounding model.
//
// Built with -ffp-contract=off -DCCMP_USE_FMA: the only fused operations are the CCMP_FMA calls in
// ccmp_detmath.h / ccmp_kin.h / this file, the same ones oracle/ccmp_oracle.c (det build) performs,
// so every kernel here is expected to agree with the CPU oracle BIT FOR BIT.
//
// Hot kernel: project_fd_kernel — KinematicChainConstraint::project with the reference's
// finite-difference Jacobian (include/closed_chain_motion_planner/base/constraints/
// ConstraintFunction.h:57-82 + OMPL's default Constraint::jacobian).
//
// Work decomposition: one wavefront = 10 independent samples x 6 lanes.
// The 6 lanes of a group are the 6 evaluation points of OMPL's 7-point central stencil
// (+h,+2h,+3h,-h,-2h,-3h) of ONE Jacobian column; the wave walks the 14 columns in lock-step, so
// arm/joint indices are wave-uniform and every kinematic constant is an SGPR operand.  A perturbed
// evaluation reuses the unperturbed prefix of the chain (bit-identical to recomputing it) from LDS
// and recomputes only the suffix.  Joint state, sines/cosines, prefix frames, tool poses, stencil
// values and the 2x14 Jacobian of each sample are staged in LDS (one 165-double record per group; odd
// stride => the 10 groups hit distinct banks, lanes of a group broadcast).  Groups pull samples from
// a global atomic queue (optionally in longest-predicted-first order, ccmp_kernels_scout.hip), so a
// group whose sample converges early refills while its neighbours keep iterating (iteration counts
// spread 15..250); when the queue runs dry the samples still in flight are handed to the
// wave-per-sample kernel (ccmp_kernels_wave.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccmp_fd_common.h"

using namespace ccmp;

namespace {

constexpr int kGroup = 6;             // lanes per sample = stencil evaluations per column
constexpr int kGroupsPerWave = 10;    // 60 of 64 lanes busy
// LDS record of one group (in doubles).  The prefix frames are kept for ONE arm at a time (the
// second arm's chain is re-run before its columns: +343 operations per iteration, -84 doubles of
// LDS per sample).  A consumed prefix slot (12 doubles) then holds the 6 lanes' residual pairs of
// that column until the arm's stencil is combined; arm 0's Jacobian entries go to kJ0, arm 1's into
// arm 0's (by then dead) sin/cos slots.  165 x 8 B x 10 groups = 13.2 KB per wave -> 12 waves per CU.
constexpr int kX = 0;                 // x[14]        current iterate
constexpr int kSC = 14;               // sc[14][2]    sin, cos of every joint of x
constexpr int kPre = 42;              // pre[7][12]   chain frame in front of joint j of the current arm: R(9), o(3);
                                      //              once consumed: (f0, f1) of the 6 stencil points of column j
constexpr int kEE = 126;              // ee[2][12]    world tool pose of each arm at x: R(9), p(3)
constexpr int kJ0 = 150;              // J[:, 0..6]   interleaved (row0, row1) per column of arm 0
constexpr int kRec = 165;             // 164 used; odd stride keeps the 10 groups on distinct LDS banks.  13200 B per wave:
                                      // the allocation granule is 512 B, 12 x 13312 = 159744 <= 160 KiB (two more doubles
                                      // per group would round to 13824 and cost a wave per CU)

#ifndef CCMP_FD_WAVES_PER_SIMD
#define CCMP_FD_WAVES_PER_SIMD 3
#endif
#ifndef CCMP_FD_STOCK_COLUMNS
#define CCMP_FD_STOCK_COLUMNS 1 // STOCK instantiation: Jacobian columns skip the products with the stock Panda's exact zeros
#endif
#ifndef CCMP_FD_ROWS
#define CCMP_FD_ROWS 1 // STOCK instantiation: the chains at x by matrix rows, three lanes per arm (0 = every lane runs both chains whole)
#endif

// One arm's chain at x (sines/cosines from LDS), joint indices at compile time (the STOCK instantiation skips the
// products with the stock Panda's exact zeros, ccmp_kin.h).  With STORE the writer lane keeps the frame in front of
// every joint (R before the joint's rotation, o including the joint's offset) in LDS.
template <int ARM, bool STORE, bool STOCK, int I>
__device__ __forceinline__ void chain_at_x_from(const ccmp_consts &K, double *rec, bool writer, double *R, double *o)
{
  if constexpr (I < 7) {
    double Rn[9];
    mulvec_acc_nz<STOCK ? kStockOff[I] : 7>(R, K.offset[ARM][I], o);
    if (STORE && writer) {
#pragma unroll
      for (int k = 0; k < 9; k++) rec[kPre + I * 12 + k] = R[k];
#pragma unroll
      for (int k = 0; k < 3; k++) rec[kPre + I * 12 + 9 + k] = o[k];
    }
    chain_rot<I, STOCK>(K.axis[ARM][I], K.aprod[ARM][I], rec[kSC + 2 * (ARM * 7 + I)], rec[kSC + 2 * (ARM * 7 + I) + 1], R, Rn);
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
    chain_at_x_from<ARM, STORE, STOCK, I + 1>(K, rec, writer, R, o);
  }
}
template <int ARM, bool STORE, bool STOCK>
__device__ __forceinline__ void chain_at_x(const ccmp_consts &K, double *rec, bool writer, double *T)
{
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, o[3] = {0, 0, 0};
  if constexpr (STOCK) {
    // unrolled: 5.6 k instructions in all, 135 VGPRs (the general formulas unrolled the same way spill)
    chain_at_x_from<ARM, STORE, true, 0>(K, rec, writer, R, o);
  } else {
    for (int i = 0; i < 7; i++) {
      double Rj[9], Rn[9];
      mulvec_acc(R, K.offset[ARM][i], o);
      if (STORE && writer) {
#pragma unroll
        for (int k = 0; k < 9; k++) rec[kPre + i * 12 + k] = R[k];
#pragma unroll
        for (int k = 0; k < 3; k++) rec[kPre + i * 12 + 9 + k] = o[k];
      }
      rot_sc(K.axis[ARM][i], K.aprod[ARM][i], rec[kSC + 2 * (ARM * 7 + i)], rec[kSC + 2 * (ARM * 7 + i) + 1], Rj);
      mul33(R, Rj, Rn);
#pragma unroll
      for (int k = 0; k < 9; k++) R[k] = Rn[k];
    }
  }
  tool_pose_t<STOCK>(K, ARM, R, o, &T[0], &T[9]);
}

// ---- the chains at x, one matrix ROW per lane (STOCK instantiation: twin arms, diag(+-1) base frames) ---------------
// Row r of (R * Rj) needs row r of R only, and o[r] += R[r,:] * offset likewise: lane `row` of an arm's three lanes
// carries one row of the running frame (3 + 1 doubles) and does a third of every product — per element the same
// operations on the same operands as mul33 / mulvec_acc, hence the same bits.  Lanes 0..2 of a group run arm 0,
// lanes 3..5 arm 1, side by side (both arms read arm 0's constants: they are bit-identical, K.twin_arms); only the
// joint's rotation matrix (13 operations) is computed by every lane.  175 + 19 operations per lane for BOTH chains
// and tool poses instead of 2 x (343 + 48) when every lane ran both chains whole.  The lanes of arm `store_arm` keep
// the frame in front of every joint in LDS (R row before the joint's rotation, o including the joint's offset) for
// that arm's Jacobian columns; with TOOL every lane also leaves its row of the arm's world tool pose in LDS.
template <bool TOOL, int I>
__device__ __forceinline__ void chain_rows_from(const ccmp_consts &K, double *rec, const double *sc, int row, bool store,
                                                double &R0, double &R1, double &R2, double &o)
{
  if constexpr (I < 7) {
    constexpr int NZ = kStockOff[I];
    if (NZ & 1) o = CCMP_FMA(R0, K.offset[0][I][0], o);
    if (NZ & 2) o = CCMP_FMA(R1, K.offset[0][I][1], o);
    if (NZ & 4) o = CCMP_FMA(R2, K.offset[0][I][2], o);
    if (store) {
      rec[kPre + I * 12 + 3 * row] = R0;
      rec[kPre + I * 12 + 3 * row + 1] = R1;
      rec[kPre + I * 12 + 3 * row + 2] = R2;
      rec[kPre + I * 12 + 9 + row] = o;
    }
    const double s = sc[2 * I], c = sc[2 * I + 1];
    double n0, n1, n2;
    if constexpr (kStockZ[I] != 0) { // mul_zrot, one row
      const double t = 1.0 - c;
      const double w = t + c;
      const double ns = -s;
      n0 = CCMP_FMA(R1, s, R0 * c);
      n1 = CCMP_FMA(R1, c, R0 * ns);
      n2 = R2 * w;
    } else { // rot_sc + one row of mul33
      double Rj[9];
      rot_sc(K.axis[0][I], K.aprod[0][I], s, c, Rj);
      n0 = dot3(R0, Rj[0], R1, Rj[3], R2, Rj[6]);
      n1 = dot3(R0, Rj[1], R1, Rj[4], R2, Rj[7]);
      n2 = dot3(R0, Rj[2], R1, Rj[5], R2, Rj[8]);
    }
    R0 = n0; R1 = n1; R2 = n2;
    chain_rows_from<TOOL, I + 1>(K, rec, sc, row, store, R0, R1, R2, o);
  }
}
template <bool TOOL>
__device__ __forceinline__ void chain_rows(const ccmp_consts &K, double *rec, int arm, int row, bool live, int store_arm,
                                           double d_lane, double bp_lane)
{
  double R0 = row == 0 ? 1.0 : 0.0, R1 = row == 1 ? 1.0 : 0.0, R2 = row == 2 ? 1.0 : 0.0, o = 0.0;
  chain_rows_from<TOOL, 0>(K, rec, rec + kSC + 14 * arm, row, live && arm == store_arm, R0, R1, R2, o);
  if constexpr (TOOL) { // tool_pose_t<true>, diag(+-1) base frame, one row
    double pf = o;
    if (kStockEe & 1) pf = CCMP_FMA(R0, K.ee[0][0], pf);
    if (kStockEe & 2) pf = CCMP_FMA(R1, K.ee[0][1], pf);
    if (kStockEe & 4) pf = CCMP_FMA(R2, K.ee[0][2], pf);
    const double *Rt = K.R_tool[0];
    const double f0 = dot3(R0, Rt[0], R1, Rt[3], R2, Rt[6]);
    const double f1 = dot3(R0, Rt[1], R1, Rt[4], R2, Rt[7]);
    const double f2 = dot3(R0, Rt[2], R1, Rt[5], R2, Rt[8]);
    if (live) {
      double *T = rec + kEE + 12 * arm;
      T[3 * row] = d_lane * f0;
      T[3 * row + 1] = d_lane * f1;
      T[3 * row + 2] = d_lane * f2;
      T[9 + row] = CCMP_FMA(d_lane, pf, bp_lane);
    }
  }
}

// ---- suffix steps of the stock Panda (STOCK instantiation) -----------------------------------------------------------
// The joints alternate: 0, 2, 4 rotate about exactly (0, 0, 1); 1, 3, 5 and 6 about a general axis; most offset
// components are exactly zero (ccmp_kin.h: kStockZ, kStockOff).  With run-time (wave-uniform) joint indices the steps
// come in three shapes, each skipping only products with components that are exact zeros for EVERY joint it serves
// (a product with an exact zero that is not skipped adds an exact zero: same bits either way):
//   z step       joints 2, 4     o += R[:,0] off.x + R[:,2] off.z  (joint 2: off.x == 0),  R = R * Rz(q)
//   general step joints 1, 3, 5  o += R[:,0] off.x                 (joints 1, 5: off.x == 0), R = R * Rot(axis, q)
//   last step    joint 6         o += R[:,0] off.x,                                           R = R * Rot(axis, q)
__device__ __forceinline__ void stock_z_step(const ccmp_consts &K, int i, double s, double c, const double *R, double *Rn, double *o)
{
  const double ox = K.offset[0][i][0], oz = K.offset[0][i][2];
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = CCMP_FMA(R[3 * r + 2], oz, CCMP_FMA(R[3 * r], ox, o[r]));
  mul_zrot(R, s, c, Rn);
}
__device__ __forceinline__ void stock_g_step(const ccmp_consts &K, int i, double s, double c, const double *R, double *Rn, double *o)
{
  const double ox = K.offset[0][i][0];
#pragma unroll
  for (int r = 0; r < 3; r++) o[r] = CCMP_FMA(R[3 * r], ox, o[r]);
  double Rj[9];
  rot_sc(K.axis[0][i], K.aprod[0][i], s, c, Rj);
  mul33(R, Rj, Rn);
}

// OMPL's default Constraint::jacobian, evaluation part, for the 7 columns of one arm: each lane evaluates
// its stencil point of column j from the cached prefix frame and parks the residual pair in LDS.
template <int ARM, bool STOCK>
__device__ __forceinline__ void jacobian_columns(const ccmp_consts &K, double *rec, bool live, int r, bool plus, int nstep)
{
  double To[12]; // the other arm's (unperturbed) tool pose: 24 VGPRs that save 12 LDS reads per column (-6.5 %, A/B)
#pragma unroll
  for (int k = 0; k < 12; k++) To[k] = rec[kEE + (1 - ARM) * 12 + k];
  for (int j = 0; j < 7; j++) {
    const double xj = rec[kX + ARM * 7 + j];
    const double axj = ccmp_abs(xj);
    const double h = 1.4901161193847656e-08 * (axj >= 1 ? axj : 1); // sqrt(eps)*max(1,|x_j|)
    const double hh = plus ? h : -h;
    double y = xj + hh;                 // y1[j] += h   /  y2[j] -= h
    if (nstep >= 2) y = y + hh;
    if (nstep >= 3) y = y + hh;
    double R[9], o[3], s, c;
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = rec[kPre + j * 12 + k];
#pragma unroll
    for (int k = 0; k < 3; k++) o[k] = rec[kPre + j * 12 + 9 + k];
    ccmp_sincos(y, &s, &c);
    const double *sc = rec + kSC + 2 * ARM * 7;
    if constexpr (STOCK && CCMP_FD_STOCK_COLUMNS) {
      // twin arms: arm 0's constants serve both.  Joint indices are wave-uniform scalars; the branches are scalar.
      // The frame ends every step in R (R -> Rn -> R, or in place): no register copies between steps.
      double Rn[9];
      int i = j + 1;
      if ((j & 1) == 0 && j < 6) { // joints 0, 2, 4: R = R * Rz(y), row by row in place (mul_zrot's operations)
        const double t = 1.0 - c;
        const double w = t + c;
        const double ns = -s;
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
          const double a = R[3 * rr], b = R[3 * rr + 1];
          R[3 * rr] = CCMP_FMA(b, s, a * c);
          R[3 * rr + 1] = CCMP_FMA(b, c, a * ns);
          R[3 * rr + 2] = R[3 * rr + 2] * w;
        }
      } else {
        double Rj[9];
        rot_sc(K.axis[0][j], K.aprod[0][j], s, c, Rj);
        mul33(R, Rj, Rn);
        if (j < 6) { // j = 1, 3, 5: the next joint (2, 4 or 6) takes the frame back to R
          if (i < 6) stock_z_step(K, i, sc[2 * i], sc[2 * i + 1], Rn, R, o);
          else stock_g_step(K, 6, sc[12], sc[13], Rn, R, o);
          i++;
        } else { // j = 6: no suffix
#pragma unroll
          for (int k = 0; k < 9; k++) R[k] = Rn[k];
        }
      }
      for (; i < 5; i += 2) { // (1, 2), (3, 4): general step then z step, R -> Rn -> R
        stock_g_step(K, i, sc[2 * i], sc[2 * i + 1], R, Rn, o);
        stock_z_step(K, i + 1, sc[2 * i + 2], sc[2 * i + 3], Rn, R, o);
      }
      if (i == 5) { // (5, 6): joint 5 has no offset at all
        double Rj[9];
        rot_sc(K.axis[0][5], K.aprod[0][5], sc[10], sc[11], Rj);
        mul33(R, Rj, Rn);
        stock_g_step(K, 6, sc[12], sc[13], Rn, R, o);
      }
    } else {
      {
        double Rj[9], Rn[9];
        rot_sc(K.axis[ARM][j], K.aprod[ARM][j], s, c, Rj);
        mul33(R, Rj, Rn);
#pragma unroll
        for (int k = 0; k < 9; k++) R[k] = Rn[k];
      }
      {
        // suffix joints two at a time, R -> Rn -> R: no register copies of the frame between steps (61 instead of
        // 76 instructions per joint; -0.6 % run time, A/B)
        int i = j + 1;
        if ((7 - i) & 1) {
          joint_step(K, ARM, i, sc[2 * i], sc[2 * i + 1], R, o);
          i++;
        }
        for (; i < 7; i += 2) {
          double Rj[9], Rn[9];
          mulvec_acc(R, K.offset[ARM][i], o);
          rot_sc(K.axis[ARM][i], K.aprod[ARM][i], sc[2 * i], sc[2 * i + 1], Rj);
          mul33(R, Rj, Rn);
          mulvec_acc(Rn, K.offset[ARM][i + 1], o);
          rot_sc(K.axis[ARM][i + 1], K.aprod[ARM][i + 1], sc[2 * i + 2], sc[2 * i + 3], Rj);
          mul33(Rn, Rj, R);
        }
      }
    }
#if defined(CCMP_FD_PROBE_MOV) || defined(CCMP_FD_PROBE_FMA)
    // probe (never in the product; tools/ab.py): what one more instruction per stencil evaluation costs THIS kernel at its
    // occupancy — N independent 32-bit moves or N independent FP64 FMAs beside the evaluation's own stream
    {
      int pv = j;
      double pa = xj, pb = xj;
#ifdef CCMP_FD_PROBE_MOV
#pragma unroll
      for (int k = 0; k < CCMP_FD_PROBE_MOV; k++) asm volatile("v_mov_b32 %0, 0x12345678" : "=v"(pv));
#endif
#ifdef CCMP_FD_PROBE_FMA
#pragma unroll
      for (int k = 0; k < CCMP_FD_PROBE_FMA / 2; k++) asm volatile("v_fma_f64 %0, %0, %2, %2\n v_fma_f64 %1, %1, %2, %2" : "+v"(pa), "+v"(pb) : "v"(h));
#endif
      if (pv == 0x7fffffff || pa + pb == 12345.678) rec[kX] = pa; // never true: keeps the probe alive
    }
#endif
    double Tw[12], t[2];
    tool_pose_t<STOCK>(K, ARM, R, o, &Tw[0], &Tw[9]);
    if (ARM == 0) chain_residual(K, &Tw[0], &Tw[9], &To[0], &To[9], t, nullptr, nullptr);
    else chain_residual(K, &To[0], &To[9], &Tw[0], &Tw[9], t, nullptr, nullptr);
    // park this evaluation in the prefix slot the group has just consumed (12 doubles = 6 lanes x (f0, f1));
    // the stencil is combined after the arm's 7 columns (stencil_combine) instead of through two dependent
    // ds_bpermute rounds per column
    if (live) {
      rec[kPre + 12 * j + 2 * r] = t[0];
      rec[kPre + 12 * j + 2 * r + 1] = t[1];
    }
  }
}

// out.col(j) = 1.5*m1 - 0.6*m2 + 0.1*m3 with m_s = (t1 - t2) / (y1[j] - y2[j]) for the 7 columns of one arm:
// 14 entries spread over the group's 6 lanes; y1/y2 are rebuilt by the same sequential adds the evaluations
// used.  Arm 0's entries go to kJ0, arm 1's into arm 0's (by now dead) sin/cos slots.
template <int ARM>
__device__ __forceinline__ void stencil_combine(double *rec, int r, bool live)
{
#pragma unroll
  for (int n = 0; n < 3; n++) {
    const int e = r + kGroup * n; // entry: column e >> 1, row e & 1
    if (e < 14 && live) {
      const int c = e >> 1, row = e & 1;
      const double xj = rec[kX + ARM * 7 + c];
      const double axj = ccmp_abs(xj);
      const double h = 1.4901161193847656e-08 * (axj >= 1 ? axj : 1);
      double y1 = xj, y2 = xj, m[3];
#pragma unroll
      for (int sidx = 0; sidx < 3; sidx++) {
        y1 = y1 + h;
        y2 = y2 - h;
        m[sidx] = (rec[kPre + 12 * c + 2 * sidx + row] - rec[kPre + 12 * c + 2 * (sidx + 3) + row]) / (y1 - y2);
      }
      const double Jv = CCMP_FMA(0.1, m[2], CCMP_FMA(-0.6, m[1], 1.5 * m[0]));
      if (ARM == 0) rec[kJ0 + 2 * c + row] = Jv;
      else rec[kSC + 2 * c + row] = Jv;
    }
  }
}

// MODE 0: project q_in -> q_out.  MODE 1: sampleUniform = ambient sample -> project -> enforceBounds.
template <int MODE, bool STOCK>
__global__ __launch_bounds__(64, CCMP_FD_WAVES_PER_SIMD) void project_fd_kernel(
    const ccmp_consts K, const double *__restrict__ q_in, double *__restrict__ q_out, uint8_t *__restrict__ ok_out,
    uint16_t *__restrict__ iters_out, double *__restrict__ q_ambient, unsigned long long B, unsigned long long *queue,
    unsigned long long seed, unsigned long long first_index, double *__restrict__ pool, unsigned long long *pool_count,
    int dump_threshold, const unsigned int *__restrict__ order, const uint16_t *__restrict__ pred, int long_remaining,
    unsigned long long pool_records)
{
  __shared__ double lds[kGroupsPerWave * kRec];
  const int lane = threadIdx.x;
  const int g = lane / kGroup;                // 0..10; lanes 60..63 form the idle "group 10"
  const int r = lane - kGroup * g;            // evaluation point 0..5 inside the group
  const bool live = g < kGroupsPerWave;
  const int leader = live ? kGroup * g : 0;
  double *rec = lds + (live ? g : 0) * kRec;  // idle lanes alias group 0 for reads, never write
  const bool writer = live && r == 0;
  const bool plus = r < 3;                    // y1 side of the stencil; r>=3 is the y2 side
  const int nstep = (plus ? r : r - 3) + 1;   // how many h-steps this lane's point is away
  // chains at x by rows (STOCK): lane r of a group carries row r % 3 of arm r / 3; its entry of the arm's diag(+-1)
  // base rotation and of the base translation are the only lane-dependent constants
  const int arm_l = plus ? 0 : 1, row_l = plus ? r : r - 3;
  double d_lane = 0.0, bp_lane = 0.0;
  if constexpr (STOCK) {
    d_lane = arm_l ? (row_l == 0 ? K.base_R[1][0] : (row_l == 1 ? K.base_R[1][4] : K.base_R[1][8]))
                   : (row_l == 0 ? K.base_R[0][0] : (row_l == 1 ? K.base_R[0][4] : K.base_R[0][8]));
    bp_lane = arm_l ? (row_l == 0 ? K.base_p[1][0] : (row_l == 1 ? K.base_p[1][1] : K.base_p[1][2]))
                    : (row_l == 0 ? K.base_p[0][0] : (row_l == 1 ? K.base_p[0][1] : K.base_p[0][2]));
  }

  unsigned long long idx = 0;
  int iter = 0, updates = 0;
  double norm1 = 0.0, norm2 = 0.0;
  bool active = false, drained = false;

  for (;;) {
    // ---- refill: groups without a sample pull the next index from the global queue ----------
    {
      const bool want = live && !active && !drained;
      unsigned long long t = 0;
      if (want && r == 0) t = atomicAdd(queue, 1ull);
      t = shfl_u64(t, leader);
      if (want) {
        if (t < B) {
          idx = order ? (unsigned long long)order[t] : t; // ticket -> sample: longest-predicted-first when scheduled
          active = true; iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
          if (MODE == 0) {
            // the 112-B row as seven 16-B pieces: lanes 0..5 of the group take 96 contiguous bytes in ONE
            // instruction, lane 0 the last piece (14 single-lane 8-B loads before: same data, 7x the requests)
            const double2 *row = reinterpret_cast<const double2 *>(q_in + idx * 14);
            const double2 a = row[r];
            rec[kX + 2 * r] = a.x;
            rec[kX + 2 * r + 1] = a.y;
            if (r == 0) {
              const double2 b = row[6];
              rec[kX + 12] = b.x;
              rec[kX + 13] = b.y;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 14; e++) {
              if (e % kGroup == r) { // e is a compile-time constant here: bounds come from SGPRs
                const double v = ambient_uniform(K, seed, first_index + idx, e);
                if (q_ambient) q_ambient[idx * 14 + e] = v;
                rec[kX + e] = v;
              }
            }
          }
        } else drained = true;
      }
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
    __syncthreads();
    // ---- hand-over: once the queue is empty the samples still in flight go to the straggler pool
    // (x, index, counters — everything else is recomputed from x) and the wave retires; the
    // wave-per-sample kernel finishes them.  State is dumped at the loop top, where the next thing
    // that happens to a sample is function(x) + the loop test — exactly where the other kernel starts.
    // every wave looks at the queue head once per iteration, so that all waves retire within one
    // iteration of the queue running dry (not only those that happen to finish a sample)
    bool dry = __builtin_amdgcn_ballot_w64(drained) != 0ull;
    if (pool != nullptr && !dry) {
      unsigned long long head = 0;
      if (lane == 0) head = __hip_atomic_load(queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      head = shfl_u64(head, 0);
      dry = head >= B;
    }
    // occupancy-driven hand-over (dump_threshold > 10: its excess over 10 is a percentage): with the queue dry the waves
    // keep iterating while the samples in flight — B minus the finished count in queue[3] — still fill that share of
    // the launch's group slots, and all hand over together once they do not
    if (pool != nullptr && dry && dump_threshold > 10) {
      unsigned long long done = 0;
      if (lane == 0) done = __hip_atomic_load(queue + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      done = shfl_u64(done, 0);
      const unsigned long long in_flight = B - done, slots = (unsigned long long)gridDim.x * kGroupsPerWave;
      if (in_flight * 100ull >= slots * (unsigned long long)(dump_threshold - 10)) dry = false;
    }
    if (pool != nullptr && dry &&
        __builtin_popcountll(__builtin_amdgcn_ballot_w64(active && r == 0)) <= dump_threshold) {
      // Two classes (round 3): samples the scout predicts to need long_remaining or more further iterations fill the pool
      // from the front, the others from the back, and the latency kernel takes the front first — the hand-over's longest
      // samples start at once instead of wherever the order of the dumps put them (a 200-round sample that started one
      // millisecond into the latency kernel's two was the end of a 32768-sample batch).  One reservation per wavefront
      // and class.  pool_records == 0: one class, filled from the front.
      bool is_long = true;
      if (pool_records != 0ull) is_long = active && pred != nullptr && (int)pred[idx] - iter >= long_remaining;
      const unsigned long long mL = __builtin_amdgcn_ballot_w64(active && r == 0 && is_long);
      const unsigned long long mS = __builtin_amdgcn_ballot_w64(active && r == 0 && !is_long);
      unsigned long long bL = 0, bS = 0;
      if (lane == 0) {
        if (mL) bL = atomicAdd(pool_count, (unsigned long long)__builtin_popcountll(mL));
        if (mS) bS = atomicAdd(pool_count + 5, (unsigned long long)__builtin_popcountll(mS));
      }
      bL = shfl_u64(bL, 0);
      bS = shfl_u64(bS, 0);
      const unsigned long long below = (1ull << leader) - 1ull;
      const unsigned long long slot = is_long ? bL + (unsigned long long)__builtin_popcountll(mL & below)
                                              : pool_records - 1ull - (bS + (unsigned long long)__builtin_popcountll(mS & below));
      if (active) {
        double *ent = pool + slot * kPoolEntry;
        for (int e = r; e < 14; e += kGroup) ent[e] = rec[kX + e];
        if (r == 0) {
          ent[14] = __longlong_as_double((long long)idx);
          ent[15] = __hiloint2double(updates, iter);
          ent[16] = norm1;
          ent[17] = norm2;
        }
      }
      break;
    }

#define CCMP_FD_BP bp_lane
#include "ccmp_fd_newton_phase1.inc"
    // ---- finished groups: jointValid, write-back --------------------------------------------
    {
      const bool fin = active && !cont;
      bool bad = false;
      if (fin) {
#pragma unroll
        for (int e = 0; e < 14; e++) {
          if (e % kGroup == r) { // e is a compile-time constant here: limits come from SGPRs
            const double v = rec[kX + e];
            if (v < K.lbe[e % 7]) bad = true;
            if (v > K.ube[e % 7]) bad = true;
          }
        }
        // row write-back in 16-B pieces: 96 contiguous bytes per group in one instruction + the last piece
        double2 *row = reinterpret_cast<double2 *>(q_out + idx * 14);
        double2 a;
        a.x = rec[kX + 2 * r];
        a.y = rec[kX + 2 * r + 1];
        if (MODE == 1) { a.x = wrap_pi(a.x); a.y = wrap_pi(a.y); }
        row[r] = a;
        if (r == 0) {
          double2 b;
          b.x = rec[kX + 12];
          b.y = rec[kX + 13];
          if (MODE == 1) { b.x = wrap_pi(b.x); b.y = wrap_pi(b.y); }
          row[6] = b;
        }
      }
      const unsigned long long badmask = __builtin_amdgcn_ballot_w64(bad);
      if (fin && r == 0) {
        const bool gbad = ((badmask >> leader) & 0x3Full) != 0ull;
        ok_out[idx] = (uint8_t)((!gbad) && (norm1 < K.tol_pos) && (norm2 < K.tol_rot));
        if (iters_out) iters_out[idx] = (uint16_t)updates;
        if (pool != nullptr && dump_threshold > 10) atomicAdd(queue + 3, 1ull); // finished count of the occupancy-driven hand-over
      }
      if (fin) active = false;
    }
    if (__builtin_amdgcn_ballot_w64(cont) == 0ull) continue; // nobody iterates: straight to refill
    __syncthreads(); // prefix frames / tool poses written by the writer lane are visible to the group

#include "ccmp_fd_newton_phase2.inc"
#undef CCMP_FD_BP
  }
}


// geodesic_group_kernel — the extend step (jy_ProjectedStateSpace::discreteGeodesic, src/base/jy_ProjectedStateSpace.cpp:
// 32-96) on the THROUGHPUT layout: ten edges x six lanes per wavefront, the Newton iteration of project_fd_kernel (the same
// device functions in the same order: same bits), and at the point where that kernel writes a finished sample back, the
// reference's bookkeeping between two projections — jointValid, the four break tests, the state list, the next
// interpolated state (geodesic_flat_kernel's, statement for statement: ccmp_kernels_geo.hip) — instead.  A Newton round
// costs this layout ~1 250 wave-instructions per edge against the latency kernel's ~2 700, and ~12 times the latency: it
// serves the SHORT edges of a bulk call (round budget, thousands of edges), beside geodesic_flat_kernel, which takes the
// front of the FP32 scout's longest-first order on the side stream (ccmp_api.cpp: geodesic_common).  interpolate == true
// semantics; first calls only (carry_out / round_budget as in the latency kernel; no carry_in); checkMotion's isSatisfied(to) comes from
// a launch in front of this one (target_ok).
// Per-edge state lives in the group's LDS record behind the projector's 165 doubles: previous accepted state (14), dist,
// running length, bound (3), and three counters — 185 doubles per group, 10 wavefronts per CU.
// (The two halves of the Newton round are ONE text, included into both kernels' loops: ccmp_fd_newton_phase1.inc / _phase2.inc.  Shared
// through functions instead, the general instantiations of both kernels came out with 2.3 KB of scratch — the allocator lost the
// scoping of the pose arrays; textual inclusion leaves both allocations as they were: 167 / 168 registers, no scratch for the stock
// instantiations.  What does spill is listed, and bounded at build time, in build.py: _SCRATCH_RULES.)
constexpr int kGPrev = kRec, kGDist = kRec + 14, kGCnt = kRec + 17, kRecG = kRec + 20; // odd stride like kRec
static_assert(kRecG % 2 == 1, "odd record stride: the ten groups stay on distinct LDS banks");

// WrapperStateSpace::interpolate through KinematicChainSpace::interpolate (KinematicChain.h:145-171), one joint
__device__ __forceinline__ double geo_interpolate(double fr, double tg, double tt)
{
  const double pi = 3.14159265358979323846;
  double diff = tg - fr, v;
  if (ccmp_abs(diff) <= pi) v = CCMP_FMA(diff, tt, fr);
  else {
    if (diff > 0.0) diff = 2.0 * pi - diff;
    else diff = -2.0 * pi - diff;
    v = CCMP_FMA(-diff, tt, fr);
    if (v > pi) v -= 2.0 * pi;
    else if (v < -pi) v += 2.0 * pi;
  }
  return v;
}

#ifndef CCMP_GEO_GROUP_WAVES_PER_SIMD
#define CCMP_GEO_GROUP_WAVES_PER_SIMD CCMP_FD_WAVES_PER_SIMD
#endif
template <bool STOCK>
__global__ __launch_bounds__(64, CCMP_GEO_GROUP_WAVES_PER_SIMD) void geodesic_group_kernel(
    const ccmp_consts K, const double delta, const double lambda, const double *__restrict__ from, const double *__restrict__ to,
    unsigned long long E, int max_states, double *__restrict__ states, int *__restrict__ n_states, uint8_t *__restrict__ ok_out,
    int *__restrict__ newton_iters, unsigned long long *queue, const unsigned int *__restrict__ order,
    double *__restrict__ carry_out, int round_budget, double *__restrict__ pool, unsigned long long *pool_count, int handover_pct,
    const uint8_t *__restrict__ target_ok)
{
  __shared__ double lds[kGroupsPerWave * kRecG];
  __shared__ double lane_bp[64]; // the lane's base-frame offset (bp_lane below): parked here instead of in scratch
  const int lane = threadIdx.x;
  const int g = lane / kGroup;
  const int r = lane - kGroup * g;
  const bool live = g < kGroupsPerWave;
  const int leader = live ? kGroup * g : 0;
  double *rec = lds + (live ? g : 0) * kRecG; // idle lanes alias group 0 for reads, never write
  const bool writer = live && r == 0;
  const bool plus = r < 3;
  const int nstep = (plus ? r : r - 3) + 1;
  const int arm_l = plus ? 0 : 1, row_l = plus ? r : r - 3;
  double d_lane = 0.0, bp_lane = 0.0;
  if constexpr (STOCK) {
    d_lane = arm_l ? (row_l == 0 ? K.base_R[1][0] : (row_l == 1 ? K.base_R[1][4] : K.base_R[1][8]))
                   : (row_l == 0 ? K.base_R[0][0] : (row_l == 1 ? K.base_R[0][4] : K.base_R[0][8]));
    bp_lane = arm_l ? (row_l == 0 ? K.base_p[1][0] : (row_l == 1 ? K.base_p[1][1] : K.base_p[1][2]))
                    : (row_l == 0 ? K.base_p[0][0] : (row_l == 1 ? K.base_p[0][1] : K.base_p[0][2]));
  }
  lane_bp[lane] = bp_lane;
  // The lane's address offsets into from / to / states and its bit masks never change either, but they are NOT held across the
  // Newton loop: under the bound of three wavefronts per SIMD the allocator parked them in scratch.  They are recomputed where they
  // are used — once per edge or projection — from a lane index the optimiser cannot see through, which keeps it from hoisting them
  // back out of the loop.
#ifndef CCMP_GEO_GROUP_OPAQUE
#define CCMP_GEO_GROUP_OPAQUE 1
#endif
  auto lane_opaque = [](int v) {
#if CCMP_GEO_GROUP_OPAQUE
    asm volatile("" : "+v"(v));
#endif
    return v;
  };

  unsigned long long edge = 0;
  int iter = 0, updates = 0;
  double norm1 = 0.0, norm2 = 0.0;
  bool active = false, drained = false;

  for (;;) {
    // ---- refill: a group without an edge takes the next ticket and sets the edge up (the reference's prologue) ----------
    {
      const bool want = live && !active && !drained;
      unsigned long long t = 0;
      if (want && r == 0) t = atomicAdd(queue, 1ull);
      t = shfl_u64(t, leader);
      if (want) {
        if (t < E) {
          edge = order ? (unsigned long long)order[t] : t;
          const double *fr = from + edge * 14ull, *tg = to + edge * 14ull;
          double *out = states + edge * (unsigned long long)max_states * 14ull;
          double acc = 0.0; // RealVectorStateSpace::distance(from, to): serial sum in the canonical order, every lane for itself
#pragma unroll
          for (int i = 0; i < 14; i++) {
            const double diff = fr[i] - tg[i];
            acc = CCMP_FMA(diff, diff, acc);
          }
          const double dist = ccmp_sqrt(acc);
          const double total = 0.0, maxd = dist * lambda;
          // target_ok (checkMotion, src/planner/stefanBiPRM.cpp:397-398): isSatisfied(to) of every edge, tested by a launch in front of
          // this one — an edge whose target fails is not traversed (n = 1, ok = 0), as in geodesic_flat_kernel
          const bool tgt = target_ok == nullptr || target_ok[edge] != 0;
          const bool enter = tgt && dist > delta; // (continuations — carry_in — are few edges and stay on geodesic_flat_kernel)
          const double tt = delta / dist;
          for (int e = lane_opaque(r); e < 14; e += kGroup) {
            const double a = fr[e];
            rec[kGPrev + e] = a;
            if (max_states > 0) out[e] = a; // geodesic->push_back(cloneState(from))
            if (enter) rec[kX + e] = geo_interpolate(a, tg[e], tt);
          }
          if (r == 0) {
            rec[kGDist] = dist;
            rec[kGDist + 1] = total;
            rec[kGDist + 2] = maxd;
            int *cnt = reinterpret_cast<int *>(rec + kGCnt);
            cnt[0] = 1; cnt[1] = 0; cnt[2] = 0; // states listed, Newton updates, Newton rounds of this call
          }
          if (enter) {
            active = true; iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
          } else if (r == 0) { // within delta already (or a continuation that has arrived): nothing to traverse
            n_states[edge] = 1;
            ok_out[edge] = (uint8_t)(tgt && dist <= delta);
            if (newton_iters) newton_iters[edge] = 0;
            if (carry_out) { carry_out[2ull * edge] = total; carry_out[2ull * edge + 1ull] = maxd; }
            if (pool) atomicAdd(queue + 3, 1ull); // finished edges: what the hand-over rule below counts down from
          }
        } else drained = true;
      }
    }
    if (__builtin_amdgcn_ballot_w64(active) == 0ull) {
      if (__builtin_amdgcn_ballot_w64(live && !drained) == 0ull) break; // nothing in flight and the queue is empty
      continue;                                                          // edges that needed no traversal: take the next tickets
    }
    __syncthreads();
    // ---- hand-over (as project_fd_kernel's): once the ticket queue is dry and the edges still in flight — E minus the finished
    // count in queue[3], which starts at the front's length — no longer fill handover_pct % of this launch's group slots, every
    // wavefront dumps its live edges at the loop top — iterate, previous state, running distances, counters: the next thing that
    // happens to them is function(x) + the loop test, where geodesic_flat_kernel's Newton routine starts — and retires; latency
    // blocks launched behind this kernel finish them.  A wavefront whose ten edges are of similar predicted length empties all at
    // once; what is left near the end are stragglers on mostly idle wavefronts, at a twelfth of the latency kernel's pace.
    if (pool != nullptr) {
      unsigned long long head = 0, fin_count = 0;
      if (lane == 0) {
        head = __hip_atomic_load(queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fin_count = __hip_atomic_load(queue + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      head = shfl_u64(head, 0);
      fin_count = shfl_u64(fin_count, 0);
      const unsigned long long in_flight = E - fin_count, slots = (unsigned long long)gridDim.x * kGroupsPerWave;
      if (head >= E && in_flight * 100ull < slots * (unsigned long long)handover_pct) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(active && r == 0);
        unsigned long long base = 0;
        if (lane == 0 && m) base = atomicAdd(pool_count, (unsigned long long)__builtin_popcountll(m));
        base = shfl_u64(base, 0);
        if (active) {
          double *ent = pool + (base + (unsigned long long)__builtin_popcountll(m & ((1ull << lane_opaque(leader)) - 1ull))) * (unsigned long long)kGeoPoolEntry;
          for (int e = lane_opaque(r); e < 14; e += kGroup) { ent[e] = rec[kX + e]; ent[14 + e] = rec[kGPrev + e]; }
          if (r == 0) {
            const int *cnt = reinterpret_cast<const int *>(rec + kGCnt);
            ent[28] = rec[kGDist]; ent[29] = rec[kGDist + 1]; ent[30] = rec[kGDist + 2];
            ent[31] = __longlong_as_double((long long)edge);
            ent[32] = __hiloint2double(cnt[0], cnt[1]);   // states listed, Newton updates of the finished projections
            ent[33] = __hiloint2double(cnt[2], iter);     // Newton rounds of the finished projections, loop counter of this one
            ent[34] = __hiloint2double(0, updates);
            ent[35] = norm1; ent[36] = norm2;
          }
        }
        break;
      }
    }

#define CCMP_FD_BP lane_bp[lane_opaque(lane)]
#include "ccmp_fd_newton_phase1.inc"
    // ---- a projection has ended: the reference's bookkeeping between two projections (jy_ProjectedStateSpace.cpp:65-90) --
    {
      const bool fin = active && !cont;
      bool bad = false;
      if (fin) {
#pragma unroll
        for (int e = 0; e < 14; e++) {
          if (e % kGroup == r) {
            const double v = rec[kX + e];
            if (v < K.lbe[e % 7]) bad = true;
            if (v > K.ube[e % 7]) bad = true;
          }
        }
      }
      const unsigned long long badmask = __builtin_amdgcn_ballot_w64(bad);
      if (fin) {
        const bool jv = ((badmask >> leader) & 0x3Full) == 0ull;         // jointValid(x)
        const bool conv = (norm1 < K.tol_pos) && (norm2 < K.tol_rot);     // project's return test
        int *cnt = reinterpret_cast<int *>(rec + kGCnt);
        int n = cnt[0], its = cnt[1] + updates;
        const int rounds = cnt[2] + updates + 1;
        double dist = rec[kGDist], total = rec[kGDist + 1];
        const double maxd = rec[kGDist + 2];
        const double *tg = to + edge * 14ull;
        double *out = states + edge * (unsigned long long)max_states * 14ull;
        double s_acc = 0.0, d_acc = 0.0; // distance(previous, scratch), distance(scratch, to): two serial sums side by side
#pragma unroll
        for (int i = 0; i < 14; i++) {
          const double xi = rec[kX + i];
          const double ds = rec[kGPrev + i] - xi, dd = xi - tg[i];
          s_acc = CCMP_FMA(ds, ds, s_acc);
          d_acc = CCMP_FMA(dd, dd, d_acc);
        }
        bool done = true, fits = true, suspended = false;
        do {
          if (!(conv && jv)) break;                        // not on manifold
          const double step = ccmp_sqrt(s_acc), newDist = ccmp_sqrt(d_acc);
          double delta_o = delta; // (opaque: the product below is otherwise formed once, in front of the loop, and parked in scratch)
          asm volatile("" : "+v"(delta_o));
          if (step > lambda * delta_o) break;              // deviated
          const double total_before = total;
          total += step;
          if (total > maxd) break;                         // wandered too far
          if (newDist >= dist) break;                      // no closer than before
          if (n >= max_states) { fits = false; n = max_states + 1; total = total_before; its -= updates; break; } // list full
          dist = newDist;
          for (int e = lane_opaque(r); e < 14; e += kGroup) out[(unsigned long long)n * 14ull + e] = rec[kX + e];
          n++;
          if (!(dist >= delta)) break;                     // arrived
          if (round_budget > 0 && rounds >= round_budget) { suspended = true; break; } // this call's share of the edge is spent
          done = false;
        } while (0);
        if (done) {
          if (r == 0) {
            n_states[edge] = n;
            ok_out[edge] = suspended ? (uint8_t)2 : (uint8_t)(fits && dist <= delta);
            if (newton_iters) newton_iters[edge] = its;
            if (carry_out) { carry_out[2ull * edge] = total; carry_out[2ull * edge + 1ull] = maxd; }
            if (pool) atomicAdd(queue + 3, 1ull);
          }
          active = false;
        } else { // the accepted state becomes `previous`; the next scratch state is interpolated towards the target
          const double tt = delta / dist;
          for (int e = lane_opaque(r); e < 14; e += kGroup) {
            const double a = rec[kX + e];
            rec[kGPrev + e] = a;
            rec[kX + e] = geo_interpolate(a, tg[e], tt);
          }
          if (r == 0) {
            cnt[0] = n; cnt[1] = its; cnt[2] = rounds;
            rec[kGDist] = dist;
            rec[kGDist + 1] = total;
          }
          iter = 0; updates = 0; norm1 = 0.0; norm2 = 0.0;
        }
      }
    }
    if (__builtin_amdgcn_ballot_w64(cont) == 0ull) continue; // nobody iterates: back to the top (refill, next function(x))
    __syncthreads();

#include "ccmp_fd_newton_phase2.inc"
#undef CCMP_FD_BP
  }
}


// Workspace words (queue heads, counters, the scout's histogram) are cleared by a kernel, not by hipMemsetAsync: a kernel
// node is what a stream capture records and every replay of the graph executes (with hipMemsetAsync the first replay of a
// captured call worked and every later one found the queues exhausted and the histogram still full).
__global__ void clear_words_kernel(unsigned int *__restrict__ w, unsigned int n)
{
  const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = 0u;
}

// ---- simple per-lane kernels (one sample per lane; all bit-identical to the oracle) -------------
// publish the completion word of a single-state call behind its result (pinned host memory, polled by the host)
__device__ __forceinline__ void publish_done(unsigned int *done_flag, unsigned int done_seq)
{
  if (done_flag != nullptr) {
    __threadfence_system();
    __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(64) void function_kernel(const ccmp_consts K, const double *__restrict__ q, double *__restrict__ f, size_t B,
                                                      unsigned int *done_flag, unsigned int done_seq)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double x[14], out[2];
#pragma unroll
  for (int e = 0; e < 14; e++) x[e] = q[i * 14 + e];
  residual(K, x, out);
  f[2 * i] = out[0];
  f[2 * i + 1] = out[1];
  publish_done(done_flag, done_seq);
}

// KinematicChainConstraint::isSatisfied (ConstraintFunction.h:114-120)
__global__ __launch_bounds__(64) void is_satisfied_kernel(const ccmp_consts K, const double *__restrict__ q, uint8_t *__restrict__ ok, size_t B,
                                                          unsigned int *done_flag, unsigned int done_seq)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double x[14], f[2];
#pragma unroll
  for (int e = 0; e < 14; e++) x[e] = q[i * 14 + e];
  residual(K, x, f);
  const bool finite = (f[0] - f[0] == 0.0) && (f[1] - f[1] == 0.0);
  ok[i] = (uint8_t)(finite && f[0] <= K.tol_pos && f[1] <= K.tol_rot);
  publish_done(done_flag, done_seq);
}

// KinematicChainConstraint::jointValid (ConstraintFunction.h:43-55)
__global__ void joint_valid_kernel(const ccmp_consts K, const double *__restrict__ q, uint8_t *__restrict__ ok, size_t B,
                                   unsigned int *done_flag, unsigned int done_seq)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  bool good = true;
  for (int e = 0; e < 14; e++) {
    const double v = q[i * 14 + e];
    const int jj = e < 7 ? e : e - 7;
    if (v < K.lbe[jj]) good = false;
    if (v > K.ube[jj]) good = false;
  }
  ok[i] = (uint8_t)good;
  publish_done(done_flag, done_seq);
}

__global__ void ambient_uniform_kernel(const ccmp_consts K, unsigned long long seed, unsigned long long first,
                                       double *__restrict__ q, size_t n /* = 14*B */)
{
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  q[t] = ambient_uniform(K, seed, first + t / 14, (int)(t % 14));
}

// sampleUniformNear / sampleGaussian ambient samples; `ref` holds one state per sample (stride 14) or a
// single state shared by all (stride 0)
__global__ void ambient_near_kernel(const ccmp_consts K, unsigned long long seed, unsigned long long first,
                                    const double *__restrict__ ref, int ref_stride, double dist, double *__restrict__ q, size_t n)
{
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t i = t / 14;
  const int j = (int)(t % 14);
  q[t] = ambient_near(K, seed, first + i, j, ref[i * (size_t)ref_stride + j], dist);
}
__global__ void ambient_gaussian_kernel(const ccmp_consts K, unsigned long long seed, unsigned long long first,
                                        const double *__restrict__ ref, int ref_stride, double stddev, double *__restrict__ q,
                                        size_t n)
{
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t i = t / 14;
  const int j = (int)(t % 14);
  q[t] = ambient_gaussian(K, seed, first + i, j, ref[i * (size_t)ref_stride + j], stddev);
}

// IKTask::compute_t_wo (src/base/constraints/ik_task.cpp:10-14): object pose from the left arm's joints,
// t_wb * FK(q) * t_o7.inverse(); out[i] = R (9, row-major) then p (3)
__global__ __launch_bounds__(64) void t_wo_kernel(const ccmp_consts K, const double *__restrict__ q, int q_stride, double *__restrict__ out, size_t B)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  double x[7], Rw[9], pw[3], R[9], p[3];
#pragma unroll
  for (int e = 0; e < 7; e++) x[e] = q[i * (size_t)q_stride + e];
  fk_arm(K, 0, x, Rw, pw);
  mul33(Rw, K.t_o7i_R, R);
  p[0] = pw[0]; p[1] = pw[1]; p[2] = pw[2];
  mulvec_acc(Rw, K.t_o7i_p, p);
#pragma unroll
  for (int k = 0; k < 9; k++) out[i * 12 + k] = R[k];
#pragma unroll
  for (int k = 0; k < 3; k++) out[i * 12 + 9 + k] = p[k];
}

__global__ void enforce_bounds_kernel(double *__restrict__ q, size_t n)
{
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  q[t] = wrap_pi(q[t]);
}

// Stable stream compaction of valid rows: block-local scan + one atomic per block would reorder
// blocks, so this is the ordered two-pass form: (1) per-block counts, (2) single-block exclusive
// scan of the counts, (3) scatter.  B/256 counts fit one block's loop comfortably (1024 at 262144).
__global__ void compact_count_kernel(const uint8_t *__restrict__ ok, size_t B, unsigned int *__restrict__ block_counts)
{
  __shared__ unsigned int wsum[4];
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool v = i < B && ok[i] != 0;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (unsigned)__builtin_popcountll(m);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ void compact_scan_kernel(unsigned int *__restrict__ block_counts, size_t nblocks,
                                    unsigned long long *__restrict__ total)
{
  // one 1024-thread block; each thread owns a contiguous chunk
  __shared__ unsigned long long part[1024];
  const size_t per = (nblocks + 1023) / 1024;
  const size_t lo = (size_t)threadIdx.x * per, hi = lo + per < nblocks ? lo + per : nblocks;
  unsigned long long s = 0;
  for (size_t k = lo; k < hi; k++) s += block_counts[k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int k = 0; k < 1024; k++) { unsigned long long v = part[k]; part[k] = run; run += v; }
    *total = run;
  }
  __syncthreads();
  unsigned long long run = part[threadIdx.x];
  for (size_t k = lo; k < hi; k++) { unsigned int v = block_counts[k]; block_counts[k] = (unsigned int)run; run += v; }
}
__global__ void compact_scatter_kernel(const double *__restrict__ q, const uint8_t *__restrict__ ok, size_t B,
                                       const unsigned int *__restrict__ block_offsets, double *__restrict__ out,
                                       size_t capacity)
{
  __shared__ unsigned int wsum[4];
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool v = i < B && ok[i] != 0;
  const unsigned long long m = __builtin_amdgcn_ballot_w64(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wsum[w] = (unsigned)__builtin_popcountll(m);
  __syncthreads();
  unsigned int base = block_offsets[blockIdx.x];
  for (int k = 0; k < w; k++) base += wsum[k];
  if (v) {
    const unsigned int pos = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (pos < capacity) { // rows past the capacity of `out` are dropped; the total still counts them
#pragma unroll
      for (int e = 0; e < 14; e++) out[(size_t)pos * 14 + e] = q[i * 14 + e];
    }
  }
}

} // namespace

// ---- launchers (called from ccmp_api.cpp) --------------------------------------------------------
extern "C" {

// the extend step's short edges on the throughput layout (geodesic_group_kernel): one-wavefront workgroups, ten edges each
hipError_t ccmp_launch_geodesic_group(const ccmp_consts *K, double delta, double lambda, const double *from, const double *to, size_t E,
                                      int max_states, double *states, int *n_states, uint8_t *ok, int *newton_iters, int nblocks,
                                      unsigned long long *queue, const unsigned int *order, double *carry_out, int round_budget,
                                      double *pool, unsigned long long *pool_count, int handover_pct, const uint8_t *target_ok, hipStream_t st)
{
  if (K->stock && K->twin_arms) // the STOCK instantiation also assumes twin arms on diag(+-1) base frames (chain_rows), like project_fd_kernel's
    hipLaunchKernelGGL(geodesic_group_kernel<true>, dim3(nblocks), dim3(64), 0, st, *K, delta, lambda, from, to, (unsigned long long)E,
                       max_states, states, n_states, ok, newton_iters, queue, order, carry_out, round_budget, pool, pool_count, handover_pct, target_ok);
  else
    hipLaunchKernelGGL(geodesic_group_kernel<false>, dim3(nblocks), dim3(64), 0, st, *K, delta, lambda, from, to, (unsigned long long)E,
                       max_states, states, n_states, ok, newton_iters, queue, order, carry_out, round_budget, pool, pool_count, handover_pct, target_ok);
  return hipGetLastError();
}

hipError_t ccmp_launch_clear_words(void *words, size_t n_u32, hipStream_t st)
{
  hipLaunchKernelGGL(clear_words_kernel, dim3((unsigned)((n_u32 + 255) / 256)), dim3(256), 0, st, (unsigned int *)words, (unsigned int)n_u32);
  return hipGetLastError();
}

hipError_t ccmp_launch_project_group(const ccmp_consts *K, int mode, const double *q_in, double *q_out, uint8_t *ok,
                                     uint16_t *iters, double *q_ambient, size_t B, unsigned long long *queue,
                                     unsigned long long seed, unsigned long long first, int nblocks, double *pool,
                                     int dump_threshold, const unsigned int *order, const uint16_t *pred, int long_remaining,
                                     size_t pool_records, hipStream_t st)
{
  // queue[0]: sample queue of this kernel; queue[1]: pool fill count (front); queue[6]: pool fill count from the back
#define CCMP_LAUNCH_GROUP(MODE, STOCK)                                                                                        \
  hipLaunchKernelGGL((project_fd_kernel<MODE, STOCK>), dim3(nblocks), dim3(64), 0, st, *K, q_in, q_out, ok, iters, q_ambient, \
                     (unsigned long long)B, queue, seed, first, pool, queue + 1, dump_threshold, order, pred, long_remaining, \
                     (unsigned long long)pool_records)
  if (mode == 0) {
    if (K->stock && K->twin_arms) CCMP_LAUNCH_GROUP(0, true);
    else CCMP_LAUNCH_GROUP(0, false);
  } else {
    if (!(K->stock && K->twin_arms)) return hipErrorInvalidValue; // the fused sampler exists for the stock structure only (ccmp_api.cpp: project_common)
    CCMP_LAUNCH_GROUP(1, true);
  }
#undef CCMP_LAUNCH_GROUP
  return hipGetLastError();
}

// done_flag (nullable) is honoured for B == 1 only: the single working thread publishes done_seq behind its result
hipError_t ccmp_launch_function(const ccmp_consts *K, const double *q, double *f, size_t B, unsigned int *done_flag,
                                unsigned int done_seq, hipStream_t st)
{
  if (B != 1) done_flag = nullptr;
  hipLaunchKernelGGL(function_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, *K, q, f, B, done_flag, done_seq);
  return hipGetLastError();
}
hipError_t ccmp_launch_is_satisfied(const ccmp_consts *K, const double *q, uint8_t *ok, size_t B, unsigned int *done_flag,
                                    unsigned int done_seq, hipStream_t st)
{
  if (B != 1) done_flag = nullptr;
  hipLaunchKernelGGL(is_satisfied_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, *K, q, ok, B, done_flag, done_seq);
  return hipGetLastError();
}
hipError_t ccmp_launch_joint_valid(const ccmp_consts *K, const double *q, uint8_t *ok, size_t B, unsigned int *done_flag,
                                   unsigned int done_seq, hipStream_t st)
{
  if (B != 1) done_flag = nullptr;
  hipLaunchKernelGGL(joint_valid_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, *K, q, ok, B, done_flag, done_seq);
  return hipGetLastError();
}
hipError_t ccmp_launch_ambient_uniform(const ccmp_consts *K, unsigned long long seed, unsigned long long first,
                                       double *q, size_t B, hipStream_t st)
{
  size_t n = B * 14;
  hipLaunchKernelGGL(ambient_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *K, seed, first, q, n);
  return hipGetLastError();
}
hipError_t ccmp_launch_ambient_ref(const ccmp_consts *K, int kind, unsigned long long seed, unsigned long long first,
                                   const double *ref, int ref_stride, double param, double *q, size_t B, hipStream_t st)
{
  size_t n = B * 14;
  if (kind == 0)
    hipLaunchKernelGGL(ambient_near_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *K, seed, first, ref,
                       ref_stride, param, q, n);
  else
    hipLaunchKernelGGL(ambient_gaussian_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, *K, seed, first, ref,
                       ref_stride, param, q, n);
  return hipGetLastError();
}
hipError_t ccmp_launch_t_wo(const ccmp_consts *K, const double *q, int q_stride, double *out, size_t B, hipStream_t st)
{
  hipLaunchKernelGGL(t_wo_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, *K, q, q_stride, out, B);
  return hipGetLastError();
}
hipError_t ccmp_launch_enforce_bounds(double *q, size_t B, hipStream_t st)
{
  size_t n = B * 14;
  hipLaunchKernelGGL(enforce_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q, n);
  return hipGetLastError();
}
hipError_t ccmp_launch_compact(const double *q, const uint8_t *ok, size_t B, double *out, size_t capacity,
                               unsigned int *block_counts, unsigned long long *total, hipStream_t st)
{
  const size_t nblocks = (B + 255) / 256;
  hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, ok, B, block_counts);
  hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, st, block_counts, nblocks, total);
  hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, q, ok, B, block_counts, out, capacity);
  return hipGetLastError();
}

} // extern "C"
