"""Host-side mirror of the reference's constraint interface for the projector hot path.

Names, argument meaning and error behaviour follow
include/closed_chain_motion_planner/base/constraints/ConstraintFunction.h:21-137 of the reference
(`KinematicChainConstraint`) and src/kinematics/grasping_point.cpp:34-65 (`loadConfig`), so that the
parity tests read like tests of the reference class.  All arithmetic happens in libccmp.so on the
GPU; PyTorch is used only to own device memory and streams.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CCMP_JAC_ANALYTIC, CCMP_JAC_FD, CcmpError, CcmpProblem, check

__all__ = ["Context", "Communicator", "KinematicChainConstraint", "ArmModel", "load_config", "CCMP_JAC_FD", "CCMP_JAC_ANALYTIC"]


def _torch():
    import torch  # deferred: importing the package must not need a GPU

    return torch


_SIZE_DEFAULT = C.c_size_t(-1).value  # CCMP_DEFAULT


class Context:
    """ccmp_ctx: one per (process, device)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(_lib.lib().ccmp_ctx_create(int(device), C.byref(self._h)), "ccmp_ctx_create")
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    @property
    def num_cus(self):
        return _lib.lib().ccmp_ctx_num_cus(self._h)

    def set_waves_per_cu(self, w):
        check(_lib.lib().ccmp_ctx_set_waves_per_cu(self._h, int(w)), "ccmp_ctx_set_waves_per_cu")

    def set_schedule(self, wave_kernel=1, small_batch=None):
        """0 = group kernel only, 1 = group kernel + latency kernel for stragglers (default), 2 = latency kernel only;
        small_batch None = the library default"""
        sb = _SIZE_DEFAULT if small_batch is None else int(small_batch)
        check(_lib.lib().ccmp_ctx_set_schedule(self._h, int(wave_kernel), sb), "ccmp_ctx_set_schedule")

    def set_option(self, name, value):
        """tuning knobs of include/ccmp.h (ccmp_ctx_set_option): "handover_threshold", "flat_kernel", "stock_kernels",
        "analytic_cap" / "analytic_small_batch" / "analytic_handover_max", "analytic_split" (+ "_min", "_max", "_pred",
        "_front", "_cap"), "clearance_per_state_max"; results never change"""
        check(_lib.lib().ccmp_ctx_set_option(self._h, name.encode(), int(value)), "ccmp_ctx_set_option(%s)" % name)

    def get_option(self, name):
        """ccmp_ctx_get_option: the value in force (also "num_cus", "side_stream_busy", "resident")"""
        return _lib.get_option(self._h, name)

    def describe(self, call_kind, n):
        """ccmp_ctx_describe: which kernels and thresholds the policy takes for a call of n samples / edges (call_kind:
        _lib.CALL_PROJECT, CALL_SAMPLE_PROJECT, CALL_PROJECT_ANALYTIC, CALL_GEODESIC, CALL_GEODESIC_BUDGET)"""
        return _lib.describe(self._h, call_kind, n)

    def set_lpt(self, mode=1, min_batch=None):
        """0 = index order, 1 = FP32 scout + longest-predicted-first (default), 2 = the same without hand-over;
        min_batch None = the library default"""
        mb = _SIZE_DEFAULT if min_batch is None else int(min_batch)
        check(_lib.lib().ccmp_ctx_set_lpt(self._h, int(mode), mb), "ccmp_ctx_set_lpt")

    def close(self):
        if self._h:
            _lib.lib().ccmp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Communicator:
    """ccmp_comm: one process, one context per GPU, RCCL (ncclCommInitAll) underneath — the collective form of the
    single-process sharding."""

    def __init__(self, contexts):
        self.contexts = list(contexts)
        self._h = C.c_void_p()
        arr = (C.c_void_p * len(self.contexts))(*[cx.handle for cx in self.contexts])
        check(_lib.lib().ccmp_comm_create(arr, len(self.contexts), C.byref(self._h)), "ccmp_comm_create")

    @property
    def handle(self):
        return self._h

    def last_timing(self):
        """(kernel_ms, gather_ms) per GPU of the last sharded call, on each GPU's own stream (ccmp_comm_last_timing)"""
        n = len(self.contexts)
        k, g = (C.c_double * n)(), (C.c_double * n)()
        check(_lib.lib().ccmp_comm_last_timing(self._h, k, g), "ccmp_comm_last_timing")
        return list(k), list(g)

    def close(self):
        if self._h:
            _lib.lib().ccmp_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ArmModel:
    """The fields of the reference's `ArmModel` (panda_model.h:7-23) the projector reads: name, index and — optionally —
    `t_wb`, the base frame in the world as a (3, 4) / (4, 4) array [R | p] (None = the frame of that index in
    src/kinematics/grasping_point.cpp:11-20, which is what the reference fills it with, ConstrainedPlanningCommon.cpp:98)."""

    def __init__(self, name, index, t_wb=None):
        self.name = str(name)
        self.index = int(index)
        self.t_wb = t_wb


def load_config(yaml_path):
    """`grasping_point::loadConfig` + `ConstrainedProblem` set-up: a ready ccmp_problem."""
    P = CcmpProblem()
    check(_lib.lib().ccmp_problem_from_yaml(str(yaml_path).encode(), C.byref(P)), "ccmp_problem_from_yaml(%s)" % yaml_path)
    return P


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _stream_handle(stream):
    if stream is None:
        return C.c_void_p(_torch().cuda.current_stream().cuda_stream)
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)


class KinematicChainConstraint:
    """`KinematicChainConstraint : ompl::base::Constraint` (co-dimension 2) backed by libccmp.

    Single-state methods (`project`, `function`, `isSatisfied`, `jointValid`) keep the reference
    signatures on numpy vectors and run a batch of one on the GPU; the `*_batch` methods take
    (B,14) float64 CUDA tensors.
    """

    def __init__(self, links=14, ctx=None, device=0, problem=None):
        if int(links) != 14:
            raise ValueError("KinematicChainConstraint: the dual-Panda chain has 14 links")
        self.n_ = 14
        self.ctx = ctx if ctx is not None else Context(device)
        self._arms = [None, None]
        self._base_max_iterations = 50  # ompl::base::Constraint::maxIterations_ default; never read by project()
        if problem is not None:
            self.problem = problem.copy()
        else:
            self.problem = None

    # -- construction helpers -----------------------------------------------------------------
    @classmethod
    def from_yaml(cls, yaml_path, ctx=None, device=0):
        return cls(14, ctx=ctx, device=device, problem=load_config(yaml_path))

    def getCoDimension(self):
        return 2

    def getAmbientDimension(self):
        return 14

    def setArmModels(self, arm1, arm2):
        """ConstraintFunction.h:122-126: arm1 = the first seven joints, arm2 the other seven, in the order given (the
        reference's callers pass std::map order, ConstrainedPlanningCommon.cpp:126).  An arm object may carry `t_wb` — a
        (3, 4) or (4, 4) array [R | p], ArmModel::t_wb of panda_model.h:15 — which then replaces the frame looked up from its
        index (src/kinematics/grasping_point.cpp:11-20)."""
        self._arms = [arm1, arm2]
        L = _lib.lib()
        if self.problem is None:
            P = CcmpProblem()
            check(L.ccmp_problem_init(C.byref(P), arm1.name.encode(), arm1.index, arm2.name.encode(), arm2.index,
                                      _dptr(np.zeros(14)), None, None, None, None), "ccmp_problem_init")
            self.problem = P
        # the reference calls this AFTER loadConfig (ConstrainedPlanningCommon.cpp:126): only the arm / base-frame
        # fields change; object poses, start state, tolerances, delta / lambda, calibration and mode are kept and
        # init_chain_ / t_o7 are recomputed for the new arms
        check(L.ccmp_set_arms(C.byref(self.problem), arm1.name.encode(), arm1.index, arm2.name.encode(), arm2.index), "ccmp_set_arms")
        for slot, arm in enumerate((arm1, arm2)):
            t_wb = getattr(arm, "t_wb", None)
            if t_wb is not None:
                T = np.asarray(t_wb, dtype=np.float64)
                R, p = np.ascontiguousarray(T[:3, :3]), np.ascontiguousarray(T[:3, 3])
                check(L.ccmp_set_base_frame(C.byref(self.problem), slot, _dptr(R), _dptr(p)), "ccmp_set_base_frame")

    def setInitialPosition(self, init_joint):
        """ConstraintFunction.h:31-40."""
        self._need_problem()
        q0 = np.ascontiguousarray(init_joint, dtype=np.float64)
        if q0.shape != (14,):
            raise ValueError("init_joint must have 14 entries")
        check(_lib.lib().ccmp_set_start(C.byref(self.problem), _dptr(q0)), "ccmp_set_start")

    def setTolerance(self, tolerance1, tolerance2):
        """ConstraintFunction.h:104-112: non-positive tolerances raise (ompl::Exception there)."""
        self._need_problem()
        rc = _lib.lib().ccmp_set_tolerance(C.byref(self.problem), float(tolerance1), float(tolerance2))
        if rc != 0:
            raise ValueError("ompl::base::Constraint::setProjectionTolerance(): tolerance must be positive.")

    def setMaxIterations(self, n):
        """Sets the *base-class* field, which the reference's project() override never reads
        (ConstrainedPlanningCommon.cpp:129 vs ConstraintFunction.h:26,68): the cap stays 250."""
        self._base_max_iterations = int(n)

    def setJacobianMode(self, mode):
        self._need_problem()
        if mode not in (CCMP_JAC_FD, CCMP_JAC_ANALYTIC):
            raise ValueError("mode must be CCMP_JAC_FD or CCMP_JAC_ANALYTIC")
        self.problem.jacobian_mode = mode

    def _need_problem(self):
        if self.problem is None:
            raise RuntimeError("call setArmModels()/from_yaml() first")
        if not 0 <= self.problem.max_iter <= 32767:
            # the C side reports uint16 iteration counts and this mirror holds them in int16 tensors: a cap the tensors
            # cannot represent is refused here instead of wrapping silently (the reference's cap is 250)
            raise ValueError("problem.max_iter = %d: this binding supports 0..32767" % self.problem.max_iter)

    # -- batched API (CUDA tensors) -------------------------------------------------------------
    def _check_q(self, q):
        torch = _torch()
        if not (isinstance(q, torch.Tensor) and q.is_cuda and q.dtype == torch.float64 and q.dim() == 2
                and q.shape[1] == 14 and q.is_contiguous()):
            raise ValueError("expected a contiguous (B,14) float64 CUDA tensor")
        if q.device.index != self.ctx.device:
            raise ValueError("tensor is on cuda:%s, context on cuda:%d" % (q.device.index, self.ctx.device))

    def project_batch(self, q, out=None, want_iters=True, stream=None):
        """ok[i], q_out[i], iters[i] = project(q[i]); q_out may alias q (in place, like the reference)."""
        self._need_problem()
        self._check_q(q)
        torch = _torch()
        B = q.shape[0]
        if out is None:
            out = torch.empty_like(q)
        else:
            self._check_q(out)
        ok = torch.empty(B, dtype=torch.uint8, device=q.device)
        it = torch.empty(B, dtype=torch.int16, device=q.device) if want_iters else None
        check(_lib.lib().ccmp_project_batch(self.ctx.handle, C.byref(self.problem), q.data_ptr(), out.data_ptr(),
                                            ok.data_ptr(), it.data_ptr() if it is not None else None, B,
                                            _stream_handle(stream)), "ccmp_project_batch")
        return out, ok, self._iters(it)

    def _iters(self, it):
        """the C side writes uint16 counts into an int16 tensor (torch's uint16 has few kernels): identical for every
        cap `_need_problem` lets through (<= 32767; the reference's is 250)"""
        return it

    def sample_project_batch(self, seed, first_index, B, want_iters=True, want_ambient=False, stream=None):
        """`jy_ProjectedStateSampler::sampleUniform` x B (jy_ProjectedStateSpace.cpp:10-15)."""
        self._need_problem()
        torch = _torch()
        dev = torch.device("cuda", self.ctx.device)
        out = torch.empty((B, 14), dtype=torch.float64, device=dev)
        ok = torch.empty(B, dtype=torch.uint8, device=dev)
        it = torch.empty(B, dtype=torch.int16, device=dev) if want_iters else None
        amb = torch.empty((B, 14), dtype=torch.float64, device=dev) if want_ambient else None
        check(_lib.lib().ccmp_sample_project_batch(self.ctx.handle, C.byref(self.problem), int(seed), int(first_index),
                                                   out.data_ptr(), ok.data_ptr(),
                                                   it.data_ptr() if it is not None else None,
                                                   amb.data_ptr() if amb is not None else None, B,
                                                   _stream_handle(stream)), "ccmp_sample_project_batch")
        return out, ok, self._iters(it), amb

    def _sample_ref(self, fn_name, seed, first_index, ref, param, B, want_iters, want_ambient, stream):
        self._need_problem()
        torch = _torch()
        dev = torch.device("cuda", self.ctx.device)
        if not (isinstance(ref, torch.Tensor) and ref.is_cuda and ref.dtype == torch.float64 and ref.is_contiguous()
                and ref.shape[-1] == 14 and ref.numel() in (14, 14 * B)):
            raise ValueError("reference state(s): contiguous float64 CUDA tensor of shape (14,) or (B,14)")
        stride = 0 if ref.numel() == 14 else 14
        out = torch.empty((B, 14), dtype=torch.float64, device=dev)
        ok = torch.empty(B, dtype=torch.uint8, device=dev)
        it = torch.empty(B, dtype=torch.int16, device=dev) if want_iters else None
        amb = torch.empty((B, 14), dtype=torch.float64, device=dev) if want_ambient else None
        check(getattr(_lib.lib(), fn_name)(self.ctx.handle, C.byref(self.problem), int(seed), int(first_index), ref.data_ptr(),
                                           stride, float(param), out.data_ptr(), ok.data_ptr(),
                                           it.data_ptr() if it is not None else None,
                                           amb.data_ptr() if amb is not None else None, B, _stream_handle(stream)), fn_name)
        return out, ok, self._iters(it), amb

    def sample_near_project_batch(self, seed, first_index, near, distance, B, want_iters=True, want_ambient=False, stream=None):
        """`jy_ProjectedStateSampler::sampleUniformNear` x B (jy_ProjectedStateSpace.cpp:17-22)."""
        return self._sample_ref("ccmp_sample_near_project_batch", seed, first_index, near, distance, B, want_iters,
                                want_ambient, stream)

    def sample_gaussian_project_batch(self, seed, first_index, mean, std_dev, B, want_iters=True, want_ambient=False,
                                      stream=None):
        """`jy_ProjectedStateSampler::sampleGaussian` x B (jy_ProjectedStateSpace.cpp:24-29)."""
        return self._sample_ref("ccmp_sample_gaussian_project_batch", seed, first_index, mean, std_dev, B, want_iters,
                                want_ambient, stream)

    def compute_t_wo_batch(self, q, stream=None):
        """`IKTask::compute_t_wo` (ik_task.cpp:10-14) for the left arm's joints q[:, :7] -> (B,12): R row-major, p."""
        self._need_problem()
        torch = _torch()
        if not (isinstance(q, torch.Tensor) and q.is_cuda and q.dtype == torch.float64 and q.dim() == 2 and q.is_contiguous()
                and q.shape[1] >= 7):
            raise ValueError("expected a contiguous (B,>=7) float64 CUDA tensor")
        out = torch.empty((q.shape[0], 12), dtype=torch.float64, device=q.device)
        check(_lib.lib().ccmp_compute_t_wo_batch(self.ctx.handle, C.byref(self.problem), q.data_ptr(), q.shape[1],
                                                 out.data_ptr(), q.shape[0], _stream_handle(stream)), "ccmp_compute_t_wo_batch")
        return out

    def discrete_geodesic_batch(self, frm, to, max_states=64, stream=None, check_target=False, carry_in=None, want_carry=False,
                                round_budget=0):
        """`jy_ProjectedStateSpace::discreteGeodesic` for E edges (jy_ProjectedStateSpace.cpp:32-96), run as with
        interpolate == true.  Returns (states (E,max_states,14), n_states (E,), ok (E,), newton_iters (E,)) and, with
        want_carry, a fifth tensor carry (E,2).
        check_target: `checkMotion` in one launch — isSatisfied(to) && discreteGeodesic(from, to)
        (src/planner/stefanBiPRM.cpp:397-398); an edge whose target is not satisfied reports ok = 0, n_states = 1.
        An edge did not reach its end when n_states == max_states + 1 (its list is full) or — with round_budget > 0, the
        bound on the Newton rounds this call spends on one edge — when ok == 2: continue it with frm = its last stored
        state, the same to and carry_in = its carry row (ccmp_geodesic_batch_ex); `continue_geodesics` below does that
        until every list is whole."""
        self._need_problem()
        self._check_q(frm)
        self._check_q(to)
        torch = _torch()
        E = frm.shape[0]
        states = torch.empty((E, max_states, 14), dtype=torch.float64, device=frm.device)
        n = torch.empty(E, dtype=torch.int32, device=frm.device)
        ok = torch.empty(E, dtype=torch.uint8, device=frm.device)
        its = torch.empty(E, dtype=torch.int32, device=frm.device)
        if carry_in is not None and not (isinstance(carry_in, torch.Tensor) and carry_in.is_cuda and carry_in.dtype == torch.float64
                                         and carry_in.is_contiguous() and tuple(carry_in.shape) == (E, 2)):
            raise ValueError("carry_in: contiguous (E,2) float64 CUDA tensor")
        if round_budget and not want_carry:
            raise ValueError("round_budget needs want_carry=True: a suspended edge is continued from its carry")
        carry = torch.empty((E, 2), dtype=torch.float64, device=frm.device) if want_carry else None
        check(_lib.lib().ccmp_geodesic_batch_ex(self.ctx.handle, C.byref(self.problem), frm.data_ptr(), to.data_ptr(), E, int(max_states),
                                                states.data_ptr(), n.data_ptr(), ok.data_ptr(), its.data_ptr(),
                                                carry_in.data_ptr() if carry_in is not None else None,
                                                carry.data_ptr() if carry is not None else None, int(round_budget),
                                                1 if check_target else 0, _stream_handle(stream)), "ccmp_geodesic_batch_ex")
        return (states, n, ok, its, carry) if want_carry else (states, n, ok, its)

    def continue_geodesics(self, to, states, n, ok, its, carry, max_states, round_budget=0, max_calls=1 << 20, cont_states=None):
        """Finishes the edges of a `discrete_geodesic_batch(..., want_carry=True)` result that did not reach their end
        (list full: n == max_states + 1; round budget spent: ok == 2), each from its last stored state.  Returns {edge index:
        (states (m,14) numpy, ok, newton iterations)} with the complete list of every such edge — what one uninterrupted
        traversal produces, bit for bit.  `max_states` is the list length of the result passed in, `cont_states` (default:
        the same) the one the continuation calls use."""
        torch = _torch()
        cs = max_states if cont_states is None else int(cont_states)
        if max_states < 2 or cs < 2:
            raise ValueError("a continuation starts from a stored state other than `from`: max_states >= 2")
        long = torch.nonzero((n > max_states) | (ok == 2)).flatten()
        out = {}
        if long.numel() == 0:
            return out
        stored = n.clamp(max=max_states)[long]  # states each of them holds
        idx = long.tolist()
        st_h, stored_h = states[long].cpu().numpy(), stored.cpu().tolist()
        parts = {e: [st_h[k, : stored_h[k]]] for k, e in enumerate(idx)}
        total_its = {e: int(v) for e, v in zip(idx, its[long].cpu().tolist())}
        rows = torch.arange(len(idx), device=to.device)
        cur_from = states[long][rows, (stored - 1).long()].contiguous()
        cur_to = to[long].contiguous()
        cur_carry = carry[long].contiguous()
        for _ in range(max_calls):
            s2, n2, ok2, it2, c2 = self.discrete_geodesic_batch(cur_from, cur_to, cs, carry_in=cur_carry, want_carry=True,
                                                                round_budget=round_budget)
            n2c, ok2c, it2c = n2.cpu().numpy(), ok2.cpu().numpy(), it2.cpu().numpy()
            s2h = s2.cpu().numpy()
            again = []
            for k, e in enumerate(idx):
                m = min(int(n2c[k]), cs)
                parts[e].append(s2h[k, 1:m])  # row 0 repeats the state the continuation started from
                total_its[e] += int(it2c[k])
                if n2c[k] > cs or ok2c[k] == 2:
                    again.append(k)
                else:
                    out[e] = (np.concatenate(parts[e], axis=0), int(ok2c[k]), total_its[e])
            if not again:
                break
            sel = torch.as_tensor(again, device=to.device)
            last = (n2.clamp(max=cs)[sel] - 1).long()
            cur_from = s2[sel, last].contiguous()
            cur_to = cur_to[sel].contiguous()
            cur_carry = c2[sel].contiguous()
            idx = [idx[k] for k in again]
        else:
            # max_calls spent with edges still open: a partial dict would let a caller take a cut list for a whole one
            raise RuntimeError("continue_geodesics: %d edges unfinished after %d calls (first: %s)" % (len(idx), max_calls, idx[:8]))
        return out

    def ambient_uniform_batch(self, seed, first_index, B, stream=None):
        self._need_problem()
        torch = _torch()
        out = torch.empty((B, 14), dtype=torch.float64, device=torch.device("cuda", self.ctx.device))
        check(_lib.lib().ccmp_ambient_uniform_batch(self.ctx.handle, C.byref(self.problem), int(seed), int(first_index),
                                                    out.data_ptr(), B, _stream_handle(stream)),
              "ccmp_ambient_uniform_batch")
        return out

    def function_batch(self, q, stream=None):
        self._need_problem()
        self._check_q(q)
        torch = _torch()
        f = torch.empty((q.shape[0], 2), dtype=torch.float64, device=q.device)
        check(_lib.lib().ccmp_function_batch(self.ctx.handle, C.byref(self.problem), q.data_ptr(), f.data_ptr(),
                                             q.shape[0], _stream_handle(stream)), "ccmp_function_batch")
        return f

    def is_satisfied_batch(self, q, stream=None):
        self._need_problem()
        self._check_q(q)
        torch = _torch()
        ok = torch.empty(q.shape[0], dtype=torch.uint8, device=q.device)
        check(_lib.lib().ccmp_is_satisfied_batch(self.ctx.handle, C.byref(self.problem), q.data_ptr(), ok.data_ptr(),
                                                 q.shape[0], _stream_handle(stream)), "ccmp_is_satisfied_batch")
        return ok

    def joint_valid_batch(self, q, stream=None):
        self._need_problem()
        self._check_q(q)
        torch = _torch()
        ok = torch.empty(q.shape[0], dtype=torch.uint8, device=q.device)
        check(_lib.lib().ccmp_joint_valid_batch(self.ctx.handle, C.byref(self.problem), q.data_ptr(), ok.data_ptr(),
                                                q.shape[0], _stream_handle(stream)), "ccmp_joint_valid_batch")
        return ok

    def enforce_bounds_batch(self, q, stream=None):
        """KinematicChainSpace::enforceBounds in place (KinematicChain.h:118-130)."""
        self._check_q(q)
        check(_lib.lib().ccmp_enforce_bounds_batch(self.ctx.handle, q.data_ptr(), q.shape[0], _stream_handle(stream)),
              "ccmp_enforce_bounds_batch")
        return q

    def compact_valid(self, q, ok, stream=None, out=None, cnt=None):
        """Rows of q with ok != 0, in order (what the host tree consumes).  `out` (optional, (cap,14) with cap <= B): a
        fixed-capacity buffer — valid rows past its capacity are dropped, `cnt` still reports how many there were."""
        self._check_q(q)
        torch = _torch()
        if out is None:
            out = torch.empty_like(q)
        else:
            self._check_q(out)
        if cnt is None:
            cnt = torch.zeros(1, dtype=torch.int64, device=q.device)
        check(_lib.lib().ccmp_compact_valid_capped(self.ctx.handle, q.data_ptr(), ok.data_ptr(), q.shape[0], out.data_ptr(),
                                                   out.shape[0], cnt.data_ptr(), _stream_handle(stream)),
              "ccmp_compact_valid_capped")
        return out, cnt

    # -- single-state API with the reference's signatures ------------------------------------------
    def setResident(self, on=True):
        """The adapter's `KinematicChainConstraint::setResident` (include/ccmp_ompl_adapter.hpp): the single-state calls below and
        single-edge calls from host buffers go through the context's resident service kernel (option "resident", include/ccmp.h) —
        no launch on the call path, the same bits.  Off by default."""
        self.ctx.set_option("resident", 1 if on else 0)

    def project(self, x):
        """bool project(Eigen::Ref<VectorXd> x) const — x (numpy, 14) is modified in place."""
        self._need_problem()
        if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.shape == (14,) and x.flags.c_contiguous):
            raise ValueError("x must be a contiguous float64 numpy vector of 14 entries (modified in place)")
        ok = np.zeros(1, dtype=np.uint8)
        check(_lib.lib().ccmp_project_host(self.ctx.handle, C.byref(self.problem), _dptr(x), _dptr(x),
                                           ok.ctypes.data_as(C.POINTER(C.c_uint8)), None, 1), "ccmp_project_host")
        return bool(ok[0])

    def function(self, x, out=None):
        self._need_problem()
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(14)
        f = np.empty(2) if out is None else out
        check(_lib.lib().ccmp_function_host(self.ctx.handle, C.byref(self.problem), _dptr(x), _dptr(f), 1),
              "ccmp_function_host")
        return f

    def isSatisfied(self, x):
        self._need_problem()
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(14)
        ok = np.zeros(1, dtype=np.uint8)
        check(_lib.lib().ccmp_is_satisfied_host(self.ctx.handle, C.byref(self.problem), _dptr(x),
                                                ok.ctypes.data_as(C.POINTER(C.c_uint8)), 1), "ccmp_is_satisfied_host")
        return bool(ok[0])

    def jointValid(self, q):
        self._need_problem()
        x = np.ascontiguousarray(q, dtype=np.float64).reshape(14)
        ok = np.zeros(1, dtype=np.uint8)
        check(_lib.lib().ccmp_joint_valid_host(self.ctx.handle, C.byref(self.problem), _dptr(x),
                                               ok.ctypes.data_as(C.POINTER(C.c_uint8)), 1), "ccmp_joint_valid_host")
        return bool(ok[0])

    def project_sharded_host(self, q, contexts):
        """(B,14) numpy -> (q_out, ok, iters) with the batch split over `contexts` (one process, several GPUs)."""
        self._need_problem()
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        out = np.empty_like(q)
        ok = np.zeros(B, dtype=np.uint8)
        it = np.zeros(B, dtype=np.uint16)
        arr = (C.c_void_p * len(contexts))(*[cx.handle for cx in contexts])
        check(_lib.lib().ccmp_project_sharded_host(arr, len(contexts), C.byref(self.problem), _dptr(q), _dptr(out),
                                                   ok.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                   it.ctypes.data_as(C.POINTER(C.c_uint16)), B), "ccmp_project_sharded_host")
        return out, ok, it

    @staticmethod
    def sharded_host_last_timing(contexts):
        """(launch_ms, start_ms) per context of the last sharded call (ccmp_sharded_host_last_timing): when the host had each
        shard's upload behind it and issued its kernels, on the host clock and on the GPU's timeline relative to the first
        context (-1 on another device).  Shards driven one after the other would show values growing with the index."""
        n = len(contexts)
        arr = (C.c_void_p * n)(*[cx.handle for cx in contexts])
        a, b = np.zeros(n), np.zeros(n)
        check(_lib.lib().ccmp_sharded_host_last_timing(arr, n, _dptr(a), _dptr(b)), "ccmp_sharded_host_last_timing")
        return a.tolist(), b.tolist()

    def sample_project_sharded_host(self, seed, first_index, B, contexts):
        self._need_problem()
        out = np.empty((B, 14))
        ok = np.zeros(B, dtype=np.uint8)
        it = np.zeros(B, dtype=np.uint16)
        arr = (C.c_void_p * len(contexts))(*[cx.handle for cx in contexts])
        check(_lib.lib().ccmp_sample_project_sharded_host(arr, len(contexts), C.byref(self.problem), int(seed), int(first_index),
                                                          _dptr(out), ok.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                          it.ctypes.data_as(C.POINTER(C.c_uint16)), B),
              "ccmp_sample_project_sharded_host")
        return out, ok, it

    def _sharded(self, comm, mode, q, seed, first_index, B, block_rows, want_full):
        n = len(comm.contexts)
        shard = -(-B // n)
        block_rows = int(block_rows) if block_rows else max(1, shard)
        out = np.empty((B, 14)) if want_full else None
        ok = np.zeros(B, dtype=np.uint8) if want_full else None
        it = np.zeros(B, dtype=np.uint16) if want_full else None
        valid = np.empty((n * block_rows, 14))
        counts = (C.c_uint64 * n)()
        nv = C.c_uint64(0)
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None
        u16 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint16)) if a is not None else None
        dp = lambda a: _dptr(a) if a is not None else None
        if mode == 0:
            rc = _lib.lib().ccmp_project_sharded(comm.handle, C.byref(self.problem), _dptr(q), B, dp(out), u8(ok), u16(it), block_rows,
                                                 _dptr(valid), valid.shape[0], counts, C.byref(nv))
        else:
            rc = _lib.lib().ccmp_sample_project_sharded(comm.handle, C.byref(self.problem), int(seed), int(first_index), B, dp(out), u8(ok),
                                                        u16(it), block_rows, _dptr(valid), valid.shape[0], counts, C.byref(nv))
        if rc == -8:
            raise OverflowError("a shard holds %d valid states, the gather blocks hold %d rows" % (max(counts), block_rows))
        check(rc, "ccmp_project_sharded" if mode == 0 else "ccmp_sample_project_sharded")
        return valid[: nv.value].copy(), list(counts), (out, ok, it)

    def project_sharded(self, q, comm, block_rows=None, want_full=True):
        """(B,14) numpy over the communicator's GPUs with the RCCL all-gather of the valid states: returns
        (valid states in global order, per-shard counts, (q_out, ok, iters))"""
        self._need_problem()
        q = np.ascontiguousarray(q, dtype=np.float64)
        return self._sharded(comm, 0, q, 0, 0, q.shape[0], block_rows, want_full)

    def sample_project_sharded(self, seed, first_index, B, comm, block_rows=None, want_full=True):
        self._need_problem()
        return self._sharded(comm, 1, None, seed, first_index, int(B), block_rows, want_full)

    def project_host(self, q):
        """(B,14) numpy in -> (q_out, ok, iters) numpy out through the host-pointer entry point."""
        self._need_problem()
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        out = np.empty_like(q)
        ok = np.zeros(B, dtype=np.uint8)
        it = np.zeros(B, dtype=np.uint16)
        check(_lib.lib().ccmp_project_host(self.ctx.handle, C.byref(self.problem), _dptr(q), _dptr(out),
                                           ok.ctypes.data_as(C.POINTER(C.c_uint8)),
                                           it.ctypes.data_as(C.POINTER(C.c_uint16)), B), "ccmp_project_host")
        return out, ok, it
