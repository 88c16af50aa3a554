"""ctypes binding of the CPU oracle (oracle/ccmp_oracle.c).

Test infrastructure: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg only.  Never imported by the closed_chain_motion_planner_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")


class OrcProblem(C.Structure):
    """Mirror of orc_problem (oracle/ccmp_oracle.h); layout-compatible with ccmp_problem."""

    _fields_ = [
        ("axis", C.c_double * 42),
        ("offset", C.c_double * 42),
        ("ee", C.c_double * 6),
        ("R_tool", C.c_double * 18),
        ("base_R", C.c_double * 18),
        ("base_p", C.c_double * 6),
        ("init_R", C.c_double * 9),
        ("init_p", C.c_double * 3),
        ("lb", C.c_double * 7),
        ("ub", C.c_double * 7),
        ("joint_eps", C.c_double),
        ("tol_pos", C.c_double),
        ("tol_rot", C.c_double),
        ("step", C.c_double),
        ("delta", C.c_double),
        ("lambda_", C.c_double),
        ("start_joint", C.c_double * 14),
        ("obj_start_R", C.c_double * 9),
        ("obj_start_p", C.c_double * 3),
        ("obj_goal_R", C.c_double * 9),
        ("obj_goal_p", C.c_double * 3),
        ("t_o7_R", C.c_double * 18),
        ("t_o7_p", C.c_double * 6),
        ("max_iter", C.c_int32),
        ("jacobian_mode", C.c_int32),
        ("arm_index", C.c_int32 * 2),
    ]


class OrcSphere(C.Structure):
    """orc_sphere == ccmp_sphere"""
    _fields_ = [("frame", C.c_int32), ("group", C.c_int32), ("c", C.c_double * 3), ("r", C.c_double)]


class OrcBox(C.Structure):
    """orc_box == ccmp_box"""
    _fields_ = [("group", C.c_int32), ("reserved", C.c_int32), ("c", C.c_double * 3), ("R", C.c_double * 9), ("half", C.c_double * 3)]


def pack_proxies(spheres, boxes, allowed):
    """spheres: iterable of (frame, group, centre, radius); boxes: (group, centre, R 3x3, half); allowed: 32 ints or None"""
    sa = (OrcSphere * max(1, len(spheres)))()
    for k, (frame, group, c, r) in enumerate(spheres):
        sa[k] = OrcSphere(int(frame), int(group), (C.c_double * 3)(*[float(v) for v in c]), float(r))
    ba = (OrcBox * max(1, len(boxes)))()
    for k, (group, c, R, half) in enumerate(boxes):
        ba[k] = OrcBox(int(group), 0, (C.c_double * 3)(*[float(v) for v in c]),
                       (C.c_double * 9)(*[float(v) for v in np.asarray(R, dtype=np.float64).reshape(9)]),
                       (C.c_double * 3)(*[float(v) for v in half]))
    al = None if allowed is None else (C.c_uint32 * 32)(*[int(v) & 0xFFFFFFFF for v in allowed])
    return sa, len(spheres), ba, len(boxes), al


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """One build (libm or det) of the oracle."""

    def __init__(self, kind="det"):
        # CCMP_ORACLE_BUILD: a sanitized build of the same sources (tools/sanitize_cpu.py)
        path = os.path.join(os.environ.get("CCMP_ORACLE_BUILD") or os.path.join(ORACLE_DIR, "build"), "libccmp_oracle_%s.so" % kind)
        if not os.path.exists(path):
            build_oracle()
        self.kind = kind
        self.lib = lib = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        pp = C.POINTER(OrcProblem)
        lib.orc_problem_init.argtypes = [pp, C.c_char_p, C.c_int, C.c_char_p, C.c_int, dp, dp, dp, dp, dp]
        lib.orc_problem_init.restype = C.c_int
        lib.orc_set_start.argtypes = [pp, dp]
        lib.orc_fk.argtypes = [pp, C.c_int, dp, dp, dp]
        lib.orc_function.argtypes = [pp, dp, dp]
        lib.orc_jacobian_fd.argtypes = [pp, dp, dp]
        lib.orc_jacobian_analytic.argtypes = [pp, dp, dp]
        lib.orc_jacobian_analytic_world.argtypes = [pp, dp, dp]
        lib.orc_solve_minnorm.argtypes = [dp, dp, dp]
        lib.orc_solve_gram.argtypes = [dp, dp, dp]
        lib.orc_project.argtypes = [pp, dp, C.POINTER(C.c_int32)]
        lib.orc_project.restype = C.c_int
        lib.orc_joint_valid.argtypes = [pp, dp]
        lib.orc_joint_valid.restype = C.c_int
        lib.orc_is_satisfied.argtypes = [pp, dp]
        lib.orc_is_satisfied.restype = C.c_int
        lib.orc_enforce_bounds.argtypes = [dp]
        lib.orc_interpolate.argtypes = [dp, dp, C.c_double, dp]
        lib.orc_distance.argtypes = [dp, dp]
        lib.orc_distance.restype = C.c_double
        lib.orc_splitmix64.argtypes = [C.c_uint64]
        lib.orc_splitmix64.restype = C.c_uint64
        lib.orc_ambient_uniform.argtypes = [pp, C.c_uint64, C.c_uint64, dp]
        lib.orc_ambient_near.argtypes = [pp, C.c_uint64, C.c_uint64, dp, C.c_double, dp]
        lib.orc_ambient_gaussian.argtypes = [pp, C.c_uint64, C.c_uint64, dp, C.c_double, dp]
        lib.orc_log.argtypes = [C.c_double]
        lib.orc_log.restype = C.c_double
        lib.orc_discrete_geodesic.argtypes = [pp, dp, dp, C.c_int, C.c_void_p, C.c_void_p, dp, C.c_int,
                                              C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        lib.orc_discrete_geodesic.restype = C.c_int
        lib.orc_discrete_geodesic_ex.argtypes = [pp, dp, dp, C.c_int, C.c_void_p, C.c_void_p, dp, C.c_int,
                                                 C.POINTER(C.c_int), C.POINTER(C.c_int64), dp, dp]
        lib.orc_discrete_geodesic_ex.restype = C.c_int
        lib.orc_compute_t_wo.argtypes = [pp, dp, dp, dp]
        lib.orc_function_batch.argtypes = [pp, dp, dp, C.c_size_t, C.c_int]
        lib.orc_project_batch.argtypes = [pp, dp, dp, C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.c_size_t, C.c_int]
        lib.orc_sample_project_batch.argtypes = [pp, C.c_uint64, C.c_uint64, dp, C.POINTER(C.c_uint8),
                                                 C.POINTER(C.c_int32), C.c_size_t, C.c_int]
        lib.orc_discrete_geodesic_batch.argtypes = [pp, dp, dp, C.c_size_t, C.c_int, dp, C.POINTER(C.c_int32),
                                                    C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.c_int]
        lib.orc_proxy_centres.argtypes = [pp, C.POINTER(OrcSphere), C.c_int, dp, dp]
        lib.orc_clearance.argtypes = [pp, C.POINTER(OrcSphere), C.c_int, C.POINTER(OrcBox), C.c_int, C.POINTER(C.c_uint32), dp, dp,
                                      C.POINTER(C.c_int32)]
        lib.orc_clearance.restype = C.c_int
        lib.orc_clearance_batch.argtypes = [pp, C.POINTER(OrcSphere), C.c_int, C.POINTER(OrcBox), C.c_int, C.POINTER(C.c_uint32), dp,
                                            C.c_size_t, dp, C.POINTER(C.c_int32)]
        lib.orc_sincos.argtypes = [C.c_double, dp, dp]
        lib.orc_atan2_nn.argtypes = [C.c_double, C.c_double]
        lib.orc_atan2_nn.restype = C.c_double
        lib.orc_is_detmath.restype = C.c_int
        lib.orc_problem_sizeof.restype = C.c_size_t
        lib.orc_set_variant.argtypes = [C.c_int, C.c_int]
        lib.orc_get_variant.argtypes = [C.c_int]
        lib.orc_get_variant.restype = C.c_int

    # -- setup ------------------------------------------------------------------------------
    def problem(self, cfg):
        """cfg: dict with start_joint, arm1{name,index}, arm2{name,index}, t_wo_* (the YAML keys)."""
        P = OrcProblem()
        q0 = np.ascontiguousarray(cfg["start_joint"], dtype=np.float64)
        sp = np.ascontiguousarray(cfg.get("t_wo_start_pos", [0, 0, 0]), dtype=np.float64)
        sq = np.ascontiguousarray(cfg.get("t_wo_start_quat", [0, 0, 0, 1]), dtype=np.float64)
        gp = np.ascontiguousarray(cfg.get("t_wo_goal_pos", [0, 0, 0]), dtype=np.float64)
        gq = np.ascontiguousarray(cfg.get("t_wo_goal_quat", [0, 0, 0, 1]), dtype=np.float64)
        rc = self.lib.orc_problem_init(C.byref(P), cfg["arm1"]["name"].encode(), int(cfg["arm1"]["index"]),
                                       cfg["arm2"]["name"].encode(), int(cfg["arm2"]["index"]),
                                       _dptr(q0), _dptr(sp), _dptr(sq), _dptr(gp), _dptr(gq))
        assert rc == 0
        return P

    def checker_problem(self, yaml_path, product_problem=None):
        """The checker's problem for a product problem that still is what the product's loader made of `yaml_path`: built
        by the ORACLE'S OWN set-up code (orc_problem_init / orc_set_start: arm order, base frames, DH constants,
        init_chain_) from the YAML file, never from the product's struct — and, when the product's problem is given,
        compared with it byte for byte (jacobian_mode aside: a switch of this library, not part of the reference's
        set-up), so that the two set-up paths check each other wherever a result is checked."""
        import yaml

        with open(yaml_path) as f:
            own = self.problem(yaml.safe_load(f))
        if product_problem is not None:
            own.jacobian_mode = product_problem.jacobian_mode
            assert bytes(own) == bytes(product_problem), "the product's set-up and the oracle's disagree on %s" % yaml_path
        return own

    def set_start(self, P, q0):
        """orc_set_start: init_chain_ and t_o7 again from the problem's present arms and base frames"""
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        self.lib.orc_set_start(C.byref(P), _dptr(q0))
        return P

    def problem_from_bytes(self, raw):
        """Adopt the bytes of a ccmp_problem built by the product library (layouts are identical).  ONLY for problems the
        caller has modified after loading (tolerances, iteration cap, calibration, arms, tilted bases): everything else goes
        through checker_problem, which does not trust the product's set-up."""
        assert len(raw) == C.sizeof(OrcProblem), (len(raw), C.sizeof(OrcProblem))
        return OrcProblem.from_buffer_copy(raw)

    # -- single-sample ------------------------------------------------------------------------
    def fk(self, P, arm, q7):
        q = np.ascontiguousarray(q7, dtype=np.float64)
        R = np.empty(9); p = np.empty(3)
        self.lib.orc_fk(C.byref(P), arm, _dptr(q), _dptr(R), _dptr(p))
        return R.reshape(3, 3), p

    def function(self, P, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = np.empty(2)
        self.lib.orc_function(C.byref(P), _dptr(x), _dptr(f))
        return f

    def jacobian(self, P, x, analytic=False):
        """analytic: False = OMPL's FD stencil, True = the product's analytic mode (kernel order), "world" = the
        independent world-frame formulation of the same derivative"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        J = np.empty(28)
        fn = {False: self.lib.orc_jacobian_fd, True: self.lib.orc_jacobian_analytic, "world": self.lib.orc_jacobian_analytic_world}[analytic]
        fn(C.byref(P), _dptr(x), _dptr(J))
        return J.reshape(2, 14)

    def solve_minnorm(self, J, f):
        J = np.ascontiguousarray(J, dtype=np.float64).reshape(28)
        f = np.ascontiguousarray(f, dtype=np.float64)
        dx = np.empty(14)
        self.lib.orc_solve_minnorm(_dptr(J), _dptr(f), _dptr(dx))
        return dx

    def solve_gram(self, J, f):
        J = np.ascontiguousarray(J, dtype=np.float64).reshape(28)
        f = np.ascontiguousarray(f, dtype=np.float64)
        dx = np.empty(14)
        self.lib.orc_solve_gram(_dptr(J), _dptr(f), _dptr(dx))
        return dx

    def project(self, P, x):
        x = np.array(x, dtype=np.float64)
        it = C.c_int32(0)
        ok = self.lib.orc_project(C.byref(P), _dptr(x), C.byref(it))
        return bool(ok), x, it.value

    def joint_valid(self, P, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return bool(self.lib.orc_joint_valid(C.byref(P), _dptr(x)))

    def is_satisfied(self, P, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return bool(self.lib.orc_is_satisfied(C.byref(P), _dptr(x)))

    def enforce_bounds(self, x):
        x = np.array(x, dtype=np.float64)
        self.lib.orc_enforce_bounds(_dptr(x))
        return x

    def interpolate(self, a, b, t):
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.empty(14)
        self.lib.orc_interpolate(_dptr(a), _dptr(b), float(t), _dptr(out))
        return out

    def distance(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        return self.lib.orc_distance(_dptr(a), _dptr(b))

    def ambient_uniform(self, P, seed, index):
        q = np.empty(14)
        self.lib.orc_ambient_uniform(C.byref(P), seed, index, _dptr(q))
        return q

    def ambient_uniform_batch(self, P, seed, first, B):
        out = np.empty((B, 14))
        for i in range(B):
            self.lib.orc_ambient_uniform(C.byref(P), seed, first + i, _dptr(out[i]))
        return out

    def ambient_ref_batch(self, P, kind, seed, first, ref, param, B):
        """kind 'near' / 'gaussian'; ref is (14,) shared or (B,14) per sample"""
        ref = np.ascontiguousarray(ref, dtype=np.float64)
        out = np.empty((B, 14))
        fn = self.lib.orc_ambient_near if kind == "near" else self.lib.orc_ambient_gaussian
        for i in range(B):
            r = ref if ref.ndim == 1 else np.ascontiguousarray(ref[i])
            fn(C.byref(P), seed, first + i, _dptr(r), float(param), _dptr(out[i]))
        return out

    VARIANTS = {"stencil": 0, "h": 1, "solve": 2, "angle": 3, "return": 4}

    def variant(self, name, value):
        """context manager: one deliberately wrong variant of the third-party arithmetic (oracle/ccmp_oracle.h ORC_VAR_*),
        reset to the restatement on exit; process-global — single-threaded use only"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self.lib.orc_set_variant(self.VARIANTS[name], int(value))
            try:
                yield self
            finally:
                self.lib.orc_set_variant(self.VARIANTS[name], 0)
        return cm()

    def discrete_geodesic(self, P, a, b, interpolate=False, max_states=256):
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros((max_states, 14))
        n = C.c_int(0); its = C.c_int64(0)
        ok = self.lib.orc_discrete_geodesic(C.byref(P), _dptr(a), _dptr(b), int(interpolate), None, None,
                                            _dptr(out), max_states, C.byref(n), C.byref(its))
        if n.value > max_states:  # the list did not fit (max_states + 1): run the edge again with room
            return self.discrete_geodesic(P, a, b, interpolate, max_states=4 * max_states)
        return bool(ok), out[: n.value].copy(), its.value

    def discrete_geodesic_ex(self, P, a, b, max_states, carry_in=None):
        """one bounded call of the resumable form: (ok, states (min(n, max_states),14), n, its, carry (2,))"""
        a = np.ascontiguousarray(a, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros((max_states, 14))
        n = C.c_int(0); its = C.c_int64(0)
        cin = None if carry_in is None else np.ascontiguousarray(carry_in, dtype=np.float64)
        cout = np.zeros(2)
        ok = self.lib.orc_discrete_geodesic_ex(C.byref(P), _dptr(a), _dptr(b), 1, None, None, _dptr(out), max_states,
                                               C.byref(n), C.byref(its), _dptr(cin) if cin is not None else None, _dptr(cout))
        return bool(ok), out[: min(n.value, max_states)].copy(), n.value, its.value, cout

    def compute_t_wo(self, P, q7):
        q = np.ascontiguousarray(q7, dtype=np.float64)
        R = np.empty(9); p = np.empty(3)
        self.lib.orc_compute_t_wo(C.byref(P), _dptr(q), _dptr(R), _dptr(p))
        return R.reshape(3, 3), p

    # -- proxy-geometry clearance ---------------------------------------------------------------
    def proxy_centres(self, P, spheres, x):
        sa, ns, _, _, _ = pack_proxies(spheres, [], None)
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty((ns, 3))
        self.lib.orc_proxy_centres(C.byref(P), sa, ns, _dptr(x), _dptr(out))
        return out

    def clearance(self, P, spheres, boxes, allowed, x):
        """-> (clearance, pair code, number of pairs tested)"""
        sa, ns, ba, nb, al = pack_proxies(spheres, boxes, allowed)
        x = np.ascontiguousarray(x, dtype=np.float64)
        clr = C.c_double(0.0); pair = C.c_int32(0)
        n = self.lib.orc_clearance(C.byref(P), sa, ns, ba, nb, al, _dptr(x), C.byref(clr), C.byref(pair))
        return clr.value, pair.value, n

    def clearance_batch(self, P, spheres, boxes, allowed, q):
        sa, ns, ba, nb, al = pack_proxies(spheres, boxes, allowed)
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        clr = np.empty(B); pair = np.empty(B, dtype=np.int32)
        self.lib.orc_clearance_batch(C.byref(P), sa, ns, ba, nb, al, _dptr(q), B, _dptr(clr), pair.ctypes.data_as(C.POINTER(C.c_int32)))
        return clr, pair

    # -- batch --------------------------------------------------------------------------------
    def function_batch(self, P, q, nthreads=8):
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        f = np.empty((B, 2))
        self.lib.orc_function_batch(C.byref(P), _dptr(q), _dptr(f), B, nthreads)
        return f

    def project_batch(self, P, q, nthreads=8):
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        out = np.empty_like(q)
        ok = np.zeros(B, dtype=np.uint8)
        it = np.zeros(B, dtype=np.int32)
        self.lib.orc_project_batch(C.byref(P), _dptr(q), _dptr(out), ok.ctypes.data_as(C.POINTER(C.c_uint8)),
                                   it.ctypes.data_as(C.POINTER(C.c_int32)), B, nthreads)
        return out, ok, it

    def sample_project_batch(self, P, seed, first, B, nthreads=8):
        out = np.empty((B, 14))
        ok = np.zeros(B, dtype=np.uint8)
        it = np.zeros(B, dtype=np.int32)
        self.lib.orc_sample_project_batch(C.byref(P), seed, first, _dptr(out),
                                          ok.ctypes.data_as(C.POINTER(C.c_uint8)),
                                          it.ctypes.data_as(C.POINTER(C.c_int32)), B, nthreads)
        return out, ok, it

    def discrete_geodesic_batch(self, P, frm, to, max_states=64, nthreads=8):
        """(E,14) x2 -> states (E,max_states,14), n_states (E,), ok (E,), newton_iters (E,); interpolate == True"""
        frm = np.ascontiguousarray(frm, dtype=np.float64); to = np.ascontiguousarray(to, dtype=np.float64)
        E = frm.shape[0]
        st = np.zeros((E, max_states, 14))
        n = np.zeros(E, dtype=np.int32); ok = np.zeros(E, dtype=np.uint8); its = np.zeros(E, dtype=np.int32)
        self.lib.orc_discrete_geodesic_batch(C.byref(P), _dptr(frm), _dptr(to), E, max_states, _dptr(st),
                                             n.ctypes.data_as(C.POINTER(C.c_int32)), ok.ctypes.data_as(C.POINTER(C.c_uint8)),
                                             its.ctypes.data_as(C.POINTER(C.c_int32)), nthreads)
        return st, n, ok, its

    def sincos(self, x):
        s = C.c_double(); c = C.c_double()
        self.lib.orc_sincos(float(x), C.byref(s), C.byref(c))
        return s.value, c.value
