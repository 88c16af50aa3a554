"""Host-side mirror of the reference's constrained state space for the projector hot path:
`jy_ProjectedStateSampler` (src/base/jy_ProjectedStateSpace.cpp:5-29), `jy_ProjectedStateSpace::
discreteGeodesic` (:32-96) and the path-matrix format the reference dumps
(PathGeometric::printAsMatrix, src/base/constraints/ConstrainedPlanningCommon.cpp:219-222), all on
top of the batched GPU entry points of KinematicChainConstraint."""
import itertools
import os
import threading

import numpy as np

__all__ = ["jy_ProjectedStateSampler", "jy_ProjectedStateSpace", "check_motion", "geodesic_interpolate", "format_path_matrix",
           "parse_path_matrix", "format_graphml", "parse_graphml", "format_graphviz", "splitmix64", "next_sampler_seed"]


def _torch():
    import torch

    return torch


_M64 = (1 << 64) - 1


def splitmix64(z):
    """one SplitMix64 output (the generator of the kernels' counter-based sampler, csrc/ccmp_fd_common.h)"""
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


# Every sampler owns an independently seeded stream, as every OMPL StateSampler owns its own ompl::RNG (seeded from a
# process-wide seed generator; src/base/jy_ProjectedStateSpace.cpp:5-8 wraps such a sampler).  Samplers created without
# an explicit seed draw one from this process-wide sequence; set CCMP_SEED for reproducible runs (ompl::RNG::setSeed).
_seed_lock = threading.Lock()
_seed_counter = itertools.count()
_process_seed = None


def next_sampler_seed():
    global _process_seed
    with _seed_lock:
        if _process_seed is None:
            env = os.environ.get("CCMP_SEED")
            _process_seed = int(env, 0) & _M64 if env else int.from_bytes(os.urandom(8), "little")
        return splitmix64((_process_seed + next(_seed_counter)) & _M64)


class jy_ProjectedStateSampler:
    """`sampleUniform(state)` / `sampleUniformNear` / `sampleGaussian` with the reference's signatures
    (state = numpy (14,), written in place).  sampleUniform is served from a buffer that ONE GPU launch
    of `batch` fused sample->project->enforceBounds fills whenever it runs empty, invisible to the
    planner; the reference ignores project()'s result here, and so does this class."""

    def __init__(self, constraint, seed=None, batch=4096):
        self.constraint_ = constraint
        self.seed = next_sampler_seed() if seed is None else int(seed) & _M64  # None: its own stream, like an ompl::RNG
        self.batch = int(batch)
        self._next_index = 0  # global sample counter: the stream of samples does not depend on `batch`
        self._buf = None
        self._pos = 0
        self.lookahead = 32   # Near / Gaussian samples drawn per launch around one reference state
        self._ref_buf = {}
        self._ref_index = 0

    def _refill(self):
        q, ok, _, _ = self.constraint_.sample_project_batch(self.seed, self._next_index, self.batch, want_iters=False)
        self._buf = q.cpu().numpy()
        self._ok = ok.cpu().numpy()
        self._next_index += self.batch
        self._pos = 0

    def sampleUniform(self, state):
        if self._buf is None or self._pos >= self._buf.shape[0]:
            self._refill()
        state[:] = self._buf[self._pos]
        self._pos += 1

    # sampleUniformNear / sampleGaussian: `near` may change from call to call, so nothing can be drawn before the call;
    # but a launch of `lookahead` samples around the same state costs what a launch of one does (one 128-thread block
    # per sample on an otherwise idle GPU), so a call that sees a new (state, parameter) draws `lookahead` samples and
    # the following calls with the same arguments are served from that buffer.  Every refill takes fresh indices of
    # the sampler's counter-based stream: no sample is ever handed out twice, whatever the call pattern.
    def _sample_ref(self, kind, state, ref, param):
        torch = _torch()
        ref = np.ascontiguousarray(ref, dtype=np.float64).reshape(14)
        key = (ref.tobytes(), float(param))
        buf = self._ref_buf.get(kind)
        if buf is None or buf[0] != key or buf[2] >= buf[1].shape[0]:
            dev_ref = torch.as_tensor(ref).to("cuda:%d" % self.constraint_.ctx.device)
            fn = self.constraint_.sample_near_project_batch if kind == "near" else self.constraint_.sample_gaussian_project_batch
            seed = self.seed ^ (0x4E454152 if kind == "near" else 0x47415553)
            q, _, _, _ = fn(seed, self._ref_index, dev_ref, float(param), self.lookahead, want_iters=False)
            self._ref_index += self.lookahead
            buf = [key, q.cpu().numpy(), 0]
            self._ref_buf[kind] = buf
        state[:] = buf[1][buf[2]]
        buf[2] += 1

    def sampleUniformNear(self, state, near, distance):
        self._sample_ref("near", state, near, distance)

    def sampleGaussian(self, state, mean, stdDev):
        self._sample_ref("gaussian", state, mean, stdDev)


class jy_ProjectedStateSpace:
    """`discreteGeodesic(from, to, interpolate, geodesic)` with the reference's semantics.  The GPU runs
    the traversal without the validity test; for interpolate == False the host then applies
    `StateValidityChecker::isValid` (MoveIt in the reference; any callable here) in order and cuts the
    list at the first rejected state — exactly where the reference's loop would have stopped
    (jy_ProjectedStateSpace.cpp:65-68)."""

    def __init__(self, constraint, isValid=None, max_states=64, seed=None):
        self.constraint_ = constraint
        self.isValid = isValid
        self.max_states = int(max_states)
        self._seed = next_sampler_seed() if seed is None else int(seed) & _M64
        self._n_samplers = itertools.count()

    def allocStateSampler(self, batch=4096):
        """`allocStateSampler` / `allocDefaultStateSampler` (jy_ProjectedStateSpace.h:41-49): every sampler of a space
        gets its own stream — seed = splitmix64(space seed + running sampler number) — so the planner's sampler, the
        valid-state sampler and a re-plan never repeat each other's samples."""
        return jy_ProjectedStateSampler(self.constraint_, seed=splitmix64((self._seed + next(self._n_samplers)) & _M64),
                                        batch=batch)

    allocDefaultStateSampler = allocStateSampler

    def setDelta(self, delta):
        self.constraint_.problem.delta = float(delta)

    def setLambda(self, lam):
        self.constraint_.problem.lambda_ = float(lam)

    def distance(self, a, b):
        return float(np.sqrt(np.sum((np.asarray(a) - np.asarray(b)) ** 2)))

    def discreteGeodesicBatch(self, frm, to, interpolate=False, check_target=False):
        """numpy (E,14) x2 -> list of (ok, states (n,14)) per edge; check_target: isSatisfied(to) is tested first, in the
        same launch (checkMotion)"""
        torch = _torch()
        dev = "cuda:%d" % self.constraint_.ctx.device
        f = torch.as_tensor(np.ascontiguousarray(frm, dtype=np.float64)).to(dev)
        t = torch.as_tensor(np.ascontiguousarray(to, dtype=np.float64)).to(dev)
        # A large batch takes the shape the library is fastest with: lists of 16 states and at most 128 Newton
        # rounds per edge in the first launch — no creeping edge can hold it — then the few edges that stopped short.
        big = f.shape[0] >= 1024
        # a continuation starts from a stored state other than `from`: lists hold at least two states (the C++ adapter clamps
        # the same way); max_states = 1 therefore behaves like 2 — nothing is ever cut, edges that need more are continued
        cap = max(2, min(self.max_states, 16) if big else self.max_states)
        states, n, ok, its, carry = self.constraint_.discrete_geodesic_batch(f, t, cap, check_target=check_target, want_carry=True,
                                                                             round_budget=128 if big else 0)
        # n_states == cap + 1: these lists did not fit and the traversal stopped there (ok == 2: the edge had spent the
        # call's Newton rounds).  Continue them from their last stored state until they are whole (ccmp_geodesic_batch_ex:
        # the same states as one uninterrupted traversal) — a cut list must never reach the validity test or the caller
        # as if it were complete.
        whole = self.constraint_.continue_geodesics(t, states, n, ok, its, carry, cap, cont_states=max(self.max_states, 2))
        n, ok = n.cpu().numpy(), ok.cpu().numpy()
        rows = [None] * len(n)
        for e, (st_e, ok_e, _) in whole.items():
            rows[e], n[e], ok[e] = st_e, st_e.shape[0], ok_e
        left = [e for e in range(len(n)) if rows[e] is None and (n[e] > cap or ok[e] == 2)]
        if left:  # a truncated or suspended edge must never be reported as reached (ok == 2 is not a bool)
            raise RuntimeError("discreteGeodesicBatch: edges %s were not traversed to their end" % left[:8])
        states = states.cpu().numpy()
        delta = self.constraint_.problem.delta
        out = []
        for e in range(states.shape[0]):
            st = rows[e] if rows[e] is not None else states[e, : n[e]]
            good = bool(ok[e])
            if not interpolate and self.isValid is not None:
                for k in range(1, st.shape[0]):
                    if not self.isValid(st[k]):
                        st = st[:k]
                        good = self.distance(st[-1], to[e]) <= delta  # the loop broke before dist was updated
                        break
            out.append((good, st.copy()))
        return out

    def discreteGeodesic(self, frm, to, interpolate=False, geodesic=None, check_target=False):
        good, st = self.discreteGeodesicBatch(np.asarray(frm).reshape(1, 14), np.asarray(to).reshape(1, 14), interpolate,
                                              check_target)[0]
        if geodesic is not None:
            del geodesic[:]
            geodesic.extend(st)
        return good


def check_motion(space, s1, s2):
    """OMPL `ConstrainedMotionValidator::checkMotion(s1, s2)` as the reference's planner calls it
    (src/planner/stefanBiPRM.cpp:397-398,463-464): isSatisfied(s2) && discreteGeodesic(s1, s2)."""
    return bool(space.discreteGeodesic(s1, s2, False, None, check_target=True))  # both tests in one launch


def geodesic_interpolate(states, t):
    """OMPL `ConstrainedStateSpace::geodesicInterpolate`: the state at fraction t of the piecewise-linear
    geodesic (used by `PathGeometric::interpolate`, src/base/constraints/ConstrainedPlanningCommon.cpp:217)."""
    states = np.asarray(states, dtype=np.float64).reshape(-1, 14)
    n = states.shape[0]
    if n == 1:
        return states[0].copy()
    d = np.sqrt(((states[1:] - states[:-1]) ** 2).sum(axis=1))
    total = d.sum()
    if total == 0.0:
        return states[0].copy()
    target, acc = t * total, 0.0
    for i in range(n - 1):
        if acc + d[i] >= target or i == n - 2:
            tt = 0.0 if d[i] == 0.0 else (target - acc) / d[i]
            return states[i] + min(max(tt, 0.0), 1.0) * (states[i + 1] - states[i])
        acc += d[i]
    return states[-1].copy()


def format_path_matrix(states):
    """`PathGeometric::printAsMatrix`: one state per line, values in C++ default stream format (%g, 6
    significant digits), each followed by a space, and one empty line after the last state — what
    scripts/execute_path.py and visualize_path.py of the reference parse."""
    lines = []
    for s in np.asarray(states, dtype=np.float64).reshape(-1, 14):
        lines.append("".join("%g " % v for v in s))
    return "\n".join(lines) + "\n\n"


def parse_path_matrix(text):
    rows = [[float(v) for v in ln.split()] for ln in text.splitlines() if ln.strip()]
    return np.array(rows, dtype=np.float64).reshape(-1, 14)


# ---- planner-graph dumps (ConstrainedProblem::dumpGraph, ConstrainedPlanningCommon.h:73-87) ---------------------------
_GRAPHML_HEAD = (
    '<?xml version="1.0" encoding="UTF-8"?>\n'
    '<graphml xmlns="http://graphml.graphdrawing.org/xmlns" xmlns:xsi="http://www.w3.org/2001/XMLSchema-instance" '
    'xsi:schemaLocation="http://graphml.graphdrawing.org/xmlns http://graphml.graphdrawing.org/xmlns/1.0/graphml.xsd">\n'
    '  <key id="key0" for="node" attr.name="coords" attr.type="string" />\n'
    '  <key id="key1" for="edge" attr.name="weight" attr.type="double" />\n'
    '  <graph id="G" edgedefault="directed" parse.nodeids="free" parse.edgeids="canonical" parse.order="nodesfirst">\n')


def format_graphml(nodes, edges, weights=None):
    """`PlannerData::printGraphML` as `dumpGraph` writes `<obj>_node_info.graphml`: node data = the state's reals in
    default stream format joined by commas, directed edges in insertion order, edge data = the edge weight."""
    out = [_GRAPHML_HEAD]
    for i, q in enumerate(np.asarray(nodes, dtype=np.float64).reshape(-1, 14)):
        out.append('    <node id="n%d">\n      <data key="key0">%s</data>\n    </node>\n' % (i, ",".join("%g" % v for v in q)))
    for k, (a, b) in enumerate(edges):
        w = 1.0 if weights is None else float(weights[k])
        out.append('    <edge id="e%d" source="n%d" target="n%d">\n      <data key="key1">%g</data>\n    </edge>\n' % (k, a, b, w))
    out.append("  </graph>\n</graphml>\n")
    return "".join(out)


def parse_graphml(text):
    """(nodes (N,14), directed edges, weights) of a `printGraphML` dump"""
    import xml.etree.ElementTree as ET

    ns = "{http://graphml.graphdrawing.org/xmlns}"
    root = ET.fromstring(text)
    ids, nodes, edges, weights = {}, [], [], []
    for n in root.iter(ns + "node"):
        ids[n.get("id")] = len(nodes)
        nodes.append([float(v) for v in n.find(ns + "data").text.split(",")])
    for e in root.iter(ns + "edge"):
        edges.append((ids[e.get("source")], ids[e.get("target")]))
        weights.append(float(e.find(ns + "data").text))
    return np.array(nodes, dtype=np.float64).reshape(-1, 14), edges, weights


def format_graphviz(n_nodes, edges):
    """`PlannerData::printGraphviz` as `dumpGraph` writes `<obj>_graph_info.dot`"""
    return "digraph G {\n" + "".join("%d;\n" % i for i in range(int(n_nodes))) + "".join("%d->%d ;\n" % (a, b) for a, b in edges) + "}\n"
