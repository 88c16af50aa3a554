"""Host-side mirror of the reference's constrained state space for the projector hot path:
`jy_ProjectedStateSampler` (src/base/jy_ProjectedStateSpace.cpp:5-29), `jy_ProjectedStateSpace::
discreteGeodesic` (:32-96) and the path-matrix format the reference dumps
(PathGeometric::printAsMatrix, src/base/constraints/ConstrainedPlanningCommon.cpp:219-222), all on
top of the batched GPU entry points of KinematicChainConstraint."""
import numpy as np

__all__ = ["jy_ProjectedStateSampler", "jy_ProjectedStateSpace", "check_motion", "geodesic_interpolate", "format_path_matrix",
           "parse_path_matrix"]


def _torch():
    import torch

    return torch


class jy_ProjectedStateSampler:
    """`sampleUniform(state)` / `sampleUniformNear` / `sampleGaussian` with the reference's signatures
    (state = numpy (14,), written in place).  sampleUniform is served from a buffer that ONE GPU launch
    of `batch` fused sample->project->enforceBounds fills whenever it runs empty, invisible to the
    planner; the reference ignores project()'s result here, and so does this class."""

    def __init__(self, constraint, seed=0, batch=4096):
        self.constraint_ = constraint
        self.seed = int(seed)
        self.batch = int(batch)
        self._next_index = 0  # global sample counter: the stream of samples does not depend on `batch`
        self._buf = None
        self._pos = 0
        self._near_index = 0

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

    def sampleUniformNear(self, state, near, distance):
        torch = _torch()
        ref = torch.as_tensor(np.ascontiguousarray(near, dtype=np.float64)).to("cuda:%d" % self.constraint_.ctx.device)
        q, _, _, _ = self.constraint_.sample_near_project_batch(self.seed ^ 0x4E454152, self._near_index, ref, distance, 1,
                                                               want_iters=False)
        self._near_index += 1
        state[:] = q.cpu().numpy()[0]

    def sampleGaussian(self, state, mean, stdDev):
        torch = _torch()
        ref = torch.as_tensor(np.ascontiguousarray(mean, dtype=np.float64)).to("cuda:%d" % self.constraint_.ctx.device)
        q, _, _, _ = self.constraint_.sample_gaussian_project_batch(self.seed ^ 0x47415553, self._near_index, ref, stdDev, 1,
                                                                   want_iters=False)
        self._near_index += 1
        state[:] = q.cpu().numpy()[0]


class jy_ProjectedStateSpace:
    """`discreteGeodesic(from, to, interpolate, geodesic)` with the reference's semantics.  The GPU runs
    the traversal without the validity test; for interpolate == False the host then applies
    `StateValidityChecker::isValid` (MoveIt in the reference; any callable here) in order and cuts the
    list at the first rejected state — exactly where the reference's loop would have stopped
    (jy_ProjectedStateSpace.cpp:65-68)."""

    def __init__(self, constraint, isValid=None, max_states=256):
        self.constraint_ = constraint
        self.isValid = isValid
        self.max_states = int(max_states)

    def setDelta(self, delta):
        self.constraint_.problem.delta = float(delta)

    def setLambda(self, lam):
        self.constraint_.problem.lambda_ = float(lam)

    def distance(self, a, b):
        return float(np.sqrt(np.sum((np.asarray(a) - np.asarray(b)) ** 2)))

    def discreteGeodesicBatch(self, frm, to, interpolate=False):
        """numpy (E,14) x2 -> list of (ok, states (n,14)) per edge"""
        torch = _torch()
        dev = "cuda:%d" % self.constraint_.ctx.device
        f = torch.as_tensor(np.ascontiguousarray(frm, dtype=np.float64)).to(dev)
        t = torch.as_tensor(np.ascontiguousarray(to, dtype=np.float64)).to(dev)
        states, n, ok, _ = self.constraint_.discrete_geodesic_batch(f, t, self.max_states)
        states, n, ok = states.cpu().numpy(), n.cpu().numpy(), ok.cpu().numpy()
        delta = self.constraint_.problem.delta
        out = []
        for e in range(states.shape[0]):
            st = states[e, : n[e]]
            good = bool(ok[e])
            if not interpolate and self.isValid is not None:
                for k in range(1, st.shape[0]):
                    if not self.isValid(st[k]):
                        st = st[:k]
                        good = self.distance(st[-1], to[e]) <= delta  # the loop broke before dist was updated
                        break
            out.append((good, st.copy()))
        return out

    def discreteGeodesic(self, frm, to, interpolate=False, geodesic=None):
        good, st = self.discreteGeodesicBatch(np.asarray(frm).reshape(1, 14), np.asarray(to).reshape(1, 14), interpolate)[0]
        if geodesic is not None:
            del geodesic[:]
            geodesic.extend(st)
        return good


def check_motion(space, s1, s2):
    """OMPL `ConstrainedMotionValidator::checkMotion(s1, s2)` as the reference's planner calls it
    (src/planner/stefanBiPRM.cpp:397-398,463-464): isSatisfied(s2) && discreteGeodesic(s1, s2)."""
    return bool(space.constraint_.isSatisfied(s2)) and bool(space.discreteGeodesic(s1, s2, False, None))


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
    significant digits), each followed by a space — what scripts/execute_path.py and visualize_path.py
    of the reference parse."""
    lines = []
    for s in np.asarray(states, dtype=np.float64).reshape(-1, 14):
        lines.append("".join("%g " % v for v in s))
    return "\n".join(lines) + "\n"


def parse_path_matrix(text):
    rows = [[float(v) for v in ln.split()] for ln in text.splitlines() if ln.strip()]
    return np.array(rows, dtype=np.float64).reshape(-1, 14)
