"""Proxy-geometry clearance: the pre-filter in front of the reference's MoveIt validity test.

`KinematicChainValidityChecker::isValid` (src/kinematics/KinematicChain.cpp:94-123 of the reference) hands the 14 joint
values to MoveIt's `PlanningScene::checkCollision` under an allowed-collision matrix; the scene holds the robot's URDF
meshes, a box ("sub_table", KinematicChain.cpp:25-30) and the grasped object attached to the left hand
(ConstrainedPlanningCommon.cpp:30-31).  MoveIt and the meshes stay on the host (out of scope); this module
is the test a planner runs BEFORE asking MoveIt: spheres attached to link frames, static boxes, an
allowed-pair matrix, and per state the smallest signed distance over the pairs that are not allowed
(`ccmp_clearance_batch`, include/ccmp.h).  The proxies are the caller's — the reference ships no link geometry — so
nothing here is compared with MoveIt; the GPU result is bit-identical to oracle/ccmp_oracle.c:orc_clearance, which
places the spheres with the projector's own forward kinematics.

`ProxyValidityChecker` mirrors the reference class's surface (`addBox`, `attachObject`, `isValid`) over that kernel.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import CcmpBox, CcmpSphere, check

__all__ = ["FRAME_WORLD", "frame", "ProxyScene", "ProxyValidityChecker", "skeleton_spheres", "default_allowed", "allow",
           "object_points_to_hand", "decode_pair", "GROUP_OBJECT", "GROUP_ENVIRONMENT"]

FRAME_WORLD = -1
MAX_SPHERES = 64
MAX_BOXES = 8
GROUP_OBJECT = 18       # by convention: groups 0..17 = frame codes of the two arms
GROUP_ENVIRONMENT = 19


def frame(arm, k):
    """CCMP_FRAME(arm, k): k = 0..6 the body of joint k (panda_link1..7), 7 the hand frame, 8 the arm's base"""
    if arm not in (0, 1) or not 0 <= k <= 8:
        raise ValueError("arm 0/1, k 0..8")
    return arm * 9 + k


def allow(allowed, g, h):
    """mark the pair of groups (g, h) as never tested (AllowedCollisionMatrix::setEntry(g, h, true))"""
    allowed[g] |= 1 << h
    allowed[h] |= 1 << g
    return allowed


def _chain_position(k):
    return -1 if k == 8 else k  # base, bodies 0..6, hand 7


def default_allowed(adjacency=3):
    """Groups = frame codes.  Pairs on the same arm whose links are at most `adjacency` apart along the chain are never
    tested (neighbouring links overlap by construction and the Panda's elbow folds links three apart against each other —
    MoveIt's SRDF disables such pairs the same way); the object group is allowed against both hands (the reference's touch links, KinematicChain.cpp:86-91)."""
    allowed = [0] * 32
    for arm in (0, 1):
        for k in range(9):
            for kk in range(9):
                if abs(_chain_position(k) - _chain_position(kk)) <= adjacency:
                    allow(allowed, arm * 9 + k, arm * 9 + kk)
    for arm in (0, 1):
        allow(allowed, GROUP_OBJECT, arm * 9 + 7)
    allow(allowed, GROUP_OBJECT, GROUP_OBJECT)
    allow(allowed, GROUP_ENVIRONMENT, GROUP_ENVIRONMENT)
    return allowed


# reject threshold that goes with the default skeleton spheres (see skeleton_spheres): they are not inscribed in the links
DEFAULT_SKELETON_MARGIN = -0.03


def skeleton_spheres(problem, radius=0.055, spacing=0.08, hand_extent=0.09, hand_radius=0.035):
    """Spheres strung along the joint-to-joint skeleton of both arms, built from the problem's own chain constants
    (joint offsets of panda_rbdl.cpp:128-130, the hand offset of :124-126): one at every joint origin, more every
    `spacing` metres along each segment, attached to the body that carries the segment; group = frame code.  The radius is
    about the physical radius of the Panda's arm links (~0.055 m); the hand's spheres are thinner (two hands hold one
    object a few centimetres apart).  They are a COARSE default, not the robot's meshes and NOT provably inscribed: the
    straight joint-to-joint segments cut corners of the real links (the forearm's (-0.0825, 0, 0.384) diagonal leaves the
    L-shaped link by centimetres), so a slightly negative clearance does not prove a collision.  ProxyValidityChecker
    therefore rejects with these spheres only below DEFAULT_SKELETON_MARGIN (an overlap deeper than 3 cm); strict
    "negative clearance = collision" semantics are for inscribed proxies the caller supplies.  What is checked: every
    state the reference itself validated with MoveIt — its recorded paths and roadmap vertices — keeps a positive
    clearance with them (tests/test_scene.py)."""
    off = np.ctypeslib.as_array(problem.offset).reshape(2, 7, 3)
    ee = np.ctypeslib.as_array(problem.ee).reshape(2, 3)
    out = []

    def along(frame_code, vec, include_end, r=radius):
        n = max(1, int(np.ceil(np.linalg.norm(vec) / spacing)))
        for t in range(0, n + (1 if include_end else 0)):
            out.append((frame_code, frame_code, tuple(vec * (t / n)), r))

    for arm in (0, 1):
        along(frame(arm, 8), off[arm, 0], False)                   # base -> joint 0, rigid with the base
        for k in range(7):
            nxt = off[arm, k + 1] if k < 6 else ee[arm]            # in body k's axes (parallel to the base at q = 0)
            if np.linalg.norm(nxt) < 1e-9:
                out.append((frame(arm, k), frame(arm, k), (0.0, 0.0, 0.0), radius))
            else:
                along(frame(arm, k), nxt, False)
        along(frame(arm, 7), np.array([0.0, 0.0, hand_extent]), True, hand_radius)  # hand and fingers along the hand frame's approach axis
    return out


def object_points_to_hand(problem, points_obj):
    """Points given in the OBJECT frame -> the hand frame of arm 0, which carries the object
    (attachObject("stefan", "panda_left_hand", ...), ConstrainedPlanningCommon.cpp:31; t_o7 of :110-111)."""
    R = np.ctypeslib.as_array(problem.t_o7_R).reshape(2, 3, 3)[0]
    p = np.ctypeslib.as_array(problem.t_o7_p).reshape(2, 3)[0]
    pts = np.asarray(points_obj, dtype=np.float64).reshape(-1, 3)
    return (pts - p) @ R  # R^T (x - p), row-wise


def decode_pair(code):
    """pair code -> (sphere index, ('sphere', j) | ('box', b)) or None"""
    if code < 0:
        return None
    i, j = code & 0xFF, code >> 8
    return (i, ("sphere", j) if j < MAX_SPHERES else ("box", j - MAX_SPHERES))


class ProxyScene:
    """ccmp_scene.  spheres: (frame, group, centre(3), radius); boxes: (group, centre(3), R(3x3), half(3))."""

    def __init__(self, constraint, spheres, boxes=(), allowed=None):
        self.constraint = constraint
        self.spheres = [(int(f), int(g), tuple(float(v) for v in c), float(r)) for f, g, c, r in spheres]
        self.boxes = [(int(g), tuple(float(v) for v in c), np.asarray(R, dtype=np.float64).reshape(3, 3).copy(),
                       tuple(float(v) for v in h)) for g, c, R, h in boxes]
        self.allowed = None if allowed is None else [int(v) & 0xFFFFFFFF for v in allowed]
        sa = (CcmpSphere * max(1, len(self.spheres)))()
        for k, (f, g, c, r) in enumerate(self.spheres):
            sa[k] = CcmpSphere(f, g, (C.c_double * 3)(*c), r)
        ba = (CcmpBox * max(1, len(self.boxes)))()
        for k, (g, c, R, h) in enumerate(self.boxes):
            ba[k] = CcmpBox(g, 0, (C.c_double * 3)(*c), (C.c_double * 9)(*R.reshape(9)), (C.c_double * 3)(*h))
        al = None if self.allowed is None else (C.c_uint32 * 32)(*self.allowed)
        self._h = C.c_void_p()
        check(_lib.lib().ccmp_scene_create(constraint.ctx.handle, sa, len(self.spheres), ba, len(self.boxes), al, C.byref(self._h)),
              "ccmp_scene_create")

    @property
    def num_pairs(self):
        return _lib.lib().ccmp_scene_num_pairs(self._h)

    def clearance_batch(self, q, margin=0.0, ok=None, want_pair=True, stream=None):
        """(clearance (B,) f64, pair (B,) i32 or None, free (B,) u8) for a (B,14) float64 CUDA tensor;
        free = (ok is None or ok) & (clearance > margin): feed it to compact_valid behind a projection"""
        import torch

        from .constraint import _stream_handle

        c = self.constraint
        c._need_problem()
        c._check_q(q)
        B = q.shape[0]
        clr = torch.empty(B, dtype=torch.float64, device=q.device)
        pair = torch.empty(B, dtype=torch.int32, device=q.device) if want_pair else None
        free = torch.empty(B, dtype=torch.uint8, device=q.device)
        check(_lib.lib().ccmp_clearance_batch(c.ctx.handle, C.byref(c.problem), self._h, q.data_ptr(),
                                              ok.data_ptr() if ok is not None else None, B, float(margin), clr.data_ptr(),
                                              pair.data_ptr() if pair is not None else None, free.data_ptr(),
                                              _stream_handle(stream)), "ccmp_clearance_batch")
        return clr, pair, free

    def clearance(self, x, margin=0.0):
        """host states (14,) or (B,14) -> (clearance, pair, free) as numpy (scalars for one state)"""
        c = self.constraint
        c._need_problem()
        x = np.ascontiguousarray(x, dtype=np.float64)
        single = x.ndim == 1
        q = x.reshape(-1, 14)
        B = q.shape[0]
        clr = np.empty(B)
        pair = np.empty(B, dtype=np.int32)
        free = np.empty(B, dtype=np.uint8)
        check(_lib.lib().ccmp_clearance_host(c.ctx.handle, C.byref(c.problem), self._h, q.ctypes.data_as(C.POINTER(C.c_double)), B,
                                             float(margin), clr.ctypes.data_as(C.POINTER(C.c_double)),
                                             pair.ctypes.data_as(C.POINTER(C.c_int32)), free.ctypes.data_as(C.POINTER(C.c_uint8))),
              "ccmp_clearance_host")
        if single:
            return float(clr[0]), int(pair[0]), bool(free[0])
        return clr, pair, free

    def close(self):
        if self._h:
            _lib.lib().ccmp_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ProxyValidityChecker:
    """The surface of `KinematicChainValidityChecker` (include/.../kinematics/KinematicChain.h:175-219) over proxy
    geometry: the reference's constructor adds the "sub_table" box (KinematicChain.cpp:25-30), `attachObject` hangs the
    object on the left hand and allows it against its touch links, `isValid(state)` answers for one state.

        chk = ProxyValidityChecker(constraint)             # both arms' skeleton spheres + sub_table
        chk.attachObject(points_in_object_frame, radius)   # optional: the grasped object as spheres
        chk.isValid(x)                                     # False = clearance <= margin
        chk.filter_batch(q, ok)                            # device flags for a whole batch

    `inner` (optional) is the exact checker to ask when the proxies do NOT refuse the state (MoveIt in the reference's
    build).  `margin` None = 0 for proxies the caller supplies (with INSCRIBED proxies the pre-filter then only ever
    rejects states that really collide) and DEFAULT_SKELETON_MARGIN = -3 cm for the default skeleton spheres, which are
    not inscribed: the pre-filter must not refuse states MoveIt would accept — that would change the planner's answers."""

    SUB_TABLE = (GROUP_ENVIRONMENT, (0.65, 0.0, 1.1), np.eye(3), (0.325, 0.5, 0.1))  # addBox(dim (0.65, 1.0, 0.2), pose)

    def __init__(self, constraint, spheres=None, boxes=None, allowed=None, margin=None, inner=None):
        self.constraint = constraint
        constraint._need_problem()
        self._spheres = list(skeleton_spheres(constraint.problem) if spheres is None else spheres)
        self._boxes = [self.SUB_TABLE] if boxes is None else list(boxes)
        self._allowed = list(default_allowed() if allowed is None else allowed)
        self.margin = float((DEFAULT_SKELETON_MARGIN if spheres is None else 0.0) if margin is None else margin)
        self.inner = inner
        self._scene = None

    def addBox(self, dim, pose_position, id=None, rotation=None, group=GROUP_ENVIRONMENT):
        """KinematicChain.h:194-211: dim = full extents, pose in the world ("/base") frame"""
        half = tuple(0.5 * float(v) for v in dim)
        R = np.eye(3) if rotation is None else np.asarray(rotation, dtype=np.float64).reshape(3, 3)
        self._boxes.append((group, tuple(float(v) for v in pose_position), R, half))
        self._scene = None

    def addSphere(self, frame_code, centre, radius, group=None):
        """group None = the frame's own group (a link) or the environment group (a world-frame sphere)"""
        if group is None:
            group = frame_code if frame_code >= 0 else GROUP_ENVIRONMENT
        self._spheres.append((frame_code, group, tuple(float(v) for v in centre), float(radius)))
        self._scene = None

    def attachObject(self, points_obj, radius, touch_both_hands=True):
        """the grasped object as spheres around `points_obj` (object frame), carried by arm 0's hand"""
        for p in object_points_to_hand(self.constraint.problem, points_obj):
            self._spheres.append((frame(0, 7), GROUP_OBJECT, tuple(p), float(radius)))
        if not touch_both_hands:
            self._allowed[GROUP_OBJECT] &= ~(1 << frame(1, 7))
            self._allowed[frame(1, 7)] &= ~(1 << GROUP_OBJECT)
        self._scene = None

    @property
    def scene(self):
        if self._scene is None:
            self._scene = ProxyScene(self.constraint, self._spheres, self._boxes, self._allowed)
        return self._scene

    def isValid(self, x):
        clr, _, free = self.scene.clearance(x, self.margin)
        if not free:
            return False
        return True if self.inner is None else bool(self.inner(x))

    def filter_batch(self, q, ok=None, stream=None):
        """device flags: ok & (clearance > margin)"""
        return self.scene.clearance_batch(q, self.margin, ok=ok, want_pair=False, stream=stream)[2]
