#!/usr/bin/env python3
"""Independent high-precision vectors for FK and the closed-chain residual (mpmath, 50 digits).

This is the builder's own second formulation, NOT a restatement of the oracle: the arm is a product
of Craig modified-DH 4x4 transforms (the textbook Panda model the reference's table at
src/kinematics/panda_rbdl.cpp:97-99 encodes) instead of the product-of-exponentials chain the
oracle and the kernels use, and the rotation error is the angle of R_c R_0^T from its trace /
skew part instead of a quaternion angularDistance.  Agreement to ~1e-13 pins FK + residual of both
the oracle and the HIP kernels to the mathematics, independently of either implementation.

    python tests/golden/make_mp_vectors.py   ->  tests/golden/mp_vectors.json
"""
import json
import os
import random

import mpmath as mp
import yaml

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))

AL = [0, -mp.pi / 2, mp.pi / 2, mp.pi / 2, -mp.pi / 2, mp.pi / 2, mp.pi / 2]
A = [0, 0, 0, mp.mpf("0.0825"), mp.mpf("-0.0825"), 0, mp.mpf("0.088")]
D = [mp.mpf("0.333"), 0, mp.mpf("0.316"), 0, mp.mpf("0.384"), 0, 0]
LB = [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973]
UB = [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973]


def dh(a, d, al, th):
    ct, st, ca, sa = mp.cos(th), mp.sin(th), mp.cos(al), mp.sin(al)
    return mp.matrix([[ct, -st, 0, a], [st * ca, ct * ca, -sa, -sa * d], [st * sa, ct * sa, ca, ca * d], [0, 0, 0, 1]])


def base(index):
    T = mp.eye(4)
    if index == 0:
        T[0, 3], T[1, 3], T[2, 3] = 0, mp.mpf("0.3"), mp.mpf("1.006")
    elif index == 1:
        T[0, 3], T[1, 3], T[2, 3] = 0, mp.mpf("-0.3"), mp.mpf("1.006")
    else:
        T[0, 3], T[1, 3], T[2, 3] = mp.mpf("1.35"), mp.mpf("0.3"), mp.mpf("1.006")
        T[0, 0] = -1
        T[1, 1] = -1
    return T


def hand_pose(index, q):
    """world pose of the hand frame: flange offset 0.107 along z7, then Rz(-pi/4)."""
    T = base(index)
    for i in range(7):
        T = T * dh(A[i], D[i], AL[i], mp.mpf(q[i]))
    F = mp.eye(4)
    F[2, 3] = mp.mpf("0.107")
    c, s = mp.cos(-mp.pi / 4), mp.sin(-mp.pi / 4)
    Rz = mp.matrix([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    return T * F * Rz


def inv(T):
    R = T[0:3, 0:3].T
    p = -R * T[0:3, 3]
    out = mp.eye(4)
    out[0:3, 0:3] = R
    out[0:3, 3] = p
    return out


def rot_angle(R):
    """angle in [0, pi] of a rotation matrix, robust near 0 and pi"""
    v = mp.matrix([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = mp.sqrt(v[0] ** 2 + v[1] ** 2 + v[2] ** 2) / 2
    c = (R[0, 0] + R[1, 1] + R[2, 2] - 1) / 2
    return mp.atan2(s, c)


def residual(idx, chain0, x):
    T1 = hand_pose(idx[0], x[:7])
    T2 = hand_pose(idx[1], x[7:])
    C = inv(T2) * T1
    dp = C[0:3, 3] - chain0[0:3, 3]
    E = C[0:3, 0:3] * chain0[0:3, 0:3].T
    return [mp.sqrt(dp[0] ** 2 + dp[1] ** 2 + dp[2] ** 2), rot_angle(E)]


def f17(x):
    return float(mp.nstr(x, 20))


def main():
    rng = random.Random(20260927)
    out = {}
    for obj in ("Wine_Bottle", "dumbbell", "stefan"):
        cfg = yaml.safe_load(open(os.path.join(HERE, "config", obj + ".yaml")))
        arms = sorted([(cfg["arm1"]["name"], cfg["arm1"]["index"]), (cfg["arm2"]["name"], cfg["arm2"]["index"])])
        idx = [arms[0][1], arms[1][1]]
        q0 = cfg["start_joint"]
        C0 = inv(hand_pose(idx[1], q0[7:])) * hand_pose(idx[0], q0[:7])
        cases = []
        xs = [list(q0)] + [[rng.uniform(LB[j % 7], UB[j % 7]) for j in range(14)] for _ in range(12)]
        # a state close to the manifold exercises the small-angle / small-norm end of the residual
        xs.append([q0[j] + 1e-4 * rng.uniform(-1, 1) for j in range(14)])
        for x in xs:
            T1 = hand_pose(idx[0], x[:7])
            T2 = hand_pose(idx[1], x[7:])
            f = residual(idx, C0, x)
            cases.append({
                "x": [float(v) for v in x],
                "R1": [f17(T1[i, j]) for i in range(3) for j in range(3)], "p1": [f17(T1[i, 3]) for i in range(3)],
                "R2": [f17(T2[i, j]) for i in range(3) for j in range(3)], "p2": [f17(T2[i, 3]) for i in range(3)],
                "f": [f17(f[0]), f17(f[1])],
            })
        out[obj] = {
            "arm_index": idx,
            "init_R": [f17(C0[i, j]) for i in range(3) for j in range(3)],
            "init_p": [f17(C0[i, 3]) for i in range(3)],
            "cases": cases,
        }
    with open(os.path.join(HERE, "mp_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote mp_vectors.json")


if __name__ == "__main__":
    main()
