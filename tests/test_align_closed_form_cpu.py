"""Closed-form known-answer tests of the third-party arithmetic the alignment stage restates because the packages are absent
(VERDICT r2 item 5): roma.rigid_points_registration (weighted Umeyama with scale), evo's PosePath3D.align_origin and the RPE rotation
metric (dust3r/utils/vo_eval.py:174-266, optimizer_group.py:242-268). Both the product restatement (geo4d_amd/align.py, host-side
functions: no GPU needed) and the oracle's (oracle/align.py) are held to the same closed-form answers."""
import math

import numpy as np
import pytest
import torch

from geo4d_amd import align as galign
from oracle import align as oalign


def _rot(axis, deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def _pose(R, t):
    M = np.eye(4)
    M[:3, :3], M[:3, 3] = R, t
    return M


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_weighted_umeyama_recovers_a_known_similarity(impl):
    g = torch.Generator().manual_seed(0)
    x = torch.randn((500, 3), generator=g)
    R = torch.from_numpy(_rot("z", 40) @ _rot("x", -25)).float()
    s, T = 1.7, torch.tensor([0.3, -1.2, 2.0])
    y = s * x @ R.t() + T
    w = torch.rand(500, generator=g) + 0.1
    def reg(a, b, ww):       # product returns (s, R, T); roma's order (the oracle keeps it) is (R, T, s)
        if impl == "product":
            return galign.rigid_points_registration(a, b, ww)
        Ro, To, so = oalign.rigid_points_registration(a, b, ww, compute_scaling=True)
        return so, Ro, To
    se, Re, Te = reg(x, y, w)
    assert abs(float(se) - s) < 1e-5 and (Re - R).abs().max() < 1e-5 and (Te - T).abs().max() < 1e-4
    # weights matter: corrupt half of the points and give them zero weight -> the same answer
    y2 = y.clone()
    y2[::2] += 5 * torch.randn((250, 3), generator=g)
    w2 = w.clone()
    w2[::2] = 0
    se, Re, Te = reg(x, y2, w2)
    assert abs(float(se) - s) < 1e-5 and (Re - R).abs().max() < 1e-5
    # a mirrored target must NOT be matched by a reflection: det(R) = +1 always
    ym = y * torch.tensor([1.0, 1.0, -1.0])
    Rm = reg(x, ym, w)[1]
    assert abs(float(torch.det(Rm.double())) - 1.0) < 1e-5


@pytest.mark.parametrize("fn", [galign.align_origin_and_rpe, oalign.align_origin_and_rpe])
def test_align_origin_and_rpe_closed_form(fn):
    S = 6
    # reference: rotates 10 deg / frame about y while translating; estimate: the same path seen from another frame, 13 deg / frame
    ref = np.stack([_pose(_rot("y", 10 * k), [0.1 * k, 0, 0.05 * k]) for k in range(S)])
    W = _pose(_rot("x", 33) @ _rot("z", -12), [2.0, -1.0, 0.5])                       # arbitrary change of world frame
    est = np.stack([W @ _pose(_rot("y", 13 * k), [0.1 * k, 0, 0.05 * k]) for k in range(S)])
    P, rpe = fn(est, ref)
    assert np.allclose(P @ est[0], ref[0], atol=1e-12)                                 # align_origin: first poses coincide
    assert np.allclose(P, ref[0] @ np.linalg.inv(est[0]), atol=1e-12)
    assert abs(rpe - 3.0) < 1e-9                                                        # every relative rotation is off by exactly 3 degrees
    # relative motion is frame independent: an exact copy in another frame has zero RPE; mixed errors give their rmse
    P0, rpe0 = fn(np.stack([W @ r for r in ref]), ref)
    assert rpe0 < 1e-6
    errs = [1.0, -2.0, 0.0, 4.0, 2.0]
    ang = np.concatenate([[0.0], np.cumsum([10 + e for e in errs])])
    est2 = np.stack([W @ _pose(_rot("y", ang[k]), [0.1 * k, 0, 0.05 * k]) for k in range(S)])
    assert abs(fn(est2, ref)[1] - math.sqrt(np.mean(np.square(errs)))) < 1e-9


def test_quaternion_round_trip_and_signed_log():
    for axis, deg in (("x", 10), ("y", 179), ("z", -120), ("y", 91)):
        R = torch.from_numpy(_rot(axis, deg)).float()
        q = galign.rotmat_to_quat(R)
        assert abs(float(q.norm()) - 1) < 1e-6 and (galign.quat_to_rotmat(q) - R).abs().max() < 1e-6
        assert (oalign.unitquat_to_rotmat(oalign.rotmat_to_unitquat(R.double())).float() - R).abs().max() < 1e-6
    x = torch.tensor([-3.0, -0.5, 0.0, 0.25, 7.0])
    assert torch.allclose(galign.signed_expm1(galign.signed_log1p(x)), x, atol=1e-6)
