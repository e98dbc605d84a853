"""RANSAC-PnP camera initialisation (geo4d_amd/pnp.py; reference init_im_poses.py:824-865 through cv2.solvePnPRansac, restated with a
seeded sampler): closed-form cases with known answers — exact pinhole projections of random and PLANAR scenes, gross outliers, the
candidate-focal choice of fast_pnp, and its failure contract."""
import numpy as np
import pytest

from geo4d_amd import pnp


def _scene(n, seed, planar=False):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, 3)) * [2.0, 1.5, 1.0] + [0, 0, 5.0]
    if planar:
        X[:, 2] = 5.0 + 0.3 * X[:, 0]
    ang = rng.uniform(-0.4, 0.4, 3)
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    R = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    t = rng.uniform(-0.5, 0.5, 3)
    return X, R, t


def _project(X, R, t, K):
    Xc = X @ R.T + t
    return np.stack([K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]], 1)


@pytest.mark.parametrize("planar", [False, True])
def test_ransac_pnp_recovers_the_pose_with_outliers(planar):
    K = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])
    X, R, t = _scene(3000, 1, planar)
    pix = _project(X, R, t, K)
    rng = np.random.default_rng(2)
    bad = rng.choice(len(X), 900, replace=False)                   # 30 % gross outliers
    pix[bad] += rng.uniform(-200, 200, (900, 2))
    ok, Re, te, inl = pnp.solve_pnp_ransac(X, pix, K, iterations=100, reproj=5.0, seed=0)
    assert ok and np.abs(Re - R).max() < 1e-5 and np.abs(te - t).max() < 1e-4
    good = np.setdiff1d(np.arange(len(X)), bad)
    assert np.isin(good, inl).mean() > 0.999 and np.isin(bad, inl).mean() < 0.05
    ok2, R2, t2, inl2 = pnp.solve_pnp_ransac(X, pix, K, iterations=100, reproj=5.0, seed=0)
    assert np.array_equal(Re, R2) and np.array_equal(inl, inl2)        # seeded: run-to-run identical


def test_fast_pnp_picks_the_focal_candidate_and_returns_cam_to_world():
    H, W, f = 240, 320, 260.0        # 3 % of the image size moves edge pixels by > 5 px: the candidates are distinguishable
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]])
    rng = np.random.default_rng(5)
    depth = rng.uniform(2.0, 6.0, (H, W))
    grid = pnp.pixel_grid(H, W).astype(np.float64)
    cam = np.concatenate([(grid - [W / 2, H / 2]) / f * depth[..., None], depth[..., None]], -1)       # camera-frame points
    _, R, t = _scene(4, 7)
    c2w = np.eye(4)
    c2w[:3, :3], c2w[:3, 3] = R.T, -R.T @ t
    pts = cam @ c2w[:3, :3].T + c2w[:3, 3]                                                              # world-frame point map
    msk = rng.uniform(size=(H, W)) > 0.2
    S = max(H, W)
    # the true focal is the "+3 % of the image size" candidate of the guess it is given (init_im_poses.py:841-842)
    got = pnp.fast_pnp(pts, f - 0.03 * S, msk, niter_PnP=100)
    assert got is not None and abs(got[0] - f) < 1e-9
    assert np.abs(got[1] - c2w).max() < 1e-5
    none_focal = pnp.fast_pnp(pts, None, msk, niter_PnP=30)                                            # 63 log-spaced candidates
    assert none_focal is not None and abs(np.log(none_focal[0] / f)) < np.log(3 * S / (S / 2)) / 62 and np.abs(none_focal[1][:3, :3] - c2w[:3, :3]).max() < 0.05
    assert pnp.fast_pnp(pts, f, np.zeros((H, W), bool)) is None                                         # < 4 points
    assert pnp.fast_pnp(rng.normal(size=(H, W, 3)), f, msk, niter_PnP=5) is None or True                # noise: no crash, no contract on the value
