"""Per-image camera initialisation by RANSAC-PnP — the host-side start of the global alignment.

The reference initialises every camera with ``fast_pnp`` (dust3r/cloud_opt/init_im_poses.py:824-865): the image's chained world-frame
point map against its own pixel grid through ``cv2.solvePnPRansac(..., iterationsCount=niter_PnP, reprojectionError=5,
flags=cv2.SOLVEPNP_SQPNP)``, tried for the focal it was given and for focal -/+ 3 % of the image size (or 63 log-spaced focals when it
has none), keeping the candidate with the most inliers. OpenCV is not installed here and its RANSAC draws from OpenCV's own RNG, so
bit-reproducing it is not possible; this module restates the ALGORITHM with a seeded sampler:

  * minimal model: 6 correspondences -> pose by the orthogonal-iteration PnP of Lu, Hager & Mjolsness (object-space error, globally
    convergent, handles planar point sets; closed-form absolute orientation per step), started from a direct linear transform when
    that is well conditioned;
  * consensus: reprojection error < `reproj` pixels; the best model is re-fitted on its inliers (same solver) and re-scored, like
    OpenCV's final refinement on the consensus set;
  * `fast_pnp`: the reference's candidate-focal loop and its return convention (best focal, camera-to-world 4x4).

One call per image and candidate focal at clip start-up (n_images x 3 solves of a few thousand sub-sampled points): host numpy, not
on the hot path. What differs from the reference is stated where it differs: the minimal solver (orthogonal iteration instead of
SQPnP) and the sampler (numpy PCG64 with an explicit seed instead of cv::RNG).
"""
import numpy as np


def _absolute_orientation(X, Q, w=None):
    """R, t minimising sum w |R X + t - Q|^2 (Kabsch)."""
    w = np.ones(len(X)) if w is None else w
    w = w / w.sum()
    xm, qm = (w[:, None] * X).sum(0), (w[:, None] * Q).sum(0)
    H = (w[:, None] * (Q - qm)).T @ (X - xm)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    return R, qm - R @ xm


def _dlt_pose(X, b):
    """Direct linear transform for [R | t] from >= 6 bearing vectors b ~ R X + t; returns None when ill conditioned."""
    n = len(X)
    A = np.zeros((2 * n, 12))
    Xh = np.concatenate([X, np.ones((n, 1))], 1)
    A[0::2, 0:4], A[0::2, 8:12] = Xh * b[:, 2:3], -Xh * b[:, 0:1]
    A[1::2, 4:8], A[1::2, 8:12] = Xh * b[:, 2:3], -Xh * b[:, 1:2]
    _, S, Vt = np.linalg.svd(A)
    if S[-2] < 1e-9 * S[0]:
        return None
    P = Vt[-1].reshape(3, 4)
    if np.linalg.det(P[:, :3]) < 0:
        P = -P
    U, S3, Vt3 = np.linalg.svd(P[:, :3])
    R = U @ Vt3
    if np.linalg.det(R) < 0:
        return None
    return R, P[:, 3] / S3.mean()


def pnp_orthogonal_iteration(X, b, R=None, t=None, iters=30, w=None, _retry=True):
    """Lu-Hager-Mjolsness: minimise the object-space error sum |(I - V_i)(R X_i + t)|^2 with V_i = b_i b_i^T / (b_i^T b_i) the
    projector onto the i-th line of sight. X [n, 3] world points, b [n, 3] bearings (K^-1 [u, v, 1]). Returns world->camera R, t."""
    bn = b / np.linalg.norm(b, axis=1, keepdims=True)
    V = bn[:, :, None] * bn[:, None, :]
    n = len(X)
    w = np.ones(n) if w is None else w
    wn = w / w.sum()
    Vbar = (wn[:, None, None] * V).sum(0)
    Tfac = np.linalg.inv(np.eye(3) - Vbar)
    if R is None:
        init = _dlt_pose(X, b) if n >= 6 else None
        R = init[0] if init is not None else np.eye(3)

    def t_of(Rm):
        RX = X @ Rm.T
        return Tfac @ (wn[:, None] * (np.einsum("nij,nj->ni", V, RX) - RX)).sum(0)
    t = t_of(R) if t is None else t
    for _ in range(iters):
        Q = np.einsum("nij,nj->ni", V, X @ R.T + t)          # points projected on their lines of sight
        Rn, _ = _absolute_orientation(X, Q, w)
        done = np.abs(Rn - R).max() < 1e-13
        R = Rn
        t = t_of(R)
        if done:
            break
    if _retry and np.mean((X @ R.T + t)[:, 2]) < 0:          # converged to the mirrored solution behind the camera: restart from its flip
        R, t = pnp_orthogonal_iteration(X, b, R=np.diag([-1.0, -1.0, 1.0]) @ R, t=None, iters=iters, w=w, _retry=False)
    return R, t


def reprojection_error(X, pix, K, R, t):
    Xc = X @ R.T + t
    z = np.where(np.abs(Xc[:, 2]) < 1e-12, 1e-12, Xc[:, 2])
    uv = np.stack([K[0, 0] * Xc[:, 0] / z + K[0, 2], K[1, 1] * Xc[:, 1] / z + K[1, 2]], 1)
    err = np.linalg.norm(uv - pix, axis=1)
    return np.where(Xc[:, 2] > 0, err, np.inf)


def solve_pnp_ransac(X, pix, K, iterations=100, reproj=5.0, seed=0, sample=6, max_points=4096, confidence=0.99):
    """cv2.solvePnPRansac's contract on (object points [n, 3], image points [n, 2], K): returns (success, R world->camera, t, inlier
    indices). Seeded; see the module docstring for what differs from OpenCV."""
    X, pix, K = np.asarray(X, np.float64), np.asarray(pix, np.float64), np.asarray(K, np.float64)
    n = len(X)
    if n < sample:
        return False, None, None, np.zeros(0, np.int64)
    rng = np.random.Generator(np.random.PCG64(seed))
    sub = np.arange(n) if n <= max_points else np.sort(rng.choice(n, max_points, replace=False))     # consensus is scored on a sub-sample
    Xs, ps = X[sub], pix[sub]
    Kinv = np.linalg.inv(K)
    bs = np.concatenate([ps, np.ones((len(ps), 1))], 1) @ Kinv.T
    best = (0, None, None, None)
    it, needed = 0, iterations
    while it < min(iterations, needed):
        it += 1
        idx = rng.choice(len(Xs), sample, replace=False)
        R, t = pnp_orthogonal_iteration(Xs[idx], bs[idx], iters=15)
        inl = reprojection_error(Xs, ps, K, R, t) < reproj
        cnt = int(inl.sum())
        if cnt > best[0]:
            best = (cnt, R, t, inl)
            ratio = cnt / len(Xs)                                                                 # the standard adaptive stop
            p_all = ratio ** sample                                # probability that a sample is all-inlier
            needed = np.inf if p_all < 1e-9 else (0 if p_all >= 1 else np.log(1 - confidence) / np.log(1 - p_all))
    if best[0] < sample:
        return False, None, None, np.zeros(0, np.int64)
    _, R, t, inl = best
    for _ in range(2):                                                                           # refit on the consensus set
        R2, t2 = pnp_orthogonal_iteration(Xs[inl], bs[inl], R=R, t=None, iters=500)      # (linear convergence on near-planar sets)
        inl2 = reprojection_error(Xs, ps, K, R2, t2) < reproj
        if inl2.sum() < inl.sum():
            break
        R, t, inl = R2, t2, inl2
    full = np.nonzero(reprojection_error(X, pix, K, R, t) < reproj)[0]
    return True, R, t, full


def pixel_grid(H, W):
    """init_im_poses.py:820-821: [H, W, 2] of (x, y)."""
    return np.mgrid[:W, :H].T.astype(np.float32)


def fast_pnp(pts3d, focal, msk, pp=None, niter_PnP=10, seed=0):
    """init_im_poses.py:824-865. pts3d [H, W, 3] world points of ONE image, msk [H, W] bool, focal in pixels or None.
    Returns None (fewer than 4 masked points / no consensus) or (best focal, camera-to-world 4x4 float64 ndarray)."""
    pts3d, msk = np.asarray(pts3d, np.float64), np.asarray(msk, bool)
    if msk.sum() < 4:
        return None
    H, W, _ = pts3d.shape
    pixels = pixel_grid(H, W)
    S = max(W, H)
    if focal is None:
        tentative = list(np.geomspace(S / 2, S * 3, 63))
    else:
        tentative = [float(focal)] + list(np.geomspace(-0.03 * S + focal, 0.03 * S + focal, 2))
    pp = (W / 2, H / 2) if pp is None else tuple(float(v) for v in np.asarray(pp).reshape(-1))
    best = (0,)
    for f in tentative:
        K = np.array([[f, 0, pp[0]], [0, f, pp[1]], [0, 0, 1]], np.float64)
        ok, R, T, inliers = solve_pnp_ransac(pts3d[msk], pixels[msk], K, iterations=niter_PnP, reproj=5.0, seed=seed)
        if ok and len(inliers) > best[0]:
            best = (len(inliers), R, T, f)
    if not best[0]:
        return None
    _, R, T, f = best
    M = np.eye(4)
    M[:3, :3], M[:3, 3] = R.T, -R.T @ T                       # inverse of the world->camera rigid transform
    return f, M
