"""TEST INFRASTRUCTURE ONLY — restatement of the Plücker ray-map -> camera step of the reference.

Follows scripts/evaluation/test_geo4d.py:539-557 (raymap_to_camera_matrix), utils/rays.py:387-433 (cameras_from_plucker:
centre crop to a square, nearest "resize" to the same size, normalised directions, frame 0 as the reference rays),
utils/rays.py:301-368 (rays_to_cameras: camera centre = least-squares intersection of the rays, rotation = optimal alignment of
the frame's directions onto the reference directions, T = -R^T c), utils/rays.py:134-155,174-188 (origins of Plücker rays =
d x m), utils/normalize.py:25-51 (intersect_skew_lines_high_dim) and utils/rays.py:579-595 (compute_optimal_rotation_alignment,
SVD Kabsch with the reflection fix). fp64 throughout. Pinned by tests/golden/rays.pt, which is produced by the reference's own
functions (pytorch3d's PerspectiveCameras stubbed as a plain container; see tests/golden/generate.py).
"""
import torch


def crop_square(x):
    """[T, H, W, 3] -> centre square (utils/rays.py:399-417, `crop:-crop`). For H == W the reference never assigns
    num_patches_x and raises UnboundLocalError; here a square frame is simply used whole."""
    T, H, W, _ = x.shape
    if H > W:
        c = (H - W) // 2
        return x[:, c:-c]
    if W > H:
        c = (W - H) // 2
        return x[:, :, c:-c]
    return x


@torch.no_grad()
def raymap_to_camera_matrix(raymap, crossmap):
    """raymap, crossmap [1, 3, T, H, W] -> camera-to-world matrices [T, 4, 4] (ref_raymap=None, as the window loop calls it)."""
    d = crop_square(raymap[0].permute(1, 2, 3, 0).double())
    m = crop_square(crossmap[0].permute(1, 2, 3, 0).double())
    T = d.shape[0]
    d = torch.nn.functional.normalize(d.reshape(T, -1, 3), dim=-1)
    m = m.reshape(T, -1, 3)
    p = torch.cross(d, m, dim=-1)                                   # ray origins (closest point to the world origin)
    eye = torch.eye(3, dtype=torch.float64)
    proj = eye - d[..., :, None] * d[..., None, :]                  # I - d d^T per ray
    A = proj.sum(dim=1)
    b = (proj @ p[..., None]).sum(dim=1)
    centers = torch.linalg.lstsq(A, b).solution[..., 0]             # [T, 3]
    ref = d[0]
    P = torch.eye(4, dtype=torch.float64).repeat(T, 1, 1)
    for i in range(T):
        Hm = d[i].T @ ref                                           # B^T A with A = reference directions, B = this frame's
        U, _, Vh = torch.linalg.svd(Hm)
        s = torch.sign(torch.linalg.det(U @ Vh))
        R = U @ torch.diag(torch.stack([torch.ones_like(s), torch.ones_like(s), s])) @ Vh
        t_w2c = -(R.T @ centers[i])
        P[i, :3, :3] = R
        P[i, :3, 3] = -(R @ t_w2c)
    return P
