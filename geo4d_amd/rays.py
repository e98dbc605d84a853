"""Plücker ray map -> camera — the names the window loop of the reference uses
(``raymap_to_camera_matrix``, scripts/evaluation/test_geo4d.py:539-557; ``cameras_from_plucker``, utils/rays.py:387-433).

The reference copies the two decoded maps to the host and runs pytorch3d / torch.linalg there, once per window, which
serialises the window loop on a device->host sync. Here the whole step is two HIP launches on the stream
(``csrc/rays.hip``: fp64 moment reduction + a 3x3 solve / Jacobi-SVD per frame) and the matrices stay on the device.
Deliberate difference: the reference raises UnboundLocalError for square frames (rays.py:399-417 never sets num_patches_x
when H == W); a square frame is simply used whole here.
"""
import torch

from . import ops


def raymap_to_camera_matrix(raymap, crossmap, ref_raymap=None):
    """raymap, crossmap [1, 3, T, H, W] fp32 on the HIP device -> camera-to-world matrices [T, 4, 4] on the same device."""
    if ref_raymap is not None:
        raise NotImplementedError("ref_raymap is never passed by the window loop (test_geo4d.py:458); only frame 0 of the "
                                  "window as reference is built")
    return ops.plucker_cameras(raymap.float(), crossmap.float())


def cameras_from_plucker(raydir, raymoment, ref_raymap=None):
    """-> (R [T,3,3], T_w2c [T,3], centers [T,3]): the fields of the PerspectiveCameras object + centres that
    utils/rays.py:cameras_from_plucker returns (R as stored there, T = -R^T c)."""
    P = raymap_to_camera_matrix(raydir, raymoment, ref_raymap)
    R, c = P[:, :3, :3], P[:, :3, 3]
    return R, -torch.bmm(R.transpose(1, 2), c[..., None])[..., 0], c
