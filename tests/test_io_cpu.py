"""N4 host I/O (geo4d_amd/io.py): the clip-loading contract of utils/funcs.py:142-179 and the on-disk result formats of
dust3r/cloud_opt/base_opt_group.py:390-464 / dust3r/utils/vo_eval.py:465-473."""
import os

import numpy as np
import torch


def test_load_video_batch_contract(tmp_path):
    from geo4d_amd.io import load_video_batch
    from PIL import Image
    frames = (np.random.RandomState(0).rand(10, 24, 32, 3) * 255).astype(np.uint8)
    npy = tmp_path / "clip.npy"
    np.save(npy, frames)
    x, fps = load_video_batch([str(npy)], frame_stride=2, video_size=(24, 32), video_frames=4)
    assert x.shape == (1, 3, 4, 24, 32) and fps == [12]
    assert torch.equal(x[0, :, 1], (torch.from_numpy(frames[2]).permute(2, 0, 1).float() / 255. - 0.5) * 2)      # index = stride * i
    xa, _ = load_video_batch([str(npy)], frame_stride=3, video_size=(24, 32), video_frames=-1)                   # all frames: 10 // 3
    assert xa.shape[2] == 3
    xp, _ = load_video_batch([str(npy)], frame_stride=4, video_size=(24, 32), video_frames=5)                    # 2 valid frames + 3 x last
    assert xp.shape[2] == 5 and torch.equal(xp[0, :, 2], xp[0, :, 1]) and torch.equal(xp[0, :, 4], xp[0, :, 1])
    d = tmp_path / "frames"
    d.mkdir()
    for i, f in enumerate(frames[:3]):
        Image.fromarray(f).save(d / f"{i:03d}.png")
    xd, _ = load_video_batch([str(d)], frame_stride=1, video_size=(24, 32), video_frames=3)
    assert torch.equal(xd, load_video_batch([str(npy)], 1, (24, 32), 3)[0])
    xr, _ = load_video_batch([str(d)], frame_stride=1, video_size=(12, 16), video_frames=2)                      # decoder-side resize
    assert xr.shape == (1, 3, 2, 12, 16) and xr.abs().max() <= 1.0


def test_writers_formats(tmp_path):
    from geo4d_amd import io
    from scipy.spatial.transform import Rotation
    T = 3
    c2w = torch.eye(4).repeat(T, 1, 1)
    rots = Rotation.from_euler("xyz", [[0.1 * i, -0.2 * i, 0.05 * i] for i in range(T)])
    c2w[:, :3, :3] = torch.from_numpy(rots.as_matrix()).float()
    c2w[:, :3, 3] = torch.tensor([[0.5 * i, 1.0, -i] for i in range(T)])
    poses = io.save_tum_poses(tmp_path / "pred_traj.txt", c2w)
    rows = [l.split() for l in open(tmp_path / "pred_traj.txt").read().strip().splitlines()]
    assert len(rows) == T and all(len(r) == 8 for r in rows) and rows[1][0] == "1.0"
    q = rots.as_quat()                                                   # scipy: x y z w ; TUM line: qw qx qy qz
    got = np.array([[float(v) for v in r] for r in rows])
    assert np.allclose(got[:, 1:4], c2w[:, :3, 3].numpy(), atol=1e-6) and np.allclose(got[:, 4], q[:, 3], atol=1e-6)
    assert np.allclose(got[:, 5:8], q[:, :3], atol=1e-6) and np.allclose(poses, got[:, 1:], atol=1e-6)
    K = torch.tensor([[[400., 0, 256], [0, 400, 160], [0, 0, 1]]]).repeat(T, 1, 1)
    io.save_intrinsics(tmp_path / "pred_intrinsics.txt", K)
    io.save_focals(tmp_path / "pred_focal.txt", K[:, 0, 0:1])
    assert open(tmp_path / "pred_intrinsics.txt").readline().split()[:3] == ["400.000000", "0.000000", "256.000000"]
    assert open(tmp_path / "pred_focal.txt").readline().strip() == "400.000000"
    depth = [torch.rand(8, 12) * 5 + 0.5 for _ in range(T)]
    io.save_depth_maps(str(tmp_path), depth)
    io.save_conf_maps(str(tmp_path), [torch.rand(8, 12) for _ in range(T)])
    assert np.allclose(np.load(tmp_path / "frame_0002.npy"), depth[2].numpy())
    for n in ("frame_colordepth_0000.png", "colored_depth_maps.gif", "conf_1.npy"):
        assert os.path.getsize(tmp_path / n) > 0


def test_glb_export_is_a_valid_binary_gltf(tmp_path):
    """save_glb (dust3r/utils/viz_demo.py:13-58, as_pointcloud=True): container layout, chunk sizes, accessor counts / bounds, the
    reference's scene transform, and the point / colour payload read back bit for bit."""
    import json
    import struct
    import numpy as np
    from geo4d_amd import io as gio
    rng = np.random.default_rng(0)
    n, H, W = 3, 6, 8
    imgs, pts = rng.uniform(size=(n, H, W, 3)).astype(np.float32), rng.normal(size=(n, H, W, 3)).astype(np.float32)
    masks = rng.uniform(size=(n, H, W)) > 0.3
    c2w = np.tile(np.eye(4), (n, 1, 1))
    c2w[:, :3, 3] = rng.normal(size=(n, 3))
    th = 0.3
    c2w[0, :3, :3] = [[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]]
    path = gio.save_glb(str(tmp_path / "scene.glb"), torch.from_numpy(imgs), torch.from_numpy(pts), masks, np.full(n, 50.0), c2w)
    raw = open(path, "rb").read()
    magic, version, total = struct.unpack("<4sII", raw[:12])
    assert magic == b"glTF" and version == 2 and total == len(raw)
    jlen, jtype = struct.unpack("<I4s", raw[12:20])
    doc = json.loads(raw[20:20 + jlen])
    blen, btype = struct.unpack("<I4s", raw[20 + jlen:28 + jlen])
    assert jtype == b"JSON" and btype == b"BIN\x00" and jlen % 4 == 0 and 28 + jlen + blen == total and doc["buffers"][0]["byteLength"] == blen
    assert len(doc["meshes"]) == 1 + n and doc["meshes"][0]["primitives"][0]["mode"] == 0 and all(m["primitives"][0]["mode"] == 1 for m in doc["meshes"][1:])
    binbuf = raw[28 + jlen:]
    acc = doc["accessors"][doc["meshes"][0]["primitives"][0]["attributes"]["POSITION"]]
    view = doc["bufferViews"][acc["bufferView"]]
    got = np.frombuffer(binbuf[view["byteOffset"]:view["byteOffset"] + view["byteLength"]], np.float32).reshape(-1, 3)
    want = np.concatenate([p[m] for p, m in zip(pts, masks)])
    assert acc["count"] == int(masks.sum()) and np.array_equal(got, want) and np.allclose(acc["min"], want.min(0)) and np.allclose(acc["max"], want.max(0))
    cacc = doc["accessors"][doc["meshes"][0]["primitives"][0]["attributes"]["COLOR_0"]]
    cview = doc["bufferViews"][cacc["bufferView"]]
    cols = np.frombuffer(binbuf[cview["byteOffset"]:cview["byteOffset"] + cview["byteLength"]], np.uint8).reshape(-1, 4)
    assert cacc["normalized"] and np.abs(cols[:, :3].astype(np.float32) / 255 - np.concatenate([im[m] for im, m in zip(imgs, masks)])).max() < 0.5 / 255 + 1e-6
    assert all(doc["accessors"][m["primitives"][0]["attributes"]["POSITION"]]["count"] == 16 for m in doc["meshes"][1:])      # 8 edges per camera
    rot = np.diag([-1.0, 1.0, -1.0, 1.0])
    M = np.array(doc["nodes"][0]["matrix"]).reshape(4, 4).T
    assert np.allclose(M, np.linalg.inv(c2w[0] @ gio.OPENGL @ rot)) and all(nd["matrix"] == doc["nodes"][0]["matrix"] for nd in doc["nodes"])
