"""Clip loading and result writers of the Geo4D CLI — host-side, no kernels.

``load_video_batch`` keeps the contract of ``utils/funcs.py:142-179`` (frame indices ``frame_stride * i``, ``video_frames=-1`` =
all frames, last-frame padding, ``(x / 255 - 0.5) * 2`` scaling, ``[b, c, t, h, w]`` layout, ``fps // frame_stride``). The
reference decodes with ``decord`` (absent in this image: used when importable); a clip may also be a directory of image files
(sorted by name), a ``.npy`` / ``.npz`` array ``[t, h, w, 3]`` uint8, or an animated image readable by PIL.

Writers follow ``dust3r/cloud_opt/base_opt_group.py:390-464`` and ``dust3r/utils/vo_eval.py:465-473`` byte for byte where the
format is text (TUM trajectories ``t x y z qw qx qy qz``, ``pred_focal.txt`` / ``pred_intrinsics.txt`` with ``%.6f``,
``frame_%04d.npy``, ``conf_%d.npy``) so that ``viser/visualizer.py`` and the evaluation scripts read them unchanged; the colour
depth previews use matplotlib's 'inferno' map over the 2-98 percentile range of inverse depth like ``vis_sequence_depth``.
"""
import os

import numpy as np
import torch

IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp")


# ---- reading ------------------------------------------------------------------------------------------------------------------
class _ArrayReader:
    def __init__(self, frames, fps=24.0):
        self.frames, self.fps = frames, fps

    def __len__(self):
        return len(self.frames)

    def get_avg_fps(self):
        return self.fps

    def get_batch(self, idx):
        return np.stack([np.asarray(self.frames[i]) for i in idx])


def _resize(frame, width, height):
    from PIL import Image
    if frame.shape[1] == width and frame.shape[0] == height:
        return frame
    return np.asarray(Image.fromarray(frame).resize((width, height), Image.BILINEAR))


def open_clip_reader(path, width, height):
    """A minimal VideoReader (len / get_avg_fps / get_batch -> uint8 [n, h, w, 3]) over whatever this host can decode."""
    if os.path.isdir(path):
        from PIL import Image
        names = sorted(n for n in os.listdir(path) if n.lower().endswith(IMAGE_EXT))
        if not names:
            raise FileNotFoundError(f"{path}: no image files")

        class _Dir(_ArrayReader):
            def get_batch(self, idx):
                return np.stack([_resize(np.asarray(Image.open(os.path.join(path, names[i])).convert("RGB")), width, height) for i in idx])
        return _Dir(names)
    ext = os.path.splitext(path)[1].lower()
    if ext in (".npy", ".npz"):
        arr = np.load(path)
        if ext == ".npz":
            arr = arr[arr.files[0]]
        assert arr.ndim == 4 and arr.shape[-1] == 3 and arr.dtype == np.uint8, "expected uint8 frames [t, h, w, 3]"
        return _ArrayReader([_resize(f, width, height) for f in arr])
    try:
        from decord import VideoReader, cpu   # the reference's decoder (utils/funcs.py:152)
    except ImportError:
        VideoReader = None
    if VideoReader is not None:
        vr = VideoReader(path, ctx=cpu(0), width=width, height=height)

        class _Decord(_ArrayReader):
            def __len__(self):
                return len(vr)

            def get_avg_fps(self):
                return vr.get_avg_fps()

            def get_batch(self, idx):
                return vr.get_batch(idx).asnumpy()
        return _Decord(None)
    if ext in (".gif", ".webp", ".png", ".tif", ".tiff"):
        from PIL import Image, ImageSequence
        im = Image.open(path)
        frames = [_resize(np.asarray(f.convert("RGB")), width, height) for f in ImageSequence.Iterator(im)]
        dur = im.info.get("duration", 0)
        return _ArrayReader(frames, fps=1000.0 / dur if dur else 24.0)
    raise RuntimeError(f"cannot decode {path}: no video decoder in this environment (decord is what the reference uses); "
                       "pass a directory of frames or a .npy / .npz uint8 array [t, h, w, 3]")


def load_video_batch(filepath_list, frame_stride, video_size=(256, 256), video_frames=16):
    """utils/funcs.py:142-179. Returns (frames [b, 3, t, h, w] in [-1, 1], fps_list)."""
    assert frame_stride > 0, "valid frame stride should be a positive interge!"
    fps_list, batch = [], []
    for filepath in filepath_list:
        reader = open_clip_reader(filepath, width=video_size[1], height=video_size[0])
        total = len(reader)
        max_valid = total // frame_stride
        required = total // frame_stride if video_frames < 0 else video_frames
        query = min(required, max_valid)
        frames = torch.from_numpy(np.ascontiguousarray(reader.get_batch([frame_stride * i for i in range(query)])))
        x = (frames.permute(3, 0, 1, 2).float() / 255. - 0.5) * 2
        if max_valid < required:
            x = torch.cat([x] + [x[:, -1:]] * (required - max_valid), dim=1)
        batch.append(x)
        fps_list.append(int(reader.get_avg_fps() / frame_stride))
    return torch.stack(batch, dim=0), fps_list


# ---- writing ------------------------------------------------------------------------------------------------------------------
def c2w_to_tumpose(c2w):
    """base_opt_group.py:29-44: camera-to-world 4x4 -> (x y z qw qx qy qz)."""
    from scipy.spatial.transform import Rotation
    c2w = np.asarray(torch.as_tensor(c2w).detach().cpu().double().numpy())
    qx, qy, qz, qw = Rotation.from_matrix(c2w[:3, :3]).as_quat()
    return np.concatenate([c2w[:3, -1], [qw, qx, qy, qz]])


def get_tum_poses(c2w_poses):
    return [np.stack([c2w_to_tumpose(p) for p in c2w_poses], 0), np.arange(len(c2w_poses)).astype(float)]


def save_tum_poses(path, c2w_poses):
    """vo_eval.py:465-473 (save_trajectory_tum_format): one line per frame, `timestamp x y z qw qx qy qz`, str() formatting."""
    poses, tt = get_tum_poses(c2w_poses)
    tostr = lambda a: " ".join(map(str, a))
    with open(path, "w") as f:
        for t, p in zip(tt, poses):
            f.write(f"{t} {tostr(p[:3])} {tostr(p[3:])}\n")
    return poses


def save_focals(path, focals):
    np.savetxt(path, torch.as_tensor(focals).detach().cpu().numpy(), fmt='%.6f')


def save_intrinsics(path, K):
    np.savetxt(path, torch.as_tensor(K).detach().cpu().reshape(-1, 9).numpy(), fmt='%.6f')


def save_conf_maps(path, conf, prefix="conf"):
    for i, c in enumerate(conf):
        np.save(f'{path}/{prefix}_{i}.npy', torch.as_tensor(c).detach().cpu().numpy())


def colorize_inverse_depth(inv_depth, colormap="inferno"):
    """vis_sequence_depth (base_opt_group.py:75-90): 2-98 percentile range over the sequence, matplotlib listed colormap."""
    import matplotlib
    d = np.asarray(inv_depth, dtype=np.float64)
    lo, hi = np.percentile(d, 2), np.percentile(d, 98)
    colors = np.asarray(matplotlib.colormaps[colormap].colors)
    idx = np.clip(((d - lo) / (hi - lo) * 255).astype(np.int64), 0, 255)
    return colors[idx]


def save_depth_maps(path, depth_maps):
    """base_opt_group.py:436-464: frame_%04d.npy (metric depth), frame_colordepth_%04d.png and colored_depth_maps.gif (inverse depth)."""
    from PIL import Image
    depth = torch.stack([torch.as_tensor(d).detach().cpu().float() for d in depth_maps])
    for i, d in enumerate(depth):
        np.save(f'{path}/frame_{i:04d}.npy', d.numpy())
    colored = colorize_inverse_depth((1 / (depth + 1e-6)).numpy())
    images = []
    for i, c in enumerate(colored):
        p = f'{path}/frame_colordepth_{i:04d}.png'
        Image.fromarray((c * 255).astype(np.uint8)).save(p)
        images.append(Image.open(p))
    images[0].save(f'{path}/colored_depth_maps.gif', save_all=True, append_images=images[1:], duration=100, loop=0)
    return depth


def save_rgb_imgs(path, imgs):
    """frame_%04d.png from float RGB in [0, 1] (base_opt_group.py:422-428 writes the same pixels through cv2 in BGR order)."""
    from PIL import Image
    for i, img in enumerate(imgs):
        Image.fromarray((np.asarray(img) * 255).astype(np.uint8)).save(f'{path}/frame_{i:04d}.png')


# ---- glb export (dust3r/demo.py:56-86 -> dust3r/utils/viz_demo.py:13-58) -------------------------------------------------------------
OPENGL = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]], dtype=np.float64)     # dust3r/viz.py:338-341


def _glb_bytes(meshes, transform):
    """Binary glTF 2.0 with one node per primitive. meshes: list of dict(mode, positions f32 [n, 3], colors u8 [n, 4] or None).
    `transform` (4x4) is applied to every node, as trimesh.Scene.apply_transform does."""
    import json
    import struct
    buf = bytearray()
    views, accessors, gl_meshes, nodes = [], [], [], []

    def add(data, target, ctype, atype, normalized=False, minmax=False):
        while len(buf) % 4:
            buf.append(0)
        views.append({"buffer": 0, "byteOffset": len(buf), "byteLength": data.nbytes, "target": target})
        buf.extend(data.tobytes())
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": int(data.shape[0]), "type": atype}
        if normalized:
            acc["normalized"] = True
        if minmax:
            acc["min"], acc["max"] = [float(v) for v in data.min(0)], [float(v) for v in data.max(0)]
        accessors.append(acc)
        return len(accessors) - 1
    for m in meshes:
        pos = np.ascontiguousarray(m["positions"], dtype=np.float32)
        attrs = {"POSITION": add(pos, 34962, 5126, "VEC3", minmax=True)}
        if m.get("colors") is not None:
            attrs["COLOR_0"] = add(np.ascontiguousarray(m["colors"], dtype=np.uint8), 34962, 5121, "VEC4", normalized=True)
        gl_meshes.append({"primitives": [{"attributes": attrs, "mode": m["mode"]}]})
        nodes.append({"mesh": len(gl_meshes) - 1, "matrix": [float(v) for v in np.asarray(transform, np.float64).T.reshape(-1)]})   # column-major
    doc = {"asset": {"version": "2.0", "generator": "geo4d_amd.io"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}],
           "nodes": nodes, "meshes": gl_meshes, "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(buf)}]}
    js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
    js += b" " * (-len(js) % 4)
    while len(buf) % 4:
        buf.append(0)
    total = 12 + 8 + len(js) + 8 + len(buf)
    return b"glTF" + struct.pack("<II", 2, total) + struct.pack("<I", len(js)) + b"JSON" + js + struct.pack("<I", len(buf)) + b"BIN\x00" + bytes(buf)


def _camera_wireframe(c2w, focal, imsize, screen_width):
    """Line list of a camera pyramid: apex at the camera centre, base = the image plane at depth `screen_width * focal / W` scaled so
    that the base is `screen_width` wide (dust3r/viz.py add_scene_cam draws a 4-sided cone of that size; here its 8 edges)."""
    W, H = imsize
    d = screen_width * focal / W
    hw, hh = screen_width / 2.0, screen_width * H / W / 2.0
    corners = np.array([[-hw, -hh, d], [hw, -hh, d], [hw, hh, d], [-hw, hh, d]], np.float64)
    world = corners @ c2w[:3, :3].T + c2w[:3, 3]
    apex = c2w[:3, 3]
    segs = []
    for k in range(4):
        segs += [apex, world[k], world[k], world[(k + 1) % 4]]
    return np.asarray(segs, np.float32)


def save_glb(path, imgs, pts3d, masks, focals, cams2world, cam_size=0.05, show_cam=True, cam_color=None):
    """convert_scene_output_to_glb(as_pointcloud=True) without trimesh: the masked points of every image with their colours as ONE
    POINTS primitive, one LINES primitive per camera (pyramid edges, viridis-like colour ramp unless `cam_color` [n][3] in 0-255), and
    the reference's scene transform inv(cams2world[0] @ OPENGL @ rot_y(180 deg)) on every node.
    imgs [n, H, W, 3] in [0, 1]; pts3d [n, H, W, 3]; masks [n, H, W] bool; focals [n]; cams2world [n, 4, 4]. Returns `path`."""
    to_np = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    imgs, pts3d, masks, focals, cams2world = (to_np(v) for v in (imgs, pts3d, masks, focals, cams2world))
    assert len(pts3d) == len(masks) <= len(imgs) <= len(cams2world) == len(np.reshape(focals, -1))
    masks = masks.astype(bool)
    pts = np.concatenate([p[m] for p, m in zip(pts3d, masks)]).reshape(-1, 3)
    col = np.concatenate([im[m] for im, m in zip(imgs, masks)]).reshape(-1, 3)
    col = np.concatenate([np.clip(col * 255.0 + 0.5, 0, 255).astype(np.uint8), np.full((len(col), 1), 255, np.uint8)], 1)
    meshes = [dict(mode=0, positions=pts, colors=col)]
    n = len(cams2world)
    if show_cam:
        H, W = imgs.shape[1:3]
        for i in range(n):
            if cam_color is not None:
                c = np.asarray(cam_color[i] if isinstance(cam_color, list) else cam_color, np.float64)
            else:
                t = i / max(n - 1, 1)                                       # dark violet -> green -> yellow (viridis end points)
                c = 255 * np.array([0.267 + 0.726 * t ** 2, 0.005 + 0.90 * t, 0.329 + 0.3 * np.sin(np.pi * t) - 0.19 * t])
            seg = _camera_wireframe(cams2world[i].astype(np.float64), float(np.reshape(focals, -1)[i]), (W, H), cam_size)
            cc = np.tile(np.concatenate([np.clip(c, 0, 255), [255.0]]).astype(np.uint8), (len(seg), 1))
            meshes.append(dict(mode=1, positions=seg, colors=cc))
    rot = np.eye(4)
    rot[:3, :3] = np.array([[-1.0, 0, 0], [0, 1, 0], [0, 0, -1.0]])                     # Rotation.from_euler('y', 180 deg)
    transform = np.linalg.inv(cams2world[0].astype(np.float64) @ OPENGL @ rot)
    data = _glb_bytes(meshes, transform)
    with open(path, "wb") as f:
        f.write(data)
    return path
