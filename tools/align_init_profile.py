#!/usr/bin/env python3
"""Where does the alignment INIT of a clip go (post_optimization(..., align=False): post-decode math, GroupAligner construction,
registration chain, focals, depth maps)? cProfile over the synthetic scene bench.py's clip mode uses. usage: align_init_profile.py [n_frames]"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geo4d_amd.align import post_optimization
from geo4d_amd.pipeline import window_slices
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
slices = window_slices(n, 4, 16)
maps, traj = bench.synthetic_scene_maps(slices, 16, 320, 512, dev)
args = dict(n_iter=500, pose_schedule="linear", temporal_smoothing_weight=0.015, translation_weight=1.0)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    scene = post_optimization(slices, maps, traj, args, align=False)
    torch.cuda.synchronize(); pr.disable()
    print(f"init pass {rep}: {time.perf_counter() - t0:.2f} s", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
