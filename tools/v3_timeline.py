"""Where a phase of the phased conv_gemm K loop spends its cycles (profiling build only: make ABLATION=1, tile hints 88 / 98).
Waves 0 and 4 of workgroup 0 stamp s_memtime four times per phase over slabs 8..23 of their first tile:
  0 start of the issue / read segment   1 after its counted wait (before the barrier)   2 after the barrier (start of the MFMA segment)
  3 end of the MFMA segment + cursor work (before the closing barrier)
Prints, per wave group and phase, the mean cycles of: issue+read+wait | barrier 1 | MFMA+cursor | barrier 2."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from geo4d_amd import ops, pack  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    which = sys.argv[1] if len(sys.argv) > 1 else "conv"
    torch.manual_seed(0)
    if which == "conv":          # L0 conv3x3 320 -> 320 on the 160x320 tile
        F_, H, W, Cin, N, tile = 16, 40, 64, 320, 320, 88
        x = torch.randn((F_ * H * W, Cin), device=dev)
        w = pack.split_bf16(torch.randn((N, 9 * Cin), device=dev) / (9 * Cin) ** 0.5)
        b = torch.randn((N,), device=dev)
        xa = ops.SplitAct.wrap(pack.split_bf16(x))
        run = lambda t: ops.conv2d(xa, w, b, F=F_, Hin=H, Win=W, KH=3, KW=3, pad=1, tile_hint=t, split_k=1)[0]
    else:                        # L2 ffout 5120 -> 1280 (plain linear, K = 5120) on the 192x256 tile
        M, K, N, tile = 2560, 5120, 1280, 98
        x = torch.randn((M, K), device=dev)
        w = pack.split_bf16(torch.randn((N, K), device=dev) / K ** 0.5)
        xa = ops.SplitAct.wrap(pack.split_bf16(x))
        run = lambda t: ops.linear(xa, w, None, tile_hint=t, split_k=1)
    ref = run(72 if tile == 88 else 71)
    ws = ops.workspace(dev)[0]
    ws.zero_()
    out = run(tile)
    torch.cuda.synchronize()
    assert torch.equal(out, ref), "the stamped build must not change the result"
    st = ws[:4096].view(torch.int64).cpu().reshape(2, 16, 16).double()
    names = ["issue+read+wait", "barrier 1", "mfma+cursor", "barrier 2"]
    for g in range(2):
        s = st[g]
        print(f"wave group {g}: slab time {((s[1:, 0] - s[:-1, 0]).mean()):.0f} cycles")
        for ph in range(4):
            a, b_, c, d = s[:, 4 * ph], s[:, 4 * ph + 1], s[:, 4 * ph + 2], s[:, 4 * ph + 3]
            if ph < 3:
                vals = [b_ - a, c - b_, d - c, s[:, 4 * ph + 4] - d]
            else:                # the closing barrier of phase 3 ends at the next slab's first stamp
                vals = [b_ - a, c - b_, d - c, s[1:, 0] - d[:-1]]
            print("  phase %d: " % ph + "  ".join("%s %6.0f" % (n, v.mean()) for n, v in zip(names, vals)))


if __name__ == "__main__":
    main()
