"""Global alignment sharded over the GPUs of one node (BASELINE.json north_star: "... dust3r/cloud_opt point-map alignment shard over the
8 GPUs"; SURVEY.md §8(e) last row).

What shards: the residual terms of ``LightPointCloudGroupOptimizer.forward`` (optimizer_group.py:440-525) are a SUM over windows
(groups) — the confidence-weighted point-map residual and the inverse-depth residual touch one window's prediction each — while the
parameters are per image (log-depth map, pose), per window (sim(3) ``pw_poses``, s / t of the depth fit, trajectory alignment) or
global (shared focal). So windows are dealt to ranks in CONTIGUOUS blocks (consecutive windows share 12 of their 16 images: a block
keeps those images on one rank), every rank evaluates the fused residual kernel over its own windows only, and per iteration
  * ONE small all-reduce sums [ loss | d/d im_poses (n x 7) | d/d im_focals | d/d pw_poses (G x 8) | late-term parameters ] (a few KB), and
  * the d/d log-depth rows of the SHARED images - images whose windows live on more than one rank: the block boundaries, 12 images each -
    travel by a NEIGHBOUR HALO EXCHANGE (round 6, the default): a rank sends its partial rows of the images it shares with rank q to q and
    receives q's (`batch_isend_irecv`: point-to-point over one xGMI link per neighbour), then every rank that touches an image adds the
    partial rows of all its touching ranks IN ASCENDING RANK ORDER - the same sum, bit for bit, on each of them, so their copies of the
    image's depth map stay identical under the replicated Adam update. 128-frame clip on 8 ranks: 12 images x 655 KB = 7.9 MB to each
    neighbour instead of the 55 MB (84 shared images) every rank pushed through the ring all-reduce of rounds 3-5 (`exchange="allreduce"`
    keeps that form for A/B runs).
Depth maps of images that belong to one rank only never leave it until the end: their gradients are complete locally, and
`gather_depthmaps` assembles the final maps from each image's owner. Every rank then applies the identical Adam update to the replicated small parameters, so the ranks
stay bit-identical without a broadcast. Pose-only terms (temporal smoothing, trajectory) are a few dozen flops: rank 0 evaluates them.

The start-up of the inverse-depth term (``_set_st_depth``, one least-absolute-deviation fit per window) shards the same way: a rank
fits its own windows, `merge_rows` all-reduces the [G, 3] table of (s, t, delta score).

Backend: torch.distributed — "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests (tests/test_dist_cpu.py drives this class
with the oracle's objective: world 2 == world 1 to fp32 round-off).
"""
import os

import torch
import torch.distributed as dist


def partition_windows(num_windows, rank, world):
    """Contiguous block of windows of `rank` (sizes differ by at most one). Every rank computes the same table."""
    base, extra = divmod(num_windows, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def make_shard(groups, n_images, rank=None, world=None, group=None):
    """The shard `post_optimization` runs under, or None = replicated optimisation: with fewer windows than ranks (a clip shorter than
    16 + 4 (world - 1) frames at stride 4) some ranks would own NO window - an empty slot list, which the fused residual kernel
    refuses while the other ranks wait in the per-iteration all-reduce. Such a clip is seconds of work: every rank runs the whole
    optimisation (bit-identical results on every rank, no collective). The decision depends on (G, world) only, so all ranks agree."""
    shard = AlignShard(groups, n_images, rank=rank, world=world, group=group)
    return shard if shard.active and len(shard.groups) >= shard.world else None


class AlignShard:
    def __init__(self, groups, n_images, rank=None, world=None, group=None, exchange=None):
        self.group = group
        self.exchange = exchange or os.environ.get("GEO4D_ALIGN_EXCHANGE", "halo")
        assert self.exchange in ("halo", "allreduce"), self.exchange
        self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if dist.is_initialized() else 0)
        self.groups = [list(g) for g in groups]
        self.G, self.n = len(self.groups), n_images
        self.blocks = [partition_windows(self.G, r, self.world) for r in range(self.world)]
        self.local_groups = self.blocks[self.rank]
        touch = [set() for _ in range(n_images)]
        for r, blk in enumerate(self.blocks):
            for g in blk:
                for i in self.groups[g]:
                    touch[i].add(r)
        self.owner = [min(t) if t else 0 for t in touch]               # the rank whose copy of an image's depth map is final
        self.shared = [i for i, t in enumerate(touch) if len(t) > 1]   # images whose depth gradient needs the all-reduce
        self.primary = self.rank == 0                                   # evaluates the pose-only terms
        self._shared_idx = None
        # halo exchange tables: the images this rank shares with each peer, and - per distinct set of touching ranks this rank belongs to -
        # the images with that set (their partial rows are added in ascending rank order)
        self.touch = [tuple(sorted(t)) for t in touch]
        mine = [i for i in self.shared if self.rank in self.touch[i]]
        self.peers = sorted({q for i in mine for q in self.touch[i] if q != self.rank})
        self.peer_images = {q: [i for i in mine if q in self.touch[i]] for q in self.peers}
        self.sets = {}
        for i in mine:
            self.sets.setdefault(self.touch[i], []).append(i)
        self._halo_idx = None

    @property
    def active(self):
        return self.world > 1

    def _all_reduce(self, t):
        if self.active:
            dist.all_reduce(t, group=self.group)
        return t

    def _global_rank(self, q):
        return dist.get_global_rank(self.group, q) if self.group is not None else q

    def _halo_exchange(self, depth):
        """d/d log-depth rows of the shared images: send this rank's partial rows to every peer that touches them, receive theirs, add the
        partial rows of all touching ranks in ascending rank order (identical result on every touching rank)."""
        if not self.peers:
            return
        dev = depth.device
        if self._halo_idx is None or self._halo_idx["dev"] != dev:
            self._halo_idx = {"dev": dev, "peer": {q: torch.tensor(v, dtype=torch.long, device=dev) for q, v in self.peer_images.items()},
                              "pos": {q: {i: k for k, i in enumerate(v)} for q, v in self.peer_images.items()},
                              "set": {s_: torch.tensor(v, dtype=torch.long, device=dev) for s_, v in self.sets.items()}}
        idx = self._halo_idx
        # (gloo has no device-tensor send / recv: the CPU tests and the 2-ranks-on-one-GPU rig stage through host memory; RCCL sends device buffers)
        stage = depth.is_cuda and dist.get_backend(self.group) == "gloo"
        send = {q: depth.index_select(0, idx["peer"][q]).contiguous() for q in self.peers}
        if stage:
            send = {q: t.cpu() for q, t in send.items()}
        recv = {q: torch.empty_like(send[q]) for q in self.peers}
        ops = []
        for q in self.peers:
            ops.append(dist.P2POp(dist.isend, send[q], self._global_rank(q), self.group))
            ops.append(dist.P2POp(dist.irecv, recv[q], self._global_rank(q), self.group))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        if stage:
            recv = {q: t.to(dev) for q, t in recv.items()}
        for s_, imgs in self.sets.items():
            rows = idx["set"][s_]
            total = None
            for r in s_:                                             # ascending rank order: every touching rank forms the same sum
                if r == self.rank:
                    part = depth.index_select(0, rows)
                else:
                    pos = torch.tensor([idx["pos"][r][i] for i in imgs], dtype=torch.long, device=dev)
                    part = recv[r].index_select(0, pos)
                total = part if total is None else total + part
            depth.index_copy_(0, rows, total)

    def reduce(self, loss, grads):
        """Sum the local objective over ranks IN PLACE: loss (0-dim) and every small gradient in one flat all-reduce; the depth-map
        gradient rows of the shared images by the neighbour halo exchange (default) or appended to that all-reduce (exchange="allreduce")."""
        if not self.active:
            return loss, grads
        depth = grads["im_depthmaps"]
        if self.exchange == "halo":
            keys = sorted(k for k in grads if k != "im_depthmaps")
            flat = self._all_reduce(torch.cat([loss.reshape(1).float()] + [grads[k].reshape(-1).float() for k in keys]))
            off = 1
            loss = flat[0]
            for k in keys:
                n = grads[k].numel()
                grads[k] = flat[off:off + n].reshape(grads[k].shape).to(grads[k].dtype)
                off += n
            self._halo_exchange(depth)
            return loss, grads
        if self._shared_idx is None or self._shared_idx.device != depth.device:
            self._shared_idx = torch.tensor(self.shared, dtype=torch.long, device=depth.device)
        keys = sorted(k for k in grads if k != "im_depthmaps")
        parts = [loss.reshape(1).float()] + [grads[k].reshape(-1).float() for k in keys]
        if len(self.shared):
            parts.append(depth.index_select(0, self._shared_idx).reshape(-1))
        flat = self._all_reduce(torch.cat(parts))
        off = 1
        loss = flat[0]
        for k in keys:
            n = grads[k].numel()
            grads[k] = flat[off:off + n].reshape(grads[k].shape).to(grads[k].dtype)
            off += n
        if len(self.shared):
            depth.index_copy_(0, self._shared_idx, flat[off:].reshape(len(self.shared), -1))
        return loss, grads

    def merge_rows(self, table, local_rows):
        """[G, k] table whose rows `local_rows` were computed here: zero the others, all-reduce -> complete table on every rank."""
        if not self.active:
            return table
        keep = torch.zeros(table.shape[0], 1, dtype=table.dtype, device=table.device)
        keep[list(local_rows)] = 1
        return self._all_reduce(table * keep)

    def gather_depthmaps(self, depth):
        """[n, HW] log-depth maps: every image from its owner rank (ranks that do not hold an image keep its initial value)."""
        if not self.active:
            return depth
        own = torch.tensor([1.0 if o == self.rank else 0.0 for o in self.owner], dtype=depth.dtype, device=depth.device).unsqueeze(1)
        return self._all_reduce(depth * own)

    def bytes_per_iteration(self, HW, n_small):
        """Bytes THIS rank sends per iteration: the small all-reduce's payload + (halo) its rows for every peer, or
        (allreduce) the rows of every shared image of the clip."""
        if self.exchange == "halo":
            return 4 * (1 + n_small) + 4 * HW * sum(len(v) for v in self.peer_images.values())
        return 4 * (1 + n_small + len(self.shared) * HW)
