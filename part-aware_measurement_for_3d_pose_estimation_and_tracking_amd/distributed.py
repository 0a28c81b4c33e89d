"""Multi-GPU: camera views are partitioned over ranks (one process per GPU); the only exchange is ONE all-gather per
frame of the per-view 2D keypoint records (RCCL over xGMI when the backend is 'nccl'; 'gloo' in the CPU tests).

Per-view work (crop, HRNet, decode) is sharded; the cross-view step (association, part-aware filter, DLT, tracker) is
replicated: after the gather every rank runs the identical deterministic frame kernel, so no broadcast of results is
needed.  The reference has no distributed code at all; this is the build's design."""
import torch
import torch.distributed as dist

NUM_JOINTS = 17


def view_partition(n_views, world):
    """rank r owns the contiguous block {v : v * world // n_views == r} (may be empty when world > n_views)."""
    return [[v for v in range(n_views) if v * world // n_views == r] for r in range(world)]


class ViewGather(object):
    """Fixed-size padded record per view: [n_det, det[max_dets*17*3]] float64 (float64 keeps caller-supplied keypoints
    lossless; records decoded from float32 heat-maps are exactly representable)."""

    def __init__(self, n_views, max_dets, world, rank, device, group=None):
        self.C, self.max_dets, self.world, self.rank, self.group = n_views, max_dets, world, rank, group
        self.parts = view_partition(n_views, world)
        self.mine = self.parts[rank]
        self.maxv = max(1, max(len(p) for p in self.parts))
        self.rec = 1 + max_dets * NUM_JOINTS * 3
        self.send = torch.zeros((self.maxv, self.rec), dtype=torch.float64, device=device)
        self.recv = torch.zeros((world * self.maxv, self.rec), dtype=torch.float64, device=device)
        rows = [0] * n_views
        for r, p in enumerate(self.parts):
            for i, v in enumerate(p):
                rows[v] = r * self.maxv + i
        self.rows = torch.tensor(rows, dtype=torch.long, device=device)
        self.n_det = torch.zeros(n_views, dtype=torch.int32, device=device)
        self.det = torch.zeros((n_views, max_dets, NUM_JOINTS, 3), dtype=torch.float64, device=device)

    def gather(self, n_det_local, det_local):
        """n_det_local (len(mine),) int, det_local (len(mine), max_dets, 17, 3) float64 on this rank's device ->
        (n_det (C,) int32, det (C, max_dets, 17, 3) float64), identical on every rank."""
        k = len(self.mine)
        if k:
            self.send[:k, 0] = n_det_local.to(torch.float64)
            self.send[:k, 1:] = det_local.reshape(k, -1)
        if self.world > 1:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            full = self.recv.index_select(0, self.rows)
        else:
            full = self.send[:self.C]
        self.n_det.copy_(full[:, 0].to(torch.int32))
        self.det.copy_(full[:, 1:].reshape(self.C, self.max_dets, NUM_JOINTS, 3))
        return self.n_det, self.det
