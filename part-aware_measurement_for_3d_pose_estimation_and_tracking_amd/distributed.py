"""Multi-GPU: the per-view / per-crop work is partitioned over ranks (one process per GPU); the only exchange is ONE
all-gather per frame of the 2D keypoint records (RCCL over xGMI when the backend is 'nccl'; 'gloo' in the CPU tests).

Two partitions: ``ViewGather`` -- whole camera views per rank (the natural split when every rank ingests its own cameras;
SURVEY 8e), and ``CropGather`` -- the frame's person crops dealt out evenly regardless of view (HRNet time is proportional
to crops, and 5 views never divide evenly over 2 / 4 / 8 ranks); ``FramePipeline(shard='crops')`` uses the latter.

Per-view work (crop, HRNet, decode) is sharded; the cross-view step (association, part-aware filter, DLT, tracker) is
replicated: after the gather every rank runs the identical deterministic frame kernel, so no broadcast of results is
needed.  The reference has no distributed code at all; this is the build's design."""
import torch
import torch.distributed as dist

NUM_JOINTS = 17


def device_identity(device):
    """(host name, device identity) of a torch CUDA device: what tells whether two ranks drive the same GPU.  The identity is the device's
    UUID where PyTorch reports one, else its PCI location."""
    import socket
    p = torch.cuda.get_device_properties(device)
    # everything that can tell two devices apart, together: a UUID that a runtime reports as zeros for every device, or PCI fields a
    # build does not expose, must not merge two GPUs into one (that would only switch the flags off -- safe, but silent); with none of
    # them available the visible index decides, which calls per-rank HIP_VISIBLE_DEVICES "shared": the safe side
    ident = (str(getattr(p, 'uuid', None)), getattr(p, 'pci_domain_id', None), getattr(p, 'pci_bus_id', None), getattr(p, 'pci_device_id', None))
    if all(x in (None, 'None', '') for x in ident) or (ident[1] is None and set(ident[0]) <= set('0-')):
        ident = ident + (p.name, torch.device(device).index)
    return socket.gethostname(), str(ident)


_share_cache = {}


def ranks_share_a_device(device, group=None, identity=None):
    """True when two ranks of the (initialised) process group drive the same physical GPU -- decided from the devices' identities, not from
    counts: a launcher that shows every rank ONE device (HIP_VISIBLE_DEVICES per rank) has device_count() == 1 on an 8-GPU node, and
    world > device_count() called that sharing.  Collective (all_gather_object), cached per (group, device).  identity: override of this
    rank's (host, device) pair (CPU tests)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return False
    key = (id(group), str(device))
    if identity is None and key in _share_cache:
        return _share_cache[key]
    me = identity if identity is not None else device_identity(device)
    everyone = [None] * dist.get_world_size(group)
    dist.all_gather_object(everyone, tuple(me), group=group)
    shared = len(set(everyone)) < len(everyone)
    if identity is None:
        _share_cache[key] = shared
    return shared


def view_partition(n_views, world):
    """rank r owns the contiguous block {v : v * world // n_views == r} (may be empty when world > n_views)."""
    return [[v for v in range(n_views) if v * world // n_views == r] for r in range(world)]


class AbiComm(object):
    """RCCL communicator owned through the C ABI (pam_comm_*): what a host without PyTorch would hold.  The 128-byte unique id is
    made on rank 0 and shipped with whatever the host has -- here a torch.distributed broadcast (any backend) when world > 1."""

    def __init__(self, world, rank, device_index, group=None):
        import ctypes as C
        from . import _lib
        self.lib = _lib.load()
        idb = (C.c_char * 128)()
        if rank == 0:
            rc = self.lib.pam_comm_unique_id(C.cast(idb, C.c_void_p))
            if rc != 0:
                raise _lib.PamError('pam_comm_unique_id failed (%d): %s' % (rc, self.lib.pam_comm_last_error().decode()))
        if world > 1:
            box = [bytes(idb.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            idb.raw = box[0]
        self.comm = C.c_void_p()
        rc = self.lib.pam_comm_init(C.byref(self.comm), world, rank, C.cast(idb, C.c_void_p), device_index)
        if rc != 0:
            raise _lib.PamError('pam_comm_init failed (%d): %s' % (rc, self.lib.pam_comm_last_error().decode()))

    def close(self):
        if self.comm:
            self.lib.pam_comm_destroy(self.comm)
            self.comm = None


class ViewGather(object):
    """Fixed-size padded record per view, float64 (keeps caller-supplied keypoints lossless; records decoded from float32 heat-maps
    are exactly representable): (max_dets + 1) rows of 17 x 3 -- the view's detection rows (y, x, score), then one row whose first
    element is the detection count.  ``send`` IS the decode target of this rank's views (``det_local`` is a view of it: the head +
    arg-max kernel writes row (view, slot) with a slot stride of max_dets + 1), the all-gather fills ``recv``, and the frame kernel
    reads ``recv`` in place through ``rows`` (pam_frame_dev_views): per frame the exchange is one count copy + the collective."""

    def __init__(self, n_views, max_dets, world, rank, device, group=None, abi=None):
        """abi: (Handle, AbiComm) -> the exchange goes through pam_allgather_keypoints (RCCL called from the library, enqueued on the
        current stream) instead of torch.distributed."""
        self.C, self.max_dets, self.world, self.rank, self.group = n_views, max_dets, world, rank, group
        self.abi = abi
        self.parts = view_partition(n_views, world)
        self.mine = self.parts[rank]
        self.maxv = max(1, max(len(p) for p in self.parts))
        self.send = torch.zeros((self.maxv, max_dets + 1, NUM_JOINTS, 3), dtype=torch.float64, device=device)
        # one rank without a communicator: nothing to exchange, the frame kernel reads the send buffer itself
        self.recv = torch.zeros((world * self.maxv, max_dets + 1, NUM_JOINTS, 3), dtype=torch.float64, device=device) \
            if (world > 1 or abi is not None) else self.send
        self.det_local = self.send[:max(1, len(self.mine))]            # contiguous: the leading records
        self.count_local = self.send[:, max_dets, 0, 0]               # strided view: one count per record
        self.void_local = self.send[:, max_dets, 0, 1]                # ... and the producer's "these keypoints are void" word beside it
        rows = [0] * n_views
        for r, p in enumerate(self.parts):
            for i, v in enumerate(p):
                rows[v] = r * self.maxv + i
        self.rows = torch.tensor(rows, dtype=torch.int32, device=device)
        self.rows_long = self.rows.long()

    def exchange(self, n_det_local, void_word=None):
        """n_det_local (len(mine),) int tensor; the detection rows are already in ``send`` (det_local).  Returns ``recv`` -- every
        rank's records, rank-major -- identical on every rank; read it through ``rows``.
        void_word: this rank's (1,) int32 device word that is non-zero when the forward that decoded these rows gave up
        (HRNetPose.void_word): it travels in the second double of every record's count row, and the frame kernel of EVERY rank skips the
        frame when any record carries it (csrc/pam_tracker.hip) -- the replicated trackers stay identical.  With one rank the frame
        kernel reads the word itself (pam_set_input_guard)."""
        k = len(self.mine)
        if k:
            self.count_local[:k].copy_(n_det_local)                    # int -> float64 in the copy: one small kernel
            if void_word is not None and (self.world > 1 or self.abi is not None):
                self.void_local[:k].copy_(void_word.expand(k))
        if self.world == 1 and self.abi is None:
            return self.recv
        if self.abi is not None:
            import ctypes as C
            handle, comm = self.abi
            rc = handle.lib.pam_allgather_keypoints(handle.raw, comm.comm, C.c_void_p(torch.cuda.current_stream(self.send.device).cuda_stream),
                                                    C.c_void_p(self.send.data_ptr()), self.maxv, C.c_void_p(self.recv.data_ptr()))
            if rc != 0:
                raise RuntimeError('pam_allgather_keypoints failed (%d): %s' % (rc, handle.lib.pam_last_error(handle.raw).decode()))
        else:
            dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1), group=self.group)
        return self.recv

    def gather(self, n_det_local, det_local):
        """Host-visible form (tests, CPU): n_det_local (len(mine),) int, det_local (len(mine), max_dets, 17, 3) float64 ->
        (n_det (C,) int32, det (C, max_dets, 17, 3) float64), identical on every rank.  The product path does not unpack: the frame
        kernel reads ``recv`` through ``rows``."""
        k = len(self.mine)
        if k and det_local.data_ptr() != self.send.data_ptr():
            self.send[:k, :self.max_dets].copy_(det_local.reshape(k, self.max_dets, NUM_JOINTS, 3))
        full = self.exchange(n_det_local).index_select(0, self.rows_long)
        return full[:, self.max_dets, 0, 0].to(torch.int32), full[:, :self.max_dets].contiguous()


def crop_partition(n_crops, world):
    """rank r owns the contiguous crop range [r*n // world, (r+1)*n // world) (sizes differ by at most one)."""
    return [(r * n_crops // world, (r + 1) * n_crops // world) for r in range(world)]


class CropGather(object):
    """Every rank decodes ITS crops into a full-shape (C, max_dets, 17, 3) float64 buffer (``self.send``; rows of other
    ranks' crops are don't-care), the buffers are all-gathered, and row (view, slot) is taken from the rank that owns
    that crop (``select_index``).  Identical result on every rank."""

    def __init__(self, n_views, max_dets, world, rank, device, group=None):
        self.C, self.max_dets, self.world, self.rank, self.group = n_views, max_dets, world, rank, group
        self.rows = n_views * max_dets
        # one spare row behind the keypoint rows: its first double carries the rank's void word through the exchange
        self._send_all = torch.zeros((self.rows + 1, NUM_JOINTS * 3), dtype=torch.float64, device=device)
        self.send = self._send_all[:self.rows].view(n_views, max_dets, NUM_JOINTS, 3)
        self.recv = torch.zeros((world, self.rows + 1, NUM_JOINTS * 3), dtype=torch.float64, device=device)
        self.det = torch.zeros((n_views, max_dets, NUM_JOINTS, 3), dtype=torch.float64, device=device)
        self.void_any = torch.zeros(1, dtype=torch.int32, device=device)     # after gather(): some rank's forward of this frame gave up

    @staticmethod
    def select_index(view_of, slot_of, n_views, max_dets, world):
        """Host helper: the frame's crop list (view, slot per crop; the order defines the partition) -> (select, parts):
        select[(view, slot)] = row of the all-gathered buffer that holds that crop (int64, C*max_dets), parts = per-rank
        crop ranges.  Empty slots point at rank 0's (unused) row."""
        import numpy as np
        parts = crop_partition(len(view_of), world)
        rows = n_views * max_dets
        owner = np.zeros(rows, dtype=np.int64)
        for r, (a, b) in enumerate(parts):
            for i in range(a, b):
                owner[int(view_of[i]) * max_dets + int(slot_of[i])] = r
        return owner * (rows + 1) + np.arange(rows, dtype=np.int64), parts

    def gather(self, select, void_word=None):
        """select: (C*max_dets,) int64 device tensor from select_index -> det (C, max_dets, 17, 3) float64, all ranks alike.
        void_word: this rank's (1,) int32 "my forward gave up" word (HRNetPose.void_word); ``void_any`` is then 1 on EVERY rank when any
        rank's word was up -- the word the frame kernel is guarded by in this mode (FramePipeline), so the replicated trackers skip the
        same frames."""
        if self.world > 1:
            if void_word is not None:
                self._send_all[self.rows, :1].copy_(void_word)
            dist.all_gather_into_tensor(self.recv.view(-1), self._send_all.view(-1), group=self.group)
            torch.index_select(self.recv.view(self.world * (self.rows + 1), -1), 0, select, out=self.det.view(self.rows, -1))
            if void_word is not None:
                self.void_any.copy_(self.recv[:, self.rows, 0].amax().reshape(1))
            return self.det
        return self.send


def check_same_call(n_crops, n_views, device, group=None):
    """Debug guard of the crop-sharded predict(): raises unless every rank was called with the same number of crops and views."""
    t = torch.tensor([n_crops, -n_crops, n_views, -n_views], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    lo_n, hi_n, lo_v, hi_v = -int(t[1]), int(t[0]), -int(t[3]), int(t[2])
    if lo_n != hi_n or lo_v != hi_v:
        raise RuntimeError('crop-sharded predict() is a collective: ranks passed %d..%d crops in %d..%d views' % (lo_n, hi_n, lo_v, hi_v))


def gather_crop_keypoints(kp_local, n_total, world, rank, group=None):
    """The drop-in surface's exchange (HRNetPose.predict under torch.distributed): the call's n_total person crops are dealt out with
    ``crop_partition``; every rank decodes its share -> kp_local (b - a, 17, 3) float32 (x, y, score) -> ONE all-gather ->
    (n_total, 17, 3), identical on every rank.  Works on CPU tensors (gloo) and CUDA tensors (RCCL)."""
    parts = crop_partition(n_total, world)
    if world == 1:
        return kp_local
    maxn = max(1, max(b - a for a, b in parts))
    send = torch.zeros((maxn, NUM_JOINTS, 3), dtype=kp_local.dtype, device=kp_local.device)
    a, b = parts[rank]
    if b > a:
        send[:b - a] = kp_local
    recv = torch.empty((world * maxn, NUM_JOINTS, 3), dtype=kp_local.dtype, device=kp_local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    idx = torch.tensor([r * maxn + i for r, (p, q) in enumerate(parts) for i in range(q - p)], dtype=torch.long, device=kp_local.device)
    return recv.index_select(0, idx)
