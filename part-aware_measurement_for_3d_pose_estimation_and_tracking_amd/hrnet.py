"""HRNet-W48 top-down 2D pose (a1): the ``HRNetPose(...).predict(...)`` seam of the reference
(/root/reference/src/ivclabpose.py:125-134,210; the backend itself is git-ignored there).

The conv stack is a plain PyTorch-ROCm module (public HRNet architecture, official state-dict key layout so
``pose_hrnet_w48_384x288.pth`` loads), run in bf16 channels-last with BatchNorm folded and replayed from a hipGraph.
Crop / resize / normalise and the heat-map arg-max decode are HIP kernels of libpam_hip.so (csrc/pam_image.hip).
Parity with the authors' modified backend is UNPINNED (no source, weights or tests in the reference): decode follows
upstream simple-HRNet (hard arg-max, linear map through the box)."""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as Fnn

from . import _lib

BN_EPS = 1e-5


def conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = Fnn.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return Fnn.relu(y + r)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = conv3x3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        y = Fnn.relu(self.bn1(self.conv1(x)))
        y = Fnn.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return Fnn.relu(y + r)


class HighResolutionModule(nn.Module):
    def __init__(self, channels, num_blocks=4, multi_scale_output=True):
        super().__init__()
        nb = len(channels)
        self.branches = nn.ModuleList(
            [nn.Sequential(*[BasicBlock(c, c) for _ in range(num_blocks)]) for c in channels])
        fuse = []
        for i in range(nb if multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(channels[j], channels[i], 1, bias=False),
                                             nn.BatchNorm2d(channels[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    row.append(None)
                else:
                    steps = []
                    for k in range(i - j):
                        if k == i - j - 1:
                            steps.append(nn.Sequential(conv3x3(channels[j], channels[i], 2), nn.BatchNorm2d(channels[i])))
                        else:
                            steps.append(nn.Sequential(conv3x3(channels[j], channels[j], 2), nn.BatchNorm2d(channels[j]),
                                                       nn.ReLU(inplace=False)))
                    row.append(nn.Sequential(*steps))
            fuse.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(fuse)

    def forward(self, xs):
        xs = [b(x) for b, x in zip(self.branches, xs)]
        out = []
        for i, row in enumerate(self.fuse_layers):
            y = None
            for j, f in enumerate(row):
                t = xs[j] if f is None else f(xs[j])
                y = t if y is None else y + t
            out.append(Fnn.relu(y))
        return out


class PoseHighResolutionNet(nn.Module):
    """HRNet-W{c}: stem -> layer1 -> 3 multi-resolution stages -> 1x1 head (Appendix D of SURVEY.md)."""

    def __init__(self, c=48, num_joints=17):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64)
        down = nn.Sequential(nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256))
        self.layer1 = nn.Sequential(Bottleneck(64, 64, downsample=down), Bottleneck(256, 64), Bottleneck(256, 64),
                                    Bottleneck(256, 64))
        ch = [c, 2 * c, 4 * c, 8 * c]
        self.transition1 = nn.ModuleList([
            nn.Sequential(conv3x3(256, ch[0]), nn.BatchNorm2d(ch[0]), nn.ReLU(inplace=False)),
            nn.Sequential(nn.Sequential(conv3x3(256, ch[1], 2), nn.BatchNorm2d(ch[1]), nn.ReLU(inplace=False)))])
        self.stage2 = nn.Sequential(HighResolutionModule(ch[:2]))
        self.transition2 = nn.ModuleList([None, None, nn.Sequential(
            nn.Sequential(conv3x3(ch[1], ch[2], 2), nn.BatchNorm2d(ch[2]), nn.ReLU(inplace=False)))])
        self.stage3 = nn.Sequential(*[HighResolutionModule(ch[:3]) for _ in range(4)])
        self.transition3 = nn.ModuleList([None, None, None, nn.Sequential(
            nn.Sequential(conv3x3(ch[2], ch[3], 2), nn.BatchNorm2d(ch[3]), nn.ReLU(inplace=False)))])
        self.stage4 = nn.Sequential(HighResolutionModule(ch), HighResolutionModule(ch),
                                    HighResolutionModule(ch, multi_scale_output=False))
        self.final_layer = nn.Conv2d(ch[0], num_joints, 1)

    def features(self, x):
        x = Fnn.relu(self.bn1(self.conv1(x)))
        x = Fnn.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        xs = [self.transition1[0](x), self.transition1[1](x)]
        xs = self.stage2[0](xs)
        xs = xs + [self.transition2[2](xs[-1])]
        for m in self.stage3:
            xs = m(xs)
        xs = xs + [self.transition3[3](xs[-1])]
        for m in self.stage4:
            xs = m(xs)
        return xs[0]

    def forward(self, x):
        return self.final_layer(self.features(x))


def init_random(model, seed=0):
    """Seeded He-normal convs, BN gamma=1 beta=0 mean=0 var=1 (real weights are unavailable offline)."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1]
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_out) ** 0.5)
                if m.bias is not None:
                    m.bias.zero_()
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
            m.running_mean.zero_(); m.running_var.fill_(1.0)
    for m in model.modules():          # damp the residual branches so activations stay O(1) through ~80 blocks
        if isinstance(m, BasicBlock):
            nn.init.constant_(m.bn2.weight, 0.3)
        elif isinstance(m, Bottleneck):
            nn.init.constant_(m.bn3.weight, 0.3)
    return model


def count_flops(c=48, num_joints=17, h=384, w=288):
    """2*MAC of every conv for one crop, counted from the module table by a shape-only forward on the meta device."""
    model = PoseHighResolutionNet(c, num_joints).to('meta')
    total = [0]

    def hook(m, inp, out):
        total[0] += 2 * out.shape[1] * out.shape[2] * out.shape[3] * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
    hs = [m.register_forward_hook(hook) for m in model.modules() if isinstance(m, nn.Conv2d)]
    with torch.no_grad():
        model(torch.empty(1, 3, h, w, device='meta'))
    for x in hs:
        x.remove()
    return total[0]


def algorithmic_work(n_crops, resolution=(384, 288), config=None):
    """Algorithmic work of one conv-stack forward (shape-only walks on the meta device):
    ``bytes``: EXECUTOR-INDEPENDENT unique HBM bytes of the module graph -- every BasicBlock / Bottleneck counted as input + its weights and
    biases + output (block interiors are not traffic the algorithm needs: a fused launch keeps them on chip), every other convolution as
    input + weights + bias + output, every fuse-layer sum as base + terms + output.  Un-fusing a block cannot raise this number.
    ``flops``; ``launches`` and ``bytes_as_executed`` (per launch: in + weights + bias [+ residual] + out) of the given executor
    configuration (HipHRNet.CONFIGS; None = the default)."""
    from .hrnet_hip import HipHRNet

    from . import _lib as _real

    class _MetaLib(object):
        def __getattr__(self, name):
            if name in ('pam_conv3x3_slab', 'pam_conv3x3_layout', 'pam_conv3x3_layout_ex'):       # pure host-side shape queries: the executor's plan depends on them
                return getattr(_real.load(), name)
            return lambda *a, **k: 0
    eng = HipHRNet.__new__(HipHRNet)
    eng.lib = _MetaLib(); eng.device = torch.device('meta'); eng.tile_cfg = -1; eng.multi_stream = False
    model = fold_batchnorm(PoseHighResolutionNet())
    model.final_layer = nn.Identity()
    HipHRNet._pack(eng, model, torch.device('meta'))
    eng.apply_config(config or HipHRNet.config_name)
    eng.count = dict(bytes=0, flops=0, launches=0)
    x = torch.empty((n_crops, 8, resolution[0], resolution[1]), dtype=torch.bfloat16, device='meta').contiguous(memory_format=torch.channels_last)
    eng._features(x)
    # the module graph itself: hooks on the blocks (whole), on the convolutions outside blocks, and the fuse sums from the module's shapes
    total = [0]
    inside = set()
    for m in model.modules():
        if isinstance(m, (BasicBlock, Bottleneck)):
            inside.update(id(c) for c in m.modules() if c is not m)
    pbytes = lambda m: sum(2 * p.numel() if p.dim() > 1 else 4 * p.numel() for p in m.parameters())

    def hook(m, inp, out):
        total[0] += 2 * (inp[0].numel() + out.numel()) + pbytes(m)
    hs = [m.register_forward_hook(hook) for m in model.modules()
          if isinstance(m, (BasicBlock, Bottleneck)) or (isinstance(m, nn.Conv2d) and id(m) not in inside)]

    def sum_hook(m, inp, out):                            # HighResolutionModule: out_i = relu(x_i + terms): base + each term at ITS resolution + out
        for i, o in enumerate(out):
            total[0] += 2 * 2 * o.numel()
            for j in range(len(out)):
                if j > i:                                 # 1x1 up-convolution's output, read through the nearest-neighbour upsample
                    total[0] += 2 * o.numel() // (4 ** (j - i))
                elif j < i:
                    total[0] += 2 * o.numel()
    hs += [m.register_forward_hook(sum_hook) for m in model.modules() if isinstance(m, HighResolutionModule)]
    with torch.no_grad():
        model.to('meta')(torch.empty((n_crops, 3, resolution[0], resolution[1]), device='meta'))
    for h in hs:
        h.remove()
    return dict(bytes=total[0], flops=eng.count['flops'], launches=eng.count['launches'], bytes_as_executed=eng.count['bytes'])


def fold_batchnorm(model):
    """Inference form: every (conv, BN) pair becomes one conv with bias (scale = gamma / sqrt(var + eps))."""
    def fold(conv, bn):
        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        new = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True)
        with torch.no_grad():
            new.weight.copy_(conv.weight * s.reshape(-1, 1, 1, 1))
            new.bias.copy_(bn.bias - bn.running_mean * s + (conv.bias * s if conv.bias is not None else 0))
        return new

    def walk(mod):
        names = list(mod._modules.keys())
        for a, b in zip(names, names[1:]):
            ma, mb = mod._modules[a], mod._modules[b]
            if isinstance(ma, nn.Conv2d) and isinstance(mb, nn.BatchNorm2d):
                mod._modules[a] = fold(ma, mb)
                mod._modules[b] = nn.Identity()
        for m in mod._modules.values():
            if m is not None:
                walk(m)
    model = model.eval()
    walk(model)
    return model


_FOLDED_RANDOM = {}


def _folded_random_model(c, nof_joints, seed):
    """The seeded random-weight network, BN folded: built once per (c, joints, seed) and process (2.7 s of CPU work), copied per object --
    a test session or a bench run constructs dozens of networks over the same weights."""
    import copy
    key = (int(c), int(nof_joints), int(seed))
    if key not in _FOLDED_RANDOM:
        _FOLDED_RANDOM[key] = fold_batchnorm(init_random(PoseHighResolutionNet(c, nof_joints), seed))
    return copy.deepcopy(_FOLDED_RANDOM[key])


class HRNetPose(object):
    """Mirror of ``backend.HRPose.SimpleHRNet.HRNetPose``: ctor (c, nof_joints, checkpoint, model_name, resolution, ...),
    ``predict(person_bbox_list, batch_size, conf_threshold) -> dump_results`` (ivclabpose.py:131-132,210)."""

    def __init__(self, c, nof_joints, checkpoint_path, model_name='HRNet', resolution=(384, 288), hrpose_args=None,
                 device=0, dtype=torch.bfloat16, use_graph=True, seed=0, max_dets=16, backend='hip', graph_bucket=4,
                 shard_crops=False, group=None, autotune=False, max_crops=32, antialias=False):
        assert model_name == 'HRNet' and int(nof_joints) == 17
        if not torch.cuda.is_available():
            raise RuntimeError('HRNetPose needs a GPU (the preprocessing / decode kernels are HIP only; no CPU fallback)')
        self.lib = _lib.load()
        # hrpose_args: the reference hands HRNetPose every visible GPU (gpu_args.gpus / .device, ivclabpose.py:107-111,131-132) and
        # lets the backend spread a batch over them inside one process.  Here the unit is one process per GPU.  shard_crops=True
        # (what ivclabpose passes when it runs under a torch.distributed job) makes predict() a COLLECTIVE over `group`: every rank
        # must call it with the SAME person_bbox_list; the call's crops are dealt out over the ranks, each runs its share on ITS
        # device (hrpose_args.device if given, else `device`) and ONE all-gather returns every rank the complete dump.  The default
        # is off: a merely initialised process group (a view-sharded host that calls predict() with per-rank inputs) changes nothing.
        import torch.distributed as dist
        self.group = group
        self.world, self.rank = (1, 0)
        if shard_crops and dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        dv = getattr(hrpose_args, 'device', None) if hrpose_args is not None else None
        if dv is None and isinstance(hrpose_args, dict):
            dv = hrpose_args.get('device')
        if dv is not None and torch.device(dv).type == 'cuda' and torch.device(dv).index is not None:
            device = torch.device(dv).index
        self.device = torch.device('cuda:%d' % device)
        self.resolution = tuple(resolution)
        self.dtype = dtype
        self.max_dets = max_dets
        if checkpoint_path and os.path.exists(checkpoint_path):
            model = PoseHighResolutionNet(c, nof_joints)
            sd = torch.load(checkpoint_path, map_location='cpu')
            model.load_state_dict(sd.get('model', sd) if isinstance(sd, dict) else sd)
            self.weights = checkpoint_path
            model = fold_batchnorm(model)
        else:
            model = _folded_random_model(c, nof_joints, seed)
            self.weights = 'random(seed=%d)' % seed
        self.head = model.final_layer.to(self.device).float()           # 1x1 head + decode stay float32
        self.head_w = self.head.weight.detach().reshape(int(nof_joints), -1).contiguous()     # [17][48] for k_head
        self.head_b = self.head.bias.detach().contiguous()
        model.final_layer = nn.Identity()
        # the conv stack = the hand-written MFMA kernels of csrc/ (hrnet_hip.HipHRNet); there is no other backend in the product: the
        # PyTorch-ROCm form of the same folded module that the tests compare against lives in tests/torch_ref.py
        if backend != 'hip':
            raise ValueError("HRNetPose has one conv backend, 'hip' (got %r)" % (backend,))
        self.backend = backend
        from .hrnet_hip import HipHRNet
        self.hip = HipHRNet(model, self.device)
        self.in_channels = 8
        self.model = None
        self.use_graph = use_graph
        # crop resize: False = plain bilinear (cv2.resize INTER_LINEAR semantics; the default: exact for boxes up to the network input, and what
        # every golden of this repository was made with); True = upstream simple-HRNet's PIL / torchvision Resize, which filters with a triangle
        # that widens with the down-scaling factor (HD Panoptic boxes taller than 384 pixels) -- csrc/pam_image.hip
        self.antialias = bool(antialias)
        self.flag_synced = {}        # (crops, kind, slot) -> the replay orders its branch streams by device-side flags (False: stream events)
        # a gate that times out in ANY replay stores 1 here (pinned host memory, csrc/pam_sync.hip): read before every replay
        self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._flag_host_np = self._flag_host.numpy()
        self.hip.flag_host_err = self._flag_host
        # ... and 1 in this DEVICE word, which the frame kernel reads in front of every frame (pam_set_input_guard): keypoints decoded from a
        # forward whose gate gave up never reach the tracker state, whoever consumes them and however far the host has run ahead.  The
        # consumers recover (DumpResults / ivclabpose: the forward is re-run with stream events; FramePipeline.results raises FrameVoid
        # with the frame to resume from) -- check_void / clear_void below.
        self.void_word = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.hip.flag_dev_void = self.void_word
        self.hip.flag_limit()                             # (the gates' bound as a device word: made here, never inside a capture)
        self.void_pending = False    # a time-out has been seen and the forwards since have not been re-run yet
        self.flag_timeouts = 0       # time-outs seen by this object after the capture-time checks
        if self.world > 1:
            # the crop-sharded predict() is a collective that one rank cannot re-run alone: stream events.  (Ranks that SHARE a device
            # oversubscribe its hardware queues -- a gate can sit in front of its own producer: FramePipeline decides that from the
            # devices' identities, distributed.ranks_share_a_device, and calls disable_flag_sync; no collective hides in this constructor.)
            self._flag_sync_failed = True
        self.flag_timing = {}        # crop count -> the race of its first flagged capture against stream events: dict(ms={mode: (flags, events)}, kept={mode: bool})
        self._alt = {}               # (crops, kind, slot) -> {mode: (graph, static_in, static_out)} when both forms exist
        self.flag_race = 'serial'    # which timing of that race picks a replay's form: 'serial' (one replay at a time: predict()'s use), 'throughput' (back to back: FramePipeline); None = no race, flags stay
        self._dead_graphs = []       # captures that lost that race or failed the flag check (never destroyed: _lib.new_graph)
        self.captures = 0            # hipGraph captures made so far (a capture inside a frame is a stall of hundreds of ms: warm())
        # predict() pads a batch to the next multiple of graph_bucket crops (repeating its last box; the padded rows are not
        # decoded): a sequence whose person count wanders then replays a handful of captured graphs instead of capturing one
        # per count (a capture is a multi-100-ms stall); 1 = exact batch sizes
        self.graph_bucket = max(1, int(graph_bucket)) if use_graph else 1
        self._graphs = {}
        # autotune: the executor configuration follows the crop count (config_for); off = one configuration for every count (results of
        # different configurations differ in the last bf16 bits, and a test that compares an eager forward with a replay needs both in ONE)
        self.autotune = bool(autotune) and use_graph
        self.tuned = {}                                   # crop count -> {'choice': configuration name} of every replay captured so far
        self._kp_pinned = {}         # crop count -> two pinned host buffers for predict()'s keypoints
        self._meta_pinned = {}       # table size -> two pinned staging buffers for predict()'s per-call tables
        # activations of the captured forwards: one arena per replay slot, sized for max_crops crops per forward (a larger forward gets a
        # new, larger arena; the captures made before keep theirs) -- hrnet_hip.ActivationArena
        self.max_crops = int(max_crops)
        self._arenas = {}
        self._arena_bytes_per_crop = None
        self._pools = {}             # graph memory pool per replay slot: graphs of ONE slot replay one after the other and may share
                                     # intermediates; the two slots of FramePipeline(pose_streams=2) replay concurrently and must not
        self.stream = torch.cuda.current_stream(self.device)

    # -- conv stack (PyTorch-ROCm; hipGraph replay per batch size) -------------------------------------------------
    def _forward(self, x, kind='heatmaps'):
        f = self.hip.features(x)                                        # (N, 48, h, w) channels-last bf16
        if kind == 'features':
            return f
        n, c, h, w = f.shape
        hm = torch.empty((n, self.head_w.shape[0], h, w), dtype=torch.float32, device=f.device, memory_format=torch.channels_last)
        rc = self.lib.pam_head_heatmaps(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), n * h * w,
                                        C.c_void_p(f.data_ptr()), c, C.c_void_p(self.head_w.data_ptr()),
                                        C.c_void_p(self.head_b.data_ptr()), self.head_w.shape[0], C.c_void_p(hm.data_ptr()))
        if rc != 0:
            raise _lib.PamError('pam_head_heatmaps failed: %d' % rc)
        return hm

    def heatmaps(self, x, slot=0):
        """x: (N,3,H,W) channels-last bf16 on the device -> (N,17,H/4,W/4) float32 (channels-last memory)."""
        return self._run(x, 'heatmaps', slot)

    def features(self, x, slot=0):
        """x as above -> (N,48,H/4,W/4) channels-last bf16: the input of ``head_decode`` (the product path: the heat-maps are
        never written).  slot: which replay instance (own static input / activations / output) -- two frames whose forwards are in
        flight at the same time (FramePipeline(pose_streams=2)) use different slots of the same weights."""
        return self._run(x, 'features', slot)

    def _head(self, f):
        n, c, h, w = f.shape
        hm = torch.empty((n, self.head_w.shape[0], h, w), dtype=torch.float32, device=f.device, memory_format=torch.channels_last)
        rc = self.lib.pam_head_heatmaps(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), n * h * w,
                                        C.c_void_p(f.data_ptr()), c, C.c_void_p(self.head_w.data_ptr()),
                                        C.c_void_p(self.head_b.data_ptr()), self.head_w.shape[0], C.c_void_p(hm.data_ptr()))
        if rc != 0:
            raise _lib.PamError('pam_head_heatmaps failed: %d' % rc)
        return hm

    def _run(self, x, kind, slot=0):
        n = x.shape[0]
        assert self.flag_race in (None, 'serial', 'throughput'), self.flag_race
        if not self.use_graph:
            with torch.no_grad():
                return self._forward(x, kind)
        self.check_void()                                  # a time-out of an EARLIER replay: stream events from here on (before any capture)
        g = self._graphs.get((n, kind, slot))
        if g is None:
            self.hip.apply_config(self.config_for(n))
            self.hip.arena = self._arena_for(n, slot)
            other = self._graphs.get((n, 'features' if kind == 'heatmaps' else 'heatmaps', slot))
            static_in = other[1] if other is not None else torch.empty_like(x)     # one input buffer per batch size and slot
            static_in.copy_(x)
            with torch.no_grad():
                s = torch.cuda.Stream(self.device)
                s.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(s):
                    for _ in range(2):
                        self._forward(static_in, kind)
                torch.cuda.current_stream(self.device).wait_stream(s)
                raced = self.flag_timing.get(n)
                want_flags = self._flag_sync_ok() and (raced is None or raced['kept']['serial'] or raced['kept']['throughput'])
                graph, static_out, flags = self._capture(static_in, kind, slot, want_flags)
                alt = None
                if flags is not None:
                    # the flagged replay once, then its error word: a gate that timed out (two branch chains on one in-order hardware queue,
                    # a profiler that serialises kernels) -> this object goes back to stream events, for this and every later capture
                    # (the flagged capture stays alive: captured graphs are never destroyed, _lib.new_graph)
                    graph.replay()
                    torch.cuda.synchronize(self.device)
                    ok = int(flags[0].item()) == 0
                    self._flag_host_np[0] = 0                                             # (a time-out of THIS replay is dealt with here:
                    if not self.void_pending:                                             #  its output is discarded)
                        self.void_word.zero_()
                    if not ok:
                        self._flag_sync_failed = True
                        self._dead_graphs.append((graph, static_out, flags))
                        graph, static_out, flags = self._capture(static_in, kind, slot, False)
                    elif self.flag_race:
                        # every flagged capture has the same forward ordered by stream events beside it, and both are timed both ways: 'throughput'
                        # = replays back to back (FramePipeline: the host runs a frame ahead; flags win by 1-6 %), 'serial' = one replay at a
                        # time (the synchronous drop-in surface: the runtime enqueues a flagged graph chain by chain, so a new branch's first
                        # kernel is enqueued ~200 us after the caller stream's, which only back-to-back replays hide -- small forwards lose
                        # more to that than the joins cost).  A replay uses the form that is faster in the object's CURRENT mode (flag_race).
                        # Under rocprofv3's kernel tracing the gates wait ~1 ms each without ever timing out: flags lose every race.
                        ev_graph, ev_out, _ = self._capture(static_in, kind, slot, False)
                        if raced is None:
                            # interleaved rounds, the median of each form; flags are kept only where they win by more than flag_margin
                            # (round 5 decided from 4 replays with a 1.7 % margin: not reproducible from box to box)
                            t = {m: self._race(graph, ev_graph, m) for m in ('serial', 'throughput')}
                            raced = self.flag_timing[n] = dict(ms=t, kept={m: t[m][0] <= (1.0 - self.flag_margin) * t[m][1] for m in t})
                        alt = {'flags': (graph, static_in, static_out), 'events': (ev_graph, static_in, ev_out)}
                self.captures += 1
            self.hip.arena = None
            g = (graph, static_in, static_out)
            self._graphs[(n, kind, slot)] = g
            self._alt[(n, kind, slot)] = alt
            self.flag_synced[(n, kind, slot)] = flags is not None if alt is None else bool(raced['kept'][self.flag_race])
        if self.check_void():                              # (a replay of the capture-time race gave up: the switch to stream events has
            g = self._graphs.get((n, kind, slot))          #  replaced or dropped this forward's flagged capture)
            if g is None:
                return self._run(x, kind, slot)
        alt = self._alt.get((n, kind, slot))
        if alt is not None:                                # two forms of this forward: the one that is faster the way this object is used now
            g = alt['flags' if self.flag_timing[n]['kept'][self.flag_race or 'throughput'] else 'events']
        graph, static_in, static_out = g
        if static_in.data_ptr() != x.data_ptr():
            static_in.copy_(x)
        graph.replay()
        return static_out

    flag_margin = float(os.environ.get('PAM_FLAG_MARGIN', '0.01'))    # flags must win the capture-time race by this fraction to be kept
    flag_race_rounds = int(os.environ.get('PAM_FLAG_RACE_ROUNDS', '5'))

    def _race(self, graph, ev_graph, mode):
        """(ms flags, ms events) of one forward captured both ways: flag_race_rounds interleaved rounds of 4 replays each (20 replays per
        form), the median round of each."""
        a, b = [], []
        for _ in range(max(1, self.flag_race_rounds)):
            a.append(self._replay_ms(graph, mode)); b.append(self._replay_ms(ev_graph, mode))
        a.sort(); b.sort()
        return a[len(a) // 2], b[len(b) // 2]

    def _replay_ms(self, graph, mode='serial', reps=4):
        """ms per replay of a captured forward (events on the caller's stream): 'serial' = the shortest of `reps` replays issued one at a
        time, 'throughput' = `reps` replays issued back to back."""
        if mode == 'throughput':
            graph.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                graph.replay()
            e1.record(); e1.synchronize()
            return e0.elapsed_time(e1) / reps
        best = float('inf')
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); graph.replay(); e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    def disable_flag_sync(self):
        """From now on every replay orders its branch streams by stream events; flagged captures leave the cache (kept alive, never
        destroyed) and are re-captured at their next use.  For callers that keep two forwards in flight at the same time: the gates
        of two flagged replays can block each other's hardware queues (FramePipeline(pose_streams=2) calls this)."""
        self._flag_sync_failed = True
        for key in list(self.flag_synced):
            alt = self._alt.pop(key, None)
            if alt is not None:                            # the stream-event form of this forward exists already
                self._dead_graphs.append(alt['flags']); self._graphs[key] = alt['events']; self.flag_synced[key] = False
            elif self.flag_synced[key]:
                self._dead_graphs.append(self._graphs.pop(key))
                self.flag_synced.pop(key)

    def check_void(self):
        """Has a gate of a replay given up since the last clear_void()?  (One read of a pinned host word: free.)  The first sighting
        switches the object to stream events for every later forward; the forwards replayed between the time-out and the switch are
        void -- their consumers on the device skipped them (void_word), and whoever holds their inputs re-runs them after clear_void()."""
        if self._flag_host_np[0] != 0:
            self._flag_host_np[0] = 0
            if not self.void_pending:
                self.flag_timeouts += 1
                import warnings
                warnings.warn('a device-side gate of a captured HRNet forward timed out (flag_max_us = %d us; a second process on this GPU?): '
                              'the affected forwards are re-run and this object orders its branch streams by stream events from now on'
                              % self.hip.flag_max_us, RuntimeWarning)
            self.void_pending = True
            self.disable_flag_sync()
        return self.void_pending

    def clear_void(self):
        """Call with the device's work drained or about to be re-issued: lowers the device word (the frame kernel applies frames again) and
        the pending mark.  Synchronises the device."""
        torch.cuda.synchronize(self.device)
        self._flag_host_np[0] = 0
        self.void_word.zero_()
        torch.cuda.synchronize(self.device)
        self.void_pending = False

    def _flag_sync_ok(self):
        return bool(self.hip.flag_sync) and not getattr(self, '_flag_sync_failed', False) and os.environ.get('PAM_FLAG_SYNC', '1') != '0'

    def _capture(self, static_in, kind, slot, flags_on):
        """Capture one forward on static_in; flags_on: the branch streams are ordered by device-side flags (hrnet_hip.HipHRNet)."""
        self.hip.flags_on = bool(flags_on)
        try:
            graph = _lib.new_graph()
            with torch.cuda.graph(graph, pool=self._pool_of(slot)):
                static_out = self._forward(static_in, kind)
            flags = self.hip._flags if flags_on else None
        finally:
            self.hip.flags_on = False
            self.hip._flags = None
        return graph, static_out, flags

    def _arena_for(self, n, slot):
        """The activation arena the n-crop capture of `slot` allocates from (created / replaced by a larger one on demand)."""
        from .hrnet_hip import ActivationArena
        if self._arena_bytes_per_crop is None:            # largest epoch of a one-crop forward, from a shape-only walk of this executor
            hip, probe = self.hip, ActivationArena()
            saved = (hip.multi_stream, hip.arena, hip.count, hip.prof)
            hip.multi_stream, hip.arena, hip.count, hip.prof = False, probe, None, None
            try:
                H, W = self.resolution
                hip._features(torch.empty((1, self.in_channels, H, W), dtype=self.dtype, device='meta').contiguous(memory_format=torch.channels_last))
            finally:
                hip.multi_stream, hip.arena, hip.count, hip.prof = saved
            self._arena_bytes_per_crop = probe.peak       # per tensor rounded up to 256 B: n crops need <= n times this
        ar = self._arenas.get(slot)
        need = n * self._arena_bytes_per_crop
        if ar is None or ar.half_bytes < need:
            # first arena: twice the first forward's need, at most max_crops crops (a rig of 31 views x 16 detections would otherwise
            # reserve 18 GB per replay slot for frames of 40 crops); warm() captures its largest bucket first and so sizes the arena once.
            # A forward beyond the arena gets a new one of TWICE its need (the captures made so far keep the old one: their kernels hold
            # its addresses, and a captured graph is never destroyed, _lib.new_graph): a crop count that creeps upwards costs at most
            # 4 x the largest forward's need in total
            cap = self.max_crops * self._arena_bytes_per_crop
            ar = ActivationArena(self.device, max(need, min(2 * need, cap)) if ar is None else 2 * need)
            self._arenas[slot] = ar
        return ar

    def bucket(self, n, cap=None):
        """Crop count of the replay that serves a call of n crops: the next multiple of graph_bucket (at most cap)."""
        b = self.graph_bucket
        m = (n + b - 1) // b * b
        return min(m, cap) if cap is not None and cap >= n else m

    def warm(self, max_crops, slots=(0,), kind='features'):
        """Capture the replay of every crop-count bucket up to max_crops NOW (largest first: one activation arena, sized once), so that
        no frame pays a capture later -- the first sight of a bucket inside a frame is a stall of hundreds of milliseconds.
        -> dict(buckets, captures, seconds, arena_bytes): what the prewarmed cache holds."""
        import time
        t0, c0 = time.perf_counter(), self.captures
        buckets = sorted({self.bucket(k) for k in range(1, int(max_crops) + 1)}, reverse=True)
        self.max_crops = max(self.max_crops, buckets[0] if buckets else 0)
        for slot in slots:
            for n in buckets:
                if (n, kind, slot) not in self._graphs:
                    self._run(self.input_buffer(n, slot), kind, slot)
        torch.cuda.synchronize(self.device)
        return dict(buckets=sorted(buckets), captures=self.captures - c0, seconds=time.perf_counter() - t0, arena_bytes=self.arena_bytes())

    def arena_bytes(self):
        """Device memory of the activation arenas of all replay slots."""
        return int(sum(2 * a.half_bytes for a in self._arenas.values()))

    def _pool_of(self, slot):
        if slot not in self._pools:
            self._pools[slot] = torch.cuda.graph_pool_handle()
        return self._pools[slot]

    def config_for(self, n):
        """Executor configuration of the n-crop forward (HipHRNet.CONFIGS).  With ``autotune`` a fixed rule from interleaved A/B runs over
        crop counts with the branch streams ordered by device-side flags (tools/ab_flags.py fused:flags_on=1 res48:flags_on=1,block2=1
        fsum:flags_on=1,fused_sums=1, end of round 5): up to 20 crops the fuse-layer sums carry their 1x1 products (k_fuse_sum, 18 launches
        fewer: 2-6 crops -3.0 ... -3.4 %, 8-16 crops -0.6 ... +0.7 %, 18 crops -1.3 %, 20 crops -1.7 %), above that they are separate launches
        (22 crops 0.0 %, 24 +0.5 %, 28 +1.6 %: the 1x1 launches hide beside the other branches' blocks and the sums are on the critical path).
        Up to 12 crops the 192- / 384-channel 3x3 layers also run with 32-channel slabs (HipHRNet.slab32: a launch of a few crops is as long
        as ONE workgroup; 2 crops -8 %, 4 -10 %, 6 -8 %, 9 -3 ... -6 %, 12 -2 %, 14 0 %, 16 +2 %).
        The 96-channel branch as streamed convolutions (round 4's choice for 8-12 crops under stream events) loses at every count now
        (+0.1 ... +5.6 %).  (Round 3 timed every configuration at the first replay of a crop count: 1.5 s per count, a choice decided by
        noise, and three dead captures per count that could never be destroyed, see _lib.new_graph.)"""
        name = type(self.hip).config_name
        if self.autotune:
            name = 'fused48_fused96_fsum_s32' if n <= self.s32_max_crops else ('fused48_fused96_fsum' if n <= self.fsum_max_crops else name)
        self.tuned[n] = {'choice': name}
        return name

    fsum_max_crops = int(os.environ.get('PAM_FSUM_MAX', '20'))      # tuning hooks of config_for's rule
    s32_max_crops = int(os.environ.get('PAM_S32_MAX', '12'))        # up to here the deep branches' 3x3 layers run with 32-channel slabs (-8 ... -10 % at 2-6 crops, -2 % at 12)

    def input_buffer(self, n, slot=0):
        """The (N,3,H,W) channels-last bf16 tensor the preprocessing kernel writes; the replay's own input when one
        exists, so no copy is needed."""
        g = self._graphs.get((n, 'features', slot)) or self._graphs.get((n, 'heatmaps', slot))
        if g is not None:
            return g[1]
        H, W = self.resolution
        return torch.empty((n, self.in_channels, H, W), dtype=self.dtype, device=self.device, memory_format=torch.channels_last)

    # -- HIP kernels around it -----------------------------------------------------------------------------------------
    def preprocess(self, frame_ptrs, frame_h, frame_w, view_of, boxes, out):
        """frame_ptrs: int64 device tensor of per-view frame addresses; view_of int32 (N), boxes float32 (N,4) xywh.  out may hold more
        crops than N (a replay bucket): the extra ones repeat the last crop."""
        H, W = self.resolution
        st = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.pam_preprocess_crops_ex(C.c_void_p(st), int(view_of.numel()), int(out.shape[0]), C.c_void_p(frame_ptrs.data_ptr()),
                                              int(frame_h), int(frame_w), C.c_void_p(view_of.data_ptr()),
                                              C.c_void_p(boxes.data_ptr()), H, W, int(out.shape[1]), C.c_void_p(out.data_ptr()),
                                              1 if self.antialias else 0)
        if rc != 0:
            raise _lib.PamError('pam_preprocess_crops failed: %d' % rc)

    def decode(self, hm, view_of, slot_of, boxes, det, kp=None):
        """hm (N,17,h,w) float32 (channels-last or contiguous) -> det (C,max_dets,17,3) float64 rows (y,x,score)."""
        n, j, h, w = hm.shape
        nchw = hm.is_contiguous()
        assert nchw or hm.is_contiguous(memory_format=torch.channels_last)
        st = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.pam_decode_heatmaps(C.c_void_p(st), n, C.c_void_p(hm.data_ptr()), 1 if nchw else 0, h, w,
                                          C.c_void_p(view_of.data_ptr()), C.c_void_p(slot_of.data_ptr()),
                                          C.c_void_p(boxes.data_ptr()), det.shape[1], C.c_void_p(det.data_ptr()),
                                          C.c_void_p(kp.data_ptr()) if kp is not None else None)
        if rc != 0:
            raise _lib.PamError('pam_decode_heatmaps failed: %d' % rc)

    soft_beta = None        # None: hard arg-max decode (the parity mode).  A float > 0: soft-arg-max with that inverse temperature
                            # (sub-pixel keypoints = softmax(beta * heat-map)-weighted mean position; confidence = the maximum).

    def head_decode(self, f, view_of, slot_of, boxes, det, kp=None, heat=None, n=None):
        """Final 1x1 convolution + arg-max decode (soft-arg-max when ``self.soft_beta`` is set) in one pass over the features f
        (N,48,h,w channels-last bf16): det rows as ``decode``; the heat-maps are written only when ``heat`` (N,17,h,w float32
        channels-last) is given.  n: decode only the first n crops of f."""
        nf, c, h, w = f.shape
        n = nf if n is None else n
        assert f.is_contiguous(memory_format=torch.channels_last) and f.dtype == torch.bfloat16 and n <= nf
        soft = self.soft_beta is not None
        need = int((self.lib.pam_head_decode_soft_scratch_bytes if soft else self.lib.pam_head_decode_scratch_bytes)(n, h, w))
        if getattr(self, '_hd_scratch', None) is None or self._hd_scratch.numel() < need:
            self._hd_scratch = torch.empty((max(need, 1),), dtype=torch.uint8, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        head = (C.c_void_p(st), n, h, w, C.c_void_p(f.data_ptr()), c, C.c_void_p(self.head_w.data_ptr()),
                C.c_void_p(self.head_b.data_ptr()), self.head_w.shape[0])
        tail = (C.c_void_p(heat.data_ptr()) if heat is not None else None,
                C.c_void_p(view_of.data_ptr()), C.c_void_p(slot_of.data_ptr()), C.c_void_p(boxes.data_ptr()),
                det.shape[1], C.c_void_p(det.data_ptr()), C.c_void_p(kp.data_ptr()) if kp is not None else None,
                C.c_void_p(self._hd_scratch.data_ptr()))
        if soft:
            rc = self.lib.pam_head_decode_soft(*(head + (C.c_float(float(self.soft_beta)),) + tail))
        else:
            rc = self.lib.pam_head_decode(*(head + tail))
        if rc != 0:
            raise _lib.PamError('pam_head_decode%s failed: %d' % ('_soft' if soft else '', rc))

    # -- the reference-shaped entry point ------------------------------------------------------------------------------
    def predict(self, person_bbox_list, batch_size=20, conf_threshold=0.4):
        """person_bbox_list[view] = list of dicts with 'bbox' [x, y, w, h] and 'data' (BGR uint8 HxWx3 ndarray or CUDA
        tensor) -> dump_results[view] = list of dicts {bbox, keypoints (51: x, y, score), keypoints_score (17), feature}.

        The returned list is a ``DumpResults``: besides the reference's dicts it keeps the decoded keypoints on the device in
        the tracker's input layout ((views, max_dets, 17, 3) float64 rows (y, x, score) + per-view counts), so that
        ``ivclabpose.PersonTrack_Project3DPose`` can hand them to the frame kernel without a host round trip when the caller
        passes the dump on unchanged."""
        V = len(person_bbox_list)
        if not self._arenas:
            self.max_crops = max(self.max_crops, int(batch_size))
        views, slots, boxes, frames, cnt = [], [], [], {}, [0] * V
        for v, persons in enumerate(person_bbox_list):
            for p in persons:
                views.append(v); slots.append(cnt[v]); cnt[v] += 1; boxes.append(list(p['bbox']))     # copied: the dump is built lazily
                if v not in frames:
                    d = p['data']
                    if not torch.is_tensor(d):
                        d = torch.from_numpy(np.ascontiguousarray(d))
                    frames[v] = d.to(self.device, non_blocking=True).contiguous()
        n = len(views)
        out = DumpResults([[] for _ in person_bbox_list])
        if n == 0:
            return out
        any_frame = next(iter(frames.values()))
        fh, fw = any_frame.shape[0], any_frame.shape[1]
        # one upload for all the small per-call tables: [view_of n | slot_of n | boxes 4n (f32 bits) | n_det V | frame ptrs V (i64)]
        meta = np.empty(6 * n + V + (V & 1) + 2 * V, dtype=np.int32)
        meta[:n] = views; meta[n:2 * n] = slots
        meta[2 * n:6 * n].view(np.float32)[:] = np.asarray(boxes, dtype=np.float32).reshape(-1)
        meta[6 * n:6 * n + V] = cnt
        o_ptr = 6 * n + V + (V & 1)
        meta[o_ptr:].view(np.int64)[:] = [frames[v].data_ptr() if v in frames else 0 for v in range(V)]
        # staged through a pinned buffer of its own (two per size, alternating): the upload is asynchronous and the host goes on issuing
        # (each buffer carries the event recorded behind its last upload: a host that runs more than two calls ahead of the GPU -- device
        # frames, dumps dropped unread -- must not overwrite a table whose copy has not been issued to the device yet)
        stage = self._meta_pinned.setdefault(meta.size, [])
        if len(stage) < 2:
            stage.append([torch.empty(meta.size, dtype=torch.int32).pin_memory(), None])
        stage.reverse()
        if stage[0][1] is not None:
            stage[0][1].synchronize()
        stage[0][0].numpy()[:] = meta
        m = stage[0][0].to(self.device, non_blocking=True)
        stage[0][1] = torch.cuda.Event()
        stage[0][1].record(torch.cuda.current_stream(self.device))
        view_of, slot_of = m[:n], m[n:2 * n]
        bx = m[2 * n:6 * n].view(torch.float32).reshape(n, 4)
        n_det = m[6 * n:6 * n + V]
        ptrs = m[o_ptr:].view(torch.int64)
        det = torch.empty((V, max(self.max_dets, max(cnt)), 17, 3), dtype=torch.float64, device=self.device)
        kp = torch.empty((n, 17, 3), dtype=torch.float32, device=self.device)
        lo, hi = 0, n
        if self.world > 1:                                # this rank's share of the call's crops (ordered by view, then person)
            from .distributed import crop_partition, check_same_call
            if os.environ.get('PAM_CHECK_SHARD', '0') == '1':    # debug: a rank with another crop list would hang or mis-assemble the gather
                check_same_call(n, V, self.device, self.group)
            lo, hi = crop_partition(n, self.world)[self.rank]
        host = self._pinned_kp(n)

        def issue():
            """The device side of this call (crop -> conv stack -> head + arg-max per batch, the exchange, the copy of the keypoints to
            pinned host memory); DumpResults runs it again when a gate of one of its forwards gave up (clear_void first)."""
            kpl = kp
            for s in range(lo, hi, batch_size):
                e = min(hi, s + batch_size)
                k = e - s
                mp = min(batch_size, (k + self.graph_bucket - 1) // self.graph_bucket * self.graph_bucket) if batch_size >= self.graph_bucket else k
                x = self.input_buffer(mp)                    # mp > k: the crop kernel repeats the last crop into the bucket's spare rows
                self.preprocess(ptrs, fh, fw, view_of[s:e], bx[s:e], x)
                self.head_decode(self.features(x), view_of[s:e], slot_of[s:e], bx[s:e], det, kp[s:e], n=k)
            if self.world > 1:
                from .distributed import gather_crop_keypoints
                kpl = gather_crop_keypoints(kp[lo:hi].contiguous(), n, self.world, self.rank, self.group)
                # the tracker's device-side input, rebuilt from the gathered rows: (view, slot) <- (y, x, score) as float64
                det[view_of.long(), slot_of.long()] = kpl[:, :, [1, 0, 2]].double()
            # The reference's contract is host lists -- but nothing needs them before the caller looks: the device -> host copy of the
            # keypoints is only ENQUEUED here (pinned buffer, this stream) and the per-person dicts are built at the first access of the
            # dump (DumpResults).  A loop that passes the dump straight on to PersonTrack_Project3DPose waits for the GPU once per frame
            # (behind the tracker kernel) instead of twice, and the tracker kernel is queued right behind the decode instead of after a
            # host round trip.
            host.copy_(kpl, non_blocking=True)
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            return done
        out.attach_pending(det, n_det, host, issue(), views, boxes, cnt)
        out._net, out._issue = self, issue
        import weakref
        self._kp_last[1] = weakref.ref(out)
        return out

    def _pinned_kp(self, n):
        """A pinned (n, 17, 3) float32 host buffer for the keypoints of one call; two per crop count, used alternately (a dump that is
        still pending when its buffer comes round again is materialised first)."""
        ring = self._kp_pinned.setdefault(n, [])
        if len(ring) < 2:
            ring.append([torch.empty((n, 17, 3), dtype=torch.float32).pin_memory(), None])
            ent = ring[-1]
        else:
            ent = ring[0]; ring.reverse()
            prev = ent[1]() if ent[1] is not None else None
            if prev is not None:
                prev._materialise()
        self._kp_last = ent
        return ent[0]


class DumpResults(list):
    """``dump_results`` of the reference (list per view of person dicts) + the same keypoints still on the device.

    The per-person dicts are built LAZILY: ``predict`` only enqueues the device -> host copy of the keypoints; the first access to the
    list's content (indexing, iteration, comparison, printing, copying, pickling ...) waits for that copy and fills the per-view lists
    in place.  Until then nobody can have edited the dicts, so the device copy is trivially the truth."""
    device_det = None        # (views, max_dets, 17, 3) float64 rows (y, x, score) at (view, slot)
    device_n_det = None      # (views,) int32
    _poses_host = None       # per view (n, 17, 3) float64 (y, x, score): what ivclabpose._unpack would rebuild from the dicts
    _pending = None          # (pinned host keypoints, event, views, boxes, per-view counts) until the first access
    _witness = None

    def attach(self, det, n_det, poses_host):
        self.device_det, self.device_n_det, self._poses_host = det, n_det, poses_host
        self._seal()

    def attach_pending(self, det, n_det, host_kp, event, views, boxes, cnt):
        self.device_det, self.device_n_det = det, n_det
        self._pending = (host_kp, event, views, boxes, cnt)

    def _seal(self):
        # complete witness of what the tracker would read from the dicts (ivclabpose._unpack: 'keypoints' AND 'keypoints_score'):
        # object identities + a hash of all 51 + 17 numbers per person -- an interior edit or a re-scored joint invalidates it
        self._witness = [[(id(it), id(it['keypoints']), id(it['keypoints_score']),
                           hash(tuple(it['keypoints'])), hash(tuple(it['keypoints_score']))) for it in items] for items in list.__iter__(self)]

    _net = None              # the HRNetPose that made this dump and the device side of its call (re-run when a gate gave up)
    _issue = None

    def redo_if_void(self):
        """The forwards of this call are void when a device-side gate of one of them timed out (HRNetPose.check_void): run the call's
        device side again -- the object is on stream events by then -- into the same buffers.  -> True when it did.  The reference's
        predict never returns keypoints it later disowns (/root/reference/src/ivclabpose.py:208-212); neither does this."""
        net = self._net
        if net is None or self._issue is None or self._pending is None or not net.check_void():
            return False
        net.clear_void()
        self._pending = (self._pending[0], self._issue()) + tuple(self._pending[2:])
        return True

    def _host_rows(self):
        """(n, 17, 3) float64 (x, y, score) rows of the call, waiting for the copy predict() enqueued."""
        self._pending[1].synchronize()
        if self.redo_if_void():
            self._pending[1].synchronize()
        return self._pending[0].numpy().astype(np.float64)

    @property
    def poses_host(self):
        if self._poses_host is None and self._pending is not None:
            kp_h, cnt = self._host_rows(), self._pending[4]
            yxs = kp_h[:, :, [1, 0, 2]]
            first = np.concatenate([[0], np.cumsum(cnt)])
            self._poses_host = [yxs[first[v]:first[v + 1]] for v in range(len(cnt))]
        return self._poses_host

    @poses_host.setter
    def poses_host(self, value):
        self._poses_host = value

    def _materialise(self):
        if self._pending is None:
            return
        kp_h = self._host_rows()
        _, _, views, boxes, cnt = self._pending
        _ = self.poses_host
        self._pending = None
        self._issue = None                                 # (the call's frames and tables are released with it)
        n = len(views)
        flat = kp_h.reshape(n, 51).tolist()
        score = kp_h[:, :, 2].tolist()
        for i in range(n):
            list.__getitem__(self, views[i]).append(dict(bbox=list(boxes[i]), keypoints=flat[i], keypoints_score=score[i], feature=[]))
        self._seal()

    def device_valid(self):
        """True while the dicts are exactly what predict() returned: never looked at yet, or same objects and every keypoint /
        keypoints_score value untouched (the tracker reads both, /root/reference/src/ivclabpose.py:236-244).  Any edit -- masking or
        re-scoring joints between PersonPoseDetect and PersonTrack_Project3DPose -- sends the call through the host dicts again."""
        if self.device_det is None:
            return False
        if self._pending is not None:
            return True
        if self._witness is None or len(self._witness) != list.__len__(self):
            return False
        for items, wit in zip(list.__iter__(self), self._witness):
            if len(items) != len(wit):
                return False
            for it, (a, b, c, hk, hs) in zip(items, wit):
                k, sc = it.get('keypoints'), it.get('keypoints_score')
                if id(it) != a or id(k) != b or id(sc) != c or len(k) != 51 or len(sc) != 17:
                    return False
                if hash(tuple(k)) != hk or hash(tuple(sc)) != hs:
                    return False
        return True


def _lazy(name):
    base = getattr(list, name)

    def method(self, *a, **k):
        self._materialise()
        return base(self, *a, **k)
    method.__name__ = name
    return method


# every way Python code (and the C helpers that go through the iterator protocol for list SUBCLASSES: list(), json, pickle, copy)
# can see the content first fills it in; __len__ is the number of views and needs no data
for _n in ('__getitem__', '__iter__', '__reversed__', '__contains__', '__eq__', '__ne__', '__lt__', '__le__', '__gt__', '__ge__', '__repr__',
           '__add__', '__iadd__', '__mul__', '__imul__', '__rmul__', '__setitem__', '__delitem__', '__reduce_ex__', 'append', 'extend',
           'insert', 'pop', 'remove', 'index', 'count', 'copy', 'sort', 'reverse', 'clear'):
    setattr(DumpResults, _n, _lazy(_n))
del _n


def _radd(self, other):
    # `[] + dump`: list.__add__ of the LEFT operand is a C fast path that reads a list subclass's items directly (it would see the empty
    # per-view lists); Python tries the right operand's __radd__ first when its type is a subclass of the left one's -- fill in here
    self._materialise()
    return list(other) + list(list.__iter__(self))


DumpResults.__radd__ = _radd


def measure_bf16_drift(net, n_crops=2, seed=2, checker_device='cpu'):
    """Self-consistency number of the bf16 conv stack (parity with the authors' backend is unpinned: no weights offline): the same
    folded weights run as a plain fp32 PyTorch module (the checker; on the CPU by default, so that no PyTorch-ROCm / MIOpen kernel
    appears in a profile of the run) vs ``net``'s product path, on seeded random crops.  -> relative L2 error of the heat-maps,
    fraction of (crop, joint) arg-max cells that moved, and the largest move in heat-map cells.  With random weights the heat-maps are
    nearly flat noise, so the arg-max fraction is a pessimistic figure."""
    dev = net.device
    cdev = torch.device(checker_device)
    seed_w = int(net.weights.split('=')[1].rstrip(')')) if net.weights.startswith('random') else 0
    if not net.weights.startswith('random'):
        raise RuntimeError('measure_bf16_drift rebuilds the fp32 module from the seed; load the checkpoint into both to use it with real weights')
    ref = fold_batchnorm(init_random(PoseHighResolutionNet(), seed=seed_w)).to(cdev).eval()
    g = torch.Generator().manual_seed(seed)
    x32 = torch.randn((n_crops, 3, net.resolution[0], net.resolution[1]), generator=g).to(dev)
    xb = x32.to(torch.bfloat16)
    x8 = torch.cat([xb, torch.zeros((n_crops, net.in_channels - 3, net.resolution[0], net.resolution[1]), dtype=torch.bfloat16, device=dev)], dim=1) \
        if net.in_channels > 3 else xb
    x8 = x8.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        if cdev.type == 'cpu':
            nthr = torch.get_num_threads()
            torch.set_num_threads(min(16, os.cpu_count() or 16))       # many-core hosts: a full team is slower by orders of magnitude
            h32 = ref(xb.float().to(cdev)).to(dev)
            torch.set_num_threads(nthr)
        else:
            h32 = ref(xb.float())
        hb = net.heatmaps(x8).float()
    return drift_statistics(h32, hb, dict(crops=n_crops, weights=net.weights,
                                          checker='same folded weights as a plain fp32 PyTorch module on the %s' % cdev.type))


def drift_statistics(h32, hb, extra=None, planted_seed=5):
    """What the bf16 stack's error does to the DECODE, in terms that bound something (random weights give nearly flat heat-maps whose
    arg-max moves at the slightest error -- that rate alone says nothing about the kernels):
      * ``rms_err``: RMS of (bf16 heat-map - fp32 heat-map) over all cells;
      * decided joints: maps whose fp32 peak stands above every cell OUTSIDE its 3 x 3 neighbourhood by a margin > k x rms_err; on those the
        bf16 arg-max should stay within one cell.  ``decided[k]`` = {frac, joints, max_cells} for k = 4, 8 (k = 8 is asserted in
        tests/test_gpu_image.py: moving such a peak takes two cells whose errors differ by 8 sigma);
      * planted peaks: one cell per map (seeded position) raised k x rms_err above the map's maximum in BOTH heat-maps, i.e. a joint the
        network is sure about by k error sigmas: ``planted_peak_max_cells[k]`` = largest distance in cells between the two decodes, k = 4,
        8, 16 (0 expected from k = 8; k = 16 asserted);
      * the unconditioned rate (``argmax_moved_frac``) stays for the record."""
    n, j, h, w = h32.shape
    f32, fb = h32.flatten(2), hb.flatten(2)
    a32, ab = f32.argmax(2), fb.argmax(2)
    moved = a32 != ab
    cells = torch.maximum((a32 // w - ab // w).abs(), (a32 % w - ab % w).abs())
    rms = float((hb - h32).pow(2).mean().sqrt())
    # margin of the fp32 peak over the rest of the map (3 x 3 neighbourhood of the peak excluded)
    yy = torch.arange(h, device=h32.device).view(1, 1, h, 1); xx = torch.arange(w, device=h32.device).view(1, 1, 1, w)
    py, px = (a32 // w).view(n, j, 1, 1), (a32 % w).view(n, j, 1, 1)
    near = ((yy - py).abs() <= 1) & ((xx - px).abs() <= 1)
    rest = h32.masked_fill(near, float('-inf')).flatten(2).max(2)[0]
    margin = f32.max(2)[0] - rest
    decided = {}
    for k in (4, 8):
        d = margin > k * rms
        decided[str(k)] = dict(frac=float(d.float().mean()), joints=int(d.sum()), max_cells=int(cells[d].max()) if bool(d.any()) else 0)
    planted = {}
    g = torch.Generator().manual_seed(planted_seed)
    pos = (torch.randint(0, h, (n, j), generator=g) * w + torch.randint(0, w, (n, j), generator=g)).to(h32.device).view(n, j, 1)
    top = f32.max(2, keepdim=True)[0]
    for k in (4, 8, 16):
        amp = (top - torch.gather(f32, 2, pos)) + k * rms              # the planted cell ends k x rms above the map's own maximum
        p32, pb = f32.scatter_add(2, pos, amp).argmax(2), fb.scatter_add(2, pos, amp).argmax(2)
        planted[str(k)] = int(torch.maximum((p32 // w - pb // w).abs(), (p32 % w - pb % w).abs()).max())
    out = dict(rel_l2_err=float((hb - h32).norm() / h32.norm()), rms_err=rms, decided=decided, planted_peak_max_cells=planted,
               argmax_moved_frac=float(moved.float().mean()), argmax_max_cells=int(cells.max()),
               argmax_mean_cells_when_moved=float(cells[moved].float().mean()) if bool(moved.any()) else 0.0,
               score_max_abs_err=float((fb.max(2)[0] - f32.max(2)[0]).abs().max()),
               headline='decided joints (fp32 peak margin > k x rms error) move by decided[k].max_cells; planted peaks of k x rms by planted_peak_max_cells[k]')
    out.update(extra or {})
    return out


def smoke_check():
    """Tiny self-check used by __graft_entry__.smoke(): 2 crops, decode vs torch arg-max, preprocess vs torch."""
    dev = torch.device('cuda:0')
    net = HRNetPose(48, 17, None, resolution=(384, 288), use_graph=False)
    g = torch.Generator().manual_seed(0)
    frame = torch.randint(0, 256, (2, 480, 640, 3), dtype=torch.uint8, generator=g).to(dev)
    ptrs = torch.tensor([frame[0].data_ptr(), frame[1].data_ptr()], dtype=torch.int64, device=dev)
    view_of = torch.tensor([0, 1], dtype=torch.int32, device=dev)
    slot_of = torch.tensor([0, 0], dtype=torch.int32, device=dev)
    boxes = torch.tensor([[100.0, 50.0, 200.0, 300.0], [300.5, 100.25, 150.0, 280.0]], dtype=torch.float32, device=dev)
    x = net.input_buffer(2)
    net.preprocess(ptrs, 480, 640, view_of, boxes, x)
    ref = reference_preprocess(frame, view_of, boxes, (384, 288))
    err = (x[:, :3].float() - ref).abs().max().item()
    assert err < 0.03, err                               # bf16 rounding of values in [-2.2, 2.7]
    assert x.shape[1] == 3 or float(x[:, 3:].float().abs().max()) == 0.0
    hm = net.heatmaps(x)
    det = torch.zeros((2, 4, 17, 3), dtype=torch.float64, device=dev)
    net.decode(hm, view_of, slot_of, boxes, det)
    exp = reference_decode(hm, boxes)
    assert torch.equal(det[:, 0], exp), (det[:, 0] - exp).abs().max()


def reference_preprocess(frames, view_of, boxes, resolution):
    """Plain torch float32 restatement of the preprocessing kernel (tests): half-pixel bilinear sample of the box."""
    H, W = resolution
    n = view_of.numel()
    out = torch.empty((n, 3, H, W), dtype=torch.float32, device=frames.device)
    fh, fw = frames.shape[1], frames.shape[2]
    mean = torch.tensor([0.485, 0.456, 0.406], device=frames.device).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=frames.device).view(3, 1, 1)
    for i in range(n):
        img = frames[int(view_of[i])].float()
        bx, by, bw, bh = [boxes[i, k] for k in range(4)]
        sx = (bx + (torch.arange(W, device=frames.device, dtype=torch.float32) + 0.5) * (bw / W) - 0.5).clamp(0, fw - 1)
        sy = (by + (torch.arange(H, device=frames.device, dtype=torch.float32) + 0.5) * (bh / H) - 0.5).clamp(0, fh - 1)
        x0 = sx.floor().long(); y0 = sy.floor().long()
        x1 = (x0 + 1).clamp(max=fw - 1); y1 = (y0 + 1).clamp(max=fh - 1)
        fx = (sx - x0.float()).view(1, W, 1); fy = (sy - y0.float()).view(H, 1, 1)
        a = img[y0][:, x0]; b = img[y0][:, x1]; c = img[y1][:, x0]; d = img[y1][:, x1]
        top = a + (b - a) * fx; bot = c + (d - c) * fx
        v = (top + (bot - top) * fy) / 255.0                    # (H,W,3) BGR
        out[i] = (v.flip(2).permute(2, 0, 1) - mean) / std
    return out


def reference_decode(hm, boxes):
    """torch restatement of the decode kernel: hard arg-max (first maximum), upstream SimpleHRNet box mapping."""
    n, j, h, w = hm.shape
    flat = hm.float().reshape(n, j, h * w)
    val, idx = flat.max(dim=2)
    # torch.max may return any maximal index on ties; recompute the first one
    first = (flat == val.unsqueeze(2)).float().argmax(dim=2)
    py = (first // w).double(); px = (first % w).double()
    b = boxes.double()
    y = (py / h * b[:, 3:4] + b[:, 1:2]).float().double()
    x = (px / w * b[:, 2:3] + b[:, 0:1]).float().double()
    return torch.stack([y, x, val.double()], dim=2)
