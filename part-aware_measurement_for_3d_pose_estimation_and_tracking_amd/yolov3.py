"""YOLOv3 person detector: the ``backend.YOLOv3.YOLOv3`` that ivclabpose constructs (/root/reference/src/ivclabpose.py:116-120)
and PersonDetect calls on the list of view images (ivclabpose.py:183-204: one result per image, rows whose first five
entries are x1, y1, x2, y2, score in frame pixels).

The backend itself, its cfg and its weights are not part of the reference tree, so this module follows the public Darknet
YOLOv3 definition (Redmon & Farhadi 2018: Darknet-53 backbone, three ``yolo`` heads, 9 anchors) -- parity unpinned, like
HRNet.  It understands Darknet ``.cfg`` text (convolutional / shortcut / route / upsample / yolo) and Darknet ``.weights``
files; with no cfg it builds the standard 416 x 416, 80-class network, with no weights a seeded random one.

Two executions of the same network: ``Darknet`` (plain PyTorch float32; the test reference) and ``HipDarknet`` (product
path: every convolution on the MFMA kernels of csrc/pam_conv.hip with BN folded, leaky-ReLU and the shortcut add fused into
the epilogue; resize, upsample+route and box decode + NMS in csrc/pam_detect.hip; one hipGraph per batch shape)."""
import ctypes as C
import os
import struct
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .hrnet_hip import ConvEngine, PackedConv

ANCHORS = [(10, 13), (16, 30), (33, 23), (30, 61), (62, 45), (59, 119), (116, 90), (156, 198), (373, 326)]


# ---- cfg ------------------------------------------------------------------------------------------------------------
def default_cfg(width=416, height=416, classes=80):
    """The standard yolov3.cfg (107 layers), generated rather than shipped."""
    out = ['[net]', 'width=%d' % width, 'height=%d' % height, 'channels=3', '']

    def conv(filters, size, stride=1, bn=1, act='leaky'):
        out.extend(['[convolutional]'] + (['batch_normalize=1'] if bn else []) +
                   ['filters=%d' % filters, 'size=%d' % size, 'stride=%d' % stride, 'pad=1', 'activation=%s' % act, ''])

    def stage(filters, blocks):
        conv(filters, 3, 2)
        for _ in range(blocks):
            conv(filters // 2, 1)
            conv(filters, 3)
            out.extend(['[shortcut]', 'from=-3', 'activation=linear', ''])

    def head(filters, mask):
        for _ in range(3):
            conv(filters, 1)
            conv(filters * 2, 3)
        conv(3 * (5 + classes), 1, bn=0, act='linear')
        out.extend(['[yolo]', 'mask = %s' % ','.join(str(m) for m in mask),
                    'anchors = %s' % ',  '.join('%d,%d' % a for a in ANCHORS), 'classes=%d' % classes, 'num=9', ''])

    conv(32, 3)
    for f, b in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        stage(f, b)
    head(512, (6, 7, 8))
    for skip, f, mask in ((61, 256, (3, 4, 5)), (36, 128, (0, 1, 2))):
        out.extend(['[route]', 'layers = -4', ''])
        conv(f, 1)
        out.extend(['[upsample]', 'stride=2', '', '[route]', 'layers = -1, %d' % skip, ''])
        head(f, mask)
    return '\n'.join(out)


def parse_cfg(text):
    """Darknet cfg text -> (net options, [layer dicts]); values stay strings except the ones the network needs."""
    blocks = []
    for line in text.splitlines():
        line = line.split('#')[0].strip()
        if not line:
            continue
        if line.startswith('['):
            blocks.append({'type': line[1:-1].strip()})
        else:
            k, v = line.split('=', 1)
            blocks[-1][k.strip()] = v.strip()
    if not blocks or blocks[0]['type'] != 'net':
        raise ValueError('cfg must start with [net]')
    net, layers = blocks[0], blocks[1:]
    for b in layers:
        t = b['type']
        if t == 'convolutional':
            for k, d in (('batch_normalize', 0), ('filters', None), ('size', None), ('stride', 1), ('pad', 0)):
                b[k] = int(b.get(k, d))
            b.setdefault('activation', 'linear')
        elif t == 'shortcut':
            b['from'] = int(b['from'])
            b.setdefault('activation', 'linear')
        elif t == 'route':
            b['layers'] = [int(x) for x in b['layers'].split(',')]
        elif t == 'upsample':
            b['stride'] = int(b.get('stride', 2))
        elif t == 'yolo':
            b['mask'] = [int(x) for x in b['mask'].split(',')]
            a = [float(x) for x in b['anchors'].split(',')]
            b['anchors'] = [(a[i], a[i + 1]) for i in range(0, len(a), 2)]
            b['classes'] = int(b['classes'])
        else:
            raise NotImplementedError('cfg layer [%s] is not supported (convolutional/shortcut/route/upsample/yolo only)' % t)
    return net, layers


# ---- float32 PyTorch form (reference for the tests; also the container of the weights) --------------------------------------
class Darknet(nn.Module):
    def __init__(self, cfg_text=None):
        super(Darknet, self).__init__()
        self.net, self.layers = parse_cfg(cfg_text if cfg_text is not None else default_cfg())
        self.width, self.height = int(self.net.get('width', 416)), int(self.net.get('height', 416))
        chans = []
        mods = nn.ModuleList()
        c_in = int(self.net.get('channels', 3))
        for i, b in enumerate(self.layers):
            m = nn.Sequential()
            t = b['type']
            if t == 'convolutional':
                pad = (b['size'] - 1) // 2 if b['pad'] else 0
                m.add_module('conv', nn.Conv2d(c_in, b['filters'], b['size'], b['stride'], pad, bias=not b['batch_normalize']))
                if b['batch_normalize']:
                    m.add_module('bn', nn.BatchNorm2d(b['filters']))
                if b['activation'] == 'leaky':
                    m.add_module('act', nn.LeakyReLU(0.1))
                elif b['activation'] != 'linear':
                    raise NotImplementedError('activation %s' % b['activation'])
                c_out = b['filters']
            elif t == 'shortcut':
                c_out = chans[i - 1]
            elif t == 'route':
                c_out = sum(chans[l if l >= 0 else i + l] for l in b['layers'])
            else:                       # upsample, yolo
                c_out = chans[i - 1]
            mods.append(m)
            chans.append(c_out)
            c_in = c_out
        self.mods, self.chans = mods, chans

    def conv_modules(self):
        return [m for m, b in zip(self.mods, self.layers) if b['type'] == 'convolutional']

    def yolo_layers(self):
        return [b for b in self.layers if b['type'] == 'yolo']

    def forward(self, x):
        """x: (N, 3, H, W) float RGB in [0, 1] -> raw head tensors [(N, 3*(5+nc), g, g)] in cfg order (coarsest first)."""
        outs, heads = [], []
        for i, (b, m) in enumerate(zip(self.layers, self.mods)):
            t = b['type']
            if t == 'convolutional':
                x = m(x)
            elif t == 'shortcut':
                x = outs[i - 1] + outs[i + b['from']]
            elif t == 'route':
                xs = [outs[l if l >= 0 else i + l] for l in b['layers']]
                x = xs[0] if len(xs) == 1 else torch.cat(xs, 1)
            elif t == 'upsample':
                x = F.interpolate(x, scale_factor=b['stride'], mode='nearest')
            else:
                heads.append(x)
            outs.append(x)
        return heads

    # Darknet .weights: int32 major, minor, revision; `seen` (uint64 when major*10+minor >= 2 else uint32); then per
    # convolutional layer, in order: [bn beta, gamma, running mean, running var | conv bias], conv weights (OIHW), float32.
    def load_darknet_weights(self, path):
        with open(path, 'rb') as f:
            major, minor, _ = struct.unpack('<3i', f.read(12))
            f.read(8 if major * 10 + minor >= 2 else 4)
            buf = np.frombuffer(f.read(), dtype='<f4')
        pos = 0

        def take(t):
            nonlocal pos
            n = t.numel()
            if pos + n > buf.size:
                raise ValueError('%s: weights file too short' % path)
            t.data.copy_(torch.from_numpy(buf[pos:pos + n].copy()).view_as(t))
            pos += n
        for m in self.conv_modules():
            if hasattr(m, 'bn'):
                take(m.bn.bias); take(m.bn.weight); take(m.bn.running_mean); take(m.bn.running_var)
            else:
                take(m.conv.bias)
            take(m.conv.weight)
        if pos != buf.size:
            raise ValueError('%s: %d floats left over (cfg / weights mismatch)' % (path, buf.size - pos))

    def save_darknet_weights(self, path):
        with open(path, 'wb') as f:
            f.write(struct.pack('<3i', 0, 2, 0)); f.write(struct.pack('<Q', 0))
            for m in self.conv_modules():
                ts = [m.bn.bias, m.bn.weight, m.bn.running_mean, m.bn.running_var] if hasattr(m, 'bn') else [m.conv.bias]
                for t in ts + [m.conv.weight]:
                    f.write(t.detach().cpu().numpy().astype('<f4').tobytes())

    def init_random(self, seed=0):
        """He-style weights with tame residual branches and mildly negative objectness, so a random net is numerically
        well-behaved (activations O(1) through 75 convolutions) and emits few boxes."""
        g = torch.Generator().manual_seed(seed)
        for i, (m, b) in enumerate(zip(self.mods, self.layers)):
            if b['type'] != 'convolutional':
                continue
            fan = m.conv.in_channels * m.conv.kernel_size[0] ** 2
            m.conv.weight.data = torch.randn(m.conv.weight.shape, generator=g) * (2.0 / fan) ** 0.5
            residual = i + 1 < len(self.layers) and self.layers[i + 1]['type'] == 'shortcut'
            if hasattr(m, 'bn'):
                m.bn.weight.data = (0.3 if residual else 1.0) * (0.8 + 0.4 * torch.rand(m.bn.weight.shape, generator=g))
                m.bn.bias.data = 0.1 * torch.randn(m.bn.bias.shape, generator=g)
                m.bn.running_mean.data = 0.1 * torch.randn(m.bn.bias.shape, generator=g)
                m.bn.running_var.data = 0.8 + 0.4 * torch.rand(m.bn.bias.shape, generator=g)
            else:                                   # a [yolo] head: objectness logits biased down
                m.conv.weight.data *= 0.2
                m.conv.bias.data = 0.1 * torch.randn(m.conv.bias.shape, generator=g)
                m.conv.bias.data[4::m.conv.out_channels // 3] -= 2.0
        return self


def fold_conv(m):
    """(conv [, bn]) -> one nn.Conv2d with bias (inference form)."""
    conv = m.conv
    new = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, bias=True)
    with torch.no_grad():
        if hasattr(m, 'bn'):
            s = m.bn.weight / torch.sqrt(m.bn.running_var + m.bn.eps)
            new.weight.copy_(conv.weight * s.reshape(-1, 1, 1, 1))
            new.bias.copy_(m.bn.bias - m.bn.running_mean * s)
        else:
            new.weight.copy_(conv.weight); new.bias.copy_(conv.bias)
    return new


# ---- product path -----------------------------------------------------------------------------------------------------
def _round_channels(c):
    return c if (c % 48 == 0 or c % 64 == 0) else (c + 63) // 64 * 64


class HipDarknet(ConvEngine):
    """The cfg's layer list compiled to kernel launches.  Channel counts the MFMA kernels cannot take (3 -> 8 on the input,
    32 / 255 -> 64 / 256 on outputs) are zero-padded at the end of the channel axis; padded channels stay exactly zero
    through leaky-ReLU and shortcut adds, and the head decode reads the real ones through its channel stride."""

    def __init__(self, model, device):
        self.lib = _lib.load()
        self.device = device
        self.count = None
        self.tile_cfg = -1
        self.unfuse_wide = True
        layers = model.layers
        n = len(layers)
        used_by = [[] for _ in range(n)]                 # consumers of each layer's output other than the next layer
        for i, b in enumerate(layers):
            if b['type'] == 'shortcut':
                used_by[i + b['from']].append(i)
            elif b['type'] == 'route':
                for l in b['layers']:
                    used_by[l if l >= 0 else i + l].append(i)
        self.plan, real, padded = [], [], []
        c_real, c_pad = int(model.net.get('channels', 3)), 8
        i = 0
        while i < n:
            b, t = layers[i], layers[i]['type']
            if t == 'convolutional':
                cin_pad = c_pad if i == 0 else padded[i - 1]
                stem = i == 0 and cin_pad == 8 and b['size'] == 3 and b['pad'] and b['stride'] in (1, 2) and b['filters'] in (32, 64)
                op = PackedConv(fold_conv(model.mods[i]), device, pad_cin_to=cin_pad,
                                pad_cout_to=None if stem else _round_channels(b['filters']))      # k_conv_stem writes 32 real channels
                nxt = layers[i + 1] if i + 1 < n else None
                fuse = (nxt is not None and nxt['type'] == 'shortcut' and nxt['activation'] == 'linear' and used_by[i] == [] and
                        b['activation'] in ('leaky', 'linear'))
                if fuse:                                    # conv + leaky, then + skip: one launch produces layer i+1
                    self.plan.append(('conv', i + 1, op, i - 1, b['activation'], i + 1 + nxt['from']))
                    real += [b['filters'], b['filters']]; padded += [op.cout, op.cout]
                    i += 2
                    continue
                self.plan.append(('conv', i, op, i - 1, b['activation'], None))
                real.append(b['filters']); padded.append(op.cout)
            elif t == 'shortcut':
                self.plan.append(('add', i, i - 1, i + b['from']))
                real.append(real[i - 1]); padded.append(padded[i - 1])
            elif t == 'route':
                src = [l if l >= 0 else i + l for l in b['layers']]
                if len(src) == 1:
                    self.plan.append(('alias', i, src[0]))
                    real.append(real[src[0]]); padded.append(padded[src[0]])
                else:
                    raise NotImplementedError('route over several layers is only supported right after [upsample]')
            elif t == 'upsample':
                nxt = layers[i + 1] if i + 1 < n else None
                if not (b['stride'] == 2 and nxt is not None and nxt['type'] == 'route' and len(nxt['layers']) == 2 and
                        nxt['layers'][0] in (-1, i) and used_by[i] == [i + 1]):
                    raise NotImplementedError('[upsample] must be stride 2 and feed `route = -1, <skip>`')
                skip = nxt['layers'][1] if nxt['layers'][1] >= 0 else i + 1 + nxt['layers'][1]
                if real[i - 1] != padded[i - 1] or real[skip] != padded[skip]:
                    raise NotImplementedError('route over channel-padded layers')
                self.plan.append(('upcat', i + 1, i - 1, skip))
                real += [real[i - 1], real[i - 1] + real[skip]]; padded += [padded[i - 1], padded[i - 1] + padded[skip]]
                i += 2
                continue
            else:                                           # yolo
                self.plan.append(('head', i, i - 1))
                real.append(real[i - 1]); padded.append(padded[i - 1])
            i += 1
        self.real, self.padded = real, padded

    def forward(self, x8):
        """x8: (N, 8, H, W) channels-last bf16 (RGB in [0,1] + 5 zero channels) -> head tensors, channels-last bf16."""
        outs, heads = {-1: x8}, []
        for step in self.plan:
            kind, dst = step[0], step[1]
            if kind == 'conv':
                _, _, op, src, act, skip = step
                if skip is not None and self.unfuse_wide and op.kh == 3 and op.stride == 1 and outs[src].shape[3] + 2 > 138:
                    # rows too wide for the rows-in-LDS 3x3 kernels (the 208-wide block of the 416 x 416 network): with the shortcut
                    # folded in, the layer fell to the classic implicit GEMM (90 us for 5 views); as a residual-free convolution it runs on
                    # the streamed implicit GEMM and the shortcut is one k_upsample_add (round 5: 90 -> 25 + 17 us)
                    y = self.conv(op, outs[src], relu=act)
                    outs[dst] = self.upsample_add(y, [outs[skip]], [0], relu=False)
                    continue
                outs[dst] = self.conv(op, outs[src], res=outs[skip] if skip is not None else None, relu=act,
                                      res_after_act=skip is not None)
            elif kind == 'add':
                outs[dst] = self.upsample_add(outs[step[2]], [outs[step[3]]], [0], relu=False)
            elif kind == 'alias':
                outs[dst] = outs[step[2]]
            elif kind == 'upcat':
                outs[dst] = self.upsample_concat(outs[step[2]], outs[step[3]])
            else:
                outs[dst] = outs[step[2]]
                heads.append(outs[dst])
        return heads


class YOLOv3(object):
    """Same constructor arguments and call form as the reference's ``backend.YOLOv3.YOLOv3`` use (ivclabpose.py:117-119,186):
    ``YOLOv3(cfg, weight, class_names, score_thresh=, nms_thresh=, use_cuda=)``; ``detector(imglist)`` -> one (k, 5) float32
    array per image, rows (x1, y1, x2, y2, score) of the ``person`` class in that image's pixels, best score first."""

    def __init__(self, cfgfile=None, weightfile=None, namesfile=None, score_thresh=0.7, nms_thresh=0.45, use_cuda=True,
                 device=0, max_det=64, seed=0, use_graph=True):
        if not use_cuda or not torch.cuda.is_available():
            raise RuntimeError('YOLOv3 needs a GPU (HIP kernels only; no CPU fallback)')
        self.lib = _lib.load()
        self.device = torch.device('cuda:%d' % device)
        if cfgfile is not None and not os.path.exists(cfgfile):
            raise FileNotFoundError(cfgfile)
        model = Darknet(open(cfgfile).read() if cfgfile is not None else None)
        if weightfile is not None:
            model.load_darknet_weights(weightfile)
            self.weights = weightfile
        else:
            model.init_random(seed)
            self.weights = 'random(seed=%d)' % seed
        model.eval()
        self.size = (model.height, model.width)
        yl = model.yolo_layers()
        if len(yl) != 3:
            raise NotImplementedError('expected three [yolo] heads, cfg has %d' % len(yl))
        self.num_classes = yl[0]['classes']
        self.anchors = np.array([[yl[h]['anchors'][m] for m in yl[h]['mask']] for h in range(3)], dtype=np.float32)   # (3,3,2)
        self.class_id = 0
        if namesfile is not None:
            names = [l.strip() for l in open(namesfile) if l.strip()]
            self.class_id = names.index('person')
        self.score_thresh, self.nms_thresh, self.max_det = float(score_thresh), float(nms_thresh), int(max_det)
        self.net = HipDarknet(model, self.device)
        self.use_graph = use_graph
        self._graphs, self._pool = {}, None
        self._det_ws = {}            # views per call -> workspace of pam_yolo_detect_ws (zeroed once; the kernel leaves its tickets zero)
        self._pinned = {}            # (views, fh, fw) -> two pinned (boxes, count) landing buffers of submit()

    # -- device path ---------------------------------------------------------------------------------------------------------
    def _run(self, ptrs, n, fh, fw, x8, boxes, count):
        st = torch.cuda.current_stream(self.device).cuda_stream
        H, W = self.size
        rc = self.lib.pam_resize_frames(C.c_void_p(st), n, C.c_void_p(ptrs.data_ptr()), fh, fw, H, W, C.c_void_p(x8.data_ptr()))
        if rc != 0:
            raise _lib.PamError('pam_resize_frames failed: %d' % rc)
        heads = self.net.forward(x8)
        hp = (C.c_void_p * 3)(*[C.c_void_p(h.data_ptr()) for h in heads])
        gh = (C.c_int32 * 3)(*[h.shape[2] for h in heads]); gw = (C.c_int32 * 3)(*[h.shape[3] for h in heads])
        cs = (C.c_int32 * 3)(*[h.shape[1] for h in heads])
        an = np.ascontiguousarray(self.anchors.reshape(-1))
        # decode + NMS with the scoring pass over several workgroups per image; the workspace belongs to (this detector, n): replays of one
        # detector run one after the other on its stream
        need = int(self.lib.pam_yolo_detect_workspace_bytes(n, gh, gw))
        ws = self._det_ws.get(n)
        if ws is None or ws.numel() < need:
            ws = self._det_ws[n] = torch.zeros(need, dtype=torch.uint8, device=self.device)
        rc = self.lib.pam_yolo_detect_ws(C.c_void_p(st), n, hp, gh, gw, cs, an.ctypes.data_as(C.c_void_p), W, H, self.num_classes,
                                         self.class_id, self.score_thresh, self.nms_thresh, fw, fh, self.max_det,
                                         C.c_void_p(boxes.data_ptr()), C.c_void_p(count.data_ptr()), C.c_void_p(ws.data_ptr()), need)
        if rc != 0:
            raise _lib.PamError('pam_yolo_detect_ws failed: %d' % rc)
        return heads

    def detect_dev(self, frames):
        """frames: (n, fh, fw, 3) uint8 BGR device tensor (contiguous) -> (boxes (n, max_det, 5) float32, count (2n,) int32),
        both on the device, asynchronous on the current stream."""
        n, fh, fw, _ = frames.shape
        key = (n, fh, fw)
        g = self._graphs.get(key)
        if g is None:
            H, W = self.size
            st_frames = torch.empty_like(frames)
            ptrs = torch.tensor([st_frames[i].data_ptr() for i in range(n)], dtype=torch.int64, device=self.device)
            x8 = torch.empty((n, 8, H, W), dtype=torch.bfloat16, device=self.device, memory_format=torch.channels_last)
            boxes = torch.zeros((n, self.max_det, 5), dtype=torch.float32, device=self.device)
            count = torch.zeros((2 * n,), dtype=torch.int32, device=self.device)
            st_frames.copy_(frames)
            graph = None
            if self.use_graph:
                s = torch.cuda.Stream(self.device)
                s.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(s):
                    self._run(ptrs, n, fh, fw, x8, boxes, count)
                torch.cuda.current_stream(self.device).wait_stream(s)
                graph = _lib.new_graph()
                if self._pool is None:
                    self._pool = torch.cuda.graph_pool_handle()
                with torch.cuda.graph(graph, pool=self._pool):
                    self._run(ptrs, n, fh, fw, x8, boxes, count)
            g = (graph, st_frames, ptrs, x8, boxes, count)
            self._graphs[key] = g
        graph, st_frames, ptrs, x8, boxes, count = g
        if st_frames.data_ptr() != frames.data_ptr():
            st_frames.copy_(frames)
        if graph is not None:
            graph.replay()
        else:
            self._run(ptrs, n, fh, fw, x8, boxes, count)
        return boxes, count

    def frame_buffer(self, n, fh, fw):
        """The graph's own (n, fh, fw, 3) uint8 input when it exists (upload straight into it), else a fresh buffer."""
        g = self._graphs.get((n, fh, fw))
        return g[1] if g is not None else torch.empty((n, fh, fw, 3), dtype=torch.uint8, device=self.device)

    def submit(self, imglist, stream=None):
        """Issue the detection of imglist (BGR uint8 HxWx3 arrays or CUDA tensors) WITHOUT waiting for it: frames into the replay's input,
        one replay per image shape, the boxes' copy to pinned host memory, one event -- all on `stream` (default: the current one).
        -> a ticket for ``collect``.  The reference's loop is detect -> pose -> track (/root/reference/src/testmodel.py:56-63); a driver
        that has the next frame's images early (the loader decodes ahead) submits frame t + 1 here before it runs frame t's pose network,
        and the detector runs under it (ivclabpose.PersonDetectAhead)."""
        single = isinstance(imglist, np.ndarray) and imglist.ndim == 3
        imgs = [imglist] if single else list(imglist)
        by_shape = {}
        for i, im in enumerate(imgs):
            by_shape.setdefault(tuple(im.shape[:2]), []).append(i)
        cur = torch.cuda.current_stream(self.device)
        st = stream if stream is not None else cur
        parts = []
        self._tick = getattr(self, '_tick', 0) + 1
        with torch.cuda.stream(st):
            if st is not cur:
                st.wait_stream(cur)                      # the images were produced / uploaded on the caller's stream
            for (fh, fw), idx in by_shape.items():
                buf = self.frame_buffer(len(idx), fh, fw)
                for k, i in enumerate(idx):
                    im = imgs[i]
                    if torch.is_tensor(im):
                        im.record_stream(st)
                    buf[k].copy_(im if torch.is_tensor(im) else torch.from_numpy(np.ascontiguousarray(im)), non_blocking=True)
                boxes, count = self.detect_dev(buf)
                # two pinned landing buffers per (views, shape), used alternately: a ticket stays readable while the next one is in flight
                pin = self._pinned.setdefault((len(idx), fh, fw), [None, None])
                k2 = self._tick & 1
                if pin[k2] is None:
                    pin[k2] = (torch.empty(boxes.shape, dtype=boxes.dtype).pin_memory(), torch.empty(count.shape, dtype=count.dtype).pin_memory())
                pin[k2][0].copy_(boxes, non_blocking=True); pin[k2][1].copy_(count, non_blocking=True)
                parts.append((idx, pin[k2]))
            ev = torch.cuda.Event()
            ev.record(st)
        return (single, len(imgs), parts, ev)

    def collect(self, ticket):
        """Wait for a ``submit`` and unpack it: one (k, 5) float32 array per image, rows (x1, y1, x2, y2, score), best score first."""
        single, n, parts, ev = ticket
        ev.synchronize()
        results = [None] * n
        for idx, (boxes, count) in parts:
            boxes, count = boxes.numpy(), count.numpy()
            for k, i in enumerate(idx):
                if count[len(idx) + k] > _lib.YOLO_MAX_CAND:    # only the first YOLO_MAX_CAND (coarse head first) entered the NMS
                    warnings.warn('image %d: %d boxes above score_thresh=%g, NMS capacity is %d -- raise the threshold' %
                                  (i, count[len(idx) + k], self.score_thresh, _lib.YOLO_MAX_CAND))
                results[i] = boxes[k, :count[k]].copy()
        return results[0] if single else results

    def __call__(self, imglist):
        return self.collect(self.submit(imglist))
