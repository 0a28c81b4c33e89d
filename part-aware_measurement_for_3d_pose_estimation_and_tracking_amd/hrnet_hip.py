"""HRNet-W48 conv stack on the hand-written MFMA kernels of csrc/pam_conv.hip (no MIOpen in the loop).

The folded (conv + bias) PyTorch module of hrnet.py is walked once into packed weights; ``forward`` then issues one
``pam_conv2d_nhwc_bf16`` per convolution -- bias, residual add and ReLU fused into its epilogue -- plus one
``pam_upsample_add_nhwc_bf16`` per fuse-layer output that has coarser inputs.  Activations are NHWC bf16 torch tensors
(channels-last); torch is used for memory and the stream only.  The whole forward is hipGraph-capturable."""
import contextlib
import ctypes as C

import torch
import torch.nn as nn

from . import _lib


class PackedConv(object):
    def __init__(self, conv, device, pad_cin_to=None, pad_cout_to=None):
        """Zero-padding input channels (pad_cin_to) or output channels (pad_cout_to: zero filters, zero bias) leaves the real
        channels unchanged; the detector uses it for Darknet's 3-, 32- and 255-channel layers."""
        w = conv.weight.detach().float()
        b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0])
        cout, cin, kh, kw = w.shape
        if pad_cin_to is not None and cin < pad_cin_to:
            w = torch.cat([w, torch.zeros(cout, pad_cin_to - cin, kh, kw)], dim=1)
            cin = pad_cin_to
        if pad_cout_to is not None and cout < pad_cout_to:
            w = torch.cat([w, torch.zeros(pad_cout_to - cout, cin, kh, kw)], dim=0)
            b = torch.cat([b, torch.zeros(pad_cout_to - cout)])
            cout = pad_cout_to
        stem = cin == 8 and cout in (32, 64) and kh == 3 and kw == 3 and conv.stride[0] in (1, 2) and conv.padding[0] == 1
        assert cin % 8 == 0 and (cout % 48 == 0 or cout % 64 == 0 or stem), (cin, cout)
        ktot = kh * kw * cin
        kpad = (ktot + 63) // 64 * 64
        wp = torch.zeros((cout, kpad), dtype=torch.float32)
        wp[:, :ktot] = w.permute(0, 2, 3, 1).reshape(cout, ktot)          # k = (ky, kx, cin), cin fastest
        self.w = wp.to(torch.bfloat16).to(device).contiguous()
        self.bias = b.to(device).contiguous()
        self.cin, self.cout, self.kh, self.kw = cin, cout, kh, kw
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        # per-chunk LDS images for k_conv3x3, built lazily per slab width (the kernel picks the slab from the layer's H x W)
        self._w_ohwi = w.permute(0, 2, 3, 1).contiguous() if (kh == 3 and kw == 3 and self.stride == 1 and self.pad == 1 and
                                                              cin in (48, 64, 96, 128, 192, 256, 384, 512)) else None
        self._images = {}
        self._device = device
        # stem convolution (8 -> 32 / 64 channels, 3x3, stride 1 / 2): the MFMA A fragments of k_conv_stem, [n-tile j][ky][lane][8]:
        # lane l holds, for output channel 4*nt*((l & 15) >> 2) + 4*j + (l & 3), the 8 input channels of tap (ky, kx = l >> 4)
        self._stem = None
        if stem:
            nt = cout // 16
            lanes = torch.arange(64)
            q, kx = lanes & 15, lanes >> 4
            frag = torch.zeros((nt, 3, 64, 8), dtype=torch.float32)
            for j in range(nt):
                ch = 4 * nt * (q >> 2) + 4 * j + (q & 3)
                for ky in range(3):
                    sel = kx < 3
                    frag[j, ky, sel] = w[ch[sel], :, ky, kx[sel]]
            self._stem = frag.to(torch.bfloat16).to(device).contiguous()

    @staticmethod
    def merged(convs, device):
        """One convolution computing several same-shaped convolutions of the same input: weights / biases concatenated along
        the output channels, in the given order."""
        c0 = convs[0]
        assert all(c.kernel_size == c0.kernel_size and c.stride == c0.stride and c.padding == c0.padding and
                   c.in_channels == c0.in_channels for c in convs)
        m = nn.Conv2d(c0.in_channels, sum(c.out_channels for c in convs), c0.kernel_size, c0.stride, c0.padding, bias=True)
        with torch.no_grad():
            m.weight.copy_(torch.cat([c.weight.detach().float() for c in convs], 0))
            m.bias.copy_(torch.cat([(c.bias.detach().float() if c.bias is not None else torch.zeros(c.out_channels)) for c in convs]))
        return PackedConv(m, device)

    layout_lib = None           # tools/ab_conv_defs.py: a build variant whose layout functions decide the image (default: the library)

    def image(self, h, w, classic=False, c96_slab=0):
        """Weight image of this layer at input h x w for the rows-in-LDS kernels (layouts: include/pam.h): the classic per-chunk image
        [cout/BN][cin/CK][BN][pitch/2] (row = 9 taps x CK channels + pad), or -- where pam_conv3x3_layout() says so and the caller does
        not force the classic kernel -- the streamed kernel's [cout/BN][cin/32][9][BN][4][8] with swizzled 16-byte pieces."""
        if self._stem is not None:
            return self._stem
        if self._w_ohwi is None:
            return None
        lib = self.layout_lib or _lib.load()
        bn_s = 0 if classic else lib.pam_conv3x3_layout_ex(int(h), int(w), self.cin, self.cout, int(c96_slab))     # > 0: streamed kernel, with this slab width
        streamed = bn_s > 0
        bn = bn_s if streamed else lib.pam_conv3x3_slab(int(h), int(w), self.cin, self.cout)
        if self.cout % bn != 0:
            return None                                  # no whole number of slabs: the generic kernel takes this layer
        self.last_streamed = streamed                    # layout of the image this call returns (conv() states it to the library)
        self.last_c96 = bn if (streamed and self.cin == 96 and self.cout == 96) else 0
        img = self._images.get((bn, streamed))
        if img is None:
            cin, cout = self.cin, self.cout
            # row j*16 + q of a slab holds channel 4*ntw*(q >> 2) + 4*j + (q & 3): with the weights as the MFMA A operand a
            # lane's accumulators are then 4*ntw contiguous output channels (16-byte epilogue accesses, see k_conv3x3)
            ntw = bn // 16
            perm = [4 * ntw * (q >> 2) + 4 * j + (q & 3) for j in range(ntw) for q in range(16)]
            if streamed:
                w5 = self._w_ohwi.reshape(cout // bn, bn, 9, cin // 32, 32)[:, perm]       # [slab][row][tap][chunk][c]
                t = w5.permute(0, 3, 2, 1, 4).reshape(cout // bn, cin // 32, 9, bn, 4, 8)  # [slab][chunk][tap][row][piece][8]
                rows = torch.arange(bn)
                src = torch.arange(4)[None, :] ^ ((rows >> 1) & 2)[:, None]                # physical piece p of row r holds logical piece p ^ ((r >> 1) & 2)
                t = torch.gather(t, 4, src[None, None, None, :, :, None].expand(t.shape))
            else:
                ck = 48 if cin == 48 else (64 if cin >= 192 else 32)
                pitch = {48: 864, 32: 608, 64: 1184}[ck] // 2
                w5 = self._w_ohwi.reshape(cout // bn, bn, 9, cin // ck, ck)[:, perm]       # [slab][co][tap][chunk][c]
                t = torch.zeros((cout // bn, cin // ck, bn, pitch), dtype=torch.float32)
                t[:, :, :, :9 * ck] = w5.permute(0, 3, 1, 2, 4).reshape(cout // bn, cin // ck, bn, 9 * ck)
            img = t.to(torch.bfloat16).to(self._device).contiguous()
            self._images[(bn, streamed)] = img
        return img


def down48_image(op):
    """Weight image of a 3x3 stride-2 convolution with 48 input channels for ``pam_conv3x3s2_c48_nhwc_bf16`` (csrc/pam_down.hip; layout:
    include/pam.h): per 48-channel slab of the output [14 k-steps][48 rows][4 pieces][8] -- K = (tap, cin) flattened and zero-padded to
    14 * 32, rows permuted and 16-byte pieces swizzled exactly as one convolution of PackedBlock's C = 48 image."""
    assert op.cin == 48 and op.kh == 3 and op.kw == 3 and op.cout % 48 == 0
    img = getattr(op, '_down48', None)
    if img is None:
        ns, nstep = op.cout // 48, 14
        rows = torch.arange(48)
        j, qq, r = rows // 16, (rows % 16) >> 2, rows & 3
        chan = torch.where(j < 2, 8 * qq + 4 * j + r, 32 + 4 * qq + r)
        src = torch.arange(4)[None, :] ^ torch.tensor([0, 2, 3, 1])[qq][:, None]
        wk = torch.zeros((op.cout, nstep * 32), dtype=torch.float32)
        wk[:, :432] = op.w[:, :432].float().cpu()                                           # [cout][k = tap * 48 + cin]
        wk = wk.reshape(ns, 48, nstep, 4, 8)[:, chan]                                        # [slab][row][k-step][piece][8]
        wk = torch.gather(wk, 3, src[None, :, None, :, None].expand(ns, 48, nstep, 4, 8)).permute(0, 2, 1, 3, 4)
        img = op._down48 = wk.to(torch.bfloat16).to(op._device).contiguous()
        assert img.numel() * 2 == ns * 43008
    return img


def streamed_image(w_ohwi, bn, device):
    """The streamed 3x3 kernels' weight image (layout: include/pam.h): [cout / bn][cin / 32][9 taps][bn rows][4 pieces][8] -- row j*16 + q
    of a slab = its channel 4*ntw*(q >> 2) + 4*j + (q & 3) (a lane's accumulators are then 4*ntw contiguous output channels), physical
    16-byte piece p of row r = the chunk's input channels 8*(p ^ ((r >> 1) & 2)) .. + 7 (LDS bank swizzle)."""
    cout, _, _, cin = w_ohwi.shape
    ntw = bn // 16
    perm = [4 * ntw * (q >> 2) + 4 * j + (q & 3) for j in range(ntw) for q in range(16)]
    w5 = w_ohwi.reshape(cout // bn, bn, 9, cin // 32, 32)[:, perm]                              # [slab][row][tap][chunk][c]
    t = w5.permute(0, 3, 2, 1, 4).reshape(cout // bn, cin // 32, 9, bn, 4, 8)                   # [slab][chunk][tap][row][piece][8]
    rows = torch.arange(bn)
    src = torch.arange(4)[None, :] ^ ((rows >> 1) & 2)[:, None]
    t = torch.gather(t, 4, src[None, None, None, :, :, None].expand(t.shape))
    return t.to(torch.bfloat16).to(device).contiguous()


class PackedUp(object):
    """The 1x1 convolutions into ONE output of an HR module's fuse layer, packed for ``pam_fuse_sum_nhwc_bf16`` (csrc/pam_fuse.hip):
    per coarser source branch the weights as MFMA A fragments [C / 16][Cs / 32][64 lanes][8] (lane l of fragment (j, ks) holds
    W[16 j + (l & 15)][32 ks + 8 (l >> 4) .. + 7]) and the float32 bias."""

    def __init__(self, convs, shifts, device):
        self.c = convs[0].weight.shape[0]
        self.shifts, self.chans, self.wimg, self.bias = list(shifts), [], [], []
        for cv in convs:
            c, cs = cv.weight.shape[0], cv.weight.shape[1]
            assert c == self.c and cv.weight.shape[2:] == (1, 1) and cs % 32 == 0 and c % 16 == 0
            w = cv.weight.detach().float().reshape(c // 16, 16, cs // 32, 4, 8).permute(0, 2, 3, 1, 4)     # [j][ks][g][q][8]
            self.wimg.append(w.reshape(c // 16, cs // 32, 64, 8).to(torch.bfloat16).to(device).contiguous())
            self.bias.append((cv.bias.detach().float() if cv.bias is not None else torch.zeros(c)).to(device).contiguous())
            self.chans.append(cs)
        n = len(convs)
        self.c_w = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) for t in self.wimg])
        self.c_b = (C.c_void_p * n)(*[C.c_void_p(t.data_ptr()) for t in self.bias])
        self.c_sh = (C.c_int32 * n)(*self.shifts)
        self.c_ch = (C.c_int32 * n)(*self.chans)


class PackedBlock(object):
    """One BasicBlock (conv3x3 -> ReLU -> conv3x3 -> + x -> ReLU) of the 48- or 96-channel branch packed for
    ``pam_basic_block2_nhwc_bf16`` (csrc/pam_block2.hip; layouts: include/pam.h): ONE buffer ``wpack`` =
    [float32 bias of conv1, conv2, padded to 1 KiB][k-step weight images of conv1][... of conv2]."""

    def __init__(self, conv1, conv2, device):
        c = conv1.weight.shape[0]
        assert c in (48, 96) and conv1.weight.shape == (c, c, 3, 3) and conv2.weight.shape == (c, c, 3, 3)
        rows = torch.arange(c)
        zb = lambda cv: cv.bias.detach().float() if cv.bias is not None else torch.zeros(c)
        head = torch.zeros(256, dtype=torch.float32)
        head[:2 * c] = torch.cat([zb(conv1), zb(conv2)])
        if c == 48:
            # k_bblock2_48: K = (tap, cin) flattened, 14 k-steps of 32 (zero tail); a k-step image = [48 rows][4 pieces][8].  Row
            # j * 16 + 4 q' + r = channel 8 q' + 4 j + r for N tiles j = 0, 1 and 32 + 4 q' + r for j = 2 (a lane ends with channels
            # 8 g .. 8 g + 7 and 32 + 4 g .. + 3: aligned 16 + 8 bytes of a pixel); physical piece p of row R holds logical piece
            # p ^ sigma[(R % 16) >> 2], sigma = (0, 2, 3, 1) (LDS bank swizzle)
            nstep = 14
            j, qq, r = rows // 16, (rows % 16) >> 2, rows & 3
            chan = torch.where(j < 2, 8 * qq + 4 * j + r, 32 + 4 * qq + r)
            src = torch.arange(4)[None, :] ^ torch.tensor([0, 2, 3, 1])[qq][:, None]
            imgs = []
            for conv in (conv1, conv2):
                w = conv.weight.detach().float().permute(0, 2, 3, 1).reshape(c, 9 * c)[chan]       # [row][k = tap * C + cin]
                wk = torch.zeros((c, nstep * 32), dtype=torch.float32)
                wk[:, :9 * c] = w
                wk = wk.reshape(c, nstep, 4, 8)                                                    # [row][k-step][piece][8]
                imgs.append(torch.gather(wk, 2, src[:, None, :, None].expand(c, nstep, 4, 8)).permute(1, 0, 2, 3))
            nbytes = 1024 + 2 * nstep * c * 64
        else:
            # k_bblock2_96: k-step images [96 rows][4 pieces][8] in the order (conv, chunk of 32 input channels, tap); row j * 16 + q =
            # output channel 24 * (q >> 2) + 4 * j + (q & 3) (a lane ends with 24 contiguous channels), physical piece p of row r holds
            # the chunk's input channels 8 * (p ^ ((r >> 1) & 2)) .. + 7
            chan = 24 * ((rows % 16) >> 2) + 4 * (rows // 16) + (rows & 3)
            src = torch.arange(4)[None, :] ^ ((rows >> 1) & 2)[:, None]
            imgs = []
            for conv in (conv1, conv2):
                w = conv.weight.detach().float().permute(0, 2, 3, 1)[chan]                         # [row][ky][kx][cin]
                w = w.reshape(c, 9, 3, 4, 8).permute(2, 1, 0, 3, 4)                                # [chunk][tap][row][piece][8]
                imgs.append(torch.gather(w, 3, src[None, None, :, :, None].expand(3, 9, c, 4, 8)))
            nbytes = 1024 + 2 * 27 * c * 64
        self.wpack = torch.cat([head.view(torch.uint8), torch.stack(imgs).to(torch.bfloat16).reshape(-1).view(torch.uint8)]).to(device).contiguous()
        assert self.wpack.numel() == nbytes
        self.c = c


class PackedTail(object):
    """The pointwise tail of a layer1 Bottleneck packed for ``pam_bottleneck_tail_nhwc_bf16`` (csrc/pam_pw.hip; layouts: include/pam.h):
    conv3 (64 -> 256) [+ the first block's 1x1 downsample as a second K chunk] and, optionally, the NEXT block's conv1 (256 -> 64)."""

    def __init__(self, conv3, down, conv1_next, device):
        assert conv3.weight.shape == (256, 64, 1, 1) and (down is None or down.weight.shape == (256, 64, 1, 1))
        assert conv1_next is None or conv1_next.weight.shape == (64, 256, 1, 1)
        zb = lambda cv: cv.bias.detach().float() if cv.bias is not None else torch.zeros(cv.weight.shape[0])
        R = torch.arange(256)
        rem = R % 64
        ch3 = 64 * (R // 64) + 16 * ((rem % 16) >> 2) + 4 * (rem // 16) + (rem & 3)                   # LDS row -> output channel
        q3 = torch.arange(8)[None, :] ^ ((R >> 1) & 7)[:, None]                                    # [row][physical piece] -> logical piece
        srcs = [conv3] + ([down] if down is not None else [])
        img = torch.zeros((len(srcs), 256, 8, 8), dtype=torch.float32)
        for c, cv in enumerate(srcs):
            w = cv.weight.detach().float().reshape(256, 64)[ch3].reshape(256, 8, 8)                # [row][logical piece][8]
            img[c] = torch.gather(w, 1, q3[:, :, None].expand(256, 8, 8))
        self.w3 = img.to(torch.bfloat16).to(device).contiguous()
        self.b3 = (zb(conv3) + (zb(down) if down is not None else 0)).to(device).contiguous()
        self.S = len(srcs)
        self.w1 = self.b1 = None
        if conv1_next is not None:
            R1 = torch.arange(64)
            ch1 = 16 * ((R1 % 16) >> 2) + 4 * (R1 // 16) + (R1 & 3)
            q1 = torch.arange(8)[None, :] ^ ((R1 >> 1) & 7)[:, None]                               # [row][physical piece] -> logical piece q
            w = conv1_next.weight.detach().float().reshape(64, 256)[ch1]                           # [row][input channel]
            im1 = torch.zeros((4, 64, 8, 8), dtype=torch.float32)
            e = torch.arange(8)
            for sl in range(4):
                cin = 64 * sl + 16 * (q1 & 3)[:, :, None] + 8 * (q1 >> 2)[:, :, None] + e[None, None, :]   # [row][piece][8] input channel
                im1[sl] = torch.gather(w, 1, cin.reshape(64, 64)).reshape(64, 8, 8)
            self.w1 = im1.to(torch.bfloat16).to(device).contiguous()
            self.b1 = zb(conv1_next).to(device).contiguous()


class PackedPointwise64(object):
    """A 64 -> 64 1x1 convolution (+ ReLU) packed for ``pam_pointwise64_relu_nhwc_bf16`` (csrc/pam_pw.hip, k_pw1)."""

    def __init__(self, conv, device):
        assert conv.weight.shape == (64, 64, 1, 1)
        R = torch.arange(64)
        ch = 16 * ((R % 16) >> 2) + 4 * (R // 16) + (R & 3)
        q = torch.arange(8)[None, :] ^ ((R >> 1) & 7)[:, None]
        w = conv.weight.detach().float().reshape(64, 64)[ch].reshape(64, 8, 8)
        self.w = torch.gather(w, 1, q[:, :, None].expand(64, 8, 8)).to(torch.bfloat16).to(device).contiguous()
        self.b = (conv.bias.detach().float() if conv.bias is not None else torch.zeros(64)).to(device).contiguous()


def conv64_image(conv, device):
    """[9 taps][64 rows][64 K] bf16 LDS image of a 64 -> 64 3x3 convolution for the fused stem / Bottleneck kernels (layout: include/pam.h):
    row 16 j + q of a tap = output channel 32 (j >> 1) + 8 (q >> 2) + 4 (j & 1) + (q & 3) -- a lane then ends with channels 32 h + 8 g .. + 7,
    the natural K order of the pointwise product that consumes its accumulators --, 16-byte pieces swizzled by (q >> 1) & 7."""
    assert conv.weight.shape == (64, 64, 3, 3)
    R = torch.arange(64)
    j, q = R // 16, R % 16
    ch = 32 * (j >> 1) + 8 * (q >> 2) + 4 * (j & 1) + (q & 3)                                   # image row -> output channel
    c = torch.arange(8)[None, :] ^ ((q >> 1) & 7)[:, None]                                      # [row][physical piece] -> logical piece
    cin = (8 * c[:, :, None] + torch.arange(8)[None, None, :]).reshape(64, 64)                  # [row][physical K position] -> input channel
    w = conv.weight.detach().float()[ch]                                                        # [row][cin][ky][kx]
    img = torch.zeros((9, 64, 64), dtype=torch.float32)
    for ky in range(3):
        for kx in range(3):
            img[ky * 3 + kx] = torch.gather(w[:, :, ky, kx], 1, cin)
    return img.to(torch.bfloat16).to(device).contiguous()


class PackedBneck(object):
    """A layer1 Bottleneck from its 3x3 convolution on, packed for ``pam_bottleneck_fused_nhwc_bf16`` (csrc/pam_bneck.hip): the 3x3's LDS
    image + the pointwise tail's images (PackedTail with one K source)."""

    def __init__(self, conv2, tail, device):
        assert conv2.stride[0] == 1 and conv2.padding[0] == 1
        self.tail = tail
        self.w2 = conv64_image(conv2, device)
        self.b2 = (conv2.bias.detach().float() if conv2.bias is not None else torch.zeros(64)).to(device).contiguous()


class PackedStem(object):
    """HRNet's stem (conv1 8 -> 64 s2, conv2 64 -> 64 s2) and layer1[0].conv1 (64 -> 64 1x1) packed for ``pam_stem_fused_nhwc_bf16``
    (csrc/pam_stem.hip; layouts: include/pam.h): conv1 and the pointwise keep the images of their own kernels, conv2 gets the
    [9 taps][64 rows][64 K] LDS image."""

    def __init__(self, conv1_packed, conv2, pw_packed, device):
        assert conv1_packed._stem is not None and conv1_packed.cout == 64 and conv1_packed.stride == 2
        assert conv2.weight.shape == (64, 64, 3, 3) and conv2.stride[0] == 2 and conv2.padding[0] == 1
        self.c1, self.pw = conv1_packed, pw_packed
        self.w2 = conv64_image(conv2, device)
        self.b2 = (conv2.bias.detach().float() if conv2.bias is not None else torch.zeros(64)).to(device).contiguous()


class ActivationArena(object):
    """Activations of the captured forwards of ONE replay slot: a bump allocator over one device buffer of two halves.  The executor
    calls ``epoch()`` at the start of the stem, of layer1's successor and of every HR module; an epoch allocates from the half the
    epoch before last used -- everything produced two epochs ago is dead by then (a module's tensors are read by that module and by the
    next one's first kernels only, and a full join of the branch streams lies between any two epochs).  Every capture of the slot (one
    per crop-count bucket) replays into the same buffer: they run one after the other, never at the same time.  Round 3 let every
    capture keep its own tensors: 197 GiB after a sweep over the 55 crop-count buckets of the Panoptic workload (tools/graph_memory.py).
    Without a device (``half_bytes`` None) the arena only measures: the largest epoch of a shape-only walk sizes the real one."""

    def __init__(self, device=None, half_bytes=None):
        self.half_bytes = half_bytes
        self.buf = torch.empty(2 * half_bytes, dtype=torch.uint8, device=device) if half_bytes else None
        self.half, self.off, self.peak = 1, 0, 0

    def epoch(self):
        self.half ^= 1
        self.off = 0

    def count(self, nbytes):
        self.off += (nbytes + 255) // 256 * 256
        self.peak = max(self.peak, self.off)

    def alloc(self, n, c, h, w):
        nbytes = 2 * n * c * h * w
        a = self.half * self.half_bytes + self.off
        self.count(nbytes)
        if self.off > self.half_bytes:
            raise _lib.PamError('activation arena too small: epoch needs > %d bytes' % self.half_bytes)
        return self.buf[a:a + nbytes].view(torch.bfloat16).as_strided((n, c, h, w), (h * w * c, 1, w * c, c))


class ConvEngine(object):
    """Kernel launchers shared by the pose network (HipHRNet) and the person detector (yolov3.HipDarknet)."""
    count = None            # set to a dict to tally algorithmic bytes / flops of one forward (bench.py)
    prof = None             # set to a list: every launch appends dict(family, sig, bytes, flops, fn) -- fn re-issues exactly that launch
                            # (bench.py times each distinct one alone for the per-family roofline)

    def _prof_add(self, x, family, sig, nbytes, flops, fn):
        if self.prof is not None and x.device.type == 'cuda':
            self.prof.append(dict(family=family, sig=(family,) + tuple(sig), bytes=nbytes, flops=flops, fn=fn))
    arena = None                # an ActivationArena: outputs are carved from it instead of torch.empty (HRNetPose's captured replays)

    def _new(self, n, c, h, w, device):
        """A fresh (n, c, h, w) channels-last bf16 activation: from the arena when one is set, else from the caching allocator (kept
        alive until the forward has been issued: another stream may still read what a freed block held)."""
        if self.arena is not None and device.type != 'meta':
            return self.arena.alloc(n, c, h, w)
        if self.arena is not None:
            self.arena.count(2 * n * c * h * w)
        y = torch.empty((n, c, h, w), dtype=torch.bfloat16, device=device, memory_format=torch.channels_last)
        if self._keep is not None:
            self._keep.append(y)
        return y

    pw64 = True                 # 64 -> 64 pointwise layers over >= 64 k pixels on the streaming kernel k_pw1 (ReLU or leaky)
    slab32 = False              # round 5: the 192- / 384-channel 3x3 layers with 32-channel slabs (pam_conv3x3_layout_small): for forwards of a few crops
    gen_streamed = True         # Darknet 3x3 layers with Cin 128 / 256 / 512 on k_conv3x3s<.., GEN> (False: the classic k_conv3x3)
    tile_cfg = -1
    c96_slab = 0                # 96 -> 96 3x3 layers: 0 = k_conv3x3, 48 / 96 = the streamed kernel with slabs of that many output channels
    _keep = None
    ACT = {None: 0, False: 0, True: 1, 'linear': 0, 'relu': 1, 'leaky': 2}

    def conv(self, op, x, res=None, relu=False, res_after_act=False, relu_from=0):
        """relu: False/True, or 'linear' | 'relu' | 'leaky' (slope 0.1); res_after_act: out = act(conv + b) + res (Darknet shortcut).
        x may be a channel slice of a wider channels-last tensor; relu_from: the activation applies to channels >= relu_from."""
        n, cin, h, w = x.shape
        if (self.down48 and op.stride == 2 and op.kh == 3 and op.kw == 3 and op.pad == 1 and op.cin == 48 and op.cout % 48 == 0 and
                self.ACT[relu] <= 1 and not res_after_act and relu_from % 8 == 0):
            return self.conv_down48(op, x, res=res, relu=bool(self.ACT[relu]), relu_from=relu_from)
        if (self.down_s and op.stride == 2 and op.kh == 3 and op.kw == 3 and op.pad == 1 and op.cin in (96, 192) and res is None and
                self.ACT[relu] <= 1 and relu_from % 16 == 0 and self.tile_cfg == -1 and
                (x.device.type == 'meta' or self.lib.pam_conv3x3s2_slab(h, w, cin, op.cout) > 0)):
            return self.conv_down_s(op, x, relu=bool(self.ACT[relu]), relu_from=relu_from)
        if (self.pw64 and op.kh == 1 and op.kw == 1 and op.stride == 1 and cin == 64 and op.cout == 64 and res is None and relu_from == 0 and
                self.ACT[relu] in (1, 2) and x.device.type != 'meta' and x.is_contiguous(memory_format=torch.channels_last) and n * h * w >= 65536):
            # a 64 -> 64 pointwise layer over many pixels (Darknet's 64 -> 32, zero-padded, at 208 x 208) is a pure stream: k_pw1
            img = op._images.get('pw64')
            if img is None:
                R = torch.arange(64)
                ch = 16 * ((R % 16) >> 2) + 4 * (R // 16) + (R & 3)
                q = torch.arange(8)[None, :] ^ ((R >> 1) & 7)[:, None]
                w64 = op.w[:, :64].float().cpu()[ch].reshape(64, 8, 8)
                img = op._images['pw64'] = torch.gather(w64, 1, q[:, :, None].expand(64, 8, 8)).to(torch.bfloat16).to(op._device).contiguous()
            y = self._new(n, 64, h, w, x.device)
            if self.count is not None:
                self.count['bytes'] += 2 * (x.numel() + y.numel() + 64 * 64) + 4 * 64; self.count['flops'] += 2 * y.numel() * 64; self.count['launches'] += 1
            rc = self.lib.pam_pointwise64_act_nhwc_bf16(C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), C.c_void_p(x.data_ptr()),
                                                        C.c_void_p(img.data_ptr()), C.c_void_p(op.bias.data_ptr()), C.c_void_p(y.data_ptr()), n * h * w, self.ACT[relu])
            if rc != 0:
                raise _lib.PamError('pam_pointwise64_act_nhwc_bf16 failed (%d)' % rc)
            return y
        in_cs = cin if x.device.type == 'meta' else x.stride(3)          # channels between neighbouring pixels
        assert cin == op.cin and (x.device.type == 'meta' or (x.stride(1) == 1 and x.stride(2) == w * in_cs and x.stride(0) == h * w * in_cs)), (x.shape, op.cin)
        ho = (h + 2 * op.pad - op.kh) // op.stride + 1
        wo = (w + 2 * op.pad - op.kw) // op.stride + 1
        y = self._new(n, op.cout, ho, wo, x.device)
        if self.count is not None:       # unique bytes this conv must move: input + weights + bias [+ residual] + output
            self.count['bytes'] += 2 * (x.numel() + y.numel() + op.cout * op.kh * op.kw * op.cin + (y.numel() if res is not None else 0)) + 4 * op.cout
            self.count['flops'] += 2 * y.numel() * op.kh * op.kw * op.cin
            self.count['launches'] += 1
        if x.device.type == 'meta':
            return y
        act = self.ACT[relu] | (4 if (res_after_act and res is not None) else 0)
        # the streamed kernels (k_conv3x3s / k_conv_gs) take the activation codes 0 / 1 only: leaky / shortcut-after-activation layers
        # (the detector's) ask for the classic kernels and the classic weight image
        tile_cfg = -2 if (self.tile_cfg == -1 and act > 1) else self.tile_cfg
        wimg = None
        if (tile_cfg == -2 and self.gen_streamed and op.kh == 3 and op.kw == 3 and op.stride == 1 and op.pad == 1 and in_cs == cin and relu_from == 0
                and op._w_ohwi is not None and self.lib.pam_conv3x3_layout_gen(h, w, cin, op.cout) > 0):
            # round 5: Darknet's 3x3 layers (leaky, shortcut after the activation) on the streamed kernel's general-activation instantiations
            bn = self.lib.pam_conv3x3_layout_gen(h, w, cin, op.cout)
            wimg = op._images.get(('gen', bn))
            if wimg is None:
                wimg = op._images[('gen', bn)] = streamed_image(op._w_ohwi, bn, op._device)
            tile_cfg = -7
        if (wimg is None and self.slab32 and tile_cfg == -1 and op.kh == 3 and op.kw == 3 and op.stride == 1 and op.pad == 1 and cin in (192, 384) and
                in_cs == cin and relu_from == 0 and op._w_ohwi is not None and self.lib.pam_conv3x3_layout_small(h, w, cin, op.cout) > 0):
            # forwards of a few crops: the deep branches' layers with 32-channel slabs (twice the workgroups, each half as long; bit-identical)
            wimg = op._images.get(('s32', 32))
            if wimg is None:
                wimg = op._images[('s32', 32)] = streamed_image(op._w_ohwi, 32, op._device)
            tile_cfg = -8
        if wimg is None:
            wimg = op.image(h, w, classic=(tile_cfg != -1), c96_slab=self.c96_slab) if (in_cs == cin and relu_from == 0) else None
        if tile_cfg == -1 and wimg is not None and op._stem is None:
            # automatic choice, but the layout of THIS image is stated: -3 streamed / -4 classic, -5 / -6 a 96 -> 96 layer streamed with
            # slabs of 48 / 96 output channels (the executor's choice, c96_slab)
            tile_cfg = ({48: -5, 96: -6}.get(getattr(op, 'last_c96', 0), -3)) if getattr(op, 'last_streamed', False) else -4
        st = torch.cuda.current_stream(x.device).cuda_stream
        launch = lambda: self.lib.pam_conv2d_nhwc_bf16_ex(
            C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), C.c_void_p(x.data_ptr()), C.c_void_p(op.w.data_ptr()),
            C.c_void_p(wimg.data_ptr()) if wimg is not None else None,
            C.c_void_p(op.bias.data_ptr()), C.c_void_p(res.data_ptr()) if res is not None else None,
            C.c_void_p(y.data_ptr()), n, h, w, op.cin, op.cout, op.kh, op.kw, op.stride, op.pad, act, tile_cfg, in_cs, relu_from)
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_conv2d_nhwc_bf16 failed (%d) for %s' % (rc, (x.shape, op.cout, op.kh, op.stride)))
        if self.prof is not None:
            kind = self.lib.pam_conv_last_kernel()
            if kind == 4:
                fam = 'k_conv_stem %d->%d' % (3, op.cout)
            elif kind in (1, 2):
                fam = '%s C=%d %dx%d' % ('k_conv3x3s' if kind == 2 else 'k_conv3x3', op.cin, h, w)
            else:
                fam = '%s %dx%d stride %d' % ('k_conv_gs' if kind == 3 else 'k_conv_igemm', op.kh, op.kw, op.stride)
            self._prof_add(x, fam, (n, h, w, op.cin, op.cout, res is not None, in_cs, relu_from),
                           2 * (x.numel() + y.numel() + op.cout * op.kh * op.kw * op.cin + (y.numel() if res is not None else 0)) + 4 * op.cout,
                           2 * y.numel() * op.kh * op.kw * op.cin, launch)
        return y

    def basic_block2(self, op, x, tile=None):
        """One BasicBlock (PackedBlock with ``wpack``: C = 48) on x through the resident-weights kernel; tile = (rows, cols) or None."""
        n, c, h, w = x.shape
        assert c == op.c, (x.shape, op.c)
        y = self._new(n, c, h, w, x.device)
        nbytes, flops = 2 * (2 * x.numel() + 2 * 9 * c * c) + 8 * c, 2 * 2 * x.numel() * 9 * c
        if self.count is not None:
            self.count['bytes'] += nbytes; self.count['flops'] += flops; self.count['launches'] += 1
        if x.device.type == 'meta':
            return y
        assert x.is_contiguous(memory_format=torch.channels_last)
        if tile is None and c == 96 and self.b96_tile:
            tile = tuple(self.b96_tile)
        if tile is None and c == 48 and self.b48_tile:
            tile = tuple(self.b48_tile)
        if tile is None:
            tile = self._bb2_tiles.get((c, n, h, w))
            if tile is None:
                t2 = (C.c_int32 * 2)()
                if self.lib.pam_basic_block2_tile(c, n, h, w, t2) != 0:
                    raise _lib.PamError('no resident-weights block tile for %s' % (tuple(x.shape),))
                tile = self._bb2_tiles[(c, n, h, w)] = (int(t2[0]), int(t2[1]))
        launch = lambda: self.lib.pam_basic_block2_nhwc_bf16(C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), C.c_void_p(x.data_ptr()),
                                                              C.c_void_p(op.wpack.data_ptr()), C.c_void_p(y.data_ptr()), n, h, w, c, tile[0], tile[1])
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_basic_block2_nhwc_bf16 failed (%d) for %s tile %s' % (rc, tuple(x.shape), tile))
        self._prof_add(x, 'k_bblock2 C=%d' % c, (n, h, w, c) + tuple(tile), nbytes, flops, launch)
        return y

    _bb2_tiles = {}             # (C, N, H, W) -> the library's tile choice (pam_basic_block2_tile searches ~H x W candidates)
    b48_tile = None             # the same for the 48-channel block
    b96_tile = None             # (rows, cols) of the 96-channel fused block's items instead of the library's choice (tuning)

    def pointwise64(self, op, x):
        """ReLU(conv1x1 64 -> 64 (x)) as a pure stream (k_pw1)."""
        n, c, h, w = x.shape
        assert c == 64
        y = self._new(n, c, h, w, x.device)
        nbytes, flops = 2 * (2 * x.numel() + 64 * 64) + 4 * 64, 2 * n * h * w * 64 * 64
        if self.count is not None:
            self.count['bytes'] += nbytes; self.count['flops'] += flops; self.count['launches'] += 1
        if x.device.type == 'meta':
            return y
        assert x.is_contiguous(memory_format=torch.channels_last)
        launch = lambda: self.lib.pam_pointwise64_relu_nhwc_bf16(C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), C.c_void_p(x.data_ptr()),
                                                                  C.c_void_p(op.w.data_ptr()), C.c_void_p(op.b.data_ptr()), C.c_void_p(y.data_ptr()), n * h * w)
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_pointwise64_relu_nhwc_bf16 failed (%d)' % rc)
        self._prof_add(x, 'k_pw1 64->64 pointwise', (n, h, w), nbytes, flops, launch)
        return y

    def stem_fused(self, op, x8):
        """(x0, y1) = stem + the first Bottleneck's conv1 in one launch (k_stem_fused): bit-identical to conv(conv1), conv(conv2), pointwise64."""
        n, c, h, w = x8.shape
        assert c == 8
        h2, w2 = ((h - 1) // 2 + 1 - 1) // 2 + 1, ((w - 1) // 2 + 1 - 1) // 2 + 1
        x0 = self._new(n, 64, h2, w2, x8.device)
        y1 = self._new(n, 64, h2, w2, x8.device)
        h1, w1 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        nbytes = 2 * (x8.numel() + x0.numel() + y1.numel() + 64 * 72 + 64 * 576 + 64 * 64) + 4 * 192
        flops = 2 * n * (h1 * w1 * 64 * 72 + h2 * w2 * 64 * (576 + 64))
        if self.count is not None:
            self.count['bytes'] += nbytes; self.count['flops'] += flops; self.count['launches'] += 1
        if x8.device.type == 'meta':
            return x0, y1
        assert x8.is_contiguous(memory_format=torch.channels_last)
        launch = lambda: self.lib.pam_stem_fused_nhwc_bf16(
            C.c_void_p(torch.cuda.current_stream(x8.device).cuda_stream), C.c_void_p(x8.data_ptr()), C.c_void_p(op.c1._stem.data_ptr()),
            C.c_void_p(op.c1.bias.data_ptr()), C.c_void_p(op.w2.data_ptr()), C.c_void_p(op.b2.data_ptr()), C.c_void_p(op.pw.w.data_ptr()),
            C.c_void_p(op.pw.b.data_ptr()), C.c_void_p(x0.data_ptr()), C.c_void_p(y1.data_ptr()), n, h, w)
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_stem_fused_nhwc_bf16 failed (%d) for %s' % (rc, tuple(x8.shape)))
        self._prof_add(x8, 'k_stem_fused stem + conv1 of layer1', (n, h, w), nbytes, flops, launch)
        return x0, y1

    def bottleneck_fused(self, op, y1, res=None, x0=None):
        """(X, y1' or None) = conv3x3 + pointwise tail of a layer1 Bottleneck in one launch (k_bneck): bit-identical to conv(c2) + bottleneck_tail.
        res: the block input (blocks 1-3); x0: the first block's 64-channel input (its downsample convolution is part of the tail)."""
        n, c, h, w = y1.shape
        t = op.tail
        assert c == 64 and (res is None) != (x0 is None) and (x0 is None) == (t.S == 1)
        assert res is None or tuple(res.shape) == (n, 256, h, w)
        X = self._new(n, 256, h, w, y1.device)
        Y = self._new(n, 64, h, w, y1.device) if t.w1 is not None else None
        M = n * h * w
        side = res if res is not None else x0
        nbytes = 2 * (y1.numel() + side.numel() + X.numel() + (Y.numel() if Y is not None else 0) + 64 * 576 + t.S * 256 * 64 +
                      (64 * 256 if t.w1 is not None else 0)) + 4 * (64 + 256 + (64 if t.w1 is not None else 0))
        flops = 2 * M * (64 * 576 + t.S * 64 * 256 + (256 * 64 if t.w1 is not None else 0))
        if self.count is not None:
            self.count['bytes'] += nbytes; self.count['flops'] += flops; self.count['launches'] += 1
        if y1.device.type == 'meta':
            return X, Y
        for q in (y1, side):
            assert q.is_contiguous(memory_format=torch.channels_last)
        launch = lambda: self.lib.pam_bottleneck_fused_nhwc_bf16(
            C.c_void_p(torch.cuda.current_stream(y1.device).cuda_stream), C.c_void_p(y1.data_ptr()),
            C.c_void_p(x0.data_ptr()) if x0 is not None else None, C.c_void_p(res.data_ptr()) if res is not None else None,
            C.c_void_p(op.w2.data_ptr()), C.c_void_p(op.b2.data_ptr()), C.c_void_p(t.w3.data_ptr()), C.c_void_p(t.b3.data_ptr()),
            C.c_void_p(t.w1.data_ptr()) if t.w1 is not None else None, C.c_void_p(t.b1.data_ptr()) if t.w1 is not None else None,
            C.c_void_p(X.data_ptr()), C.c_void_p(Y.data_ptr()) if Y is not None else None, n, h, w)
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_bottleneck_fused_nhwc_bf16 failed (%d) for %s' % (rc, tuple(y1.shape)))
        self._prof_add(y1, 'k_bneck 3x3 + bottleneck tail', (n, h, w, t.S, t.w1 is not None), nbytes, flops, launch)
        return X, Y

    def bottleneck_tail(self, op, y2, x0=None, res=None, tile_cfg=0):
        """X = ReLU(conv3(y2) [+ downsample(x0)] [+ res]); y1 = ReLU(conv1_next(X)) in one launch -> (X, y1 or None)."""
        n, c, h, w = y2.shape
        assert c == 64 and (x0 is None) == (op.S == 1), (y2.shape, op.S)
        X = self._new(n, 256, h, w, y2.device)
        Y = self._new(n, 64, h, w, y2.device) if op.w1 is not None else None
        M = n * h * w
        nbytes = 2 * (y2.numel() + (x0.numel() if x0 is not None else 0) + (res.numel() if res is not None else 0) + X.numel() +
                      (Y.numel() if Y is not None else 0) + op.S * 256 * 64 + (64 * 256 if op.w1 is not None else 0)) + 4 * (256 + (64 if op.w1 is not None else 0))
        flops = 2 * M * (op.S * 64 * 256 + (256 * 64 if op.w1 is not None else 0))
        if self.count is not None:
            self.count['bytes'] += nbytes; self.count['flops'] += flops; self.count['launches'] += 1
        if y2.device.type == 'meta':
            return X, Y
        for t in (y2, x0, res):
            assert t is None or t.is_contiguous(memory_format=torch.channels_last)
        launch = lambda: self.lib.pam_bottleneck_tail_nhwc_bf16(
            C.c_void_p(torch.cuda.current_stream(y2.device).cuda_stream), C.c_void_p(y2.data_ptr()),
            C.c_void_p(x0.data_ptr()) if x0 is not None else None, C.c_void_p(res.data_ptr()) if res is not None else None,
            C.c_void_p(op.w3.data_ptr()), C.c_void_p(op.b3.data_ptr()),
            C.c_void_p(op.w1.data_ptr()) if op.w1 is not None else None, C.c_void_p(op.b1.data_ptr()) if op.w1 is not None else None,
            C.c_void_p(X.data_ptr()), C.c_void_p(Y.data_ptr()) if Y is not None else None, M, tile_cfg)
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_bottleneck_tail_nhwc_bf16 failed (%d) for %s' % (rc, tuple(y2.shape)))
        self._prof_add(y2, 'k_pw2 bottleneck tail', (n, h, w, op.S, res is not None, op.w1 is not None), nbytes, flops, launch)
        return X, Y

    def upsample_add(self, base, terms, shifts, relu):
        n, c, h, w = base.shape
        y = self._new(n, c, h, w, base.device)
        if self.count is not None:
            self.count['bytes'] += 2 * (2 * base.numel() + sum(t.numel() for t in terms))
            self.count['launches'] += 1
        if base.device.type == 'meta':
            return y
        st = torch.cuda.current_stream(base.device).cuda_stream
        ptrs = (C.c_void_p * 3)(*[C.c_void_p(t.data_ptr()) for t in terms] + [None] * (3 - len(terms)))
        sh = (C.c_int32 * 3)(*(list(shifts) + [0] * (3 - len(shifts))))
        cs = (C.c_int32 * 3)(*([t.stride(3) for t in terms] + [0] * (3 - len(terms))))       # terms may be channel slices
        launch = lambda: self.lib.pam_upsample_add_nhwc_bf16_ex(C.c_void_p(torch.cuda.current_stream(base.device).cuda_stream),
                                                                 C.c_void_p(base.data_ptr()), len(terms), ptrs, sh, cs,
                                                                 C.c_void_p(y.data_ptr()), n, h, w, c, 1 if relu else 0)
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_upsample_add_nhwc_bf16 failed (%d)' % rc)
        self._prof_add(base, 'k_upsample_add', (n, h, w, c, len(terms)), 2 * (2 * base.numel() + sum(t.numel() for t in terms)), 0, launch)
        return y

    def fuse_sum(self, op, base, plain, srcs, relu=True, tile=(0, 0), max_wg=0):
        """One output of an HR module's fuse layer in one launch (k_fuse_sum): relu(base + sum plain + sum up(conv1x1(src))).
        op: PackedUp; plain: tensors of base's shape (channel slices allowed); srcs: the coarser branches' tensors in op's order."""
        n, c, h, w = base.shape
        assert c == op.c and len(srcs) == len(op.shifts) and len(plain) <= 2
        y = self._new(n, c, h, w, base.device)
        nbytes = 2 * (2 * base.numel() + sum(t.numel() for t in plain) + sum(t.numel() for t in srcs) + sum(c * cs for cs in op.chans)) + 4 * c * len(srcs)
        flops = sum(2 * t.shape[0] * t.shape[2] * t.shape[3] * t.shape[1] * c for t in srcs)
        if self.count is not None:
            self.count['bytes'] += nbytes; self.count['flops'] += flops; self.count['launches'] += 1
        if base.device.type == 'meta':
            return y
        for t, sh, cs in zip(srcs, op.shifts, op.chans):
            assert tuple(t.shape) == (n, cs, h >> sh, w >> sh) and t.is_contiguous(memory_format=torch.channels_last), (t.shape, base.shape, sh)
        pp = (C.c_void_p * 2)(*[C.c_void_p(t.data_ptr()) for t in plain] + [None] * (2 - len(plain)))
        pcs = (C.c_int32 * 2)(*([t.stride(3) for t in plain] + [0] * (2 - len(plain))))
        sp = (C.c_void_p * len(srcs))(*[C.c_void_p(t.data_ptr()) for t in srcs])
        launch = lambda: self.lib.pam_fuse_sum_nhwc_bf16(
            C.c_void_p(torch.cuda.current_stream(base.device).cuda_stream), C.c_void_p(base.data_ptr()), len(plain), pp, pcs, len(srcs), sp,
            op.c_sh, op.c_ch, op.c_w, op.c_b, C.c_void_p(y.data_ptr()), n, h, w, c, 1 if relu else 0, int(tile[0]), int(tile[1]), int(max_wg))
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_fuse_sum_nhwc_bf16 failed (%d) for %s' % (rc, tuple(base.shape)))
        self._prof_add(base, 'k_fuse_sum', (n, h, w, c, len(plain), len(srcs)), nbytes, flops, launch)
        return y

    down_s = True               # 3x3 stride-2 layers with 96 / 192 input channels on k_down_s (csrc/pam_down.hip); False: the generic kernels

    def conv_down_s(self, op, x, relu=False, relu_from=0):
        """3x3 stride-2 convolution of a 96- / 192-channel input (channel slices allowed) through the streamed stride-2 kernel."""
        n, cin, h, w = x.shape
        in_cs = cin if x.device.type == 'meta' else x.stride(3)
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = self._new(n, op.cout, ho, wo, x.device)
        nbytes = 2 * (x.numel() + y.numel() + op.cout * 9 * cin) + 4 * op.cout
        flops = 2 * y.numel() * 9 * cin
        if self.count is not None:
            self.count['bytes'] += nbytes; self.count['flops'] += flops; self.count['launches'] += 1
        if x.device.type == 'meta':
            return y
        assert x.stride(1) == 1 and x.stride(2) == w * in_cs and x.stride(0) == h * w * in_cs, (x.shape, x.stride())
        bn = self.lib.pam_conv3x3s2_slab(h, w, cin, op.cout)
        img = op._images.get(('s2', bn))
        if img is None:
            w_ohwi = op.w[:, :9 * cin].float().cpu().reshape(op.cout, 3, 3, cin)               # op.w: [cout][k = (ky, kx, cin)]
            img = op._images[('s2', bn)] = streamed_image(w_ohwi, bn, op._device)
        launch = lambda: self.lib.pam_conv3x3s2_nhwc_bf16(
            C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), C.c_void_p(x.data_ptr()), in_cs, C.c_void_p(img.data_ptr()),
            C.c_void_p(op.bias.data_ptr()), C.c_void_p(y.data_ptr()), n, h, w, cin, op.cout, 1 if relu else 0, int(relu_from))
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_conv3x3s2_nhwc_bf16 failed (%d) for %s -> %d' % (rc, tuple(x.shape), op.cout))
        self._prof_add(x, 'k_down_s 3x3 stride 2 C=%d %dx%d' % (cin, h, w), (n, h, w, op.cout, in_cs, relu_from), nbytes, flops, launch)
        return y

    down48 = True               # 3x3 stride-2 layers with 48 input channels on k_down48 (csrc/pam_down.hip); False: the generic kernels
    d48_tile = None             # (rows, cols, slab groups) instead of the library's choice (tuning)

    def conv_down48(self, op, x, res=None, relu=False, relu_from=0):
        """3x3 stride-2 convolution of a 48-channel input (a channel slice of a wider tensor is fine) through k_down48."""
        n, cin, h, w = x.shape
        assert cin == 48 and op.cin == 48 and op.stride == 2 and op.kh == 3 and op.pad == 1
        in_cs = cin if x.device.type == 'meta' else x.stride(3)
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = self._new(n, op.cout, ho, wo, x.device)
        nbytes = 2 * (x.numel() + y.numel() + op.cout * 9 * 48 + (y.numel() if res is not None else 0)) + 4 * op.cout
        flops = 2 * y.numel() * 9 * 48
        if self.count is not None:
            self.count['bytes'] += nbytes; self.count['flops'] += flops; self.count['launches'] += 1
        if x.device.type == 'meta':
            return y
        assert x.stride(1) == 1 and x.stride(2) == w * in_cs and x.stride(0) == h * w * in_cs, (x.shape, x.stride())
        img = down48_image(op)
        t = self.d48_tile or (0, 0, 0)
        launch = lambda: self.lib.pam_conv3x3s2_c48_nhwc_bf16(
            C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), C.c_void_p(x.data_ptr()), in_cs, C.c_void_p(img.data_ptr()),
            C.c_void_p(op.bias.data_ptr()), C.c_void_p(res.data_ptr()) if res is not None else None, op.cout if res is not None else 0,
            C.c_void_p(y.data_ptr()), op.cout, n, h, w, op.cout, 1 if relu else 0, int(relu_from), int(t[0]), int(t[1]), int(t[2]))
        rc = launch()
        if rc != 0:
            raise _lib.PamError('pam_conv3x3s2_c48_nhwc_bf16 failed (%d) for %s -> %d' % (rc, tuple(x.shape), op.cout))
        self._prof_add(x, 'k_down48 3x3 stride 2 C=48 %dx%d' % (h, w), (n, h, w, op.cout, res is not None, in_cs, relu_from), nbytes, flops, launch)
        return y

    def upsample_concat(self, a, b):
        """Darknet upsample(x2) + route: concat(nearest_up2(a), b) along channels."""
        n, ca, h2, w2 = a.shape
        _, cb, h, w = b.shape
        assert h == 2 * h2 and w == 2 * w2 and b.shape[0] == n, (a.shape, b.shape)
        y = self._new(n, ca + cb, h, w, a.device)
        if self.count is not None:
            self.count['bytes'] += 2 * (a.numel() + b.numel() + y.numel())
            self.count['launches'] += 1
        if a.device.type == 'meta':
            return y
        st = torch.cuda.current_stream(a.device).cuda_stream
        rc = self.lib.pam_upsample_concat_nhwc_bf16(C.c_void_p(st), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                                                    C.c_void_p(y.data_ptr()), n, h, w, ca, cb)
        if rc != 0:
            raise _lib.PamError('pam_upsample_concat_nhwc_bf16 failed (%d)' % rc)
        return y


class HipHRNet(ConvEngine):
    multi_stream = False
    merge_fuse = True           # merged first-level fuse convolutions (False: one launch per convolution)
    merge_up = True

    def __init__(self, folded_model, device):
        self.lib = _lib.load()
        self.device = device
        self._pack(folded_model, device)
        self.tile_cfg = -1
        # concurrency: the 2-4 branches of an HR module run on side streams (the coarse branches do not fill the chip)
        # equal priorities: a high-priority stream for the deep branches (whose short kernels wait for CUs behind the fused blocks' long items)
        # doubles the forward -- round 4: 4.2-4.6 ms vs 2.31 ms with side streams 2 / 2 + 3 at priority -1; all three: 2.65 ms
        self.side = [torch.cuda.Stream(device) for _ in range(3)]
        self.multi_stream = True
        self.count = None            # set to a dict to tally algorithmic bytes / flops of one forward (bench.py)

    def _pack(self, folded_model, device):
        m = folded_model
        P = lambda c, **kw: PackedConv(c, device, **kw)
        self.conv1 = P(m.conv1, pad_cin_to=8)
        self.conv2 = P(m.conv2)
        self.layer1 = [dict(c1=P(b.conv1), c2=P(b.conv2), c3=P(b.conv3),
                            down=P(b.downsample[0]) if b.downsample is not None else None) for b in m.layer1]
        # the same blocks for the fused pointwise tail: tail b = conv3_b [+ downsample_0] + residual + ReLU, then conv1_{b+1} + ReLU
        l1 = list(m.layer1)
        self.pw0 = PackedPointwise64(l1[0].conv1, device)
        self.stem = PackedStem(self.conv1, m.conv2, self.pw0, device)
        self.tails = [PackedTail(b.conv3, b.downsample[0] if b.downsample is not None else None,
                                 l1[i + 1].conv1 if i + 1 < len(l1) else None, device) for i, b in enumerate(l1)]
        self.bnecks = [PackedBneck(b.conv2, self.tails[i], device) for i, b in enumerate(l1)]
        self.t1 = [P(m.transition1[0][0]), P(m.transition1[1][0][0])]
        self.t2 = P(m.transition2[2][0][0])
        self.t3 = P(m.transition3[3][0][0])
        self.stage2 = [self._module(x) for x in m.stage2]
        self.stage3 = [self._module(x) for x in m.stage3]
        self.stage4 = [self._module(x) for x in m.stage4]

    def _module(self, hm):
        P = lambda c: PackedConv(c, self.device)
        branches = [[(P(b.conv1), P(b.conv2)) for b in br] for br in hm.branches]
        # the same blocks packed for the fused kernels (48- and 96-channel branches: one launch per block)
        fused = [[PackedBlock(b.conv1, b.conv2, self.device) for b in br] if br[0].conv1.out_channels in (48, 96) else None for br in hm.branches]
        fuse = []
        for i, row in enumerate(hm.fuse_layers):
            r = []
            for j, f in enumerate(row):
                if f is None:
                    r.append(None)
                elif j > i:
                    r.append(('up', P(f[0]), j - i))
                else:
                    r.append(('down', [P(step[0]) for step in f]))
            fuse.append(r)
        # the first strided conv of every down chain that starts from branch j reads the same tensor: ONE merged launch per source
        # branch (final convs of 1-conv chains first = no ReLU, then the intermediates = ReLU), the chains continue from channel slices
        merged = {}
        for j in range(len(hm.branches)):
            heads = [(i, hm.fuse_layers[i][j]) for i in range(len(hm.fuse_layers)) if i > j and hm.fuse_layers[i][j] is not None]
            if len(heads) >= 2:
                heads.sort(key=lambda t: (len(t[1]) > 1, t[0]))
                op = PackedConv.merged([f[0][0] for _, f in heads], self.device)
                parts, off = [], 0
                for i, f in heads:
                    parts.append((i, off, f[0][0].out_channels, len(f) == 1)); off += f[0][0].out_channels
                relu_from = sum(c for _, _, c, final in parts if final)
                if relu_from % 16 == 0:
                    merged[j] = dict(op=op, parts=parts, relu_from=relu_from)
        # likewise the 1x1 up-convolutions from branch j to every finer output i < j (all linear): one launch, the sums read slices
        merged_up = {}
        for j in range(len(hm.branches)):
            ups = [(i, hm.fuse_layers[i][j]) for i in range(len(hm.fuse_layers)) if i < j and hm.fuse_layers[i][j] is not None]
            if len(ups) >= 2:
                op = PackedConv.merged([f[0] for _, f in ups], self.device)
                parts, off = [], 0
                for i, f in ups:
                    parts.append((i, off, f[0].out_channels, j - i)); off += f[0].out_channels
                merged_up[j] = dict(op=op, parts=parts)
        # and the same 1x1 convolutions grouped by OUTPUT for the fused sum (k_fuse_sum: the products never reach HBM)
        fsum = []
        for i, row in enumerate(hm.fuse_layers):
            ups = [(j, row[j][0]) for j in range(len(row)) if j > i and row[j] is not None]
            fsum.append(dict(op=PackedUp([f for _, f in ups], [j - i for j, _ in ups], self.device), srcs=[j for j, _ in ups]) if ups else None)
        return dict(branches=branches, fused=fused, fuse=fuse, merged=merged, merged_up=merged_up, fsum=fsum)

    # -- network ------------------------------------------------------------------------------------------------------
    # Stream plan (stream 0 = the caller's stream; hipGraph-capturable -- pairwise event dependencies between the branch streams
    # crashed capture on ROCm 7.2, so all cross-stream ordering is a join/fork through stream 0):
    #   * stream b runs branch b of every HR module AND the fuse chains that start from branch b's output (strided-conv chains
    #     down, 1x1 convs up);
    #   * ONE join per module (a cross-stream join costs ~10 us of idle chip), then the sum of output i (one k_upsample_add over
    #     all its terms) runs on stream i, where branch i of the next module continues without further sync;
    #   * a join before every stage (the new branch's transition conv reads another stream's sum) and at the end.
    # Every tensor of a forward is kept alive until the forward has been issued (self._keep), so the caching allocator can never
    # hand a block that another stream still reads to a new tensor.
    order = (0, 1, 2, 3)        # issue order of the branches inside a module (six orders measured within 1 %)

    def _stream(self, b):
        """stream of branch b: the caller's for branch 0, side stream b - 1 otherwise"""
        return None if (b == 0 or not self.multi_stream) else self.side[b - 1]

    def _barrier(self):
        """Join and re-fork all branch streams through the caller's stream."""
        if self.multi_stream:
            cur = torch.cuda.current_stream(self.device)
            for st in self.side:
                cur.wait_stream(st)
            for st in self.side:
                st.wait_stream(cur)

    # Executor configurations the replay autotuner chooses between per crop count (HRNetPose(autotune=True)).  Round 4, interleaved A/B on
    # one device (tools/ab_crops.sh; vs round 3's grouped ring-kernel blocks, which are gone): 12 crops resident48_streamed96 -9.2 %,
    # fused48_fused96 -8.2 %; 20 crops -3.3 / -7.3 %; 40 crops -4.1 / -8.9 %; 60 crops -5.7 / -11.2 %; 112 crops -1.7 / -7.6 %; 217 crops
    # +0.3 / -6.9 %.
    CONFIGS = {
        'fused48_fused96': dict(block2=3, c96_slab=48, fused_sums=False, slab32=False),          # both fine branches: ONE fused-BasicBlock launch per block (csrc/pam_block2.hip)
        'resident48_streamed96': dict(block2=1, c96_slab=48, fused_sums=False, slab32=False),    # 48-channel branch fused, 96-channel branch as two streamed convolutions per block
        # round 5: up to 6 crops the fuse layers' 1x1 products run inside the sum launches (k_fuse_sum: 203 launches): a forward that small
        # is a chain of launch latencies (interleaved A/B: 2 crops -1.7 %, 4 -1.9 %, 6 -1.9 %, 8 -0.2 %, 12 +0.9 %, 20 +0.3 ... +1.5 %)
        'fused48_fused96_fsum': dict(block2=3, c96_slab=48, fused_sums=True, slab32=False),
        # round 5: up to 12 crops a launch of the deep branches is as long as ONE workgroup -> their 3x3 layers with 32-channel slabs (twice the
        # workgroups, each half the MFMAs and weight bytes; bit-identical): 2 crops -8 %, 4 -10 %, 6 -8 %, 9 -3 ... -6 %, 12 -2 %, 14 0 %, 16 +2 %
        'fused48_fused96_fsum_s32': dict(block2=3, c96_slab=48, fused_sums=True, slab32=True),
    }

    def apply_config(self, name):
        for k, v in self.CONFIGS[name].items():
            setattr(self, k, v)
        self.config_name = name

    config_name = 'fused48_fused96'
    block2 = 3                  # bit 0: 48-channel branch as one resident-weights fused BasicBlock launch per block (k_bblock2_48), bit 1: the
                                # 96-channel branch on the streamed-weights fused block (k_bblock2_96) -- csrc/pam_block2.hip
    c96_slab = 48               # 96 -> 96 layers that are NOT fused: k_conv3x3s with 48-channel slabs (0 = k_conv3x3)
    stamp = None                # diagnostics (tools/fwd_stamps.py): callable(tag) issued on the current stream at points of the schedule
    fs_cap = (0, 0, 0, 0)       # workgroups of output i's fused sum at most (0 = one per CU): the sums of a module's outputs run side by side
    fused_sums = False          # round 5: True = the fuse layers' 1x1 up-convolutions inside the sum launch (k_fuse_sum: 203 launches instead of 221,
                                # bit-identical); False = one merged 1x1 launch per source branch + k_upsample_add.  Measured at 20 crops,
                                # interleaved A/B: -1.5 ... +0.3 % per forward -- the 1x1 launches ran beside the other branches' blocks and
                                # were hidden, while the sums sit behind the module's join on the critical path and a persistent
                                # one-workgroup-per-CU kernel (the weights take 64-147 KB of LDS) streams slower than k_upsample_add's
                                # many small workgroups (20.9 vs 7.9 us for output 0 of a stage-4 module).  Off by default.
    knock_conv2 = 0             # diagnostics: 1 = the second convolution of every un-fused BasicBlock is not issued
    knock_up = 0                # diagnostics: 1 = the coarsest branch's merged 1x1 up-convolution is not issued (its output stays uninitialised)
    knock_out = 0               # diagnostics: bit b = skip the BasicBlocks of branch b (what a free branch would be worth: tools/ab_flags.py)

    def _st(self, tag):
        if self.stamp is not None:
            self.stamp(tag)

    # -- device-side ordering of the branch streams (round 5, csrc/pam_sync.hip) ----------------------------------------------------------
    # A module ends with a full exchange (every sum reads every branch).  As stream events in a captured hipGraph that join costs 15-21 us
    # of idle chip per module (two cross-queue hops through the caller's stream); with flags the branch streams are independent chains that
    # meet through a counter in device memory: a branch's chain ends with a signal launch, a one-wave gate launch in front of every sum polls
    # the module's counter.  HRNetPose switches this on around a capture (flags_on) and checks the error word after the first replay: a
    # gate that timed out (two chains mapped onto one in-order hardware queue, a profiler serialising kernels) means a re-capture with stream events.
    flag_sync = True            # policy: captured replays use flags (HRNetPose._run); False / PAM_FLAG_SYNC=0: stream events everywhere
    flags_on = False            # state: this forward is being issued with flags
    flag_host_err = None        # pinned int32 word that receives 1 when a gate of ANY replay times out (HRNetPose reads it before every replay)
    flag_max_us = 2000000       # a gate gives up after 2 s (a systematic deadlock, found by the check after the first replay) and raises the error word
    flag_dev_void = None        # device int32 word that receives 1 at any time-out: what the frame kernel's input guard reads (HRNetPose.void_word)
    _flag_limit = None          # the bound as a device word the gates read when they start (set_flag_limit changes it for captured gates too)
    _flags = None

    def flag_limit(self):
        if self._flag_limit is None:
            self._flag_limit = torch.tensor([int(self.flag_max_us)], dtype=torch.int32, device=self.device)
        return self._flag_limit

    def set_flag_limit(self, us):
        """Bound (microseconds) of every gate that STARTS after the current stream's work so far, captured ones included (tests force a
        time-out in replay k this way; 0 = give up at the first poll that finds a branch missing)."""
        self.flag_max_us = int(us)
        self.flag_limit().fill_(int(us))

    def _flag_begin(self):
        """counters + error word of ONE forward (word 0 = error); zeroed on the caller's stream in front of the fork -- inside a capture the
        fill is a node of the graph, i.e. it runs at every replay"""
        self._flags = torch.zeros(128, dtype=torch.int32, device=self.device)
        self._flag_next = 1
        if self._keep is not None:
            self._keep.append(self._flags)

    def _flag_new(self):
        i = self._flag_next
        self._flag_next += 1
        assert i < 128
        return i

    def _sig(self, i):
        rc = self.lib.pam_flag_signal(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), C.c_void_p(self._flags.data_ptr() + 4 * i))
        if rc != 0:
            raise _lib.PamError('pam_flag_signal failed (%d)' % rc)

    flag_per_output = False     # round 6 experiment: one counter per OUTPUT of a module (a sum waits for the blocks and chains it reads)
                                # instead of one per module (every sum waits for every branch's whole tail)

    def _sig_mask(self, base, mask):
        if not mask:
            return
        rc = self.lib.pam_flag_signal_mask(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), C.c_void_p(self._flags.data_ptr() + 4 * base),
                                           C.c_uint32(int(mask)))
        if rc != 0:
            raise _lib.PamError('pam_flag_signal_mask failed (%d)' % rc)

    def _gate(self, i, target, arrive=False):
        rc = self.lib.pam_flag_gate(C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), C.c_void_p(self._flags.data_ptr() + 4 * i),
                                    int(target), C.c_void_p(self._flags.data_ptr()), C.c_void_p(self.flag_limit().data_ptr()), 1 if arrive else 0,
                                    C.c_void_p(self.flag_host_err.data_ptr()) if self.flag_host_err is not None else None,
                                    C.c_void_p(self.flag_dev_void.data_ptr()) if self.flag_dev_void is not None else None)
        if rc != 0:
            raise _lib.PamError('pam_flag_gate failed (%d)' % rc)

    def _branch_blocks(self, mod, b, blocks, x):
        """The BasicBlocks of branch b on the current stream."""
        fused = mod['fused'][b]
        if self.knock_out & (1 << b):
            return x
        if blocks and fused is not None and (self.block2 & (1 if fused[0].c == 48 else 2)):
            for op in fused:
                x = self.basic_block2(op, x)
            return x
        for c1, c2 in blocks:
            y = self.conv(c1, x, relu=True)
            if self.knock_conv2:                                      # diagnostics: half the launches AND half the work of the un-fused branches
                x = y
                continue
            x = self.conv(c2, y, res=x, relu=True)
        return x

    def _hr_module(self, mod, xs):
        """xs[b]: tensor, or ('lazy', transition op, source tensor) for a branch this stage creates.
        Stream b runs branch b's blocks and the fuse chains that hang off its output; ONE join; sum i on stream i."""
        self._epoch()
        xs = list(xs)
        fuse = mod['fuse']
        terms = [dict() for _ in fuse]
        flags = self.flags_on and self.multi_stream
        per_out = bool(flags and self.flag_per_output)
        ctail = self._flag_new() if (flags and not per_out) else None
        cbase, contrib = None, None
        if per_out:
            cbase = self._flag_next
            for _ in fuse:
                self._flag_new()
            nb = len(mod['branches'])
            contrib = [[b for b in range(nb) if b != i and b < len(row) and row[b] is not None] for i, row in enumerate(fuse)]
        for b in [q for q in self.order if q < len(mod['branches'])]:
            with torch.cuda.stream(self._stream(b)):
                x = xs[b]
                if isinstance(x, tuple):                              # transition conv runs on the new branch's own stream
                    if flags and len(x) > 3 and x[3] is not None:
                        self._gate(x[3], 1)                           # ... behind the sum it reads, which another stream issued
                    x = self.conv(x[1], x[2], relu=True)
                self._st('b%d start' % b)
                x = self._branch_blocks(mod, b, mod['branches'][b], x)
                self._st('b%d blocks' % b)
                xs[b] = x
                mg = mod['merged'].get(b) if self.merge_fuse else None
                heads = {}
                if mg is not None:                                    # first conv of all down chains from this branch in one launch
                    y = self.conv(mg['op'], x, relu=True, relu_from=mg['relu_from'])
                    heads = {i: y[:, off:off + c] for i, off, c, _ in mg['parts']}
                fmask = 15 if self.fused_sums is True else int(self.fused_sums)   # bit i: output i's sum carries its 1x1 products (k_fuse_sum)
                if per_out:                                           # the fused sums of finer outputs read this branch's blocks' output itself
                    self._sig_mask(cbase, sum(1 << i for i, row in enumerate(fuse) if i < b and b < len(row) and row[b] is not None
                                              and row[b][0] == 'up' and (fmask >> i) & 1 and mod['fsum'][i] is not None))
                mu = mod['merged_up'].get(b) if (self.merge_fuse and self.merge_up and not fmask) else None
                if mu is not None:                                    # all 1x1 up-convolutions from this branch in one launch
                    if self.knock_up and b == len(mod['branches']) - 1:   # diagnostics: what the last finisher's tail is worth
                        y = self._new(x.shape[0], mu['op'].cout, x.shape[2], x.shape[3], x.device)
                    else:
                        y = self.conv(mu['op'], x)
                    for i, off, c, sh in mu['parts']:
                        terms[i][b] = (y[:, off:off + c], sh)
                    if per_out:
                        self._sig_mask(cbase, sum(1 << i for i, _, _, _ in mu['parts']))
                for i, row in enumerate(fuse):
                    f = row[b] if b < len(row) else None
                    if f is None or (f[0] == 'up' and (mu is not None or (fmask >> i) & 1)):
                        continue
                    if f[0] == 'up':
                        terms[i][b] = (self.conv(f[1], x), f[2])
                    else:
                        t, ops = (heads[i], f[1][1:]) if i in heads else (x, f[1])
                        k0 = len(f[1]) - len(ops)
                        for k, op in enumerate(ops):
                            t = self.conv(op, t, relu=(k0 + k < len(f[1]) - 1))
                        terms[i][b] = (t, 0)
                    if per_out:
                        self._sig_mask(cbase, 1 << i)
                self._st('b%d tail' % b)
                if flags and not per_out and b >= len(fuse):
                    self._sig(ctail)                                  # a branch without an output of its own only arrives (the last module of stage 4)
        if not flags:
            self._barrier()
        self._st('join')
        # out_i = relu(x_i + sum_{j>i} up(conv1x1(x_j)) + sum_{j<i} strided-conv-chain(x_j)), terms in branch order
        out = [None] * len(fuse)
        for i in [q for q in self.order if q < len(fuse)]:
            with torch.cuda.stream(self._stream(i)):
                if per_out:
                    self._gate(cbase + i, 1 + len(contrib[i]), arrive=True)   # this output's own producers only
                elif flags:
                    self._gate(ctail, len(mod['branches']), arrive=True)   # this stream's chain is done; wait for every other branch's
                tl = [terms[i][j] for j in sorted(terms[i])]
                fs = mod['fsum'][i] if ((15 if self.fused_sums is True else int(self.fused_sums)) >> i) & 1 else None
                if fs is not None:                                    # plain (down-chain) terms + the coarser branches through their 1x1 products
                    out[i] = self.fuse_sum(fs['op'], xs[i], [t for t, _ in tl], [xs[j] for j in fs['srcs']], relu=True, max_wg=self.fs_cap[i])
                else:
                    out[i] = self.upsample_add(xs[i], [t for t, _ in tl], [sh for _, sh in tl], relu=True) if tl else torch.relu(xs[i])
                self._st('sum%d' % i)
        return out

    def features(self, x8):
        """x8: (N, 8, H, W) channels-last bf16 (RGB + 5 zero channels) -> (N, 48, H/4, W/4) channels-last bf16."""
        self._keep = []
        return self._features(x8)

    fuse_tail = True            # layer1: conv3 + residual + next conv1 of every Bottleneck in one launch (csrc/pam_pw.hip)
    tail_cfg = 0                # its wave-tile size (0 = automatic)
    fuse_bneck0 = True          # the first block too (its downsample fragments live in registers: they do not fit LDS beside the rest): -0.7 % at 20 crops, -1.2 % at 8
    fuse_bneck = True           # blocks 1-3 of layer1: 3x3 + pointwise tail in one launch (csrc/pam_bneck.hip; needs fuse_tail): -3.0 % at 20 crops, -2.1 % at 8, -1.6 % at 40
    fuse_stem = True            # conv1 + conv2 + layer1[0].conv1 in one launch (csrc/pam_stem.hip; needs fuse_tail): -2.2 % at 20 crops, -1.5 % at 8, -2.5 % at 40
    stop_after = None           # diagnostics (tools/stage_times.py): 'stem' | 'layer1' | 'stage2' | 'stage3' -> the forward ends there

    def _end(self, xs):
        if self.multi_stream:                                                                # final join only (no re-fork: capture must end with no forked stream)
            cur = torch.cuda.current_stream(self.device)
            for st in self.side:
                cur.wait_stream(st)
        return xs[0]

    def _epoch(self):
        if self.arena is not None:
            self.arena.epoch()

    def _features(self, x8):
        if self.flags_on and self.multi_stream and x8.device.type != 'meta':
            self._flag_begin()
        x = self._head(x8)
        if self.stop_after in ('stem', 'layer1'):
            return x
        return self._body(x)

    def _head(self, x8):
        """stem + layer1: one dependent chain on the caller's stream -> the (N, 256, H/4, W/4) tensor the branches start from"""
        self._epoch()
        n8, _, h8, w8 = x8.shape
        # the fused kernels index with 32 bits: > 606 crops of 384 x 288 (N H W 512 >= 2^31 at a quarter of the resolution) take the
        # un-fused launches, which address with 64 bits
        small = n8 * ((h8 + 3) // 4) * ((w8 + 3) // 4) * 512 < 2 ** 31 and n8 * h8 * w8 * 16 < 2 ** 31
        if self.fuse_stem and self.fuse_tail and self.stop_after != 'stem' and small:
            x0, y = self.stem_fused(self.stem, x8)
        else:
            x = self.conv(self.conv1, x8, relu=True)
            x = self.conv(self.conv2, x, relu=True)
            if self.stop_after == 'stem':
                return x
            x0, y = x, None
        if self.fuse_tail:
            # layer1 as 1 + 4 x 2 launches: conv1 of the first block, then per block the 3x3 and ONE pointwise-tail launch (conv3 + residual
            # / downsample + ReLU + the next block's conv1): the 256-channel tensor is written once and read once per block
            res = None
            if y is None:
                y = self.pointwise64(self.pw0, x0)
            for i, b in enumerate(self.layer1):
                if self.fuse_bneck and small and (i > 0 or self.fuse_bneck0):
                    x, y = self.bottleneck_fused(self.bnecks[i], y, res, x0 if i == 0 else None)
                else:
                    y2 = self.conv(b['c2'], y, relu=True)
                    x, y = self.bottleneck_tail(self.tails[i], y2, x0 if i == 0 else None, res, self.tail_cfg)
                res = x
        else:
            x = x0
            for b in self.layer1:
                r = x if b['down'] is None else self.conv(b['down'], x)
                y = self.conv(b['c1'], x, relu=True)
                y = self.conv(b['c2'], y, relu=True)
                x = self.conv(b['c3'], y, res=r, relu=True)
        return x

    def _new_branch(self, op, xs):
        """The lazy transition of a stage's new branch: it reads the last output of the previous stage, which stream len(xs) - 1 issued."""
        if self.flags_on and self.multi_stream:
            f = self._flag_new()
            with torch.cuda.stream(self._stream(len(xs) - 1)):
                self._sig(f)                                          # behind that stream's sum
            return ('lazy', op, xs[-1], f)
        self._barrier()
        return ('lazy', op, xs[-1])

    def _body(self, x):
        """stages 2-4 on the branch streams"""
        self._barrier()                                               # fork: branch streams must see layer1's output
        xs = [('lazy', self.t1[0], x), ('lazy', self.t1[1], x)]
        for m in self.stage2:
            xs = self._hr_module(m, xs)
        if self.stop_after == 'stage2':
            return self._end(xs)
        xs = xs + [self._new_branch(self.t2, xs)]                     # the new branch's stream reads the last sum of stage 2
        for m in self.stage3:
            xs = self._hr_module(m, xs)
        if self.stop_after == 'stage3':
            return self._end(xs)
        xs = xs + [self._new_branch(self.t3, xs)]
        for m in self.stage4:
            xs = self._hr_module(m, xs)
        return self._end(xs)
