#!/usr/bin/env python3
"""NEEDS THE DIAGNOSTIC LIBRARY: `make -C <package>/csrc clean && make -C <package>/csrc DIAG=1`.
In-kernel phase stamps of the streamed 3x3 kernel k_conv3x3s (multiplier wave 0 of every workgroup)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
dev = torch.device("cuda:0"); NB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev
buf = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
for (h, w, cin, cout) in [(48, 36, 96, 96), (24, 18, 192, 192), (12, 9, 384, 384)]:
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=True); op = hrnet_hip.PackedConv(conv, dev)
    x = torch.randn((NB, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn((NB, cout, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    e.tile_cfg = -1
    e.lib.pam_conv_debug_stamps(None)
    for _ in range(3): e.conv(op, x, res=res, relu=True)
    buf.zero_(); e.lib.pam_conv_debug_stamps(C.c_void_p(buf.data_ptr()))
    e.conv(op, x, res=res, relu=True); torch.cuda.synchronize()
    e.lib.pam_conv_debug_stamps(None)
    st = buf.cpu().numpy().reshape(-1, 64)
    st = st[st[:, 0] > 0]
    nch = cin // 32
    d = lambda a, b: int(np.median(st[:, b] - st[:, a]))
    print('C%d blocks=%d  start -> chunk 0 %d | chunk k start -> chunk k+1 start %s last %d | epilogue %d | total med %d  span %d'
          % (cin, len(st), d(0, 3), [d(3 + 3 * k, 3 + 3 * (k + 1)) for k in range(nch - 1)], d(3 + 3 * (nch - 1), 60),
             d(60, 61), d(0, 61), st[:, 61].max() - st[:, 0].min()))
