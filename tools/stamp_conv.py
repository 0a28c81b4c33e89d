#!/usr/bin/env python3
"""NEEDS THE DIAGNOSTIC LIBRARY: `make -C <package>/csrc clean && make -C <package>/csrc DIAG=1` (the shipped build has no stamp code).
In-kernel phase stamps of k_conv3x3 (diagnostic; dbg bit 64)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn as nn
import pam
from pam import _lib, hrnet_hip
dev = torch.device("cuda:0"); NB = int(sys.argv[1]) if len(sys.argv) > 1 else 20
EXTRA = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # extra knock-out bits: 1 = no global loads, 32 = no LDS store pass, 2 = no MFMA loop
e = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); e.lib = _lib.load(); e.device = dev
buf = torch.zeros(4096 * 64, dtype=torch.int64, device=dev)
e.lib.pam_conv_debug_stamps(C.c_void_p(buf.data_ptr()))
for (h, w, cin, cout) in [(96, 72, 48, 48), (48, 36, 96, 96), (24, 18, 192, 192), (12, 9, 384, 384)]:
    conv = nn.Conv2d(cin, cout, 3, 1, 1, bias=True); op = hrnet_hip.PackedConv(conv, dev)
    x = torch.randn((NB, cin, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    res = torch.randn((NB, cout, h, w)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    e.tile_cfg = -1
    for _ in range(3): e.conv(op, x, res=res, relu=True)
    buf.zero_(); e.tile_cfg = 164 + EXTRA
    e.conv(op, x, res=res, relu=True); torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 64)
    st = st[st[:, 0] > 0]
    nch = cin // (48 if cin == 48 else (64 if cin >= 192 else 32))
    t0 = st[:, 0].min()
    d = lambda a, b: np.median((st[:, b] - st[:, a])) 
    print('C%d blocks=%d  start-spread %.0f  setup(0->1) %.0f | per chunk: wait+sync %s lstore %s sync %s compute %s | epilogue %.0f | block total med %.0f max-end %.0f (cycles @100MHz-less: s_memtime ticks)'
          % (cin, len(st), np.ptp(st[:, 0]), d(0, 1),
             [int(d(1 if c == 0 else 5 + 4 * (c - 1), 2 + 4 * c)) for c in range(nch)], [int(d(2 + 4 * c, 3 + 4 * c)) for c in range(nch)],
             [int(d(3 + 4 * c, 4 + 4 * c)) for c in range(nch)], [int(d(4 + 4 * c, 5 + 4 * c)) for c in range(nch)],
             d(60, 61), np.median(st[:, 61] - st[:, 0]), (st[:, 61].max() - t0)))
