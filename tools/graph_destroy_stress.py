#!/usr/bin/env python3
"""Does destroying captured hipGraphs corrupt the runtime's heap (development tool; the reason pam._lib.new_graph and HRNetPose._kept exist)?
Tunes crop counts 1..N in one process: 4 captures per crop count, of which the tuner drops the 3 slower ones.
  default:    the package's behaviour -- dropped captures are never destroyed
  DESTROY=1:  torch's default -- a dropped CUDAGraph is destroyed (hipGraphExecDestroy + hipGraphDestroy)
Measured on ROCm 7.2 / MI355X, N = 40: DESTROY=1 3 of 4 processes die ('double free or corruption (!prev)' or SIGSEGV inside a later
torch.cuda.synchronize / allocation); default 0 of 5."""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
net = hrnet.HRNetPose(48, 17, None, use_graph=True, autotune=True)
if os.environ.get('DESTROY') == '1':
    class _Drop(list):
        def append(self, g): pass
    net._kept = _Drop()
for n in range(1, N + 1):
    x = net.input_buffer(n)
    net.features(x)
    torch.cuda.synchronize()
print('ok', N, 'crop counts tuned', 'DESTROY' if os.environ.get('DESTROY') == '1' else 'kept', flush=True)
