#!/usr/bin/env python3
"""Which branch chains of a stage-4 module overlap: hipGraph replay of subsets of the four 4-block chains (no fuse layer), each on its
own stream.  Development tool."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flags = sys.argv[2:]            # block2 (resident-weights 48-channel block), c96 (streamed 96-channel branch), unfused
net = hrnet.HRNetPose(48, 17, None, use_graph=False, backend='hip')
hip = net.hip
hip.block2 = 'block2' in flags
hip.fuse_blocks = not ('unfused' in flags or hip.block2)
hip.c96_slab = 96 if 'c96w' in flags else (48 if 'c96' in flags else 0)
mod = hip.stage4[0]
shapes = [(48, 96, 72), (96, 48, 36), (192, 24, 18), (384, 12, 9)]
xs = [torch.randn((n, c, h, w), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for c, h, w in shapes]

big = torch.randn((8192, 8192), device=dev, dtype=torch.bfloat16)
def run(subset):
    hip._keep = []
    cur = torch.cuda.current_stream(dev)
    # a ~0.5 ms matmul in front: the host has written every packet of the replay before the first branch kernel may start, so the
    # chains are not staggered by the ~2.5 us per node the replay takes to submit (as in the full forward, where the host runs ahead)
    hip._keep.append(big @ big)
    for st in hip.side: st.wait_stream(cur)
    for b in subset:
        with torch.cuda.stream(hip._stream(b)):
            x = xs[b]
            if b == 0 and hip.fuse_blocks:
                for k in range(4): x = hip.basic_blocks([mod['fused'][0][k]], [x], hip.fuse_waves)[0]
            else:
                x = hip._branch_blocks(mod, b, mod['branches'][b], x)
    for st in hip.side: cur.wait_stream(st)

def timeit(subset, iters=20):
    run(subset); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s): run(subset)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(iters): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
res = {}
base = timeit(())
print('front matmul alone %.1f us (subtracted below)' % base)
for r in (1, 2, 3, 4):
    for sub in itertools.combinations(range(4), r):
        res[sub] = timeit(sub) - base
        print('%-12s %7.1f us   (sum of its chains alone %7.1f)' % (sub, res[sub], sum(res[(b,)] for b in sub)), flush=True)
