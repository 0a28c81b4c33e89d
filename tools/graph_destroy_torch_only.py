#!/usr/bin/env python3
"""Does destroying captured MULTI-STREAM hipGraphs corrupt the HIP runtime's heap WITHOUT any of this repository's code in the process?
(root-cause the 'never destroy a graph' rule below this library.)  Only PyTorch kernels: every iteration captures a graph whose
body forks three side streams from the capture stream, runs a chain of small matmuls on each of the four, joins -- the shape of the
HRNet executor's capture -- replays it, and drops it (hipGraphExecDestroy + hipGraphDestroy) or, with KEEP=1, keeps it alive.
Exit code 0 and 'ok' = survived.  usage: [KEEP=1] graph_destroy_torch_only.py [iterations=160]"""
import os, sys
import torch
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 160
keep = os.environ.get('KEEP') == '1'
dev = torch.device('cuda:0')
side = [torch.cuda.Stream(dev) for _ in range(3)]
ws = [torch.randn((256, 256), device=dev, dtype=torch.bfloat16) for _ in range(4)]
kept = []


def body(x, depth):
    cur = torch.cuda.current_stream(dev)
    outs = []
    for m in range(3):                                   # three "modules": fork, four chains, join
        for st in side:
            st.wait_stream(cur)
        for b in range(4):
            with torch.cuda.stream(side[b - 1] if b else cur):
                y = x
                for _ in range(depth):
                    y = torch.relu(y @ ws[b])
                outs.append(y)
        for st in side:
            cur.wait_stream(st)
        x = outs[-1] + outs[-2] + outs[-3] + outs[-4]
    return x


for it in range(n_iter):
    n = 64 + 32 * (it % 7)
    x = torch.randn((n, 256), device=dev, dtype=torch.bfloat16)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        body(x, 4)
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body(x, 4)
    g.replay(); g.replay()
    torch.cuda.synchronize()
    if keep:
        kept.append((g, out))
    else:
        del g, out
    if it % 8 == 7:                                      # ordinary allocator traffic between the destructions
        junk = [torch.empty((1 << 20) * (k + 1), device=dev) for k in range(4)]
        del junk
        torch.cuda.synchronize()
print('ok', n_iter, 'graphs', 'kept' if keep else 'destroyed', flush=True)
