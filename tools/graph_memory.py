#!/usr/bin/env python3
"""Device memory of the replay captures over a run whose crop count wanders (development tool): one HRNetPose, every
crop-count bucket of the Panoptic-31 workload (4 .. 220 crops, graph_bucket 4) captured and replayed, twice; prints the device memory
in use (hipMemGetInfo: everything, the graphs' private pools included) after the weights, after the first sweep and after the second --
the second sweep must add nothing (captures are reused), and the total is what a long-lived drop-in process holds."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pam
from pam import hrnet
dev = torch.device('cuda:0')
used = lambda: (lambda f, t: (t - f) / 2 ** 30)(*torch.cuda.mem_get_info(dev))
base = used()
net = hrnet.HRNetPose(48, 17, None, use_graph=True, autotune=True, max_crops=int(os.environ.get('MAX_CROPS', '248')))
torch.cuda.synchronize()
after_weights = used()
counts = list(range(4, 224, 4))
res = {'GiB_before': base, 'GiB_with_weights': after_weights, 'buckets': len(counts)}
for sweep in (1, 2):
    for n in counts:
        x = net.input_buffer(n)
        net.features(x)
    torch.cuda.synchronize()
    res['GiB_after_sweep_%d' % sweep] = used()
res['captures'] = len(net._graphs)
res['torch_reserved_GiB'] = torch.cuda.memory_reserved(dev) / 2 ** 30
res['configs'] = sorted(set(t['choice'] for t in net.tuned.values()))
print(json.dumps(res))
