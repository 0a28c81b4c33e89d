import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import pam
from pam import hrnet
dev = torch.device('cuda:0')
res = {}
for n in (20, 9, 35):
    nets = {}
    for name, po in (('per_module', False), ('per_output', True)):
        net = hrnet.HRNetPose(48, 17, None, use_graph=True, autotune=True)
        net.hip.flag_per_output = po
        net.flag_race = None
        nets[name] = net
    eager = hrnet.HRNetPose(48, 17, None, use_graph=False, autotune=True)
    eager.hip.apply_config(eager.config_for(n))
    x = eager.input_buffer(n); x.copy_(torch.randn(x.shape, device=dev).to(x.dtype)); x[:, 3:] = 0
    ref = eager.features(x).clone()
    for name, net in nets.items():
        y = net.features(x).clone(); torch.cuda.synchronize()
        assert torch.equal(ref, y), (name, n)
        assert net.flag_synced[(n, 'features', 0)] is True, (name, net.flag_synced)
    t = {k: [] for k in nets}
    for r in range(7):
        for name, net in nets.items():
            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(20): net.features(x)
            e1.record(); torch.cuda.synchronize()
            t[name].append(e0.elapsed_time(e1) / 20)
    a, b = np.median(t['per_module']), np.median(t['per_output'])
    print('n=%d per_module %.3f ms per_output %.3f ms  %+.2f %%  (config %s)' % (n, a, b, 100 * (b / a - 1), eager.config_for(n)), flush=True)
print('AB-DONE')
