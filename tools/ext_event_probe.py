import torch, time
dev = torch.device('cuda:0')
x = torch.randn(1 << 24, device=dev)
a = torch.cuda.Event(enable_timing=True, external=True); b = torch.cuda.Event(enable_timing=True, external=True)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    y = x * 2
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        y = x + 1
        a.record(s)
        for _ in range(10): y = y * 1.0001
        b.record(s)
        z = y + 1
for _ in range(3):
    g.replay(); torch.cuda.synchronize()
    print('elapsed inside graph: %.3f ms' % a.elapsed_time(b))
# external wait: graph waits on an event recorded outside
w = torch.cuda.Event(external=True)
side = torch.cuda.Stream(dev)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g2, stream=s):
        y2 = x + 1
        s.wait_event(w)
        z2 = y2 * 3
with torch.cuda.stream(side):
    q = x * 5
    w.record(side)
g2.replay(); torch.cuda.synchronize()
print('external wait ok', float(z2[0] - (x[0] + 1) * 3))
