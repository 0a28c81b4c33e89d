import sys, torch
dev = torch.device('cuda:0')
pat = sys.argv[1]
a = torch.ones(1 << 20, device=dev); b = torch.ones(1 << 20, device=dev); c = torch.ones(1 << 20, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a.mul_(2)
        ev = torch.cuda.Event(); ev.record(s1)
    if pat == 'side2side':
        with torch.cuda.stream(s2):
            s2.wait_event(ev); b.add_(a)
    elif pat == 'via_cur':
        cur.wait_event(ev); c.add_(a)
        with torch.cuda.stream(s2):
            b.mul_(3)
    elif pat == 'double_wait':
        with torch.cuda.stream(s2):
            s2.wait_event(ev); b.add_(a)
        cur.wait_event(ev); c.add_(a)
    elif pat == 'record_twice':
        with torch.cuda.stream(s1):
            a.mul_(2); ev2 = torch.cuda.Event(); ev2.record(s1)
        with torch.cuda.stream(s2):
            s2.wait_event(ev); b.add_(a); s2.wait_event(ev2); b.add_(a)
    cur.wait_stream(s1); cur.wait_stream(s2)
g.replay(); torch.cuda.synchronize()
print(pat, 'ok', a[0].item(), b[0].item(), c[0].item())
