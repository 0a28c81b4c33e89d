#!/usr/bin/env python3
"""Stress of the device-side flag synchronisation (development tool): replays of several crop counts / executor configurations, the
host error word checked at the end of each; --procs 2 runs two processes on the same device at the same time."""
import os, sys, argparse, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--counts', default='4,10,12,20,35'); ap.add_argument('--replays', type=int, default=300); ap.add_argument('--procs', type=int, default=1)
ap.add_argument('--child', action='store_true')
args = ap.parse_args()
if args.procs > 1 and not args.child:
    ps = [subprocess.Popen([sys.executable, __file__, '--counts', args.counts, '--replays', str(args.replays), '--child']) for _ in range(args.procs)]
    sys.exit(max(p.wait() for p in ps))
import torch
import pam
from pam import hrnet
net = hrnet.HRNetPose(48, 17, None, use_graph=True, autotune=True, max_crops=40)
bad = 0
for n in [int(c) for c in args.counts.split(',')]:
    x = net.input_buffer(n)
    x.normal_()
    t0 = time.perf_counter()
    net.features(x); torch.cuda.synchronize()
    for _ in range(args.replays):
        net.features(x)
    torch.cuda.synchronize()
    # round 6: a time-out no longer raises -- the object notes it (flag_timeouts), switches to stream events and marks the forwards since
    # as void until their consumer re-runs them (check_void / clear_void); a raw caller like this one looks after synchronising
    void = net.check_void()
    print('pid %d  n=%d  config=%s  flags=%s  time-outs so far=%d  void pending=%s  %.2f ms/replay' % (os.getpid(), n, net.tuned[n]['choice'],
          net.flag_synced.get((n, 'features', 0)), net.flag_timeouts, void, (time.perf_counter() - t0) / (args.replays + 1) * 1e3), flush=True)
    if void:
        net.clear_void()
        y1 = net.features(x).clone(); torch.cuda.synchronize()          # the re-run, with stream events by now
        y2 = net.features(x).clone(); torch.cuda.synchronize()
        bad += 0 if (torch.equal(y1, y2) and not net.check_void()) else 1
print('pid %d  done: %d time-out(s) seen and recovered, %d failure(s)' % (os.getpid(), net.flag_timeouts, bad), flush=True)
sys.exit(1 if bad else 0)
