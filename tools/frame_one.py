#!/usr/bin/env python3
"""ONE k_frame configuration, launched a few times: the program tools/pmc_frame.sh puts directly behind `rocprofv3 --pmc ... --`
(no shell, no launcher in between).  Also prints the launch time and the per-phase table from the in-kernel clocks."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pam
from pam import _lib, synth
from oracle import cpu_ref as O          # camera construction for the synthetic rig only (development tool)
ap = argparse.ArgumentParser()
ap.add_argument('--size', default='S2'); ap.add_argument('--scenes', type=int, default=1)
ap.add_argument('--warm', type=int, default=20); ap.add_argument('--iters', type=int, default=12)
args = ap.parse_args()
dev = torch.device('cuda:0')
nf = args.warm + args.iters
seq = synth.make_sequence(args.size, n_frames=nf, seed=0)
cams = O.make_cameras(seq['calib'])
cfg = dict(synth.MATCHER_CFG[synth.SIZE_TO_DATASET[args.size]]); conf = cfg.pop('CONF_THRESHOLD')
C, S = len(cams), args.scenes
h = _lib.Handle(C, _lib.make_params(cfg, conf), max_dets=8, max_tracks=16, n_scenes=S)
h.set_cameras(np.stack([c.P for c in cams]), np.stack([c.F for c in cams]), np.stack([c.RK_INV for c in cams]), np.stack([c.position for c in cams]))
nd, dd = synth.pack_frames(seq['frames'], 8)
ndt = [torch.tensor(np.tile(nd[t], (S, 1)), dtype=torch.int32, device=dev) for t in range(nf)]
ddt = [torch.tensor(np.tile(dd[t][None], (S, 1, 1, 1, 1)), dtype=torch.float64, device=dev) for t in range(nf)]
st = torch.cuda.current_stream().cuda_stream
for t in range(args.warm):
    h.frame_dev(st, t, ndt[t].data_ptr(), ddt[t].data_ptr())
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for t in range(args.warm, nf):
    h.frame_dev(st, t, ndt[t].data_ptr(), ddt[t].data_ptr())
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / args.iters * 1e3
oi, od = h.fetch(st); h.sync(st)
rec = h.decode(0)
f = od[0][:16] * 1e6
T = rec['n_tracks']; V = int(np.mean([t['V'] for t in rec['tracks']])) if T else 0
P = synth.SIZES[args.size]['P']
alg = C * P * 408 + T * ((11 + 2) * 408 + V * 408) + T * 408 + C * P * 4 + T * 68          # SURVEY 8d (same formula as bench.py)
phases = {'P0-P3 association (projection affinity, LSAP per view, add_pose, unmatched)': f[1] - f[0],
          'P4a view selection + prediction': f[4] - f[1], 'P4b epipolar conflicts + ray distances': f[5] - f[4],
          'P4c greedy filter + DLT': f[6] - f[5], 'P4d success test': f[7] - f[6], 'P4e smoothing': f[8] - f[7],
          'P4f-g append, velocity, life cycle': f[2] - f[8], 'P5 hypothesis initialisation': f[3] - f[2],
          'P6-P7 compaction + output record': f[11] - f[10]}
print(json.dumps({'workload': args.size, 'scenes': S, 'views': C, 'tracks': T, 'views_per_track': V, 'us_per_launch': us,
                  'algorithmic_bytes_per_scene': alg, 'algorithmic_GBs': alg * S / us / 1e3, 'status': rec['status'] | rec['status_sticky'],
                  'phase_us_scene0': {k: float(v) for k, v in phases.items()}, 'in_kernel_total_us_scene0': float(f[11] - f[0]),
                  'block_threads': 1024 if C > 8 else 256,
                  'fine_us': {'P0+P1': float(f[12] - f[0]), 'P2 lsap': float(f[13] - f[12]), 'P3': float(f[1] - f[13])} if os.environ.get('PAM_FINE') else None,
                  'split_us': {'launch1_tail': float(f[15] - f[4]), 'gap_1_2': float(f[12] - f[15]), 'launch2_wg0': float(f[13] - f[12]), 'gap_2_3': float(f[14] - f[13]), 'launch3_prologue': float(f[5] - f[14])} if f[15] > 0 else None}), flush=True)
h.close()
