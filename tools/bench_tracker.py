#!/usr/bin/env python3
"""k_frame latency / phase clocks for single and batched scenes (development tool)."""
import os, sys, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pam
from pam import _lib, synth
from oracle import cpu_ref as O
ap = argparse.ArgumentParser(); ap.add_argument('--sizes', default='S2,S4'); ap.add_argument('--scenes', default='1,256,2048')
args = ap.parse_args()
dev = torch.device('cuda:0')
for size in args.sizes.split(','):
    nf = 60
    seq = synth.make_sequence(size, n_frames=nf, seed=0)
    cams = O.make_cameras(seq['calib'])
    cfg = dict(synth.MATCHER_CFG[synth.SIZE_TO_DATASET[size]]); conf = cfg.pop('CONF_THRESHOLD')
    prm = _lib.make_params(cfg, conf)
    C = len(cams)
    nd, dd = synth.pack_frames(seq['frames'], 8)
    for S in [int(s) for s in args.scenes.split(',')]:
        h = _lib.Handle(C, prm, max_dets=8, max_tracks=16, n_scenes=S)
        h.set_cameras(np.stack([c.P for c in cams]), np.stack([c.F for c in cams]), np.stack([c.RK_INV for c in cams]), np.stack([c.position for c in cams]))
        ndt = [torch.tensor(np.tile(nd[t], (S, 1)), dtype=torch.int32, device=dev) for t in range(nf)]
        ddt = [torch.tensor(np.tile(dd[t][None], (S, 1, 1, 1, 1)), dtype=torch.float64, device=dev) for t in range(nf)]
        st = torch.cuda.current_stream().cuda_stream
        for t in range(20):
            h.frame_dev(st, t, ndt[t].data_ptr(), ddt[t].data_ptr())
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(True), torch.cuda.Event(True)
        a.record()
        for t in range(20, nf):
            h.frame_dev(st, t, ndt[t].data_ptr(), ddt[t].data_ptr())
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / (nf - 20) * 1e3
        oi, od = h.fetch(st); h.sync(st)
        c = od[0][:4]
        f = od[0][:12] * 1e6
        print('   fine stamps (us from start): sel %.1f conf %.1f dlt %.1f succ %.1f smooth %.1f append %.1f | upd_end %.1f init_end %.1f rec_end %.1f' % (f[4]-f[0], f[5]-f[0], f[6]-f[0], f[7]-f[0], f[8]-f[0], f[9]-f[0], f[2]-f[0], f[10]-f[0], f[11]-f[0]))
        print('%s scenes=%5d  %8.1f us/launch  %10.0f scene-frames/s  phases(us): assoc %.1f update %.1f init %.1f  tracks=%d status=%d'
              % (size, S, us, S / us * 1e6, (c[1] - c[0]) * 1e6, (c[2] - c[1]) * 1e6, (c[3] - c[2]) * 1e6, oi[0][0], oi[:, 1].max()), flush=True)
        h.close()
