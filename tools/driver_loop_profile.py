"""cProfile of bench.py's driver_loop leg on the Shelf-like workload (development tool; GPU box)."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import bench
from pam import synth
from pam.ivclabpose import Camera, fundamental_matrices
from pam.pipeline import FramePipeline
size='S2'; meta=synth.SIZES[size]; C=meta['C']; fw,fh=meta['w'],meta['h']
seq=synth.make_sequence(size, n_frames=80, seed=0)
cfg=dict(synth.MATCHER_CFG['Shelf']); conf=cfg.pop('CONF_THRESHOLD')
P32=seq['calib']['P'].astype(np.float32); K32=seq['calib']['K'].astype(np.float32); RT32=seq['calib']['RT'].astype(np.float32)
Fm=fundamental_matrices(K32,RT32)
cams=[Camera(j,P32[j],K32[j],RT32[j],Fm[j],w=fw,h=fh) for j in range(C)]
pipe=FramePipeline(cams,cfg,conf,(fh,fw),max_dets=8,max_tracks=16,shard='crops',overlap_tracker=True)
inp=bench.build_inputs(torch, synth, seq, size, 8, 1, 0, 'crops', pipe.device, 80)
r=bench.driver_loop(torch, synth, pipe, cams, cfg, conf, inp, size, 8, 20, 5)
pr=cProfile.Profile(); pr.enable()
r=bench.driver_loop(torch, synth, pipe, cams, cfg, conf, inp, size, 8, 60, 5)
pr.disable()
print(r['value'], r['serial']['value'], r['ahead']['s_per_frame_by_stage'])
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:4500])
