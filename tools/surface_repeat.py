import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import bench
from pam import synth
from pam.ivclabpose import Camera, fundamental_matrices
from pam.pipeline import FramePipeline
size='S2'; meta=synth.SIZES[size]; C=meta['C']; fw,fh=meta['w'],meta['h']
seq=synth.make_sequence(size, n_frames=70, seed=0)
cfg=dict(synth.MATCHER_CFG['Shelf']); conf=cfg.pop('CONF_THRESHOLD')
P32=seq['calib']['P'].astype(np.float32); K32=seq['calib']['K'].astype(np.float32); RT32=seq['calib']['RT'].astype(np.float32)
Fm=fundamental_matrices(K32,RT32)
cams=[Camera(j,P32[j],K32[j],RT32[j],Fm[j],w=fw,h=fh) for j in range(C)]
pipe=FramePipeline(cams,cfg,conf,(fh,fw),max_dets=8,max_tracks=16,shard='crops',overlap_tracker=True)
dev=pipe.device
inp=bench.build_inputs(torch, synth, seq, size, 8, 1, 0, 'crops', dev, 70)
# conv stack alone for reference
x=pipe.net.input_buffer(20); pipe.net.features(x); torch.cuda.synchronize()
a,b=torch.cuda.Event(True),torch.cuda.Event(True); a.record()
for _ in range(30): pipe.net.features(x)
b.record(); torch.cuda.synchronize(); print('conv stack 20 crops %.3f ms'%(a.elapsed_time(b)/30))
for rep in range(5):
    r=bench.surface_run(torch, synth, pipe, cams, cfg, conf, seq, inp, size, 8, 60, 5)
    print('surface %.1f fps %.3f ms'%(r['value'], r['ms_per_step']))
# ---- where the host time of one frame goes (perf_counter around the two calls; the first waits for nothing, the second for the GPU) ----
import contextlib, io
from pam.ivclabpose import ivclabpose
with contextlib.redirect_stdout(io.StringIO()):
    model = ivclabpose({'NAME': ''}, None, dict(cfg, NAME='Iterative'), conf, max_dets=8, max_tracks=16, device=dev.index)
model.pose_model = pipe.net; model.cameras = cams; model.tracker.set_cameras(cams)
frames = inp['frames']
pbls = []
for t in range(40):
    e = inp['per_frame'][t]
    pbl = [[] for _ in range(C)]
    for v, b in zip(e['vl'].tolist(), e['bx'].tolist()):
        pbl[v].append(dict(image_id=t, category_id=1, score=0.9, bbox=b, data=frames[v], feature=[]))
    pbls.append(pbl)
det_dev = [torch.tensor(inp['det_all'][t], dtype=torch.float64, device=dev) for t in range(40)]
det_host = [[inp['det_all'][t][v][:inp['n_det_all'][t][v]] for v in range(C)] for t in range(40)]
T = {'detect': [], 'track': [], 'gpu_idle_gap': []}
for rep in range(2):
    model.tracker.track_restart()
    for t in range(40):
        torch.cuda.synchronize()
        a = time.perf_counter()
        dump = model.PersonPoseDetect(imagelist=None, person_bbox_list=pbls[t], batch_size=20)
        b = time.perf_counter()
        dump.device_det.copy_(det_dev[t]); dump.poses_host = det_host[t]
        r = model.PersonTrack_Project3DPose(t, pbls[t], dump, 'SVD')
        c = time.perf_counter()
        if rep == 1 and t >= 5:
            T['detect'].append((b - a) * 1e6); T['track'].append((c - b) * 1e6)
print('host us per frame: PersonPoseDetect (enqueue only) median %.0f; PersonTrack_Project3DPose (enqueue + wait + 9-tuple) median %.0f' % (np.median(T['detect']), np.median(T['track'])))
# pieces of the post-wait work
tr = model.tracker
oi, od = tr._rec_i.numpy(), tr._rec_d.numpy()
a = time.perf_counter()
for _ in range(200): last = tr.handle.decode(0, oi, od)
b = time.perf_counter()
from pam.tracker import TrackView
for _ in range(200): tv = [TrackView(r_, tr.cameras, None) for r_ in last['tracks']]
c = time.perf_counter()
print('decode %.1f us, TrackViews %.1f us' % ((b - a) / 200 * 1e6, (c - b) / 200 * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for t in range(5, 25):
    dump = model.PersonPoseDetect(imagelist=None, person_bbox_list=pbls[t], batch_size=20); torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:2600])
