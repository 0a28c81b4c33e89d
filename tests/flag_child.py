"""Run as a CHILD PROCESS by tests/test_gpu_flags.py (never collected by pytest: no test_ prefix).

Everything about the device-side flags of a captured HRNet forward (csrc/pam_sync.hip) that needs a FRESH process: whether the branch
chains of a hipGraph get hardware queues of their own depends on how many streams the process has made, and a pytest session has made
dozens (a flagged capture whose first replay times out there falls back to stream events -- correct, but then nothing below is tested).
One child for all the checks; each prints a marker line the parent asserts on.

Sections: FLAGS (flagged replays == eager forward; the capture-time race), HOSTWORD (a time-out seen through the pinned word switches
the object to stream events without an exception), SURFACE (a gate time-out in call k of the drop-in loop: golden trace S2 through
ivclabpose with the real forward in front of the tracker -- 9-tuples and tracker state equal the reference's in EVERY frame, frame k
included), PIPELINE (the same inside FramePipeline with the host running ahead: the frame kernel skips, FrameVoid names the frame to
resume from, the re-submitted run ends in the same state), MEMORY (replay cache + arena bytes of a prewarmed S2 pipeline)."""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch

import pam                                                            # noqa: F401  (the package alias)
from pam import hrnet, _lib, synth


def section_flags(x, ref):
    b = hrnet.HRNetPose(48, 17, None, use_graph=True)
    b.flag_race = None                                  # no race against stream events: this object keeps its flagged captures
    for _ in range(10):
        y = b.features(x)
    torch.cuda.synchronize()
    assert b.flag_synced[(5, 'features', 0)] is True, (b.flag_synced, getattr(b, '_flag_sync_failed', None))
    assert torch.equal(ref, y)
    assert b.captures == 1 and len(b._dead_graphs) == 0 and int(b._flag_host_np[0]) == 0 and int(b.void_word.item()) == 0
    # every flagged capture has the stream-event form of the same forward beside it; both are timed one at a time and back to back
    # (interleaved rounds, medians), flags are kept only where they win by more than flag_margin, and a replay uses the form that is
    # faster the way the object is used at that moment (flag_race); a second slot follows the verdict
    d = hrnet.HRNetPose(48, 17, None, use_graph=True)
    for mode in ('serial', 'throughput', 'serial'):
        d.flag_race = mode
        yd = d.features(x).clone(); torch.cuda.synchronize()
        t = d.flag_timing[5]
        assert all(t['kept'][m] == (t['ms'][m][0] <= (1.0 - d.flag_margin) * t['ms'][m][1]) for m in ('serial', 'throughput'))
        assert len(d._dead_graphs) == 0 and set(d._alt[(5, 'features', 0)]) == {'flags', 'events'} and torch.equal(ref, yd)
        y1 = d.features(x, slot=1).clone(); torch.cuda.synchronize()
        assert torch.equal(ref, y1) and d.captures == 2
    # the ordering assumption behind the flags: a signal's release covers the signalling wave only; that the stores of the
    # kernels in front of it have left their XCD's L2 relies on the release the runtime puts at every kernel boundary of a graph chain.
    # Asserted here on every run: for every crop-count bucket of the Shelf frame and every executor configuration the flagged replay
    # and the stream-event replay of the same forward (both captured, both alive) give the same bits, five times over
    from pam import hrnet_hip
    for name in hrnet_hip.HipHRNet.CONFIGS:
        e = hrnet.HRNetPose(48, 17, None, use_graph=True)
        e.config_for = lambda n, name=name: name
        e.flag_race = 'throughput'
        for n in (4, 12, 20):
            xn = e.input_buffer(n)
            xn.copy_(torch.randn(xn.shape, generator=torch.Generator().manual_seed(n)).to(xn.device).to(xn.dtype)); xn[:, 3:] = 0
            e.features(xn)
            alt = e._alt[(n, 'features', 0)]
            assert alt is not None and set(alt) == {'flags', 'events'}, (name, n, e.flag_synced)
            for _ in range(5):
                alt['flags'][0].replay(); yf = alt['flags'][2].clone()
                alt['events'][0].replay(); ye = alt['events'][2].clone()
                torch.cuda.synchronize()
                assert torch.equal(yf, ye), (name, n)
        assert int(e._flag_host_np[0]) == 0 and int(e.void_word.item()) == 0
    # the experiment of round 6 (off: +0.2 ... +0.7 % per forward): one counter per OUTPUT of a module, several signalled per launch
    # (pam_flag_signal_mask) -- a sum then waits only for the blocks and chains it reads; same launches, same bits
    po = hrnet.HRNetPose(48, 17, None, use_graph=True)
    po.hip.flag_per_output = True
    po.flag_race = None
    for _ in range(3):
        yp = po.features(x).clone(); torch.cuda.synchronize()
        assert torch.equal(ref, yp)
    assert po.flag_synced[(5, 'features', 0)] is True and int(po._flag_host_np[0]) == 0
    print('FLAGS-OK', flush=True)
    return b


def section_hostword(b, x, ref):
    # a time-out in a LATER replay reaches the host through the pinned word: the next call switches the object to stream events (no
    # exception: the consumers of the void forwards re-run them -- SURFACE / PIPELINE below)
    b._flag_host_np[0] = 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        y4 = b.features(x).clone(); torch.cuda.synchronize()
    assert any('gate' in str(m.message) for m in w)
    assert b.void_pending and b.flag_timeouts == 1
    assert b.flag_synced[(5, 'features', 0)] is False and torch.equal(ref, y4)
    b.clear_void()
    assert not b.void_pending and int(b.void_word.item()) == 0
    print('HOSTWORD-OK', flush=True)


class GoldenPose(hrnet.HRNetPose):
    """The real crop -> conv stack -> head + arg-max chain, with the decoded rows then REPLACED on the device by the golden keypoints of
    the frame (random weights decode noise; what is under test is the chain's ordering and the guard behind it, and the tracker's
    numbers have to be the golden trace's)."""
    gold_kp = None          # (n, 17, 3) float32 (x, y, score)
    gold_rows = None        # (n, 17, 3) float64 (y, x, score)
    decodes = 0

    def head_decode(self, f, view_of, slot_of, boxes, det, kp=None, heat=None, n=None):
        hrnet.HRNetPose.head_decode(self, f, view_of, slot_of, boxes, det, kp, heat, n)
        k = int(view_of.numel())
        assert k == self.gold_rows.shape[0], 'one batch per call in this test'
        det[view_of.long(), slot_of.long()] = self.gold_rows
        if kp is not None:
            kp.copy_(self.gold_kp)
        self.decodes += 1


def section_surface(host_first, kill_at=(40, 41, 42, 90)):
    """Golden trace S2 through the drop-in facade with the pose network in front of the tracker; the gates of the forwards of frames
    `kill_at` are given a bound of 0 us (the first of them that is not a blank frame times out; the object is on stream events after
    it).  Every frame's 9-tuple and tracker state must equal the reference's.  host_first: the caller reads the dump before it passes
    it on (the re-run happens in DumpResults); else the dump goes straight to the tracker (the frame kernel skips, ivclabpose re-runs)."""
    import golden_io as G
    from trace_driver import run_trace
    from pam.ivclabpose import ivclabpose
    dev = torch.device('cuda:0')
    box = {}

    class Model(object):
        def __init__(self, cfg, conf):
            self.m = ivclabpose(person_detector={'NAME': ''}, pose_detector=None, person_matcher=dict(cfg, NAME='Iterative'), conf_threshold=conf)
            self.pose = GoldenPose(48, 17, None, use_graph=True, max_dets=self.m.tracker.max_dets)
            self.pose.flag_race = None
            self.m.pose_model = self.pose
            self.m.tracker.set_input_guard(self.pose)
            self.tracker = self.m.tracker
            g = torch.Generator().manual_seed(3)
            self.frames = None
            self.g = g
            box['model'] = self

        def GetCameraParameters(self, *a, **k):
            return self.m.GetCameraParameters(*a, **k)

        def PersonTrack_Project3DPose(self, t, pbl, dr, b):
            C = len(pbl)
            if self.frames is None:
                self.frames = [torch.randint(0, 256, (776, 1032, 3), dtype=torch.uint8, generator=self.g).to(dev) for _ in range(C)]
            rows = []
            for v in range(C):
                for p, d in zip(pbl[v], dr[v]):
                    p['data'] = self.frames[v]
                    k = np.asarray(d['keypoints'], dtype=np.float64).reshape(17, 3).copy()
                    k[:, 2] = d['keypoints_score']
                    rows.append(k)
            rows = np.stack(rows)                                                   # (n, 17, 3) (x, y, score) float64
            self.pose.gold_kp = torch.tensor(rows, dtype=torch.float32, device=dev)
            self.pose.gold_rows = torch.tensor(rows[:, :, [1, 0, 2]], dtype=torch.float64, device=dev)
            kill = t in kill_at
            if kill:
                self.pose.hip.set_flag_limit(0)                                     # every gate that finds a branch missing gives up at once
            dump = self.m.PersonPoseDetect(None, pbl, batch_size=20)
            assert dump.device_valid()
            # the 2D rows of the 9-tuple come from the float32 keypoints of the dump; the golden ones are float64: hand them over exactly
            first = np.concatenate([[0], np.cumsum([len(v) for v in pbl])])
            exact = [rows[first[v]:first[v + 1]][:, :, [1, 0, 2]] for v in range(C)]
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                if host_first:
                    _ = dump[0]                          # the HOST looks first (materialise): the re-run happens there
                dump.poses_host = exact
                out = self.m.PersonTrack_Project3DPose(t, pbl, dump, b)
            if kill:
                self.pose.hip.set_flag_limit(2000000)
            return out

    nf = 0
    for t, tr, model in run_trace('S2', Model, atol3d=1e-6):
        k = 'f%d.st.' % t
        trs = model.tracker.tracks
        assert model.tracker.last['status'] == 0
        assert [x.track_id for x in trs] == tr[k + 'ids'].tolist(), t
        assert [x.state for x in trs] == tr[k + 'state'].tolist() and [x.hits for x in trs] == tr[k + 'hits'].tolist()
        assert [x.age for x in trs] == tr[k + 'age'].tolist() and [x.time_since_update for x in trs] == tr[k + 'tsu'].tolist()
        assert [x.nhist for x in trs] == tr[k + 'nhist'].tolist() and [x.last_time for x in trs] == tr[k + 'last_time'].tolist()
        nf += 1
    m = box['model']
    # the first killed frame timed out (flags were on); after it the object is on stream events, so the later "kills" cannot time out
    assert m.pose.flag_timeouts == 1, m.pose.flag_timeouts
    assert not m.pose.void_pending and int(m.pose.void_word.item()) == 0 and m.pose._flag_sync_ok() is False
    assert m.pose.decodes == nf + 1, (m.pose.decodes, nf)                           # exactly one forward was run twice
    assert nf > 100
    print('SURFACE-%s-OK frames=%d timeouts=%d' % ('HOST' if host_first else 'DEVICE', nf, m.pose.flag_timeouts), flush=True)


def _rig(size):
    from pam.ivclabpose import Camera, fundamental_matrices
    meta = synth.SIZES[size]
    seq = synth.make_sequence(size, n_frames=40, seed=3)
    cfg = dict(synth.MATCHER_CFG[synth.SIZE_TO_DATASET[size]]); conf = cfg.pop('CONF_THRESHOLD')
    P32 = seq['calib']['P'].astype(np.float32); K32 = seq['calib']['K'].astype(np.float32); RT32 = seq['calib']['RT'].astype(np.float32)
    Fm = fundamental_matrices(K32, RT32)
    cams = [Camera(j, P32[j], K32[j], RT32[j], Fm[j], w=meta['w'], h=meta['h']) for j in range(meta['C'])]
    return seq, cams, cfg, conf, meta


def section_pipeline(kill=19, exchange='torch', records_only=False):
    """FramePipeline with the tracker of frame t under frame t + 1's conv stack and the host three frames ahead: the forward of frame
    `kill` loses its gates.  The frame kernel must skip that frame AND the ones issued behind it, results() must raise FrameVoid(first =
    kill), and the run re-submitted from there must produce, frame by frame, the records of a pipeline that never saw a time-out."""
    from pam.pipeline import FramePipeline
    seq, cams, cfg, conf, meta = _rig('S2')
    C, md = meta['C'], 8
    n_det_all, det_all = synth.pack_frames(seq['frames'], md)
    dev = torch.device('cuda:0')
    ref = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False)
    want = []
    for t in range(len(seq['frames'])):
        ref.track_step(t, torch.tensor(n_det_all[t], dtype=torch.int32, device=dev), torch.tensor(det_all[t], dtype=torch.float64, device=dev))
        want.append(ref.results())
    net = hrnet.HRNetPose(48, 17, None, use_graph=True, max_dets=md)
    pipe = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, net=net, overlap_tracker=True, exchange=exchange)
    net.flag_race = None                                # keep the flagged captures whatever the race says on this box
    if records_only:
        # the sharded mechanism alone (row e's default path, flags ON, in one process): the exchange runs through the library's own RCCL
        # all-gather on a one-rank communicator, the frame kernel reads the gathered records in place, and ONLY the void flag that
        # travels inside the records protects the tracker (the handle's own guard is taken away)
        pipe.handle.set_input_guard(None)
        assert pipe.gather.recv.data_ptr() != pipe.gather.send.data_ptr()
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (C, meta['h'], meta['w'], 3), dtype=torch.uint8, generator=g).to(dev)
    ptrs = torch.tensor([frames[v].data_ptr() for v in range(C)], dtype=torch.int64, device=dev)

    def same(a, b):
        assert a['n_tracks'] == b['n_tracks'] and a['frame_id'] == b['frame_id'] and a['status'] == b['status'] == 0
        for ta, tb in zip(a['tracks'], b['tracks']):
            for key in ('track_id', 'state', 'hits', 'age', 'time_since_update', 'emitted', 'order', 'nhist', 'last_time'):
                assert ta[key] == tb[key], (a['frame_id'], key)
            assert np.array_equal(ta['pose3d'], tb['pose3d']) and np.array_equal(ta['velocity'], tb['velocity'])
            assert np.array_equal(ta['nviews'], tb['nviews']) and np.array_equal(ta['matched_det'], tb['matched_det'])

    def submit(t, killed):
        nd = n_det_all[t]
        vl = [v for v in range(C) for _ in range(nd[v])]
        sl = [s for v in range(C) for s in range(nd[v])]
        bx = [[50.0 + 30 * s, 40.0 + 20 * v, 200.0, 380.0] for v in range(C) for s in range(nd[v])]
        if killed:
            net.hip.set_flag_limit(0)
        with pipe.frame():
            if vl:
                pipe.pose_step(ptrs, torch.tensor(vl, dtype=torch.int32, device=dev), torch.tensor(sl, dtype=torch.int32, device=dev),
                               torch.tensor(bx, dtype=torch.float32, device=dev))
            pipe.write_local(torch.tensor(det_all[t], dtype=torch.float64, device=dev))       # the golden rows over the decoded ones
            pipe.track_step(t, torch.tensor(nd, dtype=torch.int32, device=dev))
        if killed:
            net.hip.set_flag_limit(2000000)

    t, voids, checked, killed_once = 0, [], 0, False
    F = len(seq['frames'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        while t < F:
            submit(t, t == kill and not killed_once)
            killed_once |= (t == kill)
            if t % 3 == 2 or t == F - 1:                 # the host looks every third frame only
                try:
                    same(pipe.results(), want[t]); checked += 1
                except _lib.FrameVoid as e:
                    voids.append((e.first, e.last))
                    t = e.first
                    continue
            t += 1
    assert voids == [(kill, kill + 2 - (kill % 3))], voids
    assert net.flag_timeouts == 1 and not net.void_pending and int(net.void_word.item()) == 0
    assert checked >= F // 3
    print('PIPELINE-%sOK voids=%s' % ('RECORDS-' if records_only else '', voids), flush=True)


def section_memory(size='S2'):
    """What a prewarmed pipeline holds with BOTH forms of every flagged bucket alive (the verdict's question): device memory before /
    after, the activation arena, the number of captures.  (`memory:S4` on the command line: the 31-camera rig, 62 buckets up to 248 crops.)"""
    from pam.pipeline import FramePipeline
    seq, cams, cfg, conf, meta = _rig(size)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    pipe = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=4 if size == 'S2' else 8, prewarm=True, overlap_tracker=True)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    w = pipe.warmed
    both = sum(1 for a in pipe.net._alt.values() if a is not None)
    print('MEMORY-OK size=%s' % size, 'buckets=%s captures=%d both_forms=%d arena_MB=%.1f device_MB=%.1f seconds=%.1f'
          % (w['buckets'], w['captures'], both, w['arena_bytes'] / 2 ** 20, (free0 - free1) / 2 ** 20, w['seconds']), flush=True)


def main():
    want = set(sys.argv[1:]) or {'flags', 'surface', 'pipeline'}          # 'memory' runs as a child of its own (a report, and the
                                                                          # prewarm's dozen captures on top of this process's crash ROCm 7.2's graph code now and then)
    a = hrnet.HRNetPose(48, 17, None, use_graph=False)
    x = a.input_buffer(5)
    x.copy_(torch.randn(x.shape, generator=torch.Generator().manual_seed(9)).to(x.device).to(x.dtype)); x[:, 3:] = 0
    ref = a.features(x).clone()
    if 'flags' in want:
        b = section_flags(x, ref)
        section_hostword(b, x, ref)
    if 'surface' in want:
        section_surface(False)
        section_surface(True)
    if 'pipeline' in want:
        section_pipeline()
        section_pipeline(exchange='abi', records_only=True)
    for w in sorted(want):
        if w.startswith('memory'):
            section_memory(w.split(':')[1] if ':' in w else 'S2')
    print('CHILD-DONE', flush=True)


if __name__ == '__main__':
    main()
