"""GPU: the crop-sharded / tracker-overlapped form of the device pipeline gives the same records as the plain one."""
import numpy as np
import pytest
import torch

import pam

pytestmark = pytest.mark.gpu


def _rig(size):
    from pam import synth
    from pam.ivclabpose import Camera, fundamental_matrices
    meta = synth.SIZES[size]
    seq = synth.make_sequence(size, n_frames=40, seed=3)
    cfg = dict(synth.MATCHER_CFG[synth.SIZE_TO_DATASET[size]]); conf = cfg.pop('CONF_THRESHOLD')
    P32 = seq['calib']['P'].astype(np.float32); K32 = seq['calib']['K'].astype(np.float32); RT32 = seq['calib']['RT'].astype(np.float32)
    Fm = fundamental_matrices(K32, RT32)
    cams = [Camera(j, P32[j], K32[j], RT32[j], Fm[j], w=meta['w'], h=meta['h']) for j in range(meta['C'])]
    return seq, cams, cfg, conf, meta


@pytest.mark.parametrize('overlap', [False, True])
def test_crop_mode_matches_view_mode(overlap):
    from pam import synth
    from pam.distributed import CropGather
    from pam.pipeline import FramePipeline
    seq, cams, cfg, conf, meta = _rig('S2')
    C, md = meta['C'], 8
    n_det_all, det_all = synth.pack_frames(seq['frames'], md)
    ref = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False)
    new = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False, shard='crops', overlap_tracker=overlap)
    dev = ref.device
    for t in range(len(seq['frames'])):
        nd = torch.tensor(n_det_all[t], dtype=torch.int32, device=dev)
        dd = torch.tensor(det_all[t], dtype=torch.float64, device=dev)
        ref.track_step(t, nd, dd)
        a = ref.results()
        vl = [v for v in range(C) for _ in range(n_det_all[t][v])]
        sl = [s for v in range(C) for s in range(n_det_all[t][v])]
        select, _ = CropGather.select_index(vl, sl, C, md, 1)
        new.crop_gather.send.copy_(dd)
        new.track_step_crops(t, nd, torch.tensor(select, dtype=torch.int64, device=dev))
        b = new.results()
        assert a['n_tracks'] == b['n_tracks']
        for ta, tb in zip(a['tracks'], b['tracks']):
            assert ta['track_id'] == tb['track_id'] and ta['emitted'] == tb['emitted']
            if ta['emitted']:
                assert np.array_equal(ta['pose3d'], tb['pose3d'])


def test_overlap_without_per_frame_sync_matches_serial():
    """K frames issued back to back with the tracker of frame t on its own stream under frame t + 1 and NO host synchronisation in
    between (the bench's timed loop): every write to the exchange buffer must be ordered behind the previous frame's readers
    (FramePipeline.write_send / wait_track).  Final records and every intermediate frame's record equal the serial run's."""
    from pam import synth
    from pam.distributed import CropGather
    from pam.pipeline import FramePipeline
    seq, cams, cfg, conf, meta = _rig('S2')
    C, md = meta['C'], 8
    n_det_all, det_all = synth.pack_frames(seq['frames'], md)
    ser = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False, shard='crops', overlap_tracker=False)
    ovl = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False, shard='crops', overlap_tracker=True)
    dev = ser.device
    nd = [torch.tensor(n_det_all[t], dtype=torch.int32, device=dev) for t in range(len(seq['frames']))]
    dd = [torch.tensor(det_all[t], dtype=torch.float64, device=dev) for t in range(len(seq['frames']))]
    sel = []
    for t in range(len(seq['frames'])):
        vl = [v for v in range(C) for _ in range(n_det_all[t][v])]
        sl = [s for v in range(C) for s in range(n_det_all[t][v])]
        sel.append(torch.tensor(CropGather.select_index(vl, sl, C, md, 1)[0], dtype=torch.int64, device=dev))
    # a long kernel on the pose stream in front of every frame keeps the tracker stream behind, as the conv stack does in the bench
    ballast = torch.randn((4096, 4096), device=dev)
    recs = []
    for t in range(len(seq['frames'])):
        ser.write_send(dd[t]); ser.track_step_crops(t, nd[t], sel[t])
        recs.append(ser.results())
    for t in range(len(seq['frames'])):
        _ = ballast @ ballast
        ovl.write_send(dd[t]); ovl.track_step_crops(t, nd[t], sel[t])
    last = ovl.results()
    assert last['n_tracks'] == recs[-1]['n_tracks'] and last['frame_id'] == recs[-1]['frame_id']
    for ta, tb in zip(recs[-1]['tracks'], last['tracks']):
        assert ta['track_id'] == tb['track_id'] and ta['hits'] == tb['hits'] and ta['age'] == tb['age'] and ta['emitted'] == tb['emitted']
        assert np.array_equal(ta['pose3d'], tb['pose3d']) and np.array_equal(ta['velocity'], tb['velocity'])


def test_allgather_keypoints_through_the_c_abi():
    """pam_comm_* + pam_allgather_keypoints on a one-rank communicator (a single-GPU box cannot hold two RCCL ranks): the record
    rows come back unchanged, and a FramePipeline whose exchange runs through the C ABI tracks exactly like the torch one."""
    import ctypes as C
    from pam import synth, _lib
    from pam.distributed import AbiComm
    from pam.pipeline import FramePipeline
    seq, cams, cfg, conf, meta = _rig('S1')
    Cn, md = meta['C'], 8
    n_det_all, det_all = synth.pack_frames(seq['frames'], md)
    ref = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False)
    abi = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False, exchange='abi')
    assert abi.comm is not None and abi.gather.abi is not None
    dev = ref.device
    for t in range(len(seq['frames'])):
        nd = torch.tensor(n_det_all[t], dtype=torch.int32, device=dev)
        dd = torch.tensor(det_all[t], dtype=torch.float64, device=dev)
        ref.track_step(t, nd, dd); a = ref.results()
        abi.track_step(t, nd, dd); b = abi.results()
        assert torch.equal(abi.gather.recv[:Cn], abi.gather.send[:Cn])
        assert a['n_tracks'] == b['n_tracks']
        for ta, tb in zip(a['tracks'], b['tracks']):
            assert ta['track_id'] == tb['track_id'] and ta['emitted'] == tb['emitted'] and np.array_equal(ta['pose3d'], tb['pose3d'])
    abi.comm.close()


def _views_worker(rank, world, port, ret, size='S2', n_frames=None):
    import os
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)          # two ranks on ONE device: RCCL refuses that, gloo carries the gather
    from pam import synth
    from pam.pipeline import FramePipeline
    seq, cams, cfg, conf, meta = _rig(size)
    if n_frames:
        seq['frames'] = seq['frames'][:n_frames]
    md = 8
    n_det_all, det_all = synth.pack_frames(seq['frames'], md)
    ovl = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, max_tracks=16, hrnet=False, shard='views', overlap_tracker=True,
                        world=world, rank=rank)
    dev, mine = ovl.device, ovl.mine
    nd = [torch.tensor(n_det_all[t][mine], dtype=torch.int32, device=dev) for t in range(len(seq['frames']))]
    dd = [torch.tensor(det_all[t][mine], dtype=torch.float64, device=dev) for t in range(len(seq['frames']))]
    ballast = torch.randn((2048, 2048), device=dev)
    for t in range(len(seq['frames'])):                                    # no host synchronisation between frames
        _ = ballast @ ballast
        ovl.write_local(dd[t]); ovl.track_step(t, nd[t])
    last = ovl.results()
    ser = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, max_tracks=16, hrnet=False, shard='views', overlap_tracker=False)
    for t in range(len(seq['frames'])):
        ser.track_step(t, torch.tensor(n_det_all[t], dtype=torch.int32, device=dev), torch.tensor(det_all[t], dtype=torch.float64, device=dev))
        ref = ser.results()
    ok = last['n_tracks'] == ref['n_tracks'] and last['status'] == 0 and last['status_sticky'] == 0 and last['frame_id'] == ref['frame_id'] and ref['n_tracks'] > 0
    for ta, tb in zip(ref['tracks'], last['tracks']):
        ok &= ta['track_id'] == tb['track_id'] and ta['hits'] == tb['hits'] and ta['age'] == tb['age'] and ta['emitted'] == tb['emitted']
        ok &= bool(np.array_equal(ta['pose3d'], tb['pose3d'])) and bool(np.array_equal(ta['velocity'], tb['velocity']))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_view_sharded_overlapped_pipeline_two_ranks_one_device():
    """The N > 1 default of bench.py -- camera views partitioned over the ranks, ONE all-gather per frame, the frame kernel reading the
    gathered records in place, exchange + tracker of frame t on their own stream under frame t + 1 -- with two ranks (gloo, both on
    this device), K frames and no host synchronisation: every rank's final record equals the single-process serial run's."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_views_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_view_sharded_pipeline_eight_ranks_one_device_panoptic31():
    """BASELINE config 5's partition, functionally: the 31-camera S4 rig over EIGHT ranks (4,4,4,4,4,4,4,3 views: the last rank's record
    block is padded), gloo, all on this device -- the row map view -> gathered record and the padded block drive pam_frame_dev_views
    (k_frame<1024>), overlapped, no host synchronisation; every rank's final record (ids, hits, ages, 3D poses, velocities) equals the
    single-process serial run's.  A functional check: eight processes time-slice one GPU, so this says nothing about scaling -- no
    multi-GPU box exists in this environment and no hardware curve has been measured."""
    import socket
    import torch.multiprocessing as mp
    from pam.distributed import view_partition
    assert [len(p) for p in view_partition(31, 8)] == [4, 4, 4, 4, 4, 4, 4, 3]
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ret = mp.Manager().dict()
    mp.spawn(_views_worker, args=(8, port, ret, 'S4', 6), nprocs=8, join=True)
    assert all(ret[r] for r in range(8)), dict(ret)


def test_view_records_in_place_match_packed_input():
    """pam_frame_dev_views (records + row map) and pam_frame_dev (packed n_det / det) are the same step: identical records."""
    from pam import synth
    from pam.pipeline import FramePipeline
    seq, cams, cfg, conf, meta = _rig('S2')
    md = 8
    n_det_all, det_all = synth.pack_frames(seq['frames'], md)
    a = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False, shard='views')
    b = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False, shard='crops')
    dev = a.device
    for t in range(len(seq['frames'])):
        nd = torch.tensor(n_det_all[t], dtype=torch.int32, device=dev)
        dd = torch.tensor(det_all[t], dtype=torch.float64, device=dev)
        a.track_step(t, nd, dd); ra = a.results()
        st = b.stream_ptr()
        b.handle.frame_dev(st, t, nd.data_ptr(), dd.data_ptr()); b.handle.fetch(st, b.out_i.numpy(), b.out_d.numpy()); rb = b.results()
        assert np.array_equal(a.out_i.numpy(), b.out_i.numpy())
        k = a.handle.layout.dbl_hdr_words
        assert np.array_equal(a.out_d.numpy()[:, k:], b.out_d.numpy()[:, k:])


def test_prewarmed_pipeline_never_captures_inside_a_step():
    """FramePipeline(prewarm=True) captures the replay of every crop-count bucket of its rig at construction; afterwards frames with any
    number of crops run without a capture (HRNetPose.captures does not move), padded to their bucket by the crop kernel, and the records
    equal the un-bucketed pipeline's.  Prints what the warm cache costs (arena bytes, seconds)."""
    from pam import synth
    from pam.distributed import CropGather
    from pam.pipeline import FramePipeline
    seq, cams, cfg, conf, meta = _rig('S1')
    C, md = meta['C'], 4
    n_det_all, det_all = synth.pack_frames(seq['frames'], md)
    warm = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, shard='crops', prewarm=True, autotune=False)
    w = warm.warmed
    print('prewarm: buckets %s, %d captures in %.1f s, arena %.2f GiB' % (w['buckets'], w['captures'], w['seconds'], w['arena_bytes'] / 2 ** 30))
    assert w['buckets'] == [4, 8, 12] and w['captures'] == 3 and w['arena_bytes'] == warm.net.arena_bytes() > 0
    from pam import hrnet
    exact = hrnet.HRNetPose(48, 17, None, use_graph=False, max_dets=md)                # the same weights (seed 0), forwards of exactly n crops
    plain = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, shard='crops', net=exact)
    dev = warm.device
    g = torch.Generator().manual_seed(11)
    frames = [torch.randint(0, 256, (meta['h'], meta['w'], 3), dtype=torch.uint8, generator=g).to(dev) for _ in range(C)]
    ptrs = torch.tensor([f.data_ptr() for f in frames], dtype=torch.int64, device=dev)
    c0 = warm.net.captures
    seen = set()
    for t in range(24):
        vl, sl, bx = [], [], []
        for v in range(C):
            for s, kp in enumerate(seq['frames'][t][v][:((t + 2 * v) % (md + 1))]):   # 0 .. 3 persons per view, another mix every frame
                x0, y0, x1, y1 = kp[:, 0].min(), kp[:, 1].min(), kp[:, 0].max(), kp[:, 1].max()
                vl.append(v); sl.append(s); bx.append([x0, y0, max(x1 - x0, 8.0), max(y1 - y0, 8.0)])
        if not vl:
            continue
        seen.add(len(vl))
        tv = torch.tensor(vl, dtype=torch.int32, device=dev); ts = torch.tensor(sl, dtype=torch.int32, device=dev)
        tb = torch.tensor(bx, dtype=torch.float32, device=dev).reshape(-1, 4)
        warm.crop_gather.send.zero_()
        warm.pose_step_crops(ptrs, tv, ts, tb)
        assert warm.net.captures == c0, (t, len(vl))                                   # no capture inside a step
        a = warm.crop_gather.send.clone()
        plain.crop_gather.send.zero_()
        plain.pose_step_crops(ptrs, tv, ts, tb)                                        # un-bucketed: a forward of exactly len(vl) crops
        torch.cuda.synchronize()
        assert torch.equal(a, plain.crop_gather.send), t
    assert len(seen) >= 4, seen


def test_input_guard_skips_frames_leaves_the_state_untouched_and_names_the_first_void_frame():
    """pam_set_input_guard: while the producer's device word is up -- or any view record of a sharded frame carries the flag -- the frame
    kernel applies nothing: record status 16, no tracks, word [3] = the first frame of the run of void frames; once the word is down
    the re-submitted frames give exactly the records of a run that never saw it."""
    from pam import synth, _lib
    from pam.pipeline import FramePipeline
    seq, cams, cfg, conf, meta = _rig('S2')
    C, md = meta['C'], 8
    n_det_all, det_all = synth.pack_frames(seq['frames'], md)
    ref = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False)
    new = FramePipeline(cams, cfg, conf, (meta['h'], meta['w']), max_dets=md, hrnet=False)
    dev = ref.device
    word = torch.zeros(1, dtype=torch.int32, device=dev)
    new.handle.set_input_guard(word.data_ptr())
    F = len(seq['frames'])
    want = []
    for t in range(F):
        ref.track_step(t, torch.tensor(n_det_all[t], dtype=torch.int32, device=dev), torch.tensor(det_all[t], dtype=torch.float64, device=dev))
        want.append(ref.results())

    def step(t):
        new.track_step(t, torch.tensor(n_det_all[t], dtype=torch.int32, device=dev), torch.tensor(det_all[t], dtype=torch.float64, device=dev))
        return new.results()

    def same(a, b):
        assert a['n_tracks'] == b['n_tracks'] and a['status'] == b['status'] == 0 and a['frame_id'] == b['frame_id']
        for ta, tb in zip(a['tracks'], b['tracks']):
            assert ta['track_id'] == tb['track_id'] and ta['hits'] == tb['hits'] and ta['age'] == tb['age'] and ta['state'] == tb['state']
            assert np.array_equal(ta['pose3d'], tb['pose3d']) and np.array_equal(ta['velocity'], tb['velocity'])
    for t in range(12):
        same(step(t), want[t])
    word.fill_(1)                                            # the device word (one rank: the frame kernel reads it itself)
    for t in (12, 13, 14):
        with pytest.raises(_lib.FrameVoid) as e:
            step(t)
        assert (e.value.first, e.value.last) == (12, t)
        rec = new.handle.decode(0, new.out_i.numpy(), new.out_d.numpy())
        assert rec['status'] == _lib.ST_INPUT_VOID and rec['n_tracks'] == 0 and rec['status_sticky'] == 0
    word.zero_()
    for t in range(12, 20):
        same(step(t), want[t])
    new.gather.send[1, md, 0, 1] = 1.0                       # the flag inside a view record (what a rank's exchange carries)
    with pytest.raises(_lib.FrameVoid) as e:
        step(20)
    assert (e.value.first, e.value.last) == (20, 20)
    new.gather.send[1, md, 0, 1] = 0.0
    for t in range(20, F):
        same(step(t), want[t])


def test_device_identity_is_stable_and_specific():
    """distributed.device_identity: what ranks_share_a_device compares -- host name + every device-distinguishing field PyTorch reports."""
    from pam.distributed import device_identity
    a, b = device_identity(torch.device('cuda:0')), device_identity(torch.device('cuda:0'))
    assert a == b and len(a) == 2 and isinstance(a[1], str) and len(a[1]) > 8
