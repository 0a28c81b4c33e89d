"""N>1 path on CPU: world_size-2 gloo processes exercise the view partition + the single all-gather per frame."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pam
from pam.distributed import CropGather, ViewGather, crop_partition, gather_crop_keypoints, view_partition


def test_view_partition():
    assert view_partition(5, 1) == [[0, 1, 2, 3, 4]]
    assert view_partition(5, 2) == [[0, 1, 2], [3, 4]]
    assert [len(p) for p in view_partition(31, 8)] == [4, 4, 4, 4, 4, 4, 4, 3]
    p = view_partition(5, 8)
    assert sorted(v for q in p for v in q) == list(range(5)) and sum(1 for q in p if not q) == 3
    for C in (3, 5, 31):
        for W in (1, 2, 4, 8):
            q = view_partition(C, W)
            assert sorted(v for r in q for v in r) == list(range(C))
            assert all(r == sorted(r) for r in q)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, C, max_dets, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(0)                       # same stream on all ranks = the "global" truth
    n_det = rng.integers(0, max_dets + 1, size=(3, C)).astype(np.int32)
    det = rng.normal(size=(3, C, max_dets, 17, 3))
    g = ViewGather(C, max_dets, world, rank, torch.device('cpu'))
    ok = True
    for t in range(3):
        mine = g.mine
        nd, dd = g.gather(torch.tensor(n_det[t, mine], dtype=torch.int32), torch.tensor(det[t][mine]).reshape(len(mine), max_dets, 17, 3))
        ok &= bool(np.array_equal(nd.numpy(), n_det[t])) and bool(np.array_equal(dd.numpy(), det[t]))
        # what the frame kernel reads in place (pam_frame_dev_views): view v's record is recv[rows[v]] -- max_dets detection rows, then
        # the count in the first double of the extra row
        for v in range(C):
            rec = g.recv[int(g.rows[v])].numpy()
            ok &= bool(np.array_equal(rec[:max_dets], det[t][v])) and rec[max_dets, 0, 0] == n_det[t, v]
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_world2_gloo():
    for C in (5, 3):
        mgr = mp.Manager()
        ret = mgr.dict()
        port = _free_port()
        mp.spawn(_worker, args=(2, port, C, 4, ret), nprocs=2, join=True)
        assert ret[0] and ret[1]


def test_all_gather_with_a_rank_that_owns_no_view_gloo():
    """Shelf's 5 views on 8 ranks leave 3 ranks without a camera (bench.py --gpus 8); here 2 views on 3 ranks: the idle rank sends a
    padded empty record and still ends up with every view's keypoints."""
    mgr = mp.Manager()
    ret = mgr.dict()
    assert [len(p) for p in view_partition(2, 3)].count(0) == 1
    mp.spawn(_worker, args=(3, _free_port(), 2, 4, ret), nprocs=3, join=True)
    assert ret[0] and ret[1] and ret[2]


def test_crop_partition_is_balanced_and_contiguous():
    for n in (0, 1, 7, 20, 217):
        for W in (1, 2, 4, 8):
            parts = crop_partition(n, W)
            assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert [b - a for a, b in crop_partition(20, 8)] == [2, 3, 2, 3, 2, 3, 2, 3]


def _crop_worker(rank, world, port, C, max_dets, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(1)                       # same stream on all ranks = the "global" truth
    g = CropGather(C, max_dets, world, rank, torch.device('cpu'))
    ok = True
    for t in range(3):
        n_det = rng.integers(0, max_dets + 1, size=C)
        det = rng.normal(size=(C, max_dets, 17, 3))
        view_of = [v for v in range(C) for _ in range(n_det[v])]
        slot_of = [s for v in range(C) for s in range(n_det[v])]
        select, parts = CropGather.select_index(view_of, slot_of, C, max_dets, world)
        a, b = parts[rank]
        g.send.fill_(float(100 + rank))                  # rows this rank does not own hold garbage
        for i in range(a, b):
            g.send[view_of[i], slot_of[i]] = torch.tensor(det[view_of[i], slot_of[i]])
        out = g.gather(torch.tensor(select)).numpy()
        for v in range(C):
            ok &= bool(np.array_equal(out[v, :n_det[v]], det[v, :n_det[v]]))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_crop_gather_world2_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_crop_worker, args=(2, _free_port(), 5, 4, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def _kp_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(2)
    ok = True
    for n in (0, 1, 5, 20):                              # incl. a rank with nothing to decode
        full = torch.tensor(rng.normal(size=(n, 17, 3)), dtype=torch.float32)
        a, b = crop_partition(n, world)[rank]
        out = gather_crop_keypoints(full[a:b].clone(), n, world, rank)
        ok &= bool(torch.equal(out, full))
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_surface_keypoint_gather_world2_gloo():
    """HRNetPose.predict's exchange under torch.distributed (the drop-in surface on several GPUs): every rank ends with all rows."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_kp_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def _void_worker(rank, world, port, ret):
    """The producer's "these keypoints are void" word through both exchanges: raised on ONE rank it must be seen by every rank -- in every
    view record's count row (what the frame kernel checks, csrc/pam_tracker.hip) and as CropGather.void_any -- and be gone the frame after
    it was lowered; and the device-sharing rule decided from identities."""
    from pam.distributed import ranks_share_a_device
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    C, md = 5, 4
    cpu = torch.device('cpu')
    vg, cg = ViewGather(C, md, world, rank, cpu), CropGather(C, md, world, rank, cpu)
    word = torch.zeros(1, dtype=torch.int32)
    select = torch.tensor(CropGather.select_index([0, 1, 2], [0, 0, 0], C, md, world)[0])
    ok = True
    for t, raised_on in enumerate((None, 1, None, 0, None)):
        word.fill_(1 if raised_on == rank else 0)
        recv = vg.exchange(torch.ones(len(vg.mine), dtype=torch.int32), word)
        flags = [float(recv[int(vg.rows[v]), md, 0, 1]) for v in range(C)]
        owner = [r for v in range(C) for r, p in enumerate(vg.parts) if v in p]
        ok &= flags == [1.0 if owner[v] == raised_on else 0.0 for v in range(C)]
        ok &= all(float(recv[int(vg.rows[v]), md, 0, 0]) == 1.0 for v in range(C))           # the counts beside them are untouched
        cg.gather(select, word)
        ok &= int(cg.void_any) == (0 if raised_on is None else 1)
    # two ranks, two devices -> not shared; two ranks, one identity -> shared (what world > device_count() could not tell apart under
    # HIP_VISIBLE_DEVICES: there every rank counts one device)
    ok &= ranks_share_a_device(cpu, identity=('host', 'gpu-%d' % rank)) is False
    ok &= ranks_share_a_device(cpu, identity=('host', 'gpu-0')) is True
    ok &= ranks_share_a_device(cpu, identity=('host-%d' % rank, 'gpu-0')) is False
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_void_word_travels_with_both_exchanges_and_device_sharing_is_decided_from_identities_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_void_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] and ret[1]
