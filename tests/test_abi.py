"""CPU-side checks: the C-ABI library builds/loads here (no GPU) and exports every symbol include/pam.h declares; host
logic that needs no device."""
import ctypes
import os
import re

import numpy as np
import pytest

import pam
from pam import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'pam.h')).read()
    return sorted(set(re.findall(r'\b(pam_[a-z_0-9]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libpam_hip.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_params_struct_matches_header():
    assert ctypes.sizeof(_lib.PamParams) == 7 * 8 + 6 * 4 + (16 + 16 + 64 + 4) * 8
    p = _lib.make_params(synth.MATCHER_CFG['Shelf'], 0.5)
    assert p.n_taps_body == 2 and p.n_taps_arm == 4              # radius int(4*0.3+.5)=1, int(4*0.8+.5)=3
    assert abs(sum([p.taps_body[0]] + [2 * p.taps_body[i] for i in range(1, 2)]) - 1) < 1e-15
    assert p.w_lambda_t[0] == 1.0 and abs(p.w_lambda_t[3] - np.exp(-15.0)) < 1e-20
    assert p.count_gate == 10 and p.max_age == 10 and p.n_init == 3


def test_gaussian_taps_match_scipy():
    from scipy.ndimage import gaussian_filter1d
    for sigma in (0.3, 0.6, 0.8, 1.7):
        w = _lib.gaussian_taps(sigma)
        r = len(w) - 1
        imp = np.zeros(2 * r + 1); imp[r] = 1.0
        k = gaussian_filter1d(imp, sigma=sigma, mode='constant')
        np.testing.assert_allclose(k[r:], w, rtol=0, atol=1e-16)


def test_unpack_dump_layout():
    """a2: (x, y, .) dump rows become (y, x, score) float64 (ivclabpose.py:236-244)."""
    from pam.ivclabpose import ivclabpose
    seq = synth.make_sequence('S1', n_frames=2, seed=0)
    pbl, dr = synth.to_dump_results(seq['frames'][1])
    poses = ivclabpose._unpack(dr)
    for v, d in enumerate(seq['frames'][1]):
        assert poses[v].dtype == np.float64 and poses[v].shape == (len(d), 17, 3)
        assert np.array_equal(poses[v][:, :, 0], d[:, :, 1]) and np.array_equal(poses[v][:, :, 1], d[:, :, 0])
        assert np.array_equal(poses[v][:, :, 2], d[:, :, 2])
    assert ivclabpose._unpack([[], []])[0].shape == (0, 17, 3)


def test_hrnet_definition():
    from pam import hrnet
    assert abs(hrnet.count_flops() / 1e9 - 70.6) < 0.1            # 35.3 GMAC per 384x288 crop
    m = hrnet.PoseHighResolutionNet()
    keys = m.state_dict().keys()
    for k in ('conv1.weight', 'layer1.0.downsample.0.weight', 'transition1.1.0.0.weight', 'stage2.0.branches.1.3.conv2.weight',
              'stage3.3.fuse_layers.2.0.1.0.weight', 'transition3.3.0.0.weight', 'stage4.2.fuse_layers.0.3.0.weight',
              'final_layer.bias'):
        assert k in keys, k
    assert not any(k.startswith('stage4.2.fuse_layers.1') for k in keys)


def test_bench_counts_gpus_without_touching_hip(monkeypatch):
    """bench.py's self-launching parent must learn the device count from the environment / KFD topology, never from HIP."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2')
    assert b.visible_gpu_count() == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    assert b.visible_gpu_count() == 0
    monkeypatch.delenv('HIP_VISIBLE_DEVICES')
    monkeypatch.delenv('ROCR_VISIBLE_DEVICES', raising=False); monkeypatch.delenv('CUDA_VISIBLE_DEVICES', raising=False)
    assert b.visible_gpu_count() >= 0                   # no KFD here: falls back without raising
    assert b.algorithmic_bytes_per_frame(5, 4, 4, 5, 11) == 39520      # SURVEY 8d's Shelf figure


def test_one_hip_runtime_whatever_the_import_order():
    """__graft_entry__.build() loads the library before anything imported torch; the process must still end up with a
    single libamdhip64 mapped (two runtimes: the second to initialise reports 'no ROCm-capable device')."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import pam\nfrom pam import _lib\n_lib.load()\nimport torch\n"
            "libs = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}\n"
            "print(len(libs), sorted(libs))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == '1', out.stdout


def test_bench_inputs_of_every_rank_for_the_shelf_frame_on_eight_ranks():
    """bench.py --gpus 8 quotes `value` on the Shelf-like frame: 5 views over 8 ranks leave three ranks without a camera.  Their inputs
    must still be well-formed (one padded empty record, no crops) and the ranks' crops must add up to the frame's, in both partitions."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    nF, world, max_dets = 3, 8, 8
    wl = b.setup_workload(synth, 'S2', nF)
    assert len(wl['cams']) == 5 and wl['meta']['C'] == 5
    for shard in ('views', 'crops'):
        per_rank = [b.build_inputs(torch, synth, wl['seq'], 'S2', max_dets, world, r, shard, torch.device('cpu'), nF) for r in range(world)]
        for t in range(nF):
            assert sum(i['local_crops'][t] for i in per_rank) == per_rank[0]['crops_per_frame'][t] == 20
        assert per_rank[0]['parts'] == per_rank[world - 1]['parts'] and sum(per_rank[0]['parts']) == 20
        if shard == 'views':
            idle = [i for i in per_rank if not i['mine']]
            assert len(idle) == 3
            for i in idle:
                e = i['per_frame'][0]
                assert i['frames'] == [] and int(i['ptrs'].numel()) == 1 and e['vl'].numel() == 0 and e['nd'].numel() == 0
                assert tuple(e['bx'].shape) == (0, 4) and tuple(e['dd'].shape) == (1, max_dets, 17, 3)


def test_algorithmic_work_is_executor_independent_and_launch_counts_follow_the_configuration():
    """bench.py's roofline inputs (hrnet.algorithmic_work, a shape-only walk): the algorithmic bytes and FLOPs of a forward do not depend
    on what the executor fuses -- un-fusing cannot raise a roofline fraction -- while launches and bytes-as-executed do; the
    counts are the ones bench.py asserts against the recorded families."""
    from pam import hrnet, hrnet_hip
    w = {name: hrnet.algorithmic_work(20, config=name) for name in hrnet_hip.HipHRNet.CONFIGS}
    a, b = w['fused48_fused96'], w['resident48_streamed96']
    assert a['bytes'] == b['bytes'] and a['flops'] == b['flops']
    assert abs(a['flops'] / (20 * hrnet.count_flops()) - 1) < 5e-3   # the stem's padded input channels (8 for 3) are the difference
    assert a["launches"] == 221 and b["launches"] == 253 and a['bytes_as_executed'] < b['bytes_as_executed']
    c = w['fused48_fused96_fsum']                                    # the small-forward configuration: the 1x1 products inside the sum launches
    assert c['launches'] == 203 and c['bytes'] == a['bytes'] and c['flops'] == a['flops'] and c['bytes_as_executed'] < a['bytes_as_executed']
    d = w['fused48_fused96_fsum_s32']                                # + 32-channel slabs in the deep branches: the same launches and work
    assert d['launches'] == 203 and d['bytes'] == c['bytes'] and d['flops'] == c['flops']
    assert a['bytes'] < a['bytes_as_executed']                       # the deep branches' block interiors are traffic only as executed
    assert hrnet.algorithmic_work(40)['bytes'] > 1.9 * a['bytes']    # activations scale with the crops, the weights do not


def test_conv64_image_follows_the_layout_pam_h_documents():
    """The [9 taps][64 rows][64 K] weight image of the fused stem / Bottleneck kernels (include/pam.h, pam_stem_fused_nhwc_bf16): row
    16 j + q of a tap = output channel 32 (j >> 1) + 8 (q >> 2) + 4 (j & 1) + (q & 3); the row's 16-byte piece at position p holds input
    channels 8 c .. 8 c + 7 with c = p ^ ((q >> 1) & 7).  Checked element by element against a weight tensor whose every entry encodes its
    own index (host logic only: packs on the CPU)."""
    import torch
    import torch.nn as nn
    from pam import hrnet_hip
    conv = nn.Conv2d(64, 64, 3, 1, 1, bias=True)
    co, ci, ky, kx = torch.meshgrid(torch.arange(64), torch.arange(64), torch.arange(3), torch.arange(3), indexing='ij')
    with torch.no_grad():
        conv.weight.copy_((co * 64 + ci).float() + (ky * 3 + kx).float() / 16.0)     # exact in bf16? no: compare after the same rounding
    img = hrnet_hip.conv64_image(conv, torch.device('cpu')).float().reshape(9, 64, 8, 8)
    want = conv.weight.detach().to(torch.bfloat16).float()
    for tap in (0, 4, 8):
        for row in (0, 5, 17, 38, 63):
            j, q = row // 16, row % 16
            ch = 32 * (j >> 1) + 8 * (q >> 2) + 4 * (j & 1) + (q & 3)
            for p in range(8):
                c = p ^ ((q >> 1) & 7)
                assert torch.equal(img[tap, row, p], want[ch, 8 * c:8 * c + 8, tap // 3, tap % 3]), (tap, row, p)
    # every output channel appears exactly once per tap
    rows = torch.arange(64)
    chs = 32 * ((rows // 16) >> 1) + 8 * ((rows % 16) >> 2) + 4 * ((rows // 16) & 1) + (rows % 16 & 3)
    assert sorted(chs.tolist()) == list(range(64))


def test_executor_rule_by_crop_count_and_configurations_set_the_same_switches():
    """HRNetPose.config_for (host logic): up to 12 crops fused sums + 32-channel slabs in the deep branches, up to 20 fused sums, above the
    1x1 products as launches of their own; without autotune one configuration for every count.  Every configuration sets the same
    switches, so going from one to another leaves nothing behind."""
    from pam import hrnet, hrnet_hip
    net = hrnet.HRNetPose.__new__(hrnet.HRNetPose)
    net.hip = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet); net.autotune = True; net.tuned = {}
    got = [net.config_for(n) for n in (1, 12, 13, 20, 21, 217)]
    assert got == ['fused48_fused96_fsum_s32'] * 2 + ['fused48_fused96_fsum'] * 2 + ['fused48_fused96'] * 2
    assert net.tuned[20] == {'choice': 'fused48_fused96_fsum'}
    net.autotune = False
    assert {net.config_for(n) for n in (1, 12, 20, 217)} == {'fused48_fused96'}
    keys = {frozenset(c) for c in hrnet_hip.HipHRNet.CONFIGS.values()}
    assert len(keys) == 1 and {'block2', 'fused_sums', 'slab32'} <= set(next(iter(keys)))
    h = hrnet_hip.HipHRNet.__new__(hrnet_hip.HipHRNet)
    h.apply_config('fused48_fused96_fsum_s32'); assert h.slab32 is True and h.fused_sums is True
    h.apply_config('fused48_fused96'); assert h.slab32 is False and h.fused_sums is False and h.config_name == 'fused48_fused96'


def test_disable_flag_sync_swaps_to_the_stream_event_forms():
    """HRNetPose.disable_flag_sync (host logic; FramePipeline(pose_streams=2), ranks sharing a device, a gate time-out): a forward that has
    both forms keeps its stream-event capture, a flagged-only capture leaves the cache (re-captured at its next use), captures that
    already use stream events stay; nothing is destroyed."""
    from pam import hrnet
    net = hrnet.HRNetPose.__new__(hrnet.HRNetPose)
    k1, k2, k3 = (20, 'features', 0), (8, 'features', 0), (4, 'features', 0)
    net._graphs = {k1: ('F1',), k2: ('F2',), k3: ('E3',)}
    net._alt = {k1: {'flags': ('F1',), 'events': ('E1',)}, k2: None, k3: None}
    net.flag_synced = {k1: True, k2: True, k3: False}
    net.flag_timing = {20: dict(kept={'serial': True, 'throughput': True})}
    net._dead_graphs = []
    net.disable_flag_sync()
    assert net._graphs == {k1: ('E1',), k3: ('E3',)} and net.flag_synced == {k1: False, k3: False}
    assert net._dead_graphs == [('F1',), ('F2',)] and net._flag_sync_failed is True and net._alt.get(k1) is None
    assert net._flag_sync_ok.__func__ is hrnet.HRNetPose._flag_sync_ok
