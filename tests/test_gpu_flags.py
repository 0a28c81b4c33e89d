"""GPU: device-side flags of the captured HRNet forward (csrc/pam_sync.hip) and what happens when a gate gives up -- all of it in ONE fresh
child process (tests/flag_child.py says why), whose marker lines the tests below assert on."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def child():
    out = subprocess.run([sys.executable, os.path.join(HERE, 'flag_child.py')], capture_output=True, text=True, timeout=400)
    sys.stderr.write(out.stdout[-3000:])
    return out


def _has(child, marker):
    assert marker in child.stdout, (marker, child.returncode, child.stdout[-2000:], child.stderr[-4000:])


def test_replays_order_their_branch_streams_by_device_flags(child):
    """Captured forwards meet at the module ends through counters in device memory instead of stream events: same features as the eager
    forward (which uses stream events), error words zero over many replays; the capture-time race (interleaved medians, 1 % margin)."""
    _has(child, 'FLAGS-OK')


def test_a_time_out_seen_through_the_host_word_switches_to_stream_events(child):
    _has(child, 'HOSTWORD-OK')


def test_gate_time_out_cannot_emit_a_frame_on_the_drop_in_surface(child):
    """Golden trace S2 through ivclabpose with the real forward in front of the tracker and a 0-us gate bound on frame 40: the frame
    kernel refuses the frame (input guard), the forward is re-run with stream events, and EVERY frame's 9-tuple and tracker state equal
    the reference's -- dump passed straight on, and dump read by the host first."""
    _has(child, 'SURFACE-DEVICE-OK')
    _has(child, 'SURFACE-HOST-OK')


def test_gate_time_out_inside_the_pipelined_loop_skips_frames_and_names_where_to_resume(child):
    _has(child, 'PIPELINE-OK')


def test_the_void_flag_inside_the_exchanged_records_protects_every_replica(child):
    """Row (e)'s default path in one process -- flags on, view records through the library's RCCL all-gather (one-rank communicator), the
    frame kernel reading the gathered buffer in place -- with the handle's own guard removed: the flag that travels in the records alone
    makes the tracker skip the void frames, and the re-submitted run equals a run that never saw a time-out."""
    _has(child, 'PIPELINE-RECORDS-OK')


def test_child_ran_to_its_end(child):
    assert child.returncode == 0 and 'CHILD-DONE' in child.stdout, child.stderr[-3000:]


def test_prewarmed_pipeline_memory_is_reported():
    """Replay cache + arena of a prewarmed S2 pipeline with both forms of every flagged bucket alive (a fresh child of its own)."""
    out = subprocess.run([sys.executable, os.path.join(HERE, 'flag_child.py'), 'memory'], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'MEMORY-OK' in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith('MEMORY-OK')][0]
    sys.stderr.write(line + '\n')
    mb = float(line.split('device_MB=')[1].split()[0])
    assert mb < 4096, line                               # 1.3 GB on round 6's boxes (5 buckets x 2 forms, 0.7 GB of activation arena)


def test_a_capture_whose_first_replay_times_out_is_replaced_by_stream_events():
    """In this (stream-rich) process: a capture whose first replay raises the error word (a bound of 1 us, which every waiting gate
    exceeds) is replaced by one with stream events, with the same result, and nothing stays raised."""
    from pam import hrnet
    a = hrnet.HRNetPose(48, 17, None, use_graph=False)
    x = a.input_buffer(5)
    x.copy_(torch.randn(x.shape, generator=torch.Generator().manual_seed(9)).to(x.device).to(x.dtype)); x[:, 3:] = 0
    ref = a.features(x).clone()
    c = hrnet.HRNetPose(48, 17, None, use_graph=True)
    c.hip.set_flag_limit(1)
    y2 = c.features(x).clone()
    torch.cuda.synchronize()
    assert c.flag_synced[(5, 'features', 0)] is False and c.captures == 1 and len(c._dead_graphs) == 1 and torch.equal(ref, y2)
    assert c._flag_sync_ok() is False                   # and the object stays with stream events
    assert int(c._flag_host_np[0]) == 0 and int(c.void_word.item()) == 0 and not c.void_pending
