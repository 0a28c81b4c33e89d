"""DumpResults (what HRNetPose.predict returns): the device-side fast path of PersonTrack_Project3DPose may only be taken while the
dicts are exactly what predict() produced -- the reference reads 'keypoints' AND 'keypoints_score' from them
(/root/reference/src/ivclabpose.py:236-244), so any caller edit must fall back to the dicts."""
import numpy as np

from pam.hrnet import DumpResults


def _dump():
    rng = np.random.default_rng(0)
    out = DumpResults([[], []])
    for v in range(2):
        for _ in range(2):
            kp = rng.uniform(0, 300, (17, 3))
            out[v].append(dict(bbox=[0, 0, 10, 10], keypoints=kp.reshape(-1).tolist(), keypoints_score=kp[:, 2].tolist(), feature=[]))
    out.attach(object(), object(), [np.zeros((2, 17, 3))] * 2)
    return out


def test_untouched_dump_is_valid():
    assert _dump().device_valid()


def test_interior_keypoint_edit_is_detected():
    d = _dump()
    d[1][0]['keypoints'][25] += 1.0                  # not the first, not the last element
    assert not d.device_valid()


def test_rescored_joint_is_detected():
    d = _dump()
    d[0][1]['keypoints_score'][9] = 0.0              # a caller masks a wrist
    assert not d.device_valid()


def test_replaced_lists_and_persons_are_detected():
    d = _dump()
    d[0][0]['keypoints_score'] = list(d[0][0]['keypoints_score'])     # same values, another object: not provably untouched
    assert not d.device_valid()
    d = _dump()
    d[1].pop()
    assert not d.device_valid()
    d = _dump()
    d[0][0] = dict(d[0][0])
    assert not d.device_valid()


# ---- lazy form: predict() only enqueues the device -> host copy; the dicts appear at the first access ------------------------------------
class _Event(object):
    waited = 0

    def synchronize(self):
        self.waited += 1


def _pending_dump():
    import torch
    rng = np.random.default_rng(1)
    kp = rng.uniform(0, 300, (3, 17, 3)).astype(np.float32)           # rows (x, y, score) of three persons: views 0, 0, 1
    ev = _Event()
    out = DumpResults([[], []])
    out.attach_pending(object(), object(), torch.from_numpy(kp), ev, [0, 0, 1], [[0, 0, 5, 5], [1, 1, 5, 5], [2, 2, 5, 5]], [2, 1])
    return out, kp, ev


def test_pending_dump_is_valid_without_waiting_and_has_its_length():
    d, kp, ev = _pending_dump()
    assert len(d) == 2 and d.device_valid() and ev.waited == 0


def test_first_access_builds_the_reference_dicts():
    d, kp, ev = _pending_dump()
    assert [len(v) for v in d] == [2, 1] and ev.waited >= 1            # iteration is an access
    assert d[0][1]['bbox'] == [1, 1, 5, 5] and d[1][0]['feature'] == []
    assert np.array_equal(np.array(d[0][1]['keypoints']).reshape(17, 3), kp[1].astype(np.float64))
    assert d[1][0]['keypoints_score'] == kp[2][:, 2].astype(np.float64).tolist()
    assert d.device_valid()
    d[0][0]['keypoints_score'][3] = 0.0                                # an edit after the access is detected as before
    assert not d.device_valid()


def test_poses_host_needs_no_dicts_and_matches_unpack():
    from pam.ivclabpose import ivclabpose
    d, kp, ev = _pending_dump()
    ph = d.poses_host                                                  # (y, x, score) rows per view, straight from the copied block
    assert d._pending is not None and [p.shape for p in ph] == [(2, 17, 3), (1, 17, 3)]
    un = ivclabpose._unpack(d)
    for a, b in zip(ph, un):
        assert np.array_equal(a, b)


def test_pending_dump_survives_copy_pickle_and_json():
    import copy, json, pickle
    for f in (lambda d: list(d), lambda d: copy.copy(d), lambda d: pickle.loads(pickle.dumps(d)), lambda d: json.loads(json.dumps(d)), lambda d: d.copy()):
        d, kp, ev = _pending_dump()
        got = f(d)
        assert [len(v) for v in got] == [2, 1] and got[1][0]['bbox'] == [2, 2, 5, 5]


def test_pending_dump_on_the_right_of_a_list_concatenation():
    """`[] + dump`: list.__add__ of the LEFT operand reads a list subclass's items through a C fast path; DumpResults.__radd__ gets there first."""
    d, kp, ev = _pending_dump()
    got = [] + d
    assert type(got) is list and [len(v) for v in got] == [2, 1] and got[0][1]['bbox'] == [1, 1, 5, 5] and ev.waited >= 1
    d, kp, ev = _pending_dump()
    got = [['x']] + d
    assert got[0] == ['x'] and [len(v) for v in got[1:]] == [2, 1]
