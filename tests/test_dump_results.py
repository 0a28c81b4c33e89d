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
