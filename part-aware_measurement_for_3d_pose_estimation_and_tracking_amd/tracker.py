"""Device-resident iterative tracker: host-side mirror of the reference's ``IterativeTracker`` interface
(/root/reference/src/tracking/IterativeTracker.py:34-180) on top of the C ABI (include/pam.h).

All per-frame work -- association, part-aware view filter, DLT, smoothing, motion, hypothesis initialisation, life
cycle -- runs in one fused HIP kernel launch (csrc/pam_tracker.hip, k_frame); this module only packs detections,
calls ``pam_frame`` and decodes the output record."""
import numpy as np

from . import _lib

NUM_JOINTS = 17


class TrackState:
    Tentative = 1
    Confirmed = 2
    Deleted = 3


class TrackView(object):
    """Read-only snapshot of one device-resident track (the fields of IterTrack the drivers and tests read)."""

    def __init__(self, rec, cameras, frame_dets):
        self.track_id = rec['track_id']
        self.state = rec['state']
        self.hits = rec['hits']
        self.age = rec['age']
        self.time_since_update = rec['time_since_update']
        self.emitted = rec['emitted']
        self.order = rec['order']
        self.time2d = rec['time2d']
        self.matched_det = rec['matched_det']
        self.V = rec['V']
        self.nviews = rec['nviews']
        self.nhist = rec['nhist']
        self.last_time = rec['last_time']
        self.pose3d = rec['pose3d']
        self.velocity_3d = rec['velocity']

    def is_confirmed(self):
        return self.state == TrackState.Confirmed

    def is_tentative(self):
        return self.state == TrackState.Tentative

    @property
    def joints_views(self):
        out = [[] for _ in range(self.V)]
        for j in range(NUM_JOINTS):
            out[int(self.nviews[j]) - 1].append(j)
        return out


class IterativeTracker(object):
    def __init__(self, args, n_views=None, max_dets=16, max_tracks=32, max_hyps=0, device=0):
        """args: the reference's iter_args (attribute access) -- conf_threshold, epi_threshold, init_threshold,
        joint_threshold, num_joints, init_method, n_init, max_age, alpha2d, lambda_a, lambda_t, sigma, arm_sigma."""
        self.args = args
        assert int(args.num_joints) == NUM_JOINTS, 'the path is specialised for 17 COCO joints (ivclabpose.py:96)'
        assert args.init_method == 'GD', "only INIT_METHOD 'GD' exists in the reference (IterativeTracker.py:52)"
        self.cam_num = 0
        self.max_dets, self.max_tracks, self.max_hyps, self.device = max_dets, max_tracks, max_hyps, device
        self.params = _lib.make_params(dict(
            EPI_THRESHOLD=args.epi_threshold, INIT_THRESHOLD=args.init_threshold, JOINT_THRESHOLD=args.joint_threshold,
            ALPHA2D=args.alpha2d, LAMBDA_A=args.lambda_a, LAMBDA_T=args.lambda_t, N_INIT=args.n_init,
            MAX_AGE=args.max_age, SIGMA=args.sigma, ARM_SIGMA=args.arm_sigma), args.conf_threshold)
        self.handle = None
        self.cameras = None
        self.tracks = []
        self.last = None
        self._guard = None           # set_input_guard: the pose network whose void word the frame kernel reads
        if n_views is not None:
            self._open(n_views)

    def _open(self, n_views):
        self.handle = _lib.Handle(n_views, self.params, max_dets=self.max_dets, max_tracks=self.max_tracks,
                                  max_hyps=self.max_hyps, n_scenes=1, device=self.device)
        self.cam_num = n_views
        self._ndet = np.zeros((1, n_views), dtype=np.int32)
        self._det = np.zeros((1, n_views, self.max_dets, NUM_JOINTS, 3), dtype=np.float64)
        # the record of the device-side step lands in PINNED host memory: a pageable destination makes the copy a staged, blocking one
        # (one buffer in the device record's layout: one copy per frame instead of two)
        self._rec_keep, self._rec_i, self._rec_d = self.handle.pinned_record()
        if self._guard is not None:
            self.handle.set_input_guard(self._guard.void_word.data_ptr())

    def set_input_guard(self, net):
        """net: the HRNetPose whose decode feeds this tracker (None removes the guard).  While its ``void_word`` is raised -- a device-side
        gate of one of its captured forwards timed out -- the frame kernel does not apply frames (record status ST_INPUT_VOID)."""
        self._guard = net
        if self.handle is not None:
            self.handle.set_input_guard(net.void_word.data_ptr() if net is not None else None)

    def set_cameras(self, cameras):
        if self.handle is None or self.cam_num != len(cameras):
            if self.handle is not None:
                self.handle.close()
            self._open(len(cameras))
        self.cameras = cameras
        self.handle.set_cameras(np.stack([c.P for c in cameras]), np.stack([c.F for c in cameras]),
                                np.stack([c.RK_INV for c in cameras]), np.stack([c.position for c in cameras]))

    def track_restart(self):
        self.handle.reset()
        self.tracks = []

    def tracking(self, frame_id, camera_list, frame_list, boxes_list, detections_list, build3D='SVD'):
        assert build3D == 'SVD', "Please modify BUILD3D to SVD when PERSON_MATCHER == Iterative"
        if self.cameras is None or self.cameras is not camera_list:     # the reference uses the camera_list of every call
            self.set_cameras(camera_list)
        self._ndet[:] = 0
        for v, dets in enumerate(detections_list):
            n = len(dets)
            if n > self.max_dets:
                raise _lib.PamError('view %d has %d detections > max_dets=%d' % (v, n, self.max_dets))
            self._ndet[0, v] = n
            if n:
                self._det[0, v, :n] = dets
        self.handle.frame(frame_id, self._ndet, self._det)
        self.last = self.handle.decode(0)
        if self.last['status'] & _lib.ST_INPUT_VOID:
            # host keypoints are valid by construction (DumpResults re-ran a void forward before handing them out); the word was left
            # raised by a forward nobody consumed: lower it and apply the frame
            if self._guard is None:
                raise _lib.PamError('frame %d was skipped: the input guard word is raised and no pose network is attached' % frame_id)
            self._guard.check_void(); self._guard.clear_void()
            self.handle.frame(frame_id, self._ndet, self._det)
            self.last = self.handle.decode(0)
        self.tracks = [TrackView(r, self.cameras, detections_list) for r in self.last['tracks']]
        c = self.last['clocks']
        return float(c[1] - c[0]), float(c[2] - c[1]), float(c[3] - c[2])

    def tracking_dev(self, frame_id, camera_list, dev_n_det, dev_det, build3D='SVD', on_void=None):
        """The same step on detections that are already on the device: dev_n_det (views,) int32, dev_det (views, max_dets, 17, 3)
        float64 rows (y, x, score) CUDA tensors (what ``HRNetPose.predict`` keeps) -> no host packing, no host -> device copy;
        one launch + one device -> host copy of the record."""
        import torch
        assert build3D == 'SVD', "Please modify BUILD3D to SVD when PERSON_MATCHER == Iterative"
        if self.cameras is None or self.cameras is not camera_list:
            self.set_cameras(camera_list)
        if tuple(dev_det.shape) != (self.cam_num, self.max_dets, NUM_JOINTS, 3) or dev_det.dtype != torch.float64 or \
                dev_n_det.dtype != torch.int32 or not dev_det.is_contiguous():
            raise _lib.PamError('device detections %s do not match the tracker (%d views, max_dets=%d)' % (tuple(dev_det.shape), self.cam_num, self.max_dets))
        oi, od = self._rec_i, self._rec_d
        for attempt in (0, 1):
            st = torch.cuda.current_stream(dev_det.device).cuda_stream
            self.handle.frame_dev(st, frame_id, dev_n_det.data_ptr(), dev_det.data_ptr())
            self.handle.fetch(st, oi, od)
            self.handle.sync(st)
            self.last = self.handle.decode(0, oi, od)
            # on_void: the keypoints' producer re-runs its forward into the same device buffers (DumpResults.redo_if_void) -- the frame
            # kernel skipped the frame (state untouched), so the second attempt is the frame's first application
            if not (self.last['status'] & _lib.ST_INPUT_VOID) or attempt == 1 or on_void is None or not on_void():
                break
        if self.last['status'] != 0:
            raise _lib.PamError('tracker status 0x%x (1 / 2 capacity overflow, 4 infeasible assignment, 8 clamped count, 16 void input) on frame %d' % (self.last['status'], frame_id))
        self.tracks = [TrackView(r, self.cameras, None) for r in self.last['tracks']]
        c = self.last['clocks']
        return float(c[1] - c[0]), float(c[2] - c[1]), float(c[3] - c[2])
