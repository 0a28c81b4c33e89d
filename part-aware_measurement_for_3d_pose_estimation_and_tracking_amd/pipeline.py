"""On-device per-frame pipeline: crop/resize/normalise (HIP) -> HRNet-W48 conv stack (hand-written MFMA kernels, hipGraph replay) ->
head + arg-max decode (HIP) -> [all-gather of per-view keypoints when views are sharded] -> fused tracker frame kernel (HIP).
Nothing returns to the host inside a frame; the output record is copied out asynchronously.

This is the device-resident form of testmodel.py's loop body (/root/reference/src/testmodel.py:59-69):
PersonPoseDetect + PersonTrack_Project3DPose."""
import contextlib

import numpy as np
import torch

from . import _lib
from .distributed import CropGather, ViewGather
from .hrnet import HRNetPose

NUM_JOINTS = 17


class FramePipeline(object):
    def __init__(self, calib_cameras, matcher, conf_threshold, frame_hw, max_dets=8, max_tracks=16, device=0, world=1,
                 rank=0, group=None, use_graph=True, hrnet=True, seed=0, shard='views', overlap_tracker=False, net=None, exchange='torch',
                 pose_streams=1, autotune=True, prewarm=False):
        """shard: 'views' -- rank owns whole camera views (pose_step / track_step take view-local inputs); 'crops' -- the
        frame's crops are dealt out evenly over the ranks (pose_step_crops / track_step_crops take global view indices).
        overlap_tracker (either mode): exchange + tracker kernel + fetch of frame t run on their own stream, under the conv
        stack of frame t+1 (the tracker is one workgroup per scene; it rides on CUs the conv kernels leave idle).
        pose_streams = 2: the crop -> conv stack -> decode chains of consecutive frames alternate between two streams, each with its own
        replay instance (static input, activations, output) of the same weights, so frame t+1's HBM-bound stem / layer1 runs beside
        frame t's CU-bound last stages.  Frames still finish in order (the tracker stream takes them in order).
        autotune: the conv stack's replay of every crop count is the fastest of the executor's configurations on this device
        (HRNetPose(autotune=True)).
        prewarm: capture the replay of every crop-count bucket this rank can see (its views x max_dets, in steps of the network's
        graph_bucket) at construction -- `self.warmed` says what that cost -- and pad every frame's forward to its bucket, so that no
        step() ever captures (the reference's call shape has a fixed batch_size = 20, /root/reference/src/ivclabpose.py:208-212)."""
        self.device = torch.device('cuda:%d' % device)
        torch.cuda.set_device(self.device)
        self.cams = calib_cameras
        self.C = len(calib_cameras)
        self.max_dets = max_dets
        self.frame_h, self.frame_w = frame_hw
        self.world, self.rank = world, rank
        self.params = _lib.make_params(matcher, conf_threshold)
        self.handle = _lib.Handle(self.C, self.params, max_dets=max_dets, max_tracks=max_tracks, n_scenes=1, device=device)
        self.handle.set_cameras(np.stack([c.P for c in calib_cameras]), np.stack([c.F for c in calib_cameras]),
                                np.stack([c.RK_INV for c in calib_cameras]), np.stack([c.position for c in calib_cameras]))
        # net: an existing HRNetPose to share (weights, packed images, captured graphs) between several pipelines of one process
        self.net = net if net is not None else (HRNetPose(48, 17, None, resolution=(384, 288), device=device, use_graph=use_graph, seed=seed,
                                                          max_dets=max_dets, autotune=autotune,
                                                          max_crops=len(calib_cameras) * max_dets) if hrnet else None)
        self.shard = shard
        # exchange: 'torch' = torch.distributed (RCCL when the backend is nccl, gloo in the CPU tests); 'abi' = pam_allgather_keypoints,
        # the library's own RCCL call on the decode stream (view sharding only)
        self.comm = None
        if exchange == 'abi':
            from .distributed import AbiComm
            self.comm = AbiComm(world, rank, device, group)
        self.gather = ViewGather(self.C, max_dets, world, rank, self.device, group, abi=(self.handle, self.comm) if self.comm else None)
        self.crop_gather = CropGather(self.C, max_dets, world, rank, self.device, group) if shard == 'crops' else None
        self.mine = self.gather.mine
        # decode target: this rank's views only -- the leading records of the exchange's send buffer, (len(mine), max_dets + 1, 17, 3):
        # slot stride max_dets + 1, the extra row carries the view's detection count (ViewGather)
        self.det_local = self.gather.det_local
        self._rec_keep, oi, od = self.handle.pinned_record()           # one pinned buffer in the device record's layout: one copy per fetch
        self.out_i, self.out_d = torch.from_numpy(oi), torch.from_numpy(od)
        self.ev = None
        self.track_stream, self.track_overlaps = self._pick_track_stream() if overlap_tracker else (None, False)
        self.ev_pose, self.ev_track, self._track_pending = torch.cuda.Event(), torch.cuda.Event(), False
        assert pose_streams in (1, 2) and (pose_streams == 1 or overlap_tracker), 'two pose streams need the tracker on its own stream'
        self.pose_streams = [torch.cuda.Stream(self.device) for _ in range(pose_streams)] if pose_streams > 1 else None
        self._frame_no = 0
        # device-side gates (pam_sync.hip) need the forward to be the only flagged work on the device: two forwards in flight can block
        # each other's queues, and so can the replays of several ranks that share one device (more ranks than devices: one-device tests)
        # (which ranks share one is decided from the devices' identities: a launcher that shows each rank a single device has device_count() == 1)
        from .distributed import ranks_share_a_device
        self.shared_device = ranks_share_a_device(self.device, group) if world > 1 else False
        if self.net is not None and (self.pose_streams is not None or self.shared_device):
            self.net.disable_flag_sync()
        if self.net is not None:
            self.net.flag_race = 'throughput'        # the pipeline keeps the host a frame ahead of the device: time the flag race that way
            # Keypoints of a forward whose gate gave up must not reach the tracker state.  The host runs a frame ahead and cannot look, so
            # the frame kernel does: it skips every frame while the producer's void word is up (pam_set_input_guard; results() then raises
            # FrameVoid with the frame to resume from).  Sharded: the word travels with the exchange, so that every rank's replica skips
            # the same frames -- inside the view records (ViewGather.exchange), or as CropGather.void_any, which is the guard there.
            guard = self.crop_gather.void_any if (shard == 'crops' and world > 1) else self.net.void_word
            self.handle.set_input_guard(guard.data_ptr())
        self.bucketed, self.warmed = bool(prewarm), None
        if prewarm and self.net is not None:
            most = (self.C if shard == 'crops' else len(self.mine)) * max_dets
            if shard == 'crops' and world > 1:
                most = (most + world - 1) // world
            self.warmed = self.net.warm(most, slots=(0, 1) if pose_streams > 1 else (0,))

    def _pick_track_stream(self, tries=8, spin_us=400, avoid=None):
        """A stream for the exchange + tracker that REALLY runs beside the caller's (pose) stream.  HIP streams are multiplexed onto a few
        hardware queues (4 by default on ROCm 7.2) and a queue is in-order: the first stream the pool handed out sat on the pose stream's
        queue, so frame t's 80 us tracker kernel ran in FRONT of frame t + 1's crop kernel instead of under its conv stack (rocprofv3
        kernel trace, tools/frame_timeline.py: 110 us from the end of the head kernel to the next frame's first kernel, 10-20 us once
        the two streams are on different queues; +4.6 % frames/s).  Which queue a stream gets is decided at its first use and cannot be
        queried, so this measures it: two spin kernels (pam_clock_probe, one wave each), one per stream, take as long as ONE when the
        queues differ and as long as two when they are the same.  Returns (stream, overlaps); the first candidate that overlaps wins.
        The reference point is the stream that is current when the pipeline is built (frames are issued on the caller's current stream);
        with pose_streams = 2 the frames run on two pool streams of their own, which this does not test against.
        avoid: streams the new one must ALSO overlap with (the detector's stream must share a queue neither with the pose nor with the
        tracker stream)."""
        import ctypes as C
        import time
        pose = torch.cuda.current_stream(self.device)
        out = torch.zeros(4, dtype=torch.int64, device=self.device)
        lib = self.handle.lib

        def both(s, us):
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for st, o in ((pose, out), (s, out[2:])):
                if lib.pam_clock_probe(C.c_void_p(st.cuda_stream), C.c_void_p(o.data_ptr()), us) != 0:
                    raise _lib.PamError('pam_clock_probe failed')
            torch.cuda.synchronize(self.device)
            return time.perf_counter() - t0
        others = [o for o in (avoid or []) if o is not None]

        def pair(x, y, us):
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for st, o in ((x, out), (y, out[2:])):
                if lib.pam_clock_probe(C.c_void_p(st.cuda_stream), C.c_void_p(o.data_ptr()), us) != 0:
                    raise _lib.PamError('pam_clock_probe failed')
            torch.cuda.synchronize(self.device)
            return time.perf_counter() - t0
        cands = []
        for _ in range(tries):
            s = torch.cuda.Stream(self.device)
            cands.append(s)
            both(s, 20)                                  # first use: the stream gets its hardware queue here
            if min(both(s, spin_us) for _ in range(3)) < 1.5e-6 * spin_us and \
                    all(min(pair(o, s, spin_us) for _ in range(3)) < 1.5e-6 * spin_us for o in others):
                return s, True
        return cands[0], False

    # -- person detector of frame t + 1 under the pose network of frame t -----------------------------------------------------------------
    def attach_detector(self, detector, frames=None, n_crops=None, candidates=6):
        """detector: a pam.yolov3.YOLOv3.  The reference loop is detect -> pose -> track per frame (/root/reference/src/testmodel.py:59-63,
        ivclabpose.py:183-204); here frame t + 1's detection (k_resize_frames -> Darknet-53 -> k_yolo_detect, one hipGraph replay) is
        issued on a stream and hardware queue of its own as soon as frame t's crop kernel has read the detector's previous output, and
        runs under frame t's conv stack; frame t + 1's crop kernel waits (on the device) for it.
        frames + n_crops: choose the detector's stream by MEASUREMENT -- four hardware queues serve all streams and the pose replay's
        own branch chains sit on all of them, so which queue the detector shares decides how much of it hides: `candidates` fresh
        streams are each timed running the detector's replay beside the n_crops pose replay, the fastest pair wins (``det_pick``)."""
        import time
        self.detector = detector
        self.det_stream, self.det_overlaps = self._pick_track_stream(avoid=[self.track_stream])
        self.det_pick = None
        if frames is not None and n_crops and self.net is not None:
            cur = torch.cuda.current_stream(self.device)
            x = self.net.input_buffer(int(n_crops))
            self.net.features(x); detector.detect_dev(frames)
            torch.cuda.synchronize(self.device)

            def both(st, reps=6):
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                for _ in range(reps):
                    self.net.features(x)
                    with torch.cuda.stream(st):
                        detector.detect_dev(frames)
                torch.cuda.synchronize(self.device)
                return (time.perf_counter() - t0) / reps
            cands = [self.det_stream] + [torch.cuda.Stream(self.device) for _ in range(candidates - 1)]
            times = []
            for st in cands:
                both(st, 2)
                times.append(min(both(st) for _ in range(2)))
            k = int(np.argmin(times))
            self.det_stream = cands[k]
            self.det_pick = {'ms_pose_and_detector_side_by_side': [float(t * 1e3) for t in times], 'chosen': k}
        self.ev_det, self.ev_crop = [torch.cuda.Event(), torch.cuda.Event()], torch.cuda.Event()
        self._det_frames = 0

    def detect_ahead(self, frames):
        """Issue the detection of the NEXT frame (frames: its (views, H, W, 3) uint8 BGR device tensor) behind the crop kernel that was
        issued last on the current stream; returns (boxes, count) device tensors, valid once ``wait_detection`` of that frame has run."""
        cur = torch.cuda.current_stream(self.device)
        self.ev_crop.record(cur)
        with torch.cuda.stream(self.det_stream):
            self.det_stream.wait_event(self.ev_crop)
            out = self.detector.detect_dev(frames)
            self._det_frames += 1
            self.ev_det[self._det_frames & 1].record(self.det_stream)
        return out

    def wait_detection(self):
        """Order the current stream behind the latest detection issued by ``detect_ahead`` (call before the frame's crop kernel)."""
        torch.cuda.current_stream(self.device).wait_event(self.ev_det[self._det_frames & 1])

    def stream_ptr(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _pose(self, frame_ptrs, views, slot_of, boxes, det, time_events, after_crop=None):
        """crop -> conv stack -> head + arg-max for one frame on the current stream, replay slot = frame parity when two pose streams run."""
        k = (self._frame_no & 1) if self.pose_streams is not None else 0
        n = int(views.numel())
        x = self.net.input_buffer(self.net.bucket(n) if self.bucketed else n, k)     # a bucket's spare rows repeat the last crop, undecoded
        self.net.preprocess(frame_ptrs, self.frame_h, self.frame_w, views, boxes, x)
        if after_crop is not None:
            after_crop()                                # the frames have been read: a feeder may mark its buffers reusable here
        if time_events is not None:
            time_events[0].record(torch.cuda.current_stream(self.device))
        f = self.net.features(x, k)
        if time_events is not None:
            time_events[1].record(torch.cuda.current_stream(self.device))
        self.wait_track()                               # the previous frame's exchange / tracker read the buffer decode writes
        self.net.head_decode(f, views, slot_of, boxes, det, n=n)

    @contextlib.contextmanager
    def frame(self):
        """Everything of ONE frame (pose_step*, write_*, track_step*) goes inside.  With two pose streams the frame runs on the stream of
        its parity (ordered behind the caller's stream, which holds its inputs) with that parity's replay slot, so the forwards of
        consecutive frames overlap; frame t + 2 follows frame t on the same stream, which also orders the reuse of the slot."""
        if self.pose_streams is None:
            yield
            self._frame_no += 1
            return
        ps = self.pose_streams[self._frame_no & 1]
        ps.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(ps):
            yield
        self._frame_no += 1

    def pose_step(self, frame_ptrs, view_local, slot_of, boxes, time_events=None, after_crop=None):
        """HRNet side for this rank's crops.  view_local: int32 (N,) index into self.mine; writes self.det_local."""
        if int(view_local.numel()) == 0 or self.net is None:
            if after_crop is not None:
                after_crop()
            return
        self._pose(frame_ptrs, view_local, slot_of, boxes, self.det_local, time_events, after_crop)

    def write_local(self, rows):
        """Copy keypoint rows (len(mine), max_dets, 17, 3) into this rank's records, ordered behind the previous frame's readers."""
        self.wait_track()
        self.det_local[:, :self.max_dets].copy_(rows.reshape(-1, self.max_dets, NUM_JOINTS, 3)[:self.det_local.shape[0]])

    def track_step(self, frame_id, n_det_local, det_local=None, fetch=True):
        """Exchange (if sharded) + fused tracker kernel on the gathered records, read in place through the row map
        (pam_frame_dev_views); async fetch of the record.  det_local: None / ``self.det_local`` = the rows are already in the send
        buffer (decode, write_local); any other tensor is copied in first."""
        if det_local is not None and det_local.data_ptr() != self.det_local.data_ptr():
            self.write_local(det_local)

        def issue(st):
            recv = self.gather.exchange(n_det_local, self.net.void_word if self.net is not None else None)
            self.handle.frame_dev_views(st, frame_id, recv.data_ptr(), self.gather.rows.data_ptr())
            if fetch:
                self.handle.fetch(st, self.out_i.numpy(), self.out_d.numpy())
        if self.track_stream is None:
            issue(self.stream_ptr())
            return
        self.ev_pose.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.track_stream):
            self.track_stream.wait_event(self.ev_pose)
            issue(self.track_stream.cuda_stream)
            self.ev_track.record(self.track_stream)
        self._track_pending = True

    # -- crop-balanced sharding ----------------------------------------------------------------------------------------------
    def wait_track(self):
        """Order the caller's stream behind the previous frame's exchange / tracker (they read ``crop_gather.send`` / the view
        records): call before ANY write to that buffer -- decode, or a caller that fills it itself."""
        if self._track_pending:
            torch.cuda.current_stream(self.device).wait_event(self.ev_track)

    def write_send(self, rows):
        """Copy keypoint rows (C, max_dets, 17, 3) into the exchange buffer, ordered behind the previous frame's readers."""
        self.wait_track()
        self.crop_gather.send.copy_(rows)

    def pose_step_crops(self, frame_ptrs, view_of, slot_of, boxes, time_events=None, after_crop=None):
        """HRNet side for this rank's share of the frame's crops; view_of indexes ALL views (frame_ptrs has C entries).
        Decodes straight into the exchange buffer at (view, slot)."""
        if int(view_of.numel()) == 0 or self.net is None:
            if after_crop is not None:
                after_crop()                             # (a feeder's release / the next frame's detection are due on empty frames too)
            return
        self._pose(frame_ptrs, view_of, slot_of, boxes, self.crop_gather.send, time_events, after_crop)

    def track_step_crops(self, frame_id, n_det, select, fetch=True):
        """n_det (C,) int32 and select (CropGather.select_index) are the same on every rank (they follow from the frame's box
        list); the keypoint rows come from ``crop_gather.send`` of the rank that owns each crop."""
        vw = self.net.void_word if self.net is not None else None
        if self.track_stream is None:
            det = self.crop_gather.gather(select, vw)
            st = self.stream_ptr()
            self.handle.frame_dev(st, frame_id, n_det.data_ptr(), det.data_ptr())
            if fetch:
                self.handle.fetch(st, self.out_i.numpy(), self.out_d.numpy())
            return
        self.ev_pose.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.track_stream):
            self.track_stream.wait_event(self.ev_pose)
            det = self.crop_gather.gather(select, vw)
            st = self.track_stream.cuda_stream
            self.handle.frame_dev(st, frame_id, n_det.data_ptr(), det.data_ptr())
            if fetch:
                self.handle.fetch(st, self.out_i.numpy(), self.out_d.numpy())
            self.ev_track.record(self.track_stream)
        self._track_pending = True

    def results(self, strict=True):
        """Synchronise and decode the last fetched record.  strict: a non-zero status word (capacity overflow, infeasible
        assignment, clamped detection count -- include/pam.h) IN ANY FRAME since the handle was created or reset (the sticky half of
        the word) raises instead of passing silently into the caller's numbers."""
        if self.pose_streams is not None:
            for ps in self.pose_streams:
                ps.synchronize()
        if self.track_stream is not None:
            self.track_stream.synchronize()
        torch.cuda.current_stream(self.device).synchronize()
        rec = self.handle.decode(0, self.out_i.numpy(), self.out_d.numpy())
        if rec['status'] & _lib.ST_INPUT_VOID:
            # a gate of a captured forward timed out in frame rec['first_void']: the tracker (every rank's replica) skipped that frame and
            # all since; the network is on stream events by now (HRNetPose.check_void).  Lower the word and tell the caller where to resume.
            if self.net is not None:
                self.net.check_void()
                self.net.clear_void()
            raise _lib.FrameVoid(rec['first_void'], rec['frame_id'])
        if strict and (rec['status'] | rec['status_sticky']) != 0:
            raise _lib.PamError('tracker status 0x%x on frame %d, 0x%x over the run (1 track slots, 2 hypothesis slots, 4 infeasible '
                                'assignment, 8 clamped detection count)' % (rec['status'], rec['frame_id'], rec['status_sticky']))
        return rec
