"""Drop-in façade: the reference's ``ivclabpose`` / ``Camera`` surface (/root/reference/src/ivclabpose.py:35-287)
with the matching / triangulation / tracking path running on MI355X through libpam_hip.so.

Kept: constructor signature and config attribute names, ``GetCameraParameters``, ``PersonDetect``,
``PersonPoseDetect``, ``PersonTrack_Project3DPose`` and its 9-tuple.  The YOLOv3 detector is outside the path
(SURVEY 8f rank 1): with DETECT_MODELS.NONE person boxes must be supplied by the caller."""
import numpy as np
import torch

from .tracker import IterativeTracker, NUM_JOINTS


class _Args(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _cfg(obj, key):
    return obj[key] if isinstance(obj, dict) else getattr(obj, key)


class Camera(object):
    """Calibrated view: P, K, RT (float32), F[other] (float32), RK_INV (float32), position (float64)."""

    def __init__(self, cid, P, K, RT, F, w=640, h=480):
        self.cid, self.P, self.K, self.RT, self.F, self.w, self.h = cid, P, K, RT, F, w, h
        self.RK_INV = np.linalg.inv(RT[:, :3]) @ np.linalg.inv(K)
        self.position = np.linalg.inv(np.vstack([RT, [0, 0, 0, 1]]))[:3, 3]

    def undistort(self, im):
        return im

    def undistort_points(self, points2d):
        return points2d

    def projectPoints_parallel(self, points3d):
        """(n,17,3) -> (n,17,2) in (y, x); host NumPy convenience (the device does this inside k_frame)."""
        n, j = points3d.shape[0], points3d.shape[1]
        hom = np.concatenate([points3d, np.ones((n, j, 1))], axis=2).reshape(-1, 4)
        h = (self.P @ hom.T).T
        return (h[:, :2] / h[:, 2:3])[:, ::-1].reshape(n, j, 2)

    def projectPoints(self, points3d):
        return self.projectPoints_parallel(np.asarray(points3d, dtype=np.float64)[None])[0]

    projectPoints_undist = projectPoints


def fundamental_matrices(K, RT):
    """(C,C,3,3) float32, F[a][b] with x_a^T F x_b = 0, evaluated in float32 torch CPU algebra in the reference's
    operation order (ivclabpose.py:166-177) so both stacks hand the same bits to the device."""
    C = len(K)
    Kt = [torch.tensor(K[i]) for i in range(C)]
    R = [torch.tensor(RT[i][:, :3]) for i in range(C)]
    T = [torch.tensor(RT[i][:, 3]) for i in range(C)]
    Kinv_t = [torch.inverse(k).t() for k in Kt]
    F = torch.zeros(C, C, 3, 3)
    for a in range(C):
        for b in range(C):
            e = Kt[b] @ R[b] @ R[a].t() @ (T[a] - R[a] @ R[b].t() @ T[b])
            ex = torch.tensor([[0, -e[2], e[1]], [e[2], 0, -e[0]], [-e[1], e[0], 0]])
            F[a, b] += Kinv_t[a] @ (R[a] @ R[b].t()) @ Kt[b].t() @ ex
            if F[a, b].sum() == 0:
                F[a, b] += 1e-12
    return F.numpy()


class ivclabpose(object):
    def __init__(self, person_detector=None, pose_detector=None, person_matcher=None, conf_threshold=0.4,
                 max_dets=16, max_tracks=32, device=0, autotune=False):
        """autotune (not in the reference): let the pose network pick the fastest of its executor configurations per crop count on this
        device (HRNetPose(autotune=True)); off by default because two batchings of the same crops may then differ in the last bf16 bits."""
        self.person_detector = person_detector if _cfg(person_detector, 'NAME') != '' else None
        self.pose_detector = pose_detector
        self.person_matcher = person_matcher
        self.conf_threshold = conf_threshold
        self.device = device
        self.cameras = None
        self.pose_model = None
        if self.person_detector is None:
            print("Person Detector : Close.")
        elif _cfg(self.person_detector, 'NAME') == 'YOLOv3':                        # ivclabpose.py:116-120
            import os
            from .yolov3 import YOLOv3
            d = self.person_detector
            cfg, weight, names = _cfg(d, 'CFG'), _cfg(d, 'WEIGHT'), _cfg(d, 'CLASS_NAMES')
            # the reference's cfg / weight / names files are not distributed with it: a missing cfg or names file means
            # the standard YOLOv3-416 COCO layout (person = class 0), missing weights mean a seeded random network
            self.bbox_detector = YOLOv3(cfg if cfg and os.path.exists(cfg) else None,
                                        weight if weight and os.path.exists(weight) else None,
                                        names if names and os.path.exists(names) else None,
                                        score_thresh=_cfg(d, 'SCORE_THRESH'), nms_thresh=_cfg(d, 'NMS_THRESH'),
                                        use_cuda=True, device=device, max_det=max_dets)   # best max_dets boxes per view: the tracker's capacity
            print("Person Detector : ", _cfg(d, 'NAME'), '(weights: %s)' % self.bbox_detector.weights)
        else:
            raise NotImplementedError('person detector %r' % _cfg(self.person_detector, 'NAME'))
        if self.pose_detector is None:
            print("Pose Detector : Close.")
        elif _cfg(self.pose_detector, 'NAME') == 'HRPose':
            from .hrnet import HRNetPose
            # ivclabpose.py:107-111: gpu_args = every visible GPU.  One process drives ONE of them (`device`); several GPUs = several
            # ranks of a torch.distributed job, and HRNetPose.predict shards each call's crops over them (see HRNetPose.__init__).
            gpu_args = _Args(gpus=list(range(torch.cuda.device_count())), device=torch.device('cuda:%d' % device))
            self.pose_model = HRNetPose(_cfg(self.pose_detector, 'C'), _cfg(self.pose_detector, 'NUM_JOINTS'),
                                        _cfg(self.pose_detector, 'CHECKPOINT_FILE'),
                                        model_name=_cfg(self.pose_detector, 'MODEL_NAME'),
                                        resolution=tuple(_cfg(self.pose_detector, 'RESOLUTION')), hrpose_args=gpu_args,
                                        device=device, max_dets=max_dets,
                                        shard_crops=torch.distributed.is_available() and torch.distributed.is_initialized(),
                                        autotune=autotune)
            # optional key (not in the reference's YAMLs): SOFT_ARGMAX_BETA > 0 switches the decode from the hard arg-max (parity mode)
            # to the soft-arg-max of pam_head_decode_soft with that inverse temperature
            sb = self.pose_detector.get('SOFT_ARGMAX_BETA') if isinstance(self.pose_detector, dict) else getattr(self.pose_detector, 'SOFT_ARGMAX_BETA', None)
            if sb:
                self.pose_model.soft_beta = float(sb)
            print("Pose Detector : ", _cfg(self.pose_detector, 'NAME'))
        if self.person_matcher is None:
            print("Person Matcher : Close.")
            self.tracker = None
        elif _cfg(self.person_matcher, 'NAME') == 'Iterative':
            m = self.person_matcher
            a = _Args(conf_threshold=conf_threshold)
            for dst, src in (('epi_threshold', 'EPI_THRESHOLD'), ('init_threshold', 'INIT_THRESHOLD'),
                             ('joint_threshold', 'JOINT_THRESHOLD'), ('num_joints', 'NUM_JOINTS'),
                             ('init_method', 'INIT_METHOD'), ('n_init', 'N_INIT'), ('max_age', 'MAX_AGE'),
                             ('w2d', 'W2D'), ('alpha2d', 'ALPHA2D'), ('w3d', 'W3D'), ('alpha3d', 'ALPHA3D'),
                             ('lambda_a', 'LAMBDA_A'), ('lambda_t', 'LAMBDA_T'), ('sigma', 'SIGMA'),
                             ('arm_sigma', 'ARM_SIGMA')):
                a[dst] = _cfg(m, src)
            self.tracker = IterativeTracker(a, max_dets=max_dets, max_tracks=max_tracks, device=device)
            if self.pose_model is not None:
                self.tracker.set_input_guard(self.pose_model)      # keypoints of a forward whose gate timed out never reach the tracker state
            print("Person Matcher : ", _cfg(m, 'NAME'))

    # -- a18 ----------------------------------------------------------------------------------------------------------
    def GetCameraParameters(self, camera_parameter, im_width, im_height, F=None):
        P = np.asarray(camera_parameter['P']).astype(np.float32)
        K = np.asarray(camera_parameter['K']).astype(np.float32)
        RT = np.asarray(camera_parameter['RT']).astype(np.float32)
        if F is None:
            F = fundamental_matrices(K, RT)
        self.cameras = [Camera(j, P[j], K[j], RT[j], F[j], w=im_width, h=im_height) for j in range(len(P))]
        if self.tracker is not None:
            self.tracker.set_cameras(self.cameras)
        return self.cameras

    def PersonDetect(self, imglist, image_id):
        """ivclabpose.py:183-204: per image a list of person dicts, boxes clamped to the image, xywh."""
        if self.person_detector is None or _cfg(self.person_detector, 'NAME') != 'YOLOv3':
            return None
        return self._person_dicts(imglist, image_id, self.bbox_detector(imglist))

    def PersonDetectAhead(self, imglist, image_id):
        """Not in the reference: PersonDetect split in two so that a driver which already holds the NEXT frame's images (the loader decodes
        ahead) can run that frame's detector under the current frame's pose network -- issue here (a stream of its own, nothing waits),
        collect with ``PersonDetectResult``.  Same boxes as PersonDetect (same replay, same decode)."""
        if self.person_detector is None or _cfg(self.person_detector, 'NAME') != 'YOLOv3':
            return None
        if getattr(self, '_det_stream', None) is None:
            self._det_stream = torch.cuda.Stream(self.bbox_detector.device)
        return (imglist, image_id, self.bbox_detector.submit(imglist, stream=self._det_stream))

    def PersonDetectResult(self, ahead):
        if ahead is None:
            return None
        imglist, image_id, ticket = ahead
        return self._person_dicts(imglist, image_id, self.bbox_detector.collect(ticket))

    def _person_dicts(self, imglist, image_id, results):
        person_bbox_list = []
        for idx, result in enumerate(results):
            h, w = imglist[idx].shape[:2]
            person_temps = []
            for ret in result:
                x1, y1 = max(0, float(ret[0])), max(0, float(ret[1]))
                x2, y2 = min(float(ret[2]), w), min(float(ret[3]), h)
                person_temps.append(dict(image_id=image_id, category_id=1, score=float(round(float(ret[4]), 4)),
                                         bbox=[x1, y1, x2 - x1, y2 - y1], data=imglist[idx], feature=[]))
            person_bbox_list.append(person_temps)
        return person_bbox_list

    def PersonPoseDetect(self, imagelist=None, person_bbox_list=None, batch_size=20, image_id=None):
        if self.pose_model is None:
            return None
        return self.pose_model.predict(person_bbox_list, batch_size, self.conf_threshold)

    # -- a2 + a17 ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _unpack(dump_results):
        """dump dicts -> per-view (n,17,3) float64 rows (y, x, score) (ivclabpose.py:233-245)."""
        out = []
        for items in dump_results:
            if len(items) == 0:
                out.append(np.zeros((0, NUM_JOINTS, 3)))
                continue
            k = np.array([it['keypoints'] for it in items], dtype=np.float64).reshape(len(items), NUM_JOINTS, 3)
            s = np.array([it['keypoints_score'] for it in items], dtype=np.float64)
            out.append(np.stack([k[:, :, 1], k[:, :, 0], s], axis=2))
        return out

    def PersonTrack_Project3DPose(self, frame_id, person_bbox_list=None, dump_results=None, build3D='SVD'):
        dev = getattr(dump_results, 'device_det', None)
        if dev is not None and dev.shape[1] == self.tracker.max_dets and dev.shape[0] == len(self.cameras) and dump_results.device_valid():
            # the dump is the one PersonPoseDetect returned, untouched: its keypoints are still on the device in the tracker's
            # layout -> no re-packing, no host -> device copy (the dicts stay the source of truth whenever the caller edits them)
            # (the frame kernel is queued first: the one host wait of this call, behind it, also covers the keypoints' copy predict() enqueued)
            asso_time, update_time, init_time = self.tracker.tracking_dev(frame_id, self.cameras, dump_results.device_n_det, dev, build3D,
                                                                          on_void=getattr(dump_results, 'redo_if_void', None))
            poses = dump_results.poses_host
        else:
            poses = self._unpack(dump_results)
            boxes = [np.array([it['bbox'] for it in items]) for items in dump_results]
            frames = [(b[0]['data'] if len(b) else []) for b in person_bbox_list]
            asso_time, update_time, init_time = self.tracker.tracking(frame_id, self.cameras, frames, boxes, poses, build3D)
        camera_ids, pts, person_ids, pts3d, pts3d_joints_views, person3d_ids = [], [], [], [], [], []
        for tr in self.tracker.tracks:
            if not tr.emitted:
                continue
            pts3d.append(tr.pose3d.T)
            pts3d_joints_views.append(tr.joints_views)
            person3d_ids.append(tr.track_id)
            person_ids.append([tr.track_id] * len(tr.order))
            cams = [cid for cid in tr.order if tr.time2d[cid] == frame_id]
            camera_ids.append(cams)
            pts.append([poses[cid][tr.matched_det[cid]] for cid in cams])
        return (np.array(camera_ids, dtype='object'), np.array(pts, dtype='object'), person_ids, np.array(pts3d),
                pts3d_joints_views, np.array(person3d_ids), asso_time, update_time, init_time)
