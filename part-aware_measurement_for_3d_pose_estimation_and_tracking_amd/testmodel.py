#!/usr/bin/env python3
"""Demo driver with the reference's CLI (/root/reference/src/testmodel.py): ``python testmodel.py --dataset Shelf``.

Per frame: load C images -> person boxes -> HRNet 2D poses -> PersonTrack_Project3DPose, with the reference's timing
print-out.  Boxes come from PersonDetect when ``DETECT_MODEL: YOLOv3`` (testmodel.py:56-58), else -- ``DETECT_MODEL: None``,
the shipped default -- boxes (and optionally 2D poses) come from ``DATASET.PRECOMPUTED``, a pickle
{frame_id: [per view list of {'bbox': [x,y,w,h], optional 'keypoints', 'keypoints_score'}]} -- see INTEGRATION.md."""
import argparse
import os
import pickle
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import pam  # noqa: E402
from pam.dataset import GetConfig, LoadFilenames  # noqa: E402
from pam.ingest import FrameLoader  # noqa: E402


def build_model(cfg):
    from pam.ivclabpose import ivclabpose
    pipe = cfg.PIPELINE_COMBINATION
    det = cfg.DETECT_MODELS[str(pipe['DETECT_MODEL']).upper()]
    pose_key = str(pipe['POSE_MODEL']).upper()
    pose = cfg.POSE_MODELS[pose_key] if pose_key in cfg.POSE_MODELS else None
    matcher = cfg.PERSON_MATCHERS[str(pipe['PERSON_MATCHER']).upper()]
    return ivclabpose(person_detector=det, pose_detector=pose, person_matcher=matcher,
                      conf_threshold=pipe['CONF_THRESHOLD']), pipe['BUILD_3D']


def frame_inputs(model, precomputed, frame_id, imagelist, ahead=None, next_frame=None):
    """person_bbox_list + dump_result_list for one frame -> (pbl, dump, detect seconds, pose seconds, the next frame's detection ticket).
    ahead: this frame's detection, issued one frame ago (PersonDetectAhead); next_frame = (frame_id, imagelist) of the frame after this
    one when the loader has it already: its detection is issued BEFORE this frame's pose network and runs under it."""
    dt_det = 0.0
    nxt = None
    if model.person_detector is not None:                    # testmodel.py:56-58
        t0 = time.time()
        pbl = model.PersonDetectResult(ahead) if ahead is not None else model.PersonDetect(imagelist, frame_id)
        if next_frame is not None:
            nxt = model.PersonDetectAhead(next_frame[1], next_frame[0])
        dt_det = time.time() - t0
    else:
        views = precomputed.get(frame_id, [[] for _ in imagelist])
        pbl = [[dict(image_id=frame_id, category_id=1, score=float(p.get('score', 1.0)), bbox=list(p['bbox']),
                     data=imagelist[v], feature=[]) for p in persons] for v, persons in enumerate(views)]
        if all('keypoints' in p for persons in views for p in persons):
            dump = [[dict(bbox=list(p['bbox']), keypoints=list(p['keypoints']), keypoints_score=list(p['keypoints_score']),
                          feature=[]) for p in persons] for persons in views]
            return pbl, dump, 0.0, 0.0, None
    t0 = time.time()
    dump = model.PersonPoseDetect(imagelist=None, person_bbox_list=pbl, batch_size=20)
    return pbl, dump, dt_det, time.time() - t0, nxt


def test_ivclabpose_PersonTrack_Project3DPose(cfg, inputs, on_frame=None):
    dataset = cfg.DATASET
    os.makedirs(cfg.OUTPUT, exist_ok=True)
    with open(os.path.join(dataset.ROOT, dataset.CALIBRATION_FILE), 'rb') as f:
        camera_parameter = pickle.load(f)
    model, build3D = build_model(cfg)
    precomputed = {}
    if model.person_detector is None:
        with open(os.path.join(dataset.ROOT, dataset.PRECOMPUTED), 'rb') as f:
            precomputed = pickle.load(f)
    t_det = t_pose = t_track = 0.0
    start, end = dataset.TEST_RANGE
    n_views = len(dataset.FOLDERS_ORDER)
    # images are decoded a few frames ahead on worker threads; with a GPU pipeline behind them (detector or pose network) they are
    # uploaded from pinned memory on a copy stream and arrive as CUDA tensors (LOADER_DEVICE: false keeps them NumPy arrays, which
    # the overlay needs anyway).  DETECT_AHEAD (default on): frame t + 1's person detection is issued before frame t's pose network.
    draw = bool(cfg.get('VISUALIZATION') or cfg.get('SAVE_IMAGE'))
    on_gpu = (model.person_detector is not None or model.pose_model is not None) and not draw and bool(dataset.get('LOADER_DEVICE', True))
    look_ahead = model.person_detector is not None and bool(dataset.get('DETECT_AHEAD', True))
    loader = FrameLoader(dataset.TEST_DATASET, inputs, indices=range(start, end), workers=int(dataset.get('LOADER_THREADS', 8)),
                         device=(model.device if on_gpu else None))
    it = iter(loader)
    cur = next(it, None)
    ahead = None
    i = -1
    while cur is not None:
        i += 1
        frame_id, imagelist, timestamp = cur
        nxt_frame = next(it, None)
        if i == 0:
            model.GetCameraParameters(camera_parameter, imagelist[0].shape[0], imagelist[0].shape[1])
        pbl, dump, dt_det, dt_pose, ahead = frame_inputs(model, precomputed, frame_id, imagelist, ahead,
                                                         (nxt_frame[0], nxt_frame[1]) if (look_ahead and nxt_frame is not None) else None)
        cur = nxt_frame
        result = None
        dt_track = 0.0
        if any(len(v) for v in dump):                       # testmodel.py:66 guard: no pose in any view -> frame skipped
            t0 = time.time()
            result = model.PersonTrack_Project3DPose(frame_id=frame_id, person_bbox_list=pbl, dump_results=dump, build3D=build3D)
            dt_track = time.time() - t0
        if result is not None and (cfg.get('VISUALIZATION') or cfg.get('SAVE_IMAGE')):      # testmodel.py:70-75 overlay
            from pam.visualization import joints_dict, draw_points_and_skeleton
            camera_ids, pts, person_ids = result[0], result[1], result[2]
            for cids, poses_2d, pids in zip(camera_ids, pts, person_ids):
                for cid, pose_2d, pid in zip(cids, poses_2d, pids):
                    imagelist[cid] = draw_points_and_skeleton(imagelist[cid], pose_2d, joints_dict()['coco']['skeleton'],
                                                              person_index=pid, points_color_palette='gist_rainbow',
                                                              skeleton_color_palette='tab20', points_palette_samples=17,
                                                              confidence_threshold=0.0)
            if cfg.get('SAVE_IMAGE'):
                from PIL import Image
                store = os.path.join(cfg.OUTPUT, dataset.TEST_DATASET, 'Images')
                os.makedirs(store, exist_ok=True)
                for cid, im in enumerate(imagelist):
                    Image.fromarray(np.ascontiguousarray(im[..., ::-1])).save(os.path.join(store, '%s_%d.jpg' % (frame_id, cid)))
        if on_frame is not None:
            on_frame(frame_id, timestamp, result)
        if frame_id > start + 10:
            t_det += dt_det
            t_pose += dt_pose
            t_track += dt_track
    loader.close()
    n = max(1, end - start - 10)
    print("Person Detect Processing time (s/f): %f" % (t_det / n))
    print("Pose Detect Processing time (s/f): %f" % (t_pose / n))
    print("Track Processing time (s/f): %f" % (t_track / n))
    print("fps: %f" % (1 / max(1e-12, (t_det / n + t_pose / n) / n_views + t_track / n)))
    print("tracking fps: %f" % (1 / max(1e-12, t_track / n)))


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', help='Three options: CampusSeq1, Shelf, Panoptic', type=str, default='CampusSeq1')
    opt = parser.parse_args()
    cfg = GetConfig(os.path.join(_HERE, 'configs', opt.dataset, 'model_configs.yaml'))
    datas = LoadFilenames(cfg.DATASET)
    {'PersonTrack_Project3DPose': test_ivclabpose_PersonTrack_Project3DPose}[cfg.TEST_FUNCTION](cfg, datas)
