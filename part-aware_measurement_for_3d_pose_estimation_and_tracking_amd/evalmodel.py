#!/usr/bin/env python3
"""Evaluation driver with the reference's CLI (/root/reference/src/evalmodel.py): ``python evalmodel.py --dataset Shelf``.
Runs the same loop as testmodel.py, stores ``{frame_id (Panoptic: timestamp): pts3d (n,3,17)}`` like evalmodel.py:83-91,
pickles it (evalmodel.py:373-377) and prints the PCP table (evalmodel.py:120-206) for Campus / Shelf."""
import argparse
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
import pam  # noqa: E402
from pam.dataset import GetConfig, LoadFilenames  # noqa: E402
from pam.evaluation import Evaluate3DPose_PCP, EvaluatePanoptic, Write3DResult  # noqa: E402
from pam.testmodel import test_ivclabpose_PersonTrack_Project3DPose  # noqa: E402


def eval_ivclabpose_PersonTrack_Project3DPose(cfg, inputs):
    dataset = cfg.DATASET
    pipe = cfg.PIPELINE_COMBINATION
    multi_poses3d = {}

    def collect(frame_id, timestamp, result):
        key = timestamp if dataset.TEST_DATASET == 'Panoptic' else frame_id
        multi_poses3d[key] = result[3] if result is not None else []
    test_ivclabpose_PersonTrack_Project3DPose(cfg, inputs, on_frame=collect)
    path = os.path.join(cfg.OUTPUT, dataset.TEST_DATASET, 'logs', '{}_{}_{}_{}.pkl'.format(
        pipe['DETECT_MODEL'], pipe['POSE_MODEL'], pipe['PERSON_MATCHER'], os.path.basename(dataset.ROOT)))
    Write3DResult(multi_poses3d, path)
    if dataset.TEST_DATASET == 'Panoptic':
        EvaluatePanoptic(dataset.EVAL_RANGE, path, dataset.TEST_DATASET, seqs=dataset.FOLDERS_ORDER, data_root=dataset.ROOT)
    else:
        Evaluate3DPose_PCP(dataset.EVAL_RANGE, path, gt_path=dataset.ROOT, dataset_name=dataset.TEST_DATASET)


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', help='Three options: CampusSeq1, Shelf, Panoptic', type=str, default='CampusSeq1')
    opt = parser.parse_args()
    cfg = GetConfig(os.path.join(_HERE, 'configs', opt.dataset, 'model_configs.yaml'))
    datas = LoadFilenames(cfg.DATASET)
    {'PersonTrack_Project3DPose': eval_ivclabpose_PersonTrack_Project3DPose}[cfg.TEST_FUNCTION](cfg, datas)
