"""Config + data plumbing of the drivers (/root/reference/src/dataset.py:13-45) without easydict / natsort / cv2:
YAML -> attribute dict, natural-sorted per-camera file lists, PIL image loading returned as BGR uint8 like cv2.imread."""
import glob
import os
import re

import numpy as np
import yaml


class AttrDict(dict):
    """dict with attribute access, recursive (the subset of easydict.EasyDict the drivers use)."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def GetConfig(config_file):
    with open(config_file) as f:
        return AttrDict(yaml.safe_load(f))


def natural_key(s):
    return [int(t) if t.isdigit() else t.lower() for t in re.split(r'(\d+)', s)]


def LoadFilenames(dataset):
    """-> list over frames of list over cameras of file names (FOLDERS_ORDER order)."""
    per_cam = [sorted(glob.glob(os.path.join(dataset.ROOT, folder, dataset.DATA_FORMAT)), key=natural_key)
               for folder in dataset.FOLDERS_ORDER]
    n = min(len(f) for f in per_cam) if per_cam else 0
    return [[f[i] for f in per_cam] for i in range(n)]


def LoadImages(dataset, files):
    """-> (list of BGR uint8 HxWx3 arrays, timestamp) -- timestamp parsing as dataset.py:36-40."""
    base = os.path.basename(files[0])
    timestamp = int(base.split('_')[-1].split('.')[0]) if dataset == 'Panoptic' else base.split('.')[0]
    from PIL import Image
    return [np.ascontiguousarray(np.asarray(Image.open(f).convert('RGB'))[:, :, ::-1]) for f in files], timestamp
