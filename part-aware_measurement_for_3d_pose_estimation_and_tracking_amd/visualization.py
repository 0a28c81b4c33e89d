"""2D overlay of the tracked poses: the ``joints_dict`` / ``draw_points_and_skeleton`` pair that the
reference's demo driver imports from the absent ``HRPose.misc.visualization`` (/root/reference/src/testmodel.py:34,72-75).
Pure host code (NumPy + PIL; OpenCV is not a dependency here).  Points are rows (y, x, confidence) like the 2D poses
``PersonTrack_Project3DPose`` returns; images are H x W x 3 uint8 BGR arrays and a new array is returned."""
import colorsys

import numpy as np
from PIL import Image, ImageDraw

_TAB20 = [(31, 119, 180), (174, 199, 232), (255, 127, 14), (255, 187, 120), (44, 160, 44), (152, 223, 138), (214, 39, 40),
          (255, 152, 150), (148, 103, 189), (197, 176, 213), (140, 86, 75), (196, 156, 148), (227, 119, 194), (247, 182, 210),
          (127, 127, 127), (199, 199, 199), (188, 189, 34), (219, 219, 141), (23, 190, 207), (158, 218, 229)]


def joints_dict():
    """COCO 17-keypoint names and limb list (pairs of joint indices)."""
    return {'coco': {
        'keypoints': {0: 'nose', 1: 'left_eye', 2: 'right_eye', 3: 'left_ear', 4: 'right_ear', 5: 'left_shoulder',
                      6: 'right_shoulder', 7: 'left_elbow', 8: 'right_elbow', 9: 'left_wrist', 10: 'right_wrist',
                      11: 'left_hip', 12: 'right_hip', 13: 'left_knee', 14: 'right_knee', 15: 'left_ankle', 16: 'right_ankle'},
        'skeleton': [[15, 13], [13, 11], [16, 14], [14, 12], [11, 12], [5, 11], [6, 12], [5, 6], [5, 7], [6, 8], [7, 9],
                     [8, 10], [1, 2], [0, 1], [0, 2], [1, 3], [2, 4], [0, 5], [0, 6]]}}


def _palette(name, samples):
    if name == 'tab20':
        return [_TAB20[i % 20] for i in range(samples)]
    # rainbow-like palettes ('gist_rainbow', 'jet', ...): evenly spaced hues
    return [tuple(int(255 * c) for c in colorsys.hsv_to_rgb(0.83 * i / max(1, samples - 1), 1.0, 1.0)) for i in range(samples)]


def draw_points_and_skeleton(image, points, skeleton, points_color_palette='tab20', points_palette_samples=16,
                             skeleton_color_palette='Set2', skeleton_palette_samples=8, person_index=0, confidence_threshold=0.5):
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    im = Image.fromarray(np.ascontiguousarray(np.asarray(image)[..., ::-1]))           # BGR -> RGB for PIL
    draw = ImageDraw.Draw(im)
    radius = max(2, int(round(min(im.size) / 150.0)))
    limb_colors = _palette(skeleton_color_palette, skeleton_palette_samples)
    color = limb_colors[int(person_index) % len(limb_colors)]
    for a, b in skeleton:                                                              # limbs: one colour per person
        if pts[a, 2] > confidence_threshold and pts[b, 2] > confidence_threshold:
            draw.line([(pts[a, 1], pts[a, 0]), (pts[b, 1], pts[b, 0])], fill=color, width=max(1, radius // 2 + 1))
    pt_colors = _palette(points_color_palette, points_palette_samples)
    for j, (y, x, c) in enumerate(pts):                                                # joints: one colour per joint
        if c > confidence_threshold:
            draw.ellipse([x - radius, y - radius, x + radius, y + radius], fill=pt_colors[j % len(pt_colors)])
    return np.asarray(im)[..., ::-1].copy()
