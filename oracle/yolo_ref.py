"""CPU ORACLE for the person-detector side -- test infrastructure, not product code.

PARITY UNPINNED: the reference only *calls* ``backend.YOLOv3.YOLOv3`` (/root/reference/src/ivclabpose.py:116-120 ctor,
:183-204 PersonDetect); the backend, its cfg and its weights are absent from /root/reference, and there is no golden
vector for it.  This file restates the public Darknet YOLOv3 post-processing (Redmon & Farhadi 2018, darknet
src/yolo_layer.c get_yolo_box / src/box.c do_nms_sort) and OpenCV's ``cv2.resize(..., INTER_LINEAR)`` sampling in NumPy
float32, so that the HIP kernels of csrc/pam_detect.hip have a checker.  Only tests/ may import it.
"""
import numpy as np

f32 = np.float32


def bf16_round(x):
    """float32 -> nearest-even bfloat16, returned as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def resize_frames(frames, out_h, out_w):
    """(n, H, W, 3) uint8 BGR -> (n, out_h, out_w, 8) float32 holding bf16-rounded RGB/255 + 5 zero channels.
    Sampling: src = (dst + 0.5) * scale - 0.5, clamped to the image (border replicate), bilinear."""
    n, H, W, _ = frames.shape
    sx = np.clip((np.arange(out_w, dtype=f32) + f32(0.5)) * (f32(W) / f32(out_w)) - f32(0.5), f32(0), f32(W - 1)).astype(f32)
    sy = np.clip((np.arange(out_h, dtype=f32) + f32(0.5)) * (f32(H) / f32(out_h)) - f32(0.5), f32(0), f32(H - 1)).astype(f32)
    x0 = sx.astype(np.int64); y0 = sy.astype(np.int64)
    x1 = np.minimum(x0 + 1, W - 1); y1 = np.minimum(y0 + 1, H - 1)
    fx = (sx - x0.astype(f32))[None, None, :, None]; fy = (sy - y0.astype(f32))[None, :, None, None]
    img = frames[..., ::-1].astype(f32)                     # BGR -> RGB
    a = img[:, y0][:, :, x0]; b = img[:, y0][:, :, x1]; c = img[:, y1][:, :, x0]; d = img[:, y1][:, :, x1]
    top = a + (b - a) * fx; bot = c + (d - c) * fx
    v = (top + (bot - top) * fy) * f32(1.0 / 255.0)
    out = np.zeros((n, out_h, out_w, 8), dtype=f32)
    out[..., :3] = bf16_round(v.astype(f32))
    return out


def upsample_concat(a, b):
    """a (n, h, w, Ca), b (n, 2h, 2w, Cb) -> (n, 2h, 2w, Ca + Cb): nearest x2 of a, then b, along channels."""
    return np.concatenate([a.repeat(2, axis=1).repeat(2, axis=2), b], axis=3)


def sigmoid(x):
    return (f32(1.0) / (f32(1.0) + np.exp(-x.astype(f32)))).astype(f32)


def yolo_candidates(heads, anchors, net_w, net_h, num_classes, class_id, score_thresh, frame_w, frame_h):
    """heads: three (gh, gw, C >= 3*(5+nc)) float32 arrays of ONE image; anchors (3, 3, 2) network pixels.
    -> (boxes (k, 4) x1 y1 x2 y2 in frame pixels, scores (k,)) in candidate order: head, cell row-major, anchor."""
    boxes, scores = [], []
    st = 5 + num_classes
    for h, t in enumerate(heads):
        gh, gw = t.shape[:2]
        v = t[:, :, :3 * st].reshape(gh, gw, 3, st).astype(f32)
        gy, gx = np.meshgrid(np.arange(gh, dtype=f32), np.arange(gw, dtype=f32), indexing='ij')
        sc = (sigmoid(v[..., 4]) * sigmoid(v[..., 5 + class_id])).astype(f32)
        bx = ((sigmoid(v[..., 0]) + gx[..., None]) / f32(gw)).astype(f32)
        by = ((sigmoid(v[..., 1]) + gy[..., None]) / f32(gh)).astype(f32)
        aw = np.asarray(anchors, dtype=f32)[h, :, 0][None, None, :]; ah = np.asarray(anchors, dtype=f32)[h, :, 1][None, None, :]
        with np.errstate(over='ignore'):
            bw = (np.exp(v[..., 2]) * aw / f32(net_w)).astype(f32)
            bh = (np.exp(v[..., 3]) * ah / f32(net_h)).astype(f32)
        x1 = (bx - f32(0.5) * bw) * f32(frame_w); x2 = (bx + f32(0.5) * bw) * f32(frame_w)
        y1 = (by - f32(0.5) * bh) * f32(frame_h); y2 = (by + f32(0.5) * bh) * f32(frame_h)
        keep = sc > f32(score_thresh)
        boxes.append(np.stack([x1, y1, x2, y2], -1)[keep]); scores.append(sc[keep])
    return np.concatenate(boxes).astype(f32), np.concatenate(scores).astype(f32)


def greedy_nms(boxes, scores, nms_thresh, max_det, max_cand=1024):
    """Best score first (ties: lower candidate number); a survivor is dropped when IoU with a kept box > nms_thresh.
    Only the first max_cand candidates enter (the kernel's LDS capacity).  -> kept candidate indices."""
    boxes, scores = boxes[:max_cand].astype(f32), scores[:max_cand].astype(f32)
    alive = np.ones(len(scores), dtype=bool)
    area = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])).astype(f32)
    kept = []
    while len(kept) < max_det and alive.any():
        s = np.where(alive, scores, f32(-1))
        i = int(np.argmax(s))                               # first maximum = lowest index
        kept.append(i)
        iw = (np.minimum(boxes[:, 2], boxes[i, 2]) - np.maximum(boxes[:, 0], boxes[i, 0])).astype(f32)
        ih = (np.minimum(boxes[:, 3], boxes[i, 3]) - np.maximum(boxes[:, 1], boxes[i, 1])).astype(f32)
        inter = np.where((iw > 0) & (ih > 0), (iw * ih).astype(f32), f32(0)).astype(f32)
        uni = ((area + area[i]).astype(f32) - inter).astype(f32)
        with np.errstate(invalid='ignore'):
            alive &= ~(inter > (f32(nms_thresh) * uni).astype(f32))
        alive[i] = False
    return kept


def detect(heads, anchors, net_w, net_h, num_classes, class_id, score_thresh, nms_thresh, frame_w, frame_h, max_det):
    boxes, scores = yolo_candidates(heads, anchors, net_w, net_h, num_classes, class_id, score_thresh, frame_w, frame_h)
    kept = greedy_nms(boxes, scores, nms_thresh, max_det)
    return np.concatenate([boxes[kept], scores[kept, None]], axis=1).reshape(-1, 5), len(scores)
