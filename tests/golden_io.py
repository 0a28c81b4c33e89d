"""Readers for tests/golden/*.npz (written by tools/make_goldens.py from the reference)."""
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SIZES = ('S1', 'S2', 'S3', 'S4')


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def ops(size):
    """dict op -> list of record dicts."""
    z = load('ops_%s.npz' % size)
    out = {}
    for k in z.files:
        parts = k.split('.')
        if parts[1] == 'count':
            out.setdefault(parts[0], [dict() for _ in range(int(z[k]))])
    for k in z.files:
        parts = k.split('.')
        if parts[1] != 'count':
            out[parts[0]][int(parts[1])][parts[2]] = z[k]
    return out


def cameras(size):
    return load('cameras_%s.npz' % size)


def trace_frames(tr):
    """list[frame][view] -> (n,17,3) arrays in the dump layout (x, y, score)."""
    n_det = tr['in.n_det']
    dets = tr['in.dets']
    frames, pos = [], 0
    for t in range(n_det.shape[0]):
        views = []
        for c in range(n_det.shape[1]):
            n = int(n_det[t, c])
            views.append(dets[pos:pos + n].reshape(n, 17, 3))
            pos += n
        frames.append(views)
    return frames
