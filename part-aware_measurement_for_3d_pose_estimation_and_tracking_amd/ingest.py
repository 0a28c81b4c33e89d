"""Frame ingest for the drivers (reference: src/dataset.py:36-45, one cv2.imread per camera per frame on the
main thread).  With the per-frame compute at a few milliseconds, decoding C JPEGs per frame is the bottleneck of a real run,
so frames are decoded by a pool of worker threads a few frames ahead (PIL/libjpeg-turbo releases the GIL), written as BGR
uint8 HxWx3 -- cv2.imread's layout -- into pinned host buffers and, when a device is given, uploaded on a dedicated copy
stream; the consumer only waits on an event.  Frame order is preserved."""
import os
import threading
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def decode_bgr(path, out=None):
    """One image file -> BGR uint8 (H, W, 3), optionally into a preallocated array."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    if out is None:
        return np.ascontiguousarray(rgb[:, :, ::-1])
    out[...] = rgb[:, :, ::-1]
    return out


def timestamp_of(dataset_name, first_file):
    base = os.path.basename(first_file)
    return int(base.split('_')[-1].split('.')[0]) if dataset_name == 'Panoptic' else base.split('.')[0]


class FrameLoader(object):
    """Iterates (frame_index, imagelist, timestamp) over ``frames`` (list over frames of list over cameras of file names),
    ``depth`` frames ahead on ``workers`` threads.  imagelist entries are NumPy BGR arrays, or CUDA uint8 tensors when
    ``device`` is given (pinned staging + async copy on a side stream)."""

    def __init__(self, dataset_name, frames, indices=None, workers=8, depth=4, device=None):
        self.name, self.frames = dataset_name, frames
        self.indices = list(range(len(frames))) if indices is None else list(indices)
        self.depth = max(1, depth)
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers))
        self.device = device
        self._pinned = {}
        self._slot_events = {}
        self._lock = threading.Lock()
        if device is not None:
            import torch
            self.torch = torch
            self.copy_stream = torch.cuda.Stream(device)

    def _pinned_for(self, slot, cam, shape):
        key = (slot, cam, tuple(shape))
        with self._lock:
            buf = self._pinned.get(key)
            if buf is None:
                buf = self._pinned[key] = self.torch.empty(tuple(shape), dtype=self.torch.uint8).pin_memory()
        return buf

    def _decode_pinned(self, path, slot, cam):
        """Worker thread: one image file -> BGR uint8 straight into the slot's pinned staging buffer (round 5: the main thread used to
        copy every decoded image into pinned memory itself, 12 MB per Shelf frame set, which capped the device path at ~125 frame sets/s
        on the GPU hosts against 245 for the decode alone)."""
        from PIL import Image
        with Image.open(path) as im:
            rgb = np.asarray(im.convert('RGB'))
        pin = self._pinned_for(slot, cam, rgb.shape)
        np.copyto(pin.numpy(), rgb[:, :, ::-1])
        return pin

    def _submit(self, k):
        files = self.frames[self.indices[k]]
        if self.device is None:
            return [self.pool.submit(decode_bgr, f) for f in files]
        slot = k % (self.depth + 1)
        prev = self._slot_events.get(slot)
        if prev is not None:
            prev.synchronize()                           # the async copy out of this pinned slot (depth + 1 frames ago) has finished: safe to overwrite
        return [self.pool.submit(self._decode_pinned, f, slot, c) for c, f in enumerate(files)]

    def __iter__(self):
        pending = deque()
        nxt = 0
        n = len(self.indices)
        while nxt < n and len(pending) < self.depth:
            pending.append((nxt, self._submit(nxt))); nxt += 1
        while pending:
            k, futs = pending.popleft()
            imgs = [f.result() for f in futs]
            ts = timestamp_of(self.name, self.frames[self.indices[k]][0])
            if self.device is None:
                if nxt < n:
                    pending.append((nxt, self._submit(nxt))); nxt += 1
                yield self.indices[k], imgs, ts
                continue
            torch = self.torch
            slot = k % (self.depth + 1)
            out = []
            consumer = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.copy_stream):
                for pin in imgs:
                    t = pin.to(self.device, non_blocking=True)
                    # the block belongs to copy_stream's pool; the consumer reads it on ITS stream after only a wait_event, so tell
                    # the caching allocator -- otherwise the block can be handed to the next frame's copy while kernels still read it
                    t.record_stream(consumer)
                    out.append(t)
                ev = torch.cuda.Event(); ev.record(self.copy_stream)
            self._slot_events[slot] = ev
            consumer.wait_event(ev)
            if nxt < n:                                  # frame k + depth takes the slot of frame k - 1, whose copy was issued a step ago
                pending.append((nxt, self._submit(nxt))); nxt += 1
            yield self.indices[k], out, ts

    def close(self):
        self.pool.shutdown(wait=False)
