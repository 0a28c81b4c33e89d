"""ctypes binding of csrc/libpam_hip.so (the C ABI declared in include/pam.h).

There is deliberately no CPU fallback: if the library is missing or does not load, importing the product path
raises with the build command.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C <package>/csrc``."""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PAM_LIB') or os.path.join(_HERE, 'csrc', 'libpam_hip.so')     # PAM_LIB: another build of the same library (tools/ab_build.sh)

PAM_J = 17
PAM_MAX_TAPS = 16
PAM_EXP_TABLE = 64


class PamParams(C.Structure):
    _fields_ = [
        ('conf_threshold', C.c_double), ('epi_threshold', C.c_double), ('init_threshold', C.c_double),
        ('joint_threshold', C.c_double), ('alpha2d', C.c_double), ('lambda_a', C.c_double), ('lambda_t', C.c_double),
        ('n_init', C.c_int32), ('max_age', C.c_int32), ('count_gate', C.c_int32), ('n_taps_body', C.c_int32),
        ('n_taps_arm', C.c_int32), ('reserved', C.c_int32),
        ('taps_body', C.c_double * PAM_MAX_TAPS), ('taps_arm', C.c_double * PAM_MAX_TAPS),
        ('exp_lambda_a', C.c_double * PAM_EXP_TABLE), ('w_lambda_t', C.c_double * 4),
    ]


class PamOutLayout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'n_views', 'max_dets', 'max_tracks', 'n_scenes', 'int_words', 'dbl_words', 'hdr_words', 'trk_words',
        'off_order', 'off_matched', 'off_time2d', 'off_nviews', 'dbl_hdr_words', 'dbl_trk_words')]


# every symbol include/pam.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_I = C.c_int
_SIGS = {
    'pam_create': (_I, [C.POINTER(_P), _I, _I, _I, _I, _I, _I, C.POINTER(PamParams)]),
    'pam_destroy': (_I, [_P]),
    'pam_last_error': (C.c_char_p, [_P]),
    'pam_version': (C.c_char_p, []),
    'pam_set_cameras': (_I, [_P, _P, _P, _P, _P]),
    'pam_reset': (_I, [_P]),
    'pam_out_layout': (_I, [_P, C.POINTER(PamOutLayout)]),
    'pam_frame': (_I, [_P, _I, _P, _P, _P, _P]),
    'pam_frame_dev': (_I, [_P, _P, _I, _P, _P]),
    'pam_frame_dev_views': (_I, [_P, _P, _I, _P, _P]),
    'pam_fetch': (_I, [_P, _P, _P, _P]),
    'pam_sync': (_I, [_P, _P]),
    'pam_op_project': (_I, [_P, _I, _I, _P, _P]),
    'pam_op_track_affinity': (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    'pam_op_lsap': (_I, [_P, _I, _I, _P, _P, _P, _P]),
    'pam_op_epi_dist': (_I, [_P, _I, _P, _P, _P]),
    'pam_op_epi_pair': (_I, [_P, _I, _P, _I, _P, _P]),
    'pam_op_epi_dist_init': (_I, [_P, _I, _P, _P, _P]),
    'pam_op_greedy': (_I, [_P, _I, _I, _P, _P, _P, _P, _P]),
    'pam_op_dlt': (_I, [_P, _I, _P, _P, _P, _P, _P, _P]),
    'pam_op_smooth': (_I, [_P, _I, _P, _P, _P]),
    'pam_op_velocity': (_I, [_P, _I, _P, _P]),
    'pam_op_hyp_cost': (_I, [_P, _I, _P, _P, _I, _P, _P, _P]),
    'pam_preprocess_crops': (_I, [_P, _I, _P, _I, _I, _P, _P, _I, _I, _I, _P]),
    'pam_preprocess_crops_ex': (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _I, _I, _I, _P, _I]),
    'pam_decode_heatmaps': (_I, [_P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    'pam_clock_probe': (_I, [_P, _P, _I]),
    'pam_conv2d_nhwc_bf16': (_I, [_P, _P, _P, _P, _P, _P, _P] + [_I] * 11),
    'pam_conv2d_nhwc_bf16_ex': (_I, [_P, _P, _P, _P, _P, _P, _P] + [_I] * 13),
    'pam_conv3x3_slab': (_I, [_I, _I, _I, _I]),
    'pam_conv3x3_layout': (_I, [_I, _I, _I, _I]),
    'pam_conv_last_kernel': (_I, []),
    'pam_conv3x3_layout_ex': (_I, [_I, _I, _I, _I, _I]),
    'pam_conv3x3_layout_gen': (_I, [_I, _I, _I, _I]),
    'pam_conv3x3_layout_small': (_I, [_I, _I, _I, _I]),
    'pam_conv_debug_stamps': (_I, [_P]),
    'pam_upsample_add_nhwc_bf16': (_I, [_P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I]),
    'pam_upsample_add_nhwc_bf16_ex': (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    'pam_basic_block2_tile': (_I, [_I, _I, _I, _I, _P]),
    'pam_basic_block2_nhwc_bf16': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    'pam_pointwise64_relu_nhwc_bf16': (_I, [_P, _P, _P, _P, _P, C.c_longlong]),
    'pam_pointwise64_act_nhwc_bf16': (_I, [_P, _P, _P, _P, _P, C.c_longlong, _I]),
    'pam_stem_fused_nhwc_bf16': (_I, [_P] * 10 + [_I, _I, _I]),
    'pam_bottleneck_fused_nhwc_bf16': (_I, [_P] * 12 + [_I, _I, _I]),
    'pam_bottleneck_tail_nhwc_bf16': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_longlong, _I]),
    'pam_conv3x3s2_c48_tile': (_I, [_I, _I, _I, _I, _P]),
    'pam_conv3x3s2_c48_nhwc_bf16': (_I, [_P, _P, _I, _P, _P, _P, _I, _P, _I] + [_I] * 9),
    'pam_conv3x3s2_slab': (_I, [_I, _I, _I, _I]),
    'pam_conv3x3s2_nhwc_bf16': (_I, [_P, _P, _I, _P, _P, _P] + [_I] * 7),
    'pam_fuse_sum_nhwc_bf16': (_I, [_P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P] + [_I] * 8),
    'pam_flag_signal': (_I, [_P, _P]),
    'pam_flag_signal_mask': (_I, [_P, _P, C.c_uint32]),
    'pam_flag_gate': (_I, [_P, _P, _I, _P, _P, _I, _P, _P]),
    'pam_set_input_guard': (_I, [_P, _P]),
    'pam_comm_unique_id': (_I, [_P]),
    'pam_comm_init': (_I, [C.POINTER(_P), _I, _I, _P, _I]),
    'pam_comm_destroy': (_I, [_P]),
    'pam_comm_last_error': (C.c_char_p, []),
    'pam_allgather_keypoints': (_I, [_P, _P, _P, _P, _I, _P]),
    'pam_head_decode_scratch_bytes': (C.c_longlong, [_I, _I, _I]),
    'pam_head_decode': (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    'pam_head_decode_soft_scratch_bytes': (C.c_longlong, [_I, _I, _I]),
    'pam_head_decode_soft': (_I, [_P, _I, _I, _I, _P, _I, _P, _P, _I, C.c_float, _P, _P, _P, _P, _I, _P, _P, _P]),
    'pam_head_heatmaps': (_I, [_P, _I, _P, _I, _P, _P, _I, _P]),
    'pam_resize_frames': (_I, [_P, _I, _P, _I, _I, _I, _I, _P]),
    'pam_upsample_concat_nhwc_bf16': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I]),
    'pam_yolo_detect': (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, C.c_float, C.c_float, _I, _I, _I, _P, _P]),
    'pam_yolo_detect_ws': (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, C.c_float, C.c_float, _I, _I, _I, _P, _P, _P, C.c_longlong]),
    'pam_yolo_detect_workspace_bytes': (C.c_longlong, [_I, _P, _P]),
}
EXPORTS = tuple(_SIGS)
YOLO_MAX_CAND = 1024        # PAM_YOLO_MAX_CAND in include/pam.h

_lib = None


def load():
    """Load libpam_hip.so and declare signatures.  Raises RuntimeError (loudly) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('HIP extension missing: %s not built. Run `make -C %s` (needs hipcc, --offload-arch=gfx950). '
                           'There is no CPU fallback.' % (LIB_PATH, os.path.join(_HERE, 'csrc')))
    # One HIP runtime per process: libtorch_hip NEEDs "libamdhip64.so" (its bundled copy, SONAME libamdhip64.so.7) while
    # this library NEEDs "libamdhip64.so.7".  With torch loaded first the loader satisfies ours from torch's copy by
    # SONAME; the other way round it maps /opt/rocm's AND torch's, and the second runtime to initialise finds no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib



_live_graphs = None


def new_graph():
    """A torch.cuda.CUDAGraph that is NOT destroyed while the interpreter shuts down (every graph the package captures is one).

    Destroying a captured multi-stream graph on ROCm 7.2 corrupts the runtime's heap now and then (tools/graph_destroy_stress.py: 3 of
    4 processes that capture 160 forwards and destroy 120 of them die of 'double free or corruption' / SIGSEGV inside a LATER
    synchronize or allocation, 0 of 5 when nothing is destroyed).  The package therefore never drops a capture while its network lives
    (one capture per crop-count bucket and replay slot, all of a slot in ONE memory pool), and at interpreter exit every graph still
    alive gets one reference that is never returned, so torch's destructor (hipGraphExecDestroy / hipGraphDestroy) does not run in the
    teardown either and a finished run cannot turn into a non-zero exit code.  Graphs do die with their network when a process drops one
    mid-run (the test-suite does); making them immortal outright was tried and is worse: with every capture of every earlier network
    still alive, a replay in tests/test_gpu_image.py segfaults inside hipGraphLaunch (3 of 3 full-suite runs)."""
    global _live_graphs
    import torch
    if _live_graphs is None:
        import atexit
        import weakref
        _live_graphs = weakref.WeakSet()

        def _keep():
            for g in list(_live_graphs):
                C.pythonapi.Py_IncRef(C.py_object(g))
        atexit.register(_keep)

    graph = torch.cuda.CUDAGraph()
    _live_graphs.add(graph)
    return graph


class PamError(RuntimeError):
    pass


# status bits of the frame record (include/pam.h, PamOutLayout.hdr_words)
ST_TRACK_OVERFLOW, ST_HYP_OVERFLOW, ST_LSAP_INFEASIBLE, ST_NDET_CLAMPED, ST_INPUT_VOID = 1, 2, 4, 8, 16


class FrameVoid(PamError):
    """The frame step skipped frames because the producer of their keypoints declared them void (a device-side gate of the captured
    HRNet forward timed out; pam_set_input_guard): the tracker state is that of the last applied frame.  ``first`` = the first skipped
    frame -- re-submit from there (the pose network orders its branch streams by stream events by now)."""

    def __init__(self, first, last):
        PamError.__init__(self, 'frames %d..%d were not applied: their keypoints were void (gate time-out); re-submit from frame %d' % (first, last, first))
        self.first, self.last = int(first), int(last)


def gaussian_taps(sigma, truncate=4.0):
    """One-sided taps of scipy.ndimage.gaussian_filter1d's kernel (w[0] = centre), built the way SciPy builds it
    so the device convolves with identical constants (IterativeTracker.py:381-382 call sites)."""
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    return phi[r:].copy()


def make_params(matcher, conf_threshold):
    """PamParams from the reference's PERSON_MATCHERS.ITERATIVE config block (attribute or key access)."""
    g = (lambda k: matcher[k]) if isinstance(matcher, dict) else (lambda k: getattr(matcher, k))
    p = PamParams()
    p.conf_threshold = float(conf_threshold)
    p.epi_threshold = float(g('EPI_THRESHOLD'))
    p.init_threshold = float(g('INIT_THRESHOLD'))
    p.joint_threshold = float(g('JOINT_THRESHOLD'))
    p.alpha2d = float(g('ALPHA2D'))
    p.lambda_a = float(g('LAMBDA_A'))
    p.lambda_t = float(g('LAMBDA_T'))
    p.n_init = int(g('N_INIT'))
    p.max_age = int(g('MAX_AGE'))
    p.count_gate = 10
    tb, ta = gaussian_taps(g('SIGMA')), gaussian_taps(g('ARM_SIGMA'))
    if len(tb) > PAM_MAX_TAPS or len(ta) > PAM_MAX_TAPS:
        raise ValueError('SIGMA too large for %d taps' % PAM_MAX_TAPS)
    p.n_taps_body, p.n_taps_arm = len(tb), len(ta)
    for i, w in enumerate(tb):
        p.taps_body[i] = w
    for i, w in enumerate(ta):
        p.taps_arm[i] = w
    e = np.exp(g('LAMBDA_A') * np.arange(PAM_EXP_TABLE))          # np.exp, as IterativeTracker.py:148
    for i in range(PAM_EXP_TABLE):
        p.exp_lambda_a[i] = e[i]
    for t in range(4):
        p.w_lambda_t[t] = math.exp(-g('LAMBDA_T') * t)             # math.exp, as construction.py:96
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Handle(object):
    """Owns one PamHandle (one GPU, device-resident tracker state for n_scenes scenes)."""

    def __init__(self, n_views, params, max_dets=16, max_tracks=32, max_hyps=0, n_scenes=1, device=0):
        self.lib = load()
        self._h = C.c_void_p()
        self.params = params
        rc = self.lib.pam_create(C.byref(self._h), device, n_views, max_dets, max_tracks, max_hyps, n_scenes,
                                 C.byref(params))
        if rc != 0:
            raise PamError('pam_create failed (%d): %s' % (rc, self.lib.pam_last_error(None).decode()))
        self.layout = PamOutLayout()
        self._chk(self.lib.pam_out_layout(self._h, C.byref(self.layout)))
        self.C, self.max_dets, self.max_tracks, self.n_scenes = n_views, max_dets, max_tracks, n_scenes
        self.out_i = np.zeros((n_scenes, self.layout.int_words), dtype=np.int32)
        self.out_d = np.zeros((n_scenes, self.layout.dbl_words), dtype=np.float64)

    def _chk(self, rc):
        if rc != 0:
            raise PamError('libpam_hip error %d: %s' % (rc, self.lib.pam_last_error(self._h).decode()))

    def close(self):
        if self._h:
            self.lib.pam_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def raw(self):
        return self._h

    def set_cameras(self, P, F, RK_INV, position):
        P = np.ascontiguousarray(P, dtype=np.float32); F = np.ascontiguousarray(F, dtype=np.float32)
        RK = np.ascontiguousarray(RK_INV, dtype=np.float32); pos = np.ascontiguousarray(position, dtype=np.float64)
        assert P.shape == (self.C, 3, 4) and F.shape == (self.C, self.C, 3, 3) and RK.shape == (self.C, 3, 3)
        self._chk(self.lib.pam_set_cameras(self._h, _ptr(P), _ptr(F), _ptr(RK), _ptr(pos)))

    def reset(self):
        self._chk(self.lib.pam_reset(self._h))

    def frame(self, frame_id, n_det, det):
        """n_det (S,C) int32, det (S,C,max_dets,17,3) float64 (y,x,score) host arrays -> (out_i, out_d)."""
        n_det = np.ascontiguousarray(n_det, dtype=np.int32).reshape(self.n_scenes, self.C)
        det = np.ascontiguousarray(det, dtype=np.float64).reshape(self.n_scenes, self.C, self.max_dets, PAM_J, 3)
        self._chk(self.lib.pam_frame(self._h, int(frame_id), _ptr(n_det), _ptr(det), _ptr(self.out_i), _ptr(self.out_d)))
        return self.out_i, self.out_d

    def frame_dev(self, stream, frame_id, dev_n_det_ptr, dev_det_ptr):
        self._chk(self.lib.pam_frame_dev(self._h, C.c_void_p(stream), int(frame_id), C.c_void_p(dev_n_det_ptr),
                                         C.c_void_p(dev_det_ptr)))

    def frame_dev_views(self, stream, frame_id, dev_records_ptr, dev_view_row_ptr):
        """The frame on the all-gathered per-view records (ViewGather.recv), read in place through the row map."""
        self._chk(self.lib.pam_frame_dev_views(self._h, C.c_void_p(stream), int(frame_id), C.c_void_p(dev_records_ptr),
                                               C.c_void_p(dev_view_row_ptr)))

    def set_input_guard(self, dev_word_ptr):
        """While the device int32 at dev_word_ptr is non-zero the frame step skips its frame (record status ``ST_INPUT_VOID``, state
        untouched): the word is raised by the producer of the keypoints (HRNetPose.void_word).  None / 0 removes the guard."""
        self._chk(self.lib.pam_set_input_guard(self._h, C.c_void_p(dev_word_ptr or None)))

    def pinned_record(self):
        """(keep, out_i, out_d): ONE pinned host buffer laid out like the device record (int32 section padded to 8 bytes, float64 section
        behind it) and NumPy views of its two sections -- ``fetch`` into them is a single device -> host copy."""
        import torch
        L = self.layout
        ib = (4 * self.n_scenes * L.int_words + 7) & ~7
        buf = torch.zeros(ib + 8 * self.n_scenes * L.dbl_words, dtype=torch.uint8).pin_memory()
        a = buf.numpy()
        oi = a[:4 * self.n_scenes * L.int_words].view(np.int32).reshape(self.n_scenes, L.int_words)
        od = a[ib:].view(np.float64).reshape(self.n_scenes, L.dbl_words)
        return buf, oi, od

    def fetch(self, stream, out_i=None, out_d=None):
        out_i = self.out_i if out_i is None else out_i
        out_d = self.out_d if out_d is None else out_d
        self._chk(self.lib.pam_fetch(self._h, C.c_void_p(stream), _ptr(out_i), _ptr(out_d)))
        return out_i, out_d

    def sync(self, stream=0):
        self._chk(self.lib.pam_sync(self._h, C.c_void_p(stream)))

    # ---- decoding of the output record -----------------------------------------------------------------------------
    def decode(self, scene=0, out_i=None, out_d=None):
        """-> list of per-track dicts in the reference's self.tracks order."""
        L = self.layout
        oi = (self.out_i if out_i is None else out_i)[scene]
        od = (self.out_d if out_d is None else out_d)[scene]
        n = int(oi[0])
        tracks = []
        for i in range(n):
            b = oi[L.hdr_words + i * L.trk_words: L.hdr_words + (i + 1) * L.trk_words]
            dd = od[L.dbl_hdr_words + i * L.dbl_trk_words: L.dbl_hdr_words + (i + 1) * L.dbl_trk_words]
            nv2 = int(b[6])
            tracks.append(dict(
                track_id=int(b[0]), state=int(b[1]), hits=int(b[2]), age=int(b[3]), time_since_update=int(b[4]),
                emitted=bool(b[5]), order=[int(c) for c in b[L.off_order:L.off_order + nv2]], V=int(b[7]),
                nhist=int(b[8]), last_time=int(b[9]),
                matched_det=b[L.off_matched:L.off_matched + self.C].copy(),
                time2d=b[L.off_time2d:L.off_time2d + self.C].copy(),
                nviews=b[L.off_nviews:L.off_nviews + PAM_J].copy(),
                pose3d=dd[:PAM_J * 3].reshape(PAM_J, 3).copy(), velocity=dd[PAM_J * 3:].reshape(PAM_J, 3).copy()))
        # status word: bits 0-15 = this frame's bits, bits 16-31 = OR of every frame's bits since create / reset (include/pam.h)
        w = int(oi[1]) & 0xffffffff
        return dict(n_tracks=n, status=w & 0xffff, status_sticky=(w >> 16) & 0xffff, frame_id=int(oi[2]), n_hyp=int(oi[3]),
                    first_void=int(oi[3]) if (w & ST_INPUT_VOID) else None,
                    clocks=od[:4].copy(), clocks_all=od[:L.dbl_hdr_words].copy(), tracks=tracks)

    # ---- per-operator entry points (parity tests) -------------------------------------------------------------------
    def op_project(self, cid, poses3d):
        p = np.ascontiguousarray(poses3d, dtype=np.float64).reshape(-1, PAM_J, 3)
        out = np.zeros((len(p), PAM_J, 2))
        self._chk(self.lib.pam_op_project(self._h, cid, len(p), _ptr(p), _ptr(out)))
        return out

    def op_track_affinity(self, cid, tracks_pose, dt, dets):
        tp = np.ascontiguousarray(tracks_pose, dtype=np.float64); dd = np.ascontiguousarray(dets, dtype=np.float64)
        t = np.ascontiguousarray(dt, dtype=np.int32)
        out = np.zeros((len(tp), len(dd)))
        self._chk(self.lib.pam_op_track_affinity(self._h, cid, len(tp), len(dd), _ptr(tp), _ptr(t), _ptr(dd), _ptr(out)))
        return out

    def op_lsap(self, cost):
        c = np.ascontiguousarray(cost, dtype=np.float64)
        nr, nc = c.shape
        N = max(nr, nc, 1)
        rows = np.zeros(N, dtype=np.int32); cols = np.zeros(N, dtype=np.int32); n = C.c_int32(0)
        self._chk(self.lib.pam_op_lsap(self._h, nr, nc, _ptr(c), _ptr(rows), _ptr(cols), C.byref(n)))
        return rows[:n.value].copy(), cols[:n.value].copy()

    def op_epi_dist(self, cids, pose_mat):
        c = np.ascontiguousarray(cids, dtype=np.int32); pm = np.ascontiguousarray(pose_mat, dtype=np.float64)
        V = len(c); out = np.zeros((V, V, PAM_J))
        self._chk(self.lib.pam_op_epi_dist(self._h, V, _ptr(c), _ptr(pm), _ptr(out)))
        return out

    def op_epi_pair(self, c1, p1, c2, p2):
        a = np.ascontiguousarray(p1, dtype=np.float64); b = np.ascontiguousarray(p2, dtype=np.float64)
        out = np.zeros((PAM_J, 2))
        self._chk(self.lib.pam_op_epi_pair(self._h, int(c1), _ptr(a), int(c2), _ptr(b), _ptr(out)))
        return out

    def op_epi_dist_init(self, cids, pose_mat):
        c = np.ascontiguousarray(cids, dtype=np.int32); pm = np.ascontiguousarray(pose_mat, dtype=np.float64)
        V = len(c); out = np.zeros((V, V, PAM_J), dtype=np.float32)
        self._chk(self.lib.pam_op_epi_dist_init(self._h, V, _ptr(c), _ptr(pm), _ptr(out)))
        return out

    def op_greedy(self, mode, cids, aff, pose_j=None, next_pose_j=None):
        c = np.ascontiguousarray(cids, dtype=np.int32)
        V = len(c)
        if mode == 'update':
            a = np.ascontiguousarray(aff, dtype=np.float64)
            pj = np.ascontiguousarray(pose_j, dtype=np.float64).reshape(V, 3)
            nj = np.ascontiguousarray(next_pose_j, dtype=np.float64).reshape(3)
        else:
            a = np.ascontiguousarray(aff, dtype=np.float32); pj = nj = None
        k = C.c_uint32(0)
        self._chk(self.lib.pam_op_greedy(self._h, 0 if mode == 'update' else 1, V, _ptr(c), _ptr(a), _ptr(pj), _ptr(nj),
                                         C.byref(k)))
        return k.value

    def op_dlt(self, cids, Ts, pose_mat, keep_masks, next_pose):
        c = np.ascontiguousarray(cids, dtype=np.int32); t = np.ascontiguousarray(Ts, dtype=np.int32)
        pm = np.ascontiguousarray(pose_mat, dtype=np.float64); k = np.ascontiguousarray(keep_masks, dtype=np.uint32)
        nx = np.ascontiguousarray(next_pose, dtype=np.float64); out = np.zeros((PAM_J, 3))
        self._chk(self.lib.pam_op_dlt(self._h, len(c), _ptr(c), _ptr(t), _ptr(pm), _ptr(k), _ptr(nx), _ptr(out)))
        return out

    def op_smooth(self, hist, raw):
        hs = np.ascontiguousarray(hist, dtype=np.float64).reshape(-1, PAM_J, 3)
        L = len(hs)
        if L == 0:
            hs = np.zeros((1, PAM_J, 3))
        r = np.ascontiguousarray(raw, dtype=np.float64); out = np.zeros((PAM_J, 3))
        self._chk(self.lib.pam_op_smooth(self._h, L, _ptr(hs), _ptr(r), _ptr(out)))
        return out

    def op_velocity(self, hist):
        hs = np.ascontiguousarray(hist, dtype=np.float64).reshape(-1, PAM_J, 3)
        out = np.zeros((PAM_J, 3), dtype=np.float32)
        self._chk(self.lib.pam_op_velocity(self._h, len(hs), _ptr(hs), _ptr(out)))
        return out

    def op_hyp_cost(self, cids, poses, o_cid, o_pose):
        c = np.ascontiguousarray(cids, dtype=np.int32); ps = np.ascontiguousarray(poses, dtype=np.float64)
        po = np.ascontiguousarray(o_pose, dtype=np.float64)
        cost = C.c_double(0); veto = C.c_int32(0)
        self._chk(self.lib.pam_op_hyp_cost(self._h, len(c), _ptr(c), _ptr(ps), int(o_cid), _ptr(po), C.byref(cost), C.byref(veto)))
        return cost.value, bool(veto.value)
