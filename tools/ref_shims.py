"""Import the reference's matching / triangulation / tracker modules in THIS container.

Development-time tooling only.  It never ships to the GPU box and nothing in
the package, tests, bench.py or __graft_entry__.py imports it: tests consume the committed vectors under
tests/golden/ that tools/make_goldens.py wrote with the help of this module.

What is stubbed and why:
  * np.int / np.float / np.bool aliases (removed in NumPy >= 1.24, used at matching.py:246 etc.)
  * cv2: only ``computeCorrespondEpilines`` is reached on the live path (matching.py:69-73).  The stub
    restates OpenCV 4.2 ``cv::computeCorrespondEpilines`` for CV_64F points: F converted to double,
    transposed when whichImage == 2, per point a = f0*x + f1*y + f2 (etc.), nu = a*a + b*b,
    nu = nu ? 1/sqrt(nu) : 1, (a, b, c) *= nu, output shape (N, 1, 3) with the points' dtype.
    opencv-python==4.2.0.32 (requirements.txt:5) is absent here => parity is UNPINNED at this boundary.
  * numba, cvxopt, torchvision, easydict, the two absent DNN backends: import-time only.
Run in a separate process from pytest: the reference's tracking/hypothesis.py shadows PyPI ``hypothesis``.
"""
import sys
import types
import numpy as np

REF = '/root/reference/src'


def _epilines(points, which, F):
    pts = np.asarray(points)
    out_dtype = pts.dtype if pts.dtype in (np.float32, np.float64) else np.float32
    p = pts.reshape(-1, 2).astype(np.float64)
    f = np.asarray(F).astype(np.float64)
    if which == 2:
        f = f.T.copy()
    f = f.reshape(-1)
    x, y = p[:, 0], p[:, 1]
    a = f[0] * x + f[1] * y + f[2]
    b = f[3] * x + f[4] * y + f[5]
    c = f[6] * x + f[7] * y + f[8]
    nu = a * a + b * b
    with np.errstate(divide='ignore'):
        nu = np.where(nu != 0, 1.0 / np.sqrt(nu), 1.0)
    return np.stack([a * nu, b * nu, c * nu], axis=1).reshape(-1, 1, 3).astype(out_dtype)


class AttrDict(dict):
    """easydict.EasyDict stand-in (attribute access, recursive)."""
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    for n, t in (('int', int), ('float', float), ('bool', bool)):
        if not hasattr(np, n):
            setattr(np, n, t)
    for p in (REF, REF + '/eval', REF + '/utils', REF + '/tracking'):
        if p not in sys.path:
            sys.path.insert(0, p)

    class _T:
        def __call__(self, *a, **k):
            return self
    _stub('cv2', computeCorrespondEpilines=_epilines, KalmanFilter=object)
    _stub('numba', vectorize=lambda sig: (lambda f: np.vectorize(f)), float32=_T(), float64=_T(),
          boolean=_T(), jit=lambda *a, **k: (lambda f: f))
    _stub('cvxopt', glpk=None, matrix=None, spmatrix=None)
    tv = _stub('torchvision')
    tv.transforms = _stub('torchvision.transforms')
    _stub('easydict', EasyDict=AttrDict)
    _stub('backend')
    _stub('backend.YOLOv3', YOLOv3=object)
    _stub('backend.HRPose')
    _stub('backend.HRPose.SimpleHRNet', HRNetPose=object)
    _stub('natsort', natsorted=sorted)
    _stub('motmetrics')

    class _PT:
        def __init__(self):
            self.field_names = []
            self.rows = []

        def add_row(self, r):
            self.rows.append(r)

        def __str__(self):
            return '\n'.join(str(r) for r in [self.field_names] + self.rows)
    _stub('prettytable', PrettyTable=_PT)


def load_reference():
    """Returns the reference modules (ivclabpose, IterativeTracker, matching, construction, calculate,
    hypothesis)."""
    install()
    import ivclabpose as ref_ivclabpose
    import IterativeTracker as ref_tracker
    import matching as ref_matching
    import construction as ref_construction
    import calculate as ref_calculate
    import hypothesis as ref_hypothesis
    return types.SimpleNamespace(ivclabpose=ref_ivclabpose, tracker=ref_tracker, matching=ref_matching,
                                 construction=ref_construction, calculate=ref_calculate,
                                 hypothesis=ref_hypothesis)
