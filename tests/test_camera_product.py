"""a18 (Camera / GetCameraParameters, /root/reference/src/ivclabpose.py:35-46,162-181): the PRODUCT functions -- not the oracle's
copies -- against the camera goldens the reference produced (tests/golden/cameras_S*.npz, tools/make_goldens.py).  CPU only: the
calibration set-up is host code (float32 torch CPU algebra in the reference's operation order)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import golden_io as G          # noqa: E402
import pam                      # noqa: E402,F401
from pam.ivclabpose import Camera, fundamental_matrices          # noqa: E402


def _ulps32(a, b):
    """distance in float32 units in the last place."""
    a = np.ascontiguousarray(a, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia); ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize('size', G.SIZES)
def test_product_camera_setup_vs_reference_goldens(size):
    c = G.cameras(size)
    P = c['P'].astype(np.float32); K = c['K'].astype(np.float32); RT = c['RT'].astype(np.float32)
    F = fundamental_matrices(K, RT)
    assert F.dtype == np.float32 and F.shape == c['F'].shape
    # Bit-equal on this image's torch build (the goldens were produced by the reference's own torch CPU code here).  Another BLAS /
    # torch build may reorder the 3x3 products: allow a documented bound of 4 float32 ulps instead of failing on the last bit.
    u = _ulps32(F, c['F'])
    assert int(u.max()) <= 4, (size, int(u.max()))
    if int(u.max()) == 0:
        assert np.array_equal(F, c['F'])
    for j in range(len(P)):
        cam = Camera(j, P[j], K[j], RT[j], F[j])
        assert cam.RK_INV.dtype == np.float32 and cam.position.dtype == np.float64
        assert int(_ulps32(cam.RK_INV, c['RK_INV'][j]).max()) <= 4, (size, j)
        np.testing.assert_allclose(cam.position, c['position'][j], rtol=1e-12, atol=1e-12)
        # (n,17,3) -> (n,17,2) in (y, x): the host convenience projection against the straightforward formula
        pts = np.random.default_rng(j).normal(size=(2, 17, 3)) + np.array([0.0, 0.0, 1.0])
        hom = np.concatenate([pts, np.ones((2, 17, 1))], axis=2)
        ref = np.einsum('ij,nkj->nki', P[j].astype(np.float64), hom)
        np.testing.assert_allclose(cam.projectPoints_parallel(pts), (ref[..., :2] / ref[..., 2:3])[..., ::-1], rtol=1e-12, atol=1e-12)
