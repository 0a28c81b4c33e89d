"""MI355X-native per-frame multi-view 3D pose hot path (HRNet decode -> part-aware epipolar matching -> DLT),
behind the reference's ``ivclabpose`` surface.  Sub-modules are imported lazily: ``pam.synth`` is NumPy only,
``pam.ivclabpose`` / ``pam.tracker`` need the HIP C-ABI library (csrc/libpam_hip.so) and fail loudly without it."""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
__all__ = ['PKG_DIR', 'REPO_ROOT']
