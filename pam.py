"""Import alias: ``import pam`` loads the package directory
``part-aware_measurement_for_3d_pose_estimation_and_tracking_amd/`` (its name is not a valid Python identifier)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    'part-aware_measurement_for_3d_pose_estimation_and_tracking_amd')
_spec = importlib.util.spec_from_file_location('pam', os.path.join(_DIR, '__init__.py'),
                                               submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['pam'] = _mod
_spec.loader.exec_module(_mod)
