import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_runtest_teardown(item, nextitem):
    """PAM_TEST_MEM=1: log the device memory in use after every test (stderr) -- captured graphs are never destroyed
    (pam._lib.immortal_graph), so what a test's networks captured stays allocated for the rest of the session."""
    if os.environ.get('PAM_TEST_MEM') != '1':
        return
    import torch
    if torch.cuda.is_available():
        free, total = torch.cuda.mem_get_info()
        sys.stderr.write('\n[mem] %-90s used %.1f GB of %.0f\n' % (item.nodeid[-90:], (total - free) / 2**30, total / 2**30))
