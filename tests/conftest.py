import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
GPU_TESTS = set()     # node ids of the tests that carry the gpu marker (the parity report covers exactly these)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    return load


@pytest.fixture(autouse=True)
def _release_device_temporaries(request):
    import helpers
    helpers.CURRENT_TEST[0] = request.node.nodeid
    if request.node.get_closest_marker('gpu') is not None:
        GPU_TESTS.add(request.node.nodeid)
    yield
    helpers._KEEP.clear()


def pytest_sessionfinish(session, exitstatus):
    """Parity report: the observed error of every comparison the suite made, beside its bound.  Written when GPU
    tests ran (gpurun merges gpurun_out/ back; the copy that is judged is committed under profiles/rNN/)."""
    import json
    import helpers
    rows = [r for r in helpers.REPORT if r['test'] in GPU_TESTS or 'gpu' in r['test']]
    if not rows:
        return
    out_dir = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out_dir, exist_ok=True)
        worst = {}
        for r in rows:
            k = (r['test'].split('::')[-1], r['what'])
            if k not in worst or r['observed'] > worst[k]['observed']:
                worst[k] = r
        with open(os.path.join(out_dir, 'parity_report.json'), 'w') as f:
            json.dump({'exit_status': int(exitstatus), 'n_comparisons': len(rows),
                       'comparisons': sorted(worst.values(), key=lambda r: (r['test'], r['what']))}, f, indent=1)
    except OSError:
        pass
