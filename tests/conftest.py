import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    import torch
    from mdgat_matcher_amd.synth import effective_cpu_count
    torch.set_num_threads(effective_cpu_count())      # the CPU oracle: never oversubscribe a cgroup quota


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
