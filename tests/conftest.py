import os
import sys
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                   # cgroup v2 quota
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def pytest_configure(config):
    # the GPU boxes expose 256 logical CPUs; torch's default (one thread per CPU) makes the CPU-side oracle /
    # interpreter crawl when several processes share them
    import torch
    torch.set_num_threads(max(1, min(8, usable_cores())))
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle_net():
    import torch
    from oracle.weights import make_state_dict
    from oracle.net import OracleNet
    torch.manual_seed(0)
    return OracleNet(make_state_dict(seed=0))
