import os
import sys
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle_net():
    import torch
    from oracle.weights import make_state_dict
    from oracle.net import OracleNet
    torch.manual_seed(0)
    return OracleNet(make_state_dict(seed=0))
