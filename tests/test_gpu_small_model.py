"""cutie-small (cutie/config/model/small.yaml: ResNet-18 pixel encoder, multi-scale dims [256,128,64]) on the MI355X against the
oracle, which is pinned to the executed reference for this variant too (tests/golden/model_small.npz).  Tolerances: the reference's own reduced-precision envelope for
this variant (tests/golden/amp_envelope.json, see tests/test_gpu_parity.py).  (The file sorts after the base-model suites on purpose: it is the newest widening.)"""
import pytest
import torch

from cutie_amd import _lib
from cutie_amd.config import default_config
from oracle import scenarios as S
from oracle.inference import OracleProcessor, DEFAULT_CFG
from oracle.weights import MODEL_CFG_SMALL, make_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu_net():
    from cutie_amd.model.cutie import CUTIE
    _lib.set_executor_for_testing(None)
    net = CUTIE(default_config()).cuda().eval()
    net.load_weights(make_state_dict(seed=0))
    return net


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    assert torch.isfinite(a).all()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-6))


def test_small_model_on_gpu():
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.model.cutie import CUTIE
    from cutie_amd.utils.synth import SyntheticClip
    from oracle.net import OracleNet
    _lib.set_executor_for_testing(None)
    sd = make_state_dict(seed=0, m=MODEL_CFG_SMALL)
    onet = OracleNet(sd, MODEL_CFG_SMALL)
    net = CUTIE(default_config(model='small')).cuda().eval()
    net.load_weights(sd)
    assert not _lib.get_executor().is_mock
    with torch.inference_mode():
        img = SyntheticClip(128, 192, 3, 4, seed=5).frame(0).unsqueeze(0)
        ms, pix = net.encode_image(img.cuda())
        key, shr, sel = net.transform_key(ms[0])
        oms, opix = onet.encode_image(img)
        okey, oshr, osel = onet.transform_key(oms[0])
        assert [t.shape[1] for t in ms] == [256, 128, 64]
        for n, a, b in zip(['f16', 'f8', 'f4', 'pix', 'key', 'shr', 'sel'], [*ms, pix, key, shr, sel], [*oms, opix, okey, oshr, osel]):
            assert a.shape == b.shape and rel_err(a, b) < 3e-2, (n, rel_err(a, b))
    cfgs = lambda over: default_config(model='small', **over)
    outs, _ = S.run_scenario(lambda over: InferenceCore(net, cfg=cfgs(over)), 'small_fifo', device='cuda', make_cfg=cfgs)
    oouts, _ = S.run_scenario(lambda over: OracleProcessor(onet, dict(DEFAULT_CFG, **over)), 'small_fifo')
    for t, (p, o) in enumerate(zip(outs, oouts)):
        assert torch.isfinite(p).all()
        d = (p - o).abs()
        bmax, bmean, bmargin = S.trajectory_bounds('small')
        assert float(d.max()) < bmax and float(d.mean()) < bmean, (t, float(d.max()), float(d.mean()))
        assert S.q999(d) < S.trajectory_q999('small'), (t, S.q999(d))
        top2 = o.topk(2, dim=0)[0]
        confident = (top2[0] - top2[1]) > bmargin
        assert bool((p.argmax(0) == o.argmax(0))[confident].all()), t


@pytest.mark.parametrize('seed', [0, 2, 4])
def test_random_scripts_on_gpu(seed, gpu_net, oracle_net):
    """Random event scripts (oracle/fuzz_reference.py; the oracle agrees with the executed reference on them to 2e-6) through
    the HIP path: bank bookkeeping exact, probabilities within the trajectory tolerance."""
    from cutie_amd.inference.inference_core import InferenceCore
    from oracle.fuzz_reference import random_scenario
    net = gpu_net
    S.SCENARIOS['_fuzz'] = random_scenario(seed, 14)
    try:
        ps, os_ = [], []

        def psz(p):
            m = p.memory
            ps.append([sum(b.n_perm + b.n_work for b in m.buckets.values()), sum(b.n_perm for b in m.buckets.values()),
                       sum(b.n_long for b in m.buckets.values()), len(m.buckets)])

        def osz(p):
            os_.append([sum(p.work.size(b) for b in p.work.buckets), sum(p.work.perm_end[b] for b in p.work.buckets),
                        sum(p.long.size(b) for b in p.long.buckets) if p.use_long_term else 0, len(p.work.buckets)])

        oouts, _ = S.run_scenario(lambda over: OracleProcessor(oracle_net, dict(DEFAULT_CFG, **over)), '_fuzz', record=lambda t, p: osz(p))
        outs, _ = S.run_scenario(lambda over: InferenceCore(net, cfg=default_config(**over)), '_fuzz', device='cuda',
                                 record=lambda t, p: psz(p), make_cfg=lambda over: default_config(**over))
    finally:
        del S.SCENARIOS['_fuzz']
    assert ps == os_
    for t, (p, o) in enumerate(zip(outs, oouts)):
        assert torch.isfinite(p).all()
        d = (p - o).abs()
        bmax, bmean, _ = S.trajectory_bounds('base')
        assert float(d.max()) < bmax and float(d.mean()) < bmean, (seed, t, float(d.max()), float(d.mean()))


class _CudaInputs:
    """Hands the (CPU) tensors of the edge-case scripts to the processor as device tensors; everything else passes through."""

    def __init__(self, proc):
        object.__setattr__(self, '_p', proc)

    def __getattr__(self, k):
        return getattr(self._p, k)

    def step(self, image, mask=None, *a, **k):
        return self._p.step(image.cuda(), None if mask is None else mask.cuda(), *a, **k)


# (scripts whose step raises mid-frame leave that frame's features in the image feature store: the reference's own "Leaking ..." warning, image_feature_store.py:47-49)
@pytest.mark.filterwarnings('ignore:Leaking:UserWarning')
def test_edge_cases_on_gpu(gpu_net):
    """tests/golden/edge_cases.json (outcomes recorded from the executed reference) through the HIP path."""
    import json, os
    from cutie_amd.inference.inference_core import InferenceCore
    from oracle.edge_cases import CASES, INTENDED, run_case
    gold = json.load(open(os.path.join(S.GOLDEN_DIR, 'edge_cases.json')))
    net = gpu_net
    bad = {}
    for name in sorted(CASES):
        got = run_case(name, lambda over: _CudaInputs(InferenceCore(net, cfg=default_config(**over))))
        want = INTENDED[name][0] if name in INTENDED else gold[name]
        if got != want:
            bad[name] = (got, want)
    assert not bad, bad
