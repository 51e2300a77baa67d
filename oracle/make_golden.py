"""Generate the golden vectors under tests/golden/ by EXECUTING THE UNMODIFIED REFERENCE.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python -m oracle.make_golden

The reference (hkchengrex/Cutie, /root/reference) is imported on CPU/fp32 through the
hydra-free shim of SURVEY.md section 8c / Appendix B, loaded with the deterministic weights
of oracle/weights.py, and driven through the scenarios of oracle/scenarios.py.  Outputs:

  tests/golden/bike/*.jpg, 00000.png   input frames of examples/images/bike (data fixture)
  tests/golden/judo/*.jpg, *.png       examples/images/judo and examples/masks/judo (data fixture)
  tests/golden/<scenario>.npz          sub-sampled per-frame probabilities, argmax md5, memory sizes
  tests/golden/stages.npz              sampled per-stage tensors of the CUTIE facade methods
  tests/golden/state_dict_spec.json    reference state_dict key -> shape (pins oracle/weights.py)
  tests/golden/model_small.npz, state_dict_spec_small.json   the same for cutie-small (model/small.yaml, ResNet-18 pixel encoder)

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import json
import os
import shutil
import sys
import types

import numpy as np
import torch
import yaml

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, '..', 'tests', 'golden')


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    return DictConfig({k: _wrap(v) for k, v in o.items()}) if isinstance(o, dict) else o


def import_reference():
    om = types.ModuleType('omegaconf')
    om.DictConfig = DictConfig
    om.open_dict = None
    sys.modules['omegaconf'] = om
    # The repo root carries a drop-in `cutie` alias package, and the reference's `cutie` is a namespace package (no
    # __init__.py): a regular package anywhere on sys.path would win.  So: import what this script needs from the repo
    # first, then take the repo root off the path and let `cutie` resolve to the reference.
    root = os.path.abspath(os.path.join(HERE, '..'))
    if root not in [os.path.abspath(p or '.') for p in sys.path]:
        sys.path.append(root)
    import oracle.weights, oracle.scenarios, cutie_amd.utils.synth        # noqa: F401,E401
    sys.path = [REF] + [p for p in sys.path if os.path.abspath(p or '.') != root]
    for k in [k for k in sys.modules if k == 'cutie' or k.startswith('cutie.')]:
        del sys.modules[k]
    from cutie.model.utils import resnet
    resnet.load_weights_add_extra_dim = lambda *a, **k: None
    import torch.utils.model_zoo as mz
    mz.load_url = lambda *a, **k: {}
    from cutie.model.cutie import CUTIE
    from cutie.inference.inference_core import InferenceCore
    return CUTIE, InferenceCore


def reference_cfg(model_name='base', **over):
    R = REF + '/cutie/config/'
    m = yaml.safe_load(open(R + 'model/base.yaml'))
    if model_name != 'base':                                       # hydra `defaults: [base]` + overrides (model/small.yaml)
        extra = yaml.safe_load(open(R + f'model/{model_name}.yaml'))
        extra.pop('defaults')
        for k, v in extra.items():
            m[k] = {**m[k], **v} if isinstance(v, dict) else v
    m['object_transformer']['embed_dim'] = m['embed_dim']
    m['object_summarizer']['embed_dim'] = m['embed_dim']
    m['object_summarizer']['num_summaries'] = m['object_transformer']['num_queries']
    ev = yaml.safe_load(open(R + 'eval_config.yaml'))
    ev.pop('defaults')
    ev.pop('hydra')
    ev['model'] = m
    for k in ('use_long_term', 'mem_every'):
        ev[k] = ev['datasets']['d17-val'][k]
    for k, v in over.items():
        ev[k] = v
    return _wrap(ev)


def sample_tensor(t, n=96, seed=0):
    """Deterministic sparse probe of a tensor: (mean, std, n sampled entries)."""
    f = t.detach().float().flatten()
    idx = np.random.Generator(np.random.PCG64(seed)).integers(0, f.numel(), n)
    return np.concatenate([[f.mean().item(), f.std().item()], f[torch.from_numpy(idx)].numpy()]).astype(np.float32)


def memory_sizes(proc):
    """[work size, work perm, long size] summed over buckets + number of buckets (reference processor)."""
    mem = proc.memory
    w = sum(mem.work_mem.size(b) for b in mem.work_mem.buckets)
    p = sum(mem.work_mem.perm_size(b) for b in mem.work_mem.buckets)
    l = sum(mem.long_mem.size(b) for b in mem.long_mem.buckets) if mem.use_long_term else 0
    return [w, p, l, len(mem.work_mem.buckets)]


def golden_model_small(CUTIE, InferenceCore):
    """cutie-small (model/small.yaml: ResNet-18 pixel encoder): key set, encoder stage probes and one trajectory
    (the small_fifo script) -> tests/golden/model_small.npz, state_dict_spec_small.json."""
    from oracle.weights import MODEL_CFG_SMALL, make_state_dict, param_spec
    from oracle import scenarios as S
    from cutie_amd.utils.synth import SyntheticClip
    net = CUTIE(reference_cfg('small')).eval()
    ref_sd = net.state_dict()
    spec = param_spec(MODEL_CFG_SMALL)
    assert set(ref_sd.keys()) == set(spec.keys()), (set(ref_sd) ^ set(spec))
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k][0]), (k, v.shape, spec[k])
    json.dump({k: list(v.shape) for k, v in ref_sd.items()},
              open(os.path.join(GOLDEN, 'state_dict_spec_small.json'), 'w'), indent=0)
    net.load_weights({k: v.clone() for k, v in make_state_dict(seed=0, m=MODEL_CFG_SMALL).items()})
    print('reference cutie-small state_dict:', len(ref_sd), 'tensors,', sum(v.numel() for v in ref_sd.values()) / 1e6, 'M')
    sizes = []
    wrap = lambda over: reference_cfg('small', **{k: (_wrap(v) if isinstance(v, dict) else v) for k, v in over.items()})
    outs, proc = S.run_scenario(lambda over: InferenceCore(net, cfg=wrap(over)), 'small_fifo',
                                record=lambda t, p: sizes.append(memory_sizes(p)), make_cfg=wrap)
    rec = S.summarize(outs, S.SCENARIOS['small_fifo']['sub'])
    rec['mem_sizes'] = np.array(sizes, dtype=np.int64)
    with torch.inference_mode():
        img = SyntheticClip(128, 192, 3, 4, seed=5).frame(0).unsqueeze(0)
        ms, pix = net.encode_image(img)
        key, shr, sel = net.transform_key(ms[0])
        for n, t in zip(['f16', 'f8', 'f4', 'pix_feat', 'key', 'shrinkage', 'selection'], [*ms, pix, key, shr, sel]):
            rec['stage_' + n] = sample_tensor(t)
    np.savez_compressed(os.path.join(GOLDEN, 'model_small.npz'), **rec)
    print('model_small frames', len(outs), 'mem', sizes[-1], 'hist', rec[f'hist_{len(outs) - 1}'])


def main():
    torch.set_num_threads(os.cpu_count())
    os.makedirs(os.path.join(GOLDEN, 'bike'), exist_ok=True)
    for n in sorted(os.listdir(REF + '/examples/images/bike')):
        shutil.copyfile(f'{REF}/examples/images/bike/{n}', os.path.join(GOLDEN, 'bike', n))
    shutil.copyfile(f'{REF}/examples/masks/bike/00000.png', os.path.join(GOLDEN, 'bike', '00000.png'))
    for n in os.listdir(os.path.join(GOLDEN, 'bike')):
        os.chmod(os.path.join(GOLDEN, 'bike', n), 0o644)
    os.makedirs(os.path.join(GOLDEN, 'judo'), exist_ok=True)       # examples/images/judo + its four mask files (data fixture, C0b)
    for sub in ('images', 'masks'):
        for n in sorted(os.listdir(f'{REF}/examples/{sub}/judo')):
            shutil.copyfile(f'{REF}/examples/{sub}/judo/{n}', os.path.join(GOLDEN, 'judo', n))
            os.chmod(os.path.join(GOLDEN, 'judo', n), 0o644)

    CUTIE, InferenceCore = import_reference()
    from oracle.weights import make_state_dict, param_spec
    from oracle import scenarios as S

    net = CUTIE(reference_cfg()).eval()
    ref_sd = net.state_dict()
    spec = param_spec()
    assert set(ref_sd.keys()) == set(spec.keys()), (set(ref_sd) ^ set(spec))
    for k, v in ref_sd.items():
        assert tuple(v.shape) == tuple(spec[k][0]), (k, v.shape, spec[k])
    json.dump({k: list(v.shape) for k, v in ref_sd.items()},
              open(os.path.join(GOLDEN, 'state_dict_spec.json'), 'w'), indent=0)
    sd = make_state_dict(seed=0)
    net.load_weights({k: v.clone() for k, v in sd.items()})
    print('reference state_dict:', len(ref_sd), 'tensors,', sum(v.numel() for v in ref_sd.values()) / 1e6, 'M')

    # ---- scenario trajectories ------------------------------------------------------
    only = [a for a in sys.argv[1:] if a in S.SCENARIOS]          # e.g. `python oracle/make_golden.py small_interactive`
    for name, sc in S.SCENARIOS.items():
        if (only or 'model_small' in sys.argv[1:]) and name not in only:
            continue
        sizes = []
        net_sc = net
        if sc.get('weights') == 'decisive':                          # the reference under the decisive weights (oracle/make_decisive_weights.py)
            net_sc = CUTIE(reference_cfg()).eval()
            net_sc.load_weights({k: v.clone() for k, v in S.decisive_state_dict().items()})

        def make(over, net=net_sc):
            proc = InferenceCore(net, cfg=reference_cfg(**{k: (_wrap(v) if isinstance(v, dict) else v)
                                                           for k, v in over.items()}))
            if 'max_internal_size' in over:
                proc.max_internal_size = over['max_internal_size']
            return proc

        wrap = lambda over: reference_cfg(**{k: (_wrap(v) if isinstance(v, dict) else v) for k, v in over.items()})
        outs, proc = S.run_scenario(make, name, record=lambda t, p: sizes.append(memory_sizes(p)), make_cfg=wrap)
        rec = S.summarize(outs, sc['sub'])
        rec['mem_sizes'] = np.array(sizes, dtype=np.int64)
        np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **rec)
        print(name, 'frames', len(outs), 'shape', tuple(outs[-1].shape), 'mem', sizes[-1],
              'hist', rec[f'hist_{len(outs) - 1}'])

    if not only or 'model_small' in sys.argv[1:]:
        golden_model_small(CUTIE, InferenceCore)
    if only or 'model_small' in sys.argv[1:]:
        return
    # ---- per-stage probes of the facade methods -----------------------------------------
    from cutie_amd.utils.synth import SyntheticClip
    clip = SyntheticClip(128, 192, 3, 4, seed=5)
    g = torch.Generator().manual_seed(11)
    rec = {}
    with torch.inference_mode():
        img = clip.frame(0).unsqueeze(0)
        ms, pix = net.encode_image(img)
        key, shr, sel = net.transform_key(ms[0])
        for n, t in zip(['f16', 'f8', 'f4', 'pix_feat', 'key', 'shrinkage', 'selection'], [*ms, pix, key, shr, sel]):
            rec[n] = sample_tensor(t)
        K, h, w = 3, 8, 12
        masks = torch.stack([(clip.first_mask() == i + 1).float() for i in range(K)], 0).unsqueeze(0)
        masks = masks * 0.9 + 0.05
        sens = torch.randn(1, K, 256, h, w, generator=g) * 0.5
        val, nsens, summ, _ = net.encode_mask(img, pix, sens, masks)
        rec['mask_value'], rec['deep_sensory'], rec['summaries'] = sample_tensor(val), sample_tensor(nsens), sample_tensor(summ)
        ro = torch.randn(1, K, 256, h, w, generator=g) * 0.5
        fused = net.pixel_fusion(pix, ro, sens, masks)
        rec['fused'] = sample_tensor(fused)
        rq, aux = net.readout_query(fused, summ.unsqueeze(2))
        rec['readout_query'] = sample_tensor(rq)
        for i, lg in enumerate(aux['logits']):
            rec[f'aux_logits_{i}'] = sample_tensor(lg)
        s2, lg, prob = net.segment(ms, rq, sens, update_sensory=True)
        rec['seg_sensory'], rec['seg_logits'], rec['seg_prob'] = sample_tensor(s2), sample_tensor(lg), sample_tensor(prob)
        # affinity math on its own (memory_utils.py)
        from cutie.model.utils.memory_utils import get_similarity, do_softmax
        mk = torch.randn(1, 64, 500, generator=g)
        msh = torch.rand(1, 1, 500, generator=g) * 2 + 1
        qk = torch.randn(1, 64, 96, generator=g)
        qe = torch.rand(1, 64, 96, generator=g)
        sim = get_similarity(mk, msh, qk, qe)
        rec['similarity'] = sample_tensor(sim)
        aff, usage = do_softmax(sim, top_k=30, inplace=False, return_usage=True)
        rec['affinity'], rec['usage'] = sample_tensor(aff), sample_tensor(usage)
    np.savez_compressed(os.path.join(GOLDEN, 'stages.npz'), **rec)
    print('stages:', sorted(rec.keys()))


if __name__ == '__main__':
    main()
