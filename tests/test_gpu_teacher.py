"""Teacher-forced per-frame parity on the MI355X (tests/teacher.py): before every frame the oracle's complete recurrent state is
copied into the product, the product runs ONE ``step`` through the HIP path and is compared with the oracle's result for that
frame -- at the small scenario sizes for both model variants, and at the BASELINE sizes (480p / 3 objects / long-term memory
through the first consolidation; 1080p / 5 objects).

Bound: the reference's OWN single-step reduced-precision deviation, measured by oracle/make_envelope.py (``one_step`` arm: a deep
copy of the fp32 reference runs one step under bf16 / fp16 autocast from the fp32 state, at every frame of every scenario) and
committed in tests/golden/amp_envelope.json, times the safety factor stored there (1.25; rationale in oracle/make_envelope.py).  Argmax object ids must be identical
wherever the oracle's top-1 / top-2 margin exceeds twice the max bound.  Bank bookkeeping must be exact at every frame.
"""
import json
import os

import pytest
import torch

from cutie_amd import _lib
from cutie_amd.config import default_config
from oracle import scenarios as S
from oracle.inference import OracleProcessor, DEFAULT_CFG
from oracle.weights import MODEL_CFG_SMALL, make_state_dict
import teacher

pytestmark = pytest.mark.gpu
ALL_BOUNDS = json.load(open(os.path.join(S.GOLDEN_DIR, 'amp_envelope.json')))['bounds']['one_step']      # per model variant


def _nets(model):
    from cutie_amd.model.cutie import CUTIE
    from oracle.net import OracleNet
    _lib.set_executor_for_testing(None)
    if model == 'small':
        sd = make_state_dict(seed=0, m=MODEL_CFG_SMALL)
        return CUTIE(default_config(model='small')).cuda().eval(), OracleNet(sd, MODEL_CFG_SMALL), sd, (lambda over: default_config(model='small', **over))
    sd = make_state_dict(seed=0)
    return CUTIE(default_config()).cuda().eval(), OracleNet(sd), sd, (lambda over: default_config(**over))


_cache = {}


def nets(model):
    if model not in _cache:
        net, onet, sd, cfgs = _nets(model)
        net.load_weights(sd)
        _cache[model] = (net, onet, cfgs)
    return _cache[model]


# Ratchet: the worst per-frame deviations OBSERVED on the MI355X at the end of round 4 (tests/golden/observed_r04.json, written by
# `CUTIE_RECORD_OBSERVED=path pytest tests/test_gpu_teacher.py`).  A run may exceed neither the reference's reduced-precision
# envelope (above) nor 1.5 x what this build actually did -- a numerical regression that doubles the error inside the envelope fails.
_OBS_PATH = os.path.join(S.GOLDEN_DIR, 'observed_r04.json')
_RECORD = os.environ.get('CUTIE_RECORD_OBSERVED')
# (re-recording the baseline after a change that moves roundings -- round 4: the positional terms inside the transformer's projection
# convs, errors moved by a few per cent in both directions -- runs without the old baseline; the envelope bounds stay in force)
OBSERVED = json.load(open(_OBS_PATH)) if (os.path.exists(_OBS_PATH) and not (_RECORD and os.environ.get('CUTIE_REBASE_OBSERVED'))) else {}
_recorded = {}


def check_rows(rows, tag, model='base'):
    BOUNDS = ALL_BOUNDS[model]
    wmax, wmean = max(r['max'] for r in rows), max(r['mean'] for r in rows)
    if _RECORD:
        _recorded[tag] = dict(max=wmax, mean=wmean, agree_all=min(r.get('agree_all', 1.0) for r in rows))
        json.dump(_recorded, open(_RECORD, 'w'), indent=1, sort_keys=True)
    if tag in OBSERVED:
        o = OBSERVED[tag]
        assert wmax <= 1.5 * o['max'] + 2e-3 and wmean <= 1.5 * o['mean'] + 2e-4, ('ratchet', tag, wmax, wmean, o)
    worst = max(rows, key=lambda r: r['max'])
    print(tag, 'worst frame', worst['t'], 'max %.4f' % worst['max'], 'worst mean %.5f' % max(r['mean'] for r in rows),
          'bound', BOUNDS['max'], BOUNDS['mean'], '| per frame:', [(r['t'], round(r['max'], 4), round(r['mean'], 5)) for r in rows])
    for r in rows:
        assert r['sizes_equal'], (tag, r['t'], 'memory-bank sizes differ')
        assert r['max'] <= BOUNDS['max'] and r['mean'] <= BOUNDS['mean'], (tag, r)
        if 'flips_above_margin' in r:
            assert r['flips_above_margin'][BOUNDS['argmax_margin']] == 0, (tag, r)
        if 'sensory_rel' in r:
            # recurrent state after the step: max |d| / max |sensory| -- a diagnostic (the reference's own sensory deviation was not
            # recorded): loose for cutie-small, whose synthetic weights make single steps 2-3x more sensitive (see the envelope)
            assert r['sensory_rel'] < (0.15 if model == 'base' else 0.35), (tag, r)


@pytest.mark.parametrize('model', ['base', 'small'])
@pytest.mark.parametrize('name', ['small_fifo', 'small_lt', 'small_add_del', 'small_lt_overlap', 'small_clear:4', 'small_cfg_fifo:5', 'small_flip', 'small_chunk',
                                  'small_interactive:5', 'small_misc:5', 'small_video:6'])
def test_teacher_forced_scenarios(name, model):
    """name[:n] = the first n frames of a scenario (the plain part before an event the harness does not replay)."""
    from cutie_amd.inference.inference_core import InferenceCore
    net, onet, cfgs = nets(model)
    name, _, nf = name.partition(':')
    over = S.SCENARIOS[name]['cfg']
    steps, deletes = S.scenario_inputs(name)
    steps = steps[:int(nf)] if nf else steps
    rows = teacher.run_teacher_forced(steps, lambda: OracleProcessor(onet, dict(DEFAULT_CFG, **over)),
                                      lambda: InferenceCore(net, cfg=cfgs(over)), 'cuda', deletes=deletes, margins=(ALL_BOUNDS[model]['argmax_margin'],))
    check_rows(rows, f'{model}:{name}', model)


def _clip_steps(h, w, k, frames, seed=1):
    from cutie_amd.utils.synth import SyntheticClip
    clip = SyntheticClip(h, w, k, frames, seed=seed)
    return [(clip.frame(0), clip.first_mask(), clip.objects)] + [(clip.frame(t), None, None) for t in range(1, frames)]


def test_teacher_forced_480p_long_term():
    """BASELINE C2 size: 854x480, 3 objects, long-term memory with the eval_config defaults, through frame 50: 1620-token memory
    frames, the first consolidation (frame 45: 8100 candidates -> 128 prototypes) and reads from the long-term region, every
    frame value-checked against the oracle from the oracle's state."""
    from cutie_amd.inference.inference_core import InferenceCore
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    net, onet, cfgs = nets('base')
    over = dict(use_long_term=True)
    rows = teacher.run_teacher_forced(_clip_steps(480, 854, 3, 51), lambda: OracleProcessor(onet, dict(DEFAULT_CFG, **over)),
                                      lambda: InferenceCore(net, cfg=cfgs(over)), 'cuda', margins=(ALL_BOUNDS['base']['argmax_margin'],))
    assert len(rows) == 51
    check_rows(rows, '480p K=3 LT')


def test_teacher_forced_judo():
    """examples/images/judo driven like scripting_demo_add_del_objects.py -- 16 real 480p frames, ids 1..4 given
    by the mask files of frames 0, 5, 8, 13 (three buckets; the mask of frame 5 is one pixel narrower than the frames), id 1 deleted
    before frame 10 -- every frame value-checked against the oracle from the oracle's state, bank sizes exact."""
    from cutie_amd.inference.inference_core import InferenceCore
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    net, onet, cfgs = nets('base')
    over = S.SCENARIOS['judo']['cfg']
    steps, deletes = S.scenario_inputs('judo')
    assert len(steps) == 16 and [t for t, s_ in enumerate(steps) if s_[1] is not None] == [0, 5, 8, 13] and deletes == {10: [1]}
    rows = teacher.run_teacher_forced(steps, lambda: OracleProcessor(onet, dict(DEFAULT_CFG, **over)),
                                      lambda: InferenceCore(net, cfg=cfgs(over)), 'cuda', deletes=deletes, margins=(ALL_BOUNDS['base']['argmax_margin'],))
    assert len(rows) == 16
    check_rows(rows, 'judo')


def test_teacher_forced_1080p():
    """BASELINE C4 size: 1920x1080, 5 objects (8160 queries per frame), 7 frames incl. the second memory frame."""
    from cutie_amd.inference.inference_core import InferenceCore
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    net, onet, cfgs = nets('base')
    rows = teacher.run_teacher_forced(_clip_steps(1080, 1920, 5, 7, seed=2), lambda: OracleProcessor(onet, dict(DEFAULT_CFG)),
                                      lambda: InferenceCore(net, cfg=cfgs({})), 'cuda', margins=(ALL_BOUNDS['base']['argmax_margin'],))
    check_rows(rows, '1080p K=5')


def test_teacher_forced_480p_one_object():
    """BASELINE configs[1]: 854x480, ONE object (M = 1620 per-object layers: other tile-table entries than the 3-object run), 30 frames
    = 5 memory frames, every frame value-checked against the oracle from the oracle's state."""
    from cutie_amd.inference.inference_core import InferenceCore
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    net, onet, cfgs = nets('base')
    rows = teacher.run_teacher_forced(_clip_steps(480, 854, 1, 30, seed=4), lambda: OracleProcessor(onet, dict(DEFAULT_CFG)),
                                      lambda: InferenceCore(net, cfg=cfgs({})), 'cuda', margins=(ALL_BOUNDS['base']['argmax_margin'],))
    assert len(rows) == 30
    check_rows(rows, '480p K=1')


@pytest.mark.parametrize('h,w,k,size,frames', [(1080, 1920, 2, 480, 6), (100, 120, 2, 80, 8)])
def test_teacher_forced_internal_resize(h, w, k, size, frames):
    """The max_internal_size path (inference_core.py:206-228, 321-326; scripting_demo.py:21): frame and index mask are resized on the
    way in (bilinear / nearest-exact), the probabilities on the way out -- the RESIZE kernel on the GPU against the oracle's
    F.interpolate, teacher-forced."""
    from cutie_amd.inference.inference_core import InferenceCore
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    net, onet, cfgs = nets('base')
    over = dict(max_internal_size=size, mem_every=3)
    rows = teacher.run_teacher_forced(_clip_steps(h, w, k, frames, seed=6), lambda: OracleProcessor(onet, dict(DEFAULT_CFG, **over)),
                                      lambda: InferenceCore(net, cfg=cfgs(over)), 'cuda', margins=(ALL_BOUNDS['base']['argmax_margin'],))
    check_rows(rows, f'resize {h}x{w}->{size}')
