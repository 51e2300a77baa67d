"""Measure the reference's OWN reduced-precision envelope: the unmodified reference run under half-precision autocast
(what ``scripting_demo.py:13`` / ``eval_vos.py:112`` do on a GPU: ``torch.cuda.amp.autocast``) against the same reference in
fp32, on the scenario scripts of oracle/scenarios.py.  The free-running trajectory tolerance of tests/test_gpu_parity.py is
derived from this file's output, not fitted to the product.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python -m oracle.make_envelope            # -> tests/golden/amp_envelope.json

How autocast is reproduced on the CPU: ``torch.autocast('cpu', dtype)`` around every ``step`` (as the decorator of
``scripting_demo.py`` does), and ``torch.cuda.amp.autocast`` -- which the reference calls with ``enabled=False`` to keep its
fp32 islands (modules.py:62,79; object_summarizer.py:78; big_modules.py:289; tensor_utils.py:48) -- is pointed at the CPU
autocast context so that the islands stay fp32 exactly as they do on a GPU.  Two arms: bfloat16 (the product's storage type) and
float16 (what the reference's users actually run).

``python -m oracle.make_envelope one_step`` adds the teacher-forced arm: before every fp32 step a deep copy of the fp32 processor
runs the same step under autocast (a single reduced-precision step from the fp32 state, at every frame of every scenario) -- the
envelope that the teacher-forced per-frame tests (tests/teacher.py) are held to.

Recorded per scenario and arm, per frame: max / mean |prob_amp - prob_fp32|, argmax agreement over all pixels and over the pixels
whose fp32 top-1/top-2 margin exceeds each of a list of thresholds; plus the memory-bank sizes (must not depend on precision).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import json
import os
import sys

import numpy as np
import torch

from .make_golden import GOLDEN, _wrap, import_reference, memory_sizes, reference_cfg

MARGINS = [0.0, 0.02, 0.04, 0.1, 0.2, 0.3]


def main():
    torch.set_num_threads(os.cpu_count())
    CUTIE, InferenceCore = import_reference()
    from oracle.weights import MODEL_CFG_SMALL, make_state_dict
    from oracle import scenarios as S

    arms = {'bf16': torch.bfloat16, 'fp16': torch.float16}
    state = {'dtype': None}

    class _IslandCtx:
        """torch.cuda.amp.autocast(enabled=...) of the reference mapped onto the CPU autocast state."""

        def __init__(self, enabled=True, **kw):
            on = enabled and state['dtype'] is not None
            self.ctx = torch.autocast('cpu', dtype=state['dtype'] or torch.bfloat16, enabled=on)

        def __enter__(self):
            return self.ctx.__enter__()

        def __exit__(self, *a):
            return self.ctx.__exit__(*a)

    torch.cuda.amp.autocast = _IslandCtx

    def wrap(model):
        return lambda over: reference_cfg(model, **{k: (_wrap(v) if isinstance(v, dict) else v) for k, v in over.items()})

    class _Amp:
        """Processor proxy: every ``step`` runs under the arm's autocast (scripting_demo.py:13)."""

        def __init__(self, proc):
            object.__setattr__(self, '_p', proc)

        def __getattr__(self, k):
            return getattr(self._p, k)

        def step(self, *a, **k):
            if state['dtype'] is None:
                return self._p.step(*a, **k)
            with torch.autocast('cpu', dtype=state['dtype']):
                return self._p.step(*a, **k).float()

    nets = {}
    for model, mcfg in (('base', None), ('small', MODEL_CFG_SMALL)):
        net = CUTIE(reference_cfg(model)).eval()
        sd = make_state_dict(seed=0) if mcfg is None else make_state_dict(seed=0, m=mcfg)
        net.load_weights({k: v.clone() for k, v in sd.items()})
        nets[model] = net

    class _OneStep:
        """Teacher-forced arm: the fp32 processor is the teacher; before each of its steps a deep copy of its complete state
        runs the SAME step under autocast, so every frame is a single reduced-precision step from the fp32 state."""

        def __init__(self, proc, net, log):
            object.__setattr__(self, '_p', proc)
            object.__setattr__(self, '_net', net)
            object.__setattr__(self, '_log', log)

        def __getattr__(self, k):
            return getattr(self._p, k)

        def step(self, *a, **k):
            import copy
            outs = {}
            for arm, dt in arms.items():
                clone = copy.deepcopy(self._p, {id(self._net): self._net})
                state['dtype'] = dt
                with torch.autocast('cpu', dtype=dt):
                    outs[arm] = clone.step(*a, **k).float()
                state['dtype'] = None
            o = self._p.step(*a, **k)
            for arm, p in outs.items():
                d = (p - o).abs()
                self._log.setdefault(arm, []).append({'max': float(d.max()), 'mean': float(d.mean())})
            return o

    only = [a for a in sys.argv[1:]]
    one_step_mode = 'one_step' in only
    only = [a for a in only if a != 'one_step']
    jobs = [('base', n) for n in S.SCENARIOS] + [('small', n) for n in ('small_fifo', 'small_lt', 'small_add_del', 'small_lt_overlap',
                                                                         'small_cfg_fifo', 'small_clear')]
    out_path = os.path.join(GOLDEN, 'amp_envelope.json')
    if only == ['bounds']:
        return finish(json.load(open(out_path)), out_path)
    result = json.load(open(out_path)) if ((only or one_step_mode) and os.path.exists(out_path)) else {}
    for model, name in jobs:
        tag = name if model == 'base' else f'{model}:{name}'
        if only and tag not in only:
            continue
        net = nets[model]
        if one_step_mode:
            log = {}

            def make1(over):
                proc = InferenceCore(net, cfg=wrap(model)(over))
                if 'max_internal_size' in over:
                    proc.max_internal_size = over['max_internal_size']
                return _OneStep(proc, net, log)

            S.run_scenario(make1, name, make_cfg=wrap(model))
            result.setdefault('one_step_frames', {})[tag] = log
            for arm, rows in log.items():
                print(tag, 'one-step', arm, 'worst max %.4f  worst mean %.5f over %d steps' %
                      (max(r['max'] for r in rows), max(r['mean'] for r in rows), len(rows)), flush=True)
            json.dump(result, open(out_path, 'w'), indent=1)
            continue

        def make(over):
            proc = InferenceCore(net, cfg=wrap(model)(over))
            if 'max_internal_size' in over:
                proc.max_internal_size = over['max_internal_size']
            return _Amp(proc)

        runs = {}
        for arm in [None] + list(arms):
            state['dtype'] = arms[arm] if arm else None
            sizes = []
            outs, _ = S.run_scenario(make, name, record=lambda t, p: sizes.append(memory_sizes(p._p)), make_cfg=wrap(model))
            runs[arm] = (outs, sizes)
        state['dtype'] = None
        ref_outs, ref_sizes = runs[None]
        rec = {}
        for arm in arms:
            outs, sizes = runs[arm]
            frames = []
            for t, (p, o) in enumerate(zip(outs, ref_outs)):
                d = (p - o).abs()
                row = {'t': t, 'max': float(d.max()), 'mean': float(d.mean())}
                if o.shape[0] > 1:
                    top2 = o.topk(2, dim=0)[0]
                    margin = top2[0] - top2[1]
                    agree = p.argmax(0) == o.argmax(0)
                    row['agree'] = {str(m): [int((margin > m).sum()), int((agree & (margin > m)).sum())] for m in MARGINS}
                frames.append(row)
            rec[arm] = {'frames': frames, 'sizes_equal': sizes == ref_sizes,
                        'max': max(f['max'] for f in frames), 'mean': max(f['mean'] for f in frames)}
            print(tag, arm, 'worst max %.4f  worst mean %.5f  sizes_equal %s' % (rec[arm]['max'], rec[arm]['mean'], rec[arm]['sizes_equal']),
                  flush=True)
        result[tag] = rec
        json.dump(result, open(out_path, 'w'), indent=1)
    finish(result, out_path)


# product bound = SAFETY x the reference's own reduced-precision envelope.  Why not 1.0: the two are different samples of the same
# kind of perturbation (the product pre-rounds weights and keeps other fp32 islands than CPU autocast does), and the max norm over
# a few hundred frames of a recurrence with discontinuities (top-k membership, argmax-fed masks, usage-ranked pruning) is
# heavy-tailed: per scenario the product / reference-bf16 ratio of the worst frame scatters between 0.5 and 1.9.
SAFETY = 1.5         # free-running trajectories (errors compound through the recurrence): mean, 99.9th percentile, argmax margin
# The single worst pixel of a free-running trajectory is not a stable statistic: two builds of the product that differ ONLY in the fp32
# summation order of the 16-query side (round 3: CUTIE_AMD_QCHAIN=0 / 1, both fp32-class there) differ from each other on late frames by
# as much as either differs from the oracle -- small_fifo frame 4: 0.109 / 0.031, small_clear frame 4: 0.129 / 0.086, small_cfg_fifo
# worst frame: 0.070 / 0.188 (gpurun r3c22).  The single-pixel maximum therefore gets SAFETY_MAX; the 99.9th
# percentile of |dprob| per frame is held to SAFETY x the envelope's maximum (the bound the single pixel had in rounds 1-2).
SAFETY_MAX = 2.0
SAFETY_ONE_STEP = 1.25   # teacher-forced single steps


def finish(result, out_path):
    """Derive the product's free-running bounds from the measured envelope: per model variant, SAFETY x the worst deviation of
    the reference's own bf16 / fp16 autocast runs from its fp32 run over that variant's scenarios; the one-step bound from the
    first segmented frame of every scenario (frame 1: the only frame whose input state is still identical in both runs)."""
    scen = {k: v for k, v in result.items() if k not in ('bounds', 'one_step_frames')}
    bounds = {}
    for model in ('base', 'small'):
        rows = [v for k, v in scen.items() if (k.startswith('small:') if model == 'small' else ':' not in k)]
        if not rows:
            continue
        mx = max(r[a]['max'] for r in rows for a in ('bf16', 'fp16'))
        mn = max(r[a]['mean'] for r in rows for a in ('bf16', 'fp16'))
        bounds[model] = {'reference_envelope_max': mx, 'reference_envelope_mean': mn,
                         'trajectory_max': round(SAFETY_MAX * mx, 4), 'trajectory_q999': round(SAFETY * mx, 4),
                         'trajectory_mean': round(SAFETY * mn, 4), 'argmax_margin': round(2 * SAFETY * mx, 4)}
    # One-step bound (teacher-forced tests): the product stores activations in bf16, so it is held to the reference's own bf16
    # single-step deviation x SAFETY_ONE_STEP.  (The fp16 arm is recorded for information: with 10 mantissa bits it is usually
    # tighter, but its worst frame -- the GUI re-propagation step of small_clear -- is looser than any bf16 frame.)
    bounds['one_step'] = {}
    for model in ('base', 'small'):
        mine = lambda k: k.startswith('small:') if model == 'small' else ':' not in k
        one = {a: [r[a]['frames'][1] for k, r in scen.items() if mine(k) and len(r[a]['frames']) > 1] for a in ('bf16', 'fp16')}
        for k, log in result.get('one_step_frames', {}).items():    # teacher-forced arm: every frame is a single step
            if mine(k):
                for a, rows in log.items():
                    one[a] += rows
        if not one['bf16']:
            continue
        mx, mn = max(f['max'] for f in one['bf16']), max(f['mean'] for f in one['bf16'])
        bounds['one_step'][model] = {
            'reference_bf16_max': mx, 'reference_bf16_mean': mn,
            'reference_fp16_max': max(f['max'] for f in one['fp16']), 'reference_fp16_mean': max(f['mean'] for f in one['fp16']),
            'safety': SAFETY_ONE_STEP, 'max': round(SAFETY_ONE_STEP * mx, 4), 'mean': round(SAFETY_ONE_STEP * mn, 4),
            'argmax_margin': round(2 * SAFETY_ONE_STEP * mx, 4), 'steps_measured': len(one['bf16'])}
    result['bounds'] = bounds
    json.dump(result, open(out_path, 'w'), indent=1)
    print('bounds:', json.dumps(bounds, indent=1))


if __name__ == '__main__':
    main()
