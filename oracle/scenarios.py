"""Scripted clips used to pin the oracle against the executed reference and to parity-test
the HIP path against the oracle.  TEST INFRASTRUCTURE (see oracle/__init__.py).

A scenario is (cfg overrides, frame source, event script).  ``run_scenario`` drives any
processor exposing the reference ``InferenceCore`` surface (``step``, ``delete_objects``,
``output_prob_to_mask``) -- the reference itself, ``OracleProcessor`` or the HIP product.
"""
import hashlib
import os
import numpy as np
import torch

from cutie_amd.utils.synth import SyntheticClip

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')

LT_SMALL = dict(count_usage=True, max_mem_frames=4, min_mem_frames=2, num_prototypes=8,
                max_num_tokens=40, buffer_tokens=12)

SCENARIOS = {
    # configs[0] of BASELINE.json: examples/bike, 480p, (2) objects, scripting_demo.py settings
    'bike': dict(cfg=dict(max_internal_size=480), kind='bike', frames=4, sub=8),
    # the same clip under the DECISIVE weights (decisive_state_dict): the oracle's margin exceeds 0.33 on >= 95 % of every frame
    'bike_decisive': dict(cfg=dict(max_internal_size=480), kind='bike', frames=4, sub=8, weights='decisive'),
    # examples/images/judo through scripting_demo_add_del_objects.py:21-64 -- 16 real 480p frames, ids 1..4
    # given by the mask files of frames 0, 5, 8, 13 (three buckets of objects added at different times), id 1 deleted before frame 10.
    # (masks/judo/00005.png is 480 x 853 against 480 x 854 frames: the reference pads the mask on its own, inference_core.py:263)
    'judo': dict(cfg=dict(max_internal_size=480), kind='judo', frames=16, sub=8, delete_at={10: [1]}),
    # ... and under the decisive weights (round 5: a second whole-frame argmax scenario -- 16 real frames, four objects in three buckets,
    # a deletion; the executed reference's top-1 / top-2 margin exceeds 0.33 on 89 ... 100 % of the pixels of every frame)
    'judo_decisive': dict(cfg=dict(max_internal_size=480), kind='judo', frames=16, sub=8, delete_at={10: [1]}, weights='decisive'),
    # small clip, 3 objects, FIFO working memory wraps (mem_every=2, max_mem_frames=3)
    'small_fifo': dict(cfg=dict(mem_every=2, max_mem_frames=3, stagger_updates=1),
                       kind='synth', h=96, w=136, k=3, frames=14, sub=2),
    # long-term memory with scaled-down limits: consolidation and pruning both fire
    'small_lt': dict(cfg=dict(mem_every=2, use_long_term=True, long_term=LT_SMALL),
                     kind='synth', h=96, w=128, k=3, frames=44, sub=2),
    # objects added at different frames (multi-bucket), mask merge, delete
    # (scripting_demo_add_del_objects.py pattern)
    'small_add_del': dict(cfg=dict(mem_every=3), kind='synth', h=112, w=128, k=4, frames=16, sub=2,
                          add_at={0: [1], 4: [2], 7: [3, 4]}, delete_at={10: [1]}),
    # interactive surface (gui/main_controller.py:303-368, scripts/process_video.py): a corrected mask committed as permanent
    # memory (force_permanent -> prepend to the permanent part), update_config (mem_every 2 -> 3), clear_non_permanent_memory
    'small_interactive': dict(cfg=dict(mem_every=2, max_mem_frames=3), kind='synth', h=96, w=128, k=3, frames=15, sub=2,
                              add_at={0: [1, 2, 3], 5: [1, 2, 3]}, permanent_at=[5], update_config_at={8: dict(mem_every=3)},
                              clear_non_permanent_at=[11]),
    # chunk_size > 0: the reference's MemoryManager.read runs fusion + object transformer per group of 2 objects (the "others"
    # mask and the attention masks only see the group), so the result differs from chunk_size = -1
    'small_chunk': dict(cfg=dict(mem_every=2, max_mem_frames=3, chunk_size=2), kind='synth', h=96, w=136, k=3, frames=10, sub=2),
    # remaining step() arguments: soft float masks (idx_mask=False) on the first and on a later frame, an index mask that
    # re-specifies every known object (need_segment False), end=True on the last frame (no memory / sensory update)
    'small_misc': dict(cfg=dict(mem_every=3, max_mem_frames=3), kind='synth', h=96, w=120, k=3, frames=12, sub=2,
                       add_at={0: [1, 2, 3], 5: [1, 2, 3], 8: [1, 2, 3]}, float_mask_at=[0, 8], end_at=[11]),
    # GUI patterns (gui/main_controller.py:297-306,542-543): "propagate" = clear_sensory_memory() then step(image, previous
    # prob[1:], idx_mask=False) with objects=None; "reset all memory" = clear_memory() followed by a fresh index mask
    'small_clear': dict(cfg=dict(mem_every=3, max_mem_frames=3), kind='synth', h=96, w=120, k=2, frames=13, sub=2,
                        add_at={0: [1, 2], 8: [1, 2]}, repropagate_at=[4], clear_memory_at=[8]),
    # scripts/process_video.py:94-120,139-176: every given mask is first committed to PERMANENT memory (one-hot float planes,
    # objects=None, force_permanent=True), then the video runs from frame 0 and meets the same masks again; long-term memory on
    'small_video': dict(cfg=dict(mem_every=3, use_long_term=True, long_term=LT_SMALL), kind='synth', h=96, w=120, k=2,
                        frames=14, sub=2, add_at={0: [1, 2], 6: [1, 2]}, float_mask_at=[0, 6], float_kind='onehot',
                        precommit=[0, 6]),
    # long-term memory where fewer tokens are consolidated than kept (min_mem_frames > max_mem_frames / 2): the compaction of
    # the working region moves the kept tokens over a distance smaller than their extent (overlapping ranges)
    'small_lt_overlap': dict(cfg=dict(mem_every=2, use_long_term=True, long_term=dict(LT_SMALL, max_mem_frames=6, min_mem_frames=5)),
                             kind='synth', h=96, w=128, k=2, frames=30, sub=2),
    # memory limits changed mid-clip through update_config (the GUI's memory sliders, gui/main_controller.py:518-560): the token
    # limits are re-derived at the next memory frame (memory_manager.py:228-235) -- FIFO ring grown 2 -> 4 frames, then cut to 1
    'small_cfg_fifo': dict(cfg=dict(mem_every=2, max_mem_frames=3), kind='synth', h=96, w=120, k=2, frames=18, sub=2,
                           update_config_at={5: dict(max_mem_frames=5), 13: dict(max_mem_frames=2)}),
    # ... and with long-term memory: larger working set + more long-term tokens, then both cut below what is stored
    'small_cfg_lt': dict(cfg=dict(mem_every=2, use_long_term=True, long_term=LT_SMALL), kind='synth', h=96, w=120, k=2, frames=34, sub=2,
                         update_config_at={9: dict(long_term=dict(LT_SMALL, max_mem_frames=6, min_mem_frames=5, max_num_tokens=64)),
                                           23: dict(long_term=dict(LT_SMALL, max_mem_frames=3, min_mem_frames=2, max_num_tokens=24,
                                                                   buffer_tokens=4))}),
    # flip augmentation (bs = 2 in the reference) with long-term memory; width 121 -> asymmetric pad (3 | 4)
    'small_flip': dict(cfg=dict(mem_every=2, flip_aug=True, use_long_term=True, long_term=LT_SMALL),
                       kind='synth', h=96, w=121, k=2, frames=26, sub=2),
}


def decisive_state_dict(seed=0):
    """The synthetic state dict with the fitted mask-decoder head of oracle/make_decisive_weights.py (tests/golden/decisive_delta.npz):
    weights under which the oracle is decisive on almost every pixel, so that argmax object ids can be compared over whole frames."""
    from oracle.weights import make_state_dict
    sd = make_state_dict(seed=seed)
    d = np.load(os.path.join(GOLDEN_DIR, 'decisive_delta.npz'))
    for k in d.files:
        assert k in sd and tuple(sd[k].shape) == tuple(d[k].shape), k
        sd[k] = torch.from_numpy(d[k]).clone()
    return sd


def trajectory_bounds(model='base'):
    """(max |dprob|, mean |dprob|, argmax margin) a free-running trajectory of the product is held to, derived from the deviation of
    the reference's OWN bf16 / fp16 autocast runs from its fp32 run on these scenarios (oracle/make_envelope.py, committed in
    tests/golden/amp_envelope.json per model variant): mean and argmax margin at 1.5 x the envelope (margin = twice the per-class
    bound), the single worst pixel at 2 x (it is not a stable statistic, see make_envelope.SAFETY_MAX); trajectory_q999() holds
    the 99.9th percentile to 1.5 x the envelope's maximum."""
    import json
    b = json.load(open(os.path.join(GOLDEN_DIR, 'amp_envelope.json')))['bounds'][model]
    return b['trajectory_max'], b['trajectory_mean'], b['argmax_margin']


def trajectory_q999(model='base'):
    import json
    return json.load(open(os.path.join(GOLDEN_DIR, 'amp_envelope.json')))['bounds'][model]['trajectory_q999']


def q999(d):
    """99.9th percentile of a tensor of absolute deviations (kthvalue: exact, no interpolation)."""
    flat = d.reshape(-1)
    k = max(1, int(round(0.999 * flat.numel())))
    return float(flat.float().kthvalue(k)[0])


def _bike_frames():
    from PIL import Image
    d = os.path.join(GOLDEN_DIR, 'bike')
    names = sorted(n for n in os.listdir(d) if n.endswith('.jpg'))
    imgs = [torch.from_numpy(np.array(Image.open(os.path.join(d, n)).convert('RGB'))).permute(2, 0, 1).float() / 255
            for n in names]
    mask = torch.from_numpy(np.array(Image.open(os.path.join(d, '00000.png')))).long()
    return imgs, mask


def _judo_steps():
    """The inputs of scripting_demo_add_del_objects.py:29-60: every frame of examples/images/judo, with the mask file of the same
    name where one exists; objects = the unique non-zero values of that mask."""
    from PIL import Image
    d = os.path.join(GOLDEN_DIR, 'judo')
    steps = []
    for n in sorted(x for x in os.listdir(d) if x.endswith('.jpg')):
        img = torch.from_numpy(np.array(Image.open(os.path.join(d, n)).convert('RGB'))).permute(2, 0, 1).float() / 255
        mp = os.path.join(d, n[:-4] + '.png')
        if os.path.exists(mp):
            m = np.array(Image.open(mp))
            objs = [int(o) for o in np.unique(m) if o != 0]
            steps.append((img, torch.from_numpy(m).long(), objs))
        else:
            steps.append((img, None, None))
    return steps


def scenario_inputs(name):
    """-> list of (image, mask_or_None, objects_or_None), dict frame->objects to delete (before the step)"""
    sc = SCENARIOS[name]
    if sc['kind'] == 'judo':
        return _judo_steps(), sc.get('delete_at', {})
    if sc['kind'] == 'bike':
        imgs, mask = _bike_frames()
        objs = [int(o) for o in np.unique(mask.numpy()) if o != 0]
        return [(imgs[0], mask, objs)] + [(im, None, None) for im in imgs[1:]], {}
    clip = SyntheticClip(sc['h'], sc['w'], sc['k'], sc['frames'], seed=3)
    full = clip.first_mask()
    add_at = sc.get('add_at', {0: clip.objects})
    steps = []
    for t in range(sc['frames']):
        if t in add_at:
            ids = add_at[t]
            m = torch.zeros_like(full)
            for i in ids:
                y0, y1, x0, x1 = clip.rects[i - 1]
                dx = (2 * t) % 16          # rectangles drift with the texture
                m[y0:y1, max(x0 - dx, 0):max(x1 - dx, 1)] = i
            steps.append((clip.frame(t), m, list(ids)))
        else:
            steps.append((clip.frame(t), None, None))
    return steps, sc.get('delete_at', {})


def run_scenario(make_processor, name, device='cpu', record=None, make_cfg=None):
    """make_processor(cfg_overrides) -> processor.  Returns list of per-frame prob tensors (cpu fp32).
    make_cfg(cfg_overrides) -> the config object handed to update_config (defaults to the override dict merged by the
    processor factory's own rules: pass the factory's config builder)."""
    sc = SCENARIOS[name]
    proc = make_processor(sc['cfg'])
    steps, deletes = scenario_inputs(name)
    outs = []

    def planes_of(mask, objs):
        if sc.get('float_kind') == 'onehot':                         # index_numpy_to_one_hot_torch(...)[1:]
            return torch.stack([(mask == o).float() for o in objs]).to(device)
        # soft planes in tmp-id order (0.9 inside, 0.05 outside)
        return torch.stack([(mask == o).float() * 0.85 + 0.05 for o in objs]).to(device)

    with torch.inference_mode():
        for t in sc.get('precommit', ()):
            img, mask, objs = steps[t]
            proc.step(img.to(device), planes_of(mask, objs), idx_mask=False, force_permanent=True)
        for t, (img, mask, objs) in enumerate(steps):
            if t in deletes:
                proc.delete_objects(deletes[t])
            if t in sc.get('update_config_at', {}):
                over = dict(sc['cfg'])
                over.update(sc['update_config_at'][t])
                proc.update_config(make_cfg(over) if make_cfg is not None else over)
            if t in sc.get('clear_non_permanent_at', ()):
                proc.clear_non_permanent_memory()
            img = img.to(device)
            end = t in sc.get('end_at', ())
            if t in sc.get('clear_memory_at', ()):
                proc.clear_memory()
            if t in sc.get('repropagate_at', ()):
                proc.clear_sensory_memory()
                p = proc.step(img.to(device), outs[-1][1:].to(device), idx_mask=False)
                outs.append(p.detach().float().cpu())
                if record is not None:
                    record(t, proc)
                continue
            if mask is not None and t in sc.get('float_mask_at', ()):
                p = proc.step(img, planes_of(mask, objs), objects=objs, idx_mask=False, end=end)   # idx_mask=False input
            elif mask is not None:
                p = proc.step(img, mask.to(device), objects=objs, force_permanent=(t in sc.get('permanent_at', ())), end=end)
            else:
                p = proc.step(img, end=end)
            outs.append(p.detach().float().cpu())
            if record is not None:
                record(t, proc)
    return outs, proc


def summarize(outs, sub):
    """Compact golden record: sub-sampled probs (fp16), argmax md5 and class histograms."""
    rec = {}
    for t, p in enumerate(outs):
        am = p.argmax(0).to(torch.uint8).numpy()
        rec[f'prob_{t}'] = p[:, ::sub, ::sub].numpy().astype(np.float16)
        rec[f'md5_{t}'] = np.frombuffer(hashlib.md5(am.tobytes()).digest(), dtype=np.uint8)
        rec[f'hist_{t}'] = np.bincount(am.ravel(), minlength=p.shape[0]).astype(np.int64)
        # top1-top2 margin map (fp16, sub-sampled) for margin-aware argmax checks
        if p.shape[0] > 1:
            top2 = p.topk(2, dim=0)[0]
            rec[f'margin_{t}'] = (top2[0] - top2[1])[::sub, ::sub].numpy().astype(np.float16)
    return rec
