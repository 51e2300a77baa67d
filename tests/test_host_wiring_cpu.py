"""Host-side parity on CPU: the product's plan builders / memory bank / InferenceCore mirror, executed through the
torch interpreter of the op descriptors (tests/mock_exec.py), against the oracle.

This checks everything EXCEPT the HIP kernels themselves (those are compared with the same interpreter and with
the oracle on the GPU box, tests/test_gpu_*.py).  bf16 activation storage is emulated, so tolerances are the
bf16 ones (prob atol 3e-2, stage rtol 3e-2 of the tensor scale).
"""
import numpy as np
import pytest
import torch

from cutie_amd import _lib
from cutie_amd.config import default_config
from oracle import scenarios as S
from oracle.inference import OracleProcessor, DEFAULT_CFG
from oracle.weights import make_state_dict

from mock_exec import MockExecutor


@pytest.fixture(scope='module')
def product_net():
    from cutie_amd.model.cutie import CUTIE
    _lib.set_executor_for_testing(MockExecutor())
    net = CUTIE(default_config())
    net.load_weights(make_state_dict(seed=0))
    yield net
    _lib.set_executor_for_testing(None)


def rel_err(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-6))


def test_state_dict_keys_match_reference(product_net):
    import json, os
    ref = json.load(open(os.path.join(S.GOLDEN_DIR, 'state_dict_spec.json')))
    sd = product_net.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        assert list(v.shape) == ref[k], k


def test_stages_match_oracle(product_net, oracle_net):
    from cutie_amd.utils.synth import SyntheticClip
    net, onet = product_net, oracle_net
    clip = SyntheticClip(96, 128, 3, 4, seed=5)
    g = torch.Generator().manual_seed(11)
    K, h, w = 3, 6, 8
    with torch.inference_mode():
        img = clip.frame(0).unsqueeze(0)
        ms, pix = net.encode_image(img)
        oms, opix = onet.encode_image(img)
        for a, b, n in zip(ms, oms, ['f16', 'f8', 'f4']):
            assert a.shape == b.shape
            assert rel_err(a, b) < 3e-2, (n, rel_err(a, b))
        assert rel_err(pix, opix) < 3e-2
        key, shr, sel = net.transform_key(ms[0])
        okey, oshr, osel = onet.transform_key(oms[0])
        assert rel_err(key, okey) < 3e-2 and rel_err(shr, oshr) < 3e-2 and rel_err(sel, osel) < 3e-2
        masks = torch.stack([(clip.first_mask() == i + 1).float() for i in range(K)], 0).unsqueeze(0) * 0.9 + 0.05
        sens0 = torch.randn(1, K, 256, h, w, generator=g) * 0.5
        # encode_mask (use the oracle's pix_feat so stage errors do not compound)
        pix_in = opix.to(torch.bfloat16).float()
        sens = sens0.clone()
        val, nsens, summ, _ = net.encode_mask(img, pix_in, sens, masks)
        oval, onsens, osumm = onet.encode_mask(img, opix, sens0, masks)
        assert rel_err(val, oval) < 3e-2, rel_err(val, oval)
        assert rel_err(nsens, onsens) < 3e-2, rel_err(nsens, onsens)
        assert rel_err(summ, osumm) < 3e-2, rel_err(summ, osumm)
        ro = torch.randn(1, K, 256, h, w, generator=g) * 0.5
        fused = net.pixel_fusion(pix_in, ro, sens0.clone(), masks)
        ofused = onet.pixel_fusion(opix, ro, sens0, masks)
        assert rel_err(fused, ofused) < 3e-2, rel_err(fused, ofused)
        rq, aux = net.readout_query(ofused, osumm.unsqueeze(2))
        orq, oaux = onet.readout_query(ofused, osumm.unsqueeze(2), return_aux=True)
        for i in range(4):
            assert rel_err(aux['logits'][i], oaux[i]) < 5e-2, (i, rel_err(aux['logits'][i], oaux[i]))
        assert rel_err(rq, orq) < 5e-2, rel_err(rq, orq)
        sens = sens0.clone()
        s2, lg, prob = net.segment(oms, orq, sens, update_sensory=True)
        os2, olg, oprob = onet.segment(oms, orq, sens0, update_sensory=True)
        assert rel_err(s2, os2) < 3e-2, rel_err(s2, os2)
        assert (prob - oprob).abs().max() < 3e-2, float((prob - oprob).abs().max())
        assert rel_err(lg, olg) < 3e-2


def _run_product(net, name):
    from cutie_amd.inference.inference_core import InferenceCore
    sizes = []

    def make(over):
        cfg = default_config(**over)
        proc = InferenceCore(net, cfg=cfg)
        return proc

    def rec(t, p):
        m = p.memory
        sizes.append([sum(b.n_perm + b.n_work for b in m.buckets.values()), sum(b.n_perm for b in m.buckets.values()),
                      sum(b.n_long for b in m.buckets.values()), len(m.buckets)])

    outs, proc = S.run_scenario(make, name, record=rec, make_cfg=lambda over: default_config(**over))
    return outs, sizes


@pytest.mark.parametrize('name', ['small_fifo', 'small_add_del', 'small_lt', 'small_interactive', 'small_flip', 'small_chunk', 'small_misc', 'small_clear', 'small_video', 'small_lt_overlap', 'small_cfg_fifo', 'small_cfg_lt'])
def test_trajectory_matches_oracle(name, product_net, oracle_net):
    gold = np.load(S.GOLDEN_DIR + f'/{name}.npz')

    def make(over):
        cfg = dict(DEFAULT_CFG)
        cfg.update(over)
        return OracleProcessor(oracle_net, cfg)

    oouts, _ = S.run_scenario(make, name)
    outs, sizes = _run_product(product_net, name)
    assert np.array_equal(np.array(sizes), gold['mem_sizes']), (np.array(sizes).tolist(), gold['mem_sizes'].tolist())
    worst = 0.0
    for t, (p, o) in enumerate(zip(outs, oouts)):
        assert p.shape == o.shape, (t, p.shape, o.shape)
        err = float((p - o).abs().max())
        worst = max(worst, err)
        # bf16 activation storage, fp32 accumulation: within the reference's own reduced-precision envelope (S.trajectory_bounds)
        bmax, bmean, bmargin = S.trajectory_bounds('base')
        assert err < bmax and float((p - o).abs().mean()) < bmean, (name, t, err, float((p - o).abs().mean()))
        # argmax agreement wherever the oracle's top-1/top-2 margin exceeds the tolerance
        top2 = o.topk(2, dim=0)[0]
        confident = (top2[0] - top2[1]) > bmargin       # = 2 x the per-class bound: below it an argmax flip is within tolerance
        assert bool((p.argmax(0) == o.argmax(0))[confident].all()), (name, t)
    print(name, 'worst prob err', worst)


def _run_clip_small(net, seed, frames=4):
    """A short 3-object clip through InferenceCore; returns the stacked probabilities."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.utils.synth import SyntheticClip
    clip = SyntheticClip(64, 96, 2, frames, seed=seed)
    proc = InferenceCore(net, cfg=default_config(mem_every=2))
    out = [proc.step(clip.frame(0), clip.first_mask(), objects=clip.objects)]
    for t in range(1, frames):
        out.append(proc.step(clip.frame(t)))
    return torch.stack([o.float().cpu() for o in out])


def test_fork_and_concurrent_clips_match_sequential(product_net):
    """CUTIE.fork() views (shared weights, private plans) and parallel.run_concurrent (one host thread per clip) give
    exactly the sequential results."""
    from cutie_amd.parallel import run_concurrent
    with torch.inference_mode():
        seq = {c: _run_clip_small(product_net, 20 + c) for c in range(3)}
        fork = product_net.fork()
        assert fork.engine() is not product_net.engine() and fork.engine().w is product_net.engine().w
        assert torch.equal(_run_clip_small(fork, 21), seq[1])
    conc = run_concurrent(product_net, [0, 1, 2], lambda view, c: _run_clip_small(view, 20 + c), streams=3)
    for c in range(3):
        assert torch.equal(conc[c], seq[c]), c


def test_interleaved_clips_match_sequential(product_net):
    """parallel.run_interleaved: ONE host thread issues a step of every clip in flight in turn (generator clips, own fork / stream /
    frame_context table each): the sequential results, with and without look-ahead hints; frame_context.context swaps the thread's
    table and puts it back."""
    from cutie_amd import frame_context as fc
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.parallel import run_interleaved
    from cutie_amd.utils.synth import SyntheticClip

    def gen(view, c, hinted):
        clip = SyntheticClip(48, 80, 2, 9, seed=20 + c)
        frames = [clip.frame(t) for t in range(9)]
        proc = InferenceCore(view, cfg=default_config(mem_every=3))
        outs = []
        for t in range(9):
            kw = dict(next_images=frames[t + 1:t + 7]) if hinted and t + 1 < 9 else {}
            outs.append(proc.step(frames[t], *((clip.first_mask(),) if t == 0 else ()), **(dict(objects=clip.objects) if t == 0 else {}), **kw))
            yield
        return torch.stack(outs)

    def whole(g):
        try:
            while True:
                next(g)
        except StopIteration as e:
            return e.value

    ex = _lib.get_executor()
    ex.per_sample_conv = True          # (torch's CPU conv may sum differently per batch size; the HIP tiles of one K-order class do not)
    try:
        with torch.inference_mode():
            seq = {c: whole(gen(product_net, c, False)) for c in range(4)}
        for hinted in (False, True):
            got = run_interleaved(product_net, list(range(4)), lambda v, c: gen(v, c, hinted), streams=3)
            assert sorted(got) == [0, 1, 2, 3] and not product_net.engine().one_lane
            for c in range(4):
                assert torch.equal(got[c], seq[c]), (hinted, c)
    finally:
        ex.per_sample_conv = False
    a = torch.zeros(3)
    fc.remember('k', a, 1)
    ctx = fc.new_context()
    with fc.context(ctx):
        assert fc.recall('k', a) is None
        fc.remember('k', a, 2)
    assert fc.recall('k', a) == 1
    with fc.context(ctx):
        assert fc.recall('k', a) == 2


def _lockstep_case(net, cfg_kw, C=3, K=2, T=14, hinted=True, size=(48, 80), window=None):
    """(sequential per-clip probabilities, lock-step probabilities, LockstepCores) for C synthetic clips."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.inference.lockstep import LockstepCores
    from cutie_amd.utils.synth import SyntheticClip
    clips = [SyntheticClip(size[0], size[1], K, T, seed=40 + c) for c in range(C)]
    frames = [[cl.frame(t) for t in range(T)] for cl in clips]
    seq = []
    for c, cl in enumerate(clips):
        proc = InferenceCore(net, cfg=default_config(**cfg_kw))
        outs = [proc.step(frames[c][0], cl.first_mask(), objects=cl.objects)]
        for t in range(1, T):
            outs.append(proc.step(frames[c][t], end=(t == T - 1)))
        seq.append((torch.stack(outs), {k: (b.n_long, b.n_perm, b.n_work) for k, b in proc.memory.buckets.items()}))
    ls = LockstepCores(net, default_config(**cfg_kw), C)
    if window is not None:
        ls.WINDOW, ls.WINDOW_LEAD = window
    outs = [ls.step([f[0] for f in frames], [cl.first_mask() for cl in clips], [cl.objects for cl in clips])]
    for t in range(1, T):
        hint = dict(next_images=[f[t + 1:t + 10] for f in frames]) if hinted and t + 1 < T else {}
        outs.append(ls.step([f[t] for f in frames], end=(t == T - 1), **hint))
    got = [(torch.stack([o[c] for o in outs]), {k: (b.n_long, b.n_perm, b.n_work) for k, b in ls.cores[c].memory.buckets.items()}) for c in range(C)]
    return seq, got, ls


@pytest.mark.parametrize('hinted', [False, True])
def test_lockstep_clips_match_sequential(product_net, hinted):
    """inference/lockstep.py: C clips advanced in lock step through ONE plan per stage (batch = C x K objects; the per-clip couplings --
    "others" masks, foreground masks, soft aggregation, per-clip image features -- grouped inside the launches, ABI 4) give every clip the
    probabilities and the bank of its own InferenceCore run, bit for bit: FIFO memory and long-term memory with consolidations, with and
    without look-ahead hints (joint encoder windows of all clips, per-clip stacked read-outs)."""
    ex = _lib.get_executor()
    ex.per_sample_conv = True          # (torch's CPU conv may sum differently per batch size; the HIP tiles of one K-order class do not)
    try:
        with torch.inference_mode():
            for cfg_kw, T, C in ((dict(mem_every=3), 9, 3),
                                 (dict(mem_every=2, use_long_term=True, long_term=dict(S.LT_SMALL)), 15, 2)):      # (two consolidations)
                seq, got, ls = _lockstep_case(product_net, cfg_kw, C=C, T=T, hinted=hinted, window=(4, 1) if hinted else None)
                assert ls.batched_steps == T - 2, ls.batched_steps       # every frame but the first (masks) and the last (end=True)
                assert (ls.joint_passes > 0 and ls.stacked_steps >= T - 4) if hinted else ls.joint_passes == 0, (ls.joint_passes, ls.stacked_steps)
                assert not cfg_kw.get('use_long_term') or all(v[0] > 0 for v in seq[0][1].values()), 'the long-term case must consolidate'
                for c in range(len(seq)):
                    assert seq[c][1] == got[c][1], (cfg_kw, c, seq[c][1], got[c][1])
                    assert torch.equal(seq[c][0], got[c][0]), (cfg_kw, hinted, c, float((seq[c][0] - got[c][0]).abs().max()))
    finally:
        ex.per_sample_conv = False


def test_lockstep_joint_readout_over_the_banks_of_all_clips(product_net):
    """LockstepCores._ahead_joint / MemoryManager.prefetch_affinity_joint (AFF_SCORE flags&4, AFF_READOUT i8): with look-ahead hints and
    query rows per frame that fill whole 128-row blocks, the read-outs of all clips run as one pass per bank version over a frame-major
    encoder window, and pixel fusion takes them as one tensor -- still every clip's own bits and bank, with long-term consolidations."""
    ex = _lib.get_executor()
    ex.per_sample_conv = True
    try:
        with torch.inference_mode():
            cfg_kw = dict(mem_every=2, use_long_term=True, long_term=dict(S.LT_SMALL))
            seq, got, ls = _lockstep_case(product_net, cfg_kw, C=2, T=13, hinted=True, size=(96, 176), window=(4, 1))      # 6 x 11 = 66 -> 128 rows
            assert ls.JOINT and ls.batched_steps == 11
            assert ls.joint_passes >= 6 and ls.stacked_steps >= 9, (ls.joint_passes, ls.stacked_steps)
            assert all(v[0] > 0 for v in seq[0][1].values()), 'the case must consolidate'
            for c in range(len(seq)):
                assert seq[c][1] == got[c][1], (c, seq[c][1], got[c][1])
                assert torch.equal(seq[c][0], got[c][0]), (c, float((seq[c][0] - got[c][0]).abs().max()))
    finally:
        ex.per_sample_conv = False


def test_lockstep_banks_that_do_not_line_up_keep_their_own_readout_lanes(product_net):
    """MemoryManager.prefetch_affinity_joint refuses banks that differ in what the joint launches share (here: top_k of one clip changed
    through update_config): the group stays on the batched plans, every clip's look-ahead read-out runs in its own lane, and every clip
    still gets the bits of its own InferenceCore run under its own setting."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.inference.lockstep import LockstepCores
    from cutie_amd.utils.synth import SyntheticClip
    C, T = 2, 9
    clips = [SyntheticClip(48, 80, 2, T, seed=90 + c) for c in range(C)]
    frames = [[cl.frame(t) for t in range(T)] for cl in clips]
    cfgs = [default_config(mem_every=3), default_config(mem_every=3, top_k=12)]
    ex = _lib.get_executor()
    ex.per_sample_conv = True
    try:
        with torch.inference_mode():
            seq = []
            for c, cl in enumerate(clips):
                proc = InferenceCore(product_net, cfg=cfgs[c])
                seq.append(torch.stack([proc.step(frames[c][0], cl.first_mask(), objects=cl.objects)] + [proc.step(frames[c][t]) for t in range(1, T)]))
            ls = LockstepCores(product_net, cfgs[0], C)
            ls[1].update_config(cfgs[1])
            outs = [ls.step([f[0] for f in frames], [cl.first_mask() for cl in clips], [cl.objects for cl in clips])]
            for t in range(1, T):
                outs.append(ls.step([f[t] for f in frames], **(dict(next_images=[f[t + 1:t + 8] for f in frames]) if t + 1 < T else {})))
            assert ls.batched_steps == T - 1 and ls.joint_passes == 0 and ls.stacked_steps == 0, (ls.batched_steps, ls.joint_passes, ls.stacked_steps)
            for c in range(C):
                got = torch.stack([o[c] for o in outs])
                assert torch.equal(got, seq[c]), (c, float((got - seq[c]).abs().max()))
            assert not torch.equal(seq[1], torch.stack([o[0] for o in outs]))
    finally:
        ex.per_sample_conv = False


def test_lockstep_leaves_the_batched_path_when_the_clips_stop_being_uniform(product_net):
    """LockstepCores: a frame that brings masks, and everything behind it once the clips hold a second bucket (objects added mid-clip,
    kv_memory_store.py:96-117), run clip by clip through the cores' own step -- still every clip's own results; deleting the second
    bucket's object brings the group back onto the batched path."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.inference.lockstep import LockstepCores
    from cutie_amd.utils.synth import SyntheticClip
    C, T = 2, 12
    clips = [SyntheticClip(48, 80, 3, T, seed=80 + c) for c in range(C)]
    frames = [[cl.frame(t) for t in range(T)] for cl in clips]
    first = [(cl.first_mask() * (cl.first_mask() != 3).long()) for cl in clips]          # objects 1, 2 at t = 0
    late = [(cl.first_mask() * (cl.first_mask() == 3).long()) for cl in clips]           # object 3 at t = 4 (a second bucket)
    cfg_kw = dict(mem_every=2)

    def script(step, delete):
        outs = [step(0, first, [[1, 2]] * C)]
        for t in range(1, T):
            if t == 4:
                outs.append(step(t, late, [[3]] * C))
            else:
                if t == 8:
                    delete([3])
                outs.append(step(t, None, None))
        return outs
    ex = _lib.get_executor()
    ex.per_sample_conv = True
    try:
        with torch.inference_mode():
            seq = []
            for c in range(C):
                proc = InferenceCore(product_net, cfg=default_config(**cfg_kw))
                seq.append(script(lambda t, m, o: proc.step(frames[c][t], *((m[c],) if m is not None else ()), **(dict(objects=o[c]) if o is not None else {})),
                                  proc.delete_objects))
            ls = LockstepCores(product_net, default_config(**cfg_kw), C)
            counts = []

            def ls_step(t, m, o):
                r = ls.step([f[t] for f in frames], m, o, **({} if (m is not None or t + 1 >= T) else dict(next_images=[f[t + 1:t + 6] for f in frames])))
                counts.append(ls.batched_steps)
                return r
            got = script(ls_step, lambda ids: ls.delete_objects([ids] * C))
        # batched: t = 1..3; clip by clip: t = 0, 4 (masks), 5..7 (two buckets) and 8 (object 3 deleted, but the last mask still has its plane: the
        # cores index it by tmp id, memory_manager.py:90-92); batched again from t = 9
        assert counts == [0, 1, 2, 3, 3, 3, 3, 3, 3, 4, 5, 6], counts
        for c in range(C):
            for t in range(T):
                assert torch.equal(got[t][c], seq[c][t]), (c, t, float((got[t][c] - seq[c][t]).abs().max()))
    finally:
        ex.per_sample_conv = False


def test_run_batched_groups_and_uneven_clips(product_net):
    """parallel.run_batched: clips in groups of `lockstep`, a last group of one clip, a clip longer than its group's shortest -- every clip
    gets the object-id masks of its own InferenceCore run."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.parallel import run_batched
    from cutie_amd.utils.synth import SyntheticClip
    lens = [7, 9, 6]
    clips = []
    for c, T in enumerate(lens):
        cl = SyntheticClip(48, 80, 2, T, seed=70 + c)
        clips.append(dict(frames=[cl.frame(t) for t in range(T)], mask=cl.first_mask(), objects=cl.objects))
    cfg_kw = dict(mem_every=2)
    ex = _lib.get_executor()
    ex.per_sample_conv = True
    try:
        got = run_batched(product_net, default_config(**cfg_kw), clips, lockstep=2, lookahead=5)
        got2 = run_batched(product_net, default_config(**cfg_kw), clips, lockstep=2, in_flight=2, lookahead=5)       # the two groups in flight next to each other
        assert not product_net.engine().one_lane
        for c in range(len(clips)):
            assert len(got2[c]) == lens[c] and all(torch.equal(a, b) for a, b in zip(got[c], got2[c])), c
        with torch.inference_mode():
            for c, cl in enumerate(clips):
                proc = InferenceCore(product_net, cfg=default_config(**cfg_kw))
                want = [proc.output_prob_to_mask(proc.step(cl['frames'][0], cl['mask'], objects=cl['objects']), dtype=torch.uint8)]
                for t in range(1, lens[c]):
                    want.append(proc.output_prob_to_mask(proc.step(cl['frames'][t], end=(t == lens[c] - 1)), dtype=torch.uint8))
                assert len(got[c]) == lens[c]
                for t in range(lens[c]):
                    assert torch.equal(got[c][t], want[t]), (c, t)
    finally:
        ex.per_sample_conv = False


def test_max_internal_size_path(product_net):
    """The internal-resolution path (inference_core.py:206-228, 321-326): frames larger than max_internal_size are processed
    at the reduced size and the probabilities are resized back; index masks use nearest-exact."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.utils.synth import SyntheticClip
    clip = SyntheticClip(96, 144, 2, 3, seed=4)
    with torch.inference_mode():
        proc = InferenceCore(product_net, cfg=default_config(mem_every=2))
        proc.max_internal_size = 64
        p0 = proc.step(clip.frame(0), clip.first_mask(), objects=clip.objects)
        assert p0.shape == (3, 96, 144) and proc.pad == (0, 0, 0, 0)              # 64 x 96 inside
        p1 = proc.step(clip.frame(1))
        assert p1.shape == (3, 96, 144) and torch.isfinite(p1).all()
        assert float((p1.sum(0) - 1).abs().max()) < 1e-4
        # same thing spelled with torch around a core that runs at the reduced size directly
        small = torch.nn.functional.interpolate(clip.frame(0)[None], size=(64, 96), mode='bilinear', align_corners=False)[0]
        msmall = torch.nn.functional.interpolate(clip.first_mask()[None, None].float(), size=(64, 96), mode='nearest-exact')[0, 0].round().long()
        ref = InferenceCore(product_net, cfg=default_config(mem_every=2))
        r0 = ref.step(small, msmall, objects=clip.objects)
        up = torch.nn.functional.interpolate(r0[None], size=(96, 144), mode='bilinear', align_corners=False)[0]
        assert torch.allclose(p0, up, atol=1e-5)


def test_caller_visible_attributes(product_net):
    """the instance attributes reference callers read or write (scripting_demo.py:21, eval_vos.py:101,
    gui/main_controller.py:107-110,494-516) exist with the reference's meaning."""
    from cutie_amd.inference.image_feature_store import ImageFeatureStore
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.utils.synth import SyntheticClip
    cfg = default_config(mem_every=2, use_long_term=True, long_term=dict(count_usage=True, max_mem_frames=4, min_mem_frames=2,
                                                                         num_prototypes=8, max_num_tokens=40, buffer_tokens=12))
    store = ImageFeatureStore(product_net, no_warning=True)
    proc = InferenceCore(product_net, cfg=cfg, image_feature_store=store)
    assert proc.image_feature_store is store and proc.curr_ti == -1 and proc.last_mask is None
    proc.max_internal_size = -1                                   # written by scripting_demo.py:21
    clip = SyntheticClip(64, 90, 2, 4, seed=2)
    with torch.inference_mode():
        proc.step(clip.frame(0), clip.first_mask(), objects=[3, 7])
        proc.step(clip.frame(1))
        proc.step(clip.frame(2), delete_buffer=False)
    HW = 4 * 6
    m = proc.memory
    assert proc.curr_ti == 2 and proc.mem_every == 2 and tuple(proc.pad) == (3, 3, 0, 0)
    assert (m.max_mem_frames, m.min_mem_frames, m.max_long_tokens, m.num_prototypes) == (3, 1, 40, 8)
    assert m.max_work_tokens == 3 * HW and m.min_work_tokens == 1 * HW
    assert m.work_mem.perm_size(0) == HW and m.work_mem.non_perm_size(0) == HW and m.long_mem.non_perm_size(0) == 0
    assert m.work_mem.size(0) == 2 * HW and m.work_mem.engaged() and not m.long_mem.engaged() and m.work_mem.num_objects == 2
    om = proc.object_manager
    assert om.all_obj_ids == [3, 7] and om.num_obj == 2 and [o.id for o in om.obj_to_tmp_id] == [3, 7]
    assert {t: o.id for t, o in om.tmp_id_to_obj.items()} == {1: 3, 2: 7} and om.find_tmp_by_id(7) == 2
    assert proc.last_mask.shape == (1, 2, 64, 96)
    assert len(store) == 1                                        # delete_buffer=False kept frame 2's features
    store.delete(2)
    assert len(store) == 0


def test_small_model_variant():
    """cutie-small (model/small.yaml: ResNet-18 pixel encoder, ms_dims [256,128,64]): the reference's 359-key state_dict, encoder
    stages and a whole trajectory against the oracle (itself pinned to the executed reference, tests/golden/model_small.npz)."""
    import json, os
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.model.cutie import CUTIE
    from cutie_amd.utils.synth import SyntheticClip
    from oracle.net import OracleNet
    from oracle.weights import MODEL_CFG_SMALL
    sd = make_state_dict(seed=0, m=MODEL_CFG_SMALL)
    onet = OracleNet(sd, MODEL_CFG_SMALL)
    prev = _lib._executor
    _lib.set_executor_for_testing(MockExecutor())
    try:
        net = CUTIE(default_config(model='small'))
        ref = json.load(open(os.path.join(S.GOLDEN_DIR, 'state_dict_spec_small.json')))
        assert list(net.state_dict().keys()) == list(ref.keys())
        net.load_weights(sd)
        with torch.inference_mode():
            img = SyntheticClip(128, 192, 3, 4, seed=5).frame(0).unsqueeze(0)
            ms, pix = net.encode_image(img)
            key, shr, sel = net.transform_key(ms[0])
            oms, opix = onet.encode_image(img)
            okey, oshr, osel = onet.transform_key(oms[0])
            assert [t.shape[1] for t in ms] == [256, 128, 64]
            for n, a, b in zip(['f16', 'f8', 'f4', 'pix', 'key', 'shr', 'sel'], [*ms, pix, key, shr, sel], [*oms, opix, okey, oshr, osel]):
                assert a.shape == b.shape and rel_err(a, b) < 3e-2, (n, rel_err(a, b))
        cfgs = lambda over: default_config(model='small', **over)
        outs, _ = S.run_scenario(lambda over: InferenceCore(net, cfg=cfgs(over)), 'small_fifo', make_cfg=cfgs)
        oouts, _ = S.run_scenario(lambda over: OracleProcessor(onet, dict(DEFAULT_CFG, **over)), 'small_fifo')
        worst = 0.0
        for t, (p, o) in enumerate(zip(outs, oouts)):
            err = float((p - o).abs().max())
            bmax, bmean, _ = S.trajectory_bounds('small')
            assert err < bmax and float((p - o).abs().mean()) < bmean, (t, err)
            worst = max(worst, err)
        print('cutie-small worst prob err', worst)
    finally:
        _lib.set_executor_for_testing(prev)


@pytest.mark.parametrize('name', ['small_lt', 'small_interactive'])
def test_bank_contents_match_oracle(name, product_net, oracle_net):
    """SURVEY 8c(iii), beyond token counts: at the end of a trajectory the product's bank holds the same SET of tokens as the
    oracle's stores -- per region (long-term / permanent / working) and per object the value rows, and in long-term mode the raw
    keys with their usage / life counters.  The physical order differs by design (FIFO ring, in-place compaction), so rows are
    matched through a fixed random projection."""
    from cutie_amd.inference.inference_core import InferenceCore
    procs = {}
    S.run_scenario(lambda over: procs.setdefault('o', OracleProcessor(oracle_net, dict(DEFAULT_CFG, **over))), name)
    S.run_scenario(lambda over: procs.setdefault('p', InferenceCore(product_net, cfg=default_config(**over))), name,
                   make_cfg=lambda over: default_config(**over))
    o, p = procs['o'], procs['p']
    g = torch.Generator().manual_seed(5)
    rv, rk = torch.randn(256, generator=g), torch.randn(64, generator=g)

    def match(a, b, tol, what):
        assert a.shape == b.shape, (what, a.shape, b.shape)
        sa, sb = torch.sort(a)[0], torch.sort(b)[0]
        assert float((sa - sb).abs().max()) <= tol * max(1.0, float(sb.abs().max())), (what, float((sa - sb).abs().max()))

    assert sorted(o.work.buckets) == sorted(p.memory.buckets)
    for bid, bucket in p.memory.buckets.items():
        assert bucket.objects == o.work.buckets[bid]
        perm = o.work.perm_end[bid]
        n_long = o.long.size(bid) if (o.use_long_term and bid in o.long.buckets) else 0
        assert (bucket.n_long, bucket.n_perm, bucket.n_work) == (n_long, perm, o.work.size(bid) - perm)
        regions = {'perm': (bucket.perm_start, bucket.n_perm), 'work': (bucket.work_start, bucket.n_work), 'long': (0, bucket.n_long)}
        for obj in bucket.objects:
            ov = {'perm': o.work.v[obj][:, :perm], 'work': o.work.v[obj][:, perm:]}
            if n_long:
                ov['long'] = o.long.v[obj]
            for reg, t in ov.items():
                a0, n = regions[reg]
                match(bucket.values[obj][a0:a0 + n].float() @ rv, rv @ t, 3e-2, (bid, obj, reg, 'value'))
        if o.use_long_term:
            a0, n = regions['work']
            ok = o.work.k[bid][:, perm:]
            pk = bucket.rawkey[a0:a0 + n]
            match(pk @ rk, rk @ ok, 3e-2, (bid, 'work key'))
            # counters travel with their token: pair every product token with its nearest oracle key (must be a bijection)
            nn = torch.cdist(pk, ok.t().contiguous()).argmin(1)
            assert sorted(nn.tolist()) == list(range(n)), (bid, 'keys do not pair one-to-one')
            life_o, life_p = o.work.life[bid][nn], bucket.life[a0:a0 + n]
            use_o, use_p = o.work.use[bid][nn], bucket.use[a0:a0 + n]
            assert float((life_o - life_p).abs().max()) < 1e-3, (bid, 'life')
            assert float((use_o - use_p).abs().max()) < 0.25 and float((use_o - use_p).abs().mean()) < 0.02, \
                (bid, 'use', float((use_o - use_p).abs().max()))
            if n_long:
                # prototypes are the top-usage candidates (memory_manager.py:338-341): near-ties in usage may pick different
                # tokens under bf16, so the long-term sets agree for most, not all, prototypes; a shared prototype is the
                # same key (it is a copy of a working-memory key)
                lk, olk = bucket.rawkey[:n_long], o.long.k[bid].t().contiguous()
                d = torch.cdist(lk, olk).min(1)[0] / olk.norm(dim=1).mean()
                shared = float((d < 3e-2).float().mean())
                print(name, 'bucket', bid, 'shared long-term prototypes', shared)
                assert shared > 0.6, (bid, 'long-term prototypes', shared)


def test_rewrapped_tensors_take_the_slow_path(product_net):
    """The companions of a facade result (similarity operands of a key, bf16 shadow of the sensory state) are found by storage
    address (cutie_amd/frame_context.py); a caller that clones / rebuilds the tensors in between gets them recomputed instead of
    an error -- same values."""
    from cutie_amd import frame_context
    from cutie_amd.utils.synth import SyntheticClip
    net = product_net
    with torch.inference_mode():
        img = SyntheticClip(64, 96, 2, 2, seed=9).frame(0).unsqueeze(0)
        ms, pix = net.encode_image(img)
        key, shr, sel = net.transform_key(ms[0])
        fast = net.query_operands(key, sel)
        assert fast is frame_context.recall('query', key) and fast is net.query_operands(key[:, :, :, :], sel)      # any view finds it
        slow = net.query_operands(key.clone(), sel.clone())
        assert slow is not fast
        for k in ('Bhi', 'Blo'):
            assert torch.equal(slow[k].view(torch.int16), fast[k].view(torch.int16)), k
        assert torch.allclose(slow['cq'], fast['cq'], rtol=1e-6, atol=1e-6)
        # sensory state brought by the caller (no bf16 shadow registered): segment casts once and returns the same result as with
        # the state it produced itself
        K, h, w = 2, 4, 6
        g = torch.Generator().manual_seed(3)
        ro = torch.randn(1, K, 256, h, w, generator=g) * 0.3
        sens = torch.randn(1, K, 256, h, w, generator=g) * 0.3
        s1, _, p1 = net.segment(ms, ro, sens.clone(), update_sensory=True)
        s2, _, p2 = net.segment(ms, ro, sens.clone().contiguous(), update_sensory=True)
        assert torch.equal(p1, p2) and torch.equal(s1, s2)


def test_save_aux_contents(product_net, oracle_net):
    """cfg.save_aux: MemoryManager.aux carries the reference's keys (memory_manager.py:197-206; the reference itself raises there in
    eval mode -- edge case save_aux_on_read).  The attention mask must be object_transformer.py:179-205 applied to the last block's
    logits (checked against the oracle's restatement, head 0), the logits are copies, and the frame's result is unchanged."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.utils.synth import SyntheticClip
    clip = SyntheticClip(96, 128, 3, 3, seed=21)
    with torch.inference_mode():
        outs = {}
        for aux_on in (False, True):
            p = InferenceCore(product_net, cfg=default_config(save_aux=aux_on))
            p.step(clip.frame(0), clip.first_mask(), objects=clip.objects)
            outs[aux_on] = p.step(clip.frame(1)).clone()
        assert torch.equal(outs[False], outs[True])                       # the extra mask_pred head does not feed back
        aux = p.memory.aux
        assert sorted(aux) == ['attn_mask', 'p_weights', 'pixel_readout', 'q_logits', 'q_weights', 'sensory']
        assert len(aux['q_logits']) == 4 and aux['q_weights'] is None and aux['p_weights'] is None
        K, h, w = 3, 6, 8
        assert aux['attn_mask'].shape == (1, K, 16, h, w) and aux['q_logits'][-1].shape == (1, K, h, w)
        want = oracle_net._aux_mask(aux['q_logits'][-1]).view(K, oracle_net.m['num_heads'], 16, h, w)[:, 0].float()
        assert torch.equal(aux['attn_mask'][0], want)
        before = [t.clone() for t in aux['q_logits']]
        p.step(clip.frame(2))                                              # the next frame rewrites the plan's buffers, not the copies
        assert all(torch.equal(a, b) for a, b in zip(before, aux['q_logits'])) or aux is not p.memory.aux


def test_frame_context_is_per_thread_and_capped():
    """cutie_amd/frame_context.py: entries are found by storage address through any view, live per host thread (a clip is driven by one
    thread: another clip's traffic cannot evict them), and large payload kinds keep only their last `cap` entries."""
    import threading
    from cutie_amd import frame_context as fc
    a = torch.zeros(8, 4)
    fc.remember('kind', a, 'payload')
    assert fc.recall('kind', a[:, :]) == 'payload' and fc.recall('other', a) is None
    seen = []
    t = threading.Thread(target=lambda: seen.append(fc.recall('kind', a)))
    t.start(); t.join()
    assert seen == [None]                                             # another thread: not found -> that caller's slow path
    keep = [torch.zeros(4) for _ in range(6)]
    for i, x in enumerate(keep):
        fc.remember('big', x, i, cap=4)
    assert [fc.recall('big', x) for x in keep] == [None, None, 2, 3, 4, 5]
    for x in [torch.zeros(2) for _ in range(200)]:
        fc.remember('flood', x, 0)
    assert fc.recall('kind', a) is None                               # small LRU: old entries age out
    fc.forget('big', keep[-1])
    assert fc.recall('big', keep[-1]) is None


def test_slot_pool_never_recycles_what_is_still_referenced():
    """plans.SlotPool: frame f is served from slot f % SLOTS; a slot whose tensors (or views of them) are still referenced is not handed
    out again -- the call gets fresh tensors instead; the look-ahead offset addresses the next frame's slot."""
    from cutie_amd.model import plans as PL
    pool = PL.SlotPool()
    spec = {'a': ((4, 8), torch.float32, False), 'z': ((16,), torch.float32, True)}
    first = pool.get('g', spec, 'cpu')
    first['z'].fill_(3.0)
    base = first['a'].data_ptr()
    keep = first['a'][1:]                                   # a view keeps the slot busy
    del first
    again = pool.get('g', spec, 'cpu')                      # same frame, slot busy -> fresh tensors, zero where asked
    assert again['a'].data_ptr() != base and float(again['z'].abs().sum()) == 0.0
    del keep, again
    assert pool.get('g', spec, 'cpu')['a'].data_ptr() == base            # released: the slot comes back (contents are the caller's business)
    ptrs = []
    for f in range(2 * PL.SlotPool.SLOTS):
        pool.tick()
        ptrs.append(pool.get('g', spec, 'cpu')['a'].data_ptr())
    assert len(set(ptrs)) == PL.SlotPool.SLOTS and ptrs[:PL.SlotPool.SLOTS] == ptrs[PL.SlotPool.SLOTS:]
    pool.offset = 1
    ahead = pool.get('g', spec, 'cpu')['a'].data_ptr()
    pool.offset = 0
    pool.tick()
    assert pool.get('g', spec, 'cpu')['a'].data_ptr() == ahead


def test_next_weights_ranges_are_wired_to_producer_consumer_convs(monkeypatch):
    """OpList.finalize hands every conv on a producer / consumer tile the packed weights of the next conv of the list (p9 / i22) and,
    when that one has no producer waves of its own, those of the conv after it (p10 / i23); nothing else changes, twice is once."""
    from cutie_amd import ops as O
    from cutie_amd.model.weights import pack_conv
    g = torch.Generator().manual_seed(0)
    C = 64
    ws = [pack_conv(torch.randn(C, C, 3, 3, generator=g), torch.randn(C, generator=g), 'cpu', segs=[(C, C)]) for _ in range(4)]
    x = [torch.zeros((1, 16, 16, C), dtype=torch.bfloat16) for _ in range(5)]
    kw = dict(B=1, H=16, W=16, C1=C, ldx1=C, OH=16, OW=16, ldy=C, pad=1)

    def build():
        ol = O.OpList()
        for n, tile in enumerate((100, 61, 100, 100)):           # 61: an LDS-DMA tile without producer waves
            ol.conv(x[n], ws[n], x[n + 1], tile=tile, **kw)
        return ol
    monkeypatch.setattr(O, 'WEIGHT_PREFETCH', 1 << 20)
    monkeypatch.setattr(O, 'WEIGHT_PREFETCH_BLOCK', 1 << 20)
    ol = build()
    a = ol.finalize().copy()
    b = ol.finalize()
    assert a.tobytes() == b.tobytes()
    nbytes = [w.weight.numel() * 2 for w in ws]
    wp = [w.weight.data_ptr() for w in ws]
    assert (a['p'][0, 9], a['i'][0, 22], a['p'][0, 10], a['i'][0, 23]) == (wp[1], nbytes[1], wp[2], nbytes[2])
    assert (a['p'][1, 9], a['p'][1, 10]) == (0, 0)
    assert (a['p'][2, 9], a['i'][2, 22], a['p'][2, 10]) == (wp[3], nbytes[3], 0)
    assert (a['p'][3, 9], a['p'][3, 10]) == (0, 0)
    monkeypatch.setattr(O, 'WEIGHT_PREFETCH', 4096)
    assert build().finalize()['i'][0, 22] == 4096
    # 4 blocks = one per XCD at most: what is left of the per-block budget after the first range goes to the second
    monkeypatch.setattr(O, 'WEIGHT_PREFETCH', 1 << 20)
    monkeypatch.setattr(O, 'WEIGHT_PREFETCH_BLOCK', nbytes[1] + 3072 + 100)
    c = build().finalize()
    assert (c['i'][0, 22], c['i'][0, 23]) == (nbytes[1], 3172) and c['i'][2, 22] == nbytes[3]
    monkeypatch.setattr(O, 'WEIGHT_PREFETCH_BLOCK', 8192)
    c = build().finalize()
    assert (c['i'][0, 22], c['p'][0, 10], c['i'][0, 23]) == (8192, 0, 0)
    monkeypatch.setattr(O, 'WEIGHT_PREFETCH', 0)
    c = build().finalize()
    assert not c['p'][:, 9:].any() and not c['i'][:, 22:].any()
    c['p'][:, 9:11] = a['p'][:, 9:11]
    c['i'][:, 22:24] = a['i'][:, 22:24]
    assert a.tobytes() == c.tobytes()


def test_touch_is_planned_only_where_weights_go_cold(product_net):
    """plans.weights_go_cold: the size classes measured on the MI355X (its docstring), and the flag reaches the plans' op lists."""
    from cutie_amd.model import plans
    assert plans.weights_go_cold(30 * 54, 3) and plans.weights_go_cold(30 * 54) and plans.weights_go_cold(45 * 80, 2)
    assert not plans.weights_go_cold(30 * 54, 1) and not plans.weights_go_cold(23 * 40, 2) and not plans.weights_go_cold(15 * 27)
    assert not plans.weights_go_cold(68 * 120, 5) and not plans.weights_go_cold(68 * 120)
    eng = product_net.engine()
    small = plans.build_pixel_fusion(eng, 2, 4, 6)
    big = plans.build_pixel_fusion(eng, 3, 30, 54)
    assert not small.ol.touch_next_weights and big.ol.touch_next_weights
    assert not small.ol.finalize()['p'][:, 9:11].any()


def test_touch_ranges_follow_a_later_tile_change(monkeypatch):
    """Plans apply the tile table AFTER the list is finalized (Plan.autotune_convs); wire_next_weights then recomputes every range from
    the tiles in place and clears the ones that no longer apply."""
    from cutie_amd import ops as O
    from cutie_amd.model.weights import pack_conv
    g = torch.Generator().manual_seed(0)
    C = 64
    ws = [pack_conv(torch.randn(C, C, 3, 3, generator=g), torch.randn(C, generator=g), 'cpu', segs=[(C, C)]) for _ in range(4)]
    x = [torch.zeros((1, 16, 16, C), dtype=torch.bfloat16) for _ in range(5)]
    kw = dict(B=1, H=16, W=16, C1=C, ldx1=C, OH=16, OW=16, ldy=C, pad=1)
    monkeypatch.setattr(O, 'WEIGHT_PREFETCH', 1 << 20)
    monkeypatch.setattr(O, 'WEIGHT_PREFETCH_BLOCK', 1 << 20)
    ol = O.OpList()
    for n, tile in enumerate((100, 61, 100, 100)):
        ol.conv(x[n], ws[n], x[n + 1], tile=tile, **kw)
    arr = ol.finalize()
    wp = [w.weight.data_ptr() for w in ws]
    nbytes = [w.weight.numel() * 2 for w in ws]
    assert (arr['p'][0, 9], arr['p'][0, 10], arr['p'][1, 9], arr['p'][2, 9]) == (wp[1], wp[2], 0, wp[3])
    arr['i'][:, 17] = (61, 100, 61, 100)                  # what a tile table might do
    ol.wire_next_weights()
    assert (arr['p'][0, 9], arr['p'][0, 10], arr['i'][0, 22], arr['i'][0, 23]) == (0, 0, 0, 0)
    assert (arr['p'][1, 9], arr['i'][1, 22], arr['p'][1, 10], arr['i'][1, 23]) == (wp[2], nbytes[2], wp[3], nbytes[3])
    assert (arr['p'][2, 9], arr['p'][2, 10], arr['p'][3, 9], arr['p'][3, 10]) == (0, 0, 0, 0)
    ol.touch_next_weights = False
    ol.wire_next_weights()
    assert not arr['p'][:, 9:11].any() and not arr['i'][:, 22:24].any()


def test_window_encoder_plan_matches_single_frame_plan(product_net):
    """CUTIE._encode_window (B frames through one plan: the look-ahead window of InferenceCore.prefetch_window) against B runs of the
    one-frame plan, through the descriptor interpreter: every output of every frame, including the zero padding rows of the
    similarity operands, and the frame_context companions registered by _adopt_encoded."""
    from cutie_amd import frame_context
    from cutie_amd.utils.synth import SyntheticClip
    net = product_net
    clip = SyntheticClip(90, 120, 2, 4, seed=3)                 # un-padded 90 x 120 -> 96 x 128
    imgs = [clip.frame(t).float().contiguous() for t in range(3)]
    geom = (90, 120, 96, 128, 4, 3)
    with torch.inference_mode():
        recs = net._encode_window(imgs, *geom)
        assert len(recs) == 3
        for b in range(3):
            one = net._encode(imgs[b], *geom)
            for k in ('f16', 'f8', 'f4', 'pix_feat', 'key', 'shr', 'sel', 'B', 'cq', 'f8p', 'f4p', 'fuse_xt'):
                # (the interpreter's batched fp32 convs may round differently from its one-frame ones; the split-bf16 similarity
                # operand is compared as hi + lo: the low half alone is the rounding residue of the high one)
                a, c = (recs[b][k].float(), one[k].float()) if k != 'B' else (recs[b]['Bhi'].float() + recs[b]['Blo'].float(),
                                                                               one['Bhi'].float() + one['Blo'].float())
                assert a.shape == c.shape, (k, a.shape, c.shape)
                assert float((a - c).abs().max()) <= 2e-2 * float(c.abs().max()) + 1e-6, (b, k, float((a - c).abs().max()))
            hw = recs[b]['h'] * recs[b]['w']
            assert float(recs[b]['Bhi'][hw:].abs().max()) == 0 and float(recs[b]['cq'][hw:].abs().max()) == 0
            ms, pix, key, shr, sel = net._adopt_encoded(recs[b])
            assert key.shape == (1, 64, 6, 8) and shr.shape == (1, 1, 6, 8) and ms[2].shape == (1, 256, 24, 32)
            assert frame_context.recall('query', key) is not None and frame_context.recall('fuse_xt', pix) is not None
        one1 = net._encode_window(imgs[:1], *geom)              # a window of one frame: the B = 1 plan under its own key / arena
        assert float((one1[0]['key'] - recs[0]['key']).abs().max()) <= 2e-2 * float(recs[0]['key'].abs().max())


def test_lookahead_window_bookkeeping_matches_plain_order():
    """InferenceCore.prefetch_window through the descriptor interpreter (no streams; convs per sample, so that the interpreter's fp32
    convs round the same whatever the batch): complete window hints, hints shorter than the window, a schedule that changes under
    way, window and next_image hints alternating, long-term memory with the look-ahead read-out -- all bit-identical to no hints,
    and no encoded frame is left behind."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.model.cutie import CUTIE
    from cutie_amd.utils.synth import SyntheticClip
    mx = MockExecutor()
    mx.per_sample_conv = True
    prev = _lib._executor                                       # (the module fixture's interpreter: put back afterwards)
    _lib.set_executor_for_testing(mx)
    try:
        net = CUTIE(default_config())
        net.load_weights(make_state_dict(seed=0))
        n = 13
        clip = SyntheticClip(64, 96, 2, n, seed=3)
        frames = [clip.frame(t) for t in range(n)]
        decoy = [clip.frame((5 * t + 2) % n) + 0.01 for t in range(n)]
        mask = clip.first_mask()

        def run(hint, **cfg):
            proc = InferenceCore(net, cfg=default_config(mem_every=3, **cfg))
            outs = []
            for t in range(n):
                outs.append(proc.step(frames[t], *((mask,) if t == 0 else ()), **(dict(objects=clip.objects) if t == 0 else {}), **hint(t)))
            return torch.stack(outs), proc

        full = lambda t: dict(next_images=frames[t + 1:t + 9])
        short = lambda t: dict(next_images=frames[t + 1:t + 3])
        changing = lambda t: dict(next_images=[frames[t + 1]] + decoy[t + 2:t + 6]) if (t + 1 < n and t % 4 == 1) else full(t)
        mixed = lambda t: {} if t % 5 == 0 else (dict(next_image=frames[t + 1]) if (t % 5 == 1 and t + 1 < n) else full(t))
        # one read-out per bank version (MemoryManager._affinity_batch): how many frames did each batched read-out cover?
        from cutie_amd.inference.memory_manager import MemoryManager
        batches = []
        orig_batch = MemoryManager._affinity_batch

        def counting(self, bucket, q, h, w, dev, frames):
            batches.append(frames)
            return orig_batch(self, bucket, q, h, w, dev, frames)
        MemoryManager._affinity_batch = counting
        with torch.inference_mode():
            plain, _ = run(lambda t: {})
            assert not batches                                  # (no hints: every frame reads on its own)
            for name, h in (('full', full), ('short', short), ('changing', changing), ('mixed', mixed)):
                del batches[:]
                got, proc = run(h)
                assert torch.equal(got, plain), (name, float((got - plain).abs().max()))
                assert len(proc._window) == 0, name
                if name == 'full':                              # mem_every = 3: a memory frame's successor alone, the two frames behind it in one pass
                    assert batches and max(batches) == 2 and sum(batches) >= 6, batches
            from cutie_amd.inference import inference_core as IC
            IC.AFF_FIRST_ALONE = False                          # the whole memory cycle in one pass over the bank
            try:
                del batches[:]
                got, proc = run(full)
                assert torch.equal(got, plain) and len(proc._window) == 0
                assert batches and max(batches) == 3 and sum(batches) >= 8, batches
            finally:
                IC.AFF_FIRST_ALONE = True
            plain, p0 = run(lambda t: {}, use_long_term=True, long_term=S.LT_SMALL)
            del batches[:]
            got, p1 = run(full, use_long_term=True, long_term=S.LT_SMALL)
            assert batches and max(batches) == 2, batches
            assert torch.equal(got, plain)
            b0, b1 = next(iter(p0.memory.buckets.values())), next(iter(p1.memory.buckets.values()))
            assert (b0.n_long, b0.n_work, b0.n_perm) == (b1.n_long, b1.n_work, b1.n_perm)
            # (usage of a look-ahead read-out is parked in a side buffer and added as one number: a different fp32 rounding)
            assert torch.allclose(b0.use[:b0.work_start + b0.n_work], b1.use[:b1.work_start + b1.n_work], rtol=1e-5, atol=1e-6)
            assert torch.equal(b0.life[:b0.work_start + b0.n_work], b1.life[:b1.work_start + b1.n_work])
    finally:
        MemoryManager._affinity_batch = orig_batch
        _lib.set_executor_for_testing(prev)


def test_patched_affinity_plans_equal_rebuilt_ones():
    """MemoryManager keeps the affinity / commit plans of a bucket across memory frames and patches the sizes that changed (token
    ranges, tile count, counted lengths) into their descriptors.  Every descriptor array handed to the executor -- kinds, flags, ints,
    floats; pointers left out, they depend on the allocator -- must equal what a run that REBUILDS the plans hands over, frame by frame:
    long-term memory (growing ranges, a consolidation, counted long-term usage on and off), FIFO memory, with and without hints."""
    from cutie_amd.inference import memory_manager as MM
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.model.cutie import CUTIE
    from cutie_amd.utils.synth import SyntheticClip

    class Recording(MockExecutor):
        def __init__(self):
            self.seen = []

        def run(self, arr):
            self.seen.append([(int(r['kind']), int(r['flags']), r['i'].tolist(), r['f'].tolist()) for r in arr])
            super().run(arr)

    prev = _lib._executor
    n = 16
    clip = SyntheticClip(48, 80, 2, n, seed=5)
    frames = [clip.frame(t) for t in range(n)]

    def run(patch, hinted, **cfg):
        rec = Recording()
        rec.per_sample_conv = True
        _lib.set_executor_for_testing(rec)
        MM.PATCH_PLANS = patch
        net = CUTIE(default_config())
        net.load_weights(make_state_dict(seed=0))
        proc = InferenceCore(net, cfg=default_config(mem_every=2, **cfg))
        outs = []
        with torch.inference_mode():
            for t in range(n):
                kw = dict(next_images=frames[t + 1:t + 9]) if hinted and t + 1 < n else {}
                outs.append(proc.step(frames[t], *((clip.first_mask(),) if t == 0 else ()), **(dict(objects=clip.objects) if t == 0 else {}), **kw))
        return torch.stack(outs), rec.seen

    lt = dict(use_long_term=True, long_term=dict(max_mem_frames=4, min_mem_frames=2, num_prototypes=16, max_num_tokens=64, buffer_tokens=32, count_usage=True))
    lt_nocount = dict(lt, long_term=dict(lt['long_term'], count_usage=False))
    from cutie_amd import ops as O
    patched = []
    orig_patch = O.OpList.patch_ints

    def counting(self, op, start, values):
        patched.append((int(self.recs[op][0]), start, len(values)))
        return orig_patch(self, op, start, values)
    O.OpList.patch_ints = counting
    try:
        for cfg in (lt, lt_nocount, dict(max_mem_frames=3)):
            for hinted in (False, True):
                del patched[:]
                out_a, seen_a = run(True, hinted, **cfg)
                kinds = {k for k, _, _ in patched}
                assert O.AFF_SCORE in kinds and O.AFF_SELECT in kinds, (cfg, hinted, kinds)       # the plans really were patched
                if hinted and cfg.get('use_long_term'):
                    assert O.USAGE_TICK in kinds, (cfg, kinds)
                if cfg is lt_nocount:
                    assert O.MEMSET32 in kinds, kinds
                del patched[:]
                out_b, seen_b = run(False, hinted, **cfg)
                assert not patched
                assert torch.equal(out_a, out_b), (cfg, hinted)
                assert len(seen_a) == len(seen_b), (cfg, hinted, len(seen_a), len(seen_b))
                for k, (a, b) in enumerate(zip(seen_a, seen_b)):
                    assert a == b, (cfg, hinted, k, [x for x, y in zip(a, b) if x != y][:1], [y for x, y in zip(a, b) if x != y][:1])
    finally:
        O.OpList.patch_ints = orig_patch
        MM.PATCH_PLANS = True
        _lib.set_executor_for_testing(prev)


def test_announced_frame_overwritten_in_place_is_encoded_again_cpu():
    """CPU twin (descriptor interpreter) of tests/test_gpu_parity.py::test_announced_frame_overwritten_in_place_is_encoded_again: frames
    are matched by storage AND tensor version, an announced frame that is overwritten in place goes through its own encoder."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.model.cutie import CUTIE
    from cutie_amd.utils.synth import SyntheticClip
    mx = MockExecutor()
    mx.per_sample_conv = True
    prev = _lib._executor
    _lib.set_executor_for_testing(mx)
    try:
        net = CUTIE(default_config())
        net.load_weights(make_state_dict(seed=0))
        n = 9
        clip = SyntheticClip(64, 96, 2, n + 1, seed=3)
        mask = clip.first_mask()

        def run(hinted):
            frames = [clip.frame(t) for t in range(n)]
            proc = InferenceCore(net, cfg=default_config(mem_every=3))
            outs = []
            for t in range(n):
                if t == 2:
                    frames[5].copy_(clip.frame(n))
                kw = dict(next_images=frames[t + 1:t + 9]) if hinted and t + 1 < n else {}
                outs.append(proc.step(frames[t], *((mask,) if t == 0 else ()), **(dict(objects=clip.objects) if t == 0 else {}), **kw))
            return torch.stack(outs)
        plain, got = run(False), run(True)                       # (no inference_mode: the frames keep their version counters)
        assert torch.equal(got, plain)
    finally:
        _lib.set_executor_for_testing(prev)


# (the last step raises in the middle of a frame: its features stay in the image feature store, which says so when it is dropped --
# "Leaking ... in the image feature store", the reference's own warning for the same situation, image_feature_store.py:47-49)
@pytest.mark.filterwarnings('ignore:Leaking:UserWarning')
def test_mask_narrower_than_the_frame_is_padded_on_its_own(product_net, oracle_net):
    """examples/masks/judo/00005.png is 480 x 853 while the frames are 480 x 854: the reference pads frame and mask separately
    (inference_core.py:231,263), so the mask lands with ITS pad offsets in the common padded size.  Same at a small size: frame
    96 x 122 (pads 3 | 3), first mask 96 x 121 (3 | 4), a later mask for a new object 96 x 120 (4 | 4)."""
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.utils.synth import SyntheticClip
    clip = SyntheticClip(96, 122, 3, 6, seed=9)
    full = clip.first_mask()
    m0 = (full * (full != 3).long())[:, :121].contiguous()
    m3 = (full * (full == 3).long())[:, 1:121].contiguous()
    proc = InferenceCore(product_net, cfg=default_config(mem_every=2))
    oproc = OracleProcessor(oracle_net, dict(DEFAULT_CFG, mem_every=2))
    with torch.inference_mode():
        for t in range(6):
            args = (m0, [1, 2]) if t == 0 else ((m3, [3]) if t == 3 else ())
            p = proc.step(clip.frame(t), *args[:1], **(dict(objects=args[1]) if args else {}))
            o = oproc.step(clip.frame(t), *args[:1], **(dict(objects=args[1]) if args else {}))
            assert p.shape == o.shape == (4 if t >= 3 else 3, 96, 122)
            assert float((p.float() - o).abs().max()) < (1e-5 if t == 0 else 6e-2), (t, float((p.float() - o).abs().max()))
        with pytest.raises(RuntimeError):
            proc.step(clip.frame(0), full[:, :100].contiguous(), objects=[1, 2, 3])     # pads to 96 x 112: no common padded size


def test_slot_pool_drops_idle_groups():
    """plans.SlotPool: a group (stage x objects x resolution) nobody has asked for in IDLE_FRAMES frames is dropped; live ones stay."""
    from cutie_amd.model import plans
    if plans._USE_COUNT is None:
        pytest.skip('needs the storage use count')
    pool = plans.SlotPool()
    pool.IDLE_FRAMES = 100
    spec = dict(x=((4, 4), torch.float32, False))
    a = pool.get(('a',), spec, 'cpu')['x']
    for f in range(300):
        pool.tick()
        b = pool.get(('b',), spec, 'cpu')['x']
    assert ('a',) not in pool.groups and ('b',) in pool.groups and ('a',) not in pool.last_used
    assert a.shape == (4, 4)                                  # (a caller that still holds a tensor keeps it)


def test_trajectory_without_lds_dma_tiles():
    """$CUTIE_AMD_DMA_TILES=0 (a documented A/B switch) puts every conv on a register-staged tile, which carries no side jobs;
    the plan variant that skips QUERY_INIT used to hang the clearing of its accumulators on the first conv regardless and died on the
    second read-out.  The switch is read at import time, hence a child process."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, CUTIE_AMD_DMA_TILES='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-x', '-q', '-k', 'test_trajectory_matches_oracle and small_fifo',
                        '-p', 'no:cacheprovider'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
