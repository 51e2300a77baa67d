"""Teacher-forced per-frame parity harness.  TEST INFRASTRUCTURE ONLY.

Free-running trajectories with synthetic weights are chaotic (one near-tie flip in the top-k read-out moves boundary
probabilities by a few 1e-2 in the *fp32* reference itself, tests/test_oracle_golden.py), so they cannot carry a tight bound.
Here the oracle runs the clip on its own (the teacher); before every frame its complete recurrent state -- sensory memory, last
mask, object memory, working / permanent / long-term bank with usage counters, frame counters, object list -- is copied into the
product, the product runs ONE ``step`` on the same input and is compared with the oracle's result for that frame.  Every frame is
therefore a single-step comparison from an identical state: the error cannot accumulate, and a kernel that is wrong on some
geometry shows up at the first frame that uses it.

``inject_state`` writes the product's bank through the product's own layout (Bucket slabs, KEY_PREP kernel for the similarity
operands), so it works with the HIP executor and with the descriptor interpreter alike.
"""
import torch

from cutie_amd import ops as O
from cutie_amd.inference.kv_memory_store import Bucket
from cutie_amd.inference.memory_manager import MemoryManager
from cutie_amd.inference.object_manager import ObjectManager

BF16, F32 = torch.bfloat16, torch.float32


def inject_state(proc, o, _om=None):
    """Overwrite the recurrent state of the product ``InferenceCore`` with that of the ``OracleProcessor`` o -- both lanes when
    flip_aug is on (the flipped lane has its own bank / sensory state / last mask and shares the object manager)."""
    assert (proc._flip is None) == (o._flip is None), 'flip_aug must match'
    dev = proc.network.device
    lt = o.use_long_term
    proc.curr_ti, proc.last_mem_ti, proc.mem_every = o.curr_ti, o.last_mem_ti, o.mem_every
    proc._prefetched = None
    proc.image_feature_store._store.clear()
    if _om is None:
        om = ObjectManager()
        if o.obj_ids:
            om.add_new_objects([int(x) for x in o.obj_ids])
    else:
        om = _om
    proc.object_manager = om
    if o._flip is not None:
        o._flip.obj_ids = list(o.obj_ids)
        inject_state(proc._flip, o._flip, _om=om)
    mem = MemoryManager(cfg=proc.cfg, object_manager=om)
    proc.memory = mem
    mem.top_k = o.top_k
    mem.max_mem_frames = o.max_mem_frames
    if lt:
        mem.min_mem_frames, mem.num_prototypes = o.min_mem_frames, o.num_prototypes
        mem.max_long_tokens, mem.buffer_tokens = o.max_long_tokens, o.buffer_tokens
    proc.last_mask = None if o.last_mask is None else o.last_mask.to(device=dev, dtype=F32).contiguous()
    # ---- sensory / object memory
    ids = [x for x in o.obj_ids if x in o.sensory]
    if ids:
        s = torch.cat([o.sensory[x] for x in ids], 0).permute(0, 2, 3, 1).contiguous()        # [K,h,w,CS]
        mem._ids = list(ids)
        mem._sens_f32 = s.to(device=dev, dtype=F32).contiguous()
        mem._sens_bf16 = mem._sens_f32.to(BF16)
    vids = [x for x in o.obj_ids if x in o.obj_v]
    if vids:
        mem._objv_ids = list(vids)
        mem._objv = torch.cat([o.obj_v[x] for x in vids], 0).to(device=dev, dtype=F32).contiguous()
    for x, v in o.obj_v.items():
        if x not in o.obj_ids:
            mem._orphan_objv[x] = v[0].to(device=dev, dtype=F32).clone()
    mem.engaged = o.engaged
    if o.HW is None:
        return
    # ---- bank geometry
    some = next(iter(o.sensory.values())) if o.sensory else None
    if some is not None:
        mem.H, mem.W = some.shape[-2:]
    else:
        mem.H, mem.W = o.last_mask.shape[-2] // 16, o.last_mask.shape[-1] // 16
    mem.HW = o.HW
    assert mem.H * mem.W == mem.HW
    mem.max_work_tokens = o.max_work_tokens
    if lt:
        mem.min_work_tokens = o.min_work_tokens
    mem.config_stale = False
    mem._next_bucket = o.work.next_bucket
    ol = O.OpList()
    keep = []
    for b, objs in o.work.buckets.items():
        k = o.work.k[b]
        CK, N = k.shape
        CV = o.work.v[objs[0]].shape[0]
        mem.CK, mem.CV = CK, CV
        p = o.work.perm_end.get(b, 0)
        n_long = o.long.size(b) if (lt and o.long.engaged(b)) else 0
        bk = Bucket(b, list(objs), o.HW, CK, CV, dev, use_long_term=lt, work_cap=mem.max_work_tokens,
                    long_cap=mem.max_long_tokens if lt else 0)
        if p > bk.P:
            bk.grow_perm(p)
        bk.n_long, bk.n_perm, bk.n_work = n_long, p, N - p
        assert bk.n_work <= bk.Wc and n_long <= bk.L
        if not lt and bk.Wc >= o.HW:
            bk.ring = bk.n_work // o.HW

        def put(start, keys, shr, sel, use, life, vals):
            n = keys.shape[1]
            if n == 0:
                return
            rk = keys.t().contiguous().to(device=dev, dtype=F32)
            rs = shr.contiguous().to(device=dev, dtype=F32)
            keep.extend([rk, rs])
            ol.key_prep(rk, rs, bk.Ahi[start:], bk.Alo[start:], bk.scale[start:], n=n, query=False)
            if lt:
                bk.rawkey[start:start + n] = rk
                bk.rawshr[start:start + n] = rs
                if sel is not None:
                    bk.rawsel[start:start + n] = sel.t().to(device=dev, dtype=F32)
                if use is not None:
                    bk.use[start:start + n] = use.to(device=dev, dtype=F32)
                    bk.life[start:start + n] = life.to(device=dev, dtype=F32)
            for x in objs:
                bk.values[x][start:start + n] = vals[x].t().to(device=dev, dtype=BF16)

        wv = {x: o.work.v[x] for x in objs}
        put(bk.perm_start, k[:, :p], o.work.s[b][:p], None, None, None, {x: v[:, :p] for x, v in wv.items()})
        put(bk.work_start, k[:, p:], o.work.s[b][p:], o.work.e.get(b) if lt else None,
            o.work.use.get(b) if lt else None, o.work.life.get(b) if lt else None, {x: v[:, p:] for x, v in wv.items()})
        if n_long:
            has_use = b in o.long.use
            put(0, o.long.k[b], o.long.s[b], None, o.long.use[b] if has_use else torch.zeros(n_long),
                o.long.life[b] if has_use else torch.full((n_long,), 1e-7), {x: o.long.v[x] for x in objs})
        mem.buckets[b] = bk
    if len(ol):
        ol.run()
    if dev.type == 'cuda':
        torch.cuda.synchronize()


def bank_sizes(proc):
    m = proc.memory
    return [sum(b.n_perm + b.n_work for b in m.buckets.values()), sum(b.n_perm for b in m.buckets.values()),
            sum(b.n_long for b in m.buckets.values()), len(m.buckets)]


def oracle_bank_sizes(o):
    return [sum(o.work.size(b) for b in o.work.buckets), sum(o.work.perm_end[b] for b in o.work.buckets),
            sum(o.long.size(b) for b in o.long.buckets) if o.use_long_term else 0, len(o.work.buckets)]


def run_teacher_forced(steps, make_oracle, make_product, device, *, deletes=None, check_state=True, margins=(0.04, 0.1, 0.3)):
    """steps: [(image, mask | None, objects | None)].  Returns a list of per-frame dicts:
    t, max / mean |dprob|, margin-aware argmax agreement, bank sizes equal, sensory rel. error after the step."""
    deletes = deletes or {}
    oproc, proc = make_oracle(), make_product()
    rows = []
    with torch.inference_mode():
        for t, (img, mask, objs) in enumerate(steps):
            if t in deletes:
                oproc.delete_objects(deletes[t])
            inject_state(proc, oproc)
            if mask is not None:
                o = oproc.step(img, mask, objects=objs)
                p = proc.step(img.to(device), mask.to(device), objects=objs)
            else:
                o = oproc.step(img)
                p = proc.step(img.to(device))
            p = p.detach().float().cpu()
            assert p.shape == o.shape and bool(torch.isfinite(p).all()), (t, p.shape, o.shape)
            d = (p - o).abs()
            row = dict(t=t, max=float(d.max()), mean=float(d.mean()), sizes_equal=bank_sizes(proc) == oracle_bank_sizes(oproc))
            if o.shape[0] > 1:
                top2 = o.topk(2, dim=0)[0]
                margin = top2[0] - top2[1]
                agree = p.argmax(0) == o.argmax(0)
                row['flips_above_margin'] = {m: int((~agree & (margin > m)).sum()) for m in margins}
                row['agree_all'] = float(agree.float().mean())
            if check_state and oproc.sensory and proc.memory._ids:
                so = torch.cat([oproc.sensory[x] for x in proc.memory._ids], 0)                     # [K,CS,h,w]
                sp = proc.memory._sens_f32.permute(0, 3, 1, 2).float().cpu()
                row['sensory_rel'] = float((sp - so).abs().max() / so.abs().max().clamp(min=1e-6))
            rows.append(row)
    return rows
