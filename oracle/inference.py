"""Oracle frame processor: restatement of the reference per-frame runtime
(``InferenceCore`` + ``MemoryManager`` + ``KeyValueMemoryStore`` + ``ObjectManager``)
in plain torch-fp32 on CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  bs = 1; flip_aug = a second lane (see __init__); chunk_size > 0 groups
the objects in the memory read-out like memory_manager.py:169-186 (every BASELINE config runs with chunk_size = -1).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .net import OracleNet, aggregate, get_similarity, topk_softmax

DEFAULT_CFG = dict(
    # cutie/config/eval_config.yaml:16-51 with the d17-val dataset defaults (:64-72)
    mem_every=5, stagger_updates=5, chunk_size=-1, max_internal_size=-1, top_k=30,
    use_long_term=False, max_mem_frames=5,
    long_term=dict(count_usage=True, max_mem_frames=10, min_mem_frames=5, num_prototypes=128,
                   max_num_tokens=10000, buffer_tokens=2000),
)


def pad_divide_by(x, d):
    # cutie/utils/tensor_utils.py:7-22
    h, w = x.shape[-2:]
    new_h = h + d - h % d if h % d > 0 else h
    new_w = w + d - w % d if w % d > 0 else w
    lh, uh = int((new_h - h) / 2), int(new_h - h) - int((new_h - h) / 2)
    lw, uw = int((new_w - w) / 2), int(new_w - w) - int((new_w - w) / 2)
    pad = (lw, uw, lh, uh)
    return F.pad(x, pad), pad


def unpad(x, pad):
    # cutie/utils/tensor_utils.py:25-44 (3-D case)
    if pad[2] + pad[3] > 0:
        x = x[:, pad[2]:x.shape[1] - pad[3], :]
    if pad[0] + pad[1] > 0:
        x = x[:, :, pad[0]:x.shape[2] - pad[1]]
    return x


class _Store:
    """One key/value store (working or long-term).  cutie/inference/kv_memory_store.py:19-53"""

    def __init__(self, save_selection, save_usage):
        self.save_selection, self.save_usage = save_selection, save_usage
        self.next_bucket = 0
        self.buckets = {}       # bucket -> [obj ids]
        self.k, self.s, self.e = {}, {}, {}          # bucket -> [CK,N] / [N] / [CK,Nnp]
        self.use, self.life = {}, {}                 # bucket -> [Nnp]
        self.perm_end = {}                           # bucket -> int
        self.v = {}                                  # obj -> [CV,N]

    def size(self, b):
        return self.k[b].shape[-1] if b in self.k else 0

    def non_perm_size(self, b):
        return self.size(b) - self.perm_end.get(b, 0)

    def engaged(self, b=None):
        return len(self.buckets) > 0 if b is None else b in self.buckets

    @staticmethod
    def _cat(d, key, new, prepend):
        if key in d:
            d[key] = torch.cat([new, d[key]], -1) if prepend else torch.cat([d[key], new], -1)
        else:
            d[key] = new

    def add(self, key, values, shrinkage, selection, supposed_bucket=-1, as_permanent='no'):
        # kv_memory_store.py:55-149
        ne = key.shape[-1]
        if supposed_bucket >= 0:
            enabled = [supposed_bucket]
            exist = supposed_bucket in self.buckets
            for obj, val in values.items():
                if exist:
                    self._cat(self.v, obj, val, as_permanent == 'all')
                else:
                    self.v[obj] = val
            self.buckets[supposed_bucket] = list(values.keys())
        else:
            new_bucket = None
            enabled = []
            for obj, val in values.items():
                if obj in self.v:
                    self._cat(self.v, obj, val, as_permanent == 'all')
                    b = [bb for bb, ids in self.buckets.items() if obj in ids][0]
                else:
                    self.v[obj] = val
                    if new_bucket is None:
                        new_bucket = self.next_bucket
                        self.next_bucket += 1
                        self.buckets[new_bucket] = []
                    self.buckets[new_bucket].append(obj)
                    b = new_bucket
                if b not in enabled:
                    enabled.append(b)
        add_perm = {}
        for b in enabled:
            add_perm[b] = False
            if as_permanent == 'all':
                self.perm_end[b] = self.perm_end.get(b, 0) + ne
                add_perm[b] = True
            elif as_permanent == 'first' and self.perm_end.get(b, 0) == 0:
                self.perm_end[b] = ne
                add_perm[b] = True
            self.perm_end.setdefault(b, 0)
        for b in self.buckets:
            if b not in enabled:
                continue
            self._cat(self.k, b, key, add_perm[b])
            self._cat(self.s, b, shrinkage, add_perm[b])
            if not add_perm[b]:
                if self.save_selection:
                    self._cat(self.e, b, selection, False)
                if self.save_usage:
                    self._cat(self.use, b, torch.zeros(ne), False)
                    self._cat(self.life, b, torch.zeros(ne) + 1e-7, False)

    def update_usage(self, b, usage):
        # kv_memory_store.py:151-162
        if not self.save_usage:
            return
        usage = usage[self.perm_end[b]:]
        if usage.shape[-1] == 0:
            return
        self.use[b] = self.use[b] + usage
        self.life[b] = self.life[b] + 1

    def sieve(self, b, start, end, min_size):
        # kv_memory_store.py:164-204 : keep non-permanent elements outside [start, end)
        n_np = self.non_perm_size(b)
        if n_np <= min_size:
            return
        p = self.perm_end[b]
        N = self.k[b].shape[-1]
        end_abs = N + 1 if end == 0 else N + end       # python negative index
        start_abs = start + p
        self.k[b] = torch.cat([self.k[b][:, :start_abs], self.k[b][:, end_abs:]], -1)
        self.s[b] = torch.cat([self.s[b][:start_abs], self.s[b][end_abs:]], -1)
        # selection/usage arrays exclude the permanent part; the reference slices them with the same
        # *negative* end (or the out-of-range size+1), i.e. relative to their own length
        Nnp = N - p
        end_np = Nnp + 1 if end == 0 else Nnp + end
        if self.save_selection:
            self.e[b] = torch.cat([self.e[b][:, :start], self.e[b][:, end_np:]], -1)
        if self.save_usage:
            self.use[b] = torch.cat([self.use[b][:start], self.use[b][end_np:]], -1)
            self.life[b] = torch.cat([self.life[b][:start], self.life[b][end_np:]], -1)
        for obj in self.buckets[b]:
            self.v[obj] = torch.cat([self.v[obj][:, :start_abs], self.v[obj][:, end_abs:]], -1)

    def remove_obsolete(self, b, max_size):
        # kv_memory_store.py:209-242 (long-term only)
        usage = self.use[b] / self.life[b]
        _, surv = torch.topk(usage, k=max_size)
        self.k[b] = self.k[b][:, surv]
        self.s[b] = self.s[b][surv]
        for obj in self.buckets[b]:
            self.v[obj] = self.v[obj][:, surv]
        self.use[b] = self.use[b][surv]
        self.life[b] = self.life[b][surv]

    def purge_except(self, keep):
        # kv_memory_store.py:280-303
        keep = set(keep)
        dead = []
        for b, ids in self.buckets.items():
            self.buckets[b] = [o for o in ids if o in keep]
            if not self.buckets[b]:
                dead.append(b)
        self.v = {o: v for o, v in self.v.items() if o in keep}
        for b in dead:
            for d in (self.buckets, self.k, self.s, self.e, self.use, self.life):
                d.pop(b, None)


class OracleProcessor:
    """Restatement of ``InferenceCore`` (cutie/inference/inference_core.py:18-345)."""

    def __init__(self, net: OracleNet, cfg=None):
        self.net = net
        cfg = dict(DEFAULT_CFG if cfg is None else cfg)
        self.cfg = cfg
        self.mem_every = cfg['mem_every']
        su = cfg['stagger_updates']
        if su >= self.mem_every:                                   # inference_core.py:37-41
            self.stagger_ti = set(range(1, self.mem_every + 1))
        else:
            self.stagger_ti = set(np.round(np.linspace(1, self.mem_every, su)).astype(int).tolist())
        self.max_internal_size = cfg['max_internal_size']
        self.chunk_size = cfg.get('chunk_size', -1)
        self.top_k = cfg['top_k']
        self.use_long_term = cfg['use_long_term']
        lt = cfg['long_term']
        if self.use_long_term:                                      # memory_manager.py:30-38
            self.max_mem_frames = lt['max_mem_frames'] - 1
            self.min_mem_frames = lt['min_mem_frames'] - 1
            self.num_prototypes = lt['num_prototypes']
            self.max_long_tokens = lt['max_num_tokens']
            self.buffer_tokens = lt['buffer_tokens']
        else:
            self.max_mem_frames = cfg['max_mem_frames'] - 1
        self.count_lt_usage = lt['count_usage']
        # flip_aug (inference_core.py:142-143,162-165,234-235,303-305): the reference runs [image, flipped image] as a batch of
        # two through everything; the two batch elements never interact except for the averaged prediction, so the second
        # one is restated as a lane with its own memory that shares the object list
        self.flip_aug = bool(cfg.get('flip_aug', False))
        self._flip = OracleProcessor(net, dict(cfg, flip_aug=False)) if self.flip_aug else None
        self.curr_ti, self.last_mem_ti = -1, 0
        self.obj_ids = []            # tmp id = position + 1   (object_manager.py)
        self.last_mask = None
        self._reset_memory()

    def _reset_memory(self):
        self.work = _Store(self.use_long_term, self.use_long_term)
        self.long = _Store(False, self.count_lt_usage) if self.use_long_term else None
        self.sensory, self.obj_v = {}, {}
        self.engaged = False
        self.HW = None
        self.config_stale = True

    # ---- interactive surface (inference_core.py:52-69, memory_manager.py:59-75,377-383) ------------
    def clear_memory(self):
        # inference_core.py:52-55: a NEW MemoryManager built from the construction-time cfg (working / long-term / sensory /
        # object memory all gone); the object manager -- and with it the tmp ids -- survives
        if self._flip is not None:
            self._flip.clear_memory()
        self.curr_ti, self.last_mem_ti = -1, 0
        mem_every = self.mem_every
        OracleProcessor.update_config(self, self.cfg)               # memory limits back to the original cfg ...
        self.mem_every = mem_every                                  # ... but InferenceCore.mem_every is not touched
        self._reset_memory()

    def clear_sensory_memory(self):
        # inference_core.py:62-65, memory_manager.py:382-383: re-created as zeros by the next _add_memory (:360-366)
        if self._flip is not None:
            self._flip.clear_sensory_memory()
        self.curr_ti, self.last_mem_ti = -1, 0
        self.sensory = {}

    def clear_non_permanent_memory(self):
        if self._flip is not None:
            self._flip.clear_non_permanent_memory()
        self.curr_ti, self.last_mem_ti = -1, 0
        for b in list(self.work.buckets):
            self.work.sieve(b, 0, 0, 0)                               # kv_memory_store.py:305-308
        if self.use_long_term:
            for b in list(self.long.buckets):
                self.long.sieve(b, 0, 0, 0)

    def update_config(self, cfg):
        if self._flip is not None:
            self._flip.update_config(cfg)
        cfg = dict(DEFAULT_CFG, **cfg) if isinstance(cfg, dict) else cfg
        self.config_stale = True                                    # memory_manager.py:59: token limits are re-derived at the next add
        self.mem_every = cfg['mem_every']
        self.top_k = cfg['top_k']
        lt = cfg['long_term']
        assert self.use_long_term == cfg['use_long_term'] and self.count_lt_usage == lt['count_usage'], 'cannot update this'
        if self.use_long_term:
            self.max_mem_frames = lt['max_mem_frames'] - 1
            self.min_mem_frames = lt['min_mem_frames'] - 1
            self.num_prototypes = lt['num_prototypes']
            self.max_long_tokens = lt['max_num_tokens']
            self.buffer_tokens = lt['buffer_tokens']
        else:
            self.max_mem_frames = cfg['max_mem_frames'] - 1

    # ---- object manager ----------------------------------------------------------
    def _add_objects(self, objects):
        tmp = []
        for o in objects:
            if o not in self.obj_ids:
                self.obj_ids.append(o)
            tmp.append(self.obj_ids.index(o) + 1)
        assert tmp == sorted(tmp)                                    # object_manager.py:53
        return tmp

    def delete_objects(self, objects):
        # inference_core.py:330-335, memory_manager.py:298-307; a single id is accepted too (object_manager.py:59-60)
        if isinstance(objects, int):
            objects = [objects]
        if self._flip is not None:
            self._flip.obj_ids = list(self.obj_ids)
            self._flip.delete_objects(objects)
        self.obj_ids = [o for o in self.obj_ids if o not in objects]
        self.work.purge_except(self.obj_ids)
        if self.use_long_term and self.long.engaged():
            self.long.purge_except(self.obj_ids)
        self.sensory = {k: v for k, v in self.sensory.items() if k in self.obj_ids}
        if not self.work.engaged():
            self.engaged = False

    def output_prob_to_mask(self, prob):
        # inference_core.py:337-345
        mask = torch.argmax(prob, dim=0)
        out = torch.zeros_like(mask)
        for i, o in enumerate(self.obj_ids):
            out[mask == i + 1] = o
        return out

    # ---- memory read (memory_manager.py:112-208) -----------------------------------
    def _read(self, pix_feat, key, selection):
        h, w = pix_feat.shape[-2:]
        qk = key.flatten(2)[0]
        qe = selection.flatten(2)[0]
        out = {}
        for b, objs in self.work.buckets.items():
            if self.use_long_term and self.long.engaged(b):
                nl = self.long.size(b)
                mk = torch.cat([self.long.k[b], self.work.k[b]], -1)
                ms = torch.cat([self.long.s[b], self.work.s[b]], -1)
                aff, usage = topk_softmax(get_similarity(mk, ms, qk, qe), self.top_k)
                self.work.update_usage(b, usage[nl:])
                if self.count_lt_usage:
                    self.long.update_usage(b, usage[:nl])
            else:
                aff, usage = topk_softmax(get_similarity(self.work.k[b], self.work.s[b], qk, qe), self.top_k)
                if self.use_long_term:
                    self.work.update_usage(b, usage)
            vals = []
            for o in objs:
                v = self.work.v[o]
                if self.use_long_term and o in self.long.v:
                    v = torch.cat([self.long.v[o], v], -1)
                vals.append(v)
            V = torch.stack(vals, 0)                                  # [K,CV,N]
            ro_all = (V.flatten(0, 1) @ aff).view(1, len(objs), V.shape[1], h, w)
            # chunk_size > 0 (memory_manager.py:169-186): fusion and object transformer run per group of objects, so the
            # "others" mask and the transformer's foreground/background attention masks only see the objects of the group
            cs = self.chunk_size if self.chunk_size >= 1 else len(objs)
            for c0 in range(0, len(objs), cs):
                sub = objs[c0:c0 + cs]
                ro = ro_all[:, c0:c0 + cs]
                sens = torch.stack([self.sensory[o] for o in sub], 1)
                lm = self.last_mask[:, [self.obj_ids.index(o) for o in sub]]
                fused = self.net.pixel_fusion(pix_feat, ro, sens, lm)
                om = torch.stack([self.obj_v[o] for o in sub], 1).unsqueeze(2)
                rq = self.net.readout_query(fused, om)
                for i, o in enumerate(sub):
                    out[o] = rq[:, i]
                if self.cfg.get('save_aux', False):
                    # memory_manager.py:197-206 builds the aux dict from aux_features['attn_mask'], which QueryTransformer only
                    # sets in training mode (object_transformer.py:171-175): with save_aux the EXECUTED reference raises here on
                    # the first read (recorded in tests/golden/edge_cases.json)
                    raise KeyError('attn_mask')
        return out

    # ---- memory write (memory_manager.py:210-296, 309-358) -------------------------
    def _consolidate(self, b):
        mw = self.min_work_tokens
        p = self.work.perm_end[b]
        N = self.work.size(b)
        ck = self.work.k[b][:, p:N - mw]
        cs = self.work.s[b][p:N - mw]
        ce = self.work.e[b][:, :N - p - mw]
        usage = (self.work.use[b] / self.work.life[b])[:N - p - mw]
        cv = {o: self.work.v[o][:, p:N - mw] for o in self.work.buckets[b]}
        _, idx = torch.topk(usage, k=self.num_prototypes, dim=-1, sorted=True)
        pk, pe = ck[:, idx], ce[:, idx]
        aff, _ = topk_softmax(get_similarity(ck, cs, pk, pe), None)   # dense softmax over candidates
        pv = {o: v @ aff for o, v in cv.items()}
        ps = cs[None, :] @ aff
        self.work.sieve(b, 0, -mw, mw)
        self.long.add(pk, pv, ps[0], None, supposed_bucket=b)

    def _add_memory(self, image, pix_feat, prob, key, shrinkage, selection, force_permanent=False):
        # inference_core.py:71-121
        if prob.shape[1] == 0:
            return
        as_perm = 'all' if force_permanent else 'first'
        for o in self.obj_ids:
            if o not in self.sensory:
                self.sensory[o] = torch.zeros(1, self.net.m['sensory_dim'], *key.shape[-2:])
        sens = torch.stack([self.sensory[o] for o in self.obj_ids], 1)
        value, new_sens, summ = self.net.encode_mask(image, pix_feat, sens, prob, deep_update=True)
        self.engaged = True
        if self.HW is None or self.config_stale:                    # memory_manager.py:228-235
            self.config_stale = False
            self.HW = key.shape[-2] * key.shape[-1]
            self.max_work_tokens = self.max_mem_frames * self.HW
            if self.use_long_term:
                self.min_work_tokens = self.min_mem_frames * self.HW
        for i, o in enumerate(self.obj_ids):                          # streaming sum :252-271
            if o in self.obj_v:
                self.obj_v[o] = self.obj_v[o] + summ[:, i]
            else:
                self.obj_v[o] = summ[:, i].clone()
        vals = {o: value[0, i].flatten(1) for i, o in enumerate(self.obj_ids)}
        self.work.add(key.flatten(2)[0], vals, shrinkage.flatten(2)[0, 0],
                      selection.flatten(2)[0] if self.use_long_term else None, as_permanent=as_perm)
        for b in list(self.work.buckets.keys()):
            if self.use_long_term:
                if self.work.non_perm_size(b) >= self.max_work_tokens:
                    if self.long.non_perm_size(b) >= self.max_long_tokens - self.num_prototypes:
                        self.long.remove_obsolete(b, self.max_long_tokens - self.num_prototypes - self.buffer_tokens)
                    self._consolidate(b)
            else:
                self.work.sieve(b, 0, -self.max_work_tokens, self.max_work_tokens)
        self.last_mem_ti = self.curr_ti
        for i, o in enumerate(self.obj_ids):
            self.sensory[o] = new_sens[:, i]

    # ---- segmentation (inference_core.py:123-170) -----------------------------------
    def _segment(self, key, selection, pix_feat, ms, update_sensory):
        if not self.engaged:
            return torch.zeros(1, key.shape[-2] * 16, key.shape[-1] * 16)
        ro = self._read(pix_feat, key, selection)
        ro = torch.stack([ro[o] for o in self.obj_ids], 1)
        sens = torch.stack([self.sensory[o] for o in self.obj_ids], 1)
        new_sens, _, prob = self.net.segment(ms, ro, sens, update_sensory=update_sensory)
        if update_sensory:
            for i, o in enumerate(self.obj_ids):
                self.sensory[o] = new_sens[:, i]
        return prob[0]

    # ---- InferenceCore.step (inference_core.py:172-328) -------------------------------
    def step(self, image, mask=None, objects=None, *, idx_mask=True, end=False, force_permanent=False):
        if objects is None and mask is not None:
            assert not idx_mask
            objects = list(range(1, mask.shape[0] + 1))
        resize_needed = False
        if self.max_internal_size > 0:
            h, w = image.shape[-2:]
            min_side = min(h, w)
            if min_side > self.max_internal_size:
                resize_needed = True
                nh, nw = int(h / min_side * self.max_internal_size), int(w / min_side * self.max_internal_size)
                image = F.interpolate(image.unsqueeze(0), size=(nh, nw), mode='bilinear', align_corners=False)[0]
                if mask is not None:
                    if idx_mask:
                        mask = F.interpolate(mask.unsqueeze(0).unsqueeze(0).float(), size=(nh, nw),
                                             mode='nearest-exact')[0, 0].round().long()
                    else:
                        mask = F.interpolate(mask.unsqueeze(0), size=(nh, nw), mode='bilinear',
                                             align_corners=False)[0]
        self.curr_ti += 1
        image, self.pad = pad_divide_by(image, 16)
        image = image.unsqueeze(0)
        is_mem_frame = ((self.curr_ti - self.last_mem_ti >= self.mem_every) or (mask is not None)) and (not end)
        need_segment = (mask is None) or (len(self.obj_ids) > 0 and not all(o in self.obj_ids for o in objects))
        update_sensory = ((self.curr_ti - self.last_mem_ti) in self.stagger_ti) and (not end)

        ms, pix_feat = self.net.encode_image(image)
        key, shrinkage, selection = self.net.transform_key(ms[0])
        fl = self._flip
        if fl is not None:
            fl.obj_ids, fl.curr_ti = self.obj_ids, self.curr_ti
            image_f = torch.flip(image, dims=[-1])                      # of the PADDED frame (:231-235)
            ms_f, pix_f = self.net.encode_image(image_f)
            key_f, shr_f, sel_f = self.net.transform_key(ms_f[0])
        if need_segment:
            prob = self._segment(key, selection, pix_feat, ms, update_sensory)
            if fl is not None:
                prob = (prob + torch.flip(fl._segment(key_f, sel_f, pix_f, ms_f, update_sensory), dims=[-1])) / 2
        if mask is not None:
            tmp_ids = self._add_objects(objects)
            mask, _ = pad_divide_by(mask, 16)
            if need_segment:
                no_bg = prob[1:]
                if idx_mask:
                    no_bg[:, mask > 0] = 0
                else:
                    no_bg[:, mask.max(0)[0] > 0.5] = 0
                new_masks = []
                for mi, t in enumerate(tmp_ids):
                    this = (mask == objects[mi]).float() if idx_mask else mask[t]   # sic: reference indexes by tmp id (:276)
                    if t > no_bg.shape[0]:
                        new_masks.append(this.unsqueeze(0))
                    else:
                        no_bg[t - 1] = this
                mask = torch.cat([no_bg, *new_masks], dim=0)
            elif idx_mask:
                if len(objects) == 0:
                    return torch.zeros(1, key.shape[-2] * 16, key.shape[-1] * 16)
                mask = torch.stack([mask == objects[mi] for mi, _ in enumerate(tmp_ids)], dim=0)
            prob = torch.softmax(aggregate(mask, dim=0), dim=0)
        self.last_mask = prob[1:].unsqueeze(0)
        if fl is not None:
            fl.obj_ids = self.obj_ids
            fl.last_mask = torch.flip(self.last_mask, dims=[-1])
        if is_mem_frame or force_permanent:
            self._add_memory(image, pix_feat, self.last_mask, key, shrinkage, selection,
                             force_permanent=force_permanent)
            if fl is not None:
                fl._add_memory(image_f, pix_f, fl.last_mask, key_f, shr_f, sel_f, force_permanent=force_permanent)
        out = unpad(prob, self.pad)
        if resize_needed:
            out = F.interpolate(out.unsqueeze(0), size=(h, w), mode='bilinear', align_corners=False)[0]
        return out
