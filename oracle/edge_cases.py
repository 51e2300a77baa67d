"""Edge / error behaviour of the InferenceCore surface as a list of small scripts; each returns a JSON-able outcome:
('raise', exception type name) or ('ok', summary of the result).  ``oracle/make_edge_cases.py`` records the outcomes of the
EXECUTED reference into tests/golden/edge_cases.json; tests/test_edge_cases_cpu.py runs the same scripts on the product and
on the oracle.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import torch

from cutie_amd.utils.synth import SyntheticClip


def _clip():
    return SyntheticClip(80, 112, 3, 6, seed=7)


def _summary(t):
    t = t.detach().float().cpu()
    return {'shape': list(t.shape), 'all_zero': bool((t == 0).all()), 'sums_to_one': bool(((t.sum(0) - 1).abs() < 1e-3).all())}


def case_segment_without_memory(make):
    p = make({})                                                    # inference_core.py:146-150: warning + zeros [1,H16,W16]
    return _summary(p.step(_clip().frame(0)))


def case_empty_object_list(make):
    p, c = make({}), _clip()                                       # :289-295: warning + zeros
    return _summary(p.step(c.frame(0), torch.zeros_like(c.first_mask()), objects=[]))


def case_idx_mask_needs_objects(make):
    p, c = make({}), _clip()                                       # :183-186 assert not idx_mask
    return _summary(p.step(c.frame(0), c.first_mask()))


def case_unsorted_new_objects(make):
    p, c = make({}), _clip()                                       # object_manager.py:53: known objects must come first
    p.step(c.frame(0), (c.first_mask() == 2).long() * 2, objects=[2])
    return _summary(p.step(c.frame(1), c.first_mask(), objects=[1, 2]))


def case_first_frame_returns_the_mask(make):
    p, c = make({}), _clip()
    out = p.step(c.frame(0), c.first_mask(), objects=[1, 2, 3])
    am = out.argmax(0).cpu()
    return {**_summary(out), 'argmax_is_mask': bool((am == c.first_mask()).all())}


def case_mask_id_without_pixels(make):
    p, c = make({}), _clip()                                       # an object listed but absent from the mask: empty plane
    m = c.first_mask().clone()
    m[m == 2] = 0
    out = p.step(c.frame(0), m, objects=[1, 2, 3])
    return {**_summary(out), 'plane2_max_lt_half': bool(float(out[2].max()) < 0.5)}


def case_delete_unknown_object(make):
    p, c = make({}), _clip()
    p.step(c.frame(0), c.first_mask(), objects=[1, 2, 3])
    p.delete_objects([9])                                           # object_manager.py:56-77: ids not present are ignored
    return {**_summary(p.step(c.frame(1))), 'objects': list(p.object_manager.all_obj_ids) if hasattr(p, 'object_manager') else list(p.obj_ids)}


def case_delete_all_then_propagate(make):
    p, c = make({}), _clip()
    p.step(c.frame(0), c.first_mask(), objects=[1, 2, 3])
    p.delete_objects([1, 2, 3])
    return _summary(p.step(c.frame(1)))


def case_delete_all_then_new_mask(make):
    p, c = make({}), _clip()
    p.step(c.frame(0), c.first_mask(), objects=[1, 2, 3])
    p.delete_objects([1, 2, 3])
    out = p.step(c.frame(1), (c.first_mask() == 1).long() * 5, objects=[5])
    return {**_summary(out), 'mask': sorted(set(p.output_prob_to_mask(out).flatten().tolist()))}


def case_end_on_first_frame(make):
    p, c = make({}), _clip()                                       # end=True: the mask is returned but nothing is memorised
    a = _summary(p.step(c.frame(0), c.first_mask(), objects=[1, 2, 3], end=True))
    b = _summary(p.step(c.frame(1)))
    return {'first': a, 'second': b}


def case_output_prob_to_mask_ids(make):
    p, c = make({}), _clip()
    m = c.first_mask().clone()
    m[m == 1] = 7
    m[m == 3] = 200
    out = p.step(c.frame(0), m, objects=[2, 7, 200])
    ids = p.output_prob_to_mask(out)
    return {'dtype': str(ids.dtype), 'shape': list(ids.shape), 'ids': sorted(set(ids.flatten().tolist())), 'equal_mask': bool((ids.cpu() == m).all())}


def case_float_mask_without_objects(make):
    p, c = make({}), _clip()                                       # :183-186: objects default to 1..K for float masks
    planes = torch.stack([(c.first_mask() == i).float() for i in (1, 2, 3)])
    out = p.step(c.frame(0), planes, idx_mask=False)
    objs = list(p.object_manager.all_obj_ids) if hasattr(p, 'object_manager') else list(p.obj_ids)
    return {**_summary(out), 'objects': objs}


def case_odd_frame_size(make):
    p = make({})                                                    # 83 x 107: asymmetric pads, output at the input size
    c = SyntheticClip(83, 107, 2, 3, seed=3)
    a = p.step(c.frame(0), c.first_mask(), objects=[1, 2])
    b = p.step(c.frame(1))
    return {'first': _summary(a), 'second': _summary(b)}


def _mem(p):
    """[work tokens, permanent tokens] of bucket 0 through whichever surface the processor has."""
    if hasattr(p, 'memory'):
        return [p.memory.work_mem.size(0), p.memory.work_mem.perm_size(0)]
    return [p.work.size(0), p.work.perm_end.get(0, 0)]


def case_update_config_cannot_toggle_long_term(make):
    p, c = make({}), _clip()                                       # memory_manager.py:63: assert 'cannot update this'
    p.step(c.frame(0), c.first_mask(), objects=[1, 2, 3])
    cfg = p.cfg if hasattr(p, 'cfg') else None
    new = type(cfg)(cfg) if cfg is not None else {}
    new['use_long_term'] = True
    p.update_config(new)
    return 'updated'


def case_delete_objects_accepts_an_int(make):
    p, c = make({}), _clip()                                       # object_manager.py:59-60
    p.step(c.frame(0), c.first_mask(), objects=[1, 2, 3])
    p.delete_objects(2)
    out = p.step(c.frame(1))
    return {**_summary(out), 'mask': sorted(set(p.output_prob_to_mask(out).flatten().tolist()) - {0}) <= [1, 3]}


def case_clear_memory_then_propagate(make):
    p, c = make({}), _clip()
    p.step(c.frame(0), c.first_mask(), objects=[1, 2, 3])
    p.clear_memory()
    return _summary(p.step(c.frame(1)))


def case_force_permanent_on_a_propagated_frame(make):
    p, c = make(dict(mem_every=5)), _clip()                        # :308-315: (is_mem_frame or force_permanent) -> 'all'
    p.step(c.frame(0), c.first_mask(), objects=[1, 2, 3])
    out = p.step(c.frame(1), force_permanent=True)
    return {**_summary(out), 'mem': _mem(p)}


def case_int32_and_uint8_masks(make):
    p, c = make({}), _clip()
    a = p.step(c.frame(0), c.first_mask().to(torch.int32), objects=[1, 2, 3])
    b = p.step(c.frame(1), c.first_mask().to(torch.uint8), objects=[1, 2, 3])
    return {'first': _summary(a), 'second': _summary(b), 'mem': _mem(p)}


def case_memory_schedule(make):
    p, c = make(dict(mem_every=2, max_mem_frames=3)), SyntheticClip(80, 112, 2, 12, seed=7)
    sizes = []
    p.step(c.frame(0), c.first_mask(), objects=[1, 2])
    for t in range(1, 12):
        p.step(c.frame(t))
        sizes.append(_mem(p))
    return sizes                                                    # FIFO: permanent 35 + at most (3-1) x 35 working tokens


# ---- ObjectManager on its own (object_manager.py:7-149); `make` is only used to reach the processor's object manager ----------
def _om(make):
    p = make({})
    return getattr(p, 'object_manager', None)


def case_object_manager_bookkeeping(make):
    om = _om(make)
    if om is None:
        return 'no object manager'                                 # the oracle restates it as a plain list
    tmp, ids = om.add_new_objects([3, 7, 9])
    out = {'add': [tmp, ids], 'again': list(om.add_new_objects([7, 11])), 'all': om.all_obj_ids, 'num': om.num_obj,
           'has': [om.has_all([3, 11]), om.has_all([3, 4])], 'tmp_of_9': om.find_tmp_by_id(9), 'obj': repr(om.find_object_by_id(7))}
    om.delete_objects(7)
    out['after_delete'] = [om.all_obj_ids, om.find_tmp_by_id(9), sorted(om.tmp_id_to_obj)]
    cls = torch.tensor([[0, 1, 2], [3, 3, 0]])
    out['tmp_to_obj_cls'] = om.tmp_to_obj_cls(cls).tolist()
    oh = om.make_one_hot(torch.tensor([[3, 9, 0], [11, 7, 3]]))
    out['one_hot'] = [list(oh.shape), str(oh.dtype), oh.long().sum((1, 2)).tolist()]
    rd = om.realize_dict({3: torch.zeros(1, 2), 9: torch.ones(1, 2), 11: torch.full((1, 2), 2.0)})
    out['realize'] = [list(rd.shape), rd[0, :, 0].tolist()]
    return out


def case_object_manager_tmp_to_obj_mapping(make):
    om = _om(make)
    if om is None:
        return 'no object manager'
    om.add_new_objects([3, 7])
    return {str(k): v for k, v in om.get_tmp_to_obj_mapping().items()}    # object_manager.py:106-108


def case_object_manager_realize_dict_needs_every_object(make):
    om = _om(make)
    if om is None:
        raise NotImplementedError                                   # same outcome for the list-based oracle
    om.add_new_objects([1, 2])
    return list(om.realize_dict({1: torch.zeros(1)}).shape)


def case_object_manager_purge_inactive(make):
    om = _om(make)
    if om is None:
        return 'no object manager'
    om.add_new_objects([4, 5, 6])
    for _ in range(3):
        om.find_object_by_id(5).poke()
    om.find_object_by_id(6).poke()
    first = om.purge_inactive_objects(2)
    om.find_object_by_id(6).unpoke()
    second = om.purge_inactive_objects(0)
    empty = om.make_one_hot(torch.zeros(2, 2, dtype=torch.long)) if True else None
    return {'first': [first[0], list(first[1]), list(first[2])], 'second': [second[0], list(second[1]), list(second[2])],
            'left': om.all_obj_ids, 'one_hot_shape': list(empty.shape)}


def case_object_manager_empty_one_hot(make):
    om = _om(make)
    if om is None:
        return 'no object manager'
    oh = om.make_one_hot(torch.zeros(2, 3, dtype=torch.long))
    return [list(oh.shape), str(oh.dtype)]


def case_save_aux_on_read(make):
    p, c = make({'save_aux': True}), _clip()                       # memory_manager.py:197-206 (KeyError in eval mode, see INTENDED)
    p.step(c.frame(0), c.first_mask(), objects=c.objects)
    p.step(c.frame(1))
    aux = p.memory.aux
    return {k: (None if aux[k] is None else ([list(t.shape) for t in aux[k]] if isinstance(aux[k], (list, tuple)) else list(aux[k].shape)))
            for k in sorted(aux)}


# where the product deliberately does not reproduce the reference: case -> (product outcome, why)
INTENDED = {
    'object_manager_tmp_to_obj_mapping': (['ok', {'3': 1, '7': 2}],
                                          'the reference method unpacks (tmp id, object) the wrong way round and always raises'),
    'save_aux_on_read': (['ok', {'attn_mask': [1, 3, 16, 5, 7], 'p_weights': None, 'pixel_readout': [1, 3, 256, 5, 7],
                                 'q_logits': [[1, 3, 5, 7]] * 4, 'q_weights': None, 'sensory': [1, 3, 256, 5, 7]}],
                         "memory_manager.py:197-206 reads aux_features['attn_mask'], which only exists in training mode: the reference "
                         'raises KeyError on the first read with save_aux; the product returns the dict the code intends'),
}


CASES = {n[5:]: f for n, f in sorted(globals().items()) if n.startswith('case_')}


def run_case(name, make):
    try:
        with torch.inference_mode():
            return ['ok', CASES[name](make)]
    except Exception as e:                                          # noqa: BLE001 -- the exception type IS the recorded outcome
        return ['raise', type(e).__name__]
