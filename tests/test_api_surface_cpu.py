"""Drop-in boundary: the product classes accept every call the reference classes accept.
tests/golden/api_surface.json is recorded by introspecting the unmodified reference (oracle/make_api_surface.py)."""
import importlib
import inspect
import json
import os

import pytest

SURFACE = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'api_surface.json')))

# reference methods that are NOT part of the boundary: internals of the reference's storage that no caller outside the class
# uses (SURVEY 8b lists what callers touch); the product's memory bank is laid out differently (contiguous HBM buffers)
INTERNAL = {
    'cutie.inference.kv_memory_store.KeyValueMemoryStore': {
        '__init__', 'add', 'get_all_sliced', 'get_usage', 'remove_obsolete_features', 'remove_old_memory', 'sieve_by_range',
        'update_bucket_usage', 'purge_except', 'clear_non_permanent_memory', 'get_v_size'},
    'cutie.inference.memory_manager.MemoryManager': {'compress_features', 'consolidation'},
}
INTERNAL_PROPERTIES = {
    'cutie.inference.kv_memory_store.KeyValueMemoryStore': {'key', 'value', 'shrinkage', 'selection'},
}


def _load(qualname):
    mod, cls = qualname.rsplit('.', 1)
    return getattr(importlib.import_module(mod), cls)          # through the `cutie` alias package = what a caller imports


@pytest.mark.parametrize('qualname', sorted(SURFACE))
def test_class_accepts_the_reference_calls(qualname):
    cls = _load(qualname)
    ref = SURFACE[qualname]
    problems = []
    for name, params in ref['methods'].items():
        if name in INTERNAL.get(qualname, ()):
            continue
        fn = cls if (name == '__call__' and inspect.isfunction(cls)) else getattr(cls, name, None)
        if fn is None or not callable(fn):
            problems.append(f'{name}: missing')
            continue
        have = [p for p in inspect.signature(fn).parameters.values() if p.name != 'self']
        by_name = {p.name: p for p in have}
        var_pos = any(p.kind is p.VAR_POSITIONAL for p in have)
        var_kw = any(p.kind is p.VAR_KEYWORD for p in have)
        positional = [p for p in have if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
        pos_i = 0
        for rp in params:
            if rp['kind'] in ('VAR_POSITIONAL', 'VAR_KEYWORD'):
                if not (var_pos if rp['kind'] == 'VAR_POSITIONAL' else var_kw):
                    problems.append(f'{name}: *{rp["name"]} not accepted')
                continue
            if rp['kind'] == 'POSITIONAL_OR_KEYWORD':
                if pos_i < len(positional):
                    p = positional[pos_i]
                    if p.name != rp['name']:
                        problems.append(f'{name}: positional #{pos_i} is {p.name!r}, reference {rp["name"]!r}')
                elif not (var_pos and var_kw):
                    problems.append(f'{name}: positional {rp["name"]!r} not accepted')
                    continue
                else:
                    continue
                pos_i += 1
            else:                                               # keyword-only in the reference
                p = by_name.get(rp['name'])
                if p is None:
                    if not var_kw:
                        problems.append(f'{name}: keyword {rp["name"]!r} not accepted')
                    continue
            if rp['default'] is not None:
                if p.default is inspect.Parameter.empty:
                    problems.append(f'{name}: {rp["name"]} has no default (reference {rp["default"]})')
                elif repr(p.default) != rp['default']:
                    problems.append(f'{name}: default of {rp["name"]} is {p.default!r}, reference {rp["default"]}')
        # nothing the reference does not pass may be required
        known = {rp['name'] for rp in params}
        for p in have:
            if p.name not in known and p.default is inspect.Parameter.empty and p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD):
                problems.append(f'{name}: extra required parameter {p.name!r}')
    for prop in ref['properties']:
        if prop in INTERNAL_PROPERTIES.get(qualname, ()):
            continue
        if not isinstance(inspect.getattr_static(cls, prop, None), property):
            problems.append(f'property {prop}: missing')
    assert not problems, '\n'.join(problems)
