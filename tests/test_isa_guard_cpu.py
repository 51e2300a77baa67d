"""Static guard on the built gfx950 code (no GPU): the kernels of the frame must not contain `load; s_waitcnt vmcnt(0)` chains in
front of their loops again -- the dependent global round trips that round 4 found with tools/isa_waits.py and removed (profiles/
r04_isa_waits.md: conditional loads that hipcc sinks into per-element branches, run-time trip counts around loads).  The objects are the
ones `make` left in cutie_amd/csrc (the build step of __graft_entry__.build()); the code object is taken out of the fat binary and
disassembled, nothing is recompiled.  Skipped when the objects are not there."""
import os
import sys

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_waits as W                                      # noqa: E402

CSRC = os.path.join(ROOT, 'cutie_amd', 'csrc')
LLVM_OK = all(os.path.exists(W.LLVM + t) for t in ('llvm-objcopy', 'clang-offload-bundler', 'llvm-objdump'))


def _kernels(obj):
    path = os.path.join(CSRC, obj)
    if not (LLVM_OK and os.path.exists(path)):
        pytest.skip(f'{obj} not built here (or no LLVM tools)')
    return list(W.kernels_of_object(path))


def _lone_before_loops(body):
    return [h for h in W.scan(body) if not h[3]]


def test_affinity_kernels_request_their_operands_together():
    ks = dict(_kernels('affinity.o'))
    score = [n for n in ks if n.startswith('_Z16aff_score_kernelILi')]
    assert len(score) == 6, score                         # <1 | 2 query sets per wave, pass 0 | 1> + <2, pass 0 | 1, one bank per stacked frame>
    for n in score:
        assert _lone_before_loops(ks[n]) == [], (n, _lone_before_loops(ks[n]))
        pre, _ = W.wait_groups(ks[n])
        assert pre <= 1, (n, pre)                         # c_j / tau_j, the first memory group and the query operand: ONE round trip
    kp = [n for n in ks if 'key_prep_kernel' in n]
    assert len(kp) == 1 and len(_lone_before_loops(ks[kp[0]])) <= 1      # (the memory side's shrinkage load)


def test_elementwise_kernels_of_the_frame_have_no_load_wait_chains():
    ks = dict(_kernels('elementwise.o'))
    md = [n for n in ks if n.startswith('_Z21up4_softmax_md_kernelILi8ELi') and 'ELb1EE' in n and not n.startswith('_Z21up4_softmax_md_kernelILi8ELi0E')]
    f4 = [n for n in ks if n.startswith('_Z25up4_softmax_fused4_kernelILi8ELi') and not n.startswith('_Z25up4_softmax_fused4_kernelILi8ELi0E')]
    assert len(md) == 7 and len(f4) == 7, (md, f4)        # one instantiation per object count 1..7
    for n in md + f4:
        lone = _lone_before_loops(ks[n])
        assert len(lone) <= (1 if 'ILi8ELi1E' in n else 0), (n, lone)     # (K = 1: a single load is a lone load)
        assert W.wait_groups(ks[n])[1] <= 3, (n, W.wait_groups(ks[n]))    # 6 x K loads in at most K / 2 batches (was: one per load)
    for frag in ('gru4_kernel', 'area_down3_kernel', 'upsample2x_add_kernel', 'eca_apply_kernel'):
        for n in [n for n in ks if frag in n]:
            assert _lone_before_loops(ks[n]) == [], (n, _lone_before_loops(ks[n]))


def test_stem_kernel_batches_its_mask_planes():
    ks = dict(_kernels('stem.o'))
    (n,) = [n for n in ks if 'stem_kernel' in n]
    assert _lone_before_loops(ks[n]) == [] and len(W.scan(ks[n])) == 0, W.scan(ks[n])


def test_conv_pc_consumers_preload_the_bias_in_one_round_trip():
    ks = _kernels('conv_pc.o')
    pc = [(n, b) for n, b in ks if n.startswith('_Z14conv_pc_kernel')]
    assert len(pc) >= 100
    worst = max(len([h for h in _lone_before_loops(b) if 'global_load_dword ' in h[1] or 'flat_load' in h[1]]) for n, b in pc)
    assert worst == 0, worst                              # (was TNP * NCH = 8 ... 16 dependent dword loads per consumer wave)
    assert not any('flat_load' in line for n, b in pc for line in b)      # the bias pointer is a global pointer
