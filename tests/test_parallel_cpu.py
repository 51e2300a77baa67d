"""Clip-sharding logic of cutie_amd/parallel.py on CPU: two gloo processes, independent clips, result gather."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from cutie_amd.parallel import run_sharded, shard_clips
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    assert shard_clips(5, rank, world) == [c for c in range(5) if c % world == rank]

    def run_clip(c):
        g = torch.Generator().manual_seed(c)
        masks = torch.randint(0, 4, (3, 8, 10), generator=g).to(torch.uint8)
        return {'frames': 3 + c, 'seconds': 0.5 * (c + 1), 'masks': masks}

    res = run_sharded(list(range(5)), run_clip, gather_masks=True)
    if rank == 0:
        assert sorted(res.keys()) == list(range(5))
        for c in range(5):
            g = torch.Generator().manual_seed(c)
            want = torch.randint(0, 4, (3, 8, 10), generator=g).to(torch.uint8)
            assert res[c]['frames'] == 3 + c and torch.equal(res[c]['masks'], want)
    else:
        assert res is None
    # rank 0 owns NO clip (custom placement): it must still receive every mask -- the receive buffers live on the communication
    # device of the backend, not on the device of some local result
    res = run_sharded([7, 8, 9], run_clip, gather_masks=True, owner_of=lambda i: 1)
    if rank == 0:
        assert sorted(res.keys()) == [7, 8, 9]
        for c in (7, 8, 9):
            g = torch.Generator().manual_seed(c)
            want = torch.randint(0, 4, (3, 8, 10), generator=g).to(torch.uint8)
            assert res[c]['frames'] == 3 + c and torch.equal(res[c]['masks'], want)
        out.put('ok')
    else:
        assert res is None
    dist.destroy_process_group()


def test_clip_sharding_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 'ok'


def _model_worker(rank, world, port, out):
    """The bench protocol (cutie_amd.parallel.timed_steps, what bench.py runs under torch.distributed.run) and the clip-sharded
    result gather with the REAL per-frame path on every rank: InferenceCore over the launch plans, executed by the descriptor
    interpreter (no GPU here)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    torch.set_num_threads(2)
    from cutie_amd import _lib
    from cutie_amd.config import default_config
    from cutie_amd.inference.inference_core import InferenceCore
    from cutie_amd.model.cutie import CUTIE
    from cutie_amd.parallel import run_sharded, timed_steps
    from cutie_amd.utils.synth import SyntheticClip
    from cutie_amd.utils.synth_weights import make_state_dict
    from mock_exec import MockExecutor
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _lib.set_executor_for_testing(MockExecutor())
    cfg = default_config(mem_every=2)
    net = CUTIE(cfg)
    net.load_weights(make_state_dict(seed=0))

    def run_clip(c, frames=4):
        clip = SyntheticClip(48, 64, 2, frames, seed=30 + c)
        proc = InferenceCore(net, cfg=cfg)
        ids = []
        with torch.inference_mode():
            p = proc.step(clip.frame(0), clip.first_mask(), objects=clip.objects)
            ids.append(proc.output_prob_to_mask(p, dtype=torch.uint8))
            for t in range(1, frames):
                ids.append(proc.output_prob_to_mask(proc.step(clip.frame(t)), dtype=torch.uint8))
        return {'frames': frames, 'seconds': 0.1, 'masks': torch.stack(ids)}

    # 1. bench.py's timed region: one clip per rank, the slower rank (rank 1 sleeps) sets the time on every rank
    import time
    clip = SyntheticClip(48, 64, 2, 8, seed=rank)
    proc = InferenceCore(net, cfg=cfg)
    with torch.inference_mode():
        proc.step(clip.frame(0), clip.first_mask(), objects=clip.objects)

        def step(i):
            proc.step(clip.frame(1 + i % 7))
            if rank == 1:
                time.sleep(0.05)

        t0 = time.perf_counter()
        tmax = timed_steps(step, steps=3, warmup=1, device='cpu')
        mine = time.perf_counter() - t0
    both = [None, None]
    dist.all_gather_object(both, tmax)
    assert both[0] == both[1] and tmax >= 3 * 0.05 and tmax <= mine + 1e-3, (both, mine)
    # 2. clips sharded over the ranks, id masks gathered on rank 0 (direct sends): identical to the same clips run locally
    res = run_sharded(list(range(3)), run_clip, gather_masks=True)
    if rank == 0:
        assert sorted(res) == [0, 1, 2]
        for c in range(3):
            assert torch.equal(res[c]['masks'], run_clip(c)['masks']), c
        out.put('ok')
    else:
        assert res is None
    dist.destroy_process_group()


def test_bench_protocol_and_model_two_ranks_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 'ok'


def _run_bench(extra, env=None):
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-interpreter', '--steps', '3', '--warmup', '1', '--preroll', '2', '--height', '48',
           '--width', '80', '--objects', '2', '--window', '4'] + extra
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env or {})
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` -- the shape of the driver's command -- starts two ranks itself (torch.distributed.run, one process per
    GPU; here over gloo with the descriptor interpreter, `--cpu-interpreter`), rank 0 prints ONE line whose n_gpus is the size of the process
    group, with every rank's own frames/s; one rank stays one process; a launcher that started another number of ranks than --gpus asks for
    is refused (the flag used to be parsed and dropped)."""
    r, line = _run_bench(['--gpus', '2'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert line['n_gpus'] == 2 and len(line['per_rank_fps']) == 2 and line['steps'] == 3 and line['scaling'] == 'weak'
    assert line['config']['parallelism'] == 'clip-shard x2'
    assert abs(line['value'] - 2 * 3 / (line['ms_per_step'] * 3e-3)) < 0.05 * line['value']          # whole-job frames / the slowest rank's time
    r, line = _run_bench(['--gpus', '1'])
    assert r.returncode == 0 and line['n_gpus'] == 1 and len(line['per_rank_fps']) == 1, r.stderr[-2000:]
    r, line = _run_bench(['--gpus', '2'], env={'WORLD_SIZE': '3', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and line is None and '--gpus 2' in r.stderr and 'WORLD_SIZE=3' in r.stderr
