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
