"""Time the UNMODIFIED reference (/root/reference, CPU, fp32) beside the oracle on the bench workload (synthetic 854x480, 3 objects,
long-term memory, eval_config defaults), same cores, same frames: relates bench.py's `cpu_baseline` (kind "port" = the oracle, the
only CPU path that exists on the GPU box) to the reference itself.  Build container only.

    python -m oracle.time_reference [frames]      # -> tests/golden/cpu_reference_ratio.json

TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import json
import os
import sys
import time

import torch

from .make_golden import GOLDEN, _wrap, import_reference, reference_cfg


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    CUTIE, InferenceCore = import_reference()
    from cutie_amd.utils.synth import SyntheticClip
    from oracle.inference import DEFAULT_CFG, OracleProcessor
    from oracle.net import OracleNet
    from oracle.weights import make_state_dict
    sd = make_state_dict(seed=0)
    clip = SyntheticClip(480, 854, 3, n + 2, seed=1)
    net = CUTIE(reference_cfg()).eval()
    net.load_weights({k: v.clone() for k, v in sd.items()})
    lt = dict(DEFAULT_CFG['long_term'])
    ref = InferenceCore(net, cfg=reference_cfg(use_long_term=True, long_term=_wrap(lt)))
    ora = OracleProcessor(OracleNet(sd), dict(DEFAULT_CFG, use_long_term=True))
    out = {}
    with torch.inference_mode():
        for name, proc in (('reference', ref), ('oracle', ora)):
            proc.step(clip.frame(0), clip.first_mask(), objects=clip.objects)
            proc.step(clip.frame(1))
            t0 = time.perf_counter()
            for t in range(2, 2 + n):
                proc.step(clip.frame(t))
            out[name] = n / (time.perf_counter() - t0)
    rec = {'cores': cores, 'frames': n, 'reference_fps': round(out['reference'], 3), 'oracle_fps': round(out['oracle'], 3),
           'oracle_over_reference': round(out['oracle'] / out['reference'], 3),
           'workload': 'synthetic 854x480 3-object clip, long-term memory on, frames 2..%d, fp32, torch %s' % (1 + n, torch.__version__)}
    json.dump(rec, open(os.path.join(GOLDEN, 'cpu_reference_ratio.json'), 'w'), indent=1)
    print(rec)


if __name__ == '__main__':
    main()
