"""Record the outcomes of oracle/edge_cases.py on the EXECUTED reference -> tests/golden/edge_cases.json.
Run in the build container only:  python oracle/make_edge_cases.py      TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
from oracle.make_golden import GOLDEN, _wrap, import_reference, reference_cfg      # noqa: E402


def main():
    import oracle.edge_cases                                       # noqa: F401  (before the reference takes over `cutie`)
    CUTIE, InferenceCore = import_reference()
    from oracle.edge_cases import CASES, run_case
    from oracle.weights import make_state_dict
    net = CUTIE(reference_cfg()).eval()
    net.load_weights({k: v.clone() for k, v in make_state_dict(seed=0).items()})
    wrap = lambda over: reference_cfg(**{k: (_wrap(v) if isinstance(v, dict) else v) for k, v in over.items()})
    out = {name: run_case(name, lambda over: InferenceCore(net, cfg=wrap(over))) for name in CASES}
    json.dump(out, open(os.path.join(GOLDEN, 'edge_cases.json'), 'w'), indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, v)


if __name__ == '__main__':
    main()
