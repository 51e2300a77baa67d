"""The synthetic ``state_dict`` generator lives in cutie_amd/utils/synth_weights.py (benchmarks need it without importing the
oracle); re-exported here for the oracle, its golden-vector generator and the tests.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
from cutie_amd.utils.synth_weights import (CONV_GAINS, LINEAR_GAINS, MODEL_CFG, MODEL_CFG_SMALL, make_state_dict, param_spec)  # noqa: F401
