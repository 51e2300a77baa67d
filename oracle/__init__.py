"""CPU oracle for the Cutie per-frame inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cutie_amd/`` (the product) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the CPU
baseline -- never as the thing measured or shipped.

The oracle is a plain torch-fp32 restatement (written from scratch, functional
style over a flat weight dict) of the algorithm the reference implements in
``cutie/model/*`` and ``cutie/inference/*``; every function cites the reference
file:line it follows.

Parity pin: the reference ships no tests or golden vectors,
so the oracle is pinned against the *executed reference*: ``oracle/make_golden.py``
imports the unmodified reference from ``/root/reference`` (CPU, fp32), runs it on
seeded inputs with the deterministic synthetic weights (``cutie_amd/utils/synth_weights.py``) and commits
sub-sampled outputs under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks this restatement against those vectors.  ``oracle/fuzz_reference.py`` additionally compares the
oracle with the live reference on random event scripts (agreement to ~2e-6), and
``oracle/make_api_surface.py`` records the reference's public signatures for the drop-in test.
"""
