"""CPU oracle for the Epipolar Transformer hot path -- TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  The product package (`epipolar_transformers_amd`) never
does, and fails loudly when its HIP library is missing instead of falling back
here.
"""
