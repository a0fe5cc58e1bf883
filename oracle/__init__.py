"""CPU oracle for the VAE-GSLM training-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``vae-gslm_amd/``) never imports this package and fails
loudly when the HIP library is missing.

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the
reference itself, generated in the build container by
``tests/golden/gen_golden.py`` (which imports ``/root/reference``) and
committed as ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` checks
the oracle against every one of those vectors.
"""
