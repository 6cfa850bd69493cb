"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's (tum-vision/mem) pretraining hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import anything from this package, and only as the checker / the
reported CPU baseline -- never as part of the product path (``mem_amd/``).

Pinning: every function here is checked against the reference itself, imported
in the build container by ``oracle/gen_golden.py`` (which needs
``/root/reference`` and therefore never runs on the GPU box); the resulting
input/output vectors are committed under ``tests/golden/`` and re-checked by the
``-m "not gpu"`` tests.  The reference ships no tests or golden vectors of its
own, so the reference-run-here is the pin.
"""
