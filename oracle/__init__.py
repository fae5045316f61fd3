"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the LIST SDF query hot path.

Nothing in the product package may import this directory.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
only as the checker.  Parity status: PINNED by golden vectors generated from the
reference's own Python modules (``oracle/gen_golden.py`` -> ``tests/golden``).
"""
