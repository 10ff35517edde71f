"""CPU oracle for the gcn10 curve-number path.  TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package, and only as the checker / reported baseline.
PARITY UNPINNED -- see ``oracle/README.md``.
"""
