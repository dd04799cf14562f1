"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's low-bit forward path.  Nothing under
``oracle/`` is part of the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the checker.
"""
