"""TEST INFRASTRUCTURE ONLY -- the parity oracle for the CFG-DDPM hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and there only as the checker / the timed CPU baseline -- never as the
thing that is shipped or measured as the MI355X path.

* ``oracle/cpu_path.py``   CPU (torch fp32) restatement of the reference's
  algorithm, function by function, each citing the reference file:line.
* ``oracle/reference_loader.py``  loads the *real* reference classes from
  ``/root/reference`` (build container only; the GPU box has no reference).
* ``oracle/gen_golden.py`` runs the real reference and writes the golden
  vectors under ``tests/golden/`` that pin ``cpu_path`` (parity is PINNED:
  see ``tests/test_oracle_golden.py``).
"""
