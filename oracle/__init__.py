"""ORACLE - test infrastructure only.

CPU restatements of the reference's solver algorithms, used by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg as the *checker*.  Nothing under
``spcies_amd/`` imports this package; the product path is the HIP library and fails loudly
without it.
"""
