"""TEST INFRASTRUCTURE ONLY.

CPU restatement (oracle) of the Fair-MARL rollout hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package; the product (``fair_marl_amd``) never does.
"""
