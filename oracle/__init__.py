"""CPU oracle package (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  ``spf_amd`` (the product) never does.
"""
from .spf_oracle import *  # noqa: F401,F403
