"""ORACLE — TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference's perception hot path, used
solely as the checker: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package.  The product path
(``active-perception-using-neural-radiance-fields_amd/``) never imports it and
fails loudly when its HIP library is missing.

Parity status:
  * nerfacc half (ray/AABB, marcher, scans, volrend, occupancy update, vanilla
    MLP, ray generation): pinned against golden vectors captured by importing the
    reference's CPU-runnable code (tests/golden/make_golden.py) and against the
    reference tests' known answers.
  * tiny-cuda-nn half (hash grid, spherical harmonics, fully-fused MLPs):
    **parity unpinned** — tiny-cuda-nn is an un-vendored, un-pinned dependency
    (README.md:47-48, perception/models/requirements.txt:1); oracle/field.py
    restates its published algorithm and is anchored only on the reference's call
    sites (perception/models/radiance_fields/ngp.py:103-238).
"""
