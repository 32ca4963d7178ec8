"""CPU oracle for the MPC inner-solve path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy (float64) restatement of what the reference
computes on the path named in BASELINE.json's north_star:

    autompc/sysid/mlp.py      -> oracle/mlp.py      (MLP step + Jacobian)
    autompc/costs/cost.py     -> oracle/costs.py    (quadratic cost terms)
    autompc/control/mppi.py   -> oracle/mppi.py     (MPPI solve)
    autompc/control/ilqr.py   -> oracle/ilqr.py     (iLQR solve)
    autompc/utils/simulation.py + costs/cost.py:27-41 -> oracle/closed_loop.py

It exists to CHECK the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``autompc_amd/`` imports it, and the product path has no CPU fallback: without
the HIP library it raises.

Pinning: every module here is checked against golden vectors produced by the
real reference, imported in the build container by ``tests/golden/gen_golden.py``
(the reference itself cannot travel to the GPU box).  The nu>1 MPPI
generalisation has no reference to pin against (the reference raises
ValueError for ctrl_dim>1); it is pinned by the nu=1 goldens plus
a reduction property test (tests/test_oracle_mppi.py).  SINDy inference is
"parity unpinned" (pysindy is absent from the image and from /root/reference).
"""
