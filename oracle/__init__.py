"""CPU oracle for the ISDF hot path — TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
anything from this package; the product (``pyscf_isdf_amd``) never does and fails loudly when its
HIP library is missing.

What is restated here and how it is pinned
------------------------------------------
* ``ao``, ``pbc_tools``, ``fftdf``: numpy restatement of the reference's *exact* FFTDF path
  (collocation, FFT conventions, Coulomb kernel, get_j / get_k / ERI) — pinned against the
  reference's own known-answer constants (tests/test_oracle_pins.py).
* ``isdf``: the ISDF stages (interpolation points, fit, Coulomb W, J/K).  The mounted reference
  (stock PySCF 2.5.0) contains no ISDF code, so for these stages **parity is unpinned by the
  reference**; they are pinned instead by (i) the reference's pivot rule
  (pyscf/lib/scipy_helper.py:71-110) through golden pivot lists generated from that file
  (tests/golden/make_pivoted_cholesky_golden.py), (ii) algebraic identities, and (iii) convergence
  of ISDF J/K to the pinned FFTDF J/K as the number of interpolation points grows.
"""
