"""BASELINE.json's configurations on the GPU, each at the largest size one MI355X holds, checked against the reference's
EXACT exchange (pyscf/pbc/df/fft_jk.py:177-302, evaluated on the same GPU by isdf_get_k_exact, itself pinned to the
reference's fp(vk) in test_gpu_parity.py) and against the CPU oracle where that finishes in seconds.

What is asserted and why.  BASELINE.json's north star asks for J/K "within 1e-6 Eh of the reference CPU path".  J is the
reference's own formula (no fit): asserted to 1e-9 Eh.  K goes through the ISDF fit, whose error is set by the number of
interpolation points c_isdf * nao and their selection (DESIGN.md section 2 has the measured scan):
  * configs[1] (16 atoms): global selection at c = 15 meets the literal 1e-6 Eh (measured 6.1e-7) - asserted;
    the scalable path (refined selection) is asserted at 1e-6 Eh PER ATOM at the same c (this cell's grid is denser per atom
    than the headline's, 32000 against 13500 points: it needs c = 15 where the headline needs 12);
  * configs[2] (128 atoms, the headline): the configuration bench.py times (refined selection, c = 12, fit rows in two
    panels) is asserted at 1e-6 Eh PER ATOM (measured 2.8e-7 Eh/atom = 3.6e-5 Eh) and max|dK| <= 1e-4; the literal 1e-6 Eh
    is NOT reached at this size inside 30 s (c = 15: 9.3e-6 Eh in 26.3 s) and no test pretends otherwise;
  * configs[4] (64 H2O) at the largest single-GPU mesh: 1e-5 Eh per atom class with the block-Jacobi clusters (see the test);
  * configs[0], configs[3]: see the tests below.
"""
import numpy as np
import pytest
import cells  # noqa: F401
from pyscf_isdf_amd import gto, workloads
from oracle import ao as oao, isdf as oisdf, fftdf

pytestmark = pytest.mark.gpu

PER_ATOM_TOL = 1e-6          # Eh per atom, the accuracy class asserted for the scalable (refined-selection) path
NORTH_STAR_TOL = 1e-6        # Eh, BASELINE.json


def _ek(vk, dm):
    return float(np.einsum('ij,ji', vk, dm) / 4)


def test_config0_diamond_primitive_szv_40_full_size_vs_oracle_pipeline():
    """configs[0] (diamond primitive cell, gth-szv, 40^3 = 64000 points, c = 10 -> 80 points) at FULL size against the
    complete CPU oracle pipeline: collocation <= 1e-12, identical interpolation points (the oracle is fed the GPU's phi:
    symmetric crystal, ties), W <= 1e-9 relative, J <= 1e-10, K <= 1e-9 relative; the device's exact exchange against the
    oracle's FFTDF restatement <= 1e-10; and the ISDF fit error against that exact K (full pair rank is 36 < 80 points:
    the fit is exact to rounding)."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = workloads.make_cell('diamond-prim-szv-40')
    nao = cell.nao_nr()
    assert nao == 8 and int(np.prod(cell.mesh)) == 64000
    dm, c, occ = workloads.make_dm(cell)
    df = ISDF(cell, c_isdf=10, select='global')
    vj, vk = df.get_jk(dm)
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    ao_ref = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, rule='point')
    ao_gpu = df.backend.to_host(df.ao)
    assert abs(ao_gpu - ao_ref.T).max() < 1e-12
    a, mesh = cell.lattice_vectors(), cell.mesh
    ref = oisdf.build_global(ao_gpu, a, mesh, 10 * nao)
    assert np.array_equal(ref['ip'], df.ip)
    W = df.backend.to_host(df.W)
    assert abs(W - ref['W']).max() < 1e-9 * abs(ref['W']).max()
    assert abs(vj - oisdf.get_j(ao_gpu, dm, a, mesh)).max() < 1e-10
    k_or = oisdf.get_k(ref['aoP'], ref['W'], dm)
    assert abs(vk - k_or).max() < 1e-9 * abs(k_or).max()
    k_exact = fftdf.get_k(np.ascontiguousarray(ao_gpu.T), dm, a, mesh)
    assert abs(df.get_k_exact(mo_coeff=c, mo_occ=occ) - k_exact).max() < 1e-10
    assert len(df.ip) <= 80 and abs(_ek(vk, dm) - _ek(k_exact, dm)) < NORTH_STAR_TOL
    assert abs(vk - k_exact).max() < 1e-6


def test_config1_diamond222_accuracy_vs_exact_exchange():
    """configs[1] (diamond 2x2x2, gth-dzvp, 80^3) against the exact exchange on the same density: global selection at c = 15
    meets the north-star 1e-6 Eh literally (measured 6.1e-7); the scalable path (refined selection, c = 15) is within 1e-6 Eh
    per atom (measured 3.6e-6 Eh = 2.2e-7 per atom) and beats the plain local selection at equal c in max|dK| (1.1e-5 against
    1.3e-4); J equals the exact path's J (same formula) to 1e-9 Eh."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = workloads.make_cell('diamond-222-dzvp-80')
    dm, c, occ = workloads.make_dm(cell)
    df = ISDF(cell, c_isdf=15, select='global')
    vj, vk = df.get_jk(dm)
    k_exact = df.get_k_exact(mo_coeff=c, mo_occ=occ)
    assert abs(_ek(vk, dm) - _ek(k_exact, dm)) < NORTH_STAR_TOL
    ref = ISDF(cell, c_isdf=15, select='refined')
    vj2, vk2 = ref.get_jk(dm)
    assert abs(np.einsum('ij,ji', vj - vj2, dm)) / 2 < 1e-9                       # J does not depend on the fit
    assert abs(_ek(vk2, dm) - _ek(k_exact, dm)) < PER_ATOM_TOL * cell.natm
    loc = ISDF(cell, c_isdf=15, select='local')
    vk3 = loc.get_jk(dm, with_j=False)[1]
    assert abs(vk2 - k_exact).max() < 0.5 * abs(vk3 - k_exact).max()


def test_config2_headline_diamond444_accuracy_vs_exact_exchange():
    """configs[2] (diamond 4x4x4, gth-dzvp, 120^3: N = 1664, G = 1 728 000) exactly as bench.py times it - refined selection,
    c = 12 (P = 19968), block-Jacobi route with the fit rows in two panels - against the exact exchange on the benchmark
    density (48 s on the GPU): |dE_K| <= 1e-6 Eh per atom (1.28e-4 Eh; measured 3.6e-5), max|dK| <= 1e-4 (measured 6.1e-5);
    size-independent properties on top (symmetry, linearity, the route's probe check passed)."""
    import torch
    from pyscf_isdf_amd.isdf import ISDF
    if torch.cuda.get_device_properties(0).total_memory < 270 * 2 ** 30:
        pytest.skip('needs a 288 GB device')
    cell = workloads.make_cell('diamond-444-dzvp-120')
    dm, c, occ = workloads.make_dm(cell)
    df = ISDF(cell, c_isdf=12, select='refined')
    vj, vk = df.get_jk(dm)
    assert len(df.ip) == 19968 and len(np.unique(df.ip)) == 19968
    assert df.fit_route_used == 'blockjacobi' and df.n_panels == 2 and df.bj_check <= df.bj_check_tol
    assert abs(vj - vj.T).max() < 1e-8 and abs(vk - vk.T).max() < 1e-7
    vk_half = df.get_jk(-0.5 * dm, with_j=False)[1]
    assert abs(vk_half + 0.5 * vk).max() < 1e-9
    assert abs(np.einsum('ij,ji', vj, dm) / 2 - 12.140270972643) < 1e-7           # E_J: the exact formula, same as round 1
    k_exact = df.get_k_exact(mo_coeff=c, mo_occ=occ)
    assert abs(_ek(k_exact, dm) - 123.18058922) < 1e-6                           # the exact exchange itself is stable
    assert abs(_ek(vk, dm) - _ek(k_exact, dm)) < PER_ATOM_TOL * cell.natm
    assert abs(vk - k_exact).max() < 1e-4
    df.reset()


def test_config4_water64_largest_single_gpu_mesh_vs_exact_exchange():
    """configs[4] (64 H2O, gth-dzvp) needs 160^3 x 14720 rows = 482 GB of fit rows at c = 10: more than one GPU holds in one
    piece, so the single-GPU test runs the largest mesh whose build finishes in about a minute here - 108^3 (G = 1 259 712) -
    with the paneled build ready to take over when the rows do not fit.  Asserted against the exact exchange on the
    benchmark density: |dE_K| <= 1e-5 Eh per atom, the probe check passed with molecular preconditioner blocks."""
    import torch
    from pyscf_isdf_amd.isdf import ISDF
    if torch.cuda.get_device_properties(0).total_memory < 270 * 2 ** 30:
        pytest.skip('needs a 288 GB device')
    cell = workloads.make_cell('water64-dzvp-108')
    dm, c, occ = workloads.make_dm(cell)
    df = ISDF(cell, c_isdf=12, select='refined')
    vj, vk = df.get_jk(dm)
    assert df.fit_route_used in ('blockjacobi', 'cholesky')
    assert abs(vk - vk.T).max() < 1e-7
    k_exact = df.get_k_exact(mo_coeff=c, mo_occ=occ)
    assert abs(_ek(vk, dm) - _ek(k_exact, dm)) < 1e-5 * cell.natm
    df.reset()


def _mgo_density(name):
    cell = workloads.make_cell(name)
    kpts = workloads.make_kpts(name, cell)
    nk, nao = len(kpts), cell.nao_nr()
    rng = np.random.default_rng(20240203)
    occ = np.zeros(nao); occ[:cell.nelectron // 2] = 2
    dms, cs = [], []
    for k in range(nk):
        c = np.linalg.qr(rng.standard_normal((nao, nao)) + 1j * rng.standard_normal((nao, nao)))[0]
        cs.append(c)
        dms.append((c * occ).dot(c.conj().T))
    return cell, kpts, np.array(dms), np.array(cs), np.tile(occ, (nk, 1))


def test_config3_mgo_kmesh_accuracy_vs_exact_kpoint_exchange_reduced_cell():
    """configs[3] (MgO rocksalt, gth-dzvp, 2x2x2 k-mesh) on the 2x2x2 cell / 64^3: the k-point ISDF exchange against the
    reference's exact k-point exchange evaluated on the same GPU (isdf_get_k_exact_kpt, pinned to the reference's fp(vk1) in
    test_gpu_kpts.py; 64 (k1, k2) pairs x 216 x 64 complex FFT pairs).  The number of points per AO is a user knob at k-points
    (k_ip_factor): the default (2) and 4 are both measured (profiles/r03_kpoint_accuracy_mgo222.log: factor 2: dE_K +5.2e-7 Eh per
    cell, max|dK| 3.8e-5; factor 3: 7.2e-6 / 7.3e-6; factor 4: 7.1e-6 / 6.8e-6 - the matrix error falls with the factor, the
    signed energy error is not monotonic at this level).  Asserted: |dE_K| per cell <= 2e-5 Eh and max|dK| <= 1e-4 at the
    default, max|dK| <= 2e-5 at factor 4."""
    import torch
    from pyscf_isdf_amd.isdf import ISDF
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip('needs a large device')
    name = 'mgo-222-dzvp-k222'
    cell, kpts, dms, cs, occs = _mgo_density(name)
    nk = len(kpts)
    df = ISDF(cell, kpts=kpts, c_isdf=10, select='refined')
    vk_ex = df.get_k_exact(dms, mo_coeff=cs, mo_occ=occs)
    ek_ex = np.einsum('kij,kji', vk_ex, dms).real / 4 / nk
    assert abs(vk_ex - vk_ex.conj().transpose(0, 2, 1)).max() < 1e-9 * abs(vk_ex).max()
    errs = {}
    for fac in (2, 4):
        df = ISDF(cell, kpts=kpts, c_isdf=10, select='refined')
        df.k_ip_factor = fac
        vk = df.get_jk(dms, kpts=kpts, with_j=False)[1]
        errs[fac] = (abs(np.einsum('kij,kji', vk, dms).real / 4 / nk - ek_ex), abs(vk - vk_ex).max())
        print('MgO 2x2x2 k222 c=10 k_ip_factor=%d P=%d: |dE_K| %.2e Eh per cell, max|dK| %.2e' % (fac, len(df.ip), errs[fac][0], errs[fac][1]))
        df.reset()
    assert errs[2][0] < 2e-5 and errs[4][0] < 2e-5
    assert errs[2][1] < 1e-4 and errs[4][1] < 2e-5


def test_config3_mgo_kmesh_properties_reduced_cell():
    """configs[3] (MgO rocksalt, gth-dzvp, 2x2x2 k-mesh) on the 2x2x2 cell with a 64^3 mesh (the 3x3x3 / 96^3 run takes two
    minutes on one GPU and is recorded in profiles/r02_cfg4_mgo333_k222_single_gpu.log): size-independent properties of the
    k-point path - J and K Hermitian at every k, linear in the density matrices, the +-q pairing consistent
    (J, K from D and from conj(D) at -k related by complex conjugation: the 2x2x2 mesh maps onto itself under k -> -k; exact
    for J, to the fit error for K)."""
    import torch
    from pyscf_isdf_amd.isdf import ISDF
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip('needs a large device')
    name = 'mgo-222-dzvp-k222'
    cell = workloads.make_cell(name)
    kpts = workloads.make_kpts(name, cell)
    nk, nao = len(kpts), cell.nao_nr()
    rng = np.random.default_rng(20240203)
    occ = np.zeros(nao); occ[:cell.nelectron // 2] = 2
    dms = []
    for k in range(nk):
        c = np.linalg.qr(rng.standard_normal((nao, nao)) + 1j * rng.standard_normal((nao, nao)))[0]
        dms.append((c * occ).dot(c.conj().T))
    dms = np.array(dms)
    df = ISDF(cell, kpts=kpts, c_isdf=10, select='refined')
    vj, vk = df.get_jk(dms, kpts=kpts)
    assert vj.shape == vk.shape == (nk, nao, nao)
    assert abs(vj - vj.conj().transpose(0, 2, 1)).max() < 1e-9 and abs(vk - vk.conj().transpose(0, 2, 1)).max() < 1e-8 * abs(vk).max()
    vj2, vk2 = df.get_jk(np.stack([dms, -0.5 * dms]), kpts=kpts)
    assert abs(vj2[1] + 0.5 * vj).max() < 1e-9 and abs(vk2[1] + 0.5 * vk).max() < 1e-9 * abs(vk).max()
    # time reversal: the mesh is symmetric under k -> -k (mod G); with D'^{k} = conj(D^{-k}) the results obey K'^{k} = conj(K^{-k})
    a = cell.lattice_vectors()
    frac = kpts.dot(a.T) / (2 * np.pi)
    minus = [int(np.argmin(abs(((frac + f) - np.round(frac + f))).sum(axis=1))) for f in frac]      # index of -k (mod G)
    dms_tr = np.array([dms[minus[k]].conj() for k in range(nk)])
    vj_tr, vk_tr = df.get_jk(dms_tr, kpts=kpts)
    # J is exact: the relation holds to rounding.  The ISDF K obeys it to its FIT error only: -k = k + G0 on this mesh, and the
    # fitted exchange at q = k2 - k1 and at q - G0 differ by the interpolation error of exp(i G0.r) x the pair products
    # (measured 1.6e-5 relative at c = 10) - the same size as the error against the exact exchange
    assert abs(vj_tr - np.array([vj[minus[k]].conj() for k in range(nk)])).max() < 1e-9
    assert abs(vk_tr - np.array([vk[minus[k]].conj() for k in range(nk)])).max() < 1e-4 * abs(vk).max()
    ej = np.einsum('kij,kji', vj, dms).real / 2 / nk
    ek = np.einsum('kij,kji', vk, dms).real / 4 / nk
    assert ej > 0 and ek > 0
    df.reset()
