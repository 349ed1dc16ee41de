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
    panels) has hard regression bounds (|dE_K| <= 5e-5 Eh, measured 3.6e-5; max|dK| <= 1e-4) and the LITERAL 1e-6 Eh as an
    expected-failure test next to it: it is NOT reached at this size inside 30 s on one GPU (c = 15: 9.3e-6 Eh in 26.3 s);
  * SCF orbitals (configs[1] size): the (AO x occupied) pair space + robust K meets the literal 1e-6 Eh - asserted;
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


_CFG2 = {}


def test_config2_headline_diamond444_accuracy_vs_exact_exchange():
    """configs[2] (diamond 4x4x4, gth-dzvp, 120^3: N = 1664, G = 1 728 000) exactly as bench.py times it - refined selection,
    c = 18 (P = 29952), AO x AO pair space, block-Jacobi route, W in the spectral form (X X^T over the sphere inscribed in the
    reciprocal FFT box: 0.36 of the grid's terms, no panels) - against the exact exchange on the benchmark density (random
    orthogonal orbitals; 38 s on the GPU).  THE NORTH STAR'S LITERAL TOLERANCE: |dE_K| < 1e-6 Eh (measured +3.8e-7; c = 17 / 19
    read +1.6e-6 / -2.0e-6 - the plain K is not variational and its signed error scatters at that level),
    max|dK| <= 1e-5 (measured 4.4e-6); size-independent properties on top (symmetry, linearity, the route's probe check)."""
    import torch
    from pyscf_isdf_amd.isdf import ISDF
    if torch.cuda.get_device_properties(0).total_memory < 270 * 2 ** 30:
        pytest.skip('needs a 288 GB device')
    cell = workloads.make_cell('diamond-444-dzvp-120')
    dm, c, occ = workloads.make_dm(cell)
    df = ISDF(cell, c_isdf=18, select='refined')
    vj, vk = df.get_jk(dm)
    assert len(df.ip) == 29952 and len(np.unique(df.ip)) == 29952
    assert df.fit_route_used == 'blockjacobi' and df.n_panels == 1 and df.bj_check <= df.w_spectral_check_tol
    # the sphere is taken because this mesh resolves the AO pair products (share of their Coulomb energy outside: < 1e-11)
    assert 0.25 < df.w_spectral_fraction < 0.40 and df._sphere_share[1] < 1e-11
    assert abs(vj - vj.T).max() < 1e-8 and abs(vk - vk.T).max() < 1e-7
    vk_half = df.get_jk(-0.5 * dm, with_j=False)[1]
    assert abs(vk_half + 0.5 * vk).max() < 1e-9
    assert abs(np.einsum('ij,ji', vj, dm) / 2 - 12.140270972643) < 1e-7           # E_J: the exact formula, same as round 1
    df.release_fit_buffers()
    k_exact = df.get_k_exact(mo_coeff=c, mo_occ=occ)
    assert abs(_ek(k_exact, dm) - 123.18058922) < 1e-6                           # the exact exchange itself is stable
    _CFG2['dE_K'] = _ek(vk, dm) - _ek(k_exact, dm)
    _CFG2['k_exact'] = k_exact
    assert abs(_CFG2['dE_K']) < NORTH_STAR_TOL
    assert abs(vk - k_exact).max() < 1e-5
    df.reset()


def test_config2_fast_variant_c12_regression_bounds():
    """The fast variant of the same build (c = 12, P = 19968: 9.3 s per build + get_jk): hard regression bounds |dE_K| <= 5e-5 Eh
    (measured -3.60e-5 in rounds 2 and 3, classic and spectral form alike), max|dK| <= 1e-4 (6.1e-5), against the exact exchange
    of the test above (which has to run first; alone, this test is skipped)."""
    from pyscf_isdf_amd.isdf import ISDF
    if 'k_exact' not in _CFG2:
        pytest.skip('runs after test_config2_headline_diamond444_accuracy_vs_exact_exchange')
    cell = workloads.make_cell('diamond-444-dzvp-120')
    dm, c, occ = workloads.make_dm(cell)
    df = ISDF(cell, c_isdf=12, select='refined')
    vk = df.get_jk(dm, with_j=False)[1]
    k_exact = _CFG2.pop('k_exact')
    assert len(df.ip) == 19968 and df.n_panels == 1 and df.w_spectral_fraction is not None and df.bj_check <= df.w_spectral_check_tol
    assert abs(_ek(vk, dm) - _ek(k_exact, dm)) < 5e-5 and abs(vk - k_exact).max() < 1e-4
    df.reset()


def test_config1_scf_orbitals_occ_pair_space_and_robust_k_meet_the_north_star():
    """Physical orbitals (an RHF on diamond 2x2x2 / gth-dzvp / 80^3 converged with the ISDF object itself: get_pp, J, K from the
    device; S and T by plane-wave quadrature of the AO values - test-side plumbing) against the exact exchange AT those
    orbitals.  The AO x AO fit at c = 12 is off by 1e-4 Eh; the (AO x occupied) pair space (pair_space='occ': the reduction
    the reference's K makes with mo_coeff-tagged densities, fft_jk.py:206-210,235-238) brings the plain ISDF K to 2e-5 Eh
    at the same number of points, and with Dunlap's robust correction on top (error quadratic in the fit error) the LITERAL
    north-star 1e-6 Eh holds: asserted at c = 15 (at c = 12 the two SCF solutions measured so far gave 8.6e-8 and 1.5e-6 Eh
    with max|dK| 1.5e-8 / 7.8e-8: profiles/r03_scf_orbital_scan_diamond222.log, gpurun of this test)."""
    import torch
    from pyscf_isdf_amd.isdf import ISDF
    cell = workloads.make_cell('diamond-222-dzvp-80')
    nao, nocc = cell.nao_nr(), cell.nelectron // 2
    mesh = [int(x) for x in cell.mesh]
    G = int(np.prod(mesh))
    df = ISDF(cell, c_isdf=10, select='refined')
    df.collocate()
    F = torch.fft.fftn(df.ao.reshape(nao, *mesh), dim=(1, 2, 3)).reshape(nao, G)
    b = 2 * np.pi * np.linalg.inv(cell.lattice_vectors().T)
    fr = [np.fft.fftfreq(n, 1. / n) for n in mesh]
    Gv = (fr[0][:, None, None, None] * b[0] + fr[1][None, :, None, None] * b[1] + fr[2][None, None, :, None] * b[2]).reshape(-1, 3)
    g2 = df.backend.to_device(np.einsum('gi,gi->g', Gv, Gv))
    T = ((0.5 * cell.vol / G ** 2) * torch.matmul(F.conj() * g2, F.T).real).cpu().numpy()
    S = ((cell.vol / G ** 2) * torch.matmul(F.conj(), F.T).real).cpu().numpy()
    del F
    hcore = T + df.get_pp()
    import scf_helpers
    e_tot, dm = scf_helpers.rhf(hcore, S, lambda d: df.get_jk(d), nocc, 0.0, max_cycle=40, conv=1e-8)
    assert abs(np.einsum('ij,ji', dm, S) - 2 * nocc) < 1e-8
    k_exact = df.get_k_exact(dm)
    errs = {}
    for tag, space, robust, cc in (('ao', 'ao', False, 12), ('occ', 'occ', False, 12), ('occ+robust', 'occ', True, 15)):
        d2 = ISDF(cell, c_isdf=cc, select='refined')
        d2.pair_space, d2.robust_k = space, robust
        vk = d2.get_jk(dm, with_j=False)[1]
        errs[tag] = (abs(_ek(vk, dm) - _ek(k_exact, dm)), abs(vk - k_exact).max())
        print('diamond 2x2x2 SCF orbitals, c = %d, %s: |dE_K| %.2e Eh, max|dK| %.2e' % ((cc, tag) + errs[tag]))
        d2.reset()
    assert errs['occ'][1] < 0.2 * errs['ao'][1] and errs['occ'][0] < 3e-5
    assert errs['occ+robust'][0] < NORTH_STAR_TOL and errs['occ+robust'][1] < 1e-6


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
    (k_ip_factor): the default (2) and 4 are both measured (profiles/r03_kpoint_accuracy_mgo222.log: factor 2: dE_K -6.5e-6 Eh per
    cell, max|dK| 3.8e-5; factor 3: +1.8e-7 / 2.5e-6; factor 4: +9.4e-8 / 3.1e-7 - with the Nyquist-plane correction of the +-q
    pairing, which this comparison is what found: before it the matrix error stalled at 7e-6).  Asserted: |dE_K| per cell <= 2e-5 Eh
    and max|dK| <= 1e-4 at the default; the LITERAL north-star 1e-6 Eh per cell and max|dK| <= 1e-6 at factor 4."""
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
    assert errs[2][0] < 2e-5 and errs[2][1] < 1e-4
    assert errs[4][0] < NORTH_STAR_TOL and errs[4][1] < 1e-6


def test_config3_mgo333_kmesh_full_size_properties_and_sampled_exact_rows():
    """configs[3] at FULL size on one GPU: MgO 3x3x3 (54 atoms, N = 729), gth-dzvp, 96^3, 2x2x2 k-mesh, refined selection, c = 10
    with the default k_ip_factor = 2 (P = 14580, 14 W^q) - about two minutes.  Asserted: shapes, Hermiticity of J and K at every
    k, linearity, and the ISDF K against the reference's exact k-point exchange on a SAMPLE of 8 AO rows of every k-point (the
    full exact exchange is 64 x 729 x 216 complex FFT pairs of 96^3: an hour; the sample takes seconds): max|dK| on the rows
    <= 2e-4 (measured value in profiles/r03_kpoint_accuracy_mgo333_rows.log)."""
    import torch
    from pyscf_isdf_amd.isdf import ISDF
    if torch.cuda.get_device_properties(0).total_memory < 270 * 2 ** 30:
        pytest.skip('needs a 288 GB device')
    name = 'mgo-333-dzvp-k222'
    cell, kpts, dms, cs, occs = _mgo_density(name)
    nk, nao = len(kpts), cell.nao_nr()
    assert nao == 729 and nk == 8
    df = ISDF(cell, kpts=kpts, c_isdf=10, select='refined')
    rows = (360, 8)
    vk_rows = df.get_k_exact(dms, mo_coeff=cs, mo_occ=occs, rows=rows)
    vj, vk = df.get_jk(dms, kpts=kpts)
    assert vj.shape == vk.shape == (nk, nao, nao) and len(df.ip) == 14580
    assert abs(vj - vj.conj().transpose(0, 2, 1)).max() < 1e-9 and abs(vk - vk.conj().transpose(0, 2, 1)).max() < 1e-8 * abs(vk).max()
    err = abs(vk[:, rows[0]:rows[0] + rows[1]] - vk_rows).max()
    print('MgO 3x3x3 k222 c=10 P=%d: max|dK| on %d sampled rows per k-point %.2e (max|K| there %.3f)' % (len(df.ip), rows[1], err, abs(vk_rows).max()))
    assert err < 2e-4
    vk2 = df.get_jk(-0.5 * dms, kpts=kpts, with_j=False)[1]
    assert abs(vk2 + 0.5 * vk).max() < 1e-9 * abs(vk).max()
    df.reset()


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
