"""Multigrid J / LDA potential.

CPU half (-m "not gpu"): the oracle restatement of pyscf/pbc/dft/multigrid/multigrid.py on the reference's own task ladder
against the oracle FFTDF J (the reference's acceptance test, test_multigrid.py:112-131: |ref - out| < 1e-8) on the
reference's two test cells (test_multigrid.py:32-50; gth-dzvp instead of gth-dzv, which is not bundled), the product's planner,
and the product's orchestration on the CPU checker backend.  GPU half: the device path against the oracle on the same ladder
(1e-9) and against the FFTDF J of the device (1e-8)."""
import copy
import numpy as np
import pytest
from pyscf_isdf_amd import gto
from pyscf_isdf_amd import multigrid as pmg
from oracle import multigrid as omg, fftdf as offt, ao as oao


def cell_c2_orth():
    return gto.Cell(a=np.eye(3) * 3.5668, atom='C 0 0 0; C 1.8 1.8 1.8', basis='gth-dzvp', pseudo='gth-pade', precision=1e-9,
                    mesh=[48] * 3)


def cell_c2_nonorth():
    rng = np.random.default_rng(5)
    return gto.Cell(a=np.eye(3) * 3.5668 + rng.random((3, 3)), atom='C 0 0 0; C 0.8917 0.8917 0.8917', basis='gth-dzvp',
                    pseudo='gth-pade', precision=1e-9, mesh=[44, 43, 42])


def cell_he():
    # test_multigrid.py:52-58: shells that split between levels inside one contraction
    return gto.Cell(atom='He 0 0 0', basis=[[0, (1, 1, .1), (.5, .1, 1)], [1, (.8, 1)]], unit='B', precision=1e-9, mesh=[18] * 3,
                    a=np.eye(3) * 5)


def cell_he_split():
    # a sharp and a smooth primitive inside the same two contractions: the contracted functions themselves split between levels
    return gto.Cell(atom='He 0 0 0; He 2.2 2.4 2.1', basis=[[0, (6., 1, .1), (.4, .1, 1)], [1, (.8, 1)], [2, (1.1, 1)]], unit='B',
                    precision=1e-9, mesh=[30, 32, 30], a=np.eye(3) * 5 + np.array([[0, .3, 0], [0, 0, 0], [.2, 0, 0]]))


def lattice_fn_for(cell):
    def fn(bas, env):
        sub = copy.copy(cell)
        sub._bas, sub._env = bas, env
        sub._rcut = gto.estimate_rcut(sub, cell.precision)
        rcut = gto.estimate_rcut_per_shell(sub)
        return gto.get_lattice_Ls(sub, rcut=rcut.max()), rcut
    return fn


def dense_ao(cell):
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    return oao.eval_ao(cell._atm, cell._bas, cell._env, cell.get_uniform_grids(), Ls, rcut, rule='point')


def make_dm(cell, seed=2):
    nao = cell.nao_nr()
    dm = np.random.default_rng(seed).random((nao, nao)) * .2 + np.eye(nao)
    return dm + dm.T


def as_tasks(levels):
    return [dict(mesh=l.mesh, bas=l.bas, env=l.env, nH=l.nH, idx_h=l.idx_h, idx_l=l.idx_l, Ls=l.Ls, rcut=l.rcut) for l in levels]


@pytest.mark.parametrize('mk', [cell_c2_orth, cell_c2_nonorth, cell_he, cell_he_split])
def test_oracle_reference_ladder_reproduces_fftdf_j(mk):
    """The reference's own criterion on its own ladder (ratio 1.3, odd meshes from 12^3 / 32^3): 1e-8."""
    cell = mk()
    dm = make_dm(cell)
    a, mesh = cell.lattice_vectors(), cell.mesh
    tasks = omg.reference_tasks(cell._bas, cell._env, a, mesh, cell.precision, lattice_fn_for(cell))
    assert (np.asarray(tasks[-1]['mesh']) == mesh).all()
    ref = offt.get_j(dense_ao(cell), dm, a, mesh)
    assert abs(omg.get_j(tasks, cell._atm, dm, a, mesh) - ref).max() < 1e-8


def test_planner_levels_partition_the_primitives():
    """Every primitive is dense in exactly one level and sparse in every level above it; meshes descend from the dense mesh,
    factor into 2, 3, 5, 7 and never drop below the floor; contractions without a primitive in a window are dropped."""
    cell = cell_c2_orth()
    levels = pmg.multi_grids_tasks(cell, cell.mesh, split='all')
    assert (levels[0].mesh == cell.mesh).all() and len(levels) >= 2
    for hi, lo in zip(levels[:-1], levels[1:]):
        assert (lo.mesh <= hi.mesh).all() and (lo.mesh < hi.mesh).any() and hi.ke_window[0] == lo.ke_window[1]
    nprim_total = sum(cell.bas_nprim(i) for i in range(cell.nbas))
    NPRIM_OF, ANG_OF, NCTR_OF = 2, 1, 3
    dense = 0
    for lv in levels:
        assert lv.mesh.min() >= pmg.MIN_LEVEL_MESH and all(pmg._fft_friendly(int(n)) == n for n in lv.mesh[lv.mesh < cell.mesh])
        nsh_h = 0
        nfun = 0
        for row in lv.bas:                                   # dense rows come first: count rows until nH functions are covered
            if nfun >= lv.nH:
                break
            nfun += (2 * row[ANG_OF] + 1) * row[NCTR_OF]
            dense += row[NPRIM_OF]
            nsh_h += 1
        assert nfun == lv.nH and len(lv.idx_h) == lv.nH and len(set(lv.idx_h)) == lv.nH
        assert len(lv.rcut) == len(lv.bas)
    assert dense == nprim_total
    assert len(levels[-1].idx_l) == 0
    # gth-dzvp carbon: the second s / p contraction is one diffuse primitive - absent from the sharp levels
    assert levels[0].nH < cell.nao_nr()
    # the default cost model keeps a cell this small on one mesh and splits the headline cell (planning only)
    assert len(pmg.multi_grids_tasks(cell, cell.mesh)) == 1
    big = gto.diamond_supercell(4, mesh=(120,) * 3)
    lv = pmg.multi_grids_tasks(big, big.mesh)
    assert len(lv) >= 2 and sum(pmg._level_cost(l.ngrids, l.nH, l.nT) for l in lv) < pmg._level_cost(120 ** 3, big.nao_nr(), big.nao_nr())
    assert len(pmg.multi_grids_tasks(big, big.mesh, max_levels=2)) <= 2


@pytest.mark.parametrize('mk', [cell_c2_orth, cell_c2_nonorth, cell_he, cell_he_split])
def test_oracle_on_product_ladder_reproduces_fftdf_j_and_lda(mk):
    """The product's coarse ladder (ratio 3, FFT-friendly even/odd meshes) through the oracle: J to 1e-8 of the FFTDF J, the
    LDA energy / potential to 1e-7 of the dense-grid numbers (test_multigrid.py:133-142)."""
    cell = mk()
    dm = make_dm(cell)
    a, mesh = cell.lattice_vectors(), cell.mesh
    tasks = as_tasks(pmg.multi_grids_tasks(cell, mesh, split='all'))
    aoR = dense_ao(cell)
    assert abs(omg.get_j(tasks, cell._atm, dm, a, mesh) - offt.get_j(aoR, dm, a, mesh)).max() < 1e-8
    n, e, v, ecoul = omg.nr_rks_lda(tasks, cell._atm, dm, a, mesh)
    n0, e0, v0 = omg.nr_rks_lda_dense(aoR, dm, a, mesh)
    assert abs(n - n0) < 1e-7 and abs(e - e0) < 1e-7 and abs(v - v0).max() < 1e-7
    vj = offt.get_j(aoR, dm, a, mesh)
    assert abs(ecoul - 0.5 * np.einsum('ij,ji', vj, 0.5 * (dm + dm.T))) < 1e-7


def _check_product_against_oracle(df, cell, tol):
    dm = make_dm(cell)
    a, mesh = cell.lattice_vectors(), cell.mesh
    dms = np.stack([dm, make_dm(cell, seed=7)[::-1, ::-1].copy()])
    vj = df.get_jk(dms, with_k=False)[0]
    assert vj.shape == dms.shape
    tasks = as_tasks(df.tasks)
    ref = omg.get_j(tasks, cell._atm, dms, a, mesh)
    assert abs(vj - ref).max() < tol
    assert abs(df.get_jk(dm, with_k=False)[0] - ref[0]).max() < tol
    rho = df.get_rho(dm)
    assert abs(rho - omg.get_rho(tasks, cell._atm, dm, a, mesh)[0]).max() < tol * 10
    n, e, veff = pmg.nr_rks(df, 'lda,', dm, with_j=True)
    n0, e0, v0, ec0 = omg.nr_rks_lda(tasks, cell._atm, dm, a, mesh, with_j=True)
    assert abs(n - n0) < tol * 100 and abs(e - e0) < tol * 100 and abs(veff - v0).max() < tol * 10
    assert abs(veff.ecoul - ec0) < 1e-7 and veff.exc == e and veff.vj is None
    n, e, veff = pmg.nr_rks(df, 'LDA,', dm, return_j=True)
    assert abs(veff - omg.nr_rks_lda(tasks, cell._atm, dm, a, mesh)[2]).max() < tol * 10
    assert abs(veff.vj - ref[0]).max() < tol
    with pytest.raises(NotImplementedError):
        pmg.nr_rks(df, 'pbe,pbe', dm)
    # open shell: (alpha, beta) pair, spin-scaled Slater exchange, Coulomb potential of the total density
    n, e, veff = pmg.nr_uks(df, 'lda,', dms, with_j=True, return_j=True)
    n0, e0, v0, ec0 = omg.nr_uks_lda(tasks, cell._atm, dms, a, mesh, with_j=True)
    assert abs(n - n0) < tol * 100 and abs(e - e0) < tol * 100 and abs(veff - v0).max() < tol * 10 and veff.shape == dms.shape
    assert abs(veff.ecoul - ec0) < 1e-7 and abs(veff.vj - ref.sum(axis=0)).max() < tol * 2
    # a closed shell split into equal halves gives the restricted numbers
    nr, er, vr = pmg.nr_rks(df, 'lda,', dm)
    nu, eu, vu = pmg.nr_uks(df, 'lda,', np.stack([dm, dm]) * .5)
    assert abs(nr - nu) < 1e-9 and abs(er - eu) < 1e-9 and abs(vu[0] - vr).max() < 1e-9 and abs(vu[1] - vr).max() < 1e-9
    return vj, dms


@pytest.mark.parametrize('mk', [cell_he_split, cell_he])
def test_product_orchestration_on_checker_backend(mk):
    """pyscf_isdf_amd.multigrid end to end on tests/oracle_backend.py (half spectra by numpy rfftn) against the full-spectrum
    oracle on the same ladder, and the FFTDF J."""
    from oracle_backend import OracleBackend
    cell = mk()
    df = pmg.MultiGridFFTDF(cell, backend=OracleBackend())
    df.split = 'all'                                         # one level per distinct mesh: small cells get several levels
    vj, dms = _check_product_against_oracle(df, cell, 1e-10)
    assert abs(vj[0] - offt.get_j(dense_ao(cell), dms[0], cell.lattice_vectors(), cell.mesh)).max() < 1e-8
    assert not df._built                                     # J never triggered the ISDF fit
    df.reset()
    assert df.tasks is None


def make_kpts_dms(cell, hermitian=True, seed=3):
    """Two k-points k, -k as in the reference's tests (test_multigrid.py:60-67) and complex density matrices on them."""
    rng = np.random.default_rng(seed)
    k0 = rng.random(3) * 0.4
    kpts = np.array([k0, -k0])
    nao = cell.nao_nr()
    dms = rng.random((2, nao, nao)) * .2 + 1j * (rng.random((2, nao, nao)) - .5) * .1
    if hermitian:
        dms = dms + dms.conj().transpose(0, 2, 1) + np.eye(nao)
    return kpts, dms


def dense_ao_kpts(cell, kpts):
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    return oao.eval_ao(cell._atm, cell._bas, cell._env, cell.get_uniform_grids(), Ls, rcut, kpts=kpts, rule='point')


def _check_kpts_against_oracle(df, cell, tol):
    a, mesh = cell.lattice_vectors(), cell.mesh
    kpts, dms = make_kpts_dms(cell)
    vj = df.get_jk(dms, kpts=kpts, with_k=False)[0]
    assert vj.shape == dms.shape and np.iscomplexobj(vj)
    tasks = as_tasks(df.tasks)
    ref = omg.get_j_kpts(tasks, cell._atm, dms, a, mesh, kpts)
    assert abs(vj - ref).max() < tol
    assert abs(vj - vj.conj().transpose(0, 2, 1)).max() < tol           # Hermitian density -> Hermitian J
    # the reference's criterion: the FFTDF J at the same k-points (test_multigrid.py:112-125), 1e-8
    fft_ref = offt.get_jk_kpts(dense_ao_kpts(cell, kpts), dms, a, mesh, cell.get_uniform_grids(), kpts)[0]
    assert abs(vj - fft_ref).max() < 1e-8
    # band k-points, one (3,) vector and a list; a non-Hermitian density matrix (complex density, two real passes)
    band = np.array([[0.1, -0.05, 0.2], [0., 0., 0.]])
    vb = pmg.get_j_kpts(df, dms, kpts=kpts, kpts_band=band)
    assert vb.shape == (2,) + dms.shape[1:]
    assert abs(vb - omg.get_j_kpts(tasks, cell._atm, dms, a, mesh, kpts, band)).max() < tol
    assert abs(df.get_jk(dms, kpts=kpts, kpts_band=band[0], with_k=False)[0] - vb[0]).max() < 1e-12
    kpts2, dms_nh = make_kpts_dms(cell, hermitian=False)
    vnh = df.get_jk(np.stack([dms, dms_nh]), kpts=kpts, with_k=False)[0]
    assert vnh.shape == (2,) + dms.shape and abs(vnh[0] - vj).max() < 1e-12
    assert abs(vnh[1] - omg.get_j_kpts(tasks, cell._atm, dms_nh, a, mesh, kpts)).max() < tol * 10
    # LDA at k-points
    n, e, veff = pmg.nr_rks(df, 'lda,', dms, kpts=kpts, with_j=True)
    n0, e0, v0 = omg.nr_rks_lda_kpts(tasks, cell._atm, dms, a, mesh, kpts, with_j=True)
    assert abs(n - n0) < tol * 100 and abs(e - e0) < tol * 100 and abs(veff - v0).max() < tol * 10 and veff.shape == dms.shape
    pair = np.stack([dms, dms[::-1] * .5])
    n, e, veff = pmg.nr_uks(df, 'lda,', pair, kpts=kpts, with_j=True)
    n0, e0, v0, ec0 = omg.nr_uks_lda(tasks, cell._atm, pair, a, mesh, with_j=True, kpts=kpts)
    assert abs(n - n0) < tol * 100 and abs(e - e0) < tol * 100 and abs(veff - v0).max() < tol * 10 and veff.shape == pair.shape
    assert abs(veff.ecoul - ec0) < 1e-7


def _check_response(df, cell, tol):
    """nr_rks_fxc / nr_rks_fxc_st / nr_uks_fxc / cache_xc_kernel1 against the oracle, and against what they are by definition:
    the derivative of nr_rks's (nr_uks's) potential matrix along the response density (central difference)."""
    a, mesh = cell.lattice_vectors(), cell.mesh
    dm0 = make_dm(cell)
    rng = np.random.default_rng(9)
    nao = cell.nao_nr()
    dm1 = rng.standard_normal((2, nao, nao)) * 0.05                      # not symmetric: hermi = 0 input
    tasks = as_tasks(df.tasks) if df.tasks is not None else as_tasks(df.build_tasks())
    v = pmg.nr_rks_fxc(df, 'lda,', dm0, dm1, with_j=True)
    assert v.shape == dm1.shape
    assert abs(v - omg.nr_fxc_lda(tasks, cell._atm, dm0, dm1, a, mesh, 'rks', with_j=True)).max() < tol
    eps = 1e-4
    sym = 0.5 * (dm1[0] + dm1[0].T)
    fd = (pmg.nr_rks(df, 'lda,', dm0 + eps * sym, with_j=True)[2] - pmg.nr_rks(df, 'lda,', dm0 - eps * sym, with_j=True)[2]) / (2 * eps)
    assert abs(v[0] - fd).max() < 1e-6 * max(1.0, abs(v[0]).max())
    rho, vxc, fxc = pmg.cache_xc_kernel1(df, 'lda,', dm0)
    assert rho.shape == (int(np.prod(mesh)),) and vxc.shape == (1, rho.size) and fxc.shape == (1, 1, rho.size)
    assert abs(pmg.nr_rks_fxc(df, 'lda,', dm0, dm1, with_j=True, rho0=rho, fxc=fxc) - v).max() < 1e-12
    vs = pmg.nr_rks_fxc_st(df, 'lda,', dm0, dm1, singlet=True)
    assert abs(vs - omg.nr_fxc_lda(tasks, cell._atm, dm0, dm1, a, mesh, 'st')).max() < tol
    assert abs(vs - pmg.nr_rks_fxc_st(df, 'lda,', dm0, dm1, singlet=False)).max() < 1e-12
    assert abs(vs - 2 * pmg.nr_rks_fxc(df, 'lda,', dm0, dm1)).max() < 1e-10
    pair0 = np.stack([dm0 * .6, dm0 * .4])
    pair1 = np.stack([dm1[0], dm1[1], dm1[1] * .5, dm1[0] * -.3])       # (alpha responses 1, 2, beta responses 1, 2)
    vu = pmg.nr_uks_fxc(df, 'lda,', pair0, pair1, with_j=True)
    assert vu.shape == pair1.shape
    assert abs(vu - omg.nr_fxc_lda(tasks, cell._atm, pair0, pair1, a, mesh, 'uks', with_j=True)).max() < tol
    d = np.stack([0.5 * (pair1[0] + pair1[0].T), 0.5 * (pair1[2] + pair1[2].T)])
    fdu = (pmg.nr_uks(df, 'lda,', pair0 + eps * d, with_j=True)[2] - pmg.nr_uks(df, 'lda,', pair0 - eps * d, with_j=True)[2]) / (2 * eps)
    assert abs(vu[[0, 2]] - fdu).max() < 1e-6 * max(1.0, abs(vu).max())
    r2, v2, f2 = pmg.cache_xc_kernel1(df, 'lda,', pair0, spin=1)
    assert r2.shape == (2, rho.size) and v2.shape == (2, 1, rho.size) and f2.shape == (2, 1, 2, 1, rho.size)
    assert abs(f2[0, 0, 1, 0]).max() == 0 and abs(r2.sum(axis=0) - rho).max() < 1e-10


def test_product_response_on_checker_backend():
    from oracle_backend import OracleBackend
    cell = cell_he_split()
    df = pmg.MultiGridFFTDF(cell, backend=OracleBackend())
    df.split = 'all'
    _check_response(df, cell, 1e-10)
    # k-points: Hermitian and non-Hermitian response matrices
    a, mesh = cell.lattice_vectors(), cell.mesh
    kpts, dm0 = make_kpts_dms(cell)
    _, dm1 = make_kpts_dms(cell, hermitian=False, seed=11)
    tasks = as_tasks(df.tasks)
    v = pmg.nr_rks_fxc(df, 'lda,', dm0, dm1[None], with_j=True, kpts=kpts)
    assert v.shape == (1,) + dm1.shape
    assert abs(v - omg.nr_fxc_lda(tasks, cell._atm, dm0, dm1[None], a, mesh, 'rks', with_j=True, kpts=kpts)).max() < 1e-9


@pytest.mark.gpu
def test_gpu_multigrid_response_functions():
    cell = cell_he_split()
    df = pmg.MultiGridFFTDF(cell)
    df.split = 'all'
    _check_response(df, cell, 1e-9)


def dense_ao4(cell):
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    return oao.eval_ao_deriv1(cell._atm, cell._bas, cell._env, cell.get_uniform_grids(), Ls, rcut)


def test_b88_closed_form_derivatives():
    """oracle.multigrid.b88_exchange: vrho and w = de/d(grad rho) against central differences of the energy density; the
    uniform-gas limit is the Slater exchange."""
    rng = np.random.default_rng(0)
    rho = rng.random(500) * 2 + 1e-3
    grad = rng.standard_normal((3, 500)) * rho ** 1.2
    g = np.sqrt((grad ** 2).sum(axis=0))
    exc, vrho, w = omg.b88_exchange(rho, grad)
    h = 1e-6
    assert abs((omg.b88_energy_density(rho * (1 + h), g) - omg.b88_energy_density(rho * (1 - h), g)) / (2 * h * rho) - vrho).max() < 1e-8
    fd = (omg.b88_energy_density(rho, g * (1 + h)) - omg.b88_energy_density(rho, g * (1 - h))) / (2 * h * g)
    assert abs(fd[None] * grad / g - w).max() < 1e-8
    assert abs(omg.b88_exchange(np.array([1.3]), np.zeros((3, 1)))[0][0] - omg.slater_exchange(np.array([1.3]))[0][0]) < 1e-14


def test_oracle_scf_reproduces_the_reference_lda_and_b88_energies():
    """All-oracle SCF on the diamond primitive cell of pyscf/pbc/scf/test/test_newton.py:25-44 (gth-szv / gth-pade, 19^3): RKS
    'lda,' -9.7670882971475663 and RKS 'b88,' -9.9355341416893559 (both places=8 there).  Pins the closed-form functionals
    (libxc is absent) and the dense-grid GGA quadrature the ladder is compared with."""
    import scf_helpers
    from oracle import pp as opp
    cell = gto.Cell(unit='B', atom='C 0. 0. 0.; C 1.68506879 1.68506879 1.68506879',
                    a=[[0., 3.37013758, 3.37013758], [3.37013758, 0., 3.37013758], [3.37013758, 3.37013758, 0.]],
                    basis='gth-szv', pseudo='gth-pade', mesh=[19] * 3)
    a, mesh = cell.lattice_vectors(), cell.mesh
    S, T = scf_helpers.overlap_kinetic_from_ft(cell)
    ao4 = dense_ao4(cell)
    ps = [cell._pseudo.get(cell.atom_symbol(i)) for i in range(cell.natm)]
    vpp = opp.get_pp(cell._atm, cell._bas, cell._env, cell.atom_coords(), cell.atom_charges(), ps, a, mesh,
                     cell.get_uniform_grids(), [ao4[0]], np.zeros((1, 3)))[0].real
    e_nuc = scf_helpers.ewald_energy(cell)

    def make(xc):
        def veff(dm):
            vj = offt.get_j(ao4[0], dm, a, mesh)
            n, exc, vxc = xc(dm)
            return vj + vxc, 0.5 * np.einsum('ij,ji', vj, dm), exc
        return veff
    e_b88 = scf_helpers.rks(T + vpp, S, make(lambda dm: omg.nr_rks_b88_dense(ao4, dm, a, mesh)), 4, e_nuc)[0]
    assert abs(e_b88 - (-9.9355341416893559)) < 5e-8
    e_lda = scf_helpers.rks(T + vpp, S, make(lambda dm: omg.nr_rks_lda_dense(ao4[0], dm, a, mesh)), 4, e_nuc)[0]
    assert abs(e_lda - (-9.7670882971475663)) < 5e-8


def _check_gga(df, cell, tol):
    """nr_rks('b88,') of the product against the oracle on the same ladder, against the dense-grid quadrature (1e-7, the
    reference's own criterion for its GGA: test_multigrid.py:216-226) and against its definition (veff = dE_xc/dD)."""
    a, mesh = cell.lattice_vectors(), cell.mesh
    dm = make_dm(cell)
    n, e, veff = pmg.nr_rks(df, 'b88,', dm, with_j=True, return_j=True)
    tasks = as_tasks(df.tasks)
    n0, e0, v0, ec0 = omg.nr_rks_b88(tasks, cell._atm, dm, a, mesh, with_j=True)
    assert abs(n - n0) < tol * 100 and abs(e - e0) < tol * 100 and abs(veff - v0).max() < tol * 10 and abs(veff.ecoul - ec0) < 1e-7
    vj = df.get_jk(dm, with_k=False)[0]
    assert abs(veff.vj - vj).max() < 1e-9
    nd, ed, vd = omg.nr_rks_b88_dense(dense_ao4(cell), dm, a, mesh)
    vxc = pmg.nr_rks(df, 'B88,', dm)[2]
    assert abs(e - ed) < 1e-7 and abs(vxc - vd).max() < 1e-7 and abs(veff - vj - vxc).max() < 1e-9
    # open shell: the spin-scaling route of the product against the oracle's spin channels in their own variables
    pair = np.stack([dm * .6, make_dm(cell, seed=8) * .4])
    nu, eu, vu = pmg.nr_uks(df, 'b88,', pair, with_j=True)
    n0, e0, v0, ec0 = omg.nr_uks_b88(tasks, cell._atm, pair, a, mesh, with_j=True)
    assert vu.shape == pair.shape and abs(nu - n0) < tol * 100 and abs(eu - e0) < tol * 100 and abs(vu - v0).max() < tol * 10
    assert abs(vu.ecoul - ec0) < 1e-7
    nu, eu, vu = pmg.nr_uks(df, 'b88,', np.stack([dm, dm]) * .5)
    assert abs(nu - n) < 1e-9 and abs(eu - e) < 1e-9 and abs(vu[0] - vxc).max() < 1e-9 and abs(vu[1] - vxc).max() < 1e-9
    rng = np.random.default_rng(4)
    d1 = rng.standard_normal(dm.shape) * 0.05
    d1 = d1 + d1.T
    eps = 1e-4
    fd = (pmg.nr_rks(df, 'b88,', dm + eps * d1)[1] - pmg.nr_rks(df, 'b88,', dm - eps * d1)[1]) / (2 * eps)
    assert abs(fd - np.einsum('ij,ji', vxc, d1)) < 1e-6 * max(1.0, abs(fd))


def test_product_gga_on_checker_backend():
    from oracle_backend import OracleBackend
    cell = cell_he_split()
    df = pmg.MultiGridFFTDF(cell, backend=OracleBackend())
    df.split = 'all'
    _check_gga(df, cell, 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize('mk', [cell_he_split, cell_c2_orth])
def test_gpu_multigrid_gga_b88(mk):
    cell = mk()
    df = pmg.MultiGridFFTDF(cell)
    df.split = 'all'
    _check_gga(df, cell, 1e-9)


def test_product_kpts_on_checker_backend():
    """k-point J / LDA of pyscf_isdf_amd.multigrid on the CPU checker backend: stacked real / imaginary planes against the
    oracle's complex arithmetic on the same ladder, and the oracle's FFTDF J at the k-points."""
    from oracle_backend import OracleBackend
    cell = cell_he_split()
    df = pmg.MultiGridFFTDF(cell, backend=OracleBackend())
    df.split = 'all'
    _check_kpts_against_oracle(df, cell, 1e-10)
    assert not df._built


@pytest.mark.gpu
@pytest.mark.parametrize('mk', [cell_he_split, cell_c2_nonorth])
def test_gpu_multigrid_kpts_match_oracle_and_fftdf(mk):
    """The device's k-point J / LDA through the ladder (eval_ao_k on the level cells, isdf_rho_pair and isdf_gemm_nt on stacked
    planes) against the oracle on the same ladder (1e-9) and the oracle's FFTDF J (1e-8)."""
    cell = mk()
    df = pmg.MultiGridFFTDF(cell)
    df.split = 'all'
    _check_kpts_against_oracle(df, cell, 1e-9)
    assert not df._built


@pytest.mark.gpu
@pytest.mark.parametrize('mk', [cell_c2_orth, cell_c2_nonorth, cell_he, cell_he_split])
def test_gpu_multigrid_j_rho_lda_match_oracle_and_fftdf(mk):
    """The device path (multigrid.hip + eval_ao.hip + gemm_f64.hip through the C ABI) against the oracle on the same ladder
    (1e-9) and against the device's own FFTDF-formula J (1e-8, the reference's criterion)."""
    from pyscf_isdf_amd import ISDF
    cell = mk()
    df = pmg.MultiGridFFTDF(cell)
    df.split = 'all'
    vj, dms = _check_product_against_oracle(df, cell, 1e-9)
    assert not df._built
    plain = ISDF(cell, c_isdf=4, select='global')
    ref = plain.get_jk(dms, with_k=False)[0]
    assert abs(vj - ref).max() < 1e-8


@pytest.mark.gpu
def test_gpu_uniform_grid_matches_cell_grid():
    """isdf_uniform_grid: order and folding of cell.get_uniform_grids (cell.py:874-898), even / odd and unequal meshes."""
    from pyscf_isdf_amd.backend import HipBackend
    be = HipBackend(0)
    cell = cell_c2_nonorth()
    for mesh in ([44, 43, 42], [5, 8, 3], [1, 2, 7]):
        got = be.to_host(be.uniform_grid(mesh, cell.lattice_vectors()))
        ref = cell.get_uniform_grids(mesh)
        assert got.shape == (3, len(ref)) and abs(got.T - ref).max() < 1e-14


@pytest.mark.gpu
def test_gpu_multigrid_pairs_with_isdf_k():
    """get_jk(with_k=True): J from the ladder, K from the parent's interpolation - the same K as the plain ISDF object."""
    from pyscf_isdf_amd import ISDF
    cell = cell_c2_orth()
    dm = make_dm(cell)
    df = pmg.MultiGridFFTDF(cell, c_isdf=8, select='global')
    df.split = 'all'
    vj, vk = df.get_jk(dm)
    plain = ISDF(cell, c_isdf=8, select='global')
    vj0, vk0 = plain.get_jk(dm)
    assert abs(vj - vj0).max() < 1e-8 and abs(vk - vk0).max() < 1e-9


def reference_gga_kpts_case():
    """Cell, k-points and density matrices of test_multigrid.py:28-67,216-227 (numpy.random.seed(2) sequence)."""
    np.random.seed(2)
    np.random.random((3, 3))                                   # the reference draws its non-orthogonal lattice first
    kpts = np.random.random((2, 3))
    kpts[1] = -kpts[0]
    dzvp = gto.Cell(a=np.eye(3) * 3.5668, atom='C 0 0 0; C 1.8 1.8 1.8', basis='gth-dzvp', pseudo='gth-pade', precision=1e-9,
                    mesh=[48] * 3)
    # gth-dzv is gth-dzvp without the polarisation shell (pyscf/pbc/gto/basis/gth-dzv.dat:40-46 vs gth-dzvp.dat:60-68)
    cell = gto.Cell(a=np.eye(3) * 3.5668, atom='C 0 0 0; C 1.8 1.8 1.8', basis={'C': [b for b in dzvp._basis['C'] if b[0] < 2]},
                    pseudo='gth-pade', precision=1e-9, mesh=[48] * 3)
    nao = cell.nao_nr()
    dm = np.random.random((2, nao, nao)) * .2
    dm1 = dm + np.eye(nao)
    return cell, kpts, dm1 + dm1.transpose(0, 2, 1)


def _check_gga_kpts(df, cell, kpts, dms, tol):
    a, mesh = cell.lattice_vectors(), cell.mesh
    n, e, veff = pmg.nr_rks(df, 'b88,', dms, kpts=kpts, with_j=True)
    tasks = as_tasks(df.tasks)
    n0, e0, v0 = omg.nr_rks_b88_kpts(tasks, cell._atm, dms, a, mesh, kpts, with_j=True)
    assert veff.shape == dms.shape and abs(n - n0) < tol * 100 and abs(e - e0) < tol * 100 and abs(veff - v0).max() < tol * 10
    assert abs(veff - veff.conj().transpose(0, 2, 1)).max() < tol * 10
    return veff


def test_product_gga_kpts_on_checker_backend():
    from oracle_backend import OracleBackend
    cell = cell_he_split()
    kpts, dms = make_kpts_dms(cell)
    df = pmg.MultiGridFFTDF(cell, backend=OracleBackend())
    df.split = 'all'
    _check_gga_kpts(df, cell, kpts, dms, 1e-10)


@pytest.mark.gpu
def test_gpu_multigrid_gga_kpts_and_the_reference_constant():
    """k-point 'b88,' on the device against the oracle on the same ladder (He cell with d functions, 1e-9), and the reference's own
    case (test_multigrid.py:216-227): fp(nr_rks(..., with_j=True)) = -0.05697304864467462+0.6990367789096609j, which the reference
    asserts to 7 places for its dense-grid answer and to 1e-7 between that and its multigrid."""
    from oracle import pbc_tools as otools
    cell = cell_he_split()
    kpts, dms = make_kpts_dms(cell)
    df = pmg.MultiGridFFTDF(cell)
    df.split = 'all'
    _check_gga_kpts(df, cell, kpts, dms, 1e-9)
    cell, kpts, dms = reference_gga_kpts_case()
    df = pmg.MultiGridFFTDF(cell)
    df.split = 'all'
    n, e, veff = pmg.nr_rks(df, 'b88,', dms, kpts=kpts, with_j=True)
    assert len(df.tasks) > 1
    assert abs(otools.fp(veff) - (-0.05697304864467462 + 0.6990367789096609j)) < 2e-7


def test_oracle_kpoint_b88_potential_reproduces_the_reference_constant():
    """The reference's only stored constant for a multigrid XC POTENTIAL (test_multigrid.py:216-227): C2 in the cubic 3.5668 A cell,
    gth-dzv / gth-pade, 48^3, two k-points k, -k and density matrices drawn after numpy.random.seed(2):
    fp(vxc['b88,'] + vj) = -0.05697304864467462+0.6990367789096609j (places=7), computed there by KNumInt + FFTDF.  The oracle's
    dense-grid k-point B88 quadrature + FFTDF J reproduce it: the closed-form Becke POTENTIAL (vrho, de/d grad rho) is pinned to
    libxc's, and through the oracle comparisons (1e-9) so is the device's Gamma-point GGA potential."""
    from oracle import pbc_tools as otools
    cell, kpts, dm = reference_gga_kpts_case()
    assert cell.nao_nr() == 16
    a, mesh = cell.lattice_vectors(), cell.mesh
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    coords = cell.get_uniform_grids()
    ao4 = [oao.eval_ao_deriv1(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpt=k) for k in kpts]
    n, exc, vxc = omg.nr_rks_b88_dense_kpts(ao4, dm, a)
    vj = offt.get_jk_kpts([x[0] for x in ao4], dm, a, mesh, coords, kpts)[0]
    assert abs(otools.fp(vxc + vj) - (-0.05697304864467462 + 0.6990367789096609j)) < 1e-7


def _reference_embed_then_real_field(field, n, N):
    """The reference's embedding of a level field (mesh n) into the dense mesh N: level spectrum at the numpy.fft.fftfreq indices
    of the dense mesh - Nyquist entries of an even level mesh at -n/2 only (multigrid.py:669-673) - and the REAL part of the
    inverse transform, which is all the reference ever uses of it (multigrid.py:1096, :915).  Returns the dense real field
    (numpy ifft normalisation)."""
    from oracle import multigrid as omg
    sub = np.fft.fftn(field.reshape(n))
    raw = np.zeros(N, dtype=complex)
    gx, gy, gz = omg._freq_index(n, N)
    raw[gx[:, None, None], gy[:, None], gz] = sub
    return np.fft.ifftn(raw).real


def _reference_restrict_real_field(dense_field, n, N):
    """The reference's way back (multigrid.py:905-915): spectrum of the dense real field, the entries at the level's fftfreq
    indices, inverse transform on the level mesh, real part."""
    from oracle import multigrid as omg
    gx, gy, gz = omg._freq_index(n, N)
    sub = np.fft.fftn(dense_field.reshape(N))[gx[:, None, None], gy[:, None], gz]
    return np.fft.ifftn(sub).real


NYQUIST_CASES = [((4, 6, 8), (8, 12, 16)), ((6, 4, 8), (9, 8, 8)), ((4, 4, 4), (4, 4, 10)), ((5, 6, 4), (10, 6, 9)),
                 ((8, 6, 10), (8, 6, 10))]


def _check_nyquist_embed_restrict(be, tol):
    import torch
    rng = np.random.default_rng(11)
    for n, N in NYQUIST_CASES:
        gc = N[0] * N[1] * (N[2] // 2 + 1)
        # level field with deliberate Nyquist content (white noise has all of it)
        f = rng.standard_normal((2, int(np.prod(n))))
        spec = be.zeros((2, gc), dtype=torch.complex128)
        be.mg_embed_density(be.to_device(f), n, 1.0, spec, N, accumulate=True)
        dense = be.empty((2, int(np.prod(N))))
        be.mg_restrict_potential(spec, N, N, 1.0 / np.prod(N), dense)          # same mesh: the plain inverse transform
        got = be.to_host(dense)
        for i in range(2):
            ref = _reference_embed_then_real_field(f[i], n, N).ravel()
            assert abs(got[i] - ref).max() < tol * abs(ref).max(), (n, N)
        # way back: a dense real field (all frequencies present) cut down to the level mesh
        d = rng.standard_normal((2, int(np.prod(N))))
        spec2 = be.zeros((2, gc), dtype=torch.complex128)
        be.mg_embed_density(be.to_device(d), N, 1.0, spec2, N, accumulate=False)
        lev = be.empty((2, int(np.prod(n))))
        be.mg_restrict_potential(spec2, N, n, 1.0 / np.prod(n), lev)
        got = be.to_host(lev)
        for i in range(2):
            ref = _reference_restrict_real_field(d[i], n, N).ravel()
            assert abs(got[i] - ref).max() < tol * abs(ref).max(), (n, N)


def test_checker_embed_restrict_follow_the_reference_on_nyquist_planes():
    """Level fields with full Nyquist content through the checker backend's embed / restrict == the reference's index-list
    embedding + .real (1e-13): even level meshes in x, y and z, dense meshes that are finer, equal or odd."""
    from oracle_backend import OracleBackend
    _check_nyquist_embed_restrict(OracleBackend(), 1e-13)


@pytest.mark.gpu
def test_gpu_embed_restrict_follow_the_reference_on_nyquist_planes():
    """The same through isdf_mg_embed_density / isdf_mg_restrict_potential (half spectra on the device): the Hermitian form of
    the reference's placement, weight 1/2 on proper Nyquist entries, +-n/2 averaged on the way back."""
    from pyscf_isdf_amd.backend import HipBackend
    _check_nyquist_embed_restrict(HipBackend(0), 1e-13)


def test_vwn_correlation_closed_form_and_reference_scf_energy():
    """'lda,vwn' (Slater exchange + VWN5 correlation; libxc is absent, the closed form of Vosko-Wilk-Nusair eq. 4.4 stands in):
    (i) v_c = d(rho eps_c)/d rho by central differences; (ii) the all-oracle RKS on the diamond primitive cell of
    pyscf/pbc/dft/test/test_krks.py:59-71,112-119 (gth-szv / gth-pade, 17^3) reproduces the reference's
    e_tot = -10.221426445656439 (places=7 there; measured 2.9e-9)."""
    import scf_helpers
    from oracle import pp as opp
    rho = np.array([1e-6, 1e-3, 0.05, 0.3, 1.7, 20.0])
    e, v = omg.vwn_correlation(rho)
    h = 1e-6
    fd = ((rho * (1 + h)) * omg.vwn_correlation(rho * (1 + h))[0] - (rho * (1 - h)) * omg.vwn_correlation(rho * (1 - h))[0]) / (2 * h * rho)
    assert abs(fd - v).max() < 1e-9
    assert np.all(e < 0) and omg.vwn_correlation(np.array([0.0, 1e-30]))[0].max() == 0.0
    cell = gto.Cell(unit='A', atom='C 0. 0. 0.; C 0.8917 0.8917 0.8917', a=[[0., 1.7834, 1.7834], [1.7834, 0., 1.7834], [1.7834, 1.7834, 0.]],
                    basis='gth-szv', pseudo='gth-pade', mesh=[17] * 3)
    a, mesh = cell.lattice_vectors(), cell.mesh
    S, T = scf_helpers.overlap_kinetic_from_ft(cell)
    ao4 = dense_ao4(cell)
    ps = [cell._pseudo.get(cell.atom_symbol(i)) for i in range(cell.natm)]
    vpp = opp.get_pp(cell._atm, cell._bas, cell._env, cell.atom_coords(), cell.atom_charges(), ps, a, mesh,
                     cell.get_uniform_grids(), [ao4[0]], np.zeros((1, 3)))[0].real
    e_nuc = scf_helpers.ewald_energy(cell)

    def veff(dm):
        vj = offt.get_j(ao4[0], dm, a, mesh)
        n, exc, vxc = omg.nr_rks_lda_dense(ao4[0], dm, a, mesh, xc='lda,vwn')
        return vj + vxc, 0.5 * np.einsum('ij,ji', vj, dm), exc
    e_tot = scf_helpers.rks(T + vpp, S, veff, 4, e_nuc)[0]
    assert abs(e_tot - (-10.221426445656439)) < 5e-8


def _check_vwn(df, cell, tol):
    """nr_rks('lda,vwn') of the product against the oracle on the same ladder and against the dense-grid quadrature (1e-7)."""
    a, mesh = cell.lattice_vectors(), cell.mesh
    dm = make_dm(cell)
    n, e, veff = pmg.nr_rks(df, 'lda,vwn', dm, with_j=True)
    n0, e0, v0, ec0 = omg.nr_rks_lda(as_tasks(df.tasks), cell._atm, dm, a, mesh, with_j=True, xc='lda,vwn')
    assert abs(n - n0) < tol * 100 and abs(e - e0) < tol * 100 and abs(veff - v0).max() < tol * 10
    vxc = pmg.nr_rks(df, 'LDA,VWN', dm)[2]
    nd, ed, vd = omg.nr_rks_lda_dense(dense_ao4(cell)[0], dm, a, mesh, xc='lda,vwn')
    assert abs(e - ed) < 1e-7 and abs(vxc - vd).max() < 1e-7
    # it is a different functional from the exchange alone, and the open-shell form is refused, not silently wrong
    assert abs(e - pmg.nr_rks(df, 'lda,', dm)[1]) > 1e-3
    with pytest.raises(NotImplementedError):
        pmg.nr_uks(df, 'lda,vwn', np.stack([dm, dm]) * .5)


def test_product_vwn_on_checker_backend():
    from oracle_backend import OracleBackend
    cell = cell_he_split()
    df = pmg.MultiGridFFTDF(cell, backend=OracleBackend())
    df.split = 'all'
    _check_vwn(df, cell, 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize('mk', [cell_he_split, cell_c2_orth])
def test_gpu_multigrid_lda_vwn(mk):
    cell = mk()
    df = pmg.MultiGridFFTDF(cell)
    df.split = 'all'
    _check_vwn(df, cell, 1e-9)
