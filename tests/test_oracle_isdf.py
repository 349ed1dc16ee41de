"""CPU tests of the ISDF oracle: pinned to the reference's pivot rule through golden vectors made
from the reference's own pivoted_cholesky_python, to algebraic identities, and to the pinned exact
FFTDF J/K through convergence.  No GPU."""
import json
import os
import numpy as np
import pytest
import cells
from pyscf_isdf_amd import gto
from oracle import ao as oao, isdf as oisdf, fftdf, pbc_tools as tools, c_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pivoted_cholesky_golden.json')


def _make_ao(seed, nao, m):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((nao, m)) * np.exp(-3.0 * rng.random(m))


@pytest.mark.parametrize('impl', ['numpy', 'c'])
def test_pivots_match_reference_golden(impl):
    """Implicit-Gram selection == the reference's pivoted Cholesky on the explicit Gram matrix
    (same pivots, same rank under tol=-1, same factor diagonal)."""
    with open(GOLD) as f:
        gold = json.load(f)
    for case in gold['cases']:
        ao = _make_ao(case['seed'], case['nao'], case['m'])
        fn = oisdf.select_ip if impl == 'numpy' else c_oracle.select_ip
        piv, L = fn(ao, case['m'], tol=-1.0, tie_rtol=0.0)
        assert len(piv) == case['rank']
        assert list(piv) == case['piv']
        assert np.allclose(L[np.arange(len(piv)), piv], case['diag'], rtol=1e-7, atol=0)


def test_explicit_gram_pivoted_cholesky_matches_reference_golden():
    """oracle.pivoted_cholesky_gram (the restatement of isdf_select_ip_gram) on the explicitly formed Gram matrix ==
    the reference's pivoted_cholesky_python on the same matrix (golden pivots, rank and factor diagonal)."""
    with open(GOLD) as f:
        gold = json.load(f)
    for case in gold['cases']:
        ao = _make_ao(case['seed'], case['nao'], case['m'])
        piv, L = oisdf.pivoted_cholesky_gram(ao.T.dot(ao) ** 2, case['m'], tol=-1.0, tie_rtol=0.0)
        assert len(piv) == case['rank'] and list(piv) == case['piv']
        assert np.allclose(L[np.arange(len(piv)), piv], case['diag'], rtol=1e-7, atol=0)
        # and it is the implicit selection's answer too
        assert list(oisdf.select_ip(ao, case['m'], tol=-1.0, tie_rtol=0.0)[0]) == list(piv)


def test_refined_selection_host_logic_with_checker_backend():
    """select='refined' through the host driver (no GPU): per-atom candidates (refine_over x too many), one pivoted
    Cholesky restricted to them; the points equal the oracle's two-stage restatement, are grouped by atom for the
    block-Jacobi route, and K is closer to the exact exchange than with the plain local selection at equal P."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    from pyscf_isdf_amd._common import partition_grid_by_atom
    cell = cells.cell_diamond_prim('gth-szv', (12, 12, 12))
    nao = cell.nao_nr()
    rng = np.random.default_rng(1)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:4] = 2
    dm = (c * occ).dot(c.T)
    rcut = gto.estimate_rcut_per_shell(cell)
    coords = cell.get_uniform_grids()
    ao = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, gto.get_lattice_Ls(cell, rcut=rcut.max()), rcut, rule='point')
    aoT = np.ascontiguousarray(ao.T)
    k_exact = fftdf.get_k(ao, dm, cell.lattice_vectors(), cell.mesh)
    errs = {}
    for sel in ('local', 'refined'):
        df = ISDF(cell, c_isdf=4, select=sel, backend=OracleBackend())
        df.refine_over = 2.0
        errs[sel] = abs(df.get_jk(dm, with_j=False)[1] - k_exact).max()
        assert len(df.ip) == 4 * nao and len(set(df.ip)) == len(df.ip)
        if sel == 'refined':
            owner = partition_grid_by_atom(coords, cell.atom_coords(), cell.lattice_vectors())
            perm = np.argsort(owner, kind='stable')
            off = np.append(0, np.cumsum(np.bincount(owner, minlength=cell.natm)))
            cand = np.concatenate([perm[off[b]:off[b + 1]][oisdf.select_ip(aoT[:, perm[off[b]:off[b + 1]]], 2 * 4 * 4)[0]]
                                   for b in range(cell.natm)])
            chosen = oisdf.refine_selection(aoT, cand, 4 * nao)
            assert sorted(chosen) == sorted(df.ip)
            # stored atom by atom (the preconditioner blocks of the block-Jacobi route), pivot order inside an atom
            own = owner[df.ip]
            assert (np.diff(own) >= 0).all()
            for b in range(cell.natm):
                assert list(df.ip[own == b]) == [g for g in chosen if owner[g] == b]
    assert errs['refined'] < errs['local']


def test_c_oracle_equals_numpy_oracle():
    ao = _make_ao(42, 12, 3000)
    p1, L1 = oisdf.select_ip(ao, 60)
    p2, L2 = c_oracle.select_ip(ao, 60)
    assert np.array_equal(p1, p2)
    assert abs(L1 - L2).max() < 1e-10 * abs(L1).max()


def test_tie_rule_picks_lowest_index():
    base = _make_ao(7, 4, 200)
    ao = np.concatenate([base, base], axis=1)
    for fn in (oisdf.select_ip, c_oracle.select_ip):
        piv, _ = fn(ao, 8)
        assert (piv < 200).all()


@pytest.fixture(scope='module')
def hec():
    cell = cells.cell_he_c()
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    ao = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, rule='point')
    return cell, ao, np.ascontiguousarray(ao.T)


def test_fit_identities(hec):
    cell, ao, aoT = hec
    piv, L = oisdf.select_ip(aoT, 15)
    theta = oisdf.fit_theta(L, piv)
    assert abs(theta[:, piv] - np.eye(15)).max() < 1e-12          # interpolation property
    ne = oisdf.fit_theta_normal_equations(aoT, piv)                # SURVEY 7.1-3 form
    assert abs(theta - ne).max() < 1e-7 * abs(ne).max()
    # residual of the pair densities decreases with the number of points
    i, j = np.tril_indices(cell.nao_nr())
    pairs = aoT[i] * aoT[j]
    errs = []
    for k in (6, 12, 18, 21):
        p, Lk = oisdf.select_ip(aoT, k)
        th = oisdf.fit_theta(Lk, p)
        errs.append(abs(pairs - pairs[:, p].dot(th)).max())
    assert errs[0] > errs[1] > errs[2] > errs[3] and errs[3] < 1e-10


def test_isdf_converges_to_pinned_fftdf(hec):
    """At full rank ISDF is exact: K and the ERIs reproduce the reference's known answers
    (test_fft.py:645, :695) through the ISDF formulas — this pins W's normalisation."""
    cell, ao, aoT = hec
    a, mesh = cell.lattice_vectors(), cell.mesh
    r = oisdf.build_global(aoT, a, mesh, 40)
    assert len(r['ip']) == 21
    dm = np.eye(cell.nao_nr())
    vk = oisdf.get_k(r['aoP'], r['W'], dm)
    assert abs(tools.fp(vk) - 4.290076429522121) < 1e-8
    assert abs(tools.fp(oisdf.isdf_eri_s4(r['aoP'], r['W'])) - 0.80425358275734926) < 1e-8
    assert abs(tools.fp(oisdf.get_j(aoT, dm, a, mesh)) - 3.7955873127283377) < 1e-8
    errs = [abs(oisdf.get_k(*(lambda q: (q['aoP'], q['W']))(oisdf.build_global(aoT, a, mesh, k)), dm)
                - fftdf.get_k(ao, dm, a, mesh)).max() for k in (12, 18, 21)]
    assert errs[0] > errs[1] > errs[2]


def test_local_blocks_with_global_fit(hec):
    """Per-atom selection + regularised global fit (the scalable variant) also converges."""
    cell, ao, aoT = hec
    a, mesh = cell.lattice_vectors(), cell.mesh
    owner = oisdf.partition_by_atom(cell.get_uniform_grids(), cell.atom_coords(), a)
    assert set(np.unique(owner)) == {0, 1}
    r = oisdf.build_local_select_global_fit(aoT, a, mesh, owner, [8, 16], reg_rel=1e-12)
    dm = np.eye(cell.nao_nr())
    vk = oisdf.get_k(r['aoP'], r['W'], dm)
    # 24 points for 21 independent pair products: the diagonal shift bounds the accuracy here
    assert abs(vk - fftdf.get_k(ao, dm, a, mesh)).max() < 1e-5


def test_auto_route_logic_with_checker_backend():
    """Host logic of fit_route='auto' (no GPU): the probe check accepts the block-Jacobi route, and when its mismatch
    exceeds the tolerance (forced here with a tiny tolerance) the build warns and falls back to the Cholesky route;
    the checker backend implements the same stage methods as the HIP one."""
    import warnings
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he2_triclinic()
    dm = np.eye(cell.nao_nr())
    ref = ISDF(cell, c_isdf=8, select='local', backend=OracleBackend()); ref.fit_route = 'cholesky'
    k_ref = ref.get_jk(dm, with_j=False)[1]
    ok = ISDF(cell, c_isdf=8, select='local', backend=OracleBackend())      # 64 points for 36 pair products
    k_ok = ok.get_jk(dm, with_j=False)[1]
    assert ok.fit_route == 'auto' and ok.fit_route_used == 'blockjacobi' and 0 < ok.bj_check <= ok.bj_check_tol
    # both routes solve the same regularised normal equations: they differ by (amplified) rounding only
    assert abs(k_ok - k_ref).max() < 1e-7 * abs(k_ref).max()
    df = ISDF(cell, c_isdf=8, select='local', backend=OracleBackend())
    df.bj_check_tol = 1e-13
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        k_auto = df.get_jk(dm, with_j=False)[1]
    assert df.fit_route_used == 'cholesky' and df.bj_check > df.bj_check_tol
    assert any('probe check' in str(w.message) for w in rec)
    assert abs(k_auto - k_ref).max() < 1e-12
    assert ISDF(cell, c_isdf=15, select='local', backend=OracleBackend())._fit_routes() == ['cholesky']
    with pytest.raises(ValueError):
        bad = ISDF(cell, c_isdf=4, select='local', backend=OracleBackend()); bad.fit_route = 'nonsense'; bad.build()


def test_range_separated_get_jk_host_logic_with_checker_backend():
    """get_jk(omega=...): long-range + short-range = full Coulomb for J and K, W rebuilt once per omega from the same
    fit and cached, cache dropped by the next build; the oracle's kernel follows pyscf/pbc/tools/pbc.py:408-418."""
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-szv', (12, 12, 12))
    nao = cell.nao_nr()
    rng = np.random.default_rng(1)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    aoT = oao.eval_ao(cell._atm, cell._bas, cell._env, cell.get_uniform_grids(), gto.get_lattice_Ls(cell, rcut=gto.estimate_rcut_per_shell(cell).max()),
                      gto.estimate_rcut_per_shell(cell), rule='point').T
    for route in ('cholesky', 'auto'):
        df = ISDF(cell, c_isdf=6, select='local', backend=OracleBackend())
        df.fit_route = route
        vj, vk = df.get_jk(dm)
        vjl, vkl = df.get_jk(dm, omega=0.4)
        vjs, vks = df.get_jk(dm, omega=-0.4)
        assert sorted(df._W_omega) == [-0.4, 0.4]
        assert abs(vjl + vjs - vj).max() < 1e-12 and abs(vkl + vks - vk).max() < 1e-7 * abs(vk).max()
        assert 1e-3 < abs(vkl).max() / abs(vk).max() < 0.5                       # the long-range part is a real fraction
        # J against the reference formula with the attenuated kernel
        assert abs(vjl - oisdf.get_j(aoT, dm, cell.lattice_vectors(), cell.mesh, omega=0.4)).max() < 1e-10
        # K against the oracle's W with the attenuated kernel on the same points (Cholesky route: same regularised fit)
        if route == 'cholesky':
            th = oisdf.fit_theta_global_chol(aoT, df.ip, reg_rel=df.reg_used) if hasattr(oisdf, 'fit_theta_global_chol') else None
            if th is not None:
                W = oisdf.build_W(th, cell.lattice_vectors(), cell.mesh, omega=0.4)
                k_or = oisdf.get_k(np.ascontiguousarray(aoT[:, df.ip].T), W, dm)
                assert abs(vkl - k_or).max() < 1e-8 * abs(k_or).max()
        with df.range_coulomb(0.4) as rsh:                                       # FFTDF.range_coulomb surface
            assert abs(rsh.get_jk(dm)[1] - vkl).max() < 1e-14 and rsh.cell is df.cell
        assert df.to_gpu() is df
        df.build()
        assert df._W_omega == {}


def test_robust_k_host_logic_with_checker_backend():
    """robust_k through the host driver (no GPU): forces Theta itself, keeps V = conv(Theta) in the fit buffer, assembles
    K1 + K2 - K_isdf in row batches; equals the oracle's restatement and beats the plain ISDF K against the exact one."""
    import cells
    from oracle_backend import OracleBackend
    from oracle import fftdf
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-szv', (12, 12, 12))
    nao = cell.nao_nr()
    rng = np.random.default_rng(1)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:4] = 2
    dm = (c * occ).dot(c.T)
    rcut = gto.estimate_rcut_per_shell(cell)
    ao = oao.eval_ao(cell._atm, cell._bas, cell._env, cell.get_uniform_grids(), gto.get_lattice_Ls(cell, rcut=rcut.max()), rcut, rule='point')
    aoT = np.ascontiguousarray(ao.T)
    a, mesh = cell.lattice_vectors(), cell.mesh
    k_exact = fftdf.get_k(ao, dm, a, mesh)
    plain = ISDF(cell, c_isdf=3, select='local', backend=OracleBackend())
    k_plain = plain.get_jk(dm, with_j=False)[1]
    df = ISDF(cell, c_isdf=3, select='local', backend=OracleBackend())
    df.robust_k = True
    k_rob = df.get_jk(dm, with_j=False)[1]
    assert df._want_theta and not df.explicit_theta and df._V is not None and df._fit_state is None
    th = oisdf.fit_theta_global_chol(aoT, df.ip, reg_rel=df.reg_used)
    assert abs(k_rob - oisdf.get_k_robust(aoT, df.ip, th, dm, a, mesh)).max() < 1e-12
    assert abs(k_rob - k_exact).max() < 0.5 * abs(k_plain - k_exact).max()
    with pytest.raises(NotImplementedError):
        df.get_jk(dm, omega=0.2)
    sh = ISDF(cell, c_isdf=3, select='local', backend=OracleBackend())          # the grid-sharded layout, one rank
    sh.robust_k = True
    sh.force_sharded = True
    sh.fft_batch = 5
    assert abs(sh.get_jk(dm, with_j=False)[1] - k_rob).max() < 1e-12


def test_paneled_build_host_logic_with_checker_backend():
    """More interpolation points than HBM holds fit rows for (max_resident_rows forces it): the rows are produced panel by
    panel, M' is assembled from diagonal panel blocks and recomputed-batch x resident-panel blocks, the probe check is
    accumulated on the way.  Same W and K as the single-pass block-Jacobi build, same range-separated rebuild, for 2, 3 and
    5 panels with ragged FFT batches; a panel budget below one preconditioner block is an error."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = gto.diamond_supercell(2, 'gth-szv', (12, 12, 12))
    nao = cell.nao_nr()
    rng = np.random.default_rng(1)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    ref = ISDF(cell, c_isdf=2, select='refined', backend=OracleBackend())
    ref.bj_check_tol = 1e-6
    k0 = ref.get_jk(dm, with_j=False)[1]
    W0 = ref.W.numpy().copy()
    assert ref.fit_route_used == 'blockjacobi' and ref.n_panels == 1
    P = len(ref.ip)
    seen = set()
    for rows, npan in ((P // 2 + 3, 2), (P // 3 + 5, 3), (P // 5 + 8, 5)):
        df = ISDF(cell, c_isdf=2, select='refined', backend=OracleBackend())
        df.max_resident_rows, df.fft_batch, df.bj_check_tol = rows, 11, 1e-6
        k1 = df.get_jk(dm, with_j=False)[1]
        assert df.n_panels == len(df._fit_state['panels']) >= npan - 1 and df._fit_state['kind'] == 'blockjacobi-paneled'
        seen.add(df.n_panels)
        assert np.array_equal(df.ip, ref.ip)
        assert max(y - x for x, y in df._fit_state['panels']) <= rows
        assert abs(df.W.numpy() - W0).max() < 1e-9 * abs(W0).max()
        assert abs(k1 - k0).max() < 1e-10 * abs(k0).max()
        assert abs(df.bj_check - ref.bj_check) < 0.5 * ref.bj_check + 1e-13
        kl, ks = df.get_jk(dm, omega=0.4, with_j=False)[1], df.get_jk(dm, omega=-0.4, with_j=False)[1]
        assert abs(kl + ks - k1).max() < 1e-8 * abs(k1).max()
    assert len(seen) >= 2 and min(seen) >= 2
    small = ISDF(cell, c_isdf=2, select='local', backend=OracleBackend())
    small.max_resident_rows = 3
    with pytest.raises(MemoryError):
        small.build()


def test_pair_transforms_and_loop_host_logic_with_checker_backend():
    """The rest of the with_df surface (pyscf/pbc/df/fft.py:317-345): get_ao_pairs_G reproduces the reference's fp(eri)
    through (ij|kl) = sum_G conj((G|ij)) coulG vol/G^2 (G|kl) (fft_ao2mo.py:154-184, pin test_fft.py:695), compact and full
    layouts agree, get_mo_pairs_G is the transformed tensor, and loop() yields three-index blocks whose squares sum to the
    object's own ERIs (the contract of FFTDF.loop), with get_naoaux() rows in total.  (ao2mo_7d: tests/test_oracle_kisdf.py.)"""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    nao = cell.nao_nr()
    df = ISDF(cell, c_isdf=4, select='global', backend=OracleBackend())
    G = int(np.prod(cell.mesh))
    coulG = tools.get_coulG(cell.lattice_vectors(), cell.mesh)
    pc = df.get_ao_pairs_G(compact=True)
    assert pc.shape == (G, nao * (nao + 1) // 2) and pc.dtype == np.complex128
    eri = (pc.conj().T * (coulG * cell.vol / G ** 2)).dot(pc).real
    assert abs(tools.fp(eri) - 0.80425358275734926) < 1e-8
    pf = df.get_ao_pairs_G(compact=False)
    i, j = np.tril_indices(nao)
    assert pf.shape == (G, nao * nao) and abs(pf.reshape(G, nao, nao)[:, i, j] - pc).max() < 1e-12
    sl = df.get_ao_pairs_G(shls_slice=(0, 1, 1, 3))
    loc = cell.ao_loc_nr()
    assert abs(sl - pf.reshape(G, nao, nao)[:, loc[0]:loc[1], loc[1]:loc[3]].reshape(G, -1)).max() < 1e-12
    rng = np.random.default_rng(0)
    ci, cj = rng.standard_normal((nao, 2)), rng.standard_normal((nao, 3))
    pm = df.get_mo_pairs_G((ci, cj))
    assert abs(pm - np.einsum('gpq,pi,qj->gij', pf.reshape(G, nao, nao), ci, cj).reshape(G, -1)).max() < 1e-10
    # a pair of different k-points: the transform of conj(phi^k1) phi^k2 exp(-i q.r), exact exchange-type integral against oracle
    kpts = np.array([[0.1, 0.2, -0.1], [0.3, -0.1, 0.2]])
    pk = df.get_ao_pairs_G(kpts)
    assert pk.shape == (G, nao * nao)
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    a1 = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpts, rule='point')
    ref = tools.fft((a1[0].conj()[:, :, None] * a1[1][:, None, :]).reshape(G, -1).T * np.exp(-1j * coords.dot(kpts[1] - kpts[0])), cell.mesh).T
    assert abs(pk - ref).max() < 1e-10
    blocks = list(df.loop(blksize=7))
    L = np.vstack(blocks)
    assert all(len(b) <= 7 for b in blocks) and L.shape[1] == nao * (nao + 1) // 2 and len(L) <= df.get_naoaux()
    assert abs(L.T.dot(L) - df.get_ao_eri(compact=True)).max() < 1e-9


def test_vcut_sph_exchange_host_logic_with_checker_backend():
    """exxdiv='vcut_sph' at the Gamma point through the host driver: K with the kernel 4 pi/G^2 (1 - cos(|G| Rc)), G = 0 ->
    2 pi Rc^2, Rc = (3 vol / 4 pi)^(1/3) (pyscf/pbc/tools/pbc.py:312-317), equals the oracle's W built with that kernel on
    the same points; J and the plain K are unchanged; the variant is cached until the next build."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    nao = cell.nao_nr()
    rng = np.random.default_rng(2)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    df = ISDF(cell, c_isdf=3, select='local', backend=OracleBackend())
    df.fit_route = 'cholesky'
    vj0, vk0 = df.get_jk(dm)
    vj1, vk1 = df.get_jk(dm, exxdiv='vcut_sph')
    assert abs(vj1 - vj0).max() < 1e-13 and list(df._W_omega) == ['vcut_sph']
    rc = (3 * cell.vol / (4 * np.pi)) ** (1. / 3)
    aoT = np.ascontiguousarray(oao.eval_ao(cell._atm, cell._bas, cell._env, cell.get_uniform_grids(),
                                           gto.get_lattice_Ls(cell, rcut=gto.estimate_rcut_per_shell(cell).max()),
                                           gto.estimate_rcut_per_shell(cell), rule='point').T)
    th = oisdf.fit_theta_global_chol(aoT, df.ip, reg_rel=df.reg_used)
    W = oisdf.build_W(th, cell.lattice_vectors(), cell.mesh, rc=rc)
    k_or = oisdf.get_k(np.ascontiguousarray(aoT[:, df.ip].T), W, dm)
    assert abs(vk1 - k_or).max() < 1e-9 * abs(k_or).max()
    assert abs(vk1 - vk0).max() > 1e-3 * abs(vk0).max()
    assert abs(df.get_jk(dm, with_j=False)[1] - vk0).max() < 1e-13           # the plain W is still in place
    df.build()
    assert df._W_omega == {}


def _diamond_case(nocc=4, c_isdf=4, mesh=(12, 12, 12)):
    cell = cells.cell_diamond_prim('gth-szv', mesh)
    nao = cell.nao_nr()
    rng = np.random.default_rng(1)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:nocc] = 2
    dm = (c * occ).dot(c.T)
    rcut = gto.estimate_rcut_per_shell(cell)
    coords = cell.get_uniform_grids()
    ao = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, gto.get_lattice_Ls(cell, rcut=rcut.max()), rcut, rule='point')
    return cell, ao, np.ascontiguousarray(ao.T), dm, c, occ


def _tag(dm, mo_coeff, mo_occ):
    class Tagged(np.ndarray):
        pass
    t = np.asarray(dm).view(Tagged)
    t.mo_coeff, t.mo_occ = mo_coeff, mo_occ
    return t


@pytest.mark.parametrize('route', ['cholesky', 'blockjacobi'])
def test_occ_pair_space_host_logic_with_checker_backend(route):
    """pair_space='occ' through the host driver (no GPU): build() stops after the candidate stage; get_jk with an MO-tagged
    density (the tag of pyscf/pbc/df/fft_jk.py:206-210) picks the points from the product Gram matrix
    (phi^T phi) o (psi^T psi) of the candidates, fits in that pair space and returns the oracle's K for the same points;
    the same density again does not refit, another density does; an untagged positive semidefinite density gives the
    same fit through its eigenvectors."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    from pyscf_isdf_amd._common import partition_grid_by_atom
    cell, ao, aoT, dm, c, occ = _diamond_case(nocc=2)
    nao = cell.nao_nr()
    a, mesh = cell.lattice_vectors(), cell.mesh
    df = ISDF(cell, c_isdf=4, select='refined', backend=OracleBackend())
    df.pair_space = 'occ'
    df.fit_route = route
    df.build()
    assert df._fit_pending and df.W is None and df.ip is None
    vk = df.get_jk(_tag(dm, c, occ), with_j=False)[1]
    assert not df._fit_pending and len(set(df.ip)) == len(df.ip) <= 4 * nao      # the pair space of 2 orbitals x 8 AOs has rank <= 16
    # the oracle's restatement of the two stages and of the fit
    coords = cell.get_uniform_grids()
    owner = partition_grid_by_atom(coords, cell.atom_coords(), a)
    perm = np.argsort(owner, kind='stable')
    off = np.append(0, np.cumsum(np.bincount(owner, minlength=cell.natm)))
    cand = np.concatenate([perm[off[b]:off[b + 1]][oisdf.select_ip(aoT[:, perm[off[b]:off[b + 1]]], 2 * 4 * 4)[0]]
                           for b in range(cell.natm)])
    psi = oisdf.occupied_on_grid(aoT, c, occ)
    chosen = oisdf.refine_selection_occ(aoT, psi, cand, 4 * nao)
    assert sorted(chosen) == sorted(df.ip)
    ip = df.ip
    if route == 'cholesky':
        theta = oisdf.fit_theta_occ_chol(aoT, psi, ip, reg_rel=df.reg_rel)
        W = oisdf.build_W(theta, a, mesh)
    else:
        cnt = np.bincount(owner[ip], minlength=cell.natm)
        W = oisdf.build_W_blockjacobi_occ(aoT, psi, ip, np.append(0, np.cumsum(cnt)), a, mesh, reg_rel=df.reg_rel)
    k_or = oisdf.get_k(np.ascontiguousarray(aoT[:, ip].T), W, dm)
    assert abs(vk - k_or).max() < 1e-8 * abs(k_or).max()
    # same density again: no refit (the W buffer is not rewritten); untagged: same occupied space through the eigenvectors
    Wid = df.W.data_ptr()
    serial = dict(df.timings)
    vk2 = df.get_jk(dm, with_j=False)[1]
    assert df.W.data_ptr() == Wid and df.timings.get('S3_fit') == serial.get('S3_fit')
    assert abs(vk2 - vk).max() < 1e-12
    # another occupied space: refit, and the fit is the better one for ITS density
    occ2 = np.zeros(nao); occ2[2:4] = 2
    dm2 = (c * occ2).dot(c.T)
    k_exact2 = fftdf.get_k(ao, dm2, a, mesh)
    k_stale = oisdf.get_k(np.ascontiguousarray(aoT[:, ip].T), W, dm2)
    vk3 = df.get_jk(_tag(dm2, c, occ2), with_j=False)[1]
    assert df.timings['S3_fit'] > serial['S3_fit']
    assert abs(vk3 - k_exact2).max() < abs(k_stale - k_exact2).max()
    # occ_refit = 'once' keeps the fit
    df.occ_refit = 'once'
    vk4 = df.get_jk(_tag(dm, c, occ), with_j=False)[1]
    assert abs(vk4 - oisdf.get_k(np.ascontiguousarray(aoT[:, df.ip].T), df.backend.to_host(df.W), dm)).max() < 1e-12
    t_fit = df.timings['S3_fit']
    df.get_jk(_tag(dm2, c, occ2), with_j=False)
    assert df.timings['S3_fit'] == t_fit


def test_occ_pair_space_beats_ao_pair_space_at_equal_points():
    """The (AO x occupied) pair space is what K needs (fft_jk.py:235-238): at equal numbers of points its fit reproduces the
    exact exchange of THAT density better than the AO x AO fit (oracle arithmetic, diamond primitive cell)."""
    cell, ao, aoT, dm, c, occ = _diamond_case(nocc=1)
    a, mesh = cell.lattice_vectors(), cell.mesh
    k_exact = fftdf.get_k(ao, dm, a, mesh)
    P = 20
    piv_a, L_a = oisdf.select_ip(aoT, P, tol=0.0)
    th_a = oisdf.fit_theta(L_a, piv_a)
    err_ao = abs(oisdf.get_k(np.ascontiguousarray(aoT[:, piv_a].T), oisdf.build_W(th_a, a, mesh), dm) - k_exact).max()
    psi = oisdf.occupied_on_grid(aoT, c, occ)
    ip = oisdf.refine_selection_occ(aoT, psi, np.arange(aoT.shape[1]), P, tol=0.0)
    th_o = oisdf.fit_theta_occ_chol(aoT, psi, ip, reg_rel=1e-13)
    err_occ = abs(oisdf.get_k(np.ascontiguousarray(aoT[:, ip].T), oisdf.build_W(th_o, a, mesh), dm) - k_exact).max()
    assert err_occ < 0.2 * err_ao


def test_occ_pair_space_switches_back_to_ao_pairs_when_needed():
    """pair_space='occ': what W is contracted with decides the pair space.  After a fit for an MO-tagged density, AO integrals
    (get_ao_eri) and a density WITHOUT occupied-orbital form (indefinite: a response / difference density) are served from an
    AO x AO fit - the (AO x occupied) fit does not represent those pairs - and the next tagged density gets its own fit back."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell, ao, aoT, dm, c, occ = _diamond_case(nocc=2)
    nao = cell.nao_nr()
    tdm = _tag(dm, c, occ)
    ref = ISDF(cell, c_isdf=4, select='refined', backend=OracleBackend())
    ref.fit_route = 'cholesky'
    eri_ao = ref.get_ao_eri()
    rng = np.random.default_rng(3)
    d1 = rng.standard_normal((nao, nao)); d1 = d1 + d1.T                     # indefinite
    k_ao = ref.get_jk(d1, with_j=False)[1]
    df = ISDF(cell, c_isdf=4, select='refined', backend=OracleBackend())
    df.pair_space, df.fit_route = 'occ', 'cholesky'
    vk_occ = df.get_jk(tdm, with_j=False)[1]
    assert df._fit_dm is not None
    assert abs(df.get_ao_eri() - eri_ao).max() < 1e-9 * abs(eri_ao).max() and df._fit_dm is None
    assert abs(df.get_jk(d1, with_j=False)[1] - k_ao).max() < 1e-9 * abs(k_ao).max()
    vk_again = df.get_jk(tdm, with_j=False)[1]
    assert df._fit_dm is not None and abs(vk_again - vk_occ).max() < 1e-10 * abs(vk_occ).max()


def test_spectral_W_host_logic_with_checker_backend():
    """W = X X^T from the half spectra of the fit rows (fit_route.FitRouteMixin._spectral_plan / _finish_W_spectral): with the
    whole box kept it is the classic W to the (amplified) rounding of the route; the sphere is taken by 'auto' only when the
    mesh resolves the AO pair products, and a forced sphere on a mesh that does not is caught by the probe check."""
    import warnings
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-szv', (24, 24, 24))
    nao = cell.nao_nr()
    rng = np.random.default_rng(1)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T

    def run(**kw):
        df = ISDF(cell, c_isdf=6, select='local', backend=OracleBackend())
        for k, v in kw.items():
            setattr(df, k, v)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter('always')
            vk = df.get_jk(dm, with_j=False)[1]
        return df, vk, [str(w.message) for w in rec]
    ref, k_ref, _ = run(w_spectral=False)
    assert ref.w_spectral_fraction is None and ref.fit_route_used == 'blockjacobi'
    box, k_box, _ = run(w_sphere=0, w_spectral_check_tol=3e-8)
    assert box.w_spectral_fraction > 1.0 and box._fit_state['kind'] == 'blockjacobi-spectral'
    assert abs(k_box - k_ref).max() < 1e-7 * abs(k_ref).max() and box.bj_check <= box.bj_check_tol
    # the plan: every point of the half spectrum with a positive kernel value, multiplicity 1 on the kz = 0 and Nyquist planes
    plan = box._spectral_plan()
    n2h = 24 // 2 + 1
    assert plan['npts'] == 24 * 24 * n2h - 1 and plan['ldx'] % 128 == 0 and plan['ldx'] >= 2 * plan['npts']
    # range separation rebuilds W through the same spectral state (long range + short range = full)
    kl = box.get_jk(dm, with_j=False, omega=0.4)[1]
    ks = box.get_jk(dm, with_j=False, omega=-0.4)[1]
    assert abs(kl + ks - k_box).max() < 1e-7 * abs(k_box).max()
    # 'auto' on this 24^3 mesh: 7e-8 of the products' Coulomb energy sits outside the sphere -> the classic build
    auto, k_auto, _ = run()
    assert auto.w_spectral_fraction is None and 1e-9 < auto._sphere_share[1] < 1e-6 and np.array_equal(k_auto, k_ref)
    # a forced sphere is about 0.3 of the box for the fcc cell; K moves by what the corners carried
    sph, k_sph, _ = run(w_sphere=100.0, bj_check_tol=1e-6, w_spectral_check_tol=1e-6)
    assert 0.25 < sph.w_spectral_fraction < 0.35 and 1e-10 < abs(k_sph - k_ref).max() < 1e-6
    # on a coarse mesh the probe check rejects it and the classic build takes over
    cell12 = cells.cell_diamond_prim('gth-szv', (12, 12, 12))
    df = ISDF(cell12, c_isdf=6, select='local', backend=OracleBackend())
    df.w_sphere = 100.0
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        k12 = df.get_jk(dm, with_j=False)[1]
    assert any('spectral build' in str(w.message) for w in rec) and df.w_spectral_fraction is None or df._fit_state['kind'] != 'blockjacobi-spectral'
    ref12 = ISDF(cell12, c_isdf=6, select='local', backend=OracleBackend()); ref12.w_spectral = False
    assert abs(k12 - ref12.get_jk(dm, with_j=False)[1]).max() < 1e-12


def test_oracle_spectral_W_is_the_parseval_form_of_build_W(hec):
    """oracle.isdf.build_W_spectral: whole box = build_W to rounding (plain, range-separated and spherically truncated kernels);
    inside the sphere it is exact for band-limited rows and close for the smooth rows of a fit; and it is what the checker
    backend's packed rows give through the object's plan (multiplicities of the half spectrum included)."""
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell, ao, aoT = hec
    a, mesh = cell.lattice_vectors(), [int(x) for x in cell.mesh]
    rng = np.random.default_rng(4)
    rows = rng.standard_normal((7, aoT.shape[1]))
    for kw in (dict(), dict(omega=0.5), dict(omega=-0.5), dict(rc=3.0)):
        W0, W1 = oisdf.build_W(rows, a, mesh, **kw), oisdf.build_W_spectral(rows, a, mesh, **kw)
        assert abs(W0 - W1).max() < 1e-12 * abs(W0).max()
    # band-limited rows: nothing outside the sphere, so the truncated sum is the whole sum
    G = aoT.shape[1]
    z = np.fft.fftn(rows.reshape(7, *mesh), axes=(1, 2, 3))
    f = [np.fft.fftfreq(n, 1.0 / n) for n in mesh]
    low = (abs(f[0])[:, None, None] <= 3) & (abs(f[1])[None, :, None] <= 3) & (abs(f[2])[None, None, :] <= 3)
    smooth = np.fft.ifftn(z * low, axes=(1, 2, 3)).real.reshape(7, G)
    Ws, Wb = oisdf.build_W_spectral(smooth, a, mesh, sphere_pct=100.0), oisdf.build_W(smooth, a, mesh)
    assert abs(Ws - Wb).max() < 1e-12 * abs(Wb).max()
    assert abs(oisdf.build_W_spectral(rows, a, mesh, sphere_pct=100.0) - oisdf.build_W(rows, a, mesh)).max() > 1e-6 * abs(W0).max()
    # the product path: plan of the object + packed rows of the checker backend
    df = ISDF(cell, c_isdf=4, select='local', backend=OracleBackend())
    for pct in (0, 100.0, 60.0):
        df.w_sphere = pct
        plan = df._spectral_plan()
        X = df.backend.empty((7, plan['ldx']))
        df.backend.spectral_rows(df.backend.to_device(smooth if pct else rows), np.asarray(mesh), plan['idx'], plan['scale'], X, batch=7)
        Wx = X.numpy().dot(X.numpy().T)
        ref = oisdf.build_W_spectral(smooth if pct else rows, a, mesh, sphere_pct=pct)
        assert abs(Wx - ref).max() < 1e-12 * abs(ref).max()


def test_spectral_state_falls_back_for_kernels_without_a_spectral_form():
    """exxdiv='vcut_ws' (a kernel table with negative entries has no sqrt) and 'vcut_sph' (non-negative: spectral) requested from
    an object whose fit was built in the spectral form: the truncated-kernel W is rebuilt from the same factors - through the
    paneled classic product for vcut_ws, through X X^T for vcut_sph - and K equals the classic object's."""
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-szv', (12, 12, 12))
    nao = cell.nao_nr()
    rng = np.random.default_rng(3)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    ref = ISDF(cell, c_isdf=5, select='local', backend=OracleBackend()); ref.w_spectral, ref.fit_route = False, 'blockjacobi'
    sp = ISDF(cell, c_isdf=5, select='local', backend=OracleBackend()); sp.w_sphere, sp.fit_route = 0, 'blockjacobi'
    k0 = ref.get_jk(dm, with_j=False)[1]
    k1 = sp.get_jk(dm, with_j=False)[1]
    assert sp._fit_state['kind'] == 'blockjacobi-spectral' and abs(k1 - k0).max() < 1e-7 * abs(k0).max()
    for ex in ('vcut_sph', 'vcut_ws'):
        a = ref.get_jk(dm, with_j=False, exxdiv=ex)[1]
        b = sp.get_jk(dm, with_j=False, exxdiv=ex)[1]
        assert abs(a - k0).max() > 1e-3 * abs(k0).max()                  # the truncation does change K
        assert abs(a - b).max() < 1e-7 * abs(a).max()
    assert sp._fit_state['kind'] == 'blockjacobi-spectral'               # the state survives the fallback
