"""k-point ISDF oracle against the reference's exact k-point exchange (oracle/fftdf.get_jk_kpts, which
is pinned to pyscf/pbc/df/test/test_fft.py:670-676).  No GPU."""
import numpy as np
import pytest
import cells
from pyscf_isdf_amd import gto
from oracle import ao as oao, fftdf, kisdf


def _setup(nk=2):
    cell = cells.cell_he2_triclinic()           # s and p shells, triclinic lattice, 17^3
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    rng = np.random.default_rng(11)
    kpts = rng.random((nk, 3)) * 0.6
    kpts[0] = 0.0                                # Gamma + a generic k-point
    aos = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpts, rule='point')
    aos = [np.asarray(x, dtype=complex) for x in aos]
    nao = cell.nao_nr()
    c = rng.standard_normal((nk, nao, nao)) + 1j * rng.standard_normal((nk, nao, nao))
    dms = np.einsum('kpi,kqi->kpq', c[:, :, :3], c[:, :, :3].conj())       # Hermitian, rank 3
    return cell, coords, kpts, aos, dms


def test_kisdf_converges_to_exact_k_point_exchange():
    cell, coords, kpts, aos, dms = _setup(2)
    a, mesh = cell.lattice_vectors(), cell.mesh
    vj_ref, vk_ref = fftdf.get_jk_kpts(aos, dms, a, mesh, coords, kpts)
    errs = []
    for nip in (40, 120, 400):
        r = kisdf.build(aos, coords, kpts, a, mesh, nip)
        vk = kisdf.get_k_kpts(r['aoP'], r['W'], r['qindex'], dms)
        errs.append(abs(vk - vk_ref).max())
    assert errs[0] > errs[1] > errs[2]
    assert errs[2] < 1e-7 * abs(vk_ref).max()
    assert abs(vk_ref - vk_ref.conj().transpose(0, 2, 1)).max() < 1e-10     # Hermitian K for Hermitian D


def test_kisdf_band_kpoints_and_interpolation_property():
    cell, coords, kpts, aos, dms = _setup(2)
    a, mesh = cell.lattice_vectors(), cell.mesh
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    kb = np.array([[0.11, -0.2, 0.05]])
    ao_b = [np.asarray(x, dtype=complex) for x in
            oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kb, rule='point')]
    vj_ref, vk_ref = fftdf.get_jk_kpts(aos, dms, a, mesh, coords, kpts, ao_band=ao_b, kpts_band=kb)
    # the band k-point's periodic parts must be in the fitted set: include it in the selection basis
    X_all = aos + ao_b
    k_all = np.vstack([kpts, kb])
    X = kisdf.periodic_stack(X_all, coords, k_all)
    piv, L = kisdf.select_ip(X, 300)
    theta = kisdf.fit_theta(X, piv)
    assert abs(theta[:, piv] - np.eye(len(piv))).max() < 1e-5          # interpolation property (ill-conditioned A_PP)
    qs, qidx = kisdf.unique_q(kpts, kb)
    Ws = [kisdf.build_Wq(theta, a, mesh, q, coords[piv]) for q in qs]
    vk = kisdf.get_k_kpts([ao[piv] for ao in aos], Ws, qidx, dms, aoP_band=[ao_b[0][piv]])
    assert abs(vk - vk_ref).max() < 1e-5 * abs(vk_ref).max()             # 300 points: fitting error, not full rank


def test_kpts_band_host_logic_with_checker_backend():
    """get_jk(kpts, kpts_band) through the host driver (no GPU): band k-points join the stacked periodic parts, q = k2 - kb,
    result shapes as df_jk._format_jks (df_jk.py:1426-1444); converges to the reference's constant for this call
    (pyscf/pbc/df/test/test_fft.py:663-676) as the point set grows."""
    import cells
    from oracle_backend import OracleBackend
    from oracle import pbc_tools as otools
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    np.random.seed(1)
    kpts = np.random.random((4, 3))
    kpts[3] = kpts[0] - kpts[1] + kpts[2]
    np.random.seed(1)
    kpts_band = np.random.random((2, 3))
    nao, nk = cell.nao_nr(), 4
    mo_coeff = np.random.random((nk, nao, nao))
    mo_occ = np.array(np.random.random((nk, nao)) > .6, dtype=np.double)
    dms = np.einsum('kpi,ki,kqi->kpq', mo_coeff, mo_occ, mo_coeff)
    df = ISDF(cell, kpts=kpts, c_isdf=20, select='global', backend=OracleBackend())
    df.select_tol = 0.0
    df.k_ip_factor = 2
    vj, vk = df.get_jk(dms, kpts=kpts, kpts_band=kpts_band)
    assert vj.shape == vk.shape == (2, nao, nao) and df._nk_stack == 4          # the band points are k-points here
    assert abs(otools.fp(vk) - (10.239828255099447 + 2.1190549216896182j)) < 1e-4
    v1j, v1k = df.get_jk(dms, kpts=kpts, kpts_band=np.array([0.1, 0.2, 0.3]), with_j=False)
    assert v1j is None and v1k.shape == (nao, nao) and df._nk_stack == 5


def test_kpoint_ao_eri_host_logic_with_checker_backend():
    """k-point AO ERIs from the factorisation through the host driver (no GPU): converge to the reference's constant for four
    momentum-conserving k-points (pyscf/pbc/df/test/test_fft.py:702-703)."""
    import cells
    from oracle_backend import OracleBackend
    from oracle import pbc_tools as otools
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    np.random.seed(1)
    kpts = np.random.random((4, 3))
    kpts[3] = kpts[0] - kpts[1] + kpts[2]
    df = ISDF(cell, kpts=kpts, c_isdf=20, select='global', backend=OracleBackend())
    df.select_tol = 0.0
    df.k_ip_factor = 2
    eri4 = df.get_ao_eri(kpts)
    assert eri4.shape == (36, 36)
    assert abs(otools.fp(eri4) - (0.33709288394542991 - 0.94185725001175313j)) < 2e-4
    with pytest.raises(ValueError):
        df.get_ao_eri(kpts[:3])


def test_supercell_kmesh_cross_check_with_checker_backend():
    """k2gamma: a [2,1,1] k-mesh on the He2 cell == the Gamma point of the 2x1x1 supercell (the cross-check of
    pyscf/pbc/scf/test/test_khf.py:73, here on J and K themselves): the k-blocked J maps onto the supercell J to 1e-9
    (both exact), energies per cell agree, K agrees to the fit error at (numerical) full rank, and the phase table is unitary."""
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    from pyscf_isdf_amd import k2gamma
    cell = cells.cell_he2_triclinic()
    cell.mesh = np.array([9, 9, 9])
    kmesh = [2, 1, 1]
    kpts = cell.make_kpts(kmesh)
    nao, nk = cell.nao_nr(), 2
    assert k2gamma.kpts_to_kmesh(cell, kpts) == kmesh
    scell, phase = k2gamma.get_phase(cell, kpts)
    assert abs(phase.conj().T.dot(phase) - np.eye(nk)).max() < 1e-14 and scell.nao_nr() == nk * nao
    assert list(scell.mesh) == [18, 9, 9]
    rng = np.random.default_rng(8)
    c = rng.standard_normal((nk, nao, 2))
    dms = np.einsum('kpi,kqi->kpq', c, c)                       # real symmetric at the two time-reversal invariant points
    dm_sc = k2gamma.to_supercell_ao_integrals(cell, kpts, dms)
    assert abs(dm_sc.imag).max() < 1e-13
    assert abs(k2gamma.to_kpts_ao_integrals(cell, kpts, dm_sc) - dms).max() < 1e-13
    dfk = ISDF(cell, kpts=kpts, c_isdf=30, select='global', backend=OracleBackend())
    dfk.select_tol, dfk.k_ip_factor = 0.0, 2
    vj, vk = dfk.get_jk(dms, kpts=kpts)
    dfs = ISDF(scell, c_isdf=30, select='global', backend=OracleBackend())
    dfs.select_tol = 0.0
    vjs, vks = dfs.get_jk(dm_sc.real)
    vj_map = k2gamma.to_supercell_ao_integrals(cell, kpts, vj)
    vk_map = k2gamma.to_supercell_ao_integrals(cell, kpts, vk)
    assert abs(vj_map.imag).max() < 1e-10 and abs(vj_map.real - vjs).max() < 1e-9
    ej_k = np.einsum('kij,kji', vj, dms).real / 2 / nk
    ej_s = np.einsum('ij,ji', vjs, dm_sc.real) / 2 / nk
    assert abs(ej_k - ej_s) < 1e-10
    assert abs(vk_map.real - vks).max() < 1e-4 * abs(vks).max()
    ek_k = np.einsum('kij,kji', vk, dms).real / 4 / nk
    ek_s = np.einsum('ij,ji', vks, dm_sc.real) / 4 / nk
    assert abs(ek_k - ek_s) < 1e-5 * abs(ek_s)
    # exxdiv='vcut_sph' (pbc.py:312-317): the k-mesh's cutoff sphere has the volume of nk cells = the supercell's own, so
    # the identity holds for the truncated kernel as well; J is untouched, K changes by a finite amount
    vjc, vkc = dfk.get_jk(dms, kpts=kpts, exxdiv='vcut_sph')
    vjsc, vksc = dfs.get_jk(dm_sc.real, exxdiv='vcut_sph')
    assert abs(vjc - vj).max() < 1e-12 and abs(vjsc - vjs).max() < 1e-12
    assert abs(k2gamma.to_supercell_ao_integrals(cell, kpts, vkc).real - vksc).max() < 1e-4 * abs(vksc).max()
    assert abs(vksc - vks).max() > 1e-3 * abs(vks).max()
    assert sorted(k for k in dfk._W_omega) == ['vcut_sph'] and 'vcut_sph' in dfs._W_omega
    # exxdiv='vcut_ws' on this triclinic lattice: the reference refuses it (test_pbc.py:43-52, pbc.py:466-473) and so do we
    with pytest.raises(RuntimeError):
        dfk.get_jk(dms, kpts=kpts, exxdiv='vcut_ws')


def test_supercell_kmesh_identity_with_the_wigner_seitz_kernel():
    """exxdiv='vcut_ws' (pbc.py:318-346): the kernel is truncated to the Wigner-Seitz cell of the nk-fold lattice, which IS the
    supercell's own lattice - so a [2,1,1] k-mesh on the cubic He/C test cell and the Gamma point of its 2x1x1 supercell must give
    the same exchange (to the fit error), different from the untruncated one; J is untouched."""
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    from pyscf_isdf_amd import k2gamma
    cell = cells.cell_he_c()
    cell.mesh = np.array([9, 9, 9])
    kpts = cell.make_kpts([2, 1, 1])
    nao, nk = cell.nao_nr(), 2
    scell, phase = k2gamma.get_phase(cell, kpts)
    rng = np.random.default_rng(9)
    c = rng.standard_normal((nk, nao, 2))
    dms = np.einsum('kpi,kqi->kpq', c, c)
    dm_sc = k2gamma.to_supercell_ao_integrals(cell, kpts, dms).real
    dfk = ISDF(cell, kpts=kpts, c_isdf=25, select='global', backend=OracleBackend())
    dfk.select_tol, dfk.k_ip_factor = 0.0, 2
    dfs = ISDF(scell, c_isdf=25, select='global', backend=OracleBackend())
    dfs.select_tol = 0.0
    vj, vk = dfk.get_jk(dms, kpts=kpts)
    vjw, vkw = dfk.get_jk(dms, kpts=kpts, exxdiv='vcut_ws')
    vjs, vks = dfs.get_jk(dm_sc)
    vjsw, vksw = dfs.get_jk(dm_sc, exxdiv='vcut_ws')
    assert abs(vjw - vj).max() < 1e-12 and abs(vjsw - vjs).max() < 1e-12
    assert abs(k2gamma.to_supercell_ao_integrals(cell, kpts, vk).real - vks).max() < 2e-4 * abs(vks).max()
    assert abs(k2gamma.to_supercell_ao_integrals(cell, kpts, vkw).real - vksw).max() < 2e-4 * abs(vksw).max()
    assert abs(vksw - vks).max() > 1e-3 * abs(vks).max()


def test_ao2mo_7d_matches_per_quartet_pair_transforms():
    """ao2mo_7d on the reference's own test system (pyscf/pbc/df/test/test_fft.py:817-847: two He, s + p shells, 3 Bohr cube,
    mesh 6^3, k-mesh [1,3,1], random complex MO coefficients with seed 1) equals, quartet by quartet (the fourth k-point from
    momentum conservation modulo a reciprocal lattice vector), the integral assembled from get_mo_pairs_G - the same
    consistency the reference asserts against its ao2mo."""
    from oracle_backend import OracleBackend
    from oracle import pbc_tools as otools
    from pyscf_isdf_amd import gto
    from pyscf_isdf_amd.isdf import ISDF
    cell = gto.Cell(atom=[('He', (2., 2.2, 2.)), ('He', (1.2, 1., 1.))], a=np.eye(3) * 3.0, unit='Bohr', mesh=(6, 6, 6),
                    basis={'He': [[0, (1.2, 1)], [1, (0.6, 1)]]})
    kpts = cell.make_kpts([1, 3, 1])
    nk, nao = len(kpts), cell.nao_nr()
    np.random.seed(1)
    mo = np.random.random((nk, nao, nao)) + np.random.random((nk, nao, nao)) * 1j
    df = ISDF(cell, kpts=kpts, c_isdf=5, select='global', backend=OracleBackend())
    out = df.ao2mo_7d(mo, kpts)
    assert out.shape == (nk, nk, nk, nao, nao, nao, nao) and out.dtype == np.complex128
    kcons = df.get_kconserv(kpts)
    G = int(np.prod(cell.mesh))
    worst = 0.0
    for ki in range(nk):
        for kj in range(nk):
            q = kpts[kj] - kpts[ki]
            coulG = otools.get_coulG(cell.lattice_vectors(), cell.mesh, q) * cell.vol / G ** 2
            pij = df.get_mo_pairs_G((mo[ki], mo[kj]), kpts[[ki, kj]])                  # FFT of conj(i) j exp(-i q.r)
            for kk in range(nk):
                kl = kcons[ki, kj, kk]
                plk = df.get_mo_pairs_G((mo[kl], mo[kk]), kpts[[kl, kk]], q=q)       # FFT of conj(l) k exp(-i q.r) = conj of the kl pair's transform
                ref = (pij.T * coulG).dot(plk.conj()).reshape(nao, nao, nao, nao).transpose(0, 1, 3, 2)
                worst = max(worst, abs(out[ki, kj, kk] - ref).max())
    assert worst < 1e-10


def test_robust_k_at_kpoints_host_logic_and_error_reduction():
    """robust_k at k-points through the host driver on the checker backend (no GPU): K = K1 + K1^H - K_isdf with
    V^q = conv_q(Theta) equals the oracle's direct formula (oracle.kisdf.get_k_robust_kpts) on the same points, and it is much
    closer to the reference's exact k-point exchange (fft_jk.py:250-292, oracle/fftdf.py) than the plain ISDF K at the same P."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he2_triclinic()
    cell.mesh = np.array([9, 9, 9])
    kpts = cell.make_kpts([2, 1, 1])
    nao = cell.nao_nr()
    rng = np.random.default_rng(2)
    c = rng.standard_normal((2, nao, nao)) + 1j * rng.standard_normal((2, nao, nao))
    dms = np.einsum('kpi,kqi->kpq', c[:, :, :2], c[:, :, :2].conj())
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    ao_k = [np.asarray(x, dtype=complex) for x in oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpts, rule='point')]
    k_exact = fftdf.get_jk_kpts(ao_k, dms, cell.lattice_vectors(), cell.mesh, coords, kpts)[1]
    plain = ISDF(cell, kpts=kpts, c_isdf=5, select='global', backend=OracleBackend())
    vk_plain = plain.get_jk(dms, kpts=kpts, with_j=False)[1]
    df = ISDF(cell, kpts=kpts, c_isdf=5, select='global', backend=OracleBackend())
    df.robust_k = True
    vk = df.get_jk(dms, kpts=kpts, with_j=False)[1]
    assert np.array_equal(df.ip, plain.ip)
    theta = kisdf.fit_theta(kisdf.periodic_stack(ao_k, coords, kpts), df.ip, df.reg_used)
    ref = kisdf.get_k_robust_kpts(ao_k, coords, kpts, cell.lattice_vectors(), cell.mesh, df.ip, theta, dms)
    assert abs(vk - ref).max() < 1e-8 * abs(ref).max()
    assert abs(vk - vk.conj().transpose(0, 2, 1)).max() < 1e-10 * abs(vk).max()
    assert abs(vk - k_exact).max() < 0.1 * abs(vk_plain - k_exact).max()


def test_even_mesh_pair_correction_host_logic():
    """W^{-q} = conj(W^q) holds index by index only off the Nyquist planes of an even mesh (the reference's table labels index n/2
    as -n/2 for both signs of q and zeroes it on the wrap-around edge for one of them, pbc.py:272-302).  With the Nyquist-plane terms
    added ('auto') the paired build reproduces the build that makes every W^q from its own table (1e-10); the uncorrected pairing
    of round 2 does not."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he2_triclinic()
    cell.mesh = np.array([8, 10, 9])
    kpts = cell.make_kpts([2, 2, 1])
    nao = cell.nao_nr()
    rng = np.random.default_rng(4)
    c = rng.standard_normal((4, nao, nao)) + 1j * rng.standard_normal((4, nao, nao))
    dms = np.einsum('kpi,kqi->kpq', c[:, :, :2], c[:, :, :2].conj())
    res = {}
    for mode in (False, 'auto', 'uncorrected'):
        df = ISDF(cell, kpts=kpts, c_isdf=4, select='global', backend=OracleBackend())
        df.kpt_pair_q = mode
        res[mode] = df.get_jk(dms, kpts=kpts, with_j=False)[1]
        if mode == 'auto':
            assert df._pair_correct and len(df._Wq) == len(df._qs)          # every q stored: primaries + corrected twins
    scale = abs(res[False]).max()
    assert abs(res['auto'] - res[False]).max() < 1e-10 * scale
    assert abs(res['uncorrected'] - res[False]).max() > 1e-7 * scale
