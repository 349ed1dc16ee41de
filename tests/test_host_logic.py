"""Host-side logic (no GPU): cell tables, grid partition, shape conventions, argument errors."""
import numpy as np
import pytest
import cells
from pyscf_isdf_amd import gto
from pyscf_isdf_amd.isdf import ISDF, partition_grid_by_atom
from oracle import isdf as oisdf


def test_cp2k_basis_parser_shell_layout():
    cell = cells.cell_diamond_prim('gth-dzvp')
    # C DZVP-GTH: s(2 ctr, 4 prim), p(2 ctr, 4 prim), d(1 ctr, 1 prim) -> 2 + 6 + 5 = 13 AOs per atom
    assert cell.nao_nr() == 26 and cell.nbas == 6
    assert [cell.bas_angular(i) for i in range(3)] == [0, 1, 2]
    assert [cell.bas_nctr(i) for i in range(3)] == [2, 2, 1]
    assert cell.bas_exp(0)[0] > cell.bas_exp(0)[-1]          # descending exponents (mole.py:995-1000)
    assert cell.nelectron == 8                                 # GTH-PADE q4 per carbon
    szv = cells.cell_diamond_prim('gth-szv')
    assert szv.nao_nr() == 8


def test_supercell_counts_match_baseline_table():
    c = gto.diamond_supercell(2, 'gth-dzvp', (8, 8, 8))
    assert c.natm == 16 and c.nao_nr() == 208                 # BASELINE.md section 2, cfg 2
    assert abs(c.vol - 8 * gto.diamond_primitive().vol) < 1e-9


def test_grid_order_and_Gv_follow_fftfreq():
    cell = cells.cell_he2_triclinic()
    coords = cell.get_uniform_grids([3, 4, 5])
    a = cell.lattice_vectors()
    # index (1, 0, 0) is +1/3 a0, index (2,0,0) wraps to -1/3 a0 (cell.py:889-893)
    assert np.allclose(coords[1 * 20], a[0] / 3) and np.allclose(coords[2 * 20], -a[0] / 3)
    Gv = cell.get_Gv([3, 4, 5])
    b = cell.reciprocal_vectors()
    assert np.allclose(Gv[1], b[2]) and np.allclose(Gv[3], -2 * b[2])
    assert np.allclose(a.dot(b.T), 2 * np.pi * np.eye(3))


def test_partition_matches_bruteforce_oracle():
    cell = cells.cell_diamond_prim('gth-szv', (10, 10, 10))
    coords = cell.get_uniform_grids()
    a = cell.lattice_vectors()
    own = partition_grid_by_atom(coords, cell.atom_coords(), a)
    ref = oisdf.partition_by_atom(coords, cell.atom_coords(), a)
    assert np.array_equal(own, ref)
    assert set(np.unique(own)) == {0, 1}


def test_isdf_surface_and_errors():
    cell = cells.cell_he_c()
    df = ISDF(cell)
    for name in ('build', 'reset', 'dump_flags', 'check_sanity', 'get_jk', 'get_naoaux', 'update_mf', 'get_ao_eri', 'get_eri',
                 'ao2mo', 'get_mo_eri', 'get_ao_pairs_G', 'get_ao_pairs', 'get_mo_pairs_G', 'get_mo_pairs', 'loop', 'ao2mo_7d',
                 'get_nuc', 'get_pp', 'range_coulomb', 'to_gpu'):          # pyscf/pbc/df/fft.py:298-359
        assert callable(getattr(df, name))
    assert list(df.mesh) == [21, 21, 21] and df.grids.weights.shape == (9261,)
    assert abs(df.grids.weights.sum() - cell.vol) < 1e-9
    with pytest.raises(NotImplementedError):                       # range separation: exxdiv=None only
        ISDF(cell, kpts=np.array([[0.1, 0., 0.], [0., 0., 0.]])).get_jk(np.zeros((2, 6, 6)), omega=0.3, exxdiv='ewald')
    with pytest.raises(NotImplementedError):
        df.get_jk(np.eye(6), exxdiv='no-such-treatment')
    with pytest.raises(NotImplementedError):
        ISDF(cell, kpts=np.array([[0.1, 0., 0.], [0., 0., 0.]])).get_jk(np.zeros((2, 6, 6)), exxdiv='no-such-treatment')


def test_unique_q():
    from pyscf_isdf_amd import pbc_tools
    cell = cells.cell_he2_triclinic()
    kpts = cell.make_kpts([2, 2, 1])
    qs, idx = pbc_tools.unique_q(kpts)
    assert idx.shape == (4, 4) and len(qs) == 9
    for i1 in range(4):
        for i2 in range(4):
            assert abs(qs[idx[i1, i2]] - (kpts[i2] - kpts[i1])).max() < 1e-12


def test_madelung_simple_cubic_textbook_value():
    """The reference has no numeric pin for madelung (test_pbc.py:172-183 only checks supercell vs
    k-mesh consistency); pin to the textbook simple-cubic value 2.837297479480620 / L and to the
    reference's own consistency property."""
    L = 3.7
    cell = gto.Cell(atom='He 0 0 0', a=np.eye(3) * L, basis={'He': [[0, [1.0, 1.0]]]}, mesh=[5] * 3, unit='B')
    assert abs(gto.madelung(cell) * L - 2.837297479480620) < 1e-10
    fcc = gto.Cell(atom='He 0 0 0', a=np.array([[0, 1.7834, 1.7834], [1.7834, 0, 1.7834], [1.7834, 1.7834, 0]]),
                   basis={'He': [[0, [1.0, 1.0]]]}, mesh=[5] * 3)
    sup = gto.super_cell(fcc, [2, 3, 5], mesh=[5] * 3)
    assert abs(gto.madelung(sup) - gto.madelung(fcc, nk=(2, 3, 5))) < 1e-9


def test_reset_drops_every_view_of_the_build_and_rebuilds():
    """reset() must not leave a stale fit state pointing at the old (214 GiB at headline size) fit buffer; a rebuild after
    reset gives the same answer."""
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-szv', (10, 10, 10))
    nao = cell.nao_nr()
    rng = np.random.default_rng(0)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    df = ISDF(cell, c_isdf=3, select='local', backend=OracleBackend())
    df.robust_k = True
    k0 = df.get_jk(dm, with_j=False)[1]
    assert df._V is not None and df._bufs
    df.reset()
    for name in ('ao', 'aoP', 'W', 'ip', '_fit_state', '_kfit_state', '_V', '_Wq', '_aoP_k', '_k_built', '_band_built'):
        assert getattr(df, name) is None, name
    assert df._bufs == {} and df._W_omega == {} and not df._built
    assert abs(df.get_jk(dm, with_j=False)[1] - k0).max() < 1e-12


def test_switching_kpoint_sets_does_not_reuse_range_separated_Wq():
    """get_ao_eri on a new k-point set rebuilds through build(): the range-separated W^q cached for the OLD set (keyed by
    omega only) must be gone, so get_jk(omega) on the new set equals a fresh object's."""
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he2_triclinic()
    cell.mesh = np.array([9, 9, 9])
    nao = cell.nao_nr()
    kA = cell.make_kpts([2, 1, 1])
    kB = np.array([[0.11, 0.07, -0.05]])
    rng = np.random.default_rng(4)
    dA = rng.standard_normal((2, nao, nao)); dA = dA + dA.transpose(0, 2, 1)
    dB = rng.standard_normal((1, nao, nao)); dB = dB + dB.transpose(0, 2, 1)
    df = ISDF(cell, kpts=kA, c_isdf=6, select='global', backend=OracleBackend())
    df.get_jk(dA, kpts=kA, omega=0.5)
    assert list(df._W_omega) == [0.5]
    df.get_ao_eri(kB)
    assert df._W_omega == {} and df._k_built.shape == (1, 3)
    vk = df.get_jk(dB, kpts=kB, omega=0.5, with_j=False)[1]
    fresh = ISDF(cell, kpts=kB, c_isdf=6, select='global', backend=OracleBackend())
    assert abs(vk - fresh.get_jk(dB, kpts=kB, omega=0.5, with_j=False)[1]).max() < 1e-12


def test_occupied_orbitals_of_a_density_matrix():
    """ISDF._occupied_orbitals: the occupied space get_jk fits when pair_space='occ' - from the mo_coeff / mo_occ tag exactly as
    the reference's K takes it (pyscf/pbc/df/fft_jk.py:206-210: mo_coeff[:, mo_occ > 0] * sqrt(mo_occ)), for one matrix and for a
    stack (UHF-like tags); from the eigenvectors of an untagged symmetric positive semidefinite matrix; None (-> AO pairs) for
    anything without such a form."""
    cell = cells.cell_he_c()
    df = ISDF(cell)
    nao = cell.nao_nr()
    rng = np.random.default_rng(0)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:2] = 2

    class Tagged(np.ndarray):
        pass
    dm = (c * occ).dot(c.T)
    t = dm.view(Tagged)
    t.mo_coeff, t.mo_occ = c, occ
    o = df._occupied_orbitals(t)
    assert o.shape == (nao, 2) and abs(o.dot(o.T) - dm).max() < 1e-14
    assert abs(o - c[:, :2] * np.sqrt(2.0)).max() < 1e-14
    # a stack of two matrices with their own orbitals: side by side
    occ_b = np.zeros(nao); occ_b[1:2] = 1
    pair = np.stack([dm, (c * occ_b).dot(c.T)]).view(Tagged)
    pair.mo_coeff, pair.mo_occ = np.stack([c, c]), np.stack([occ, occ_b])
    o2 = df._occupied_orbitals(pair)
    assert o2.shape == (nao, 3) and abs(o2.dot(o2.T) - (dm + (c * occ_b).dot(c.T))).max() < 1e-14
    # untagged: eigenvectors; the projector is what matters
    o3 = df._occupied_orbitals(dm)
    assert o3.shape == (nao, 2) and abs(o3.dot(o3.T) - dm).max() < 1e-12
    # no occupied-orbital form: indefinite, non-symmetric, complex, or more than N/2 orbitals
    assert df._occupied_orbitals(dm - 0.1 * np.eye(nao)) is None
    assert df._occupied_orbitals(dm + np.triu(np.ones((nao, nao)), 1) * 1e-3) is None
    assert df._occupied_orbitals(dm + 1e-3j * (c.dot(c.T) > 0)) is None
    assert df._occupied_orbitals(np.eye(nao)) is None
