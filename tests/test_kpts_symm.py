"""k-point symmetry: irreducible k-points pinned to the reference's own numbers
(pyscf/pbc/lib/test/test_kpts_ksymm.py:59-93), the rotations checked on matrices and densities evaluated with the oracle's
collocation on symmetric grids (the content of its test_transform / test_symmetrize_density, which need an SCF object)."""
import warnings
import numpy as np
import scipy.linalg
import pytest
from pyscf_isdf_amd import gto, kpts_symm
from pyscf_isdf_amd._common import tag_array
from oracle import ao as oao, pbc_tools as tools


def _si_cell():
    """test_kpts_ksymm.py:29-42 (the basis does not enter the k-point symmetry; Si is not among the bundled GTH sets)."""
    a = [[0.0, 2.6935121974, 2.6935121974], [2.6935121974, 0.0, 2.6935121974], [2.6935121974, 2.6935121974, 0.0]]
    return gto.Cell(atom='Si 0 0 0; Si 1.3467560987 1.3467560987 1.3467560987', a=a,
                    basis={'Si': [[0, [0.5, 1]], [1, [0.4, 1]]]}, mesh=[20] * 3)


def test_space_group_of_diamond_structure():
    ops = kpts_symm.search_space_group_ops(_si_cell())
    assert len(ops) == 48 and sum(o.trans_is_zero for o in ops) == 24 and ops[0].is_eye
    assert any(o.rot_is_inversion and not o.trans_is_zero for o in ops)           # inversion about the bond centre
    # closure: the product of two operations is an operation (translations modulo the lattice)
    keys = {o._key() for o in ops}
    for o1 in ops[::5]:
        for o2 in ops[::7]:
            prod = kpts_symm.SpaceGroupOp(o1.rot.dot(o2.rot), o1.trans + o2.trans.dot(o1.rot.T))
            assert prod._key() in keys
        inv = o1.inv()
        assert kpts_symm.SpaceGroupOp(inv.rot, inv.trans)._key() in keys


def test_make_kpts_ibz_matches_reference_pins():
    """test_kpts_ksymm.py:59-93: counts and fingerprints of kpts_ibz for the 16^3 mesh of the Si cell."""
    cell = _si_cell()
    kmesh = [16] * 3
    kpts = cell.make_kpts(kmesh, space_group_symmetry=True)
    assert kpts.nkpts_ibz == 145
    for star, star_op in zip(kpts.stars, kpts.stars_ops):
        for i, k in enumerate(star):
            assert star_op[i] == kpts.stars_ops_bz[k]
    assert abs(tools.fp(kpts.kpts_ibz) - 2.211640884021115) < 1e-9
    assert abs(kpts.weights_ibz.sum() - 1) < 1e-12 and len(kpts) == 145
    kpts1 = cell.make_kpts(kmesh, space_group_symmetry=True, time_reversal_symmetry=True, symmorphic=True)
    assert kpts1.time_reversal and abs(kpts1.kpts_ibz - kpts.kpts_ibz).max() < 1e-9
    kpts2 = cell.make_kpts(kmesh, space_group_symmetry=True, time_reversal_symmetry=False, symmorphic=True)
    assert kpts2.nkpts_ibz == 245 and abs(tools.fp(kpts2.kpts_ibz) - -2.0196383066365353) < 1e-9
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')              # the shifted mesh has lower symmetry than the lattice (the reference warns too)
        kpts3 = cell.make_kpts(kmesh, with_gamma_point=False, space_group_symmetry=True)
        kpts4 = cell.make_kpts(kmesh, with_gamma_point=False, space_group_symmetry=True, symmorphic=True)
    assert kpts3.nkpts_ibz == 408 and abs(tools.fp(kpts3.kpts_ibz) - -2.581114561328012) < 1e-9
    assert kpts4.nkpts_ibz == 816 and abs(tools.fp(kpts4.kpts_ibz) - -1.124492399508386) < 1e-9
    kpts5 = cell.make_kpts(kmesh, time_reversal_symmetry=True)
    assert kpts5.nkpts_ibz == 2052
    # the docstring example of kpts.py:789-801
    he = gto.Cell(atom='He 0 0 0', a=np.eye(3) * 2.0, basis={'He': [[0, [1.0, 1]]]}, mesh=[8] * 3)
    k = kpts_symm.make_kpts(he, np.array([[0, 0, 0], [.5, 0, 0], [0, .5, 0], [0, 0, .5]]).dot(he.reciprocal_vectors()),
                            space_group_symmetry=True)
    assert k.nkpts_ibz == 2 and abs(k.kpts_scaled_ibz - np.array([[0, 0, 0], [0, 0, .5]])).max() < 1e-12


def test_D_matrices_are_an_orthogonal_representation():
    rng = np.random.default_rng(3)
    def rot():
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        return q
    for l in range(4):
        R1, R2 = rot(), -rot()
        D1, D2, D12 = (kpts_symm.rotation_Dmat(R, l) for R in (R1, R2, R1.dot(R2)))
        assert D1.shape == (2 * l + 1, 2 * l + 1)
        assert abs(D1.T.dot(D1) - np.eye(2 * l + 1)).max() < 1e-12
        assert abs(D1.dot(D2) - D12).max() < 1e-12
        assert abs(kpts_symm.rotation_Dmat(-np.eye(3), l) - (-1) ** l * np.eye(2 * l + 1)).max() < 1e-13


def _matrices(cell, kp):
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    aos = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kp.kpts)
    w = cell.vol / len(coords)
    v = np.zeros(len(coords))
    for ra in cell.atom_coords():
        for L in Ls:
            d = coords - (ra + L)
            v += np.exp(-0.8 * np.einsum('gx,gx->g', d, d))
    S = np.array([w * ao.conj().T.dot(ao) for ao in aos])
    V = np.array([w * (ao.conj().T * v).dot(ao) for ao in aos])
    return aos, S, V


def _cell_hex_spdf():
    a = np.array([[3.2, 0, 0], [-1.6, 1.6 * np.sqrt(3), 0], [0, 0, 5.1]])
    return gto.Cell(atom=[['He', (0, 0, 0)], ['C', tuple(np.array([1 / 3, 2 / 3, 0.5]).dot(a))]], a=a, mesh=[12, 12, 20],
                    basis={'He': [[0, [0.9, 1]], [3, [0.8, 1]]], 'C': [[1, [0.7, 1]], [2, [0.9, 1]], [3, [1.1, 1]]]})


@pytest.mark.parametrize('case', ['diamond_nonsymmorphic', 'diamond_symmorphic_trs', 'diamond_symmorphic', 'trs_only', 'hex_spdf'])
def test_operators_density_matrices_and_densities_rotate_to_the_full_zone(case):
    """What the reference's test_transform / test_symmetrize_density assert with an SCF object (test_kpts_ksymm.py:95-143), on
    matrices made here: overlap and a symmetric local potential in the Bloch AO basis at every k-point of the zone (s - f
    shells, fractional translations, time reversal), smeared density matrices from their eigenvectors, and the density."""
    if case.startswith('diamond'):
        cell = gto.diamond_primitive('gth-szv' if case != 'diamond_nonsymmorphic' else 'gth-dzvp', (16, 16, 16))
        kw = {'diamond_nonsymmorphic': dict(space_group_symmetry=True, time_reversal_symmetry=True),
              'diamond_symmorphic_trs': dict(space_group_symmetry=True, time_reversal_symmetry=True, symmorphic=True),
              'diamond_symmorphic': dict(space_group_symmetry=True, symmorphic=True)}[case]
        kp = cell.make_kpts([3, 3, 3], **kw)
        assert (kp.nop, kp.time_reversal, kp.nkpts_ibz) == {'diamond_nonsymmorphic': (48, False, 4), 'diamond_symmorphic_trs': (24, True, 4),
                                                           'diamond_symmorphic': (24, False, 5)}[case]
    elif case == 'trs_only':
        cell = gto.diamond_primitive('gth-szv', (16, 16, 16))
        kp = cell.make_kpts([3, 2, 1], time_reversal_symmetry=True)
        assert kp.nkpts_ibz == 4 and kp.time_reversal
    else:
        cell = _cell_hex_spdf()
        kp = cell.make_kpts([3, 3, 2], space_group_symmetry=True, time_reversal_symmetry=True)
        assert kp.nop == 12 and kp.nkpts_ibz == 6
    assert abs(kp.weights_ibz.sum() - 1) < 1e-12
    aos, S, V = _matrices(cell, kp)
    for M in (S, V):
        assert abs(kp.transform_fock(M[kp.ibz2bz]) - M).max() < 1e-12
    assert abs(kp.transform_1e_operator(np.stack([S, V])[:, kp.ibz2bz]) - np.stack([S, V])).max() < 1e-12      # sets in front
    if case == 'diamond_nonsymmorphic':
        return                                                   # (dzvp: the generalised eigenproblem below is ill conditioned)
    dms, mos, occs = [], [], []
    for k in range(kp.nkpts):
        e, c = scipy.linalg.eigh(V[k] - 0.3 * S[k].dot(S[k]), S[k])
        f = 2 / (1 + np.exp((e - e.mean()) / 0.05))
        dms.append((c * f).dot(c.conj().T)); mos.append(c); occs.append(f)
    dms = np.array(dms)
    dm_bz = kp.transform_dm(dms[kp.ibz2bz])
    assert abs(dm_bz - dms).max() < 1e-9
    # rotated orbitals span the same density matrices; occupations and energies follow their representative
    occ_ibz = kp.check_mo_occ_symmetry(occs, tol=1e-8)
    mo_bz = kp.transform_mo_coeff([mos[k] for k in kp.ibz2bz])
    occ_bz = kp.transform_mo_occ(occ_ibz)
    for k in range(kp.nkpts):
        assert abs((mo_bz[k] * occ_bz[k]).dot(mo_bz[k].conj().T) - dms[k]).max() < 1e-9
    tagged = kp.transform_dm(tag_array(dms[kp.ibz2bz], mo_coeff=[mos[k] for k in kp.ibz2bz], mo_occ=occ_ibz))
    assert len(tagged.mo_coeff) == kp.nkpts and len(tagged.mo_occ) == kp.nkpts
    assert abs(kp.dm_at_ref_cell(dms[kp.ibz2bz]).imag).max() < 1e-9
    # density: sum over the zone = symmetrised sum over the irreducible k-points
    rho = sum(np.einsum('gi,ij,gj->g', aos[k], dms[k], aos[k].conj()) for k in range(kp.nkpts)).real / kp.nkpts
    rs = 0
    for ki, k in enumerate(kp.ibz2bz):
        rs = rs + kp.symmetrize_density(np.einsum('gi,ij,gj->g', aos[k], dms[k], aos[k].conj()), ki, cell.mesh)
    rs = rs / kp.nkpts
    assert abs(rs.imag).max() < 1e-10 and abs(rs.real - rho).max() < 1e-10
    broken = [o.copy() for o in occs]
    star = max(kp.stars, key=len)
    if len(star) > 1:
        broken[star[-1]][0] += 0.1
        with pytest.raises(RuntimeError):
            kp.check_mo_occ_symmetry(broken)


def test_mesh_incompatible_operations_are_dropped_and_reported():
    cell = gto.diamond_primitive('gth-szv', (18, 18, 18))            # 18 is not a multiple of 4: the glide operations go
    with pytest.warns(UserWarning, match='not compatible'):
        kp = cell.make_kpts([2, 2, 2], space_group_symmetry=True)
    assert kp.nop == 24 and all(o.trans_is_zero for o in kp.ops)
