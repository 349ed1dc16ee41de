"""GPU parity tests: every stage of libmi355_isdf.so (through the C ABI) against the CPU oracle on
the same seeded inputs.  Tolerances are stated per test; FP64 throughout."""
import numpy as np
import pytest
import cells
from pyscf_isdf_amd import gto
from oracle import ao as oao, isdf as oisdf, fftdf, pbc_tools as tools

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    from pyscf_isdf_amd.backend import HipBackend
    return HipBackend(0)


def _oracle_ao(cell, coords=None):
    if coords is None:
        coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    ao = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, rule='point')
    return np.ascontiguousarray(ao.T), coords, Ls, rcut


def _gpu_ao(be, cell, coords, Ls, rcut):
    G = len(coords)
    ao = be.empty((cell.nao_nr(), G))
    be.eval_ao(cell._atm, cell._bas, cell._env, Ls, rcut, be.to_device(np.ascontiguousarray(coords.T)), ao)
    return ao


@pytest.mark.parametrize('mk', [cells.cell_he_c, cells.cell_he2_triclinic, cells.cell_c2_ccpvdz,
                                lambda: cells.cell_diamond_prim('gth-dzvp', (12, 12, 12))])
def test_eval_ao(be, mk):
    """S1: collocation (s, p, d shells; cubic, triclinic and fcc lattices).  Tolerance 1e-12 absolute
    (values are O(1); the only difference allowed is the last-ulp of exp)."""
    cell = mk()
    ref, coords, Ls, rcut = _oracle_ao(cell)
    got = be.to_host(_gpu_ao(be, cell, coords, Ls, rcut))
    assert got.shape == ref.shape
    assert abs(got - ref).max() < 1e-12


def _cell_spdf():
    return gto.Cell(atom='He 0.3 0.1 0.2; He 1.9 2.2 1.4', basis={'He': [[0, [1.1, 1.0]], [1, [0.9, 1.0]], [2, [1.3, 1.0]],
                                                                     [3, [1.2, 0.7], [0.5, 0.4]]]},
                    a=np.array([[4.2, 0.3, 0.], [0.1, 4.0, 0.2], [0.4, 0., 4.4]]), mesh=[15, 14, 16], unit='B')


def test_eval_ao_f_shells_values_derivatives_and_kpoints(be):
    """l = 3 (f shells, contracted, triclinic cell) through all four collocation kernels against the oracle (1e-12): values,
    Cartesian first derivatives, Bloch sums at a k-point and their derivatives.  The oracle's f combination is pinned by
    orthonormality and finite differences (tests/test_oracle_pins.py) - the reference holds no f-shell fixture on this path."""
    import torch
    cell = _cell_spdf()
    assert cell.nao_nr() == 2 * 16
    ref, coords, Ls, rcut = _oracle_ao(cell)
    got = be.to_host(_gpu_ao(be, cell, coords, Ls, rcut))
    assert abs(got - ref).max() < 1e-12
    G, nao = len(coords), cell.nao_nr()
    d_c = be.to_device(np.ascontiguousarray(coords.T))
    ao4 = be.empty((4, nao, G))
    be.eval_ao_deriv1(cell._atm, cell._bas, cell._env, Ls, rcut, d_c, ao4)
    ref4 = oao.eval_ao_deriv1(cell._atm, cell._bas, cell._env, coords, Ls, rcut)
    assert abs(be.to_host(ao4) - ref4.transpose(0, 2, 1)).max() < 1e-11
    kpt = np.array([0.21, -0.13, 0.34])
    ur, ui = be.empty((nao, G)), be.empty((nao, G))
    be.eval_ao_k(cell._atm, cell._bas, cell._env, Ls, rcut, kpt, False, d_c, ur, ui)
    refk = np.asarray(oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpt.reshape(1, 3), rule='point')[0])
    assert abs(be.to_host(ur) + 1j * be.to_host(ui) - refk.T).max() < 1e-12
    buf = be.empty((2, 4, nao, G))
    be.eval_ao_k_deriv1(cell._atm, cell._bas, cell._env, Ls, rcut, kpt, False, d_c, buf[0], buf[1])
    refk4 = oao.eval_ao_deriv1(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpt=kpt)
    assert abs(be.to_host(buf[0]) + 1j * be.to_host(buf[1]) - refk4.transpose(0, 2, 1)).max() < 1e-11


def test_eval_ao_ragged_tail_and_permuted_coords(be):
    """Grid size not a multiple of the workgroup (9261 = 36*256 + 45) and a shuffled point order."""
    cell = cells.cell_he_c()
    coords = cell.get_uniform_grids()
    perm = np.random.default_rng(3).permutation(len(coords))[:5000]
    ref, c2, Ls, rcut = _oracle_ao(cell, coords[perm])
    got = be.to_host(_gpu_ao(be, cell, c2, Ls, rcut))
    assert abs(got - ref).max() < 1e-12


def test_gather_cols(be):
    rng = np.random.default_rng(0)
    src = rng.standard_normal((7, 1000))
    idx = rng.permutation(1000)[:333].astype(np.int64)
    dst = be.empty((7, 333))
    be.gather_cols(be.to_device(src), be.to_device(idx), dst)
    assert np.array_equal(be.to_host(dst), src[:, idx])


@pytest.mark.parametrize('mk', [cells.cell_he_c, cells.cell_he2_triclinic, cells.cell_c2_ccpvdz])
def test_eval_ao_deriv1(be, mk):
    """isdf_eval_ao_deriv1 (values + Cartesian gradients, s / p / d shells) against the oracle (1e-12 of the largest entry) and,
    on the reference's own cell, against its constant fp(ao, deriv=1) = 8.8004405892746433 (test_numint.py:98-100)."""
    import torch
    cell = mk()
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    ref = oao.eval_ao_deriv1(cell._atm, cell._bas, cell._env, coords, Ls, rcut)          # (4, G, nao)
    G, nao = len(coords), cell.nao_nr()
    ao4 = be.to_device(np.full((4, nao, G + 3), 5.0))
    be.eval_ao_deriv1(cell._atm, cell._bas, cell._env, Ls, rcut, be.to_device(np.ascontiguousarray(coords.T)), ao4)
    got = be.to_host(ao4)
    assert (got[:, :, G:] == 5.0).all()
    got = got[:, :, :G].transpose(0, 2, 1)
    assert abs(got - ref).max() < 1e-12 * max(1.0, abs(ref).max())
    ao0 = _gpu_ao(be, cell, coords, Ls, rcut)
    assert abs(be.to_host(ao0).T - got[0]).max() < 1e-14
    if mk is cells.cell_c2_ccpvdz:
        from oracle import pbc_tools as otools
        assert abs(otools.fp(got) - 8.8004405892746433) < 1e-8


def test_partition_by_atom_matches_oracle(be):
    """isdf_partition_by_atom (device Voronoi partition, minimum image, ties to the lowest atom index) == the oracle's brute
    force on an fcc cell full of exact ties, on a triclinic cell, and on a supercell (bit-exact integer output)."""
    from pyscf_isdf_amd._common import partition_grid_by_atom
    for cell in (cells.cell_diamond_prim('gth-szv', (12, 12, 12)), cells.cell_he2_triclinic(), gto.diamond_supercell(2, 'gth-szv', (16, 16, 16))):
        coords = cell.get_uniform_grids()
        a = cell.lattice_vectors()
        got = be.partition_by_atom(coords, cell.atom_coords(), a)
        assert got.dtype == np.int32 and got.shape == (len(coords),)
        assert np.array_equal(got, oisdf.partition_by_atom(coords, cell.atom_coords(), a))
        assert np.array_equal(got, partition_grid_by_atom(coords, cell.atom_coords(), a))


def _select_gpu(be, aoT, blk_off, nip, tie_rtol=1e-10, tol=-1.0):
    import torch
    kmax = int(max(nip))
    m = aoT.shape[1]
    L = be.zeros((kmax, m))
    piv = be.empty((len(nip), kmax), dtype=torch.int64)
    rank = be.select_ip(be.to_device(aoT), blk_off, nip, tol, tie_rtol, L, piv)
    return rank, be.to_host(piv), be.to_host(L), L, piv


def test_select_ip_global_matches_oracle(be):
    """S2 on an asymmetric cell: identical pivot list; Cholesky rows within 1e-9 relative to max|L|."""
    cell = cells.cell_he_c()
    aoT = _oracle_ao(cell)[0]
    k = 18
    piv_ref, L_ref = oisdf.select_ip(aoT, k)
    rank, piv, L, _, _ = _select_gpu(be, aoT, [0, aoT.shape[1]], [k])
    assert rank[0] == len(piv_ref)
    assert np.array_equal(piv[0, :rank[0]], piv_ref)
    assert abs(L[:rank[0]] - L_ref).max() < 1e-9 * abs(L_ref).max()


def test_select_ip_rank_deficient_stops(be):
    """6 AOs -> 21 independent pair products: asking for 40 points must stop at rank 21 like
    pivoted_cholesky_python's tolerance rule (scipy_helper.py:88-99)."""
    cell = cells.cell_he_c()
    aoT = _oracle_ao(cell)[0]
    piv_ref, L_ref = oisdf.select_ip(aoT, 40)
    rank, piv, L, _, _ = _select_gpu(be, aoT, [0, aoT.shape[1]], [40])
    assert len(piv_ref) == 21 and rank[0] == 21
    assert np.array_equal(piv[0, :21], piv_ref)


def test_select_ip_blocks_match_oracle(be):
    """Batched blocks of unequal size (incl. one smaller than a workgroup and an empty request)."""
    rng = np.random.default_rng(5)
    nao = 9
    sizes = [700, 130, 1025, 64]
    nip = [25, 10, 0, 30]
    blk_off = np.append(0, np.cumsum(sizes))
    aoT = rng.standard_normal((nao, blk_off[-1])) * np.exp(-rng.random(blk_off[-1]) * 3)
    rank, piv, L, _, _ = _select_gpu(be, aoT, blk_off, nip)
    for b in range(4):
        pr, Lr = oisdf.select_ip(aoT[:, blk_off[b]:blk_off[b + 1]], nip[b]) if nip[b] else (np.zeros(0, int), None)
        assert rank[b] == len(pr)
        assert np.array_equal(piv[b, :rank[b]], pr)
        if nip[b]:
            got = L[:rank[b], blk_off[b]:blk_off[b + 1]]
            assert abs(got - Lr).max() < 1e-9 * abs(Lr).max()


def test_select_ip_tie_rule_lowest_index(be):
    """Exactly duplicated columns tie to the last bit: the lowest index must win on every step."""
    rng = np.random.default_rng(11)
    base = rng.standard_normal((5, 300))
    aoT = np.concatenate([base, base, base], axis=1)        # columns i, i+300, i+600 identical
    rank, piv, L, _, _ = _select_gpu(be, aoT, [0, 900], [12])
    assert rank[0] == 12 and (piv[0, :12] < 300).all()
    piv_ref, _ = oisdf.select_ip(aoT, 12)
    assert np.array_equal(piv[0, :12], piv_ref)


def _select_gram_gpu(be, A, nip, tie_rtol=1e-10, tol=-1.0, panel=0):
    import torch
    piv = be.empty((nip,), dtype=torch.int64)
    dA = be.to_device(np.ascontiguousarray(A))
    r = be.select_ip_gram(dA, nip, tol, tie_rtol, piv, panel=panel)
    return r, be.to_host(piv)[:r], be.to_host(dA)


@pytest.mark.parametrize('panel', [0, 16, 7])
def test_select_ip_gram_matches_reference_golden(be, panel):
    """Refined stage on the explicit Gram matrix, through the C ABI, against the REFERENCE's own output
    (tests/golden/pivoted_cholesky_golden.json, made by pyscf/lib/scipy_helper.py:71-110 on the same matrices,
    tie_rtol = 0 = the reference rule): identical pivot lists and ranks — for one panel and for several panels with
    trailing updates in between."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pivoted_cholesky_golden.json')) as f:
        gold = json.load(f)
    for case in gold['cases']:
        rng = np.random.default_rng(case['seed'])
        ao = rng.standard_normal((case['nao'], case['m'])) * np.exp(-3.0 * rng.random(case['m']))
        r, piv, _ = _select_gram_gpu(be, ao.T.dot(ao) ** 2, case['m'], tie_rtol=0.0, panel=panel)
        assert r == case['rank']
        assert list(piv) == case['piv']


def test_select_ip_gram_matches_oracle_ragged_and_ties(be):
    """m not a multiple of the workgroup (1000), 300 pivots in panels of 64 (four trailing updates): identical pivots
    to the oracle restatement and the same residual matrix where no pivot has been taken; a rank-deficient matrix
    stops at its rank; exactly duplicated columns: the lowest index wins on every step."""
    rng = np.random.default_rng(21)
    ao = rng.standard_normal((40, 1000)) * np.exp(-2.0 * rng.random(1000))
    A = ao.T.dot(ao) ** 2
    pr, Lr = oisdf.pivoted_cholesky_gram(A, 300)
    r, piv, _ = _select_gram_gpu(be, A, 300, panel=64)
    assert r == 300 and np.array_equal(piv, pr)
    ao6 = rng.standard_normal((6, 700))                       # 21 independent pair products
    pr, _ = oisdf.pivoted_cholesky_gram(ao6.T.dot(ao6) ** 2, 50)
    r, piv, _ = _select_gram_gpu(be, ao6.T.dot(ao6) ** 2, 50, panel=8)
    assert len(pr) == 21 and r == 21 and np.array_equal(piv, pr)
    base = rng.standard_normal((5, 300))
    dup = np.concatenate([base, base, base], axis=1)
    r, piv, _ = _select_gram_gpu(be, dup.T.dot(dup) ** 2, 12, panel=5)
    assert r == 12 and (piv < 300).all()
    assert np.array_equal(piv, oisdf.pivoted_cholesky_gram(dup.T.dot(dup) ** 2, 12)[0])


def test_select_ip_gram_block_lower_update_several_strips(be):
    """m = 4500 > the 2048-column strips of the trailing update: only the block-lower part of the matrix is kept current and
    pivot columns are read through the symmetric rule (row p where it lies in or below the column's strip, column p
    otherwise).  700 pivots in panels of 256 (two trailing updates): identical pivots to the oracle."""
    rng = np.random.default_rng(33)
    ao = rng.standard_normal((60, 4500)) * np.exp(-2.0 * rng.random(4500))
    A = ao.T.dot(ao) ** 2
    pr, _ = oisdf.pivoted_cholesky_gram(A, 700)
    r, piv, _ = _select_gram_gpu(be, A, 700)
    assert r == 700 and np.array_equal(piv, pr)


def test_refined_selection_end_to_end_matches_oracle_pipeline():
    """select='refined' on the GPU == the same host driver over the CPU oracle (identical points; K within 1e-9
    relative), on a rattled cell (no symmetry ties) with candidates 2x over-complete."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    nao = cell.nao_nr()
    rng = np.random.default_rng(3)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    out = {}
    for name, backend in (('gpu', None), ('cpu', OracleBackend())):
        df = ISDF(cell, c_isdf=3, select='refined', backend=backend)
        df.refine_over = 2.0
        df.fit_route = 'cholesky'
        out[name] = (df.get_jk(dm, with_j=False)[1], df.ip.copy())
    assert np.array_equal(out['gpu'][1], out['cpu'][1])
    assert abs(out['gpu'][0] - out['cpu'][0]).max() < 1e-9 * abs(out['cpu'][0]).max()


def test_fit_from_chol(be):
    """S3a: Theta = T^-1 L equals the normal-equation fit and is the identity on the points."""
    cell = cells.cell_he_c()
    aoT = _oracle_ao(cell)[0]
    k = 15
    rank, piv, L, dL, dpiv = _select_gpu(be, aoT, [0, aoT.shape[1]], [k])
    be.fit_from_chol(dL, k, aoT.shape[1], dpiv[0].contiguous())
    theta = be.to_host(dL)
    ref = oisdf.fit_theta(*oisdf.select_ip(aoT, k)[::-1])
    assert abs(theta - ref).max() < 1e-9 * abs(ref).max()
    assert abs(theta[:, piv[0]] - np.eye(k)).max() < 1e-10
    ne = oisdf.fit_theta_normal_equations(aoT, piv[0])
    assert abs(theta - ne).max() < 1e-6 * abs(ne).max()


def test_fit_global(be):
    """S3b: Cholesky fit for an arbitrary point set vs numpy lstsq (conditioning-limited: 1e-7 rel)."""
    cell = cells.cell_he_c()
    aoT = _oracle_ao(cell)[0]
    G = aoT.shape[1]
    ip = oisdf.select_ip(aoT, 14)[0]
    ip = np.sort(ip)                                         # arbitrary order
    theta = be.empty((14, G))
    aoP = be.empty((14, aoT.shape[0]))
    reg = be.fit_global(be.to_device(aoT), G, be.to_device(ip), 0.0, theta, aoP)
    assert reg == 0.0
    assert np.array_equal(be.to_host(aoP), aoT[:, ip].T)
    ref = oisdf.fit_theta_normal_equations(aoT, ip)
    assert abs(be.to_host(theta) - ref).max() < 1e-7 * abs(ref).max()


@pytest.mark.parametrize('mk', [cells.cell_he_c, cells.cell_he2_triclinic,
                                lambda: cells.cell_diamond_prim('gth-szv', (12, 10, 8))])
def test_coulomb_W(be, mk):
    """S4+S5 on odd meshes, a triclinic lattice and an even non-orthogonal mesh (Nyquist planes):
    W vs the oracle's complex-FFT construction, 1e-10 relative to max|W|."""
    cell = mk()
    aoT = _oracle_ao(cell)[0]
    G = aoT.shape[1]
    k = min(12, cell.nao_nr() * 2)
    piv, L = oisdf.select_ip(aoT, k)
    theta = oisdf.fit_theta(L, piv)
    a = cell.lattice_vectors()
    ref = oisdf.build_W(theta, a, cell.mesh)
    k = len(piv)
    W = be.empty((k, k))
    for batch in (5, k):                                     # ragged last batch and single batch
        W.zero_()
        be.coulomb_W(be.to_device(theta), cell.mesh, a, 0, k, batch, W)
        assert abs(be.to_host(W) - ref).max() < 1e-10 * abs(ref).max()
    # symmetric shortcut: block-upper part + mirror
    W.zero_()
    be.coulomb_W(be.to_device(theta), cell.mesh, a, 0, k, 5, W, upper_only=True)
    be.symmetrize_upper(W)
    got = be.to_host(W)
    assert abs(got - ref).max() < 1e-10 * abs(ref).max() and abs(got - got.T).max() == 0
    # row range (multi-GPU sharding of P)
    W.zero_()
    be.coulomb_W(be.to_device(theta), cell.mesh, a, 3, 4, 3, W)
    got = be.to_host(W)
    assert abs(got[3:7] - ref[3:7]).max() < 1e-10 * abs(ref).max() and abs(got[:3]).max() == 0 and abs(got[7:]).max() == 0


@pytest.mark.parametrize('mesh,nrow', [((12, 12, 12), 5), ((21, 21, 21), 3), ((10, 9, 8), 37), ((11, 9, 7), 4), ((16, 20, 24), 7),
                                       ((26, 14, 22), 2), ((5, 3, 2), 9), ((1, 4, 6), 3), ((17, 8, 8), 3), ((40, 40, 40), 33),
                                       ((6, 45, 50), 3), ((4, 27, 25), 5), ((3, 120, 120), 2), ((2, 128, 128), 1)])
def test_own_fft_convolution_matches_oracle_and_hipfft(be, mesh, nrow):
    """S4 through the hand-written five-pass FFT (fft_conv.hip: radices 4/2/3/5 and the generic 7/11/13 butterflies, odd and
    even lengths, odd line counts, several tiles) against the oracle's complex FFT + .real on a triclinic lattice
    (<= 1e-12 of the largest value) and against the hipFFT path of the same library (own_fft = 0); (17, 8, 8) has a prime
    factor above 13 and must take the hipFFT fallback with the same answer."""
    rng = np.random.default_rng(sum(mesh) + nrow)
    a = np.array([[4.1, 0.3, -0.2], [0.5, 3.7, 0.4], [-0.3, 0.6, 4.4]])
    G = int(np.prod(mesh))
    rows = rng.standard_normal((nrow, G))
    ref = oisdf.coulomb_V(rows, a, np.asarray(mesh))
    out = {}
    try:
        for own in (2, 1, 0):                        # 2: the three-pass plane form (2-3-5 smooth meshes that fit LDS), else as 1
            be.set_option('own_fft', own)
            d = be.to_device(rows)
            be.coulomb_rows(d, np.asarray(mesh), a, max(1, nrow // 2 + 1))          # two batches, the second one smaller
            out[own] = be.to_host(d)
    finally:
        be.set_option('own_fft', 2)
    scale = abs(ref).max()
    assert abs(out[2] - ref).max() < 1e-12 * scale
    assert abs(out[1] - ref).max() < 1e-12 * scale
    assert abs(out[0] - ref).max() < 1e-12 * scale
    # out of place keeps the input
    d_in = be.to_device(rows)
    d_out = be.empty((nrow, G))
    be.coulomb_rows(d_in, np.asarray(mesh), a, nrow, out=d_out)
    assert np.array_equal(be.to_host(d_in), rows) and abs(be.to_host(d_out) - ref).max() < 1e-12 * scale


def test_get_j_matches_fftdf_pin(be):
    """S6 vs the oracle AND the reference's known-answer fp(vj) (test_fft.py:643-644)."""
    cell = cells.cell_he_c()
    aoT = _oracle_ao(cell)[0]
    G = aoT.shape[1]
    nao = cell.nao_nr()
    rng = np.random.default_rng(1)
    dms = np.stack([np.eye(nao), rng.standard_normal((nao, nao))])
    dms[1] = dms[1] + dms[1].T
    vj = be.empty((2, nao, nao))
    be.get_j(be.to_device(aoT), G, cell.mesh, cell.lattice_vectors(), be.to_device(dms), vj)
    got = be.to_host(vj)
    ref = oisdf.get_j(aoT, dms, cell.lattice_vectors(), cell.mesh)
    assert abs(got - ref).max() < 1e-11
    assert abs(tools.fp(got[0]) - 3.7955873127283377) < 1e-8


def test_get_k(be):
    """S7 vs the oracle on random W (symmetric) and dm; also a row-sharded partial sum."""
    rng = np.random.default_rng(2)
    P, nao = 300, 17
    aoP = rng.standard_normal((P, nao))
    W = rng.standard_normal((P, P)); W = W + W.T
    dms = rng.standard_normal((2, nao, nao))
    vk = be.empty((2, nao, nao))
    be.get_k(be.to_device(aoP), be.to_device(W), 0, P, be.to_device(dms), vk)
    ref = oisdf.get_k(aoP, W, dms)
    assert abs(be.to_host(vk) - ref).max() < 1e-10 * abs(ref).max()
    part = np.zeros_like(ref)
    for r0, nr in ((0, 100), (100, 77), (177, 123)):
        be.get_k(be.to_device(aoP), be.to_device(W), r0, nr, be.to_device(dms), vk)
        part += be.to_host(vk)
    assert abs(part - ref).max() < 1e-10 * abs(ref).max()


@pytest.mark.parametrize('select', ['global', 'local'])
def test_isdf_object_end_to_end(select):
    """The drop-in object: build + get_jk on the reference's He/C fixture.  At full rank (21 pair
    products) ISDF is exact, so K must reproduce the reference's fp(vk) pin (test_fft.py:645)."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    nao = cell.nao_nr()
    df = ISDF(cell, c_isdf=4, select=select)
    dm = np.eye(nao)
    vj, vk = df.get_jk(dm, exxdiv=None)
    assert vj.shape == (nao, nao) and vj.dtype == np.float64 and vk.dtype == np.float64
    assert abs(tools.fp(vj) - 3.7955873127283377) < 1e-8
    if select == 'global':
        assert len(df.ip) == 21
        assert abs(tools.fp(vk) - 4.290076429522121) < 1e-7
    # shapes: (nset, nao, nao) in -> same out; with_k=False -> None
    vj2, vk2 = df.get_jk(np.stack([dm, 2 * dm]), with_k=False)
    assert vk2 is None and vj2.shape == (2, nao, nao) and abs(vj2[1] - 2 * vj).max() < 1e-10
    # complex density matrix at the Gamma point: J, K are linear in D (real AOs) -> complex128 results
    rng = np.random.default_rng(0)
    di = rng.standard_normal((nao, nao))
    vjc, vkc = df.get_jk(dm + 1j * di, hermi=0)
    vji, vki = df.get_jk(di, hermi=0)
    assert vjc.dtype == np.complex128 and abs(vjc - (vj + 1j * vji)).max() < 1e-12 and abs(vkc - (vk + 1j * vki)).max() < 1e-12


def test_isdf_object_vs_oracle_pipeline_diamond():
    """Symmetric crystal (diamond, gth-dzvp, d shells).  The oracle pipeline is fed the GPU's AO
    values so that both sides see bit-identical inputs; pivots must then agree and J/K energies
    agree to 1e-9 Eh (target in BASELINE.json: 1e-6 Eh)."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-dzvp', (12, 12, 12))
    nao = cell.nao_nr()
    df = ISDF(cell, c_isdf=6, select='global')
    df.build()
    aoT = df.backend.to_host(df.ao)
    ref_ao = _oracle_ao(cell)[0]
    assert abs(aoT - ref_ao).max() < 1e-12
    a = cell.lattice_vectors()
    ref = oisdf.build_global(aoT, a, cell.mesh, 6 * nao)
    assert np.array_equal(ref['ip'], df.ip)
    rng = np.random.default_rng(20240203)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:cell.nelectron // 2] = 2
    dm = (c * occ).dot(c.T)
    vj, vk = df.get_jk(dm)
    vj_ref = oisdf.get_j(aoT, dm, a, cell.mesh)
    vk_ref = oisdf.get_k(ref['aoP'], ref['W'], dm)
    assert abs(np.einsum('ij,ji', vj - vj_ref, dm)) / 2 < 1e-9
    assert abs(np.einsum('ij,ji', vk - vk_ref, dm)) / 4 < 1e-9
    assert abs(vj - vj_ref).max() < 1e-9 and abs(vk - vk_ref).max() < 1e-8


@pytest.mark.parametrize('M,N,K,scaled', [(128, 128, 4096, False), (130, 257, 4112, True), (7, 300, 9261, True),
                                          (384, 520, 65536, False), (64, 64, 17, False), (1, 1, 1, True),
                                          # 256-row tile variant (aligned, K % 32 == 0, M fills the tile)
                                          (512, 300, 8192, False), (1000, 130, 65536, True), (256, 128, 32, False)])
def test_gemm_nt_mfma(be, M, N, K, scaled):
    """The FP64 MFMA kernel behind W and vj: C = alpha A (B.*s)^T + beta C against numpy, incl. ragged
    tiles, odd K / unaligned rows (generic path) and multi-slab reduction.  1e-12 relative to |A||B|."""
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((N, K)); C0 = rng.standard_normal((M, N))
    s = rng.standard_normal(K) if scaled else None
    ref = 0.7 * A.dot((B * s).T if scaled else B.T) - 0.3 * C0
    C = be.to_device(C0)
    be.gemm_nt(be.to_device(A), be.to_device(B), C, alpha=0.7, beta=-0.3, kscale=be.to_device(s) if scaled else None)
    scale = np.sqrt(K) * 10
    assert abs(be.to_host(C) - ref).max() < 1e-12 * scale


@pytest.mark.parametrize('M,N,K', [(256, 128, 32), (512, 130, 64), (1000, 1002, 96), (1280, 4226, 32), (2304, 700, 416),
                                   # outside the kernel's shape (row padding, odd K): the rocBLAS route of the same entry point
                                   (300, 128, 64), (512, 200, 40)])
def test_gemm_nn_mfma(be, M, N, K):
    """The opt-in FP64 MFMA NN kernel for the pair-density rows (option gemm_nn_own): C = A B with A (M, K) and B (K, N) row-major, ragged column tiles,
    super-tile padding (4 x 8 tiles per XCD round) and strided operands, against numpy; 1e-12 relative to |A||B|."""
    rng = np.random.default_rng(M + 3 * N + K)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((K, N + 6))
    dB = be.to_device(B)[:, 2:2 + N]                     # a column window: ldb > N, 16-byte aligned start
    C = be.to_device(np.full((M, N + 4), 7.0))
    be.set_option('gemm_nn_own', 1)
    try:
        be.gemm_nn(be.to_device(A), dB, C[:, :N])
    finally:
        be.set_option('gemm_nn_own', 0)
    out = be.to_host(C)
    assert abs(out[:, :N] - A.dot(B[:, 2:2 + N])).max() < 1e-12 * np.sqrt(K) * 10
    assert (out[:, N:] == 7.0).all()                     # nothing written past the last column


def test_pair_gram_rows_squared_epilogue(be):
    """isdf_pair_gram_rows on a shape the own NN kernel takes (512 points, 64 functions): the square is applied in the
    kernel's epilogue; same numbers as the product-then-square of the reference formulation."""
    rng = np.random.default_rng(11)
    P, nao, ng = 512, 64, 3000
    aoP = rng.standard_normal((P, nao)); ao = rng.standard_normal((nao, ng))
    B = be.empty((P, ng))
    be.set_option('gemm_nn_own', 1)
    try:
        be.pair_gram_rows(be.to_device(aoP), be.to_device(ao), ng, B)
    finally:
        be.set_option('gemm_nn_own', 0)
    ref = aoP.dot(ao) ** 2
    assert abs(be.to_host(B) - ref).max() < 1e-11 * abs(ref).max()


def test_W_from_factor_both_kinds(be):
    """W without forming Theta: S^-1 [w conv(Y) Y^T] S^-T must give the same K as the explicit-Theta
    route (W itself may differ in directions K cannot see)."""
    import torch
    cell = cells.cell_diamond_prim('gth-szv', (12, 12, 12))
    aoT = _oracle_ao(cell)[0]
    G = aoT.shape[1]
    a = cell.lattice_vectors()
    nao = cell.nao_nr()
    rng = np.random.default_rng(4)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    # kind 1: selection rows L and T
    k = 20
    piv, L = oisdf.select_ip(aoT, k)
    W_ref = oisdf.build_W(oisdf.fit_theta(L, piv), a, cell.mesh)
    dL = be.to_device(L)
    T = be.empty((k, k))
    be.gather_T(dL, k, be.to_device(piv), T)
    assert abs(be.to_host(T) - np.triu(L[:, piv])).max() == 0
    W = be.empty((k, k))
    be.coulomb_W(dL, cell.mesh, a, 0, k, k, W, upper_only=True)
    be.symmetrize_upper(W)
    be.W_from_factor(T, 1, W)
    aoP = np.ascontiguousarray(aoT[:, piv].T)
    k_ref = oisdf.get_k(aoP, W_ref, dm)
    assert abs(oisdf.get_k(aoP, be.to_host(W), dm) - k_ref).max() < 1e-10 * abs(k_ref).max()
    assert abs(be.to_host(W) - oisdf.W_from_factor(np.triu(L[:, piv]), L, a, cell.mesh)).max() < 1e-9 * abs(W_ref).max()
    # kind 0: Cholesky factor + forward solve only
    ip = np.sort(piv)
    d_ao = be.to_device(aoT)
    aoPd = be.empty((k, nao)); chol = be.empty((k, k)); Y = be.empty((k, G))
    reg = be.fit_prepare(d_ao, be.to_device(ip), 1e-12, aoPd, chol)
    be.fit_apply(chol, aoPd, d_ao, G, Y, forward_only=True)
    be.coulomb_W(Y, cell.mesh, a, 0, k, 7, W, upper_only=True)
    be.symmetrize_upper(W)
    be.W_from_factor(chol, 0, W)
    theta = oisdf.fit_theta_global_chol(aoT, ip, reg)
    W_ref = oisdf.build_W(theta, a, cell.mesh)
    aoP = np.ascontiguousarray(aoT[:, ip].T)
    k_ref = oisdf.get_k(aoP, W_ref, dm)
    assert abs(oisdf.get_k(aoP, be.to_host(W), dm) - k_ref).max() < 1e-10 * abs(k_ref).max()


def test_explicit_theta_and_factor_routes_agree():
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-szv', (12, 12, 12))
    nao = cell.nao_nr()
    dm = np.eye(nao)
    out = []
    for explicit in (True, False):
        for select in ('local', 'global'):
            df = ISDF(cell, c_isdf=5, select=select)
            df.explicit_theta = explicit
            df.fit_route = 'cholesky'
            out.append(df.get_jk(dm)[1])
    assert abs(out[0] - out[2]).max() < 1e-10 and abs(out[1] - out[3]).max() < 1e-10


@pytest.mark.parametrize('route', ['cholesky', 'blockjacobi', 'auto'])
def test_sharded_code_path_on_one_gpu(route):
    """The multi-GPU orchestration (slice collocation, re-evaluated selection blocks, slice fit,
    all-to-all staging, row convolution, partial W, row-sharded K) executed on ONE rank must give
    the single-GPU result; the collectives degenerate to copies."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-dzvp', (12, 12, 12))
    nao = cell.nao_nr()
    rng = np.random.default_rng(9)
    dm = rng.standard_normal((2, nao, nao)); dm = dm + dm.transpose(0, 2, 1)
    ref = ISDF(cell, c_isdf=4, select='local')
    ref.fit_route = route
    vj0, vk0 = ref.get_jk(dm)
    df = ISDF(cell, c_isdf=4, select='local')
    df.fit_route = route
    df.force_sharded = True
    df.fft_batch = 37                                        # ragged row batches
    vj1, vk1 = df.get_jk(dm)
    assert np.array_equal(ref.ip, df.ip)
    if route == 'auto':
        # the probe check runs in both layouts and sees the same numbers
        assert ref.bj_check is not None and abs(ref.bj_check - df.bj_check) < 0.5 * ref.bj_check + 1e-12
        if ref.fit_route_used != df.fit_route_used:          # borderline check value: nothing to compare
            return
    else:
        assert ref.fit_route_used == df.fit_route_used == route
    # the block-Jacobi route amplifies the (layout-dependent) summation order of M' by cond(A')
    tol = 1e-9 if ref.fit_route_used == 'cholesky' else 1e-6
    assert abs(vj0 - vj1).max() < 1e-10 and abs(vk0 - vk1).max() < tol * abs(vk0).max()
    # range separation goes through the same sharded S4/S5
    wj0, wk0 = ref.get_jk(dm, omega=0.3)
    wj1, wk1 = df.get_jk(dm, omega=0.3)
    assert abs(wj0 - wj1).max() < 1e-10 and abs(wk0 - wk1).max() < tol * abs(wk0).max()


def test_sharded_spectral_form_on_one_gpu():
    """The grid-sharded build's spectral form of W (points dealt in whole blocks, rows made on the fly, K slices of X exchanged,
    probe check alongside, no resident rows) executed on ONE rank with the real kernels: same points, K equal to the single-GPU
    spectral build to the route's noise; the range-separated rebuild goes through the same state."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-dzvp', (24, 24, 24))
    nao = cell.nao_nr()
    rng = np.random.default_rng(9)
    dm = rng.standard_normal((2, nao, nao)); dm = dm + dm.transpose(0, 2, 1)

    def make(sharded):
        df = ISDF(cell, c_isdf=6, select='refined')
        df.w_sphere, df.w_spectral_check_tol, df.fft_batch = 0, 3e-8, 40
        df.force_sharded = sharded
        return df
    ref, df = make(False), make(True)
    vj0, vk0 = ref.get_jk(dm)
    vj1, vk1 = df.get_jk(dm)
    assert np.array_equal(ref.ip, df.ip) and ref.fit_route_used == df.fit_route_used == 'blockjacobi'
    assert ref.w_spectral_fraction == df.w_spectral_fraction > 1.0
    assert df._fit_state['kind'] == 'blockjacobi-spectral' and df._fit_state['theta'] is None
    assert ref.bj_check is not None and df.bj_check is not None and df.bj_check <= 3e-8
    assert abs(vj0 - vj1).max() < 1e-10 and abs(vk0 - vk1).max() < 1e-7 * abs(vk0).max()
    wj0, wk0 = ref.get_jk(dm, omega=0.3)
    wj1, wk1 = df.get_jk(dm, omega=0.3)
    assert abs(wj0 - wj1).max() < 1e-10 and abs(wk0 - wk1).max() < 1e-7 * abs(wk0).max()


@pytest.mark.parametrize('omega', [0.4, -0.4, 0.11])
def test_range_separated_get_jk(omega):
    """get_jk(omega=...) (FFTDF.get_jk with range_coulomb, pyscf/pbc/df/fft.py:298-303): the device kernel table carries the
    factor of pyscf/pbc/tools/pbc.py:408-418.  J against the reference formula, K against the oracle's W built with the
    attenuated kernel on the same points, on the triclinic He2 cell (even, non-orthogonal mesh: symmetrised half spectrum);
    long range + short range = full."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he2_triclinic()
    nao = cell.nao_nr()
    rng = np.random.default_rng(2)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    aoT = _oracle_ao(cell)[0]
    a, mesh = cell.lattice_vectors(), cell.mesh
    df = ISDF(cell, c_isdf=4, select='local')
    df.fit_route = 'cholesky'
    vj0, vk0 = df.get_jk(dm)
    vj, vk = df.get_jk(dm, omega=omega)
    assert vj.dtype == np.float64 and abs(vj - oisdf.get_j(aoT, dm, a, mesh, omega=omega)).max() < 1e-10
    th = oisdf.fit_theta_global_chol(aoT, df.ip, reg_rel=df.reg_used)
    k_or = oisdf.get_k(np.ascontiguousarray(aoT[:, df.ip].T), oisdf.build_W(th, a, mesh, omega=omega), dm)
    assert abs(vk - k_or).max() < 1e-8 * abs(k_or).max()
    vj2, vk2 = df.get_jk(dm, omega=-omega)
    assert abs(vj + vj2 - vj0).max() < 1e-11 and abs(vk + vk2 - vk0).max() < 1e-9 * abs(vk0).max()
    assert sorted(df._W_omega) == sorted([round(omega, 10), round(-omega, 10)])
    auto = ISDF(cell, c_isdf=4, select='local')                       # block-Jacobi route: same W up to its noise
    assert abs(auto.get_jk(dm, omega=omega)[1] - vk).max() < 1e-6 * abs(vk).max()


def test_robust_k_matches_oracle_and_reduces_the_error(be):
    """robust_k (Dunlap's correction, K = K1 + K2 - K_isdf): the device build blocks (isdf_gemm_nn, isdf_hadamard_rows) against
    numpy, the assembled K against the oracle's restatement on the same points, and the point of it: closer to the exact
    exchange than the plain ISDF K at equal rank; a non-symmetric density matrix goes through the two-pass form."""
    from pyscf_isdf_amd.isdf import ISDF
    rng = np.random.default_rng(4)
    A, B, C0 = rng.standard_normal((7, 13)), rng.standard_normal((13, 301)), rng.standard_normal((7, 301))
    dC = be.to_device(C0)
    be.gemm_nn(be.to_device(A), be.to_device(B), dC, alpha=0.5, beta=-1.0)
    assert abs(be.to_host(dC) - (0.5 * A.dot(B) - C0)).max() < 1e-13
    X, Y = rng.standard_normal((5, 77)), rng.standard_normal((5, 77))
    dX = be.to_device(X)
    be.hadamard_rows(dX, be.to_device(Y))
    assert np.array_equal(be.to_host(dX), X * Y)
    cell = cells.cell_diamond_prim('gth-dzvp', (12, 12, 12))
    nao = cell.nao_nr()
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:cell.nelectron // 2] = 2
    dm = (c * occ).dot(c.T)
    plain = ISDF(cell, c_isdf=4, select='local')
    k_plain = plain.get_jk(dm, with_j=False)[1]
    df = ISDF(cell, c_isdf=4, select='local')
    df.robust_k = True
    k_rob = df.get_jk(dm, with_j=False)[1]
    assert np.array_equal(plain.ip, df.ip) and df._want_theta and df._V is not None
    aoT = df.backend.to_host(df.ao)
    a, mesh = cell.lattice_vectors(), cell.mesh
    th = oisdf.fit_theta_global_chol(aoT, df.ip, reg_rel=df.reg_used)
    k_or = oisdf.get_k_robust(aoT, df.ip, th, dm, a, mesh)
    assert abs(k_rob - k_or).max() < 1e-8 * abs(k_or).max()
    k_exact = fftdf.get_k(np.ascontiguousarray(aoT.T), dm, a, mesh)
    e_plain, e_rob = abs(k_plain - k_exact).max(), abs(k_rob - k_exact).max()
    assert e_rob < 0.35 * e_plain
    dn = dm + 0.1 * rng.standard_normal((nao, nao))                       # not symmetric
    k_n = df.get_jk(dn, hermi=0, with_j=False)[1]
    assert abs(k_n - oisdf.get_k_robust(aoT, df.ip, th, dn, a, mesh)).max() < 1e-8 * abs(k_n).max()
    with pytest.raises(NotImplementedError):
        df.get_jk(dm, omega=0.3)
    sh = ISDF(cell, c_isdf=4, select='local')                              # the grid-sharded layout on one rank
    sh.robust_k = True
    sh.force_sharded = True
    sh.fft_batch = 37
    assert abs(sh.get_jk(dm, with_j=False)[1] - k_rob).max() < 1e-9 * abs(k_rob).max()


def test_exxdiv_vcut_sph_and_vcut_ws_match_oracle_pipeline():
    """exxdiv='vcut_sph' / 'vcut_ws' at the Gamma point on the GPU (half-spectrum kernel table with the truncated kernel, W rebuilt once
    from the same fit) == the same host driver over the CPU oracle: K within 1e-8 relative; J untouched."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    nao = cell.nao_nr()
    rng = np.random.default_rng(3)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    out = {}
    for name, backend in (('gpu', None), ('cpu', OracleBackend())):
        df = ISDF(cell, c_isdf=3, select='local', backend=backend)
        df.fit_route = 'cholesky'
        vj0, vk0 = df.get_jk(dm)
        vj1, vk1 = df.get_jk(dm, exxdiv='vcut_sph')
        assert abs(vj1 - vj0).max() < 1e-12 and abs(vk1 - vk0).max() > 1e-3 * abs(vk0).max()
        vj2, vk2 = df.get_jk(dm, exxdiv='vcut_ws')                  # Wigner-Seitz truncation: tabulated kernel, Gamma half table
        assert abs(vj2 - vj0).max() < 1e-12 and sorted(df._W_omega) == ['vcut_sph', 'vcut_ws']
        out[name] = (vk1, vk2)
    for i in (0, 1):
        assert abs(out['gpu'][i] - out['cpu'][i]).max() < 1e-8 * abs(out['cpu'][i]).max()


def test_exxdiv_ewald_adds_madelung_SDS():
    """exxdiv='ewald' = exxdiv=None + madelung * S D S (df_jk.py:1446-1452) with the grid-quadrature overlap."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    nao = cell.nao_nr()
    rng = np.random.default_rng(3)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    df = ISDF(cell, c_isdf=3, select='global')
    vk0 = df.get_jk(dm, exxdiv=None, with_j=False)[1]
    vk1 = df.get_jk(dm, exxdiv='ewald', with_j=False)[1]
    aoT = _oracle_ao(cell)[0]
    S = aoT.dot(aoT.T) * cell.vol / aoT.shape[1]
    ref = gto.madelung(cell) * S.dot(dm).dot(S)
    assert abs((vk1 - vk0) - ref).max() < 1e-10


def test_full_size_properties_diamond222():
    """BASELINE configs[1] at full size (diamond 2x2x2, gth-dzvp, 80^3, c=10: N=208, G=512000, P=2080),
    checked through size-independent properties: J and K symmetric, linear in D, both fit routes
    (explicit Theta vs factor) give the same K energy, W symmetric, exact J energy positive, and the
    ISDF K energy within the c=10 fitting error of what the global-selection yardstick gives."""
    from pyscf_isdf_amd import workloads
    from pyscf_isdf_amd.isdf import ISDF
    cell = workloads.make_cell('diamond-222-dzvp-80')
    dm, c, occ = workloads.make_dm(cell)
    nao = cell.nao_nr()
    assert nao == 208 and int(np.prod(cell.mesh)) == 512000
    df = ISDF(cell, c_isdf=10, select='local')
    vj, vk = df.get_jk(dm)
    assert len(df.ip) == 2080 and len(np.unique(df.ip)) == 2080
    assert abs(vj - vj.T).max() < 1e-9 and abs(vk - vk.T).max() < 1e-8
    W = df.backend.to_host(df.W)
    assert abs(W - W.T).max() < 1e-8 * abs(W).max()
    vj2, vk2 = df.get_jk(np.stack([dm, -0.5 * dm]))
    assert abs(vj2[1] + 0.5 * vj).max() < 1e-9 and abs(vk2[1] + 0.5 * vk).max() < 1e-8
    ej, ek = np.einsum('ij,ji', vj, dm) / 2, np.einsum('ij,ji', vk, dm) / 4
    assert ej > 0 and ek > 0
    df2 = ISDF(cell, c_isdf=10, select='local')
    df2.explicit_theta = True
    ek2 = np.einsum('ij,ji', df2.get_jk(dm, with_j=False)[1], dm) / 4
    assert abs(ek - ek2) < 1e-7                     # same W through the fit routes (auto -> S3c here; 1e-6 Eh target)
    df3 = ISDF(cell, c_isdf=10, select='global')
    ek3 = np.einsum('ij,ji', df3.get_jk(dm, with_j=False)[1], dm) / 4
    assert abs(ek - ek3) < 2e-3                     # local vs global selection: both within the c=10 fitting error


def test_exact_exchange_on_gpu_matches_reference_pin():
    """isdf_get_k_exact = the reference's FFTDF K (occupied-orbital form): reproduces the reference's
    fp(vk) for dm = I (test_fft.py:645) and the oracle's exact K for a rank-deficient random dm."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    nao = cell.nao_nr()
    df = ISDF(cell, c_isdf=2, select='global')
    vk = df.get_k_exact(np.eye(nao))
    assert abs(tools.fp(vk) - 4.290076429522121) < 1e-8
    rng = np.random.default_rng(6)
    c = rng.standard_normal((nao, 2))
    occ = np.array([2.0, 1.0])
    dm = (c * occ).dot(c.T)
    aoT = _oracle_ao(cell)[0]
    ref = fftdf.get_k(aoT.T, dm, cell.lattice_vectors(), cell.mesh, mo_coeff=c, mo_occ=occ)
    assert abs(df.get_k_exact(mo_coeff=c, mo_occ=occ, max_rows=3) - ref).max() < 1e-10
    assert abs(df.get_k_exact(dm) - ref).max() < 1e-10


def test_ao_eri_and_ao2mo_from_the_factorisation():
    """get_ao_eri at full rank reproduces the reference's fp(eri) (test_fft.py:692-695); ao2mo is the same
    tensor transformed."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    nao = cell.nao_nr()
    df = ISDF(cell, c_isdf=4, select='global')
    eri = df.get_ao_eri(compact=True)
    assert eri.shape == (21, 21) and abs(tools.fp(eri) - 0.80425358275734926) < 1e-7
    rng = np.random.default_rng(0)
    c = rng.standard_normal((nao, 3))
    full = df.get_ao_eri(compact=False).reshape(nao, nao, nao, nao)
    ref = np.einsum('pqrs,pi,qj,rk,sl->ijkl', full, c, c, c, c).reshape(9, 9)
    assert abs(df.ao2mo(c) - ref).max() < 1e-10


def test_symmetrize_mean_and_probe_rows(be):
    rng = np.random.default_rng(3)
    for P in (1, 31, 70):                                         # ragged 32 x 32 tiles
        M = rng.standard_normal((P, P))
        d = be.to_device(M)
        be.symmetrize_mean(d)
        assert np.array_equal(be.to_host(d), (M + M.T) / 2)
        d = be.to_device(M)
        be.symmetrize_mean(d, antisymmetric=True)
        assert np.array_equal(be.to_host(d), (M - M.T) / 2)
    _check_probe_rows_and_factor_solve(be, rng, 37, 5, 90, np.array([0, 12, 12, 30, 37], dtype=np.int32))
    _check_probe_rows_and_factor_solve(be, rng, 301, 8, 1003, np.array([0, 100, 170, 301], dtype=np.int32))   # skinny kernel:
    _check_probe_rows_and_factor_solve(be, rng, 130, 11, 700, np.array([0, 65, 130], dtype=np.int32))        # ragged; n > 8 -> dgemm


def _check_probe_rows_and_factor_solve(be, rng, P, n, ng, off):
    # probe rows: T <- A'^-1 D^-1 t (A' = D^-1 A D^-T), F = T Y'
    Z = rng.standard_normal((P, 2 * P))
    A = Z.dot(Z.T) + 0.5 * np.eye(P)
    D = be.empty((P, P))
    dA = be.to_device(A)
    be.block_chol(dA, off, 0.0, D)
    be.block_solve(D, off, 0, 0, dA); be.block_solve(D, off, 1, 1, dA)
    Ap = be.to_host(dA)
    be.chol_inplace(dA, 0.0)
    Dh = np.zeros((P, P))
    for b in range(len(off) - 1):
        sl = slice(off[b], off[b + 1])
        Dh[sl, sl] = np.linalg.cholesky(A[sl, sl]) if sl.stop > sl.start else 0
    T = rng.standard_normal((n, P)); Y = rng.standard_normal((P, ng))
    dT, dF = be.to_device(T), be.empty((n, ng))
    be.bj_probe_rows(dT, dA, D, off, be.to_device(Y), ng, dF)
    E = np.linalg.solve(Ap, np.linalg.solve(Dh, T.T)).T
    assert abs(be.to_host(dT) - E).max() < 1e-11 * abs(E).max()
    assert abs(be.to_host(dF) - E.dot(Y)).max() < 1e-11 * abs(E.dot(Y)).max()
    # factor_solve on a column block with a row stride (the sharded finishing solves)
    X = rng.standard_normal((P, 11))
    big = be.to_device(np.hstack([X, np.zeros((P, 3))]))
    view = big[:, :11]
    be.factor_solve(dA, view)
    assert abs(be.to_host(big)[:, :11] - np.linalg.solve(Ap, X)).max() < 1e-11 * abs(X).max()
    assert abs(be.to_host(big)[:, 11:]).max() == 0


@pytest.mark.parametrize('m', [1, 63, 64, 65, 130, 515, 1100])
@pytest.mark.parametrize('subst', [0, 1])
def test_triangular_solves_both_implementations(be, m, subst):
    """isdf_block_solve with one block through rocBLAS dtrsm (default) and through the substitution blocks of trsm.hip
    (isdf_set_option "trsm_substitution"): left/right, L / L^T, ragged 64-row blocks and 512-row panels, odd numbers of
    right-hand sides and a row stride; and the factor of an ill-conditioned Gram matrix (residual at rounding level)."""
    import scipy.linalg
    be.set_option('trsm_substitution', subst)
    try:
        _check_triangular_solves(be, m)
    finally:
        be.set_option('trsm_substitution', 0)
    with pytest.raises(Exception):
        be.set_option('no_such_option', 1)


def _check_triangular_solves(be, m):
    import scipy.linalg
    rng = np.random.default_rng(m)
    L = np.tril(rng.standard_normal((m, m))) * 0.3 + np.diag(1.0 + rng.random(m))
    D = be.to_device(L + np.triu(rng.standard_normal((m, m)), 1))          # garbage above the diagonal must be ignored
    off = np.array([0, m], dtype=np.int32)
    n = 777
    for side in (0, 1):
        for trans in (0, 1):
            X = rng.standard_normal((m, n) if side == 0 else (n, m))
            buf = be.to_device(np.hstack([X, np.zeros((X.shape[0], 5))]))
            view = buf[:, :X.shape[1]]
            be.block_solve(D, off, side, trans, view)
            if side == 0:
                ref = scipy.linalg.solve_triangular(L, X, lower=True, trans='T' if trans else 'N')
            else:
                ref = scipy.linalg.solve_triangular(L, X.T, lower=True, trans='N' if trans else 'T').T
            got = be.to_host(buf)
            assert abs(got[:, :X.shape[1]] - ref).max() < 1e-10 * abs(ref).max(), (side, trans)
            assert abs(got[:, X.shape[1]:]).max() == 0
    if m >= 130:
        # Cholesky factor of a Gram matrix with condition number 1e12: residual of L x = b at rounding level
        Z = rng.standard_normal((m, m)) * np.logspace(0, -6, m)
        Lc = np.linalg.cholesky(Z.dot(Z.T) + 1e-13 * np.eye(m))
        Xc = rng.standard_normal((m, 50))
        d = be.to_device(Xc)
        be.block_solve(be.to_device(Lc), off, 0, 0, d)
        res = Lc.dot(be.to_host(d)) - Xc
        assert abs(res).max() < 1e-9 * (abs(Lc).max() * abs(be.to_host(d)).max())


def test_block_chol_shifts_a_block_that_is_not_positive_definite(be):
    """D is only a preconditioner: a numerically indefinite diagonal block gets a larger shift instead of an error."""
    rng = np.random.default_rng(8)
    P = 24
    off = np.array([0, 10, 24], dtype=np.int32)
    Z = rng.standard_normal((P, 40))
    A = Z.dot(Z.T)
    A[:10, :10] -= (np.linalg.eigvalsh(A[:10, :10])[0] + 1e-9 * A.diagonal().max()) * np.eye(10)   # smallest eigenvalue < 0
    assert np.linalg.eigvalsh(A[:10, :10])[0] < 0 < np.linalg.eigvalsh(A[10:, 10:])[0]
    D = be.empty((P, P))
    used = be.block_chol(be.to_device(A), off, 0.0, D)
    assert 1e-14 <= used <= 1e-4
    Dh = be.to_host(D)
    L0 = np.tril(Dh[:10, :10]); L1 = np.tril(Dh[10:, 10:])
    md = A.diagonal().max()
    assert abs(L0.dot(L0.T) - (A[:10, :10] + used * md * np.eye(10))).max() < 1e-10 * md
    assert abs(L1.dot(L1.T) - A[10:, 10:]).max() < 1e-12 * md                    # the healthy block is not shifted
    assert abs(Dh[:10, 10:]).max() == 0 and abs(Dh[10:, :10]).max() == 0


def test_block_jacobi_route_building_blocks(be):
    """S3c primitives vs numpy on a random SPD problem with unequal blocks, and the assembled W vs the oracle."""
    rng = np.random.default_rng(12)
    P, n = 37, 50
    off = np.array([0, 10, 10, 26, 37], dtype=np.int32)          # includes an empty block
    Z = rng.standard_normal((P, 60))
    A = Z.dot(Z.T) + 0.1 * np.eye(P)
    D = be.empty((P, P))
    be.block_chol(be.to_device(A), off, 0.0, D)
    Dh = be.to_host(D)
    ref = np.zeros((P, P))
    for b in range(4):
        s = slice(off[b], off[b + 1])
        if s.stop > s.start:
            ref[s, s] = np.linalg.cholesky(A[s, s])
    assert abs(np.tril(Dh) - ref).max() < 1e-12
    X = rng.standard_normal((P, n))
    Dm = ref + np.eye(P) * 0     # block lower-triangular matrix
    for side, trans in ((0, 0), (0, 1), (1, 0), (1, 1)):
        x = be.to_device(X if side == 0 else X.T.copy())
        be.block_solve(D, off, side, trans, x)
        op = Dm.T if trans else Dm
        want = np.linalg.solve(op, X) if side == 0 else X.T.dot(np.linalg.inv(op))
        assert abs(be.to_host(x) - want).max() < 1e-10 * abs(want).max()
    # kind 2 + kind 0 = A^-1 M A^-1
    M = rng.standard_normal((P, P)); M = M + M.T
    F = be.to_device(A)
    be.chol_inplace(F, 0.0)
    W = be.to_device(M)
    be.W_from_factor(F, 2, W)
    be.W_from_factor(F, 0, W)
    Ai = np.linalg.inv(A)
    assert abs(be.to_host(W) - Ai.dot(M).dot(Ai)).max() < 1e-9 * abs(Ai.dot(M).dot(Ai)).max()


@pytest.mark.parametrize('sizes,n', [([156, 130, 7, 64, 201], 1000), ([16], 33), ([100, 100], 64), ([330, 5], 257), ([280], 100),
                                     ([500, 40], 70), ([1100], 40)])
def test_block_invert_and_mfma_block_apply(be, sizes, n):
    """isdf_block_invert + isdf_block_apply (explicit block inverses applied with v_mfma_f64_16x16x4, ragged blocks, ragged
    column tiles, the three column-tile widths) against numpy triangular solves: Dinv exact zeros above the diagonal and
    outside the blocks, X <- D_b^-1 X_b to 1e-12 relative; also on a diagonal sub-range (the paneled build's use)."""
    import scipy.linalg
    rng = np.random.default_rng(sum(sizes) + n)
    off = np.append(0, np.cumsum(sizes)).astype(np.int32)
    P = int(off[-1])
    D = np.zeros((P, P))
    for b in range(len(sizes)):
        s = slice(off[b], off[b + 1])
        M = rng.standard_normal((sizes[b], sizes[b]))
        D[s, s] = np.linalg.cholesky(M.dot(M.T) + sizes[b] * np.eye(sizes[b]))
    X = rng.standard_normal((P, n))
    dD, dX = be.to_device(D), be.to_device(X)
    dI = be.empty((P, P))
    be.block_invert(dD, off, dI)
    Dinv = be.to_host(dI)
    ref = X.copy()
    for b in range(len(sizes)):
        s = slice(off[b], off[b + 1])
        assert abs(Dinv[s, s] - scipy.linalg.solve_triangular(D[s, s], np.eye(sizes[b]), lower=True)).max() < 1e-12
        ref[s] = scipy.linalg.solve_triangular(D[s, s], X[s], lower=True)
    mask = np.zeros((P, P), dtype=bool)
    for b in range(len(sizes)):
        s = slice(off[b], off[b + 1])
        mask[s, s] = np.tril(np.ones((sizes[b], sizes[b]), dtype=bool))
    assert np.all(Dinv[~mask] == 0.0)
    be.block_apply(dI, off, dX)
    assert abs(be.to_host(dX) - ref).max() < 1e-12 * abs(ref).max()
    # the round-2 kernel (Dinv from L2 per column tile) behind the option gives the same answer
    be.set_option('block_apply_reg', 0)
    try:
        dX2 = be.to_device(X)
        be.block_apply(dI, off, dX2)
        assert abs(be.to_host(dX2) - ref).max() < 1e-12 * abs(ref).max()
    finally:
        be.set_option('block_apply_reg', 1)
    # the fused form: rows <- Dinv_b (aoP ao)^2 with the square applied while the block apply stages its input
    aoP, ao = rng.standard_normal((P, 9)), rng.standard_normal((9, n))
    B = aoP.dot(ao) ** 2
    refB = B.copy()
    for b in range(len(sizes)):
        s = slice(off[b], off[b + 1])
        refB[s] = scipy.linalg.solve_triangular(D[s, s], B[s], lower=True)
    dB = be.empty((P, n))
    be.pair_rows_block_apply(be.to_device(aoP), be.to_device(ao), n, dI, off, dB)
    assert abs(be.to_host(dB) - refB).max() < 1e-12 * abs(refB).max()
    if len(sizes) > 2:                                    # blocks 1..2 only, through views (leading dimensions of the parents)
        r0, r1 = int(off[1]), int(off[3])
        dX2 = be.to_device(X)
        be.block_apply(dI[r0:r1, r0:r1], (off[1:4] - r0).astype(np.int32), dX2[r0:r1])
        got = be.to_host(dX2)
        assert abs(got[r0:r1] - ref[r0:r1]).max() < 1e-12 * abs(ref).max() and np.array_equal(got[:r0], X[:r0]) and np.array_equal(got[r1:], X[r1:])


def test_auto_route_probe_check_accepts_and_falls_back():
    """fit_route='auto': the probe check accepts the block-Jacobi route; when its mismatch exceeds the tolerance
    (forced here with a tiny tolerance) the build warns and rebuilds W with the Cholesky route."""
    import warnings
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he2_triclinic()
    nao = cell.nao_nr()
    dm = np.eye(nao)
    ref = ISDF(cell, c_isdf=8, select='local'); ref.fit_route = 'cholesky'
    k_ref = ref.get_jk(dm, with_j=False)[1]
    ok = ISDF(cell, c_isdf=8, select='local')                 # 64 points for 36 pair products: A is rank deficient
    k_ok = ok.get_jk(dm, with_j=False)[1]
    assert ok.fit_route_used == 'blockjacobi' and 0 < ok.bj_check <= ok.bj_check_tol
    # both routes solve the same regularised normal equations: they differ by (amplified) rounding only,
    # and the probe mismatch is of the size of that difference
    err = abs(k_ok - k_ref).max() / abs(k_ref).max()
    assert err < 1e-6 and ok.bj_check < 1e-6
    df = ISDF(cell, c_isdf=8, select='local')
    df.bj_check_tol = 1e-13
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        k_auto = df.get_jk(dm, with_j=False)[1]
    assert df.fit_route_used == 'cholesky' and df.bj_check > df.bj_check_tol
    assert any('probe check' in str(w.message) for w in rec)
    assert abs(k_auto - k_ref).max() < 1e-10 * abs(k_ref).max()
    hi = ISDF(cell, c_isdf=15, select='local')
    assert hi._fit_routes() == ['cholesky']                   # above bj_max_c the trial is skipped


def test_block_jacobi_route_end_to_end():
    """ISDF with fit_route='blockjacobi' (no triangular solve over the grid) gives the same K as the Cholesky
    route and as the oracle's restatement of S3c."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-dzvp', (12, 12, 12))
    nao = cell.nao_nr()
    rng = np.random.default_rng(20240203)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:cell.nelectron // 2] = 2
    dm = (c * occ).dot(c.T)
    ref = ISDF(cell, c_isdf=6, select='local')
    ref.fit_route = 'cholesky'
    vk0 = ref.get_jk(dm, with_j=False)[1]
    df = ISDF(cell, c_isdf=6, select='local')
    assert df.fit_route == 'auto'                            # the default: S3c + probe check, S3b when it fails
    df.fit_route = 'blockjacobi'
    vk1 = df.get_jk(dm, with_j=False)[1]
    assert np.array_equal(ref.ip, df.ip)
    # the two routes regularise differently (shift on A_PP vs on the block-scaled A'): agreement to 1e-6 relative
    assert abs(vk1 - vk0).max() < 1e-6 * abs(vk0).max()
    assert abs(np.einsum('ij,ji', vk1 - vk0, dm)) / 4 < 1e-6
    aoT = df.backend.to_host(df.ao)
    owner_counts = np.array([len(df.ip) // cell.natm] * cell.natm)
    off = np.append(0, np.cumsum(owner_counts))
    W_or = oisdf.build_W_blockjacobi(aoT, df.ip, off, cell.lattice_vectors(), cell.mesh, reg_rel=df.reg_rel)
    k_or = oisdf.get_k(np.ascontiguousarray(aoT[:, df.ip].T), W_or, dm)
    assert abs(vk1 - k_or).max() < 1e-8 * abs(k_or).max()
    # the same build with the substitution solves of trsm.hip instead of rocBLAS dtrsm
    df.backend.set_option('trsm_substitution', 1)
    try:
        df.build()
        vk2 = df.get_jk(dm, with_j=False)[1]
    finally:
        df.backend.set_option('trsm_substitution', 0)
    assert abs(vk2 - vk1).max() < 1e-7 * abs(vk1).max()


def test_paneled_build_matches_single_pass():
    """The paneled S3c/S4/S5 (fit rows produced panel by panel when HBM cannot hold them all; forced here with
    max_resident_rows) against the single-pass block-Jacobi build on the same points: W within 1e-8, K within 1e-9 relative,
    the same probe-check value, for 2 and 4 panels with ragged FFT batches."""
    from pyscf_isdf_amd import gto as g
    from pyscf_isdf_amd.isdf import ISDF
    cell = g.diamond_supercell(2, 'gth-szv', (20, 20, 20))
    nao = cell.nao_nr()
    rng = np.random.default_rng(2)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    ref = ISDF(cell, c_isdf=6, select='refined')
    k0 = ref.get_jk(dm, with_j=False)[1]
    W0 = ref.backend.to_host(ref.W)
    P = len(ref.ip)
    assert ref.fit_route_used == 'blockjacobi' and ref.n_panels == 1
    for rows in (P // 2 + 10, P // 4 + 10):
        df = ISDF(cell, c_isdf=6, select='refined')
        df.max_resident_rows, df.fft_batch = rows, 37
        k1 = df.get_jk(dm, with_j=False)[1]
        assert df.n_panels >= 2 and np.array_equal(df.ip, ref.ip)
        assert abs(df.backend.to_host(df.W) - W0).max() < 1e-8 * abs(W0).max()
        assert abs(k1 - k0).max() < 1e-9 * abs(k0).max()
        assert abs(df.bj_check - ref.bj_check) < 0.5 * ref.bj_check + 1e-12
        df.reset()


def test_rccl_code_paths_execute_on_one_gpu():
    """Every collective of pyscf_isdf_amd/parallel.py on the real backend: a child process initialises
    torch.distributed with backend "nccl" (RCCL), world_size 1, and runs the grid-sharded build (pipelined list
    all_to_all around the convolution, all_reduce of W, broadcast of the factors, MAX all_reduce of the route decision,
    all_gather_object of the point lists) and the q-sharded k-point build with Comm(always=True); results must equal the
    single-GPU path (tests/nccl_one_rank.py).  world_size > 1 over RCCL is only reachable in the driver's scaling bench."""
    import os
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nccl_one_rank.py')
    r = subprocess.run([sys.executable, script, str(port)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'RCCL-ONE-RANK OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_two_ranks_over_rccl_when_two_gpus_are_visible():
    """world_size 2 over RCCL (one GPU per rank): the grid-sharded build against the single-GPU path.  Skipped on one-GPU boxes."""
    import os
    import socket
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nccl_two_rank.py')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), script], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'RCCL-TWO-RANK OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_occ_pair_space_kernels(be):
    """isdf_gram_prod / isdf_pair_prod_rows / isdf_factor_solve_half against numpy: the Gram products of the (AO x occupied)
    pair space (fft_jk.py:206-210,235-238), with row counts and column counts that are not multiples of any tile and more
    columns than one chunk of the second factor (the chunk is at least 1024 columns).  1e-13 relative (plain GEMMs)."""
    rng = np.random.default_rng(5)
    P, nao, nocc, G = 77, 23, 6, 70001
    ao = rng.standard_normal((nao, G)) * np.exp(-2.0 * rng.random(G))
    psi = rng.standard_normal((nocc, nao)).dot(ao)
    ip = rng.permutation(G)[:P]
    aoP, psiP = np.ascontiguousarray(ao[:, ip].T), np.ascontiguousarray(psi[:, ip].T)
    d_aoP, d_psiP, d_ao, d_psi = (be.to_device(x) for x in (aoP, psiP, ao, psi))
    A = be.empty((P, P))
    be.gram_prod(d_aoP, d_psiP, A)
    refA = aoP.dot(aoP.T) * psiP.dot(psiP.T)
    assert abs(be.to_host(A) - refA).max() < 1e-13 * abs(refA).max()
    B = be.empty((P, G))
    be.pair_prod_rows(d_aoP, d_psiP, d_ao, d_psi, G, B)
    refB = aoP.dot(ao) * psiP.dot(psi)
    assert abs(be.to_host(B) - refB).max() < 1e-13 * abs(refB).max()
    # a row slice of the points against a column range (what the paneled build asks for)
    B2 = be.empty((30, 5000))
    be.pair_prod_rows(d_aoP[10:40], d_psiP[10:40], d_ao, d_psi, 5000, B2)
    assert abs(be.to_host(B2) - refB[10:40, :5000]).max() < 1e-13 * abs(refB).max()
    # Cholesky halves: L^-1 X and L^-T X for the factor of the regularised Gram matrix
    be.shift_diag(A, 1e-6)
    reg = be.chol_inplace(A, 0.0)
    assert reg == 0.0
    Lr = np.linalg.cholesky(refA + 1e-6 * refA.diagonal().max() * np.eye(P))
    X = rng.standard_normal((P, 3001))
    dX = be.to_device(X)
    be.factor_solve_half(A, False, dX)
    import scipy.linalg
    Y = scipy.linalg.solve_triangular(Lr, X, lower=True)
    assert abs(be.to_host(dX) - Y).max() < 1e-9 * abs(Y).max()
    be.factor_solve_half(A, True, dX)
    Z = scipy.linalg.solve_triangular(Lr, Y, lower=True, trans='T')
    assert abs(be.to_host(dX) - Z).max() < 1e-8 * abs(Z).max()


@pytest.mark.parametrize('route', ['cholesky', 'blockjacobi', 'paneled'])
def test_occ_pair_space_end_to_end_matches_oracle_pipeline(route):
    """pair_space='occ' on the GPU == the same host driver over the CPU oracle: identical points from the product Gram matrix
    of the candidates, K within 1e-9 relative, for the Cholesky route, the block-Jacobi route and the paneled block-Jacobi
    build (rows recomputed panel by panel in the (AO x occupied) pair space); the fit follows the density (second density:
    refit) and K of the fitted density is closer to the exact exchange than the AO x AO fit's at equal points."""
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    nao = cell.nao_nr()
    rng = np.random.default_rng(3)
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:2] = 2
    dm = (c * occ).dot(c.T)

    class Tagged(np.ndarray):
        pass
    tdm = dm.view(Tagged)
    tdm.mo_coeff, tdm.mo_occ = c, occ
    out = {}
    for name, backend in (('gpu', None), ('cpu', OracleBackend())):
        df = ISDF(cell, c_isdf=2, select='refined', backend=backend)
        df.pair_space = 'occ'
        df.refine_over = 2.0
        df.fit_route = 'blockjacobi' if route == 'paneled' else route
        if route == 'paneled':
            df.max_resident_rows = 8
            df.fft_batch = 4
        vk = df.get_jk(tdm, with_j=False)[1]
        if route == 'paneled':
            assert df.n_panels >= 2
        out[name] = (vk, df.ip.copy())
        if name == 'gpu':
            occ2 = np.zeros(nao); occ2[2:4] = 2
            t2 = (c * occ2).dot(c.T).view(Tagged)
            t2.mo_coeff, t2.mo_occ = c, occ2
            vk_b = df.get_jk(t2, with_j=False)[1]
            aoT = _oracle_ao(cell)[0]
            k_ex = fftdf.get_k(np.ascontiguousarray(aoT.T), np.asarray(t2), cell.lattice_vectors(), cell.mesh)
            ref = ISDF(cell, c_isdf=2, select='refined')
            ref.fit_route = df.fit_route
            vk_ao = ref.get_jk(np.asarray(t2), with_j=False)[1]
            assert abs(vk_b - k_ex).max() < abs(vk_ao - k_ex).max()
    assert np.array_equal(out['gpu'][1], out['cpu'][1])
    assert abs(out['gpu'][0] - out['cpu'][0]).max() < 1e-9 * abs(out['cpu'][0]).max()


def test_max_device_memory_forces_panels_without_changing_the_result():
    """max_device_memory (the role of the reference's max_memory-driven blocking, numint.py:1236-1257: less memory = more
    blocks, same numbers): capping the build's device memory makes the block-Jacobi route produce its fit rows in panels; K is
    the single-pass K to rounding."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-dzvp', (24, 24, 24))
    nao = cell.nao_nr()
    rng = np.random.default_rng(2)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    ref = ISDF(cell, c_isdf=8, select='refined')
    ref.fit_route = 'blockjacobi'
    k0 = ref.get_jk(dm, with_j=False)[1]
    assert ref.n_panels == 1
    G = int(np.prod(cell.mesh))
    df = ISDF(cell, c_isdf=8, select='refined')
    df.fit_route, df.fft_batch = 'blockjacobi', 32
    df.w_spectral = False                      # the classic form: rows in panels (the spectral form, below, fits under the same cap)
    P = 8 * nao
    # room for about half of the rows next to the fixed allowances of FitRouteMixin._resident_rows
    df.max_device_memory = 8 * nao * G + 4 * 8 * P * P + (3 << 30) + 40 * 32 * G + 8 * G * (P // 2 + 24)
    k1 = df.get_jk(dm, with_j=False)[1]
    assert df.n_panels >= 2 and np.array_equal(df.ip, ref.ip)
    assert abs(k1 - k0).max() < 1e-9 * abs(k0).max()
    sp = ISDF(cell, c_isdf=8, select='refined')
    sp.fit_route, sp.fft_batch, sp.w_sphere, sp.max_device_memory = 'blockjacobi', 32, 0, df.max_device_memory
    k2 = sp.get_jk(dm, with_j=False)[1]
    # the whole box in spectral form is LARGER than the rows (2 (n2/2 + 1) / n2 of them): under this cap it stays classic, in panels
    assert sp.n_panels >= 2 and sp.w_spectral_fraction is None and abs(k2 - k0).max() < 1e-9 * abs(k0).max()
    sp.w_sphere, sp.bj_check_tol, sp.w_spectral_check_tol = 100.0, 1e-5, 1e-5           # the sphere (0.3 of the box) fits in one piece
    k3 = sp.get_jk(dm, with_j=False)[1]
    sp.build()
    k3 = sp.get_jk(dm, with_j=False)[1]
    assert sp.n_panels == 1 and 0.2 < sp.w_spectral_fraction < 0.4 and abs(k3 - k0).max() < 1e-4 * abs(k0).max()


def test_candidate_stage_skips_rows_that_vanish_on_a_block_without_changing_the_points(be):
    """The per-atom selections leave out the AO rows that are identically zero on the atom's block of grid points (shells
    truncated at rcut): exact zeros drop out of every dot product, so points and K are the same to the last bit; and the
    row maxima behind the decision are the true ones."""
    from pyscf_isdf_amd.isdf import ISDF
    atoms = '; '.join('He %g %g %g' % (x, y, z) for x in (0.3, 4.2) for y in (0.1, 4.4) for z in (0.2, 4.1))
    cell = gto.Cell(atom=atoms, basis={'He': [[0, [2.2, 1]], [0, [1.1, 1]], [1, [1.6, 1]]]}, a=np.eye(3) * 8.0, mesh=[30] * 3)
    nao = cell.nao_nr()
    rng = np.random.default_rng(5)
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    out = {}
    for flag in (False, True):
        df = ISDF(cell, c_isdf=6, select='refined')
        df.cand_skip_zero_rows = flag
        vk = df.get_jk(dm, with_j=False)[1]
        out[flag] = (df.ip.copy(), vk, getattr(df, '_cand_rows_kept', None))
    kept, total = out[True][2]
    assert total == nao and kept < 0.8 * nao and out[False][2] is None
    assert np.array_equal(out[False][0], out[True][0])
    assert np.array_equal(out[False][1], out[True][1])
    # the kernel behind it against numpy on ragged blocks (one of them empty)
    src = rng.standard_normal((37, 1000)); src[5, 100:300] = 0.0; src[:, 700:] *= 1e-3
    off = np.array([0, 100, 300, 300, 707, 1000], dtype=np.int64)
    got = be.block_row_absmax(be.to_device(src), off)
    ref = np.stack([abs(src[:, a:b]).max(axis=1) if b > a else np.zeros(37) for a, b in zip(off[:-1], off[1:])], axis=1)
    assert got.shape == (37, 5) and np.array_equal(got, ref)


def test_spectral_W_matches_the_classic_product(be):
    """W = X X^T from the packed half spectra (isdf_spectral_rows + isdf_gemm_nt) against w conv(rows) rows^T (isdf_coulomb_W) on the
    same rows: the whole box to rounding, the sphere against the classic product made with the sphere-truncated kernel table
    (option coul_sphere) to rounding; then the object end to end - whole box = classic K to the route's noise, 'auto' declines
    the sphere on a mesh that does not resolve the pair products."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-dzvp', (24, 24, 24))
    mesh = np.asarray(cell.mesh, dtype=np.int32)
    a = np.asarray(cell.lattice_vectors())
    G = int(np.prod(mesh))
    rng = np.random.default_rng(8)
    rows = be.to_device(rng.standard_normal((70, G)))
    Wc = be.empty((70, 70))
    be.coulomb_W(rows, mesh, a, 0, 70, 32, Wc)
    holder = ISDF(cell, c_isdf=4, select='local', backend=be)
    for pct, ksphere in ((0, 0), (100.0, 100), (70.0, 70)):
        holder.w_sphere = pct
        plan = holder._spectral_plan()
        X = be.empty((70, plan['ldx']))
        be.spectral_rows(rows, mesh, plan['idx'], plan['scale'], X, batch=32)
        Ws = be.empty((70, 70))
        be.gemm_nt(X, X, Ws)
        be.set_option('coul_sphere', ksphere)
        try:
            Wk = be.empty((70, 70))
            be.coulomb_W(rows, mesh, a, 0, 70, 32, Wk)
        finally:
            be.set_option('coul_sphere', 0)
        ws, wk = be.to_host(Ws), be.to_host(Wk)
        assert abs(ws - wk).max() < 1e-11 * abs(wk).max()
        if pct == 0:
            assert plan['npts'] == 24 * 24 * 13 - 1 and abs(wk - be.to_host(Wc)).max() == 0.0
        else:
            assert plan['fraction'] < 0.35 and abs(wk - be.to_host(Wc)).max() > 1e-9 * abs(wk).max()      # white noise is not band limited
        # against numpy on the half spectrum
        z = np.fft.rfftn(be.to_host(rows).reshape(70, *mesh), axes=(1, 2, 3)).reshape(70, -1)
        v = z[:, be.to_host(plan['idx'])] * be.to_host(plan['scale'])
        x = be.to_host(X)
        assert abs(x[:, 0:2 * plan['npts']:2] - v.real).max() < 1e-10 * abs(v).max() and abs(x[:, 2 * plan['npts']:]).max() == 0.0
    nao = cell.nao_nr()
    dm = rng.standard_normal((nao, nao)); dm = dm + dm.T
    ref = ISDF(cell, c_isdf=8, select='refined'); ref.w_spectral = False
    k0 = ref.get_jk(dm, with_j=False)[1]
    box = ISDF(cell, c_isdf=8, select='refined'); box.w_sphere, box.w_spectral_check_tol = 0, 3e-8
    k1 = box.get_jk(dm, with_j=False)[1]
    assert box._fit_state['kind'] == 'blockjacobi-spectral' and box.w_spectral_fraction > 1.0
    assert np.array_equal(box.ip, ref.ip) and abs(k1 - k0).max() < 1e-8 * abs(k0).max()
    auto = ISDF(cell, c_isdf=8, select='refined')
    k2 = auto.get_jk(dm, with_j=False)[1]
    assert auto.w_spectral_fraction is None and auto._sphere_share[1] > 1e-9 and abs(k2 - k0).max() < 1e-12 * abs(k0).max()
