"""GPU parity of the k-point path (through the C ABI) against oracle/kisdf.py and the reference's exact
k-point exchange (oracle/fftdf.get_jk_kpts, pinned to test_fft.py:670-676)."""
import numpy as np
import pytest
import torch
import cells
from pyscf_isdf_amd import gto, pbc_tools
from oracle import ao as oao, fftdf, kisdf, c_oracle, pbc_tools as otools

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    from pyscf_isdf_amd.backend import HipBackend
    return HipBackend(0)


def _setup(nk=2, seed=11):
    cell = cells.cell_he2_triclinic()
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    rng = np.random.default_rng(seed)
    kpts = rng.random((nk, 3)) * 0.6
    kpts[0] = 0.0
    aos = [np.asarray(x, dtype=complex) for x in
           oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpts, rule='point')]
    nao = cell.nao_nr()
    c = rng.standard_normal((nk, nao, nao)) + 1j * rng.standard_normal((nk, nao, nao))
    dms = np.einsum('kpi,kqi->kpq', c[:, :, :3], c[:, :, :3].conj())
    return cell, coords, Ls, rcut, kpts, aos, dms


def _gpu_X(be, cell, coords, Ls, rcut, kpts, periodic=True):
    nk, nao, G = len(kpts), cell.nao_nr(), len(coords)
    nh = nk * nao
    X = be.empty((2 * nh, G))
    cs = be.to_device(np.ascontiguousarray(coords.T))
    for k in range(nk):
        be.eval_ao_k(cell._atm, cell._bas, cell._env, Ls, rcut, kpts[k], periodic, cs,
                     X[k * nao:(k + 1) * nao], X[nh + k * nao:nh + (k + 1) * nao])
    return X


def test_eval_ao_k(be):
    """Bloch AOs and their periodic parts at Gamma and a generic k: 1e-12."""
    cell, coords, Ls, rcut, kpts, aos, dms = _setup()
    nk, nao = len(kpts), cell.nao_nr()
    nh = nk * nao
    for periodic in (False, True):
        X = be.to_host(_gpu_X(be, cell, coords, Ls, rcut, kpts, periodic))
        for k in range(nk):
            ref = aos[k].T * (np.exp(-1j * coords.dot(kpts[k])) if periodic else 1.0)
            got = X[k * nao:(k + 1) * nao] + 1j * X[nh + k * nao:nh + (k + 1) * nao]
            assert abs(got - ref).max() < 1e-12
    assert abs(kisdf.periodic_stack(aos, coords, kpts) - X).max() < 1e-12


def test_select_and_fit_complex_mode(be):
    """Complex-mode S2 (identical pivots vs the plain-C oracle on the same input) and S3."""
    cell, coords, Ls, rcut, kpts, aos, dms = _setup()
    X = kisdf.periodic_stack(aos, coords, kpts)
    nh = X.shape[0] // 2
    G = X.shape[1]
    k = 60
    piv_ref, L_ref = c_oracle.select_ip_cplx(X, k)
    piv_np, _ = kisdf.select_ip(X, k)
    assert np.array_equal(piv_ref, piv_np)
    dX = be.to_device(X)
    L = be.zeros((k, G))
    piv = be.empty((1, k), dtype=torch.int64)
    rank = be.select_ip_cplx(dX, nh, [0, G], [k], -1.0, 1e-10, L, piv)
    assert rank[0] == k and np.array_equal(be.to_host(piv)[0], piv_ref)
    assert abs(be.to_host(L) - L_ref).max() < 1e-9 * abs(L_ref).max()
    # fit (explicit Theta) vs numpy
    ip = np.sort(piv_ref)
    aoP = be.empty((k, 2 * nh)); chol = be.empty((k, k)); theta = be.empty((k, G))
    reg = be.fit_prepare_cplx(dX, nh, be.to_device(ip), 0.0, aoP, chol)
    be.fit_apply_cplx(chol, aoP, nh, dX, G, theta)
    ref = kisdf.fit_theta(X, ip, reg)
    assert abs(be.to_host(theta) - ref).max() < 1e-7 * abs(ref).max()


def test_coulG_q_kernel_matches_reference_pins_and_oracle(be):
    """isdf_coulG_q (index-space wrap-around) against the REFERENCE's own constants for tools.get_coulG
    (pyscf/pbc/tools/test/test_pbc.py:54-77: random triclinic lattice + random k, mesh [11,9,7]; cubic cell with q on the
    mesh edge, with and without wrap-around), and against the oracle restatement for generic q, even meshes and omega.
    Tolerance 1e-9 on the fingerprints (the reference's assertAlmostEqual places), 1e-12 relative element-wise."""
    np.random.seed(19)
    kpt = np.random.random(3)
    a = (np.array(((0., 1.7834, 1.7834), (1.7834, 0., 1.7834), (1.7834, 1.7834, 0.))) + np.random.random((3, 3)).T) / 0.52917721092
    mesh = [11, 9, 7]
    got = be.to_host(be.coulG_q(mesh, a, kpt))
    assert abs(otools.fp(got) - 62.75448804333378) < 1e-9 * 62.75
    assert abs(got - otools.get_coulG(a, mesh, kpt)).max() < 1e-12 * abs(got).max()
    eye = np.eye(3)
    q = np.array([0, np.pi, 0])
    assert abs(otools.fp(be.to_host(be.coulG_q(mesh, eye, q))) - 4.6737453679713905) < 1e-9
    assert abs(otools.fp(be.to_host(be.coulG_q(mesh, eye, q, wrap_around=False))) - 4.5757877990664744) < 1e-9
    # generic q, even and odd mesh sizes, range separation; the Gamma table too
    cell = cells.cell_he2_triclinic()
    al = cell.lattice_vectors()
    for m in ([9, 9, 9], [10, 8, 6], [12, 5, 7]):
        for qv in (np.zeros(3), np.array([0.13, -0.2, 0.31]), cell.make_kpts([2, 2, 2])[5], -cell.make_kpts([2, 2, 2])[7]):
            for om in (None, 0.4, -0.4):
                ref = otools.get_coulG(al, m, qv, omega=om)
                got = be.to_host(be.coulG_q(m, al, qv, omega=om))
                assert np.array_equal(got == 0, ref == 0)                      # the same zeroed (edge / G = 0) entries
                assert abs(got - ref).max() < 1e-12 * abs(ref).max()
    with pytest.raises(Exception):
        be.coulG_q([9, 9, 9], eye, np.array([0, 2 * np.pi * 5, 0]))             # outside the first FFT box (pbc.py:281)
    # exxdiv='vcut_sph' (pbc.py:312-317): 4 pi/|q+G|^2 (1 - cos(|q+G| Rc)), |q+G| = 0 -> 2 pi Rc^2
    be.set_coulomb_cutoff(5.3)
    try:
        for qv in (np.zeros(3), np.array([0.13, -0.2, 0.31])):
            ref = otools.get_coulG(al, [10, 8, 6], qv, rc=5.3)
            got = be.to_host(be.coulG_q([10, 8, 6], al, qv))
            assert abs(got - ref).max() < 1e-11 * abs(ref).max() and abs(got[0] - ref[0]) < 1e-12 * ref[0]
    finally:
        be.set_coulomb_cutoff(0.0)
    # exxdiv='vcut_ws': the REFERENCE's constant for the diamond primitive cell, mesh 11^3, 2x2x2 k-mesh, k = kpts[2]
    # (pyscf/pbc/tools/test/test_pbc.py:26-41), through the product's table (pbc_tools.wigner_seitz_kernel) and the device lookup
    a_d = np.array([[0., 1.7834, 1.7834], [1.7834, 0., 1.7834], [1.7834, 1.7834, 0.]]) / 0.52917721092
    dcell = gto.Cell(atom=[('C', (0., 0., 0.)), ('C', (0.8917, 0.8917, 0.8917))], a=a_d * 0.52917721092, basis='gth-szv',
                     mesh=(11, 11, 11), pseudo='gth-pade')
    k2 = dcell.make_kpts([2, 2, 2])[2]
    ws = pbc_tools.wigner_seitz_kernel(dcell.lattice_vectors(), [2, 2, 2])
    be.set_coulomb_ws(ws)
    try:
        got = be.to_host(be.coulG_q([11, 11, 11], dcell.lattice_vectors(), k2))
        got0 = be.to_host(be.coulG_q([11, 11, 11], dcell.lattice_vectors(), np.zeros(3)))
    finally:
        be.set_coulomb_ws(None)
    assert abs(otools.fp(got) - 1.3245365170998518) < 1e-8
    ows = otools.precompute_exx(dcell.lattice_vectors(), [2, 2, 2])
    assert abs(got - otools.get_coulG(dcell.lattice_vectors(), [11, 11, 11], k2, ws=ows)).max() < 1e-11 * abs(got).max()
    assert abs(got0 - otools.get_coulG(dcell.lattice_vectors(), [11, 11, 11], np.zeros(3), ws=ows)).max() < 1e-11 * abs(got0).max()


def test_coulomb_Wq(be):
    """M^q and W^q for q = 0 and a generic q against the oracle's complex FFT construction."""
    cell, coords, Ls, rcut, kpts, aos, dms = _setup()
    X = kisdf.periodic_stack(aos, coords, kpts)
    a, mesh = cell.lattice_vectors(), cell.mesh
    G = X.shape[1]
    piv, _ = kisdf.select_ip(X, 24)
    theta = kisdf.fit_theta(X, piv)
    P = len(piv)
    w = cell.vol / G
    for q in (np.zeros(3), kpts[1] - kpts[0], kpts[0] - kpts[1]):
        ref = kisdf.build_Wq(theta, a, mesh, q, coords[piv])
        Wre = be.empty((P, P)); Wim = be.empty((P, P)); Wc = be.empty((P, P), dtype=torch.complex128)
        be.coulomb_Wq(be.to_device(theta), mesh, be.coulG_q(mesh, cell.lattice_vectors(), q), w, 0, P, 7, Wre, Wim, upper_only=True)
        be.symmetrize_hermitian(Wre, Wim)
        be.finish_Wq(Wre, Wim, be.to_device(np.exp(-1j * coords[piv].dot(q))), Wc)
        assert abs(be.to_host(Wc) - ref).max() < 1e-10 * abs(ref).max()


@pytest.mark.parametrize('mesh', [(12, 10, 9), (16, 15, 20), (6, 45, 50), (5, 96, 96), (8, 27, 25), (9, 7, 8)])
def test_coulomb_Wq_own_fft_matches_hipfft_and_numpy(be, mesh):
    """isdf_coulomb_Wq through the own k-point FFT (real-input forward through the half spectrum, table-expanding multiply,
    complex (y, z) plane inverse) against hipFFT Z2Z (own_fft = 0) and against numpy, for a kernel table WITHOUT inversion
    symmetry (what q != 0 gives); (9, 7, 8) has a factor 7 and takes the hipFFT form either way."""
    rng = np.random.default_rng(sum(mesh))
    G = int(np.prod(mesh))
    P = 11
    theta = rng.standard_normal((P, G))
    tab = rng.random(G) + 0.1
    w = 0.37
    V = np.fft.ifftn(np.fft.fftn(theta.reshape(P, *mesh), axes=(1, 2, 3)) * tab.reshape(mesh), axes=(1, 2, 3)).reshape(P, G)
    ref = w * V.dot(theta.T)
    out = {}
    try:
        for own in (2, 0):
            be.set_option('own_fft', own)
            Wre = be.empty((P, P)); Wim = be.empty((P, P))
            be.coulomb_Wq(be.to_device(theta), np.asarray(mesh), be.to_device(tab), w, 0, P, 4, Wre, Wim)
            out[own] = be.to_host(Wre) + 1j * be.to_host(Wim)
    finally:
        be.set_option('own_fft', 2)
    scale = abs(ref).max()
    assert abs(out[2] - ref).max() < 1e-12 * scale and abs(out[0] - ref).max() < 1e-12 * scale


@pytest.mark.parametrize('select', ['global', 'local', 'refined'])
def test_isdf_kpts_end_to_end(select):
    """ISDF(cell, kpts).get_jk: J exact (vs the reference formula), K converging to the exact k-point
    exchange; and equal to the oracle's k-ISDF on the same points."""
    from pyscf_isdf_amd.isdf import ISDF
    cell, coords, Ls, rcut, kpts, aos, dms = _setup()
    a, mesh = cell.lattice_vectors(), cell.mesh
    vj_ref, vk_ref = fftdf.get_jk_kpts(aos, dms, a, mesh, coords, kpts)
    df = ISDF(cell, kpts=kpts, c_isdf=20, select=select)
    df.k_ip_factor = 2
    vj, vk = df.get_jk(dms, kpts=kpts)
    assert vj.shape == dms.shape and vj.dtype == np.complex128 and vk.dtype == np.complex128
    assert abs(vj - vj_ref).max() < 1e-10
    assert abs(vk - vk_ref).max() < 2e-5 * abs(vk_ref).max()
    # same points through the oracle's k-ISDF formulas (explicit Theta, regularised like the product)
    X = kisdf.periodic_stack(aos, coords, kpts)
    theta = kisdf.fit_theta(X, df.ip, df.reg_used)
    qs, qidx = kisdf.unique_q(kpts)
    Ws = [kisdf.build_Wq(theta, a, mesh, q, coords[df.ip]) for q in qs]
    vk_or = kisdf.get_k_kpts([ao[df.ip] for ao in aos], Ws, qidx, dms)
    assert abs(vk - vk_or).max() < 1e-8 * abs(vk_or).max()


def test_isdf_kpts_fit_routes_agree():
    """k-point build: the block-Jacobi route and the Cholesky route give the same K; with bj_auto_kpts 'auto' verifies the
    block-Jacobi route on W^{q=0} (real: the Gamma-point probe check with the densities sum_k u^k* R u^k)."""
    from pyscf_isdf_amd.isdf import ISDF
    cell, coords, Ls, rcut, kpts, aos, dms = _setup()
    out = {}
    for route in ('cholesky', 'blockjacobi'):
        df = ISDF(cell, kpts=kpts, c_isdf=10, select='local')
        df.fit_route = route
        out[route] = (df.get_jk(dms, kpts=kpts, with_j=False)[1], df.ip.copy())
    assert np.array_equal(out['cholesky'][1], out['blockjacobi'][1])
    d = out['cholesky'][0] - out['blockjacobi'][0]
    assert abs(d).max() < 1e-6 * abs(out['cholesky'][0]).max()
    df = ISDF(cell, kpts=kpts, c_isdf=10, select='local')
    vk = df.get_jk(dms, kpts=kpts, with_j=False)[1]
    assert df.fit_route == 'auto' and df.fit_route_used == 'cholesky'       # k-points: opt-in (bj_auto_kpts)
    assert abs(vk - out['cholesky'][0]).max() < 1e-12 * abs(vk).max()
    df = ISDF(cell, kpts=kpts, c_isdf=10, select='local')
    df.bj_auto_kpts = True
    vk = df.get_jk(dms, kpts=kpts, with_j=False)[1]
    assert df.fit_route_used == 'blockjacobi' and 0 < df.bj_check <= df.bj_check_tol
    assert abs(vk - out['blockjacobi'][0]).max() < 1e-12 * abs(vk).max()
    df = ISDF(cell, kpts=kpts, c_isdf=10, select='local')
    df.bj_auto_kpts = True
    df.bj_check_tol = 1e-14                                  # force the fallback
    import warnings
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        vk = df.get_jk(dms, kpts=kpts, with_j=False)[1]
    assert df.fit_route_used == 'cholesky' and len(rec) == 1
    assert abs(vk - out['cholesky'][0]).max() < 1e-12 * abs(vk).max()


def test_range_separated_kpoint_jk():
    """get_jk(omega=...) with k-points: W^q rebuilt with the attenuated kernel (pbc.py:408-418) from the same fit; J against
    the reference formula with that kernel, long range + short range = full for J and K."""
    from pyscf_isdf_amd.isdf import ISDF
    cell, coords, Ls, rcut, kpts, aos, dms = _setup()
    a, mesh = cell.lattice_vectors(), cell.mesh
    df = ISDF(cell, kpts=kpts, c_isdf=6, select='local')
    vj, vk = df.get_jk(dms, kpts=kpts)
    vjl, vkl = df.get_jk(dms, kpts=kpts, omega=0.5)
    vjs, vks = df.get_jk(dms, kpts=kpts, omega=-0.5)
    assert sorted(df._W_omega) == [-0.5, 0.5]
    assert abs(vjl + vjs - vj).max() < 1e-11 and abs(vkl + vks - vk).max() < 1e-10 * abs(vk).max()
    assert 0.05 < abs(vkl).max() / abs(vk).max() < 0.9
    # J with the attenuated kernel, reference formula (fft_jk.py:63-107)
    G = len(coords)
    rho = sum(np.einsum('gi,ij,gj->g', ao, d, ao.conj()) for ao, d in zip(aos, dms)) / len(kpts)
    vR = otools.ifft(otools.get_coulG(a, mesh, omega=0.5) * otools.fft(rho, mesh), mesh) * (cell.vol / G)
    vj_ref = np.array([ao.conj().T.dot(vR[:, None] * ao) for ao in aos])
    assert abs(vjl - vj_ref).max() < 1e-10


def test_non_hermitian_density_matrices():
    """hermi=0: the density is complex; J from its real and imaginary parts equals the reference formula (fft_jk.py:63-107),
    K (which never assumed a Hermitian D) the oracle's k-ISDF on the same points."""
    from pyscf_isdf_amd.isdf import ISDF
    cell, coords, Ls, rcut, kpts, aos, dms = _setup()
    rng = np.random.default_rng(3)
    dmn = dms + 0.3 * (rng.standard_normal(dms.shape) + 1j * rng.standard_normal(dms.shape))
    a, mesh = cell.lattice_vectors(), cell.mesh
    vj_ref, vk_ref = fftdf.get_jk_kpts(aos, dmn, a, mesh, coords, kpts)
    df = ISDF(cell, kpts=kpts, c_isdf=20, select='global')
    df.k_ip_factor = 2
    vj, vk = df.get_jk(dmn, hermi=0, kpts=kpts)
    assert abs(vj - vj_ref).max() < 1e-10
    assert abs(vk - vk_ref).max() < 5e-5 * abs(vk_ref).max()


def test_kpts_band_reproduces_the_reference_pin():
    """get_jk(kpts=4 random k, kpts_band=2 k): the reference's own constant for this call (pyscf/pbc/df/test/
    test_fft.py:555-557,663-676, exact FFTDF exchange) is reproduced by the ISDF path once the point set reaches the
    numerical rank of the pair space (global selection, select_tol=0): J exactly, K to the fitting error left at
    P = rank.  Also the shapes df_jk._format_jks gives (df_jk.py:1426-1444)."""
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
    df = ISDF(cell, kpts=kpts, c_isdf=60, select='global')
    df.select_tol = 0.0
    df.reg_rel = 1e-13
    df.k_ip_factor = 2
    vj, vk = df.get_jk(dms, kpts=kpts, kpts_band=kpts_band)
    assert vj.shape == vk.shape == (2, nao, nao) and vk.dtype == np.complex128
    assert abs(otools.fp(vk) - (10.239828255099447 + 2.1190549216896182j)) < 5e-6
    # exact J and the exact K of the oracle's restatement of the reference algorithm on the same inputs
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    ao_k = [np.asarray(x, dtype=complex) for x in oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpts, rule='point')]
    ao_b = [np.asarray(x, dtype=complex) for x in oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpts_band, rule='point')]
    vj_ref, vk_ref = fftdf.get_jk_kpts(ao_k, dms, cell.lattice_vectors(), cell.mesh, coords, kpts, ao_band=ao_b, kpts_band=kpts_band)
    assert abs(vj - vj_ref).max() < 1e-9
    assert abs(vk - vk_ref).max() < 2e-5 * abs(vk_ref).max()
    # a single band vector that is not one of the k-points: (nao, nao) results, the stack grows by one k-point
    v1j, v1k = df.get_jk(dms, kpts=kpts, kpts_band=np.array([0.1, 0.2, 0.3]))
    assert v1j.shape == v1k.shape == (nao, nao) and df._nk_stack == 5
    # and back to the plain k-point call
    v2j, v2k = df.get_jk(dms, kpts=kpts)
    assert v2k.shape == (4, nao, nao) and abs(v2k[:2] - vk).max() < 1e-4 * abs(vk).max()


def test_kpoint_ao_eri_reproduces_the_reference_pins():
    """get_ao_eri with k-points from the factorisation: the reference's constants for one k-point and for four k-points
    that conserve momentum (pyscf/pbc/df/test/test_fft.py:555-557,690-703) at full rank of the pair space."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he_c()
    np.random.seed(1)
    kpts = np.random.random((4, 3))
    kpts[3] = kpts[0] - kpts[1] + kpts[2]
    nao = cell.nao_nr()
    df = ISDF(cell, kpts=kpts, c_isdf=60, select='global')
    df.select_tol = 0.0
    df.reg_rel = 1e-13
    df.k_ip_factor = 2
    eri4 = df.get_ao_eri(kpts)
    assert eri4.shape == (nao * nao, nao * nao) and eri4.dtype == np.complex128
    assert abs(otools.fp(eri4) - (0.33709288394542991 - 0.94185725001175313j)) < 5e-6
    d1 = ISDF(cell, kpts=kpts[:1], c_isdf=40, select='global')
    d1.select_tol = 0.0
    d1.reg_rel = 1e-13
    eri1 = d1.get_ao_eri(kpts[0])
    assert abs(otools.fp(eri1) - (2.9346374584901898 - 0.20479054936744959j)) < 1e-6
    with pytest.raises(ValueError):
        df.get_ao_eri(np.array([kpts[0], kpts[1], kpts[2], kpts[2]]))          # momentum not conserved
    # MO integrals = the AO integrals transformed (test_fft.py:766-775 checks the same identity on the reference)
    np.random.seed(5)
    mo = np.random.random((nao, nao)) + np.random.random((nao, nao)) * 1j
    mo1 = mo[:, :nao // 2 + 1]
    eri_mo = df.ao2mo((mo1, mo, mo, mo1), kpts)
    ref = np.einsum('pqrs,pi,qj,rk,sl->ijkl', eri4.reshape((nao,) * 4), mo1.conj(), mo, mo.conj(), mo1)
    assert eri_mo.shape == (mo1.shape[1] * nao, nao * mo1.shape[1])
    assert abs(eri_mo - ref.reshape(eri_mo.shape)).max() < 1e-9 * abs(ref).max()


def test_select_complex_mode_panel_from_global_memory(be):
    """Many AOs x k-points: the pivot panel (2 x 9000 doubles) no longer fits the LDS staging budget and
    is broadcast from global memory; pivots must still equal the plain-C oracle's."""
    rng = np.random.default_rng(8)
    nh, m, k = 4500, 1500, 12
    X = rng.standard_normal((2 * nh, m)) * np.exp(-2.0 * rng.random(m))
    piv_ref, L_ref = c_oracle.select_ip_cplx(X, k)
    L = be.zeros((k, m))
    piv = be.empty((1, k), dtype=torch.int64)
    rank = be.select_ip_cplx(be.to_device(X), nh, [0, m], [k], -1.0, 1e-10, L, piv)
    assert rank[0] == k and np.array_equal(be.to_host(piv)[0], piv_ref)
    assert abs(be.to_host(L) - L_ref).max() < 1e-9 * abs(L_ref).max()


def test_get_nuc_reference_pins():
    """ISDF.get_nuc against the reference's known answers for four random k-points
    (pyscf/pbc/df/test/test_fft.py:555-557,589-599): pins the k-point collocation kernel and the
    phi^H v phi contraction directly to reference constants."""
    from pyscf_isdf_amd.isdf import ISDF
    from oracle import pbc_tools as otools
    cell = cells.cell_he_c()
    np.random.seed(1)
    kpts = np.random.random((4, 3))
    kpts[3] = kpts[0] - kpts[1] + kpts[2]
    v1 = ISDF(cell).get_nuc(kpts)
    assert v1.shape == (4, 6, 6) and v1.dtype == np.complex128
    assert abs(otools.fp(v1[0]) - (-5.7646608099493841 + 0.19126294430138713j)) < 1e-8
    assert abs(otools.fp(v1[1]) - (-5.6567258309199193 + 0.86813371243952175j)) < 1e-8
    assert abs(otools.fp(v1[2]) - (-6.1528952645454895 + 0.09517054428060109j)) < 1e-8
    assert abs(otools.fp(v1[3]) - (-5.7445962879770942 + 0.24611951427601772j)) < 1e-8
    v0 = ISDF(cell).get_nuc()
    assert v0.shape == (6, 6) and v0.dtype == np.float64 and abs(v0 - v0.T).max() < 1e-12


def test_get_pp_reference_pins():
    """ISDF.get_pp (GTH local + non-local on the device) against the reference's known answers for four
    random k-points (pyscf/pbc/df/test/test_fft.py:601-611) and the oracle at Gamma."""
    from pyscf_isdf_amd.isdf import ISDF
    from oracle import pbc_tools as otools, pp as opp
    cell = cells.cell_he_c()
    np.random.seed(1)
    kpts = np.random.random((4, 3))
    kpts[3] = kpts[0] - kpts[1] + kpts[2]
    v1 = ISDF(cell).get_pp(kpts)
    assert v1.shape == (4, 6, 6)
    assert abs(otools.fp(v1[0]) - (-5.6240249083785869 + 0.22094834302524968j)) < 1e-8
    assert abs(otools.fp(v1[1]) - (-5.5387702576467603 + 1.0439333717227581j)) < 1e-8
    assert abs(otools.fp(v1[2]) - (-6.0530899866313366 + 0.2817289667029651j)) < 1e-8
    assert abs(otools.fp(v1[3]) - (-5.6011543542444446 + 0.27597306418805201j)) < 1e-8
    # Gamma point, d shells, two pseudo-atoms: vs the oracle
    cell = cells.cell_diamond_prim('gth-dzvp', (12, 12, 12))
    v0 = ISDF(cell).get_pp()
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    ao = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, rule='point')
    ps = [cell._pseudo.get(cell.atom_symbol(i)) for i in range(cell.natm)]
    ref = opp.get_pp(cell._atm, cell._bas, cell._env, cell.atom_coords(), cell.atom_charges(), ps, cell.lattice_vectors(),
                     cell.mesh, coords, [ao], np.zeros((1, 3)))[0]
    assert v0.dtype == np.float64 and abs(v0 - ref).max() < 1e-10


def test_supercell_kmesh_cross_check_on_gpu():
    """k2gamma on the GPU: a [2,2,1] k-mesh on the He2 triclinic cell against the Gamma point of its 2x2x1 supercell (the kind of
    cross-check pyscf/pbc/scf/test/test_khf.py:73 makes, here for J and K themselves).  J is exact on both sides: the mapped
    k-blocked J equals the supercell J to 1e-9 and the energies per cell to 1e-10; the two ISDF K agree to their fit error
    (both at numerical full rank) - which ties the k-point kernels (collocation with phases, coulG(q) with wrap-around,
    Z2Z convolution, W^q, complex Hadamard K) to the Gamma-point kernels through an identity, not through a shared oracle."""
    from pyscf_isdf_amd.isdf import ISDF
    from pyscf_isdf_amd import k2gamma
    cell = cells.cell_he2_triclinic()
    cell.mesh = np.array([10, 9, 9])
    kmesh = [2, 2, 1]
    kpts = cell.make_kpts(kmesh)
    nao, nk = cell.nao_nr(), 4
    scell, phase = k2gamma.get_phase(cell, kpts)
    rng = np.random.default_rng(8)
    c = rng.standard_normal((nk, nao, 2))
    dms = np.einsum('kpi,kqi->kpq', c, c)                       # all four points are time-reversal invariant: real D^k
    dm_sc = k2gamma.to_supercell_ao_integrals(cell, kpts, dms)
    assert abs(dm_sc.imag).max() < 1e-13
    dfk = ISDF(cell, kpts=kpts, c_isdf=30, select='global')
    dfk.select_tol, dfk.k_ip_factor = 0.0, 4
    vj, vk = dfk.get_jk(dms, kpts=kpts)
    dfs = ISDF(scell, c_isdf=30, select='global')
    dfs.select_tol = 0.0
    vjs, vks = dfs.get_jk(dm_sc.real)
    vj_map = k2gamma.to_supercell_ao_integrals(cell, kpts, vj)
    vk_map = k2gamma.to_supercell_ao_integrals(cell, kpts, vk)
    assert abs(vj_map.imag).max() < 1e-9 and abs(vj_map.real - vjs).max() < 1e-9
    assert abs(np.einsum('kij,kji', vj, dms).real - np.einsum('ij,ji', vjs, dm_sc.real)) / 2 / nk < 1e-10
    assert abs(vk_map.real - vks).max() < 1e-4 * abs(vks).max()
    ek_k = np.einsum('kij,kji', vk, dms).real / 4 / nk
    ek_s = np.einsum('ij,ji', vks, dm_sc.real) / 4 / nk
    assert abs(ek_k - ek_s) < 1e-5 * abs(ek_s)


def test_exact_kpoint_exchange_on_device_reproduces_the_reference_pin():
    """isdf_get_k_exact_kpt (the reference's k-point exchange loop, fft_jk.py:250-292, in periodic parts on the device) pinned
    DIRECTLY to the reference's constant for 4 random k-points and 2 band k-points with MO-tagged random density matrices
    (pyscf/pbc/df/test/test_fft.py:555-557,663-676): fp(vk1) = 10.239828255099447+2.1190549216896182j to 1e-8, and to the
    oracle's restatement element-wise (1e-10); a row sample (rows=) equals the same rows of the full result; untagged
    Hermitian positive semidefinite density matrices go through their eigenvectors."""
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
    df = ISDF(cell, kpts=kpts)
    vk = df.get_k_exact(dms, mo_coeff=mo_coeff, mo_occ=mo_occ, kpts_band=kpts_band)
    assert vk.shape == (2, nao, nao)
    assert abs(otools.fp(vk) - (10.239828255099447 + 2.1190549216896182j)) < 1e-8
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    ao_k = [np.asarray(x, dtype=complex) for x in oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpts, rule='point')]
    ao_b = [np.asarray(x, dtype=complex) for x in oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpts_band, rule='point')]
    vk_ref = fftdf.get_jk_kpts(ao_k, dms, cell.lattice_vectors(), cell.mesh, coords, kpts, ao_band=ao_b, kpts_band=kpts_band)[1]
    assert abs(vk - vk_ref).max() < 1e-10 * abs(vk_ref).max()
    part = df.get_k_exact(dms, mo_coeff=mo_coeff, mo_occ=mo_occ, kpts_band=kpts_band, rows=(1, 3), max_rows=int(mo_occ.sum(axis=1).max()))
    assert abs(part - vk[:, 1:4]).max() < 1e-12
    # SCF k-points themselves, orbitals recovered from Hermitian positive semidefinite matrices
    rng = np.random.default_rng(5)
    dmh = []
    for k in range(nk):
        c = np.linalg.qr(rng.standard_normal((nao, nao)) + 1j * rng.standard_normal((nao, nao)))[0][:, :2]
        dmh.append(2 * c.dot(c.conj().T))
    dmh = np.array(dmh)
    vk2 = df.get_k_exact(dmh)
    vk2_ref = fftdf.get_jk_kpts(ao_k, dmh, cell.lattice_vectors(), cell.mesh, coords, kpts)[1]
    assert abs(vk2 - vk2_ref).max() < 1e-10 * abs(vk2_ref).max()


def test_robust_k_at_kpoints_on_device():
    """robust_k at k-points on the GPU (isdf_coulomb_rows_q + plane GEMMs + isdf_zhadamard_planes): K = K1 + K1^H - K_isdf equals
    the oracle's direct formula on the same points (1e-8), SCF k-points and band k-points, and cuts the error against the exact
    k-point exchange evaluated on the device by more than 10x at equal P."""
    from pyscf_isdf_amd.isdf import ISDF
    from oracle import kisdf as okisdf
    cell = cells.cell_he2_triclinic()
    kpts = cell.make_kpts([2, 1, 1])
    band = np.array([[0.11, -0.07, 0.23]])
    nao = cell.nao_nr()
    rng = np.random.default_rng(2)
    c = rng.standard_normal((2, nao, nao)) + 1j * rng.standard_normal((2, nao, nao))
    dms = np.einsum('kpi,kqi->kpq', c[:, :, :2], c[:, :, :2].conj())
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    ao_k = [np.asarray(x, dtype=complex) for x in oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kpts, rule='point')]
    ao_b = [np.asarray(x, dtype=complex) for x in oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=band, rule='point')]
    plain = ISDF(cell, kpts=kpts, c_isdf=6, select='global')
    vk_plain = plain.get_jk(dms, kpts=kpts, with_j=False)[1]
    k_exact = plain.get_k_exact(dms)
    df = ISDF(cell, kpts=kpts, c_isdf=6, select='global')
    df.robust_k = True
    vk = df.get_jk(dms, kpts=kpts, with_j=False)[1]
    theta = okisdf.fit_theta(okisdf.periodic_stack(ao_k, coords, kpts), df.ip, df.reg_used)
    ref = okisdf.get_k_robust_kpts(ao_k, coords, kpts, cell.lattice_vectors(), cell.mesh, df.ip, theta, dms)
    assert abs(vk - ref).max() < 1e-8 * abs(ref).max()
    assert abs(vk - k_exact).max() < 0.1 * abs(vk_plain - k_exact).max()
    # band k-point: the stack holds the band k-point too (its own points and fit), the oracle follows with the same points
    vkb = df.get_jk(dms, kpts=kpts, kpts_band=band, with_j=False)[1]
    assert vkb.shape == (1, nao, nao)
    stack = okisdf.periodic_stack(ao_k + ao_b, coords, np.vstack([kpts, band]))
    theta_b = okisdf.fit_theta(stack, df.ip, df.reg_used)
    refb = okisdf.get_k_robust_kpts(ao_k, coords, kpts, cell.lattice_vectors(), cell.mesh, df.ip, theta_b, dms, ao_band=ao_b, kpts_band=band)
    assert abs(vkb - refb).max() < 1e-8 * abs(refb).max()


def test_even_mesh_pair_correction_on_device():
    """The +-q pairing with its Nyquist-plane correction (isdf_nyquist_spectra + scaled MFMA products) on an even, anisotropic mesh:
    K equals the build that makes every W^q from its own kernel table to 1e-10, on both fit routes; the uncorrected pairing of
    round 2 differs at the level of the Nyquist content."""
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he2_triclinic()
    cell.mesh = np.array([8, 10, 9])
    kpts = cell.make_kpts([2, 2, 1])
    nao = cell.nao_nr()
    rng = np.random.default_rng(4)
    c = rng.standard_normal((4, nao, nao)) + 1j * rng.standard_normal((4, nao, nao))
    dms = np.einsum('kpi,kqi->kpq', c[:, :, :2], c[:, :, :2].conj())
    for select, route in (('global', 'auto'), ('local', 'blockjacobi')):
        res = {}
        for mode in (False, 'auto', 'uncorrected'):
            df = ISDF(cell, kpts=kpts, c_isdf=4, select=select)
            df.kpt_pair_q, df.fit_route, df.bj_auto_kpts = mode, route, route == 'blockjacobi'
            res[mode] = df.get_jk(dms, kpts=kpts, with_j=False)[1]
        scale = abs(res[False]).max()
        assert abs(res['auto'] - res[False]).max() < 1e-10 * scale, (select, abs(res['auto'] - res[False]).max() / scale)
        assert abs(res['uncorrected'] - res[False]).max() > 1e-7 * scale


def test_kpoint_symmetry_jk_on_the_irreducible_kpoints():
    """A kpts_symm.KPoints object in place of the k-point array (pyscf/pbc/scf/khf_ksymm.py:210-237): density matrices on the
    irreducible k-points in, J / K on the irreducible k-points out; rotated to the full zone (transform_fock) they are the J / K
    of the plain call with the full-zone density matrices - J and the exact exchange to rounding (both are symmetric when the
    FFT mesh is), the ISDF K to its fitting error (the interpolation points are not a symmetric set).  Diamond, 3 x 3 x 3:
    4 irreducible k-points of 27, 48 operations with fractional translations."""
    import scipy.linalg
    from pyscf_isdf_amd.isdf import ISDF
    cell = gto.diamond_primitive('gth-szv', (16, 16, 16))
    kp = cell.make_kpts([3, 3, 3], space_group_symmetry=True, time_reversal_symmetry=True)
    assert kp.nkpts_ibz == 4 and kp.nop == 48
    nao, nk = cell.nao_nr(), kp.nkpts
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    aos = oao.eval_ao(cell._atm, cell._bas, cell._env, coords, Ls, rcut, kpts=kp.kpts)
    w = cell.vol / len(coords)
    v = np.zeros(len(coords))
    for ra in cell.atom_coords():
        for L in Ls:
            d = coords - (ra + L)
            v -= np.exp(-0.8 * np.einsum('gx,gx->g', d, d))
    mos, occs, dms = [], [], []
    for ao in aos:                                        # a symmetric model Hamiltonian's two lowest bands, doubly occupied
        S, V = w * ao.conj().T.dot(ao), w * (ao.conj().T * v).dot(ao)
        e, c = scipy.linalg.eigh(V, S)
        occ = np.zeros(nao); occ[:4] = 2.0
        mos.append(c); occs.append(occ); dms.append((c * occ).dot(c.conj().T))
    dm_bz = np.array(dms)
    dm_ibz = dm_bz[kp.ibz2bz]
    assert abs(kp.transform_dm(dm_ibz) - dm_bz).max() < 1e-9
    df = ISDF(cell, kpts=kp, c_isdf=25, select='local')
    df.k_ip_factor = 2
    assert df.kpts.shape == (27, 3)
    vj_i, vk_i = df.get_jk(dm_ibz)
    assert vj_i.shape == vk_i.shape == (4, nao, nao)
    with pytest.raises(RuntimeError):
        df.get_jk(dm_bz)
    ref = ISDF(cell, kpts=kp.kpts, c_isdf=25, select='local')
    ref.k_ip_factor = 2
    vj_b, vk_b = ref.get_jk(dm_bz)
    assert abs(vj_i - vj_b[kp.ibz2bz]).max() < 1e-9 and abs(vk_i - vk_b[kp.ibz2bz]).max() < 1e-8
    # the device collocation drops shell images below the cell's precision block by block - not a symmetric rule, so the rotated
    # images agree to that truncation (6e-7 here), not to rounding as with the oracle's distance rule (tests/test_kpts_symm.py)
    assert abs(kp.transform_fock(vj_i) - vj_b).max() < 5e-6
    # exact exchange: symmetric to the same level; the ISDF K deviates from it - and from its own rotated images - by the fit error
    kx_b = ref.get_k_exact(dm_bz, mo_coeff=np.array(mos), mo_occ=np.array(occs))
    kx_i = ref.get_k_exact(dm_bz, mo_coeff=np.array(mos), mo_occ=np.array(occs), kpts_band=kp.kpts_ibz)
    assert abs(kx_i - kx_b[kp.ibz2bz]).max() < 1e-9
    assert abs(kp.transform_fock(kx_i) - kx_b).max() < 5e-6
    err = abs(vk_b - kx_b).max()
    assert err < 2e-4 * abs(kx_b).max()
    assert abs(kp.transform_fock(vk_i) - vk_b).max() < 4 * err
    # energies through the weights of the irreducible k-points
    ek_bz = np.einsum('kij,kji', vk_b, dm_bz).real / nk
    ek_ibz = np.einsum('k,kij,kji', kp.weights_ibz, vk_i, dm_ibz).real
    assert abs(ek_ibz - ek_bz) < 4 * err * nao
