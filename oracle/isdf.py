"""Oracle: the ISDF stages on the CPU (numpy).  TEST INFRASTRUCTURE ONLY.

The mounted reference holds no ISDF code, so this file IS the specification
the HIP kernels are checked against; conventions are anchored to the reference where it has them:

* pivot rule / stopping tolerance: pyscf/lib/scipy_helper.py:71-110 (argmax of the residual diagonal,
  L[k,k] = sqrt(D[k]), D -= L[:,k]^2, tol = n*eps*max(diag) when tol < 0);
* grid order, weights w = vol/G: pyscf/pbc/gto/cell.py:874-898, pyscf/pbc/dft/gen_grid.py:88-96;
* FFT / Coulomb kernel: pyscf/pbc/tools/pbc.py:149-211,230-420 (see oracle/pbc_tools.py);
* normalisation of W: pyscf/pbc/df/fft_ao2mo.py:154-184, (ij|kl) = sum_r [ifft(fft(rho_ij) coulG vol/G)].real rho_kl;
* J exactly as pyscf/pbc/df/fft_jk.py:63-107; output shapes as pyscf/pbc/df/df_jk.py:1426-1444.

Layout: ``aoT`` is (nao, m) — AO-major, grid index contiguous (the device layout).

Interpolation points are chosen by pivoted Cholesky of the pair-density Gram matrix
A(r,r') = (sum_mu phi_mu(r) phi_mu(r'))^2, which is never formed: its diagonal is
(sum_mu phi_mu(r)^2)^2 and column p is (aoT^T aoT[:,p])^2.  The Cholesky rows L (k, m) double as the
fit: with T = L[:, piv] (upper triangular), the least-squares interpolation vectors
Theta = A_PP^-1 A_P equal T^-1 L.

Tie rule: exact argmax is not reproducible across CPU/GPU arithmetic on symmetric crystals, where
symmetry-equivalent grid points tie to the last bit.  ``tie_rtol`` picks the LOWEST grid index whose
residual is within (1 - tie_rtol) of the maximum; tie_rtol = 0 is the reference rule.
"""
import numpy as np
import scipy.linalg
from . import pbc_tools as tools

TIE_RTOL = 1e-10


def select_ip(aoT, k, tol=-1.0, tie_rtol=TIE_RTOL):
    """Pivoted Cholesky of the implicit Gram matrix.  Returns (piv[int64 rank], L[rank, m])."""
    aoT = np.asarray(aoT, dtype=float)
    nao, m = aoT.shape
    k = min(k, m)
    d = np.einsum('ig,ig->g', aoT, aoT) ** 2
    if tol < 0:
        tol = m * np.finfo(float).eps * d.max()
    L = np.zeros((k, m))
    piv = np.zeros(k, dtype=np.int64)
    alive = np.ones(m, dtype=bool)
    rank = 0
    for j in range(k):
        dmax = d.max()
        if dmax <= tol:
            break
        p = int(np.argmax(d >= dmax * (1.0 - tie_rtol)))
        piv[j] = p
        col = aoT.T.dot(aoT[:, p]) ** 2
        if j:
            col -= L[:j].T.dot(L[:j, p])
        dp = np.sqrt(d[p])
        row = col / dp
        row[~alive] = 0.0          # residual rows of earlier pivots are exactly zero
        row[p] = dp
        L[j] = row
        d -= row * row
        alive[p] = False
        d[~alive] = -1.0           # never selectable again
        rank += 1
    return piv[:rank], L[:rank]


def pivoted_cholesky_gram(A, k, tol=-1.0, tie_rtol=TIE_RTOL):
    """Pivoted Cholesky of an EXPLICIT symmetric positive semidefinite matrix — the reference's
    pivoted_cholesky_python (pyscf/lib/scipy_helper.py:71-110: argmax of the residual diagonal, L[k,k] = sqrt(D[k]),
    D -= L[:,k]^2, tol = n*eps*max(diag) when tol < 0) plus the tie rule above (tie_rtol = 0: the reference rule).
    Restates include/mi355_isdf.h isdf_select_ip_gram.  Returns (piv[rank], L[rank, m])."""
    A = np.array(A, dtype=float)
    m = A.shape[0]
    k = min(k, m)
    d = A.diagonal().copy()
    if tol < 0:
        tol = m * np.finfo(float).eps * d.max()
    L = np.zeros((k, m))
    piv = np.zeros(k, dtype=np.int64)
    alive = np.ones(m, dtype=bool)
    rank = 0
    for j in range(k):
        dmax = d.max()
        if not dmax > tol:
            break
        p = int(np.argmax((d >= dmax * (1.0 - tie_rtol)) & (d > 0)))
        piv[j] = p
        col = A[p] - (L[:j].T.dot(L[:j, p]) if j else 0.0)
        dp = np.sqrt(d[p])
        row = col / dp
        row[~alive] = 0.0
        row[p] = dp
        L[j] = row
        d = np.maximum(d - row * row, 0.0)
        alive[p] = False
        d[~alive] = -1.0
        rank += 1
    return piv[:rank], L[:rank]


def refine_selection(aoT, cand, k, tol=-1.0, tie_rtol=TIE_RTOL):
    """select='refined', second stage: the final k points among the candidate grid indices ``cand`` by pivoted Cholesky of
    the pair-density Gram matrix restricted to the candidates.  Returns the chosen grid indices in pivot order."""
    aoC = aoT[:, cand]
    piv, _ = pivoted_cholesky_gram(aoC.T.dot(aoC) ** 2, k, tol=tol, tie_rtol=tie_rtol)
    return np.asarray(cand)[piv]


def fit_theta(L, piv):
    """Theta (k, m) = T^-1 L with T = L[:, piv] upper triangular."""
    T = np.triu(L[:, piv])
    return scipy.linalg.solve_triangular(T, L, lower=False)


def fit_theta_normal_equations(aoT, piv):
    """The textbook fit  (phi_P phi_P^T)^2 Theta = (phi_P phi^T)^2  (SURVEY 7.1-3), for cross-checks."""
    aoP = aoT[:, piv]
    A = aoP.T.dot(aoP) ** 2
    B = aoP.T.dot(aoT) ** 2
    return np.linalg.lstsq(A, B, rcond=None)[0]


def coulomb_V(theta, a, mesh, omega=None, rc=None, ws=None):
    """V_P = ifft(coulG * fft(Theta_P)).real, rows over P (k, G); omega: range separation (pbc.py:408-418); rc: spherical
    truncation (exxdiv='vcut_sph', pbc.py:312-317)."""
    coulG = tools.get_coulG(a, mesh, omega=omega, rc=rc, ws=ws)
    return tools.ifft(tools.fft(theta, mesh) * coulG, mesh).real


def build_W(theta, a, mesh, omega=None, rc=None):
    G = theta.shape[1]
    w = abs(np.linalg.det(a)) / G
    V = coulomb_V(theta, a, mesh, omega, rc)
    return w * V.dot(theta.T)


def build_W_spectral(theta, a, mesh, sphere_pct=0.0, omega=None, rc=None):
    """build_W in its Parseval form: with the conventions of pbc/tools/pbc.py:149-211 (fft unscaled, ifft 1/N)
    w sum_r Theta_P(r) conv(Theta_Q)(r) = (w / N) sum_G coulG(G) fft(Theta_P)(G) conj(fft(Theta_Q)(G)) = (X X^T)_PQ with
    X_P = sqrt(w coulG / N) fft(Theta_P) (real and imaginary parts as separate columns).  sphere_pct > 0 keeps only the G inside
    that percentage of the radius of the sphere inscribed in the reciprocal FFT box (faces at the frequencies +-(n_i - 1) // 2):
    what include/mi355_isdf.h isdf_spectral_rows + isdf_gemm_nt compute (pyscf_isdf_amd/fit_route.py _spectral_plan)."""
    mesh = [int(x) for x in mesh]
    a = np.asarray(a, dtype=float)
    N = int(np.prod(mesh))
    w = abs(np.linalg.det(a)) / N
    coulG = tools.get_coulG(a, mesh, omega=omega, rc=rc)
    keep = coulG > 0
    if sphere_pct > 0:
        b = 2 * np.pi * np.linalg.inv(a).T
        Gv = tools.get_Gv(b, mesh)
        g2 = np.einsum('gi,gi->g', Gv, Gv)
        rmin = min(2 * np.pi * ((n - 1) // 2) / np.linalg.norm(a[i]) for i, n in enumerate(mesh)) * sphere_pct / 100.0
        keep &= g2 <= rmin * rmin * (1 + 1e-12)
    z = tools.fft(theta, mesh)[:, keep] * np.sqrt(w * coulG[keep] / N)
    X = np.concatenate([z.real, z.imag], axis=1)
    return X.dot(X.T)


def W_from_factor(S, Y, a, mesh):
    """W for Theta = S^-1 Y (S upper triangular) without forming Theta:
    W = S^-1 [w conv(Y) Y^T] S^-T  (include/mi355_isdf.h isdf_W_from_factor)."""
    M = build_W(Y, a, mesh)
    Z = scipy.linalg.solve_triangular(S, M, lower=False)
    return scipy.linalg.solve_triangular(S, Z.T, lower=False).T


# ---- global (dense) ISDF ---------------------------------------------------------------------
def build_global(aoT, a, mesh, nip, tie_rtol=TIE_RTOL):
    piv, L = select_ip(aoT, nip, tie_rtol=tie_rtol)
    theta = fit_theta(L, piv)
    W = build_W(theta, a, mesh)
    return dict(ip=piv, theta=theta, W=W, aoP=np.ascontiguousarray(aoT[:, piv].T))


# ---- block-local ISDF ------------------------------------------------------------------------
def partition_by_atom(coords, atom_coords, a, tie_atol=1e-9):
    """Voronoi partition of grid points by nearest atom under the minimum-image convention.
    Ties (|d - dmin| <= tie_atol) go to the lowest atom index.  Returns owner[G] (int32)."""
    Ts = np.array([[i, j, k] for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)], dtype=float).dot(a)
    G = len(coords)
    dmin = np.full(G, np.inf)
    owner = np.zeros(G, dtype=np.int32)
    for ia, R in enumerate(atom_coords):
        da = np.full(G, np.inf)
        for T in Ts:
            d = coords - (R + T)
            da = np.minimum(da, np.einsum('gx,gx->g', d, d))
        da = np.sqrt(da)
        better = da < dmin - tie_atol
        owner[better] = ia
        dmin = np.where(better, da, dmin)
    return owner


def build_local(aoT, a, mesh, owner, nip_per_block, tie_rtol=TIE_RTOL):
    """Block-local ISDF.  ``owner[g]`` = block id of grid point g; ``nip_per_block[b]`` = points to pick.

    Returns ip (global grid indices, block-major), blocks (list of dicts with idx, theta_b, ip_local),
    W (P,P) and aoP (P, nao).  Theta is block-sparse: Theta[P in b, idx_b] = theta_b, zero elsewhere.
    """
    nblk = len(nip_per_block)
    G = aoT.shape[1]
    blocks, ips = [], []
    for b in range(nblk):
        idx = np.nonzero(owner == b)[0]
        piv, L = select_ip(aoT[:, idx], nip_per_block[b], tie_rtol=tie_rtol)
        th = fit_theta(L, piv)
        blocks.append(dict(idx=idx, theta=th, ip_local=piv))
        ips.append(idx[piv])
    ip = np.concatenate(ips)
    P = len(ip)
    w = abs(np.linalg.det(a)) / G
    W = np.zeros((P, P))
    coulG = tools.get_coulG(a, mesh)
    r0 = 0
    for b in range(nblk):
        kb = len(blocks[b]['ip_local'])
        dense = np.zeros((kb, G))
        dense[:, blocks[b]['idx']] = blocks[b]['theta']
        V = tools.ifft(tools.fft(dense, mesh) * coulG, mesh).real
        c0 = 0
        for b2 in range(nblk):
            kb2 = len(blocks[b2]['ip_local'])
            W[r0:r0 + kb, c0:c0 + kb2] = w * V[:, blocks[b2]['idx']].dot(blocks[b2]['theta'].T)
            c0 += kb2
        r0 += kb
    return dict(ip=ip, blocks=blocks, W=W, aoP=np.ascontiguousarray(aoT[:, ip].T))


def fit_theta_global_chol(aoT, ip, reg_rel=0.0):
    """Regularised Cholesky fit used by the scalable variant:
    Theta = [(aoP^T aoP)^2 + reg_rel * max(diag) * I]^-1 (aoP^T ao)^2  (include/mi355_isdf.h S3b)."""
    aoP = aoT[:, ip]
    A = aoP.T.dot(aoP) ** 2
    if reg_rel > 0:
        A = A + reg_rel * np.diag(A).max() * np.eye(len(ip))
    B = aoP.T.dot(aoT) ** 2
    return scipy.linalg.cho_solve(scipy.linalg.cho_factor(A), B)


def build_local_select_global_fit(aoT, a, mesh, owner, nip_per_block, reg_rel=1e-12, tie_rtol=TIE_RTOL,
                                  select=None):
    """Interpolation points chosen independently inside each block (grid points regrouped block by
    block with a stable sort, exactly like the product), then ONE global least-squares fit."""
    select = select or select_ip
    perm = np.argsort(owner, kind='stable')
    counts = np.bincount(owner, minlength=len(nip_per_block))
    off = np.append(0, np.cumsum(counts))
    ips = []
    for b in range(len(nip_per_block)):
        idx = perm[off[b]:off[b + 1]]
        k = min(int(nip_per_block[b]), len(idx))
        if k == 0:
            continue
        piv, _ = select(aoT[:, idx], k, tie_rtol=tie_rtol)
        ips.append(idx[piv])
    ip = np.concatenate(ips)
    theta = fit_theta_global_chol(aoT, ip, reg_rel)
    W = build_W(theta, a, mesh)
    return dict(ip=ip, theta=theta, W=W, aoP=np.ascontiguousarray(aoT[:, ip].T))


def build_W_blockjacobi(aoT, ip, blk_off, a, mesh, reg_rel=1e-12, block_shift=0.0):
    """W without any triangular solve over the grid (include/mi355_isdf.h S3c):
    A <- A + reg (the fit's regularisation, as in the Cholesky route), D = blockdiag(chol(A_bb)), Y' = D^-1 B,
    M' = w conv(Y') Y'^T, A' = D^-1 A D^-T, W = D^-T [A'^-1 M' A'^-1] D^-1."""
    aoP = aoT[:, ip]
    A = aoP.T.dot(aoP) ** 2
    A = A + reg_rel * A.diagonal().max() * np.eye(len(ip))
    B = aoP.T.dot(aoT) ** 2
    P = len(ip)
    D = np.zeros((P, P))
    for b in range(len(blk_off) - 1):
        s = slice(blk_off[b], blk_off[b + 1])
        if s.stop > s.start:
            D[s, s] = np.linalg.cholesky(A[s, s] + block_shift * A.diagonal().max() * np.eye(s.stop - s.start))
    Yp = scipy.linalg.solve_triangular(D, B, lower=True)
    Mp = build_W(Yp, a, mesh)
    Ap = scipy.linalg.solve_triangular(D, scipy.linalg.solve_triangular(D, A, lower=True).T, lower=True).T
    cf = scipy.linalg.cho_factor(Ap)
    Wp = scipy.linalg.cho_solve(cf, scipy.linalg.cho_solve(cf, Mp).T).T
    return scipy.linalg.solve_triangular(D, scipy.linalg.solve_triangular(D, Wp, lower=True, trans='T').T, lower=True, trans='T').T


def theta_dense_from_blocks(blocks, G):
    P = sum(len(b['ip_local']) for b in blocks)
    th = np.zeros((P, G))
    r0 = 0
    for b in blocks:
        kb = len(b['ip_local'])
        th[r0:r0 + kb][:, b['idx']] = b['theta']
        r0 += kb
    return th


# ---- (AO x occupied orbital) pair space --------------------------------------------------------------
# The reference's K only ever forms the pair densities phi_mu psi_i of the OCCUPIED orbitals when the density matrix carries
# them (pyscf/pbc/df/fft_jk.py:206-210: mo_coeff[:, mo_occ > 0] * sqrt(mo_occ); :235-238, :276-287).  Interpolating that pair
# space instead of all AO pairs changes the Gram matrix from (phi^T phi)^2 to (phi^T phi) o (psi^T psi); points, fit, W and K
# keep their formulas (K = phi_P^T [(phi_P D phi_P^T) o W] phi_P with D = sum_i psi_i psi_i^T).
def occupied_on_grid(aoT, mo_coeff, mo_occ):
    """psi (nocc, m) = (C[:, occ > 0] sqrt(occ))^T phi; several density matrices: stack their orbitals."""
    occ = np.asarray(mo_occ, dtype=float)
    c = np.asarray(mo_coeff, dtype=float)[:, occ > 0] * np.sqrt(occ[occ > 0])
    return c.T.dot(aoT)


def refine_selection_occ(aoT, psi, cand, k, tol=-1.0, tie_rtol=TIE_RTOL):
    """refine_selection on the Gram matrix of the (AO x occupied) pair products restricted to the candidates."""
    aoC, psC = aoT[:, cand], psi[:, cand]
    piv, _ = pivoted_cholesky_gram(aoC.T.dot(aoC) * psC.T.dot(psC), k, tol=tol, tie_rtol=tie_rtol)
    return np.asarray(cand)[piv]


def fit_theta_occ_chol(aoT, psi, ip, reg_rel=0.0):
    """Theta = [(aoP^T aoP) o (psiP^T psiP) + reg]^-1 [(aoP^T ao) o (psiP^T psi)]  (fit_theta_global_chol in the
    (AO x occupied) pair space; include/mi355_isdf.h isdf_gram_prod / isdf_pair_prod_rows / isdf_factor_solve_half)."""
    aoP, psP = aoT[:, ip], psi[:, ip]
    A = aoP.T.dot(aoP) * psP.T.dot(psP)
    if reg_rel > 0:
        A = A + reg_rel * np.diag(A).max() * np.eye(len(ip))
    B = aoP.T.dot(aoT) * psP.T.dot(psi)
    return scipy.linalg.cho_solve(scipy.linalg.cho_factor(A), B)


def build_W_blockjacobi_occ(aoT, psi, ip, blk_off, a, mesh, reg_rel=1e-12, block_shift=0.0):
    """build_W_blockjacobi with the product Gram matrices of the (AO x occupied) pair space."""
    aoP, psP = aoT[:, ip], psi[:, ip]
    A = aoP.T.dot(aoP) * psP.T.dot(psP)
    A = A + reg_rel * A.diagonal().max() * np.eye(len(ip))
    B = aoP.T.dot(aoT) * psP.T.dot(psi)
    P = len(ip)
    D = np.zeros((P, P))
    for b in range(len(blk_off) - 1):
        s = slice(blk_off[b], blk_off[b + 1])
        if s.stop > s.start:
            D[s, s] = np.linalg.cholesky(A[s, s] + block_shift * A.diagonal().max() * np.eye(s.stop - s.start))
    Yp = scipy.linalg.solve_triangular(D, B, lower=True)
    Mp = build_W(Yp, a, mesh)
    Ap = scipy.linalg.solve_triangular(D, scipy.linalg.solve_triangular(D, A, lower=True).T, lower=True).T
    cf = scipy.linalg.cho_factor(Ap)
    Wp = scipy.linalg.cho_solve(cf, scipy.linalg.cho_solve(cf, Mp).T).T
    return scipy.linalg.solve_triangular(D, scipy.linalg.solve_triangular(D, Wp, lower=True, trans='T').T, lower=True, trans='T').T


# ---- J / K -------------------------------------------------------------------------------------
def get_j(aoT, dm, a, mesh, omega=None):
    """Exact J (FFTDF formula) in the (nao, G) layout."""
    dms = np.asarray(dm, dtype=float)
    shape = dms.shape
    dms = dms.reshape(-1, shape[-2], shape[-1])
    G = aoT.shape[1]
    w = abs(np.linalg.det(a)) / G
    coulG = tools.get_coulG(a, mesh, omega=omega)
    vj = np.empty_like(dms)
    for i, d in enumerate(dms):
        rho = np.einsum('ig,ig->g', d.dot(aoT), aoT)
        vR = tools.ifft(coulG * tools.fft(rho, mesh), mesh).real * w
        vj[i] = (aoT * vR).dot(aoT.T)
    return vj.reshape(shape)


def get_k(aoP, W, dm):
    """K_mn = sum_PQ phi_m(r_P) [ (phi_P D phi_P^T)_PQ W_PQ ] phi_n(r_Q);  aoP is (P, nao)."""
    dms = np.asarray(dm, dtype=float)
    shape = dms.shape
    dms = dms.reshape(-1, shape[-2], shape[-1])
    vk = np.empty_like(dms)
    for i, d in enumerate(dms):
        M = aoP.dot(d).dot(aoP.T) * W
        vk[i] = aoP.T.dot(M).dot(aoP)
    return vk.reshape(shape)


def isdf_eri_s4(aoP, W):
    """(ij|kl) ~ sum_PQ phi_iP phi_jP W_PQ phi_kQ phi_lQ, packed s4 like fftdf.get_ao_eri_s4."""
    nao = aoP.shape[1]
    i, j = np.tril_indices(nao)
    X = aoP[:, i] * aoP[:, j]            # (P, npair)
    return X.T.dot(W).dot(X)


def get_k_robust(aoT, ip, theta, dm, a, mesh):
    """K with Dunlap's robust correction: for rho_mn ~ sum_P Theta_P rho_mn(r_P),
    (mu lam|sig nu) ~ (fit|exact) + (exact|fit) - (fit|fit):  K = K1 + K2 - K_isdf,
    K1_mn = sum_P phi_m(P) sum_g w V_P(g) [phi_P D phi(g)] phi_n(g), V_P = conv(Theta_P), K2 = K1(D^T)^T."""
    dm = np.asarray(dm, dtype=float)
    G = aoT.shape[1]
    w = abs(np.linalg.det(a)) / G
    aoP = np.ascontiguousarray(aoT[:, ip].T)
    V = coulomb_V(theta, a, mesh)
    W = w * V.dot(theta.T)

    def k1(D):
        F = aoP.dot(D).dot(aoT) * V                       # (P, G)
        return aoP.T.dot(w * F.dot(aoT.T))
    return k1(dm) + k1(dm.T).T - get_k(aoP, W, dm)

