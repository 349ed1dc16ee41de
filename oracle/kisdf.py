"""Oracle: ISDF with k-points (numpy).  TEST INFRASTRUCTURE ONLY.

Not in the reference; the formulas below are derived from the reference's exact
k-point exchange, pyscf/pbc/df/fft_jk.py:177-302 (pair density conj(ao1)*exp(-i q.r)*ao2 with
q = k2 - k1 :271-288, Coulomb kernel get_coulG(cell, q) :267-270, weight 1/nkpts * vol/G :217), and are
pinned by convergence to that path (tests/test_oracle_kisdf.py).

Bloch AOs phi^k = exp(i k.r) u^k.  The periodic parts of all pair products
conj(u^{k1}_m(r)) u^{k2}_n(r) are interpolated with ONE set of real, k-independent vectors Theta_P(r):

    conj(phi^{k1}_m(r)) phi^{k2}_n(r) ~ sum_P Theta_P(r) exp(i q.(r - r_P)) conj(phi^{k1}_m(r_P)) phi^{k2}_n(r_P)

Points and fit come from the real Gram matrix A(r,r') = |S(r,r')|^2, S = sum_{k,m} conj(u^k_m(r)) u^k_m(r'),
i.e. with the real stacked set X = [Re u; Im u] (2*nk*nao rows):  Re S = X^T X,  Im S = X^T X',
X' = [Im u; -Re u].  Then

    W^q_PQ = exp(-i q.(r_P - r_Q)) * (vol/G) * sum_r V^q_P(r) Theta_Q(r),   V^q_P = ifft(coulG(q) fft(Theta_P))
    K^{k1}  = 1/nk sum_{k2} aoP_{k1}^H [ (aoP_{k2} D^{k2} aoP_{k2}^H) .* W^{k2-k1} ] aoP_{k1}
    J       as pyscf/pbc/df/fft_jk.py:33-109 (exact).
"""
import numpy as np
import scipy.linalg
from . import pbc_tools as tools

TIE_RTOL = 1e-10


def periodic_stack(ao_kpts, coords, kpts):
    """X (2*nk*nao, G): rows [Re u^k_m] then [Im u^k_m], u^k = exp(-i k.r) phi^k."""
    us = []
    for ao, k in zip(ao_kpts, np.reshape(kpts, (-1, 3))):
        us.append((np.asarray(ao) * np.exp(-1j * coords.dot(k))[:, None]).T)     # (nao, G)
    u = np.vstack(us)
    return np.vstack([u.real, u.imag])


def _rot(X):
    nh = X.shape[0] // 2
    return np.vstack([X[nh:], -X[:nh]])


def select_ip(X, k, tol=-1.0, tie_rtol=TIE_RTOL):
    """Pivoted Cholesky of A = (X^T X)^2 + (X^T X')^2 (implicit).  Returns piv, L."""
    Xr = _rot(X)
    m = X.shape[1]
    k = min(k, m)
    d = np.einsum('ig,ig->g', X, X) ** 2
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
        col = X.T.dot(X[:, p]) ** 2 + Xr.T.dot(X[:, p]) ** 2
        if j:
            col -= L[:j].T.dot(L[:j, p])
        dp = np.sqrt(d[p])
        row = col / dp
        row[~alive] = 0.0
        row[p] = dp
        L[j] = row
        d -= row * row
        alive[p] = False
        d[~alive] = -1.0
        rank += 1
    return piv[:rank], L[:rank]


def fit_theta(X, ip, reg_rel=0.0):
    Xp = X[:, ip]
    Xr = _rot(X)
    A = Xp.T.dot(Xp) ** 2 + Xr[:, ip].T.dot(Xp) ** 2
    if reg_rel > 0:
        A = A + reg_rel * np.diag(A).max() * np.eye(len(ip))
    B = Xp.T.dot(X) ** 2 + Xp.T.dot(Xr) ** 2
    return scipy.linalg.cho_solve(scipy.linalg.cho_factor(A), B)


def coulomb_Vq(theta, a, mesh, q):
    coulG = tools.get_coulG(a, mesh, q)
    return tools.ifft(tools.fft(theta, mesh) * coulG, mesh)


def build_Wq(theta, a, mesh, q, r_ip):
    G = theta.shape[1]
    w = abs(np.linalg.det(a)) / G
    M = w * coulomb_Vq(theta, a, mesh, q).dot(theta.T)
    ph = np.exp(-1j * r_ip.dot(q))
    return M * ph[:, None] * ph.conj()[None, :]


def unique_q(kpts, kpts_band=None, decimals=9):
    """Distinct difference vectors k2 - k1 (k1 in band, k2 in kpts) and the index map [k1][k2] -> iq."""
    kpts = np.reshape(kpts, (-1, 3))
    band = kpts if kpts_band is None else np.reshape(kpts_band, (-1, 3))
    qs, index = [], np.zeros((len(band), len(kpts)), dtype=int)
    for i1, k1 in enumerate(band):
        for i2, k2 in enumerate(kpts):
            q = k2 - k1
            for iq, qq in enumerate(qs):
                if abs(qq - q).max() < 10.0 ** (-decimals):
                    index[i1, i2] = iq
                    break
            else:
                qs.append(q)
                index[i1, i2] = len(qs) - 1
    return np.array(qs), index


def get_k_kpts(aoP_kpts, Ws, qindex, dms, aoP_band=None):
    """K at the band k-points; aoP_* lists of (P, nao) complex; dms (nk, nao, nao)."""
    nk = len(aoP_kpts)
    band = aoP_kpts if aoP_band is None else aoP_band
    out = []
    for i1, a1 in enumerate(band):
        vk = 0
        for i2, a2 in enumerate(aoP_kpts):
            X = a2.dot(dms[i2]).dot(a2.conj().T)
            vk = vk + a1.conj().T.dot(X * Ws[qindex[i1, i2]]).dot(a1)
        out.append(vk / nk)
    return np.array(out)


def build(ao_kpts, coords, kpts, a, mesh, nip, kpts_band=None, reg_rel=0.0, tie_rtol=TIE_RTOL):
    X = periodic_stack(ao_kpts, coords, kpts)
    piv, L = select_ip(X, nip, tie_rtol=tie_rtol)
    theta = fit_theta(X, piv, reg_rel)
    qs, qindex = unique_q(kpts, kpts_band)
    r_ip = coords[piv]
    Ws = [build_Wq(theta, a, mesh, q, r_ip) for q in qs]
    return dict(ip=piv, theta=theta, qs=qs, qindex=qindex, W=Ws, aoP=[np.ascontiguousarray(ao[piv]) for ao in ao_kpts])


def get_k_robust_kpts(ao_kpts, coords, kpts, a, mesh, ip, theta, dms, ao_band=None, kpts_band=None):
    """K at k-points with Dunlap's robust correction: K = K1 + K1^H - K_isdf (Hermitian density matrices), where K_isdf is
    get_k_kpts with the W^q of ``theta`` and (in periodic parts u = exp(-i k.r) phi, q = k2 - k1)
        K1^{k1}_{pq} = w/nk sum_{k2} sum_P conj(u1_p(r_P)) sum_g V^q_P(g) [sum_ls u2_l(r_P) D^{k2}_ls conj(u2_s(g))] u1_q(g),
    V^q_P = conv_q(Theta_P): the fitted pair density on one side of the reference's exchange integral
    (pyscf/pbc/df/fft_jk.py:250-292), the exact one on the other; the error becomes quadratic in the fit error."""
    kpts = np.reshape(kpts, (-1, 3))
    band = kpts if kpts_band is None else np.reshape(kpts_band, (-1, 3))
    ao_band = ao_kpts if ao_band is None else ao_band
    nk = len(kpts)
    G = theta.shape[1]
    w = abs(np.linalg.det(a)) / G
    u2s = [(np.asarray(ao) * np.exp(-1j * coords.dot(k))[:, None]).T for ao, k in zip(ao_kpts, kpts)]      # (nao, G)
    u1s = [(np.asarray(ao) * np.exp(-1j * coords.dot(k))[:, None]).T for ao, k in zip(ao_band, band)]
    qs, qindex = unique_q(kpts, kpts_band)
    r_ip = coords[ip]
    Ws = [build_Wq(theta, a, mesh, q, r_ip) for q in qs]
    aoP = [np.ascontiguousarray(np.asarray(ao)[ip]) for ao in ao_kpts]
    aoPb = [np.ascontiguousarray(np.asarray(ao)[ip]) for ao in ao_band]
    k_isdf = get_k_kpts(aoP, Ws, qindex, dms, aoP_band=aoPb)
    out = []
    for i1, u1 in enumerate(u1s):
        k1 = 0
        for i2, u2 in enumerate(u2s):
            V = coulomb_Vq(theta, a, mesh, kpts[i2] - band[i1])                     # (P, G) complex
            F = u2[:, ip].T.dot(dms[i2]).dot(u2.conj())                             # (P, G)
            k1 = k1 + u1[:, ip].conj().dot((V * F).dot(u1.T))
        k1 = k1 * (w / nk)
        out.append(k1 + k1.conj().T - k_isdf[i1])
    return np.array(out)
