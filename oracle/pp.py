"""Oracle: GTH pseudopotential AO matrix (local + non-local), the reference's FFTDF.get_pp.
TEST INFRASTRUCTURE ONLY.

Follows pyscf/pbc/df/fft.py:64-152 (get_pp), pyscf/pbc/gto/pseudo/pp.py:58-93 (local part in G space),
pyscf/pbc/gto/pseudo/pp_int.py:51-71 (first term and the G=0 value), pp.py:150-190 (_qli projector
polynomials) and pyscf/pbc/gto/cell.py:613-643 (structure factor).  The analytic Fourier transform of
the AOs (ft_ao, libcint) is restated from the closed form for real solid-harmonic Gaussians:
    FT[ S_lm(r - R) exp(-a |r-R|^2) ](q) = (-i)^l (pi/a)^{3/2} (2a)^{-l} S_lm(q) exp(-q^2/4a) exp(-i q.R).
Pinned by pyscf/pbc/df/test/test_fft.py:601-611.
"""
import numpy as np
from . import ao as oao
from . import pbc_tools as tools

ATOM_OF, ANG_OF, NPRIM_OF, NCTR_OF, PTR_EXP, PTR_COEFF, BAS_SLOTS = 0, 1, 2, 3, 5, 6, 8
PTR_COORD, ATM_SLOTS = 1, 6


def qli(x, l, i):
    s = np.sqrt
    tab = {(0, 0): lambda x: 4 * s(2.) + 0 * x, (0, 1): lambda x: 8 * s(2 / 15.) * (3 - x ** 2),
           (0, 2): lambda x: 16 / 3. * s(2 / 105.) * (15 - 10 * x ** 2 + x ** 4),
           (1, 0): lambda x: 8 * s(1 / 3.) + 0 * x, (1, 1): lambda x: 16 * s(1 / 105.) * (5 - x ** 2),
           (1, 2): lambda x: 32 / 3. * s(1 / 1155.) * (35 - 14 * x ** 2 + x ** 4),
           (2, 0): lambda x: 8 * s(2 / 15.) + 0 * x, (2, 1): lambda x: 16 / 3. * s(2 / 105.) * (7 - x ** 2),
           (2, 2): lambda x: 32 / 3. * s(2 / 15015.) * (63 - 18 * x ** 2 + x ** 4)}
    return tab[(l, i)](x)


def solid_harmonics(l, q):
    """libcint real solid harmonics (same convention as oracle/ao.py), list of arrays over points q (n,3)."""
    return oao._angular(l, q[:, 0], q[:, 1], q[:, 2]) if l >= 2 else \
        ([oao.FAC_S * np.ones(len(q))] if l == 0 else [oao.FAC_P * q[:, 0], oao.FAC_P * q[:, 1], oao.FAC_P * q[:, 2]])


def ft_ao(atm, bas, env, q):
    """Analytic FT of the (non-periodic) AOs at wave vectors q (n,3): (n, nao) complex."""
    atm = np.asarray(atm).reshape(-1, ATM_SLOTS)
    bas = np.asarray(bas).reshape(-1, BAS_SLOTS)
    loc = oao.ao_loc(bas)
    out = np.zeros((len(q), loc[-1]), dtype=complex)
    q2 = np.einsum('gx,gx->g', q, q)
    for ib in range(len(bas)):
        l, npr, nc = bas[ib, ANG_OF], bas[ib, NPRIM_OF], bas[ib, NCTR_OF]
        es = env[bas[ib, PTR_EXP]:bas[ib, PTR_EXP] + npr]
        cs = env[bas[ib, PTR_COEFF]:bas[ib, PTR_COEFF] + npr * nc].reshape(nc, npr)
        R = env[atm[bas[ib, ATOM_OF], PTR_COORD]:atm[bas[ib, ATOM_OF], PTR_COORD] + 3]
        rad = np.array([(np.pi / e) ** 1.5 * (2 * e) ** (-l) * np.exp(-q2 / (4 * e)) for e in es])   # (nprim, n)
        rad = cs.dot(rad)                                                                           # (nctr, n)
        ph = (-1j) ** l * np.exp(-1j * q.dot(R))
        S = solid_harmonics(l, q)
        deg = 2 * l + 1
        for k in range(nc):
            for m in range(deg):
                out[:, loc[ib] + k * deg + m] = rad[k] * S[m] * ph
    return out


def get_vlocG(charges, pseudo_of_atom, a, mesh, Gv):
    """(natm, G): local pseudopotential kernel, sign as in the reference (positive)."""
    coulG = tools.get_coulG(a, mesh)
    G2 = np.einsum('gx,gx->g', Gv, Gv)
    out = np.zeros((len(charges), len(G2)))
    for ia, Z in enumerate(charges):
        out[ia] = Z * coulG
        pp = pseudo_of_atom[ia]
        if pp is None:
            continue
        rloc, nexp, cexp = pp[1], pp[2], pp[3]
        out[ia] *= np.exp(-0.5 * rloc ** 2 * G2)
        out[ia, G2 == 0] = -2 * np.pi * Z * rloc ** 2
        x = G2 * rloc ** 2
        polys = [1, 3 - x, 15 - 10 * x + x ** 2, 105 - 105 * x + 21 * x ** 2 - x ** 3]
        cf = sum(cexp[i] * polys[i] for i in range(nexp))
        out[ia] -= (2 * np.pi) ** 1.5 * rloc ** 3 * np.exp(-0.5 * x) * cf
    return out


def get_pp(atm, bas, env, atom_coords, charges, pseudo_of_atom, a, mesh, coords, ao_kpts, kpts):
    """GTH pseudopotential matrices, one per k-point ((nao,nao) complex; real at Gamma)."""
    kpts = np.reshape(kpts, (-1, 3))
    a = np.asarray(a, dtype=float)
    vol = abs(np.linalg.det(a))
    b = 2 * np.pi * np.linalg.inv(a.T)
    Gv = tools.get_Gv(b, mesh)
    SI = np.exp(-1j * np.dot(atom_coords, Gv.T))
    vlocG = -np.einsum('ij,ij->j', SI, get_vlocG(charges, pseudo_of_atom, a, mesh, Gv))
    vlocR = tools.ifft(vlocG, mesh).real
    out = []
    for k, kpt in enumerate(kpts):
        ao = ao_kpts[k]
        vpp = ao.conj().T.dot(vlocR[:, None] * ao)
        Gk = Gv + kpt
        Gr = np.linalg.norm(Gk, axis=1)
        aokG = ft_ao(atm, bas, env, Gk) * (1. / vol) ** .5
        vnl = 0
        for ia, pp in enumerate(pseudo_of_atom):
            if pp is None:
                continue
            for l, (rl, nl, hl) in enumerate(pp[5:]):
                if nl == 0:
                    continue
                # projector "GTO" with exponent rl^2/2 and coefficient rl^(l+1.5) pi^1.25 evaluated at Gk
                S = np.array(solid_harmonics(l, Gk))                                     # (2l+1, G)
                base = rl ** (l + 1.5) * np.pi ** 1.25 * np.exp(-.5 * rl ** 2 * Gr ** 2) * S
                pY = np.array([base * qli(Gr * rl, l, i) for i in range(nl)])            # (nl, 2l+1, G)
                SPG = np.einsum('g,nmg->nmg', SI[ia].conj(), pY)
                SPG_ao = np.einsum('nmg,gp->nmp', SPG, aokG)
                tmp = np.einsum('ij,jmp->imp', np.asarray(hl), SPG_ao)
                vnl = vnl + np.einsum('imp,imq->pq', SPG_ao.conj(), tmp)
        vnl = vnl * (1. / vol)
        if abs(kpt).sum() < 1e-9:
            out.append(vpp.real + np.real(vnl))
        else:
            out.append(vpp + vnl)
    return out
