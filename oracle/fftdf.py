"""Oracle: the reference's exact FFTDF J/K/ERI (the c -> infinity limit of ISDF).  TEST INFRASTRUCTURE ONLY.

Follows pyscf/pbc/df/fft_jk.py:33-109 (get_j_kpts), :177-302 (get_k_kpts incl. the MO-tagged DM
shortcut :235-238,256-259, phases :271-274,288 and weight :217), pyscf/pbc/df/df_jk.py:1411-1444
(shape formatting) and pyscf/pbc/df/fft_ao2mo.py:154-184 (ERI normalisation, s4 packing).
``ao*`` arguments are (G, nao) arrays in the reference's layout.
"""
import numpy as np
from . import pbc_tools as tools


def get_j(aoR, dm, a, mesh):
    """Γ-point J for real dm (nao,nao) or (nset,nao,nao)."""
    dms = np.asarray(dm, dtype=float)
    shape = dms.shape
    dms = dms.reshape(-1, shape[-2], shape[-1])
    G = aoR.shape[0]
    vol = abs(np.linalg.det(a))
    coulG = tools.get_coulG(a, mesh)
    vj = np.empty_like(dms)
    for i, d in enumerate(dms):
        rho = np.einsum('gi,ij,gj->g', aoR, d, aoR, optimize=True)
        vR = tools.ifft(coulG * tools.fft(rho, mesh), mesh).real * (vol / G)
        vj[i] = aoR.T.dot(vR[:, None] * aoR)
    return vj.reshape(shape)


def get_k(aoR, dm, a, mesh, mo_coeff=None, mo_occ=None, blksize=8):
    """Γ-point K for real dm; uses the occupied-orbital shortcut when mo_coeff/mo_occ are given."""
    dms = np.asarray(dm, dtype=float)
    shape = dms.shape
    dms = dms.reshape(-1, shape[-2], shape[-1])
    nset, nao = dms.shape[:2]
    G = aoR.shape[0]
    vol = abs(np.linalg.det(a))
    weight = vol / G
    coulG = tools.get_coulG(a, mesh)
    ao1T = np.ascontiguousarray(aoR.T)
    ao2T = ao1T
    if mo_coeff is not None and nset == 1:
        occ = np.asarray(mo_occ)
        c = np.asarray(mo_coeff)[:, occ > 0] * np.sqrt(occ[occ > 0])
        ao2T = c.T.dot(ao1T)
        ao_dms = [ao2T]
    else:
        ao_dms = [d.dot(ao2T) for d in dms]
    naoj = ao2T.shape[0]
    vR_dm = np.empty((nset, nao, G))
    for p0 in range(0, nao, blksize):
        p1 = min(nao, p0 + blksize)
        rho1 = np.einsum('ig,jg->ijg', ao1T[p0:p1], ao2T)
        vG = tools.fft(rho1.reshape(-1, G), mesh)
        vG *= coulG
        vR = tools.ifft(vG, mesh).real.reshape(p1 - p0, naoj, G)
        for i in range(nset):
            vR_dm[i, p0:p1] = np.einsum('ijg,jg->ig', vR, ao_dms[i])
    vk = np.empty_like(dms)
    for i in range(nset):
        vk[i] = weight * vR_dm[i].dot(ao1T.T)
    return vk.reshape(shape)


def get_jk_kpts(ao_kpts, dm_kpts, a, mesh, coords, kpts, ao_band=None, kpts_band=None,
                mo_coeff=None, mo_occ=None):
    """k-point J and K (complex), dm_kpts (nk,nao,nao); returns (vj, vk) each (nband,nao,nao)."""
    kpts = np.asarray(kpts).reshape(-1, 3)
    dms = np.asarray(dm_kpts)
    nk, nao = dms.shape[0], dms.shape[-1]
    G = ao_kpts[0].shape[0]
    vol = abs(np.linalg.det(a))
    if ao_band is None:
        ao_band, kpts_band = ao_kpts, kpts
    kpts_band = np.asarray(kpts_band).reshape(-1, 3)
    nband = len(kpts_band)
    # J
    coulG0 = tools.get_coulG(a, mesh)
    rho = np.zeros(G, dtype=np.complex128)
    for k in range(nk):
        rho += np.einsum('gi,ij,gj->g', ao_kpts[k], dms[k], ao_kpts[k].conj(), optimize=True)
    rho *= 1. / nk
    vR = tools.ifft(coulG0 * tools.fft(rho, mesh), mesh) * (vol / G)
    vj = np.array([ao.conj().T.dot(vR[:, None] * ao) for ao in ao_band])
    # K
    weight = 1. / nk * (vol / G)
    vk = np.zeros((nband, nao, nao), dtype=np.complex128)
    ao2_kpts = [np.ascontiguousarray(ao.T) for ao in ao_kpts]
    ao1_kpts = [np.ascontiguousarray(ao.T) for ao in ao_band]
    if mo_coeff is not None:
        mo = [np.asarray(mo_coeff[k])[:, np.asarray(mo_occ[k]) > 0] * np.sqrt(np.asarray(mo_occ[k])[np.asarray(mo_occ[k]) > 0])
              for k in range(nk)]
        ao2_kpts = [mo[k].T.dot(ao) for k, ao in enumerate(ao2_kpts)]
    for k2, ao2T in enumerate(ao2_kpts):
        if ao2T.size == 0:
            continue
        ao_dm = ao2T.conj() if mo_coeff is not None else dms[k2].dot(ao2T.conj())
        for k1, ao1T in enumerate(ao1_kpts):
            q = kpts[k2] - kpts_band[k1]
            coulG = tools.get_coulG(a, mesh, q)
            expmikr = np.exp(-1j * np.dot(coords, q)) if abs(q).sum() > 1e-9 else np.array(1.)
            vR_dm = np.empty((nao, G), dtype=np.complex128)
            for p in range(nao):
                rho1 = (ao1T[p].conj() * expmikr)[None, :] * ao2T
                v = tools.ifft(tools.fft(rho1, mesh) * coulG, mesh)
                vR_dm[p] = np.einsum('jg,jg->g', v, ao_dm)
            vR_dm *= expmikr.conj()
            vk[k1] += weight * vR_dm.dot(ao1T.T)
    return vj, vk


def get_ao_eri_s4(aoR, a, mesh):
    """Γ-point compact (s4) ERI matrix (npair,npair), fft_ao2mo.py:45-99,154-184."""
    G, nao = aoR.shape
    vol = abs(np.linalg.det(a))
    coulG = tools.get_coulG(a, mesh)
    i, j = np.tril_indices(nao)
    pairs = (aoR[:, i] * aoR[:, j]).T                    # (npair, G)
    vR = tools.ifft(tools.fft(pairs, mesh) * (coulG * (vol / G)), mesh).real
    return vR.dot(pairs.T)
