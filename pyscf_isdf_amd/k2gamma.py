"""Supercell <-> k-mesh correspondence (the role of pyscf/pbc/tools/k2gamma.py:39-97): a Monkhorst-Pack mesh of Nk k-points on
a cell describes the same crystal as the Gamma point of the Nk-fold supercell.  With R the translations that generate the
supercell (in the order gto.super_cell places the images: cartesian product over the axes) a Bloch AO is

    phi^k_mu(r) = sum_R exp(i k.R) phi^sc_{R mu}(r),

so k-blocked AO matrices O^k (overlap, Fock, J, K, density matrices alike) map to the supercell AO basis (R mu) as

    O^sc_{R mu, S nu} = 1/Nk sum_k exp(i k.(R - S)) O^k_{mu nu}        (to_supercell_ao_integrals),

and traces obey Tr(D^sc O^sc) = sum_k Tr(D^k O^k) = Nk x the value per cell.  Used to cross-check the k-point path against the
Gamma-point path (tests: supercell <-> k-mesh, as pyscf/pbc/scf/test/test_khf.py:73 does for the SCF energy)."""
import numpy as np
from . import gto


def kpts_to_kmesh(cell, kpts, tol=1e-6):
    """Number of distinct fractional k components per reciprocal axis (k2gamma.kpts_to_kmesh)."""
    frac = np.reshape(kpts, (-1, 3)).dot(np.asarray(cell.lattice_vectors()).T) / (2 * np.pi)
    frac = frac - np.floor(frac + tol)
    return [len(np.unique(np.round(frac[:, i] / tol).astype(np.int64))) for i in range(3)]


def translation_vectors(cell, kmesh):
    """Translations R (Nk, 3) of the cell that tile the kmesh supercell, in gto.super_cell's image order."""
    n = gto.cartesian_prod([np.arange(int(m)) for m in kmesh])
    return n.dot(np.asarray(cell.lattice_vectors(), dtype=float))


def get_phase(cell, kpts, kmesh=None, mesh=None):
    """(supercell, phase): phase[R, k] = exp(i k.R) / sqrt(Nk)  (k2gamma.get_phase).  mesh: FFT mesh of the supercell
    (default kmesh x cell.mesh, which makes the two grids identical point sets)."""
    kpts = np.reshape(kpts, (-1, 3))
    if kmesh is None:
        kmesh = kpts_to_kmesh(cell, kpts)
    R = translation_vectors(cell, kmesh)
    if len(R) != len(kpts):
        raise ValueError('k-points do not form the %s mesh' % (kmesh,))
    phase = np.exp(1j * R.dot(kpts.T)) / np.sqrt(len(R))
    scell = gto.super_cell(cell, kmesh, mesh=np.asarray(kmesh) * np.asarray(cell.mesh) if mesh is None else mesh)
    return scell, phase


def to_supercell_ao_integrals(cell, kpts, ao_ints, kmesh=None):
    """(Nk, nao, nao) k-blocked AO matrices -> the (Nk nao, Nk nao) supercell matrix (k2gamma.to_supercell_ao_integrals)."""
    _, phase = get_phase(cell, kpts, kmesh)
    nR, nk = phase.shape
    nao = np.asarray(ao_ints).shape[-1]
    out = np.einsum('Rk,kij,Sk->RiSj', phase, np.asarray(ao_ints), phase.conj())
    return out.reshape(nR * nao, nR * nao)


def to_kpts_ao_integrals(cell, kpts, ao_sc, kmesh=None):
    """The inverse map: O^k_{mu nu} = sum_{R S} exp(-i k.(R - S)) O^sc_{R mu, S nu} / Nk (translation-invariant O^sc)."""
    _, phase = get_phase(cell, kpts, kmesh)
    nR, nk = phase.shape
    nao = np.asarray(ao_sc).shape[0] // nR
    m = np.asarray(ao_sc).reshape(nR, nao, nR, nao)
    return np.einsum('Rk,RiSj,Sk->kij', phase.conj(), m, phase)
