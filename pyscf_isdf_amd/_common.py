"""Small host-side helpers shared by the ISDF orchestration modules."""
import numpy as np
from . import gto


class UniformGrids:
    """The few attributes of pyscf.pbc.dft.gen_grid.UniformGrids (gen_grid.py:63-137) callers read."""

    def __init__(self, cell, mesh):
        self.cell = cell
        self.mesh = np.asarray(mesh)
        self._coords = None
        self.non0tab = None

    @property
    def coords(self):
        if self._coords is None:
            self._coords = self.cell.get_uniform_grids(self.mesh)
        return self._coords

    @property
    def weights(self):
        ngrids = int(np.prod(self.mesh))
        return np.full(ngrids, self.cell.vol / ngrids)


def partition_grid_by_atom(coords, atom_coords, a, tie_atol=1e-9):
    """owner[g] = index of the nearest atom (minimum image); ties within ``tie_atol`` go to the
    lowest atom index.  KD-tree over the 27 nearest images of every atom."""
    from scipy.spatial import cKDTree
    Ts = gto.cartesian_prod([[-1, 0, 1]] * 3).astype(float).dot(a)
    natm = len(atom_coords)
    pts = (atom_coords[None, :, :] + Ts[:, None, :]).reshape(-1, 3)
    ids = np.tile(np.arange(natm), len(Ts))
    k = min(8, len(pts))
    dist, idx = cKDTree(pts).query(coords, k=k, workers=-1)
    cand = ids[idx]                                        # (G, k) atom ids by increasing distance
    tied = dist <= dist[:, :1] + tie_atol
    cand = np.where(tied, cand, natm)
    return cand.min(axis=1).astype(np.int32)


def _monkhorst_pack_size(cell, kpts, tol=1e-5):
    """Number of distinct k-point fractions per reciprocal axis (pyscf/pbc/tools/pbc.py:get_monkhorst_pack_size)."""
    skpts = np.linalg.solve(cell.reciprocal_vectors().T, np.reshape(kpts, (-1, 3)).T).T.round(decimals=6)
    return tuple(len(np.unique(np.round(skpts[:, i] / tol).astype(int))) for i in range(3))


def _aoslice_by_atom(cell):
    if hasattr(cell, 'aoslice_by_atom'):
        s = np.asarray(cell.aoslice_by_atom())
        return s[:, -2:] if s.shape[1] == 4 else s
    raise AttributeError('cell lacks aoslice_by_atom')


def _default_fft_batch(G, P, free_bytes=None):
    """Rows per FFT batch: up to ~7 GiB for the real batch (+ as much for its half spectrum), a
    multiple of the GEMM's 128-row tile so that no MFMA work is wasted on padding.  With free_bytes
    the batch also has to fit what is left: 8 G per row for V, ~8 G for the half spectrum and as much
    again for the FFT's work area, 2 GiB kept back for the GEMM's slab buffers."""
    nb = int((7 << 30) // (8 * G))
    if free_bytes is not None:
        nb = min(nb, int(max(0, free_bytes - (2 << 30)) // (24 * G)))
    nb = min(P, nb, 1024)
    if nb >= 128:
        nb -= nb % 128
    return max(1, nb)


class TaggedArray(np.ndarray):
    """ndarray with attributes (the role of pyscf.lib.tag_array: veff with ecoul / exc / vj / vk, density matrices with
    mo_coeff / mo_occ)."""

    def __new__(cls, a, **tags):
        obj = np.asarray(a).view(cls)
        obj.__dict__.update(tags)
        return obj

    def __array_finalize__(self, obj):
        if obj is not None and hasattr(obj, '__dict__'):
            self.__dict__.update(getattr(obj, '__dict__', {}))


def tag_array(a, **tags):
    return TaggedArray(a, **tags)
