"""``ISDF``: a drop-in for ``pyscf.pbc.df.FFTDF`` whose K build runs through interpolative separable
density fitting on MI355X.

Public surface = FFTDF's (pyscf/pbc/df/fft.py:155-359): ``__init__(cell, kpts)``, attributes
``cell, kpts, mesh, grids, stdout, verbose, max_memory, exxdiv, blockdim``, and ``build()``,
``reset()``, ``dump_flags()``, ``check_sanity()``, ``get_jk(dm, hermi, kpts, kpts_band, with_j,
with_k, omega, exxdiv)``, ``get_naoaux()``, ``get_ao_eri()/get_eri()``, ``update_mf()``; SCF callers
(pyscf/pbc/scf/hf.py:649-698) only ever use these.  Array conventions of get_jk follow
pyscf/pbc/df/df_jk.py:1411-1444: the result has the shape of ``dm``; Γ point + real dm -> float64.

What runs where: this file is host orchestration only (which stage, which buffers, which rank);
every stage executes in libmi355_isdf.so via ``backend.HipBackend``.  There is no CPU path.
"""
import sys
import time
import numpy as np
import torch
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


class ISDF:
    _keys = {'cell', 'kpts', 'grids', 'mesh', 'blockdim', 'exxdiv', 'c_isdf', 'select', 'tie_rtol'}

    def __init__(self, cell, kpts=np.zeros((1, 3)), c_isdf=10, select='local', backend=None, comm=None):
        self.cell = cell
        self.stdout = getattr(cell, 'stdout', None) or sys.stdout
        self.verbose = getattr(cell, 'verbose', 0)
        self.max_memory = getattr(cell, 'max_memory', 4000)
        self.kpts = np.asarray(kpts).reshape(-1, 3)
        self.grids = UniformGrids(cell, cell.mesh)
        self.blockdim = 240
        self.exxdiv = None
        self.c_isdf = c_isdf
        self.select = select              # 'local': per-atom Voronoi blocks + global fit; 'global': one block
        self.tie_rtol = 1e-10
        self.reg_rel = 1e-12             # relative diagonal shift of A_PP in the global fit
        self.reg_used = 0.0
        self.fft_batch = None             # rows per FFT batch (None: sized from free memory)
        self._backend = backend
        self._comm = comm
        self._rsh_df = {}
        self._built = False
        self.timings = {}
        # device state
        self.ao = None        # (nao, G)
        self.aoP = None       # (P, nao)
        self.W = None         # (P, P)
        self.ip = None        # np.int64[P] grid indices of the interpolation points

    # ---- FFTDF-compatible plumbing ---------------------------------------------------------------
    @property
    def mesh(self):
        return self.grids.mesh

    @mesh.setter
    def mesh(self, mesh):
        self.grids.mesh = np.asarray(mesh)
        self.grids._coords = None

    @property
    def backend(self):
        if self._backend is None:
            from .backend import HipBackend
            dev = 0
            if self._comm is not None:
                dev = self._comm.local_rank
            self._backend = HipBackend(dev)
        return self._backend

    def reset(self, cell=None):
        if cell is not None:
            self.cell = cell
        self.grids = UniformGrids(self.cell, self.cell.mesh)
        self.ao = self.aoP = self.W = self.ip = None
        self._rsh_df = {}
        self._built = False
        return self

    def dump_flags(self, verbose=None):
        out = self.stdout
        out.write('\n******** %s ********\n' % self.__class__)
        out.write('mesh = %s (%d PWs)\n' % (self.mesh, np.prod(self.mesh)))
        out.write('c_isdf = %s  select = %s  tie_rtol = %g\n' % (self.c_isdf, self.select, self.tie_rtol))
        out.write('len(kpts) = %d\n' % len(self.kpts))
        return self

    def check_sanity(self):
        if getattr(self.cell, 'dimension', 3) != 3:
            raise RuntimeError('ISDF is implemented for 3-D periodic cells only')
        if not self._is_gamma(self.kpts):
            raise NotImplementedError('ISDF on MI355X: only the Gamma point is implemented in this round')
        return self

    @staticmethod
    def _is_gamma(kpts):
        return kpts is None or abs(np.asarray(kpts)).sum() < 1e-9

    def get_naoaux(self):
        return 0 if self.ip is None else len(self.ip)

    def update_mf(self, mf):
        mf = mf.copy() if hasattr(mf, 'copy') else mf
        mf.with_df = self
        return mf

    # ---- build ---------------------------------------------------------------------------------
    def _tick(self, name, t0):
        self.backend.synchronize()
        t1 = time.perf_counter()
        self.timings[name] = self.timings.get(name, 0.0) + (t1 - t0)
        return t1

    def nip_per_atom(self):
        aosl = _aoslice_by_atom(self.cell)
        return (np.asarray(aosl[:, 1] - aosl[:, 0]) * self.c_isdf).astype(np.int32)

    def build(self):
        self.check_sanity()
        cell, be = self.cell, self.backend
        self.timings = {}
        self.ao = self.aoP = self.W = None
        be.empty_cache()                 # the (P, G) fit buffer needs one contiguous segment
        t0 = time.perf_counter()
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        nao = cell.nao_nr()
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        coords = self.grids.coords
        rcut = gto.estimate_rcut_per_shell(cell)
        Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
        self._ao_args = (np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut)
        t0 = self._tick('host_setup', t0)

        # S1 collocation
        coords_soa = be.to_device(np.ascontiguousarray(coords.T))
        self.ao = be.empty((nao, G))
        be.eval_ao(*self._ao_args, coords_soa, self.ao)
        del coords_soa
        t0 = self._tick('S1_eval_ao', t0)

        # S2 + S3 selection and fit
        if self.select == 'global':
            P = int(min(self.c_isdf * nao, G))
            theta = be.empty((P, G))
            piv = be.empty((1, P), dtype=torch.int64)
            rank = be.select_ip(self.ao, [0, G], [P], -1.0, self.tie_rtol, theta, piv)
            P = int(rank[0])
            t0 = self._tick('S2_select_ip', t0)
            theta = theta[:P]
            piv = piv[0, :P].contiguous()
            be.fit_from_chol(theta, P, G, piv)
            self.ip = be.to_host(piv).astype(np.int64)
            self.aoP = be.empty((P, nao))
            tmp = be.empty((nao, P))
            be.gather_cols(self.ao, piv, tmp)
            self.aoP.copy_(tmp.T)
            del tmp
            t0 = self._tick('S3_fit', t0)
        elif self.select == 'local':
            owner = partition_grid_by_atom(coords, cell.atom_coords(), a)
            perm = np.argsort(owner, kind='stable').astype(np.int64)
            counts = np.bincount(owner, minlength=cell.natm)
            blk_off = np.append(0, np.cumsum(counts)).astype(np.int64)
            nip = np.minimum(self.nip_per_atom(), counts).astype(np.int32)
            kmax = int(nip.max())
            t0 = self._tick('host_partition', t0)
            d_perm = be.to_device(perm)
            ao_sel = be.empty((nao, G))
            be.gather_cols(self.ao, d_perm, ao_sel)
            L = be.empty((kmax, G))
            piv = be.empty((cell.natm, kmax), dtype=torch.int64)
            rank = be.select_ip(ao_sel, blk_off, nip, -1.0, self.tie_rtol, L, piv)
            del ao_sel, L
            be.empty_cache()
            piv_h = be.to_host(piv)
            ip = np.concatenate([perm[blk_off[b] + piv_h[b, :rank[b]]] for b in range(cell.natm)])
            self.ip = ip.astype(np.int64)
            P = len(ip)
            t0 = self._tick('S2_select_ip', t0)
            theta = be.empty((P, G))
            self.aoP = be.empty((P, nao))
            self.reg_used = be.fit_global(self.ao, G, be.to_device(self.ip), self.reg_rel, theta, self.aoP)
            t0 = self._tick('S3_fit', t0)
        else:
            raise ValueError("select must be 'local' or 'global'")

        # S4 + S5 Coulomb convolution and W
        self.W = be.empty((P, P))
        batch = self.fft_batch or _default_fft_batch(G, P)
        be.coulomb_W(theta, mesh, a, 0, P, batch, self.W, upper_only=True)
        be.symmetrize_upper(self.W)
        del theta
        be.empty_cache()
        t0 = self._tick('S4S5_coulomb_W', t0)
        self._built = True
        return self

    # ---- J / K ---------------------------------------------------------------------------------
    def get_jk(self, dm, hermi=1, kpts=None, kpts_band=None, with_j=True, with_k=True, omega=None,
               exxdiv=None):
        if omega is not None:
            raise NotImplementedError('range-separated Coulomb kernel (omega) is not implemented for ISDF')
        if kpts is None:
            kpts = self.kpts if self._is_gamma(self.kpts) else self.kpts
        if not self._is_gamma(kpts) or not self._is_gamma(kpts_band):
            raise NotImplementedError('ISDF on MI355X: only the Gamma point is implemented in this round')
        if exxdiv is not None and exxdiv != 'None':
            raise NotImplementedError("exxdiv=%r: only exxdiv=None is implemented (SURVEY 7.3-8)" % (exxdiv,))
        if not self._built:
            self.build()
        be = self.backend
        dm_in = np.asarray(dm)
        if np.iscomplexobj(dm_in):
            if abs(dm_in.imag).max() > 1e-12:
                raise NotImplementedError('complex density matrices at the Gamma point are not supported')
            dm_in = dm_in.real
        nao = self.cell.nao_nr()
        dms = np.ascontiguousarray(dm_in.reshape(-1, nao, nao), dtype=np.float64)
        nset = dms.shape[0]
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        a = np.asarray(self.cell.lattice_vectors(), dtype=float)
        d_dm = be.to_device(dms)
        vj = vk = None
        t0 = time.perf_counter()
        if with_j:
            d_vj = be.empty((nset, nao, nao))
            be.get_j(self.ao, G, mesh, a, d_dm, d_vj)
            t0 = self._tick('S6_get_j', t0)
            vj = be.to_host(d_vj).reshape(dm_in.shape)
        if with_k:
            d_vk = be.empty((nset, nao, nao))
            P = self.W.shape[0]
            be.get_k(self.aoP, self.W, 0, P, d_dm, d_vk)
            t0 = self._tick('S7_get_k', t0)
            vk = be.to_host(d_vk).reshape(dm_in.shape)
        return vj, vk

    # ---- ERIs from the factorisation (small systems; reached from SCF.get_jk's incore branch,
    #      pyscf/pbc/scf/hf.py:670-679) ------------------------------------------------------------
    def get_ao_eri(self, kpts=None, compact=True):
        if not self._is_gamma(kpts):
            raise NotImplementedError
        if not self._built:
            self.build()
        aoP = self.backend.to_host(self.aoP)
        W = self.backend.to_host(self.W)
        nao = aoP.shape[1]
        if compact:
            i, j = np.tril_indices(nao)
            X = aoP[:, i] * aoP[:, j]
        else:
            X = np.einsum('pi,pj->pij', aoP, aoP).reshape(len(aoP), -1)
        return X.T.dot(W).dot(X)

    get_eri = get_ao_eri

    def get_pp(self, kpts=None):
        raise NotImplementedError('hcore terms are outside the ISDF hot path; use FFTDF.get_pp')

    def get_nuc(self, kpts=None):
        raise NotImplementedError('hcore terms are outside the ISDF hot path; use FFTDF.get_nuc')


def _aoslice_by_atom(cell):
    if hasattr(cell, 'aoslice_by_atom'):
        s = np.asarray(cell.aoslice_by_atom())
        return s[:, -2:] if s.shape[1] == 4 else s
    raise AttributeError('cell lacks aoslice_by_atom')


def _default_fft_batch(G, P):
    """Rows per FFT batch: ~6 GiB for the real batch + ~6 GiB for its half spectrum."""
    nb = int((6 << 30) // (8 * G))
    return max(1, min(P, nb, 1024))
