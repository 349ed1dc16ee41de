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
import os
import sys
import time
import warnings
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
        self.select_tol = -1.0           # stop when the largest residual diagonal <= tol; < 0: m*eps*max diag (scipy_helper.py:88-90)
        self.reg_rel = 1e-12             # relative diagonal shift of A_PP in the global fit
        self.reg_used = 0.0
        self.k_ip_factor = None           # k-points: points = c_isdf * nao * k_ip_factor (default min(nk, 2))
        self.kpts_band = None             # band k-points of the last k-point build (set by get_jk(kpts_band=...))
        self.force_sharded = False       # run the multi-GPU code path even on one rank (tests)
        self.fit_route = 'auto'          # 'cholesky': forward solve over the grid (S3b), always safe;
                                         # 'blockjacobi': no solve over the grid (S3c), amplifies rounding by cond(A');
                                         # 'auto': S3c, verified with probe densities, S3b when the check fails
        self.bj_check_tol = 3e-8         # 'auto': largest accepted relative mismatch of the probe energies (tracks
                                         # max|dK|/|K| of the two routes within a factor of a few, profiles/r01_bj_*)
        self.bj_auto_kpts = False        # k-points: 'auto' = Cholesky route unless this is set (the fit is < 10 % of a k-point
                                         # build, and a failed check costs a second pass over all q: MgO 2x2x2 reads 7e-8)
        self.bj_max_c = 12               # 'auto': do not even try S3c above this c_isdf (cond(A') grows ~100x per +5)
        self.bj_nprobe = 8
        self.bj_cluster_radius = 2.4     # Bohr; atoms closer than this share a preconditioner block (X-H bonds)
        self.bj_group = 1                # merge this many consecutive clusters into one block (experiments)
        self.bj_check = None             # the measured mismatch of the last 'auto' build
        self.fit_route_used = None
        self.block_shift = 0.0           # relative diagonal shift of the per-atom blocks in the S3c route (raised per
                                         # block when a block is not positive definite: D is only a preconditioner)
        self.block_shift_used = 0.0
        self.explicit_theta = False      # True: form Theta itself (second O(P^2 G) solve); same W in exact arithmetic
        self.fft_batch = None             # rows per FFT batch (None: sized from free memory)
        self._backend = backend
        self._comm = comm
        self._rsh_df = {}
        self._built = False
        self._bufs = {}
        self._ovlp = None
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
            if os.environ.get('ISDF_ONE_GPU'):           # rehearsal: several ranks share device 0
                dev = 0
            self._backend = HipBackend(dev)
        return self._backend

    def reset(self, cell=None):
        if cell is not None:
            self.cell = cell
        self.grids = UniformGrids(self.cell, self.cell.mesh)
        self.ao = self.aoP = self.W = self.ip = None
        self._bufs = {}
        self._ovlp = None
        if self._backend is not None:
            self._backend.empty_cache()
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
    def _buffer(self, name, shape, dtype=torch.float64):
        """Persistent device buffer, reused across builds: releasing and re-mapping the (P, G) fit
        buffer (214 GiB at 4x4x4) costs seconds per build, so the big buffers live until reset()."""
        n = int(np.prod(shape))
        buf = self._bufs.get(name)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            self._bufs.pop(name, None)
            buf = None
            self.backend.empty_cache()
            buf = self.backend.empty((n,), dtype=dtype)
            self._bufs[name] = buf
        return buf[:n].view(*shape)

    # ---- S3c helpers (block-Jacobi route) ---------------------------------------------------------------
    def _bj_prepare(self, ao, nh, d_ip, ip_off, aoP, scratch=None):
        """aoP <- ao[:, ip]^T;  returns (Aprime_factor, Dblk): the per-atom block factors D and the Cholesky
        factor of A' = D^-1 A D^-T (+ reg_rel)."""
        be = self.backend
        P = aoP.shape[0]
        be.gather_aoP(ao, d_ip, aoP)
        A = self._buffer('factor', (P, P))
        be.gram_sq(aoP, A, nh)
        # the fit's regularisation goes onto A itself, before the block scaling: both routes then solve the same
        # (A + reg I) x = b and differ by rounding only; A' gets a further shift only if its factorisation fails
        be.shift_diag(A, self.reg_rel)
        Dblk = self._buffer('Dblk', (P, P))
        self.block_shift_used = be.block_chol(A, ip_off, self.block_shift, Dblk)
        be.block_solve(Dblk, ip_off, 0, 0, A)            # A' = D^-1 A D^-T
        be.block_solve(Dblk, ip_off, 1, 1, A)
        extra = be.chol_inplace(A, 0.0, scratch=scratch)
        self.reg_used = self.reg_rel + extra
        return A, Dblk

    def _bj_rows(self, aoP, nh, ao, ng, Dblk, ip_off, out):
        """out (P, ng) <- Y' = D^-1 (aoP ao)^2 on ng grid columns."""
        be = self.backend
        be.pair_gram_rows(aoP, ao, ng, out, nh)
        be.block_solve(Dblk, ip_off, 0, 0, out)

    def _bj_finish(self, Afac, Dblk, ip_off, W, antisymmetric=False):
        """W <- D^-T [A'^-1 W A'^-1] D^-1 (W holds M' on entry)."""
        be = self.backend
        be.W_from_factor(Afac, 2, W)
        be.W_from_factor(Afac, 0, W)
        be.block_solve(Dblk, ip_off, 0, 1, W)
        be.block_solve(Dblk, ip_off, 1, 0, W)
        # the rounding noise along null(A) is not (anti)symmetric; the mean keeps it inside null(A) x null(A)
        be.symmetrize_mean(W, antisymmetric)

    def _bj_clusters(self):
        """Atoms grouped for the S3c preconditioner: single linkage (minimum image) below bj_cluster_radius Bohr.  The
        default joins X-H bonds only: a hydrogen's 50 points are nearly dependent on its neighbour's, so per-atom
        blocks leave A' = D^-1 A D^-T badly conditioned on molecular systems (64 H2O: probe mismatch 1e-6 with
        per-atom blocks), while diamond (C-C 2.9 Bohr) keeps one block per atom.  Returns a list of atom-index lists,
        ordered by their first atom; the interpolation points are stored cluster by cluster."""
        cell = self.cell
        natm = cell.natm
        parent = list(range(natm))

        def find(i):
            while parent[i] != i:
                parent[i] = parent[parent[i]]
                i = parent[i]
            return i
        r = float(self.bj_cluster_radius or 0.0)
        if r > 0 and natm > 1:
            a = np.asarray(cell.lattice_vectors(), dtype=float)
            frac = np.asarray(cell.atom_coords(), dtype=float).dot(np.linalg.inv(a))
            d = frac[:, None, :] - frac[None, :, :]
            d -= np.round(d)
            dist = np.linalg.norm(d.dot(a), axis=2)
            for i, j in zip(*np.nonzero(np.triu(dist < r, 1))):
                ri, rj = find(int(i)), find(int(j))
                if ri != rj:
                    parent[max(ri, rj)] = min(ri, rj)
        groups = {}
        for i in range(natm):
            groups.setdefault(find(i), []).append(i)
        return [groups[k] for k in sorted(groups)]

    def _bj_blocks(self, counts, clusters):
        """Offsets of the preconditioner blocks for points stored cluster by cluster (counts: points per atom);
        bj_group consecutive clusters are merged on top."""
        per = [int(sum(counts[b] for b in cl)) for cl in clusters]
        off = np.append(0, np.cumsum(per)).astype(np.int32)
        g = max(1, int(self.bj_group))
        if g > 1:
            off = np.unique(np.append(off[::g], off[-1])).astype(np.int32)
        return off

    def _bj_finish_sharded(self, Afac, Dblk, ip_off, W):
        """_bj_finish with the two-sided P x P solves split over the ranks by column blocks C_r (they are 4 P^3 flop,
        0.7 s at P = 16640, and would otherwise be replicated):  Z[:, C_r] = A'^-1 M'[:, C_r];  all_reduce;
        W'[:, C_r] = A'^-1 Z[C_r, :]^T (M' is symmetric) and the left block solve;  all_reduce;  the right block solve
        (block diagonal, cheap) and the symmetrisation replicated."""
        be, comm = self.backend, self.comm
        P = W.shape[0]
        c0, c1 = comm.split_range(P)
        X = W[:, c0:c1].clone()                      # (P, c) columns of M'
        W.zero_()
        if c1 > c0:
            be.factor_solve(Afac, X)
            W[:, c0:c1] = X
        comm.all_reduce_sum(W)                       # Z = A'^-1 M' on every rank
        if c1 > c0:
            X.copy_(W[c0:c1, :].T)                   # Z[C_r, :]^T = (M' A'^-1)[:, C_r]
        W.zero_()
        if c1 > c0:
            be.factor_solve(Afac, X)                 # A'^-1 M' A'^-1 [:, C_r]
            be.block_solve(Dblk, ip_off, 0, 1, X)    # D^-T (.)
            W[:, c0:c1] = X
        comm.all_reduce_sum(W)
        del X
        be.block_solve(Dblk, ip_off, 1, 0, W)        # (.) D^-1
        be.symmetrize_mean(W)

    def _fit_routes(self):
        if self.fit_route not in ('auto', 'blockjacobi', 'cholesky'):
            raise ValueError("fit_route must be 'auto', 'blockjacobi' or 'cholesky'")
        if self.explicit_theta or self.fit_route == 'cholesky':
            return ['cholesky']
        if self.fit_route == 'blockjacobi':
            return ['blockjacobi']
        if self.c_isdf > self.bj_max_c:
            return ['cholesky']
        return ['blockjacobi', 'cholesky']

    def _bj_probe_mismatch(self, aoT_P, Afac, Dblk, ip_off, Yp, ng, grid_slice, W=None):
        """A-posteriori check of the S3c route: for random symmetric R_j the density t_j = diag(phi_P R_j phi_P^T)
        at the points has the Coulomb energy  t^T W t  through the matrix and  w sum_g f conv(f), f = Theta^T t,
        through the fitted density itself (vector operations only: one pass over Y', nprobe FFTs).  The second form
        does not see the cond(A')-amplified rounding of M'; their largest relative difference is returned.
        aoT_P: (nao, P), or a list of such planes whose densities are added (k-points: Re/Im u^k at the points, the
        density sum_k u^k* R u^k with real symmetric R; W = the real plane of W^{q=0})."""
        be, comm = self.backend, self.comm
        cell = self.cell
        planes = aoT_P if isinstance(aoT_P, (list, tuple)) else [aoT_P]
        nao, P = planes[0].shape
        W = self.W if W is None else W
        n = int(self.bj_nprobe)
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        # the probe matrices R_j (random symmetric, fixed seed) are kept on the device: drawing n nao^2 normals
        # costs 0.2 s at nao = 1664
        cached = getattr(self, '_probe_R', None)
        if cached is None or cached[0] != (n, nao):
            rng = np.random.default_rng(20240203)
            R = rng.standard_normal((n, nao, nao))
            self._probe_R = ((n, nao), be.to_device(R + R.transpose(0, 2, 1)))
        d_R = self._probe_R[1]
        T = be.zeros((n, P))
        tmp = be.empty((n, P))
        for pl in planes:
            be.rho(pl, P, d_R, tmp)
            T += tmp
        del tmp
        T0 = T.clone()
        # matrix side: t^T W t
        TW = be.empty((n, P))
        be.gemm_nt(T0, W, TW)
        e_mat = np.einsum('jp,jp->j', be.to_host(TW), be.to_host(T0))
        # density side
        F = be.empty((n, ng))
        be.bj_probe_rows(T, Afac, Dblk, ip_off, Yp, ng, F)
        if grid_slice is None:
            CF = be.empty((n, G))
            be.coulomb_rows(F, mesh, a, n, out=CF)
            E = be.empty((n, n))
            be.gemm_nt(F, CF, E)
            e_fit = cell.vol / G * np.diag(be.to_host(E))
        else:
            # the fitted densities live on grid slices: zero-padded all_reduce, replicated FFT (as the sharded J)
            g0, g1 = grid_slice
            full = be.zeros((n, G))
            full[:, g0:g1] = F
            comm.all_reduce_sum(full)
            be.coulomb_rows(full, mesh, a, n)
            E = be.empty((n, n))
            be.gemm_nt(F, full[:, g0:g1].contiguous(), E)
            comm.all_reduce_sum(E)
            e_fit = cell.vol / G * np.diag(be.to_host(E))
        return float(abs(e_mat - e_fit).max() / abs(e_fit).max())

    def _tick(self, name, t0):
        self.backend.synchronize()
        t1 = time.perf_counter()
        self.timings[name] = self.timings.get(name, 0.0) + (t1 - t0)
        return t1

    def nip_per_atom(self):
        aosl = _aoslice_by_atom(self.cell)
        return (np.asarray(aosl[:, 1] - aosl[:, 0]) * self.c_isdf).astype(np.int32)

    @property
    def comm(self):
        if self._comm is None:
            from .parallel import Comm
            self._comm = Comm()
        return self._comm

    def build(self):
        self.check_sanity()
        if not self._is_gamma(self.kpts) or not self._is_gamma(self.kpts_band):
            return self._build_kpts()
        if self.comm.size > 1 or self.force_sharded:
            return self._build_sharded()
        cell, be = self.cell, self.backend
        self.timings = {}
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
        self.ao = self._buffer('ao', (nao, G))
        be.eval_ao(*self._ao_args, coords_soa, self.ao)
        del coords_soa
        t0 = self._tick('S1_eval_ao', t0)

        # S2 + S3 selection and fit
        if self.select == 'global':
            P = int(min(self.c_isdf * nao, G))
            theta = self._buffer('theta', (P, G))
            piv = be.empty((1, P), dtype=torch.int64)
            rank = be.select_ip(self.ao, [0, G], [P], self.select_tol, self.tie_rtol, theta, piv)
            P = int(rank[0])
            t0 = self._tick('S2_select_ip', t0)
            theta = theta[:P]
            piv = piv[0, :P].contiguous()
            if self.explicit_theta:
                be.fit_from_chol(theta, P, G, piv)
                factor = None
            else:
                # keep the Cholesky rows L and apply T^-1 to the small (P, P) matrix instead of the (P, G) one
                factor = (self._buffer('factor', (P, P)), 1)
                be.gather_T(theta, P, piv, factor[0])
            self.ip = be.to_host(piv).astype(np.int64)
            self.aoP = self._buffer('aoP', (P, nao))
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
            # the block-major copy of phi and the Cholesky rows are scratch that dies before the fit:
            # they live inside the (P, G) fit buffer, which is not in use yet
            Pmax = int(nip.sum())
            scratch = self._buffer('theta', (max(Pmax, nao + kmax), G))
            ao_sel = scratch[:nao]
            L = scratch[nao:nao + kmax]
            be.gather_cols(self.ao, d_perm, ao_sel)
            piv = be.empty((cell.natm, kmax), dtype=torch.int64)
            rank = be.select_ip(ao_sel, blk_off, nip, self.select_tol, self.tie_rtol, L, piv)
            del ao_sel, L, scratch
            piv_h = be.to_host(piv)
            clusters = self._bj_clusters()
            ip = np.concatenate([perm[blk_off[b] + piv_h[b, :rank[b]]] for cl in clusters for b in cl])
            self.ip = ip.astype(np.int64)
            P = len(ip)
            t0 = self._tick('S2_select_ip', t0)
            theta = self._buffer('theta', (max(Pmax, nao + kmax), G))[:P]
            self.aoP = self._buffer('aoP', (P, nao))
            d_ip = be.to_device(self.ip)
            self.W = self._buffer('W', (P, P))
            for route in self._fit_routes():
                if route == 'blockjacobi':
                    # S3c: no triangular solve over the grid.  theta <- Y' = D^-1 (aoP ao)^2
                    ip_off = self._bj_blocks(rank, clusters)
                    Afac, Dblk = self._bj_prepare(self.ao, 0, d_ip, ip_off, self.aoP, scratch=self.W)
                    self._bj_rows(self.aoP, 0, self.ao, G, Dblk, ip_off, theta)
                else:
                    chol = self._buffer('factor', (P, P))
                    self.reg_used = be.fit_prepare(self.ao, d_ip, self.reg_rel, self.aoP, chol)
                    # forward solve only (Y = Lr^-1 B); the backward solve is applied to the (P, P) matrix below
                    be.fit_apply(chol, self.aoP, self.ao, G, theta, forward_only=not self.explicit_theta)
                t0 = self._tick('S3_fit', t0)
                batch = self.fft_batch or _default_fft_batch(G, P, be.free_bytes())
                be.coulomb_W(theta, mesh, a, 0, P, batch, self.W, upper_only=True)
                be.symmetrize_upper(self.W)
                if route == 'blockjacobi':
                    self._bj_finish(Afac, Dblk, ip_off, self.W)
                elif not self.explicit_theta:
                    be.W_from_factor(chol, 0, self.W)
                t0 = self._tick('S4S5_coulomb_W', t0)
                self.fit_route_used = route
                if route == 'blockjacobi' and self.fit_route == 'auto':
                    aoT = be.empty((nao, P))
                    be.gather_cols(self.ao, d_ip, aoT)
                    self.bj_check = self._bj_probe_mismatch(aoT, Afac, Dblk, ip_off, theta, G, None)
                    del aoT
                    t0 = self._tick('S5_route_check', t0)
                    if self.bj_check <= self.bj_check_tol:
                        break
                    warnings.warn('ISDF: block-Jacobi fit route failed its probe check (mismatch %.2e > %.2e); '
                                  'rebuilding W with the Cholesky route' % (self.bj_check, self.bj_check_tol))
            del theta
            self._built = True
            return self
        else:
            raise ValueError("select must be 'local' or 'global'")

        # S4 + S5 Coulomb convolution and W (global selection)
        self.W = self._buffer('W', (P, P))
        batch = self.fft_batch or _default_fft_batch(G, P)
        be.coulomb_W(theta, mesh, a, 0, P, batch, self.W, upper_only=True)
        be.symmetrize_upper(self.W)
        if factor is not None:
            be.W_from_factor(factor[0], factor[1], self.W)
        self.fit_route_used = 'selection-cholesky'
        del theta
        t0 = self._tick('S4S5_coulomb_W', t0)
        self._built = True
        return self

    # ---- J / K ---------------------------------------------------------------------------------
    def get_jk(self, dm, hermi=1, kpts=None, kpts_band=None, with_j=True, with_k=True, omega=None,
               exxdiv=None):
        if omega is not None:
            raise NotImplementedError('range-separated Coulomb kernel (omega) is not implemented for ISDF')
        if kpts is None:
            kpts = self.kpts
        if not self._is_gamma(kpts) or not self._is_gamma(self.kpts) or not self._is_gamma(kpts_band):
            if self._is_gamma(kpts) and np.asarray(dm).ndim == 2:
                dm = np.asarray(dm)[None]                       # Gamma-point density, band structure requested
            return self._get_jk_kpts(dm, hermi, kpts, kpts_band, with_j, with_k, exxdiv)
        if self.kpts_band is not None:                          # back from a band calculation: Gamma-only build again
            self.kpts_band = None
            self._built = False
        if exxdiv is None:
            exxdiv = self.exxdiv
        if exxdiv not in (None, 'None', 'ewald'):
            raise NotImplementedError("exxdiv=%r: only None and 'ewald' are implemented" % (exxdiv,))
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
        if self.comm.size > 1 or self.force_sharded:
            return self._get_jk_sharded(d_dm, dm_in.shape, with_j, with_k, exxdiv)
        if with_j:
            d_vj = be.empty((nset, nao, nao))
            be.get_j(self.ao, G, mesh, a, d_dm, d_vj)
            t0 = self._tick('S6_get_j', t0)
            vj = be.to_host(d_vj).reshape(dm_in.shape)
        if with_k:
            d_vk = be.empty((nset, nao, nao))
            P = self.W.shape[0]
            be.get_k(self.aoP, self.W, 0, P, d_dm, d_vk)
            if exxdiv == 'ewald':
                self._add_ewald_exxdiv(d_dm, d_vk)
            t0 = self._tick('S7_get_k', t0)
            vk = be.to_host(d_vk).reshape(dm_in.shape)
        return vj, vk

    def get_k_exact(self, dm=None, mo_coeff=None, mo_occ=None, max_rows=None):
        """The reference's exact exchange (FFTDF.get_jk's K, fft_jk.py:177-302) evaluated on the GPU with
        the same device primitives — N*nocc FFT pairs.  Used to measure the ISDF fitting error at full
        size.  Needs the occupied orbitals (mo_coeff, mo_occ) or a positive semidefinite dm."""
        if not self._is_gamma(self.kpts):
            raise NotImplementedError
        if self.ao is None:
            self.build()
        be = self.backend
        nao = self.cell.nao_nr()
        if mo_coeff is None:
            mo_coeff = getattr(dm, 'mo_coeff', None)
            mo_occ = getattr(dm, 'mo_occ', None)
        if mo_coeff is None:
            s, u = np.linalg.eigh(np.asarray(dm, dtype=float))
            if s.min() < -1e-10 * abs(s).max():
                raise ValueError('get_k_exact needs occupied orbitals or a positive semidefinite density matrix')
            keep = s > 1e-12 * s.max()
            c = u[:, keep] * np.sqrt(s[keep])
        else:
            occ = np.asarray(mo_occ, dtype=float)
            c = np.asarray(mo_coeff, dtype=float)[:, occ > 0] * np.sqrt(occ[occ > 0])
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        a = np.asarray(self.cell.lattice_vectors(), dtype=float)
        nocc = c.shape[1]
        if max_rows is None:
            max_rows = max(nocc, int((4 << 30) // (8 * G)) // nocc * nocc)
        vk = be.empty((nao, nao))
        if self.comm.size > 1 or self.force_sharded:
            raise NotImplementedError('get_k_exact is a single-GPU verification path')
        be.get_k_exact(self.ao, G, be.to_device(np.ascontiguousarray(c)), mesh, a, 0, nao, max_rows, vk)
        return be.to_host(vk)

    def overlap(self):
        """AO overlap by quadrature on the FFT grid, S = (vol/G) ao ao^T (device, cached).  The
        reference takes the analytic lattice-sum overlap (df_jk.py:1447); on the meshes this path
        runs at the two agree to the grid's quadrature error."""
        if self._ovlp is None:
            be = self.backend
            nao = self.ao.shape[0]
            G = int(np.prod(self.mesh))
            S = be.empty((nao, nao))
            be.gemm_nt(self.ao, self.ao, S, alpha=self.cell.vol / G)
            if self.comm.size > 1 or self.force_sharded:
                self.comm.all_reduce_sum(S)
            self._ovlp = S
        return self._ovlp

    def _add_ewald_exxdiv(self, d_dm, d_vk):
        """vk += madelung * S D S  (pyscf/pbc/df/df_jk.py:1446-1452, Gamma point)."""
        S = self.overlap()
        mad = gto.madelung(self.cell)
        for i in range(d_dm.shape[0]):
            d_vk[i] += mad * (S @ d_dm[i] @ S)       # N^3, negligible; torch matmul as plumbing

    # ---- multi-GPU: grid-sharded build, row-sharded K (DESIGN.md "Multi-GPU") -------------------------
    def _build_sharded(self):
        """Every rank owns a contiguous slice S_r of the grid (natural order).

        S1  collocation on the slice                               no communication
        S2  per-atom selection, atom blocks dealt round-robin       all_gather of the point lists (P ints)
        S3  A_PP Cholesky replicated (P^3/3, small); fit on slice   no communication
        S4  rows of Theta assembled by all-to-all, FFT convolution, scattered back by all-to-all
        S5  W_r = w V[:, S_r] Theta[:, S_r]^T                       all_reduce(W)  (RCCL over xGMI)
        Streaming over row batches bounds memory at any rank count.
        """
        cell, be, comm = self.cell, self.backend, self.comm
        if self.select != 'local':
            raise NotImplementedError("multi-GPU build needs select='local'")
        R, rk = comm.size, comm.rank
        self.timings = {}
        t0 = time.perf_counter()
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        nao = cell.nao_nr()
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        coords = self.grids.coords
        rcut = gto.estimate_rcut_per_shell(cell)
        Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
        ao_args = (np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut)
        g0, g1 = comm.split_range(G)
        self._slice = (g0, g1)
        ng = g1 - g0
        t0 = self._tick('host_setup', t0)

        # S1 on the slice
        self.ao = self._buffer('ao', (nao, ng))
        be.eval_ao(*ao_args, be.to_device(np.ascontiguousarray(coords[g0:g1].T)), self.ao)
        t0 = self._tick('S1_eval_ao', t0)

        # S2 selection on this rank's atom blocks
        owner = partition_grid_by_atom(coords, cell.atom_coords(), a)
        perm = np.argsort(owner, kind='stable').astype(np.int64)
        counts = np.bincount(owner, minlength=cell.natm)
        blk_off = np.append(0, np.cumsum(counts)).astype(np.int64)
        nip = np.minimum(self.nip_per_atom(), counts).astype(np.int32)
        mine = [b for b in range(cell.natm) if b % R == rk and nip[b] > 0]
        t0 = self._tick('host_partition', t0)
        my_ips = {}
        if mine:
            idx = np.concatenate([perm[blk_off[b]:blk_off[b + 1]] for b in mine])
            loc_off = np.append(0, np.cumsum([counts[b] for b in mine])).astype(np.int64)
            ao_sel = be.empty((nao, len(idx)))
            be.eval_ao(*ao_args, be.to_device(np.ascontiguousarray(coords[idx].T)), ao_sel)
            kmax = int(max(nip[b] for b in mine))
            L = be.empty((kmax, len(idx)))
            piv = be.empty((len(mine), kmax), dtype=torch.int64)
            rank = be.select_ip(ao_sel, loc_off, [nip[b] for b in mine], self.select_tol, self.tie_rtol, L, piv)
            piv_h = be.to_host(piv)
            for k, b in enumerate(mine):
                my_ips[b] = idx[loc_off[k] + piv_h[k, :rank[k]]]
            del ao_sel, L, piv
        all_ips = comm.all_gather_object(my_ips)
        merged = {}
        for d in all_ips:
            merged.update(d)
        clusters = self._bj_clusters()
        self.ip = np.concatenate([merged[b] for cl in clusters for b in cl]).astype(np.int64)
        P = len(self.ip)
        t0 = self._tick('S2_select_ip', t0)

        # S3: phi at the points = columns of the slice collocations (every point lies in exactly one slice; zero-padded
        # all_reduce of 8 P N bytes).  Taking them from the SAME evaluation as the fit's right-hand sides keeps
        # B[:, ip] == A_PP to the last bit (a separate collocation differs by the image-screening tolerance, which
        # the fit amplifies by cond(A)).  P x P factorisations replicated, rows of the fit on the slice.
        aoP_T = be.zeros((nao, P))
        mine_p = np.nonzero((self.ip >= g0) & (self.ip < g1))[0]
        if len(mine_p):
            loc = be.empty((nao, len(mine_p)))
            be.gather_cols(self.ao, be.to_device(self.ip[mine_p] - g0), loc)
            aoP_T[:, be.to_device(mine_p)] = loc
            del loc
        comm.all_reduce_sum(aoP_T)
        self.aoP = self._buffer('aoP', (P, nao))
        theta = self._buffer('theta', (P, ng))
        ar = be.to_device(np.arange(P, dtype=np.int64))
        ip_off = self._bj_blocks([len(merged[b]) for b in range(cell.natm)], clusters)
        for route in self._fit_routes():
            # the P x P factorisations run on rank 0 and are broadcast (2 x 8 P^2 bytes): every rank then holds the
            # same bits, and the shift ladders' decisions cannot diverge between ranks
            if route == 'blockjacobi':
                Afac = self._buffer('factor', (P, P))
                Dblk = self._buffer('Dblk', (P, P))
                def root_factorise():
                    self._bj_prepare(aoP_T, 0, ar, ip_off, self.aoP, scratch=self._buffer('W', (P, P)))
                    return self.reg_used
                reg = comm.run_on_root(root_factorise)
                if comm.rank != 0:
                    be.gather_aoP(aoP_T, ar, self.aoP)
                comm.broadcast(Afac)
                comm.broadcast(Dblk)
                self.reg_used = comm.agree_max(reg or 0.0)
                self._bj_rows(self.aoP, 0, self.ao, ng, Dblk, ip_off, theta)
            else:
                chol = self._buffer('factor', (P, P))
                reg = comm.run_on_root(lambda: be.fit_prepare(aoP_T, ar, self.reg_rel, self.aoP, chol))
                if comm.rank != 0:
                    be.gather_aoP(aoP_T, ar, self.aoP)
                comm.broadcast(chol)
                self.reg_used = comm.agree_max(reg or 0.0)
                be.fit_apply(chol, self.aoP, self.ao, ng, theta, forward_only=not self.explicit_theta)
            t0 = self._tick('S3_fit', t0)

            # S4 + S5 streamed over row batches: rank q convolves rows P_q[t*nb : (t+1)*nb] in step t
            w = cell.vol / G
            self.W = self._buffer('W', (P, P))
            self.W.zero_()
            slices = [comm.split_range(G, r) for r in range(R)]
            rows = [comm.split_range(P, r) for r in range(R)]
            nb = self.fft_batch or _default_fft_batch(G, max(1, P // R))
            nsteps = max(-(-(hi - lo) // nb) for lo, hi in rows)
            for t in range(nsteps):
                bat = [(min(lo + t * nb, hi), min(lo + (t + 1) * nb, hi)) for lo, hi in rows]   # rows handled by rank q
                nrow = [hi - lo for lo, hi in bat]
                # all-to-all 1: send Theta[bat_q, S_r] to q; receive Theta[bat_r, S_q] from q
                send = [theta[lo:hi] for lo, hi in bat]
                recv = [be.empty((nrow[rk], s1 - s0)) for s0, s1 in slices]
                comm.all_to_all(recv, send)
                full = be.empty((nrow[rk], G))
                for (s0, s1), piece in zip(slices, recv):
                    full[:, s0:s1] = piece
                del recv
                if nrow[rk]:
                    be.coulomb_rows(full, mesh, a, max(1, nrow[rk]))
                # all-to-all 2: send V[bat_r, S_q] to q; receive V[bat_q, S_r] from q
                send = [full[:, s0:s1].contiguous() for s0, s1 in slices]
                recv = [be.empty((nrow[q], ng)) for q in range(R)]
                comm.all_to_all(recv, send)
                del full, send
                for q in range(R):
                    if nrow[q]:
                        # W[bat_q, c0:] = w V[bat_q, S_r] Theta[c0:, S_r]^T  (partial over this rank's slice).
                        # W is symmetric: only the columns from the batch's first row on are computed and
                        # the lower part is mirrored after the all-reduce (half the flops).
                        c0 = bat[q][0]
                        be.gemm_nt(recv[q], theta[c0:], self.W[bat[q][0]:bat[q][1], c0:], alpha=w, beta=0.0)
                del recv
            comm.all_reduce_sum(self.W)
            be.symmetrize_upper(self.W)
            if route == 'blockjacobi':
                self._bj_finish_sharded(Afac, Dblk, ip_off, self.W)
            elif not self.explicit_theta:
                be.W_from_factor(chol, 0, self.W)
            t0 = self._tick('S4S5_coulomb_W', t0)
            self.fit_route_used = route
            if route == 'blockjacobi' and self.fit_route == 'auto':
                # replicated W, all-reduced probe energies; the max over ranks makes the decision identical everywhere
                self.bj_check = comm.agree_max(self._bj_probe_mismatch(aoP_T, Afac, Dblk, ip_off, theta, ng, (g0, g1)))
                t0 = self._tick('S5_route_check', t0)
                if self.bj_check <= self.bj_check_tol:
                    break
                warnings.warn('ISDF: block-Jacobi fit route failed its probe check (mismatch %.2e > %.2e); '
                              'rebuilding W with the Cholesky route' % (self.bj_check, self.bj_check_tol))
        del theta, aoP_T
        self._built = True
        return self

    def _get_jk_sharded(self, d_dm, out_shape, with_j, with_k, exxdiv=None):
        cell, be, comm = self.cell, self.backend, self.comm
        nao = cell.nao_nr()
        nset = d_dm.shape[0]
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        g0, g1 = self._slice
        vj = vk = None
        t0 = time.perf_counter()
        if with_j:
            # rho on the slice -> all_reduce of the zero-padded density -> potential (replicated FFT) ->
            # vj partial from the slice -> all_reduce
            rho = be.zeros((nset, G))
            rho_loc = be.empty((nset, g1 - g0))
            be.rho(self.ao, g1 - g0, d_dm, rho_loc)
            rho[:, g0:g1] = rho_loc
            comm.all_reduce_sum(rho)
            be.coulomb_potential(rho, mesh, a)
            d_vj = be.empty((nset, nao, nao))
            be.vj_from_vR(self.ao, g1 - g0, rho[:, g0:g1].contiguous(), d_vj)
            comm.all_reduce_sum(d_vj)
            t0 = self._tick('S6_get_j', t0)
            vj = be.to_host(d_vj).reshape(out_shape)
        if with_k:
            P = self.W.shape[0]
            r0, r1 = comm.split_range(P)
            d_vk = be.empty((nset, nao, nao))
            be.get_k(self.aoP, self.W, r0, r1 - r0, d_dm, d_vk)
            comm.all_reduce_sum(d_vk)
            if exxdiv == 'ewald':
                self._add_ewald_exxdiv(d_dm, d_vk)
            t0 = self._tick('S7_get_k', t0)
            vk = be.to_host(d_vk).reshape(out_shape)
        return vj, vk


    # ---- k-points (BASELINE configs[3]); DESIGN.md "k-points" -----------------------------------------
    def _build_kpts(self):
        """Periodic parts u^k of all Bloch AOs -> real points/Theta (complex-mode S2/S3) -> one complex
        W^q per difference vector q = k2 - k1.  The q list is split over the ranks (each rank holds the
        fit, builds its share of the W^q and later the K terms that use them)."""
        from . import pbc_tools
        cell, be, comm = self.cell, self.backend, self.comm
        self.timings = {}
        t0 = time.perf_counter()
        kpts_scf = np.asarray(self.kpts, dtype=float).reshape(-1, 3)
        # band k-points (kpts_band of get_jk) join the stack: the fit must also represent conj(u^{kb}) u^{k}
        band = kpts_scf if self.kpts_band is None else np.asarray(self.kpts_band, dtype=float).reshape(-1, 3)
        kall = [k for k in kpts_scf]
        self._band_index = []
        for kb in band:
            hit = [i for i, k in enumerate(kall) if abs(k - kb).max() < 1e-9]
            if hit:
                self._band_index.append(hit[0])
            else:
                kall.append(kb)
                self._band_index.append(len(kall) - 1)
        kpts = np.array(kall)
        nk = len(kpts)                       # size of the stack; the first len(kpts_scf) entries carry density
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        nao = cell.nao_nr()
        nh = nk * nao
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        coords = self.grids.coords
        rcut = gto.estimate_rcut_per_shell(cell)
        Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
        ao_args = (np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut)
        t0 = self._tick('host_setup', t0)

        coords_soa = be.to_device(np.ascontiguousarray(coords.T))
        X = self._buffer('aok', (2 * nh, G))
        for k in range(nk):
            be.eval_ao_k(*ao_args, kpts[k], True, coords_soa, X[k * nao:(k + 1) * nao], X[nh + k * nao:nh + (k + 1) * nao])
        del coords_soa
        self.ao = X
        t0 = self._tick('S1_eval_ao', t0)

        # S2 selection (complex mode); the number of points scales with the number of distinct pair
        # families: c_isdf * nao * nk by default (capped by the grid)
        kfac = self.k_ip_factor or min(nk, 2)
        P_target = int(min(self.c_isdf * nao * kfac, G))
        if self.select == 'global':
            theta = self._buffer('theta', (P_target, G))
            piv = be.empty((1, P_target), dtype=torch.int64)
            rank = be.select_ip_cplx(X, nh, [0, G], [P_target], self.select_tol, self.tie_rtol, theta, piv)
            ip_dev = piv[0, :int(rank[0])].contiguous()
            self.ip = be.to_host(ip_dev).astype(np.int64)
        else:
            owner = partition_grid_by_atom(coords, cell.atom_coords(), a)
            perm = np.argsort(owner, kind='stable').astype(np.int64)
            counts = np.bincount(owner, minlength=cell.natm)
            blk_off = np.append(0, np.cumsum(counts)).astype(np.int64)
            nip = np.minimum(self.nip_per_atom() * kfac, counts).astype(np.int32)
            kmax = int(nip.max())
            Xs = be.empty((2 * nh, G))
            be.gather_cols(X, be.to_device(perm), Xs)
            L = be.empty((kmax, G))
            piv = be.empty((cell.natm, kmax), dtype=torch.int64)
            rank = be.select_ip_cplx(Xs, nh, blk_off, nip, self.select_tol, self.tie_rtol, L, piv)
            del Xs, L
            piv_h = be.to_host(piv)
            clusters = self._bj_clusters()
            self.ip = np.concatenate([perm[blk_off[b] + piv_h[b, :rank[b]]] for cl in clusters for b in cl]).astype(np.int64)
            ip_dev = be.to_device(self.ip)
        P = len(self.ip)
        t0 = self._tick('S2_select_ip', t0)

        # S3 global fit, forward solve only (Y); the factor is applied to the (P, P) matrices
        Y = self._buffer('theta', (max(P, P_target), G))[:P]
        aoP_X = self._buffer('aoP', (P, 2 * nh))
        # q list: W^{-q} = conj(W^q) (Theta is real, coulG_{-q}[-G] = coulG_q[G]): build one of each +-q pair,
        # the primaries dealt round-robin over the ranks
        self._qs, self._qindex = pbc_tools.unique_q(kpts_scf, band)      # index[k1 in band][k2 in kpts]
        nq = len(self._qs)
        w = cell.vol / G
        batch = self.fft_batch or max(1, min(P, int((3 << 30) // (8 * G)) // 128 * 128 or 64))
        partner = -np.ones(nq, dtype=int)
        for iq in range(nq):
            for jq in range(nq):
                if abs(self._qs[iq] + self._qs[jq]).max() < 1e-9:
                    partner[iq] = jq
        primary = [iq for iq in range(nq) if partner[iq] < 0 or partner[iq] >= iq]
        self._q_owner = np.zeros(nq, dtype=int)
        for n, iq in enumerate(primary):
            self._q_owner[iq] = n % comm.size
            if partner[iq] >= 0:
                self._q_owner[partner[iq]] = n % comm.size
        r_ip = coords[self.ip]
        Wre = self._buffer('Wre', (P, P))
        Wim = self._buffer('Wim', (P, P))

        # S3 + S4 + S5, route by route.  'auto' means the Cholesky route here unless bj_auto_kpts is set; then: block-
        # Jacobi, verified on W^{q=0}, Cholesky when the check fails (the fit is replicated, so every rank takes the
        # agreed decision after its share of the q list)
        routes = self._fit_routes() if self.select != 'global' else ['cholesky']
        if self.fit_route == 'auto' and not self.bj_auto_kpts:
            routes = ['cholesky']
        for route in routes:
            if route == 'blockjacobi':
                ip_off = self._bj_blocks(rank, clusters)
                Afac, Dblk = self._bj_prepare(X, nh, ip_dev, ip_off, aoP_X)
                self._bj_rows(aoP_X, nh, X, G, Dblk, ip_off, Y)
            else:
                chol = self._buffer('factor', (P, P))
                self.reg_used = be.fit_prepare_cplx(X, nh, ip_dev, self.reg_rel, aoP_X, chol)
                be.fit_apply_cplx(chol, aoP_X, nh, X, G, Y, forward_only=not self.explicit_theta)
            t0 = self._tick('S3_fit', t0)
            self._Wq = {}
            check = 0.0
            for iq in primary:
                if self._q_owner[iq] != comm.rank:
                    continue
                q = self._qs[iq]
                coulG = be.to_device(pbc_tools.get_coulG(cell, q, mesh))
                be.coulomb_Wq(Y, mesh, coulG, w, 0, P, batch, Wre, Wim, upper_only=True)
                be.symmetrize_hermitian(Wre, Wim)
                if route == 'blockjacobi':
                    self._bj_finish(Afac, Dblk, ip_off, Wre)
                    self._bj_finish(Afac, Dblk, ip_off, Wim, antisymmetric=True)
                    if self.fit_route == 'auto' and abs(q).max() < 1e-9:
                        # W^0 is real: the Gamma-point probe check with the densities sum_k u^k* R u^k
                        t1 = self._tick('S4S5_coulomb_W', t0)
                        planes = [aoP_X[:, o:o + nao].T.contiguous() for o in range(0, 2 * nh, nao)]
                        check = self._bj_probe_mismatch(planes, Afac, Dblk, ip_off, Y, G, None, W=Wre)
                        del planes
                        t0 = self._tick('S5_route_check', t1)
                elif not self.explicit_theta:
                    be.W_from_factor(chol, 0, Wre)
                    be.W_from_factor(chol, 0, Wim)
                Wc = be.empty((P, P), dtype=torch.complex128)
                be.finish_Wq(Wre, Wim, be.to_device(np.exp(-1j * r_ip.dot(q))), Wc)
                self._Wq[iq] = Wc
            self.fit_route_used = route
            if route == 'blockjacobi' and self.fit_route == 'auto':
                self.bj_check = comm.agree_max(check)
                if self.bj_check <= self.bj_check_tol:
                    break
                warnings.warn('ISDF: block-Jacobi fit route failed its probe check (mismatch %.2e > %.2e); '
                              'rebuilding the W^q with the Cholesky route' % (self.bj_check, self.bj_check_tol))
                t0 = self._tick('S4S5_coulomb_W', t0)

        # Bloch AOs at the points: phi^k(r_P) = exp(i k.r_P) u^k(r_P), (P, nao) complex per k
        uP = be.to_host(aoP_X)                                   # (P, 2 nh)
        self._aoP_k = []
        for k in range(nk):
            u = uP[:, k * nao:(k + 1) * nao] + 1j * uP[:, nh + k * nao:nh + (k + 1) * nao]
            self._aoP_k.append(be.to_device(np.ascontiguousarray(u * np.exp(1j * r_ip.dot(kpts[k]))[:, None])))
        self._q_partner = partner
        t0 = self._tick('S4S5_coulomb_W', t0)
        self._built = True
        self._k_built = kpts_scf.copy()
        self._band_built = None if self.kpts_band is None else band.copy()
        self._nk_stack = nk
        return self

    def _get_jk_kpts(self, dm, hermi, kpts, kpts_band, with_j, with_k, exxdiv):
        """k-point J and K (pyscf/pbc/df/fft_jk.py:33-109,177-302 semantics).  dm (nk, N, N) or (nset, nk, N, N); with
        kpts_band the result lives on the band k-points, (nband, N, N) [(N, N) for a single (3,) band vector], as
        df_jk._format_jks shapes it (pyscf/pbc/df/df_jk.py:1426-1444)."""
        ex = exxdiv if exxdiv is not None else self.exxdiv
        if ex not in (None, 'None', 'ewald'):
            raise NotImplementedError("k-point ISDF: only exxdiv=None and 'ewald' are implemented")
        cell, be, comm = self.cell, self.backend, self.comm
        kpts = np.asarray(kpts, dtype=float).reshape(-1, 3)
        band_in = None if kpts_band is None else np.asarray(kpts_band, dtype=float)
        band = None if band_in is None else band_in.reshape(-1, 3)

        def same(x, y):
            if x is None or y is None:
                return x is None and y is None
            return x.shape == y.shape and abs(x - y).max() < 1e-9
        if not self._built or getattr(self, '_k_built', None) is None or not same(kpts, self._k_built) \
                or not same(band, getattr(self, '_band_built', None)):
            self.kpts = kpts
            self.kpts_band = band
            self.build()
        nk = len(kpts)
        nks = self._nk_stack                                     # k-points in the stacked periodic parts
        bidx = list(range(nk)) if band is None else list(self._band_index)
        nband = len(bidx)
        nao = cell.nao_nr()
        nh = nks * nao
        dm_in = np.asarray(dm)
        dms = np.asarray(dm_in, dtype=np.complex128).reshape(-1, nk, nao, nao)
        nset = dms.shape[0]
        if hermi != 1 and with_j:
            if abs(dms - dms.conj().transpose(0, 1, 3, 2)).max() > 1e-10:
                raise NotImplementedError('non-Hermitian density matrices (complex density) are not implemented for J')
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        X = self.ao

        def planes(k):
            return X[k * nao:(k + 1) * nao], X[nh + k * nao:nh + (k + 1) * nao]
        out_shape = dm_in.shape if band is None else \
            (dm_in.shape[:-3] + ((nband,) if band_in.ndim > 1 else ()) + (nao, nao))
        vj = vk = None
        t0 = time.perf_counter()
        if with_j:
            vj = np.zeros((nset, nband, nao, nao), dtype=np.complex128)
            for s in range(nset):
                rho = be.zeros((1, G))
                for k in range(nk):
                    dT = dms[s, k].T
                    be.rho_k(*planes(k), G, be.to_device(np.ascontiguousarray(dT.real)),
                             be.to_device(np.ascontiguousarray(dT.imag)), 1.0 / nk, rho)
                be.coulomb_potential(rho, mesh, a)
                for ib, kb in enumerate(bidx):
                    vre = be.empty((nao, nao))
                    vim = be.empty((nao, nao))
                    be.vj_k(*planes(kb), G, rho, vre, vim)
                    vj[s, ib] = be.to_host(vre) + 1j * be.to_host(vim)
            t0 = self._tick('S6_get_j', t0)
            vj = vj.reshape(out_shape)
        if with_k:
            d_vk = be.zeros((nset, nband, nao, nao), dtype=torch.complex128)
            for s in range(nset):
                d_dm = [be.to_device(np.ascontiguousarray(dms[s, k])) for k in range(nk)]
                for i1, k1 in enumerate(bidx):
                    for k2 in range(nk):
                        iq = self._qindex[i1, k2]
                        if self._q_owner[iq] != comm.rank:
                            continue
                        if iq in self._Wq:
                            Wq = self._Wq[iq]
                        else:                      # stored as its time-reversal partner: W^{-q} = conj(W^q)
                            Wq = torch.conj_physical(self._Wq[self._q_partner[iq]])
                        be.get_k_pair(self._aoP_k[k1], self._aoP_k[k2], d_dm[k2], Wq, 1.0 / nk, d_vk[s, i1])
            if comm.size > 1:
                flat = torch.view_as_real(d_vk)
                comm.all_reduce_sum(flat)
            vk = be.to_host(d_vk)
            if ex == 'ewald':
                # vk[k] += madelung * S^k D^k S^k (pyscf/pbc/df/df_jk.py:1446-1465) for the band k-points that are
                # k-points of the density; S^k by quadrature on the grid from the periodic parts (the phases cancel)
                mad = gto.madelung(cell, _monkhorst_pack_size(cell, kpts))
                w_const = be.to_device(np.full((1, G), cell.vol / G))
                for ib, kb in enumerate(bidx):
                    if kb >= nk:
                        continue
                    sre, sim = be.empty((nao, nao)), be.empty((nao, nao))
                    be.vj_k(*planes(kb), G, w_const, sre, sim)
                    Sk = be.to_host(sre) + 1j * be.to_host(sim)
                    for s in range(nset):
                        vk[s, ib] += mad * Sk.dot(dms[s, kb]).dot(Sk)
            t0 = self._tick('S7_get_k', t0)
            vk = vk.reshape(out_shape)
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

    def ao2mo(self, mo_coeffs, kpts=None, compact=False):
        """(ij|kl) in the MO basis from the factorisation (FFTDF.ao2mo surface, pyscf/pbc/df/fft.py:319):
        sum_PQ X_ij,P W_PQ X_kl,Q with X_ij,P = (aoP C_i)_P (aoP C_j)_P.  Gamma point, s1 layout."""
        if not self._is_gamma(kpts) or not self._is_gamma(self.kpts):
            raise NotImplementedError
        if compact:
            raise NotImplementedError('compact MO integrals are not implemented; use compact=False')
        if not self._built:
            self.build()
        if isinstance(mo_coeffs, np.ndarray) and mo_coeffs.ndim == 2:
            mo_coeffs = (mo_coeffs,) * 4
        aoP = self.backend.to_host(self.aoP)
        W = self.backend.to_host(self.W)
        ci, cj, ck, cl = [aoP.dot(np.asarray(c)) for c in mo_coeffs]
        Xij = np.einsum('pi,pj->pij', ci, cj).reshape(len(aoP), -1)
        Xkl = np.einsum('pk,pl->pkl', ck, cl).reshape(len(aoP), -1)
        return Xij.T.dot(W).dot(Xkl)

    get_mo_eri = ao2mo

    def get_pp(self, kpts=None):
        """GTH pseudopotential AO matrix (G=0 removed), pyscf/pbc/df/fft.py:64-152: local part on the FFT
        grid, non-local part from projector/AO overlaps in reciprocal space — both on the device
        (pp.hip); only the final nproj-sized contraction with the h_ij matrices runs on the host.
        Returns (nao,nao) for Gamma / a single k-point, else (nk,nao,nao)."""
        cell, be = self.cell, self.backend
        pseudo = getattr(cell, '_pseudo', None) or {}
        if kpts is None:
            kpts_lst, single = np.zeros((1, 3)), True
        else:
            kpts_lst = np.reshape(kpts, (-1, 3))
            single = np.ndim(kpts) == 1
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        nao = cell.nao_nr()
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        acoords = np.asarray(cell.atom_coords(), dtype=float)
        charges = np.asarray(cell.atom_charges(), dtype=float)
        pp_par = np.zeros((cell.natm, 8))
        proj_tab, proj_rl, blocks = [], [], []          # blocks: (row0, l, nl, h) per atom and l-channel
        row = 0
        for ia in range(cell.natm):
            pp = pseudo.get(cell.atom_symbol(ia))
            pp_par[ia, 1] = charges[ia]
            if pp is None:
                continue
            rloc, nexp, cexp = pp[1], pp[2], pp[3]
            pp_par[ia, 0], pp_par[ia, 2], pp_par[ia, 3] = 1.0, rloc, nexp
            pp_par[ia, 4:4 + nexp] = cexp
            for l, (rl, nl, hl) in enumerate(pp[5:]):
                if nl == 0:
                    continue
                if l > 2 or nl > 3:
                    raise NotImplementedError('projectors with l > 2 or more than 3 per channel')
                blocks.append((row, l, nl, np.asarray(hl, dtype=float)))
                for ii in range(nl):
                    proj_tab.append((ia, l, ii))
                    proj_rl.append(rl)
                    row += 2 * l + 1
        vlocR = be.empty((1, G))
        be.pp_local_potential(acoords, pp_par, mesh, a, vlocR)
        rcut = gto.estimate_rcut_per_shell(cell)
        Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
        ao_args = (np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut)
        coords_soa = be.to_device(np.ascontiguousarray(self.grids.coords.T))
        ur, ui = be.empty((nao, G)), be.empty((nao, G))
        out = []
        for k in kpts_lst:
            gamma = abs(k).sum() < 1e-9
            if gamma:
                be.eval_ao(*ao_args, coords_soa, ur)
                v = be.empty((1, nao, nao))
                be.vj_from_vR(ur, G, vlocR, v)
                vpp = be.to_host(v)[0].astype(np.complex128)
            else:
                be.eval_ao_k(*ao_args, k, True, coords_soa, ur, ui)
                vre, vim = be.empty((nao, nao)), be.empty((nao, nao))
                be.vj_k(ur, ui, G, vlocR, vre, vim)
                vpp = be.to_host(vre) + 1j * be.to_host(vim)
            if proj_tab:
                ov = be.empty((row, nao), dtype=torch.complex128)
                be.pp_projector_overlaps(ao_args[0], ao_args[1], ao_args[2], acoords, k, np.array(proj_tab), np.array(proj_rl),
                                         mesh, a, ov)
                S = be.to_host(ov)
                vnl = np.zeros((nao, nao), dtype=np.complex128)
                for row0, l, nl, hl in blocks:
                    deg = 2 * l + 1
                    blk = S[row0:row0 + nl * deg].reshape(nl, deg, nao)
                    vnl += np.einsum('imp,ij,jmq->pq', blk.conj(), hl, blk)
                vpp = vpp + vnl / cell.vol
            out.append(vpp.real if gamma else vpp)
        return out[0] if single else np.asarray(out)

    def get_nuc(self, kpts=None):
        """Nuclear-attraction AO matrix with the G=0 term removed, pyscf/pbc/df/fft.py:39-62:
        vne^k = ao_k^H (vneR ao_k),  vneR = ifft(coulG * sum_a (-Z_a) exp(-i G.R_a)).real.
        The potential is assembled on the host (O(G natm)); the contraction runs on the device with the
        J kernels (isdf_vj_from_vR / isdf_vj_k).  Returns (nao,nao) for a single k-point (or Gamma),
        else (nk,nao,nao), like the reference."""
        from . import pbc_tools
        cell, be = self.cell, self.backend
        if kpts is None:
            kpts_lst, single = np.zeros((1, 3)), True
        else:
            kpts_lst = np.reshape(kpts, (-1, 3))
            single = np.ndim(kpts) == 1
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        nao = cell.nao_nr()
        Gv = cell.get_Gv(mesh)
        charge = -np.asarray(cell.atom_charges(), dtype=float)
        SI = np.exp(-1j * np.dot(cell.atom_coords(), Gv.T))
        rhoG = charge.dot(SI)
        vneG = rhoG * pbc_tools.get_coulG(cell, np.zeros(3), mesh)
        vneR = np.fft.ifftn(vneG.reshape(*mesh)).real.ravel()
        d_v = be.to_device(vneR.reshape(1, G))
        rcut = gto.estimate_rcut_per_shell(cell)
        Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
        ao_args = (np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut)
        coords_soa = be.to_device(np.ascontiguousarray(self.grids.coords.T))
        out = []
        ur = be.empty((nao, G))
        ui = be.empty((nao, G))
        for k in kpts_lst:
            if abs(k).sum() < 1e-9:
                be.eval_ao(*ao_args, coords_soa, ur)
                v = be.empty((1, nao, nao))
                be.vj_from_vR(ur, G, d_v, v)
                out.append(be.to_host(v)[0])
            else:
                be.eval_ao_k(*ao_args, k, True, coords_soa, ur, ui)
                vre, vim = be.empty((nao, nao)), be.empty((nao, nao))
                be.vj_k(ur, ui, G, d_v, vre, vim)
                out.append(be.to_host(vre) + 1j * be.to_host(vim))
        return out[0] if single else np.asarray(out)


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
