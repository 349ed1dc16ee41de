"""``ISDF``: a drop-in for ``pyscf.pbc.df.FFTDF`` whose K build runs through interpolative separable
density fitting on MI355X.

Public surface = FFTDF's (pyscf/pbc/df/fft.py:155-359): ``__init__(cell, kpts)``, attributes
``cell, kpts, mesh, grids, stdout, verbose, max_memory, exxdiv, blockdim``, and ``build()``,
``reset()``, ``dump_flags()``, ``check_sanity()``, ``get_jk(dm, hermi, kpts, kpts_band, with_j,
with_k, omega, exxdiv)``, ``get_naoaux()``, ``get_ao_eri()/get_eri()``, ``update_mf()``; SCF callers
(pyscf/pbc/scf/hf.py:649-698) only ever use these.  Array conventions of get_jk follow
pyscf/pbc/df/df_jk.py:1411-1444: the result has the shape of ``dm``; Γ point + real dm -> float64.

What runs where: this file and its mixins are host orchestration only (which stage, which buffers, which rank);
every stage executes in libmi355_isdf.so via ``backend.HipBackend``.  There is no CPU path.
Modules: ``isdf`` (the object, Gamma-point single-GPU build and get_jk, ERIs, range separation), ``fit_route`` (fit routes and
the probe check), ``sharded`` (grid-sharded multi-GPU build), ``kpoints`` (k-points, band k-points, k-point ERIs),
``hcore`` (get_nuc / get_pp).
"""
import os
import sys
import time
import warnings
import numpy as np
import torch
from . import gto
from ._common import (UniformGrids, partition_grid_by_atom, _monkhorst_pack_size, _aoslice_by_atom,  # noqa: F401
                      _default_fft_batch)
from .fit_route import FitRouteMixin
from .sharded import ShardedMixin
from .kpoints import KPointMixin
from .hcore import HcoreMixin
from .eri_surface import EriSurfaceMixin


class ISDF(FitRouteMixin, ShardedMixin, KPointMixin, HcoreMixin, EriSurfaceMixin):
    _keys = {'cell', 'kpts', 'grids', 'mesh', 'blockdim', 'exxdiv', 'c_isdf', 'select', 'tie_rtol'}

    def __init__(self, cell, kpts=np.zeros((1, 3)), c_isdf=12, select='refined', backend=None, comm=None):
        self.cell = cell
        self.stdout = getattr(cell, 'stdout', None) or sys.stdout
        self.verbose = getattr(cell, 'verbose', 0)
        self.max_memory = getattr(cell, 'max_memory', 4000)
        self.kpts_symm = None             # a kpts_symm.KPoints object passed as kpts: get_jk then takes the density matrices on
                                          # the irreducible k-points and returns J, K there (khf_ksymm.py:210-237)
        if hasattr(kpts, 'kpts_ibz') and hasattr(kpts, 'transform_dm'):
            self.kpts_symm = kpts
            kpts = kpts.kpts
        self.kpts = np.asarray(kpts).reshape(-1, 3)
        self.grids = UniformGrids(cell, cell.mesh)
        self.blockdim = 240
        self.exxdiv = None
        self.c_isdf = c_isdf
        self.select = select              # 'local': per-atom Voronoi blocks + global fit; 'global': one block;
                                          # 'refined': local candidates (refine_over x too many), then ONE pivoted Cholesky
                                          # restricted to the candidate set picks the final points
        self.refine_over = 2.0            # 'refined': candidates per atom = refine_over * c_isdf * nao_atom
        self.pair_space = 'ao'            # 'ao': interpolate all AO pairs (the fit does not depend on the density: build once, any
                                          # number of get_jk); 'occ': interpolate the (AO x occupied orbital) pairs of the density
                                          # get_jk is called with (mo_coeff/mo_occ tag, fft_jk.py:206-210; untagged: eigenvectors of
                                          # a positive semidefinite low-rank D) - build() stops after the candidate stage, the
                                          # final pick, the fit and W are made in get_jk (Gamma point, select 'local'/'refined')
        self.occ_refit = 'always'         # pair_space='occ': 'always' = refit whenever get_jk sees another occupied space;
                                          # 'once' = keep the first fit until the next build()
        self.w_spectral = True            # block-Jacobi route on one GPU: W = X X^T from the half spectra of the fit rows inside a sphere of the
                                          # reciprocal FFT box (fit_route._spectral_plan): no inverse transform, about half the P^2 G product
        self.w_sphere = 'auto'            # percent of the inscribed sphere's radius kept (0: the whole box = the reference's sum to rounding);
                                          # 'auto': 100 when the mesh resolves the AO pair products (share of their Coulomb energy outside the
                                          # sphere <= w_sphere_tol, measured once per mesh), the classic build otherwise
        self.w_sphere_tol = 1e-11
        self.w_sort_bins = 256            # packed points sorted into this many shells of |G|^2, outermost first (0: as they lie in the half spectrum)
        self.w_spectral_check_tol = 5e-9  # the spectral form is kept only when the route's probe mismatch stays below this (the classic form
                                          # is held to bj_check_tol): its rounding, amplified like the classic form's, shows up there first
        self.w_spectral_max_c = 18        # the spectral form carries a few times the classic product's rounding (both operands come out of a
                                          # transform), amplified like it by cond(A')^2: above this c_isdf the probe check rejects it at configs[2]
        self.cand_skip_zero_rows = True   # the per-atom selections skip the AO rows that are identically zero on the atom's block of
                                          # grid points (the collocation truncates every shell at its rcut): same pivots, less traffic
        self.cand_ao_cutoff = None        # 'refined', Bohr: the CANDIDATE stage of an atom's block sees only the AOs of atoms
                                          # within this distance (minimum image); None: all AOs.  The final pick always uses all.
        self.tie_rtol = 1e-10
        self.select_tol = -1.0           # stop when the largest residual diagonal <= tol; < 0: m*eps*max diag (scipy_helper.py:88-90)
        self.reg_rel = 1e-12             # relative diagonal shift of A_PP in the global fit
        self.reg_used = 0.0
        self.k_ip_factor = None           # k-points: points = c_isdf * nao * k_ip_factor (default min(nk, 2))
        self.kpt_pair_q = 'auto'          # k-points: True = build one W^q per +-q pair and use W^{-q} = conj(W^q) (half the products);
                                          # exact on odd meshes only - on an even mesh the Nyquist index has no partner and the
                                          # wrap-around rule zeroes it for one sign of q (MgO 2x2x2 / 64^3: 7e-6 in K); False =
                                          # every q from its own kernel table; 'auto' = pair on all-odd meshes only
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
        self.robust_k = False            # True: K with Dunlap's robust correction (error quadratic in the fit error): needs
                                         # Theta itself and keeps V = conv(Theta) on the device; K costs 4 N G P flop
        self.explicit_theta = False      # True: form Theta itself (second O(P^2 G) solve); same W in exact arithmetic
        self.fft_batch = None             # rows per FFT batch (None: sized from free memory)
        self.block_apply_mfma = True      # block solves over the grid through explicit block inverses on the matrix cores
        self.max_resident_rows = None     # fit rows held in HBM at once (None: from free memory); fewer than the number of
                                          # points -> the rows are produced panel by panel (block-Jacobi route)
        self.max_device_memory = None     # bytes the build may occupy on the device (None: what is free).  The reference's
                                          # max_memory (MB of HOST memory, numint.py:1236-1257) sizes its grid blocks; here the
                                          # blocking unit is the panel of fit rows: a smaller cap means more panels, same W
        self.n_panels = 1
        self._backend = backend
        self._comm = comm
        self._rsh_df = {}
        self._built = False
        self._W_omega = {}               # range-separated W per omega (get_jk(omega=...)), valid until the next build
        self._fit_state = None
        self._fit_pending = False        # pair_space='occ': candidates selected, the fit waits for a density
        self._psi = self._psiP = None    # pair_space='occ': occupied orbitals on the grid (nocc, G) and at the points (P, nocc)
        self._fit_dm = None              # the projector sum_i psi_i psi_i^T the current fit was made for (host, N x N)
        self._V = None                   # robust_k: V = conv(Theta) (P, G), in the fit buffer
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
    def _want_theta(self):
        """Theta itself is formed when asked for, and always for robust_k (its correction needs V = conv(Theta))."""
        return bool(self.explicit_theta or self.robust_k)

    @property
    def _sharded(self):
        """The grid-sharded (multi-GPU) code path: more than one rank, or forced for tests (force_sharded; a Comm that
        issues its collectives even with one rank)."""
        return self.comm.size > 1 or self.force_sharded or getattr(self.comm, 'always', False)

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
        self._drop_build_state()
        self._bufs = {}
        self._rows_plan = (None,)
        if self._backend is not None:
            self._backend.empty_cache()
        self._rsh_df = {}
        return self

    def release_fit_buffers(self):
        """Give back the big buffers of the fit (rows, recompute scratch, P x P matrices) and the library's workspaces, keeping
        the collocation phi: what get_k_exact needs, and nothing else.  The object has to be rebuilt before the next get_jk."""
        ao = self.ao
        self._drop_build_state()
        self.ao = ao
        for name in ('theta', 'rows_scratch', 'W', 'factor', 'Dblk', 'Dinv', 'aoP', 'psi', 'psiP'):
            self._bufs.pop(name, None)
        self._Dinv = None
        self._Dinv_key = None
        if self._backend is not None:
            self._backend.release_workspace()
            self._backend.empty_cache()
        return self

    def _drop_build_state(self):
        """Forget everything a build produced (views of the persistent buffers included: the 214 GiB fit buffer must not
        stay referenced from a stale fit state when reset() lets the buffers go).  Called by reset() and at the top of
        every build path."""
        self.ao = self.aoP = self.W = self.ip = None
        self._fit_state = None
        self._fit_pending = False
        self._psi = self._psiP = None
        self._fit_dm = None
        self._sel = None
        self._Dinv = None
        self._Dinv_key = None
        self._kfit_state = None
        self._V = None
        self._Wq = None
        self._aoP_k = None
        self._W_omega = {}
        self._k_built = None
        self._band_built = None
        self._ovlp = None
        self._built = False
        self._build_serial = getattr(self, '_build_serial', 0) + 1

    def dump_flags(self, verbose=None):
        out = self.stdout
        out.write('\n******** %s ********\n' % self.__class__)
        out.write('mesh = %s (%d PWs)\n' % (self.mesh, np.prod(self.mesh)))
        out.write('c_isdf = %s  select = %s  pair_space = %s  tie_rtol = %g\n' % (self.c_isdf, self.select, self.pair_space, self.tie_rtol))
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

    def collocate(self):
        """S1 alone: phi on the grid (what get_k_exact needs), without selecting points or fitting."""
        cell, be = self.cell, self.backend
        self._drop_build_state()
        G = int(np.prod(self.mesh))
        rcut = gto.estimate_rcut_per_shell(cell)
        Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
        self.ao = self._buffer('ao', (cell.nao_nr(), G))
        be.eval_ao(np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut,
                   be.to_device(np.ascontiguousarray(self.grids.coords.T)), self.ao)
        return self

    def build(self):
        self.check_sanity()
        self._drop_build_state()
        if not self._is_gamma(self.kpts) or not self._is_gamma(self.kpts_band):
            return self._build_kpts()
        if self._sharded:
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
        if self.pair_space not in ('ao', 'occ'):
            raise ValueError("pair_space must be 'ao' or 'occ'")
        if self.select == 'global':
            if self.pair_space == 'occ':
                raise NotImplementedError("pair_space='occ' is implemented for select='local' and 'refined'")
            P = int(min(self.c_isdf * nao, G))
            theta = self._buffer('theta', (P, G))
            piv = be.empty((1, P), dtype=torch.int64)
            rank = be.select_ip(self.ao, [0, G], [P], self.select_tol, self.tie_rtol, theta, piv)
            P = int(rank[0])
            t0 = self._tick('S2_select_ip', t0)
            theta = theta[:P]
            piv = piv[0, :P].contiguous()
            if self._want_theta:
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
        elif self.select in ('local', 'refined'):
            self._select_candidates(t0)
            if self.pair_space == 'occ':
                # the final pick, the fit and W need the occupied orbitals: made by get_jk (_ensure_fit)
                self._fit_pending = True
                self._built = True
                return self
            self._pick_and_fit()
            return self
        else:
            raise ValueError("select must be 'local', 'refined' or 'global'")

        # S4 + S5 Coulomb convolution and W (global selection)
        self.W = self._buffer('W', (P, P))
        self._fit_state = dict(kind='explicit', theta=theta) if factor is None else \
            dict(kind='selection', theta=theta, T=factor[0])
        self._finish_W(self.W)
        self.fit_route_used = 'selection-cholesky'
        del theta
        t0 = self._tick('S4S5_coulomb_W', t0)
        self._keep_V_for_robust_k(t0)
        self._built = True
        return self

    def _select_candidates(self, t0):
        """S2, first stage, for select='local' / 'refined': Voronoi partition of the grid by atoms and the per-atom
        pivoted-Cholesky selections on the AO-pair Gram matrix (for 'refined': refine_over x too many points each, the
        CANDIDATES).  Does not depend on the density; the result stays in self._sel for _pick_and_fit."""
        cell, be = self.cell, self.backend
        nao, G = self.ao.shape
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        owner = be.partition_by_atom(self.grids.coords, cell.atom_coords(), a)
        perm = np.argsort(owner, kind='stable').astype(np.int64)
        counts = np.bincount(owner, minlength=cell.natm)
        blk_off = np.append(0, np.cumsum(counts)).astype(np.int64)
        nip_final = np.minimum(self.nip_per_atom(), counts).astype(np.int32)
        nip = nip_final
        if self.select == 'refined':
            nip = np.minimum(np.ceil(self.nip_per_atom() * float(self.refine_over)).astype(np.int64), counts).astype(np.int32)
        kmax = int(nip.max())
        t0 = self._tick('host_partition', t0)
        d_perm = be.to_device(perm)
        # the block-major copy of phi and the Cholesky rows are scratch that dies before the fit:
        # they live inside the (P, G) fit buffer, which is not in use yet
        Pmax = int(nip_final.sum())
        # how many fit rows HBM can hold next to everything else: above that the rows are produced panel by panel
        # (fit_route.FitRouteMixin._finish_W_paneled; block-Jacobi route only)
        # (decided once per problem size: later builds find the persistent buffers already allocated)
        key = (Pmax, G, nao, self.max_resident_rows, self.fft_batch, self.pair_space, self.max_device_memory)
        if getattr(self, '_rows_plan', (None,))[0] != key:
            self._rows_plan = (key,) + self._resident_rows(G, Pmax)
        rows_single, rows_panel = self._rows_plan[1:]
        paneled = Pmax > rows_single and not self._want_theta and self.fit_route != 'cholesky'
        rows_buf = max(min(Pmax, rows_panel) if paneled else Pmax, 2 * nao + kmax)
        # spectral form of W (block-Jacobi route): X (P, ldx) takes the place of the rows - about half their size
        spectral = None
        routes = self._fit_routes()                       # (also validates fit_route)
        # (max_resident_rows is the tests' and experiments' handle on the paneled build: it keeps the classic form)
        if self.w_spectral and not self._want_theta and (routes[0] == 'blockjacobi' or (paneled and 'cholesky' != self.fit_route)) \
                and not self._sharded and not self.max_resident_rows and self.c_isdf <= self.w_spectral_max_c:
            plan = self._spectral_plan()
            if plan is not None and -(-Pmax * plan['ldx'] // G) <= rows_panel:
                spectral = dict(ldx=plan['ldx'], fraction=plan['fraction'])
                paneled = False
                rows_buf = max(-(-Pmax * plan['ldx'] // G), 2 * nao + kmax)
        scratch = self._buffer('theta', (rows_buf, G))
        ao_sel = scratch[:nao]
        L = scratch[nao:nao + kmax]
        be.gather_cols(self.ao, d_perm, ao_sel)
        if self.select == 'refined' and self.cand_ao_cutoff:
            ao_sel = self._local_ao_rows(ao_sel, scratch[nao + kmax:], blk_off, a)
        elif self.cand_skip_zero_rows:
            ao_sel = self._nonzero_ao_rows(ao_sel, scratch[nao + kmax:], blk_off)
        piv = be.empty((cell.natm, kmax), dtype=torch.int64)
        rank = be.select_ip(ao_sel, blk_off, nip, self.select_tol, self.tie_rtol, L, piv)
        del ao_sel, L, scratch
        piv_h = be.to_host(piv)
        self._tick('S2_select_candidates' if self.select == 'refined' else 'S2_select_ip', t0)
        self._sel = dict(owner=owner, perm=perm, blk_off=blk_off, nip_final=nip_final, piv_h=piv_h, rank=rank, paneled=paneled,
                         rows_buf=rows_buf, rows_panel=rows_panel, spectral=spectral, rows_single=rows_single)

    def _pick_and_fit(self, orbitals=None):
        """S2 second stage + S3 + S4 + S5 from the candidates of _select_candidates: the final points ('refined': one pivoted
        Cholesky of the candidates' Gram matrix), the fit (block-Jacobi / Cholesky route, paneled when the rows exceed HBM) and W.
        orbitals: None = the AO x AO pair space; (N, nocc) occupied orbital coefficients (already scaled with sqrt(occ)) = the
        (AO x occupied) pair space of that density (pair_space='occ')."""
        cell, be, sel = self.cell, self.backend, self._sel
        nao, G = self.ao.shape
        t0 = time.perf_counter()
        perm, blk_off, piv_h, rank, owner = sel['perm'], sel['blk_off'], sel['piv_h'], sel['rank'], sel['owner']
        paneled, rows_buf, rows_panel = sel['paneled'], sel['rows_buf'], sel['rows_panel']
        self._fit_state = None
        self._W_omega = {}
        self._V = None
        self._psi = self._psiP = None
        if orbitals is not None:
            # psi = C_occ^T phi on the grid (fft_jk.py:235-238 forms the same rows, mo = ao C)
            nocc = orbitals.shape[1]
            self._psi = self._buffer('psi', (nocc, G))
            be.gemm_nn(be.to_device(np.ascontiguousarray(orbitals.T)), self.ao, self._psi)
            t0 = self._tick('S2_occupied_on_grid', t0)
        clusters = self._bj_clusters()
        if self.select == 'refined':
            rank = self._refine_selection(perm, blk_off, piv_h, rank, int(sel['nip_final'].sum()), owner)
            ip = np.concatenate([self._refined_by_atom[b] for cl in clusters for b in cl])
        else:
            ip = np.concatenate([perm[blk_off[b] + piv_h[b, :rank[b]]] for cl in clusters for b in cl])
        self.ip = ip.astype(np.int64)
        P = len(ip)
        t0 = self._tick('S2_select_ip', t0)
        self.aoP = self._buffer('aoP', (P, nao))
        d_ip = be.to_device(self.ip)
        if self._psi is not None:
            self._psiP = self._buffer('psiP', (P, self._psi.shape[0]))
            be.gather_aoP(self._psi, d_ip, self._psiP)
        self.W = self._buffer('W', (P, P))
        self.w_spectral_fraction = None
        if sel.get('spectral') is not None:
            if self._build_spectral(rank, clusters, d_ip, t0):
                self._built = True
                return self
            # the probe check failed: the classic build below has the Cholesky route to fall back to
            t0 = time.perf_counter()
            rows_single, rows_panel = sel['rows_single'], sel['rows_panel']
            paneled = P > rows_single
            rows_buf = max(min(P, rows_panel) if paneled else P, 1)
        if paneled:
            self._build_paneled(rank, clusters, d_ip, rows_buf, rows_panel, t0)
            self._built = True
            return self
        theta = self._buffer('theta', (rows_buf, G))[:P]
        for route in self._fit_routes():
            if route == 'blockjacobi':
                # S3c: no triangular solve over the grid.  theta <- Y' = D^-1 (aoP ao)^2
                ip_off = self._bj_blocks(rank, clusters)
                Afac, Dblk = self._bj_prepare(self.ao, 0, d_ip, ip_off, self.aoP, scratch=self.W)
                self._bj_rows(self.aoP, 0, self.ao, G, Dblk, ip_off, theta)
            else:
                chol = self._buffer('factor', (P, P))
                # forward solve only (Y = Lr^-1 B); the backward solve is applied to the (P, P) matrix below
                self._chol_fit(d_ip, chol, theta, forward_only=not self._want_theta)
            t0 = self._tick('S3_fit', t0)
            if route == 'blockjacobi':
                self._fit_state = dict(kind='blockjacobi', theta=theta, Afac=Afac, Dblk=Dblk, ip_off=ip_off)
            else:
                self._fit_state = dict(kind='explicit' if self._want_theta else 'cholesky', theta=theta, chol=chol)
            self._finish_W(self.W)
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
        self._keep_V_for_robust_k(t0)
        del theta
        self._built = True
        return self

    def _chol_fit(self, d_ip, chol, theta, forward_only):
        """S3b: chol <- Cholesky factor of the regularised Gram matrix of the points, theta <- Lr^-1 (rows) (and Lr^-T of that
        unless forward_only), in the AO x AO or the (AO x occupied) pair space."""
        be = self.backend
        G = self.ao.shape[1]
        if self._psi is None:
            self.reg_used = be.fit_prepare(self.ao, d_ip, self.reg_rel, self.aoP, chol)
            be.fit_apply(chol, self.aoP, self.ao, G, theta, forward_only=forward_only)
            return
        be.gather_aoP(self.ao, d_ip, self.aoP)
        be.gram_prod(self.aoP, self._psiP, chol)
        be.shift_diag(chol, self.reg_rel)
        self.reg_used = self.reg_rel + be.chol_inplace(chol, 0.0, scratch=self.W)
        be.pair_prod_rows(self.aoP, self._psiP, self.ao, self._psi, G, theta)
        be.factor_solve_half(chol, False, theta)
        if not forward_only:
            be.factor_solve_half(chol, True, theta)

    # ---- pair_space='occ': the fit follows the density ----------------------------------------------------------
    def _occupied_orbitals(self, dm):
        """(N, nocc) coefficients C_occ sqrt(occ) of the density get_jk was called with - from its mo_coeff / mo_occ tag
        exactly as the reference's K does (pyscf/pbc/df/fft_jk.py:206-210), else from the eigenvectors of a symmetric positive
        semidefinite D of rank <= N/2 (several density matrices: their orbitals side by side).  None when the density has no
        such form (a difference or response density, full rank): the AO x AO pair space is used then."""
        nao = self.cell.nao_nr()
        mo_coeff, mo_occ = getattr(dm, 'mo_coeff', None), getattr(dm, 'mo_occ', None)
        cols = []
        if mo_coeff is not None and mo_occ is not None:
            mo_coeff, mo_occ = np.asarray(mo_coeff), np.asarray(mo_occ)
            if np.iscomplexobj(mo_coeff) and abs(mo_coeff.imag).max() > 1e-12:
                return None
            mo_coeff = mo_coeff.real.reshape(-1, nao, mo_coeff.shape[-1])
            mo_occ = np.asarray(mo_occ, dtype=float).reshape(len(mo_coeff), -1)
            for c, occ in zip(mo_coeff, mo_occ):
                cols.append(c[:, occ > 0] * np.sqrt(occ[occ > 0]))
        else:
            dms = np.asarray(dm)
            if np.iscomplexobj(dms):
                if abs(dms.imag).max() > 1e-12:
                    return None
                dms = dms.real
            for d in dms.reshape(-1, nao, nao):
                if abs(d - d.T).max() > 1e-10 * max(abs(d).max(), 1e-300):
                    return None
                ev, u = np.linalg.eigh(0.5 * (d + d.T))
                if ev.min() < -1e-10 * abs(ev).max():
                    return None
                keep = ev > 1e-12 * ev.max()
                cols.append(u[:, keep] * np.sqrt(ev[keep]))
        c = np.ascontiguousarray(np.hstack(cols), dtype=np.float64)
        if c.shape[1] == 0 or c.shape[1] > nao // 2:
            return None
        return c

    def _ensure_fit(self, dm=None):
        """pair_space='occ': make sure the fit (final points, rows, W) suits what is about to be contracted with W.
        dm with an occupied-orbital form (MO tag, or positive semidefinite low rank): the (AO x occupied) pair space of THAT
        density - refit when the occupied space differs from the fitted one (occ_refit='always') or keep the first such fit
        ('once').  dm = None (AO integrals: get_ao_eri / ao2mo) or a density without such a form (a response or difference
        density): the AO x AO pair space - an (AO x occupied) fit does not represent those pairs, so it is replaced."""
        if self.pair_space != 'occ' or self._sel is None or self.ao is None:
            return
        orb = None if dm is None else self._occupied_orbitals(dm)
        want = None if orb is None else orb.dot(orb.T)
        if not self._fit_pending:
            have = self._fit_dm
            if want is None:
                if have is None:
                    return                                            # the AO-pair fit is in place
            elif have is None:
                if self.occ_refit == 'once':
                    return                                            # an AO-pair fit came first and is kept
            elif self.occ_refit == 'once' or abs(want - have).max() <= 1e-12 * abs(want).max():
                return
        self._fit_dm = want
        if self._sel.get('sharded'):
            self._pick_and_fit_sharded(orb)
        else:
            self._pick_and_fit(orb)
        self._fit_pending = False

    def _build_spectral(self, rank, clusters, d_ip, t0):
        """S3c + S4 + S5 in the spectral form (fit_route.FitRouteMixin._finish_W_spectral): block-Jacobi route, the probe check
        alongside.  Returns False when the check fails (the caller then runs the classic build, which can fall back to the
        Cholesky route)."""
        be = self.backend
        P = len(self.ip)
        nao, G = self.ao.shape
        ip_off = self._bj_blocks(rank, clusters)
        Afac, Dblk = self._bj_prepare(self.ao, 0, d_ip, ip_off, self.aoP, scratch=self.W)
        self._fit_state = dict(kind='blockjacobi-spectral', Afac=Afac, Dblk=Dblk, ip_off=ip_off)
        t0 = self._tick('S3_fit', t0)
        probe = None
        if self.fit_route == 'auto':
            aoT = be.empty((nao, P))
            be.gather_cols(self.ao, d_ip, aoT)
            T0, E = self._bj_probe_vectors(aoT, Afac, Dblk, ip_off)
            del aoT
            probe = (E, be.empty((E.shape[0], G)))
        self._finish_W_spectral(self.W, probe=probe)
        t0 = self._tick('S4S5_coulomb_W', t0)
        self.fit_route_used = 'blockjacobi'
        self.n_panels = 1
        if probe is None:
            return True
        self.bj_check = self._bj_probe_energies(T0, probe[1], self.W, None)
        self._tick('S5_route_check', t0)
        tol = min(self.bj_check_tol, self.w_spectral_check_tol)
        if self.bj_check <= tol:
            return True
        warnings.warn('ISDF: block-Jacobi fit route failed its probe check (mismatch %.2e > %.2e) in the spectral build; '
                      'rebuilding W the classic way' % (self.bj_check, tol))
        self.w_spectral_fraction = None
        return False

    def _build_paneled(self, rank, clusters, d_ip, rows_buf, rows_panel, t0):
        """S3c + S4 + S5 with the fit rows produced panel by panel (more points than HBM holds rows for): block-Jacobi
        route only - its rows depend on their own preconditioner block alone, so a panel can be recomputed at will.  The
        probe check runs alongside (its combination rows are accumulated panel by panel).  There is no Cholesky route to fall
        back to here (its rows depend on all earlier rows); a failed check rebuilds ONCE with pairs of clusters merged into
        one preconditioner block (a better-conditioned A', twice the block-solve flops) and raises if that fails too."""
        be = self.backend
        P = len(self.ip)
        nao, G = self.ao.shape
        group0 = int(self.bj_group)
        try:
            for attempt in range(2):
                ip_off = self._bj_blocks(rank, clusters)
                Afac, Dblk = self._bj_prepare(self.ao, 0, d_ip, ip_off, self.aoP, scratch=self.W)
                panels = self._panel_plan(ip_off, min(rows_panel, rows_buf))
                rows = self._buffer('theta', (rows_buf, G))
                self._fit_state = dict(kind='blockjacobi-paneled', rows=rows, panels=panels, Afac=Afac, Dblk=Dblk, ip_off=ip_off)
                t0 = self._tick('S3_fit', t0)
                probe = None
                if self.fit_route == 'auto':
                    aoT = be.empty((nao, P))
                    be.gather_cols(self.ao, d_ip, aoT)
                    T0, E = self._bj_probe_vectors(aoT, Afac, Dblk, ip_off)
                    del aoT
                    probe = (E, be.empty((E.shape[0], G)))
                self._finish_W_paneled(self.W, probe=probe)
                t0 = self._tick('S4S5_coulomb_W', t0)
                self.fit_route_used = 'blockjacobi'
                self.n_panels = len(panels)
                if probe is None:
                    return
                self.bj_check = self._bj_probe_energies(T0, probe[1], self.W, None)
                t0 = self._tick('S5_route_check', t0)
                if self.bj_check <= self.bj_check_tol:
                    return
                if attempt == 0:
                    warnings.warn('ISDF: block-Jacobi fit route failed its probe check (mismatch %.2e > %.2e) in a paneled build, where '
                                  'the Cholesky route is not available: rebuilding with pairs of clusters merged into one '
                                  'preconditioner block' % (self.bj_check, self.bj_check_tol))
                    self.bj_group = 2 * group0
            raise RuntimeError('ISDF: the paneled block-Jacobi build failed its probe check twice (mismatch %.2e > bj_check_tol = '
                               '%.2e): W would carry rounding noise of that relative size.  Use fewer points (c_isdf), more GPUs '
                               '(the rows then fit and the Cholesky route is available), or raise bj_check_tol knowingly.'
                               % (self.bj_check, self.bj_check_tol))
        finally:
            self.bj_group = group0

    def _local_ao_rows(self, ao_sel, out_rows, blk_off, a):
        """Candidate stage with local AOs: for the block of atom b keep only the AO rows of atoms within cand_ao_cutoff of b
        (minimum image), packed to the front and zero-padded to a common row count - the selection kernels stream
        8 (rows + j) m bytes per pivot, and far AOs contribute next to nothing to a block's pair-density Gram matrix.
        ao_sel: (nao, G) block-major; returns a (nloc_max, G) view of out_rows."""
        cell = self.cell
        aosl = _aoslice_by_atom(cell)
        frac = np.asarray(cell.atom_coords(), dtype=float).dot(np.linalg.inv(a))
        d = frac[:, None, :] - frac[None, :, :]
        d -= np.round(d)
        near = np.linalg.norm(d.dot(a), axis=2) < float(self.cand_ao_cutoff)
        lists = [np.concatenate([np.arange(aosl[c, 0], aosl[c, 1]) for c in np.nonzero(near[b])[0]]) for b in range(cell.natm)]
        nloc = max(len(x) for x in lists)
        loc = out_rows[:nloc]
        loc.zero_()
        for b, rows in enumerate(lists):
            s0, s1 = int(blk_off[b]), int(blk_off[b + 1])
            if s1 > s0:
                loc[:len(rows), s0:s1] = ao_sel[self.backend.to_device(rows.astype(np.int64)), s0:s1]
        return loc

    def _nonzero_ao_rows(self, ao_sel, out_rows, blk_off):
        """Per block of grid points keep only the AO rows that are not identically zero there (packed to the front, zero-padded to
        a common row count, relative order kept): the collocation truncates every shell at its rcut, so for a cell larger than the
        reach of its AOs a large part of every block's rows are exact zeros - terms fma(0, 0, s) = s of the selection's dot products.
        ao_sel: (nao, G) block-major; returns a (nloc_max, G) view of out_rows, or ao_sel itself when little would be saved."""
        be = self.backend
        natm = len(blk_off) - 1
        mx = be.block_row_absmax(ao_sel, blk_off)
        lists = [np.nonzero(mx[:, b] > 0.0)[0] for b in range(natm)]
        nloc = max([len(x) for x in lists] + [1])
        self._cand_rows_kept = (nloc, ao_sel.shape[0])
        if nloc > 0.95 * ao_sel.shape[0]:
            return ao_sel
        loc = out_rows[:nloc]
        loc.zero_()
        for b, rows in enumerate(lists):
            s0, s1 = int(blk_off[b]), int(blk_off[b + 1])
            if s1 > s0 and len(rows):
                loc[:len(rows), s0:s1] = ao_sel[be.to_device(rows.astype(np.int64)), s0:s1]
        return loc

    def _refine_selection(self, perm, blk_off, piv_h, rank, P_target, owner):
        """select='refined' on one GPU: the per-atom selections (refine_over x too many points each) are only CANDIDATES;
        see _refine_pick.  Leaves the chosen grid indices per atom (in pivot order) in self._refined_by_atom and returns the
        points per atom."""
        be = self.backend
        natm = self.cell.natm
        cand = np.concatenate([perm[blk_off[b] + piv_h[b, :rank[b]]] for b in range(natm)]).astype(np.int64)
        aoC = be.empty((len(cand), self.ao.shape[0]))
        d_cand = be.to_device(cand)
        be.gather_aoP(self.ao, d_cand, aoC)
        psiC = None
        if self._psi is not None:                 # (AO x occupied) pair space: the candidates' Gram matrix is a product
            psiC = be.empty((len(cand), self._psi.shape[0]))
            be.gather_aoP(self._psi, d_cand, psiC)
        # the candidate Gram matrix (17 GB at c = 14) lives in the fit-row buffer, which is not in use yet
        rows_buf = self._bufs.get('theta')
        gram = None if rows_buf is None or rows_buf.numel() < len(cand) ** 2 else rows_buf[:len(cand) ** 2].view(len(cand), len(cand))
        chosen = self._refine_pick(aoC, cand, P_target, gram=gram, psiC=psiC)
        del aoC, gram, psiC
        own = owner[chosen]
        self._refined_by_atom = [chosen[own == b] for b in range(natm)]
        return np.array([len(x) for x in self._refined_by_atom], dtype=np.int32)

    def _refine_pick(self, aoC, cand, P_target, gram=None, nh=0, psiC=None):
        """One pivoted Cholesky of the pair-density Gram matrix restricted to the candidate set (aoC: AO values at the
        candidates, (m, nao); isdf_gram_sq + isdf_select_ip_gram, pivot rule pyscf/lib/scipy_helper.py:71-110) picks the
        final P_target points.  Returns their grid indices in pivot order."""
        be = self.backend
        m = len(cand)
        P_target = min(int(P_target), m)
        A = be.empty((m, m)) if gram is None else gram
        if psiC is not None:
            be.gram_prod(aoC, psiC, A)
        else:
            be.gram_sq(aoC, A, nh)                  # nh > 0: k-point (complex) mode, aoC = [Re u | Im u] at the candidates
        piv2 = be.empty((P_target,), dtype=torch.int64)
        r2 = be.select_ip_gram(A, P_target, self.select_tol, self.tie_rtol, piv2)
        chosen = np.asarray(cand)[be.to_host(piv2)[:r2]]
        del A, piv2
        return chosen

    def _keep_V_for_robust_k(self, t0):
        """robust_k: the fit buffer (Theta) becomes V = conv(Theta), in place (one more pass of batched FFTs)."""
        if not self.robust_k:
            return
        be = self.backend
        theta = self._fit_state['theta']
        mesh = np.asarray(self.mesh, dtype=np.int32)
        a = np.asarray(self.cell.lattice_vectors(), dtype=float)
        be.coulomb_rows(theta, mesh, a, self._last_fft_batch)
        self._V = theta
        self._fit_state = None                   # Theta is gone: no range-separated rebuild from this fit
        self._tick('S5_conv_for_robust_k', t0)

    def range_coulomb(self, omega):
        """FFTDF.range_coulomb (pyscf/pbc/df/fft.py:337-359): a context manager whose object answers get_jk / get_ao_eri-less
        calls with the range-separated kernel.  Here it is a view of this object that adds ``omega`` to get_jk."""
        import contextlib
        parent = self

        class _RangeSeparatedView:
            def __getattr__(self, name):
                return getattr(parent, name)

            def get_jk(self, dm, hermi=1, kpts=None, kpts_band=None, with_j=True, with_k=True, omega=None, exxdiv=None):
                return parent.get_jk(dm, hermi, kpts, kpts_band, with_j, with_k, omega if omega is not None else omega_, exxdiv)
        omega_ = omega

        @contextlib.contextmanager
        def ctx():
            yield _RangeSeparatedView()
        return ctx()

    def to_gpu(self):
        """FFTDF.to_gpu (pyscf/pbc/df/fft.py): this object already runs on the GPU."""
        return self

    def _robust_k_correction(self, d_dm, d_vk):
        """K <- K1 + K2 - K_isdf (Dunlap's robust form of the fitted exchange: error quadratic in the fit error),
        K1_mn = sum_P phi_m(P) sum_g w V_P(g) [phi_P D phi(g)] phi_n(g),  V_P = conv(Theta_P) (kept in the fit buffer by
        build() when robust_k is set),  K2 = K1(D^T)^T.  Two N x G x P products per density matrix, batched over P."""
        be, comm = self.backend, self.comm
        V = self._V
        if V is None:
            raise RuntimeError('robust_k needs a build with robust_k=True (Theta explicit, V = conv(Theta) kept)')
        P, ng = V.shape                                                     # ng = this rank's grid columns (all of them on one GPU)
        nao = self.cell.nao_nr()
        w = self.cell.vol / int(np.prod(self.mesh))
        sharded = self._sharded
        nb = max(1, min(P, int((6 << 30) // (8 * ng))))
        aoPT = self.aoP.T.contiguous()                                      # (N, P)
        for s in range(d_dm.shape[0]):
            D = d_dm[s]
            passes = [D] if bool(torch.allclose(D, D.T, rtol=0, atol=1e-13 * float(D.abs().max()) + 1e-300)) else [D, D.T.contiguous()]
            ks = []
            for Dp in passes:
                # [phi_P D phi(g)] = sum_i s_i psi_i(P) psi_i(g) through the eigenvectors of a symmetric D when its rank is low
                # (an SCF density: nocc << N) - the first of the two big products then costs 2 P r G instead of 2 P N G flop
                left, right = None, self.ao
                if len(passes) == 1:
                    ev, U = torch.linalg.eigh(Dp)
                    keep = ev.abs() > 1e-12 * float(ev.abs().max())
                    r = int(keep.sum())
                    if 0 < r <= nao // 2:
                        Ur = U[:, keep].contiguous()                         # (N, r)
                        left = be.empty((P, r))
                        be.gemm_nn(self.aoP, (Ur * ev[keep]).contiguous(), left)     # phi_P U s
                        right = be.empty((r, ng))
                        be.gemm_nn(Ur.T.contiguous(), self.ao, right)        # psi = U^T phi on the local columns
                if left is None:
                    left = be.empty((P, nao))
                    be.gemm_nn(self.aoP, Dp, left)                          # phi_P D
                K1 = be.zeros((nao, nao))
                for r0 in range(0, P, nb):
                    r1 = min(P, r0 + nb)
                    F = be.empty((r1 - r0, ng))
                    be.gemm_nn(left[r0:r1], right, F)                       # [phi_P D phi](P, g) on the local columns
                    be.hadamard_rows(F, V[r0:r1])
                    Kt = be.empty((r1 - r0, nao))
                    be.gemm_nt(F, self.ao, Kt, alpha=w)                     # sum_g w (.) phi_n(g)
                    be.gemm_nn(aoPT[:, r0:r1], Kt, K1, beta=1.0)            # sum_P phi_m(P) (.)
                    del F, Kt
                del left, right
                if sharded:
                    comm.all_reduce_sum(K1)                                 # the grid sum was over this rank's slice
                ks.append(K1)
            k2 = ks[0].T if len(ks) == 1 else ks[1].T
            d_vk[s] = ks[0] + k2 - d_vk[s]

    def _finish_W(self, W):
        """S4 + S5 for the fit held in self._fit_state (rows Y / Y' / Theta in the fit buffer + the factors that go with
        them): W <- w conv(rows) rows^T, then the route's P x P finishing.  Uses the Coulomb kernel the backend is set to
        (plain, or range-separated for get_jk(omega=...): the fit itself does not depend on the kernel)."""
        be, st = self.backend, self._fit_state
        if st['kind'] == 'blockjacobi-paneled':
            return self._finish_W_paneled(W)
        if st['kind'] == 'blockjacobi-spectral':
            return self._finish_W_spectral(W)
        theta = st['theta']
        P, G = theta.shape
        mesh = np.asarray(self.mesh, dtype=np.int32)
        a = np.asarray(self.cell.lattice_vectors(), dtype=float)
        batch = self.fft_batch or _default_fft_batch(G, P, be.free_bytes())
        self._last_fft_batch = batch
        be.coulomb_W(theta, mesh, a, 0, P, batch, W, upper_only=True)
        be.symmetrize_upper(W)
        if st['kind'] == 'blockjacobi':
            self._bj_finish(st['Afac'], st['Dblk'], st['ip_off'], W)
        elif st['kind'] == 'cholesky':
            be.W_from_factor(st['chol'], 0, W)
        elif st['kind'] == 'selection':
            be.W_from_factor(st['T'], 1, W)

    def _vcut_sph_radius(self, nk):
        """Rc of exxdiv='vcut_sph': the sphere with the volume of the nk-fold cell (pbc.py:313)."""
        return float((3.0 * nk * self.cell.vol / (4.0 * np.pi)) ** (1.0 / 3.0))

    def _ws_kernel(self, nk):
        """Table of the Wigner-Seitz truncated kernel for an nk k-mesh on this cell (pbc_tools.wigner_seitz_kernel), cached."""
        from . import pbc_tools
        key = tuple(int(x) for x in nk)
        cache = getattr(self, '_ws_tables', None)
        if cache is None or cache[0] is not self.cell:
            cache = self._ws_tables = (self.cell, {})
        if key not in cache[1]:
            cache[1][key] = pbc_tools.wigner_seitz_kernel(self.cell.lattice_vectors(), key)
        return cache[1][key]

    def _W_kernel_variant(self, key, omega=None, cutoff=None, ws=None):
        """W rebuilt from the current fit with another Coulomb kernel (range-separated: omega; spherically truncated:
        cutoff; Wigner-Seitz truncated: ws), cached under ``key`` until the next build.  The fit does not depend on the kernel."""
        be = self.backend
        if key not in self._W_omega:
            t0 = time.perf_counter()
            be.set_coulomb_omega(omega or 0.0)
            be.set_coulomb_cutoff(cutoff or 0.0)
            be.set_coulomb_ws(ws)
            try:
                W = be.empty(tuple(self.W.shape))
                (self._finish_W_sharded if self._fit_state.get('sharded') else self._finish_W)(W)
            finally:
                be.set_coulomb_omega(0.0)
                be.set_coulomb_cutoff(0.0)
                be.set_coulomb_ws(None)
            self._W_omega[key] = W
            self._tick('S4S5_coulomb_W_variant', t0)
        return self._W_omega[key]

    def _get_jk_omega(self, dm, hermi, kpts, kpts_band, with_j, with_k, omega, exxdiv):
        """Range-separated J/K (FFTDF.get_jk(omega=...), pyscf/pbc/df/fft.py:298-303): the Coulomb kernel carries
        exp(-G^2/4 omega^2) (omega > 0, long range) or 1 - exp(...) (omega < 0, short range) as pbc.py:408-418 has it.
        The fit does not depend on the kernel: only S4/S5 are redone, once per omega (cached until the next build)."""
        ex = exxdiv if exxdiv is not None else self.exxdiv
        if ex not in (None, 'None'):
            raise NotImplementedError('range-separated J/K: only exxdiv=None is implemented')
        be = self.backend
        if self.robust_k:
            raise NotImplementedError('range-separated J/K with robust_k is not implemented')
        if not self._built:
            self.build()
        if with_k and self.pair_space == 'occ':
            self._ensure_fit(dm)
        key = round(float(omega), 10)
        be.set_coulomb_omega(omega)
        try:
            if key not in self._W_omega:
                t0 = time.perf_counter()
                W = be.empty(tuple(self.W.shape))
                (self._finish_W_sharded if self._fit_state.get('sharded') else self._finish_W)(W)
                self._W_omega[key] = W
                self._tick('S4S5_coulomb_W_omega', t0)
            W_plain, self.W = self.W, self._W_omega[key]
            try:
                return self.get_jk(dm, hermi, kpts, kpts_band, with_j, with_k, None, exxdiv)
            finally:
                self.W = W_plain
        finally:
            be.set_coulomb_omega(0.0)

    # ---- J / K ---------------------------------------------------------------------------------
    def get_jk(self, dm, hermi=1, kpts=None, kpts_band=None, with_j=True, with_k=True, omega=None,
               exxdiv=None):
        if kpts is None:
            kpts = self.kpts_symm if self.kpts_symm is not None else self.kpts
        if hasattr(kpts, 'kpts_ibz') and hasattr(kpts, 'transform_dm'):
            # k-point symmetry (pyscf/pbc/scf/khf_ksymm.py:210-237): density matrices on the irreducible k-points, rotated to
            # the full zone on the host; J and K on the irreducible k-points (or on kpts_band when given)
            ndm = np.asarray(dm).shape[-3] if np.asarray(dm).ndim >= 3 else 1
            if ndm != kpts.nkpts_ibz:
                raise RuntimeError('number of input density matrices does not match the number of irreducible k-points: '
                                   '%d vs %d' % (ndm, kpts.nkpts_ibz))
            dm_bz = kpts.transform_dm(dm)
            band = kpts.kpts_ibz if kpts_band is None else kpts_band
            return self.get_jk(dm_bz, hermi, kpts.kpts, band, with_j, with_k, omega, exxdiv)
        if omega is not None and abs(omega) > 0:
            if not self._is_gamma(kpts) or not self._is_gamma(self.kpts) or not self._is_gamma(kpts_band):
                if self._is_gamma(kpts) and np.asarray(dm).ndim == 2:
                    dm = np.asarray(dm)[None]
                return self._get_jk_kpts(dm, hermi, kpts, kpts_band, with_j, with_k, exxdiv, omega=omega)
            return self._get_jk_omega(dm, hermi, kpts, kpts_band, with_j, with_k, omega, exxdiv)
        if not self._is_gamma(kpts) or not self._is_gamma(self.kpts) or not self._is_gamma(kpts_band):
            if self._is_gamma(kpts) and np.asarray(dm).ndim == 2:
                dm = np.asarray(dm)[None]                       # Gamma-point density, band structure requested
            return self._get_jk_kpts(dm, hermi, kpts, kpts_band, with_j, with_k, exxdiv)
        if self.kpts_band is not None:                          # back from a band calculation: Gamma-only build again
            self.kpts_band = None
            self._built = False
        if exxdiv is None:
            exxdiv = self.exxdiv
        if exxdiv not in (None, 'None', 'ewald', 'vcut_sph', 'vcut_ws'):
            raise NotImplementedError("exxdiv=%r: None, 'ewald', 'vcut_sph' and 'vcut_ws' are implemented" % (exxdiv,))
        if not self._built:
            self.build()
        if with_k and self.pair_space == 'occ':
            self._ensure_fit(dm)
        be = self.backend
        if exxdiv in ('vcut_sph', 'vcut_ws') and with_k:
            # K with a truncated kernel (spherical: pbc.py:312-317; Wigner-Seitz: pbc.py:318-346): its own W, built once from
            # the same fit; J keeps 1/r
            if self.robust_k:
                raise NotImplementedError("exxdiv=%r with robust_k is not implemented" % exxdiv)
            vj = self.get_jk(dm, hermi, kpts, kpts_band, True, False, None, 'None')[0] if with_j else None
            if exxdiv == 'vcut_sph':
                Wv = self._W_kernel_variant('vcut_sph', cutoff=self._vcut_sph_radius(1))
            else:
                Wv = self._W_kernel_variant('vcut_ws', ws=self._ws_kernel((1, 1, 1)))
            W_plain, self.W = self.W, Wv
            try:
                vk = self.get_jk(dm, hermi, kpts, kpts_band, False, True, None, 'None')[1]
            finally:
                self.W = W_plain
            return vj, vk
        if exxdiv in ('vcut_sph', 'vcut_ws'):
            exxdiv = None
        dm_in = np.asarray(dm)
        if np.iscomplexobj(dm_in):
            if abs(dm_in.imag).max() > 1e-12:
                # J and K are linear in D and the Gamma-point AOs are real: real and imaginary parts separately
                re = self.get_jk(dm_in.real, hermi, kpts, kpts_band, with_j, with_k, omega, exxdiv)
                im = self.get_jk(dm_in.imag, hermi, kpts, kpts_band, with_j, with_k, omega, exxdiv)
                return tuple(None if r is None else r + 1j * i for r, i in zip(re, im))
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
        if self._sharded:
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
            if self.robust_k:
                self._robust_k_correction(d_dm, d_vk)
            if exxdiv == 'ewald':
                self._add_ewald_exxdiv(d_dm, d_vk)
            t0 = self._tick('S7_get_k', t0)
            vk = be.to_host(d_vk).reshape(dm_in.shape)
        return vj, vk

    def get_k_exact(self, dm=None, mo_coeff=None, mo_occ=None, max_rows=None, kpts_band=None, rows=None):
        """The reference's exact exchange (FFTDF.get_jk's K, fft_jk.py:177-302) evaluated on the GPU with
        the same device primitives — N*nocc FFT pairs.  Used to measure the ISDF fitting error at full
        size.  Needs the occupied orbitals (mo_coeff, mo_occ) or a positive semidefinite dm."""
        if not self._is_gamma(self.kpts) or not self._is_gamma(kpts_band):
            if self._sharded:
                raise NotImplementedError('the exact k-point exchange is a single-process verification path')
            return self._get_k_exact_kpts(dm, mo_coeff, mo_occ, kpts_band=kpts_band, rows=rows, max_rows=max_rows)
        be, comm = self.backend, self.comm
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
        if self._sharded:
            # several ranks: every rank collocates the WHOLE grid (the FFTs are over all of it) and takes a share of the AO rows
            # of K; all-reduce of the zero-padded result.  Lets bench.py --gpus N report the accuracy entry too.
            cell = self.cell
            rcut = gto.estimate_rcut_per_shell(cell)
            Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
            ao = be.empty((nao, G))
            be.eval_ao(np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut,
                       be.to_device(np.ascontiguousarray(self.grids.coords.T)), ao)
            i0, i1 = comm.split_range(nao)
            vk = be.zeros((nao, nao))
            if i1 > i0:
                be.get_k_exact(ao, G, be.to_device(np.ascontiguousarray(c)), mesh, a, i0, i1 - i0, max_rows, vk)
            del ao
            comm.all_reduce_sum(vk)
            return be.to_host(vk)
        if self.ao is None:
            self.collocate()
        vk = be.empty((nao, nao))
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
            if self._sharded:
                self.comm.all_reduce_sum(S)
            self._ovlp = S
        return self._ovlp

    def _add_ewald_exxdiv(self, d_dm, d_vk):
        """vk += madelung * S D S  (pyscf/pbc/df/df_jk.py:1446-1452, Gamma point)."""
        S = self.overlap()
        mad = gto.madelung(self.cell)
        for i in range(d_dm.shape[0]):
            d_vk[i] += mad * (S @ d_dm[i] @ S)       # N^3, negligible; torch matmul as plumbing


    # ---- ERIs from the factorisation (small systems; reached from SCF.get_jk's incore branch,
    #      pyscf/pbc/scf/hf.py:670-679) ------------------------------------------------------------
    def get_ao_eri(self, kpts=None, compact=True):
        """AO ERIs from the factorisation (FFTDF.get_ao_eri surface, pyscf/pbc/df/fft.py:317).  Gamma point: real, s4-compact
        (nao(nao+1)/2 squared) or s1; with k-points (one, or four that conserve momentum): complex s1, see kpoints.py."""
        if not self._is_gamma(kpts) or not self._is_gamma(self.kpts):
            return self._get_ao_eri_kpts(self.kpts if kpts is None else kpts)
        if not self._built:
            self.build()
        self._ensure_fit()                        # pair_space='occ' and no fit yet: integrals come from the AO x AO pair space
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
        sum_PQ X_ij,P W_PQ X_kl,Q with X_ij,P = (aoP C_i)_P (aoP C_j)_P.  s1 layout; Gamma point (real) or k-points (kpoints.py)."""
        if isinstance(mo_coeffs, np.ndarray) and mo_coeffs.ndim == 2:
            mo_coeffs = (mo_coeffs,) * 4
        if not self._is_gamma(kpts) or not self._is_gamma(self.kpts):
            if compact:
                raise NotImplementedError('k-point MO integrals have no compact form')
            return self._get_ao_eri_kpts(self.kpts if kpts is None else kpts, mo_coeffs=mo_coeffs)
        if compact:
            raise NotImplementedError('compact MO integrals are not implemented; use compact=False')
        if not self._built:
            self.build()
        self._ensure_fit()
        aoP = self.backend.to_host(self.aoP)
        W = self.backend.to_host(self.W)
        ci, cj, ck, cl = [aoP.dot(np.asarray(c)) for c in mo_coeffs]
        Xij = np.einsum('pi,pj->pij', ci, cj).reshape(len(aoP), -1)
        Xkl = np.einsum('pk,pl->pkl', ck, cl).reshape(len(aoP), -1)
        return Xij.T.dot(W).dot(Xkl)

    get_mo_eri = ao2mo


