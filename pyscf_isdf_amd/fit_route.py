"""The fit routes of the ISDF build: block-Jacobi helpers (S3c), route selection, and the
probe-density check that guards the block-Jacobi route.  Mixed into ``isdf.ISDF``; host orchestration only."""
import numpy as np


class FitRouteMixin:
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

    def _fit_routes(self):
        if self.fit_route not in ('auto', 'blockjacobi', 'cholesky'):
            raise ValueError("fit_route must be 'auto', 'blockjacobi' or 'cholesky'")
        if self._want_theta or self.fit_route == 'cholesky':
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
