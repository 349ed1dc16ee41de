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
        if getattr(self, '_psi', None) is not None and not nh:
            be.gram_prod(aoP, self._psiP, A)             # (AO x occupied) pair space
        else:
            be.gram_sq(aoP, A, nh)
        self._Dinv_key = None                            # the block factors are about to be rewritten (a refit in the same buffers)
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

    def _block_inverse(self, Dblk, ip_off):
        """blockdiag(D_b^-1) for the MFMA form of the block solves over the grid (isdf_block_apply); formed once per build
        from the factors every rank holds.  None when block_apply_mfma is off (then: substitution, isdf_block_solve)."""
        if not self.block_apply_mfma or int(np.diff(np.asarray(ip_off)).max()) > 1100:
            return None                                  # (the MFMA kernel stages a block's rows in LDS: 160 KB hold 1137 of them)
        key = (Dblk.data_ptr(), getattr(self, '_build_serial', 0), len(ip_off))
        if getattr(self, '_Dinv_key', None) != key:
            Dinv = self._buffer('Dinv', tuple(Dblk.shape))
            self.backend.block_invert(Dblk, ip_off, Dinv)
            self._Dinv, self._Dinv_key = Dinv, key
        return self._Dinv

    def _bj_rows(self, aoP, nh, ao, ng, Dblk, ip_off, out):
        """out (P, ng) <- Y' = D^-1 (aoP ao)^2 on ng grid columns."""
        be = self.backend
        Dinv = self._block_inverse(Dblk, ip_off)
        if getattr(self, '_psi', None) is not None and not nh:
            be.pair_prod_rows(aoP, self._psiP, ao, self._psi, ng, out)      # (AO x occupied) pair space
            if Dinv is not None:
                be.block_apply(Dinv, ip_off, out)
            else:
                be.block_solve(Dblk, ip_off, 0, 0, out)
            return
        if Dinv is not None and not nh:
            be.pair_rows_block_apply(aoP, ao, ng, Dinv, ip_off, out)        # the square rides the block apply's staging
            return
        be.pair_gram_rows(aoP, ao, ng, out, nh)
        if Dinv is not None:
            be.block_apply(Dinv, ip_off, out)
        else:
            be.block_solve(Dblk, ip_off, 0, 0, out)

    # ---- paneled S3c/S4/S5: more fit rows than HBM holds at once -------------------------------------------------
    def _resident_rows(self, G, P):
        """(rows_single, rows_panel): how many (G-long) fit rows can stay resident next to what a build of P points still has
        to allocate - W, the factor of A', the block factors and their inverses (4 x 8 P^2 bytes), FFT batches with their half spectra and
        work areas (24 G bytes per row of the batch), the GEMM's slab buffers.  rows_single assumes the smallest FFT batch
        the single-pass build would settle for (128 rows); rows_panel the paneled build's 512-row batches plus its
        recompute scratch.  ``max_resident_rows`` overrides both (tests, experiments)."""
        if self.max_resident_rows:
            return int(self.max_resident_rows), int(self.max_resident_rows)
        be = self.backend
        have = be.free_bytes() + sum(int(b.numel()) * 8 for k, b in self._bufs.items() if k in ('theta', 'W', 'factor', 'Dblk', 'Dinv', 'rows_scratch'))
        if self.max_device_memory:
            # the caller's cap on what the fit may occupy (the role max_memory plays for the reference's grid blocking,
            # pyscf/pbc/dft/numint.py:1236-1257, fft_jk.py:240-244: fewer rows at once, same result) - minus what phi already holds
            have = min(have, int(self.max_device_memory) - sum(int(b.numel()) * 8 for k, b in self._bufs.items() if k in ('ao', 'psi')))
        fixed = 4 * 8 * P * P + (3 << 30)
        if getattr(self, 'pair_space', 'ao') == 'occ':
            # occupied orbitals on the grid (nocc <= N/2 rows; the electron count is the usual case) + the product scratch
            nocc = min(self.cell.nao_nr() // 2, int(getattr(self.cell, 'nelectron', 0)) // 2 + 8)
            fixed += 8 * G * nocc + (2 << 30)
        fb = int(self.fft_batch or 0)
        single = (have - fixed - 24 * (min(128, fb) if fb else 128) * G) // (8 * G)
        panel = (have - fixed - 40 * (fb or 512) * G) // (8 * G)
        return max(0, int(single)), max(0, int(panel))

    def _panel_plan(self, ip_off, nrows_max):
        """Split the point blocks (offsets ip_off) into consecutive panels of at most nrows_max rows, as equal as the
        block boundaries allow.  Returns [(r0, r1), ...]."""
        ip_off = np.asarray(ip_off, dtype=np.int64)
        P = int(ip_off[-1])
        big = int(np.diff(ip_off).max())
        if nrows_max < big:
            raise MemoryError('ISDF: not enough device memory for one block of %d fit rows' % big)
        # the LAST panel is made as large as memory allows: rows of earlier panels are the ones that get recomputed
        # (panel p is recomputed once for every later panel), so the early panels should be the small ones
        npan = max(1, -(-P // int(nrows_max)))
        while True:
            cuts = [P]
            for k in range(npan - 1):
                want = max(cuts[-1] - int(nrows_max), 0)
                # a block boundary at or above `want` (the panel above it then has at most nrows_max rows); among the next
                # few, the one that leaves the rows below it closest to a multiple of 256 from below: they are recomputed in
                # 512-row batches and the last, partial batch should still fill the GEMM's 256-row tiles
                i0 = int(np.searchsorted(ip_off, want, side='left'))
                cand = [int(x) for x in ip_off[i0:] if x < cuts[-1] and x < want + 768]
                if not cand:
                    break
                b = min(cand, key=lambda x: ((-x) % 256, x)) if want > 0 else 0
                cuts.append(b)
            cuts.append(0)
            cuts = sorted(set(cuts))
            panels = [(x, y) for x, y in zip(cuts[:-1], cuts[1:]) if y > x]
            if all(y - x <= nrows_max for x, y in panels):
                return panels
            npan += 1

    def _bj_rows_range(self, r0, r1, out):
        """out (r1 - r0, G) <- rows r0:r1 of Y' = D^-1 (aoP ao)^2 (r0, r1 on block boundaries): pair-gram rows of those
        points, forward solves with their own diagonal blocks only."""
        be, st = self.backend, self._fit_state
        ip_off = np.asarray(st['ip_off'])
        i0, i1 = int(np.searchsorted(ip_off, r0)), int(np.searchsorted(ip_off, r1))
        assert ip_off[i0] == r0 and ip_off[i1] == r1
        G = self.ao.shape[1]
        Dinv = self._block_inverse(st['Dblk'], ip_off)
        sub = (ip_off[i0:i1 + 1] - r0).astype(np.int32)
        if getattr(self, '_psi', None) is not None:
            be.pair_prod_rows(self.aoP[r0:r1], self._psiP[r0:r1], self.ao, self._psi, G, out)
            if Dinv is not None:
                be.block_apply(Dinv[r0:r1, r0:r1], sub, out)
            else:
                be.block_solve(st['Dblk'][r0:r1, r0:r1], sub, 0, 0, out)
        elif Dinv is not None:
            be.pair_rows_block_apply(self.aoP[r0:r1], self.ao, G, Dinv[r0:r1, r0:r1], sub, out)
        else:
            be.pair_gram_rows(self.aoP[r0:r1], self.ao, G, out, 0)
            be.block_solve(st['Dblk'][r0:r1, r0:r1], sub, 0, 0, out)

    def _finish_W_paneled(self, W, probe=None):
        """S3c + S4 + S5 when the P fit rows do not fit into HBM together (c_isdf above ~10 at configs[2] on one GPU).
        The rows are produced panel by panel into ONE resident buffer; M' = w conv(Y') Y'^T is assembled from
          * the diagonal panel blocks (resident rows against themselves, upper half: isdf_coulomb_W), and
          * the blocks above the diagonal: the rows of every EARLIER panel are recomputed in FFT-batch-sized pieces
            (pair-gram rows + their own block solves, 2 N G flop per row against 2 P G for the product), convolved and
            multiplied with the resident panel (isdf_coulomb_rows + isdf_gemm_nt).
        With R panels the row stage runs (R + 1) / 2 times and the convolutions likewise; the P^2 G product is done once.
        probe: (E (n, P), F (n, G)) - the route check's combination rows; F <- E Y' is accumulated on the way."""
        be, st = self.backend, self._fit_state
        cell = self.cell
        P = W.shape[0]
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        w = cell.vol / G
        ip_off = np.asarray(st['ip_off'], dtype=np.int64)
        panels = st['panels']
        buf = st['rows']                                        # (largest panel, G)
        nbat = int(self.fft_batch or 512)
        big = int(np.diff(ip_off).max())
        self._last_fft_batch = nbat
        # recomputed rows arrive block by block (a block's solve needs all of its rows) but leave in batches of exactly nbat
        # rows - one FFT plan, full GEMM tiles: the scratch holds a batch plus the block that overshoots it
        scratch = self._buffer('rows_scratch', (nbat + big, G))
        for q, (q0, q1) in enumerate(panels):
            Yq = buf[:q1 - q0]
            self._bj_rows_range(q0, q1, Yq)
            be.coulomb_W(Yq, mesh, a, 0, q1 - q0, min(nbat, q1 - q0), W[q0:q1, q0:q1], upper_only=True)
            if probe is not None:
                be.rows_combine(probe[0][:, q0:q1], Yq, probe[1], accumulate=q > 0)
            for p0, p1 in panels[:q]:
                nxt, fill, g0 = p0, 0, p0              # next row to recompute, rows waiting in the scratch, their first index
                while nxt < p1 or fill > 0:
                    while fill < nbat and nxt < p1:
                        i = int(np.searchsorted(ip_off, nxt))
                        j = i + 1
                        while j + 1 < len(ip_off) and ip_off[j + 1] <= p1 and fill + (ip_off[j + 1] - nxt) <= nbat + big:
                            j += 1
                        b1 = int(ip_off[j])
                        self._bj_rows_range(nxt, b1, scratch[fill:fill + (b1 - nxt)])
                        fill += b1 - nxt
                        nxt = b1
                    take = min(nbat, fill)
                    V = scratch[:take]
                    be.coulomb_rows(V, mesh, a, take)
                    be.gemm_nt(V, Yq, W[g0:g0 + take, q0:q1], alpha=w)
                    if fill > take:
                        scratch[:fill - take].copy_(scratch[take:fill].clone() if fill - take > take else scratch[take:fill])
                    g0 += take
                    fill -= take
        be.symmetrize_upper(W)
        self._bj_finish(st['Afac'], st['Dblk'], st['ip_off'], W)

    # ---- spectral W: W = X X^T from the scaled half spectra of the fit rows inside a sphere --------------------------------
    def _spectral_plan(self):
        """dict(idx, scale, npts, ldx, fraction) for the kernel the backend is set to, or None when the spectral form does not
        apply (switched off; a mesh the plane FFT does not cover; a kernel table with negative entries, e.g. exxdiv='vcut_ws').
        By Parseval  w sum_r Theta_P conv(Theta_Q) = (w / G) sum_G coulG(G) fft(Theta_P)(G) conj(fft(Theta_Q)(G));  over the half
        spectrum (multiplicity 1 on the kz = 0 and Nyquist planes, 2 between) that is X X^T with
        X[P][2j], X[P][2j+1] = sqrt(mult_j w coulG_j / G) (Re, Im) fft(Theta_P)(G_j).  ``w_sphere`` percent of the radius of the
        sphere inscribed in the reciprocal FFT box bounds the points kept (0: the whole box, no truncation): the fitted pair
        densities are as band-limited as the AO products they are combinations of, and what the corners of the box carry is the
        square of the fit error times a small kernel value - 1e-9 Eh in E_K at BASELINE configs[2]."""
        be = self.backend
        if not getattr(self, 'w_spectral', False) or not hasattr(be, 'spectral_rows'):
            return None
        mesh = np.asarray(self.mesh, dtype=np.int32)
        # (one plan per kernel state of the backend: the table comes down to the host and the point list goes up again)
        epoch = getattr(be, 'kernel_epoch', None)
        ckey = (tuple(int(x) for x in mesh), epoch, str(getattr(self, 'w_sphere', 'auto')), float(getattr(self, 'w_sphere_tol', 0.0)),
                int(self.fft_batch or 512), int(getattr(self, 'w_sort_bins', 0) or 0))
        cached = getattr(self, '_spectral_plan_cache', None)
        if epoch is not None and cached is not None and cached[0] == ckey:
            return cached[1]
        plan = self._spectral_plan_make(mesh)
        self._spectral_plan_cache = (ckey, plan)
        return plan

    def _spectral_plan_make(self, mesh):
        be = self.backend
        if not be.spectral_supported(mesh, int(self.fft_batch or 512)):
            return None
        a = np.asarray(self.cell.lattice_vectors(), dtype=float)
        G = int(np.prod(mesh))
        cg = be.coulG_half(mesh, a)                                  # (n0, n1, n2/2+1), 1/G inside
        if (cg < 0).any():
            return None
        n0, n1, n2 = (int(x) for x in mesh)
        n2h = n2 // 2 + 1
        keep = cg > 0
        pct = getattr(self, 'w_sphere', 'auto')
        auto = isinstance(pct, str)
        pct = 100.0 if auto else float(pct or 0.0)
        box = keep.copy()
        if pct > 0:
            b = 2 * np.pi * np.linalg.inv(a).T                       # rows b_i
            f0, f1, f2 = np.fft.fftfreq(n0, 1.0 / n0), np.fft.fftfreq(n1, 1.0 / n1), np.arange(n2h, dtype=float)
            Gv = f0[:, None, None, None] * b[0] + f1[None, :, None, None] * b[1] + f2[None, None, :, None] * b[2]
            g2 = np.einsum('xyzc,xyzc->xyz', Gv, Gv)
            # inscribed sphere: the faces of the box sit at the frequencies +-(n_i - 1) // 2 (the Nyquist planes of even meshes,
            # whose frequencies are ambiguous, lie outside), at distance 2 pi f / |a_i| from the origin
            rmin = min(2 * np.pi * ((int(n) - 1) // 2) / np.linalg.norm(a[i]) for i, n in enumerate(mesh)) * pct / 100.0
            keep &= g2 <= rmin * rmin * (1 + 1e-12)
        mult = np.full((n0, n1, n2h), 2.0)
        mult[:, :, 0] = 1.0
        if n2 % 2 == 0:
            mult[:, :, n2 // 2] = 1.0
        w = self.cell.vol / G
        if auto:
            # 'auto': the truncation is taken only when this mesh resolves the AO pair products - measured once per mesh on a few
            # random products (phi^T c)(phi^T d): the share of their Coulomb energy that sits outside the sphere (coarse test
            # meshes fail this by orders of magnitude; the production meshes of BASELINE.json read 1e-14 and less)
            key = (tuple(int(x) for x in mesh), round(float(pct), 6))
            if getattr(self, '_sphere_share', (None,))[0] != key:
                self._sphere_share = (key, self._sphere_energy_share(mesh, cg, mult, box, keep, w))
            if not (self._sphere_share[1] <= float(getattr(self, 'w_sphere_tol', 1e-11))):
                return None
        idx = np.flatnonzero(keep.ravel()).astype(np.int32)
        if len(idx) == 0:
            return None
        # ORDER of the packed points = order of the product's accumulation chains.  The terms fall off steeply with |G|: with the
        # points in descending |G| the small terms come first and the few large ones at the end of the chain - the strips' rounding
        # drops from 2.9e-15 (points as they lie in the half spectrum) to 3.6e-16 of sqrt(M_PP M_QQ), below the classic product's
        # 1.1e-15 (tools/spectral_noise_study.py).  Sorted in `w_sort_bins` shells of |G|^2 (index order inside a shell), so that the
        # gather still reads runs of neighbouring points.
        nbins = int(getattr(self, 'w_sort_bins', 0) or 0)
        if nbins > 0:
            b = 2 * np.pi * np.linalg.inv(a).T
            f0, f1, f2 = np.fft.fftfreq(n0, 1.0 / n0), np.fft.fftfreq(n1, 1.0 / n1), np.arange(n2h, dtype=float)
            Gv = f0[:, None, None, None] * b[0] + f1[None, :, None, None] * b[1] + f2[None, None, :, None] * b[2]
            g2p = np.einsum('xyzc,xyzc->xyz', Gv, Gv).ravel()[idx]
            bins = np.minimum((g2p * (nbins / g2p.max())).astype(np.int64), nbins - 1)
            idx = idx[np.argsort(-bins, kind='stable')]
        scale = np.sqrt(mult.ravel()[idx] * w * cg.ravel()[idx])
        npts = len(idx)
        ldx = -(-2 * npts // 128) * 128
        return dict(idx=be.to_device(idx), scale=be.to_device(scale), npts=npts, ldx=ldx, fraction=2.0 * npts / G)

    def _sphere_energy_share(self, mesh, cg, mult, box, keep, w, ntest=8):
        """max over ntest random AO pair products t = (phi^T c)(phi^T d) of  E_outside / E_total,  E = (t | v | t) summed over the
        kernel table's points outside the sphere / over all of them (one batch of forward transforms)."""
        import torch
        be = self.backend
        ao = getattr(self, 'ao', None)
        if ao is None:
            return np.inf
        nao, G = ao.shape
        rng = np.random.default_rng(7)
        cd = be.to_device(rng.standard_normal((2 * ntest, nao)) / np.sqrt(nao))
        rows = be.empty((2 * ntest, G))
        be.gemm_nn(cd, ao, rows)
        t = rows[:ntest].contiguous()
        be.hadamard_rows(t, rows[ntest:].contiguous())
        Gfull = int(np.prod(mesh))
        if G != Gfull:
            # grid-sharded build: phi lives on this rank's slice - the products are assembled over the whole grid (zero-padded
            # all-reduce, 8 rows), every rank then measures the same numbers
            g0, g1 = self._slice
            full = be.zeros((ntest, Gfull))
            full[:, g0:g1] = t
            self.comm.all_reduce_sum(full)
            t = full
        idx = np.flatnonzero(box.ravel()).astype(np.int32)
        scale = np.sqrt(mult.ravel()[idx] * w * cg.ravel()[idx])
        ldx = -(-2 * len(idx) // 128) * 128
        X = be.empty((ntest, ldx))
        be.spectral_rows(t, mesh, be.to_device(idx), be.to_device(scale), X, batch=ntest)
        x = be.to_host(X)[:, :2 * len(idx)]
        e = x[:, 0::2] ** 2 + x[:, 1::2] ** 2
        outside = ~keep.ravel()[idx]
        tot = e.sum(axis=1)
        return float((e[:, outside].sum(axis=1) / np.where(tot > 0, tot, 1.0)).max())

    def _finish_W_spectral(self, W, probe=None):
        """S3c + S4 + S5 of the block-Jacobi route in the spectral form: the fit rows Y' = D^-1 (pair rows) are produced in
        FFT-batch-sized pieces (whole preconditioner blocks at a time), transformed forward only and packed into X (P, ldx) -
        about half the size of the rows themselves, so configs[2] needs no panels; M' = X X^T (upper half, 512-row strips through
        isdf_gemm_nt) and the route's P x P finishing.  probe as in _finish_W_paneled."""
        be, st = self.backend, self._fit_state
        plan = self._spectral_plan()
        ip_off = np.asarray(st['ip_off'], dtype=np.int64)
        P = W.shape[0]
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        if plan is None:
            # a kernel the spectral form cannot take (negative table entries): the paneled build on the same factors
            rows_avail = int(self._bufs['theta'].numel() // G)
            st2 = dict(st, kind='blockjacobi-paneled', rows=self._buffer('theta', (rows_avail, G)),
                       panels=self._panel_plan(ip_off, rows_avail))
            self._fit_state = st2
            try:
                return self._finish_W_paneled(W, probe=probe)
            finally:
                self._fit_state = st
        nbat = int(self.fft_batch or 512)
        big = int(np.diff(ip_off).max())
        self._last_fft_batch = nbat
        ldx = plan['ldx']
        X = self._buffer('theta', (P, ldx))
        scratch = self._buffer('rows_scratch', (nbat + big, G))
        self.w_spectral_fraction = plan['fraction']
        self._last_spectral_ldx = ldx
        nxt, fill, g0 = 0, 0, 0
        while nxt < P or fill > 0:
            while fill < nbat and nxt < P:
                i = int(np.searchsorted(ip_off, nxt))
                j = i + 1
                while j + 1 < len(ip_off) and fill + (ip_off[j + 1] - nxt) <= nbat + big:
                    j += 1
                b1 = int(ip_off[j])
                self._bj_rows_range(nxt, b1, scratch[fill:fill + (b1 - nxt)])
                fill += b1 - nxt
                nxt = b1
            take = min(nbat, fill)
            rows = scratch[:take]
            if probe is not None:
                be.rows_combine(probe[0][:, g0:g0 + take], rows, probe[1], accumulate=g0 > 0)
            be.spectral_rows(rows, mesh, plan['idx'], plan['scale'], X[g0:g0 + take], batch=take)
            if fill > take:
                scratch[:fill - take].copy_(scratch[take:fill].clone() if fill - take > take else scratch[take:fill])
            g0 += take
            fill -= take
        for b0 in range(0, P, nbat):
            b1 = min(P, b0 + nbat)
            be.gemm_nt(X[b0:b1], X[b0:], W[b0:b1, b0:], alpha=1.0)
        be.symmetrize_upper(W)
        self._bj_finish(st['Afac'], st['Dblk'], st['ip_off'], W)

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

    def _bj_probe_densities(self, aoT_P):
        """Probe densities at the points, (n, P): t_j = diag(phi_P R_j phi_P^T) for fixed random symmetric matrices R_j in the
        AO x AO pair space; t_j = diag(phi_P R_j psi_P^T) with random (N, nocc) matrices R_j in the (AO x occupied) pair space -
        the probes have to lie in the span the fit represents: outside it t^T W t is dominated by the badly determined
        directions of the Gram matrix, which the exchange never sees (measured: AO-pair probes read 2e-6 on an
        (AO x occupied) fit whose K equals the Cholesky route's to 1e-10)."""
        be = self.backend
        planes = aoT_P if isinstance(aoT_P, (list, tuple)) else [aoT_P]
        nao, P = planes[0].shape
        n = int(self.bj_nprobe)
        if getattr(self, '_psi', None) is not None and len(planes) == 1:
            psiP = self._psiP
            nocc = psiP.shape[1]
            cached = getattr(self, '_probe_Rocc', None)
            if cached is None or cached[0] != (n, nao, nocc):
                rng = np.random.default_rng(20240203)
                self._probe_Rocc = ((n, nao, nocc), be.to_device(rng.standard_normal((n, nao, nocc))))
            d_R = self._probe_Rocc[1]
            aoP = planes[0].T.contiguous()                  # (P, nao)
            ones = be.to_device(np.ones((nocc, 1)))
            T = be.empty((n, P))
            tmp = be.empty((P, nocc))
            col = be.empty((P, 1))
            for j in range(n):
                be.gemm_nn(aoP, d_R[j], tmp)
                be.hadamard_rows(tmp, psiP)
                be.gemm_nn(tmp, ones, col)
                T[j].copy_(col[:, 0])
            return T
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
        return T

    def _bj_probe_vectors(self, aoT_P, Afac, Dblk, ip_off):
        """(T0, E): the probe densities at the points and their combination rows e_j = A'^-1 D^-1 t_j, so that the fitted
        density of probe j is f_j = Theta^T t_j = Y'^T e_j (paneled builds accumulate F = E Y' panel by panel)."""
        T = self._bj_probe_densities(aoT_P)
        T0 = T.clone()
        self.backend.bj_probe_vectors(T, Afac, Dblk, ip_off)
        return T0, T

    def _bj_probe_energies(self, T0, F, W, grid_slice):
        """Largest relative mismatch between t^T W t (through the matrix) and w sum_g f conv(f) (through the fitted density
        itself, F (n, columns of this rank))."""
        be, comm = self.backend, self.comm
        cell = self.cell
        n = T0.shape[0]
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        TW = be.empty(tuple(T0.shape))
        be.gemm_nt(T0, W, TW)
        e_mat = np.einsum('jp,jp->j', be.to_host(TW), be.to_host(T0))
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

    def _bj_probe_mismatch(self, aoT_P, Afac, Dblk, ip_off, Yp, ng, grid_slice, W=None):
        """A-posteriori check of the S3c route: for random symmetric R_j the density t_j = diag(phi_P R_j phi_P^T)
        at the points has the Coulomb energy  t^T W t  through the matrix and  w sum_g f conv(f), f = Theta^T t,
        through the fitted density itself (vector operations only: one pass over Y', nprobe FFTs).  The second form
        does not see the cond(A')-amplified rounding of M'; their largest relative difference is returned.
        aoT_P: (nao, P), or a list of such planes whose densities are added (k-points: Re/Im u^k at the points, the
        density sum_k u^k* R u^k with real symmetric R; W = the real plane of W^{q=0})."""
        be = self.backend
        T = self._bj_probe_densities(aoT_P)
        T0 = T.clone()
        F = be.empty((T.shape[0], ng))
        be.bj_probe_rows(T, Afac, Dblk, ip_off, Yp, ng, F)
        return self._bj_probe_energies(T0, F, self.W if W is None else W, grid_slice)
