"""Grid-sharded multi-GPU build and J/K of ``isdf.ISDF``: one process per GPU, collectives through
``parallel.Comm`` (RCCL on GPUs, gloo in CPU tests).  Host orchestration only."""
import time
import warnings
import numpy as np
import torch
from . import gto
from ._common import partition_grid_by_atom, _default_fft_batch


class ShardedMixin:
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

    # ---- multi-GPU: grid-sharded build, row-sharded K (DESIGN.md "Multi-GPU") -------------------------
    def _build_sharded(self):
        """Every rank owns a contiguous slice S_r of the grid (natural order).

        S1  collocation on the slice                               no communication
        S2  per-atom selection, atom blocks dealt round-robin       all_gather of the point lists (P ints)
        S3  A_PP Cholesky replicated (P^3/3, small); fit on slice   no communication
        S4  rows of Theta assembled by all-to-all, FFT convolution, scattered back by all-to-all
        S5  W_r = w V[:, S_r] Theta[:, S_r]^T                       all_reduce(W)  (RCCL over xGMI)
        Streaming over row batches bounds memory at any rank count.
        pair_space='occ': stops after the candidate stage like the single-GPU build; get_jk makes the pick and the fit for its
        density (_ensure_fit -> _pick_and_fit_sharded).
        """
        cell, be, comm = self.cell, self.backend, self.comm
        if self.select not in ('local', 'refined'):
            raise NotImplementedError("multi-GPU build needs select='local' or 'refined'")
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
        owner = be.partition_by_atom(coords, cell.atom_coords(), a)
        perm = np.argsort(owner, kind='stable').astype(np.int64)
        counts = np.bincount(owner, minlength=cell.natm)
        blk_off = np.append(0, np.cumsum(counts)).astype(np.int64)
        nip_final = np.minimum(self.nip_per_atom(), counts).astype(np.int32)
        nip = nip_final
        if self.select == 'refined':                      # the per-atom selections are candidates (isdf.ISDF._refine_pick)
            nip = np.minimum(np.ceil(self.nip_per_atom() * float(self.refine_over)).astype(np.int64), counts).astype(np.int32)
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
        self._tick('S2_select_candidates' if self.select == 'refined' else 'S2_select_ip', t0)
        # the fit rows of this rank's slice have to fit next to phi, W and the factors (the paneled build is single-GPU: with
        # R ranks a slice holds c_isdf up to about 10 R at configs[2])
        Pmax = int(nip_final.sum())
        need = 8 * Pmax * ng + 4 * 8 * Pmax * Pmax
        have = be.free_bytes() + sum(int(b.numel()) * 8 for k, b in self._bufs.items() if k in ('theta', 'W', 'factor', 'Dblk', 'Dinv'))
        rows_do_not_fit = comm.agree_max(1.0 if need > have else 0.0) > 0.0
        spectral_may = self.w_spectral and not self._want_theta and self.c_isdf <= self.w_spectral_max_c and self.fit_route != 'cholesky'
        if rows_do_not_fit and not spectral_may:
            raise MemoryError('ISDF: %d fit rows of %d grid columns (%.0f GB with the P x P matrices) do not fit on rank %d '
                              '(%.0f GB obtainable); use more ranks or fewer points' % (Pmax, ng, need / 1e9, rk, have / 1e9))
        self._sel = dict(sharded=True, owner=owner, merged=merged, nip_final=nip_final, rows_do_not_fit=rows_do_not_fit)
        if self.pair_space == 'occ':
            self._fit_pending = True
            self._built = True
            return self
        return self._pick_and_fit_sharded()

    def _pick_and_fit_sharded(self, orbitals=None):
        """S2 second stage + S3 + S4 + S5 of the grid-sharded build from the candidates in self._sel; orbitals: see
        isdf.ISDF._pick_and_fit (the occupied orbitals on this rank's slice are psi = C^T phi[:, S_r], no communication)."""
        cell, be, comm, sel = self.cell, self.backend, self.comm, self._sel
        owner, merged, nip_final = sel['owner'], sel['merged'], sel['nip_final']
        nao, ng = self.ao.shape
        g0, g1 = self._slice
        t0 = time.perf_counter()
        self._fit_state = None
        self._W_omega = {}
        self._V = None
        self._psi = self._psiP = None
        if orbitals is not None:
            self._psi = self._buffer('psi', (orbitals.shape[1], ng))
            be.gemm_nn(be.to_device(np.ascontiguousarray(orbitals.T)), self.ao, self._psi)
            t0 = self._tick('S2_occupied_on_grid', t0)
        none = np.zeros(0, dtype=np.int64)                 # an atom that owns no grid points / no AOs selects nothing
        if self.select == 'refined':
            # phi at the candidates from the slice collocations (zero-padded all_reduce, 8 m N bytes); the pivoted
            # Cholesky of the candidate Gram matrix runs on rank 0 and its answer is shared: one decision for all ranks
            cand = np.concatenate([merged.get(b, none) for b in range(cell.natm)]).astype(np.int64)
            aoC = self._slice_columns(cand, g0, g1).T.contiguous()
            psiC = None if self._psi is None else self._slice_columns(cand, g0, g1, src=self._psi).T.contiguous()
            chosen = comm.run_on_root(lambda: self._refine_pick(aoC, cand, int(nip_final.sum()), psiC=psiC))
            chosen = comm.all_gather_object(chosen)[0]
            del aoC, psiC
            own = owner[chosen]
            merged = {b: chosen[own == b] for b in range(cell.natm)}
        clusters = self._bj_clusters()
        self.ip = np.concatenate([merged.get(b, none) for cl in clusters for b in cl]).astype(np.int64)
        P = len(self.ip)
        t0 = self._tick('S2_select_ip', t0)

        # S3: phi at the points = columns of the slice collocations (every point lies in exactly one slice; zero-padded
        # all_reduce of 8 P N bytes).  Taking them from the SAME evaluation as the fit's right-hand sides keeps
        # B[:, ip] == A_PP to the last bit (a separate collocation differs by the image-screening tolerance, which
        # the fit amplifies by cond(A)).  P x P factorisations replicated, rows of the fit on the slice.
        aoP_T = self._slice_columns(self.ip, g0, g1)
        self.aoP = self._buffer('aoP', (P, nao))
        if self._psi is not None:
            self._psiP = self._buffer('psiP', (P, self._psi.shape[0]))
            self._psiP.copy_(self._slice_columns(self.ip, g0, g1, src=self._psi).T)
        ar = be.to_device(np.arange(P, dtype=np.int64))
        ip_off = self._bj_blocks([len(merged.get(b, none)) for b in range(cell.natm)], clusters)
        routes = self._fit_routes()
        # spectral form of W (block-Jacobi route): the fit rows are produced batch by batch inside _finish_W_sharded_spectral and
        # never held as a whole - a rank keeps its K slice of X (P x ldx / R) instead of P x G / R rows
        self.w_spectral_fraction = None
        plan = None
        # (tried first also above bj_max_c when the route is left to 'auto' - as the single-GPU build does where the rows would need
        # panels: the probe check decides, and the Cholesky route below is the fallback)
        if self.w_spectral and not self._want_theta and self.c_isdf <= self.w_spectral_max_c and \
                (routes[0] == 'blockjacobi' or sel.get('rows_do_not_fit') or self.fit_route == 'auto'):
            plan = self._spectral_plan()
            if comm.agree_max(0.0 if plan is not None else 1.0) != 0.0:
                plan = None
        if plan is not None:
            Afac = self._buffer('factor', (P, P))
            Dblk = self._buffer('Dblk', (P, P))

            def root_factorise():
                self._bj_prepare(aoP_T, 0, ar, ip_off, self.aoP, scratch=self._buffer('W', (P, P)))
                return self.reg_used
            reg = comm.run_on_root(root_factorise)
            if comm.rank != 0:
                be.gather_aoP(aoP_T, ar, self.aoP)
                self._Dinv_key = None
            comm.broadcast(Afac)
            comm.broadcast(Dblk)
            self.reg_used = comm.agree_max(reg or 0.0)
            t0 = self._tick('S3_fit', t0)
            self._fit_state = dict(kind='blockjacobi-spectral', theta=None, sharded=True, Afac=Afac, Dblk=Dblk, ip_off=ip_off, chol=None)
            probe = None
            if self.fit_route == 'auto':
                T0, E = self._bj_probe_vectors(aoP_T, Afac, Dblk, ip_off)
                probe = (E, be.empty((E.shape[0], ng)))
            self.W = self._buffer('W', (P, P))
            self._finish_W_sharded_spectral(self.W, plan, probe=probe)
            t0 = self._tick('S4S5_coulomb_W', t0)
            self.fit_route_used = 'blockjacobi'
            ok = True
            if probe is not None:
                self.bj_check = comm.agree_max(self._bj_probe_energies(T0, probe[1], self.W, (g0, g1)))
                t0 = self._tick('S5_route_check', t0)
                tol = min(self.bj_check_tol, self.w_spectral_check_tol)
                ok = self.bj_check <= tol
                if not ok:
                    warnings.warn('ISDF: block-Jacobi fit route failed its probe check (mismatch %.2e > %.2e) in the spectral build; '
                                  'rebuilding W the classic way' % (self.bj_check, tol))
                    self.w_spectral_fraction = None
            if ok:
                del aoP_T
                self._built = True
                return self
            if sel.get('rows_do_not_fit'):
                raise MemoryError('ISDF: the spectral form failed its probe check and the classic form\'s %d fit rows of %d grid '
                                  'columns do not fit on rank %d; use more ranks or fewer points' % (P, ng, comm.rank))
        theta = self._buffer('theta', (P, ng))
        for route in routes:
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
                    self._Dinv_key = None
                comm.broadcast(Afac)
                comm.broadcast(Dblk)
                self.reg_used = comm.agree_max(reg or 0.0)
                self._bj_rows(self.aoP, 0, self.ao, ng, Dblk, ip_off, theta)
            else:
                chol = self._buffer('factor', (P, P))
                if self._psi is None:
                    reg = comm.run_on_root(lambda: be.fit_prepare(aoP_T, ar, self.reg_rel, self.aoP, chol))
                else:
                    def root_chol():
                        be.gather_aoP(aoP_T, ar, self.aoP)
                        be.gram_prod(self.aoP, self._psiP, chol)
                        be.shift_diag(chol, self.reg_rel)
                        return self.reg_rel + be.chol_inplace(chol, 0.0, scratch=self._buffer('W', (P, P)))
                    reg = comm.run_on_root(root_chol)
                if comm.rank != 0:
                    be.gather_aoP(aoP_T, ar, self.aoP)
                comm.broadcast(chol)
                self.reg_used = comm.agree_max(reg or 0.0)
                if self._psi is None:
                    be.fit_apply(chol, self.aoP, self.ao, ng, theta, forward_only=not self._want_theta)
                else:
                    be.pair_prod_rows(self.aoP, self._psiP, self.ao, self._psi, ng, theta)
                    be.factor_solve_half(chol, False, theta)
                    if self._want_theta:
                        be.factor_solve_half(chol, True, theta)
            t0 = self._tick('S3_fit', t0)

            self._fit_state = dict(kind=route if route == 'blockjacobi' else ('explicit' if self._want_theta else 'cholesky'),
                                   theta=theta, sharded=True,
                                   Afac=Afac if route == 'blockjacobi' else None, Dblk=Dblk if route == 'blockjacobi' else None,
                                   ip_off=ip_off, chol=None if route == 'blockjacobi' else chol)
            self.W = self._buffer('W', (P, P))
            self._finish_W_sharded(self.W)
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
        self._keep_V_for_robust_k_sharded(t0)
        del theta, aoP_T
        self._built = True
        return self

    def _slice_columns(self, idx, g0, g1, src=None):
        """phi (or the rows of ``src``, e.g. the occupied orbitals on the slice) at the grid points idx, (rows, len(idx)), on
        every rank: each point lies in exactly one rank's slice [g0, g1); zero-padded all_reduce of 8 rows len(idx) bytes."""
        be, comm = self.backend, self.comm
        src = self.ao if src is None else src
        idx = np.asarray(idx, dtype=np.int64)
        out = be.zeros((src.shape[0], len(idx)))
        mine = np.nonzero((idx >= g0) & (idx < g1))[0]
        if len(mine):
            loc = be.empty((src.shape[0], len(mine)))
            be.gather_cols(src, be.to_device(idx[mine] - g0), loc)
            out[:, be.to_device(mine)] = loc
            del loc
        comm.all_reduce_sum(out)
        return out

    def _finish_W_sharded(self, W):
        """S4 + S5 of the grid-sharded build for the fit held in self._fit_state (rows on this rank's grid slice + the
        replicated factors): two all-to-alls around the row convolution, partial W over the slice, all-reduce, finishing.
        Uses the Coulomb kernel the backend is set to (plain, or range-separated for get_jk(omega=...))."""
        cell, be, comm = self.cell, self.backend, self.comm
        st = self._fit_state
        if st['kind'] == 'blockjacobi-spectral':
            # a rebuild of W from the same fit with another kernel (range separation): the rows are recomputed as in the first build
            plan = self._spectral_plan()
            if comm.agree_max(0.0 if plan is not None else 1.0) != 0.0:
                raise NotImplementedError('this Coulomb kernel has no spectral form (negative table entries) and the fit was built '
                                          'without resident rows; set w_spectral = False for this kernel')
            return self._finish_W_sharded_spectral(W, plan)
        theta = st['theta']
        P, ng = theta.shape
        R, rk = comm.size, comm.rank
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        # S4 + S5 streamed over row batches: rank q convolves rows P_q[t*nb : (t+1)*nb] in step t.  Two streams: the
        # exchange/FFT pipeline of step t+1 (all-to-all, row assembly, convolution, all-to-all) runs on a side stream
        # while the MFMA products of step t run on the work stream; every buffer is allocated once (two slots).
        w = cell.vol / G
        W.zero_()
        slices = [comm.split_range(G, r) for r in range(R)]
        rows = [comm.split_range(P, r) for r in range(R)]
        nb = self.fft_batch or _default_fft_batch(G, max(1, P // R))
        if not self.fft_batch:
            # two slots of three batch-sized buffers + the FFT's half spectrum and work area, 2 GiB kept back
            nb = min(nb, int(max(0, be.free_bytes() - (2 << 30)) // (72 * G)))
        nb = max(1, min(nb, max(hi - lo for lo, hi in rows)))
        nsteps = max(-(-(hi - lo) // nb) for lo, hi in rows)
        nslot = min(2, nsteps)
        pieces = [be.empty((nb * G,)) for _ in range(nslot)]         # per slot: the R exchanged pieces of this rank's batch
        full = [be.empty((nb, G)) for _ in range(nslot)]             # the batch's rows over the whole grid
        recvV = [be.empty((R * nb * ng,)) for _ in range(nslot)]     # V[bat_q, S_r] from every rank q

        def batches(t):
            bat = [(min(lo + t * nb, hi), min(lo + (t + 1) * nb, hi)) for lo, hi in rows]   # rows handled by rank q
            return bat, [hi - lo for lo, hi in bat]

        def views(flat, nrow_of, width_of):
            out, off = [], 0
            for q in range(R):
                n = nrow_of(q) * width_of(q)
                out.append(flat[off:off + n].view(nrow_of(q), width_of(q)))
                off += n
            return out

        def exchange_and_convolve(t, slot):
            bat, nrow = batches(t)
            mine = nrow[rk]
            # all-to-all 1: send Theta[bat_q, S_r] to q; receive Theta[bat_r, S_q] from q
            recv = views(pieces[slot], lambda q: mine, lambda q: slices[q][1] - slices[q][0])
            comm.all_to_all(recv, [theta[lo:hi] for lo, hi in bat])
            rows_full = full[slot][:mine]
            for (s0, s1), piece in zip(slices, recv):
                rows_full[:, s0:s1] = piece
            if mine:
                be.coulomb_rows(rows_full, mesh, a, mine)
            # all-to-all 2: send V[bat_r, S_q] to q; receive V[bat_q, S_r] from q
            for (s0, s1), piece in zip(slices, recv):
                piece.copy_(rows_full[:, s0:s1])
            got = views(recvV[slot], lambda q: nrow[q], lambda q: ng)
            comm.all_to_all(got, recv)
            return bat, nrow, got

        def products(bat, nrow, got):
            for q in range(R):
                if nrow[q]:
                    # W[bat_q, c0:] = w V[bat_q, S_r] Theta[c0:, S_r]^T  (partial over this rank's slice).
                    # W is symmetric: only the columns from the batch's first row on are computed and
                    # the lower part is mirrored after the all-reduce (half the flops).
                    c0 = bat[q][0]
                    be.gemm_nt(got[q], theta[c0:], W[bat[q][0]:bat[q][1], c0:], alpha=w, beta=0.0)

        side = be.new_stream()
        ready = be.record_event()                    # the fit rows (and W.zero_) are complete on the work stream
        staged, consumed = {}, {}
        for t in range(nsteps + 1):
            if t < nsteps:
                with be.on_stream(side):
                    be.wait_event(ready if t == 0 else None)
                    be.wait_event(consumed.get(t - nslot))          # the slot's buffers were read by the products of step t - nslot
                    out = exchange_and_convolve(t, t % nslot)
                    staged[t] = (out, be.record_event())
            if t >= 1:
                out, ev = staged.pop(t - 1)
                be.wait_event(ev)
                products(*out)
                consumed[t - 1] = be.record_event()
        del staged, consumed, pieces, full, recvV
        comm.all_reduce_sum(W)
        be.symmetrize_upper(W)
        if st['kind'] == 'blockjacobi':
            self._bj_finish_sharded(st['Afac'], st['Dblk'], st['ip_off'], W)
        elif st['kind'] == 'cholesky':
            be.W_from_factor(st['chol'], 0, W)

    def _finish_W_sharded_spectral(self, W, plan, probe=None):
        """The spectral form of W (fit_route.FitRouteMixin._finish_W_spectral) on the grid-sharded build.  The points are dealt to
        the ranks in whole preconditioner blocks; in every step each rank computes, ON ITS GRID SLICE, the fit rows of every
        rank's current batch of blocks (pair rows + block solves: nothing is held beyond the step), the first all-to-all assembles
        rank q's batch over the whole grid on rank q, which transforms it forward and packs it into X[bat_q, :]; the second
        all-to-all hands every rank r its K slice X[:, K_r] of ALL rows (0.36 of the classic form's volume); when all batches are
        through, W_r = X[:, K_r] X[:, K_r]^T (upper half) and one all-reduce.  The products cannot start before the last batch has
        arrived (every row meets every row), so the exchange is not hidden behind them as in the classic form - it is shorter
        instead, and a rank holds P x ldx / R doubles of X where the classic form holds P x G / R of rows.
        probe: (E (n, P), F (n, ng)) - the route check's combination rows; F <- E Y' on this rank's slice is accumulated on the way."""
        cell, be, comm = self.cell, self.backend, self.comm
        st = self._fit_state
        ip_off = np.asarray(st['ip_off'], dtype=np.int64)
        P = int(ip_off[-1])
        ng = self.ao.shape[1]
        R, rk = comm.size, comm.rank
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        ldx = int(plan['ldx'])
        self.w_spectral_fraction = plan['fraction']
        self._last_spectral_ldx = ldx
        self._last_fft_batch = int(self.fft_batch or 512)
        slices = [comm.split_range(G, r) for r in range(R)]
        ks = [tuple(16 * x for x in comm.split_range(ldx // 16, r)) for r in range(R)]      # K slices (ldx is a multiple of 128)
        kw = ks[rk][1] - ks[rk][0]
        nbat = int(self.fft_batch or 512)
        big = int(np.diff(ip_off).max())
        # block-aligned shares of the points, and inside a share batches of whole blocks of at most max(nbat, big) rows
        nblk = len(ip_off) - 1
        cuts = [int(np.searchsorted(ip_off, P * r / R, side='left')) for r in range(R)] + [nblk]
        cuts = [min(max(c, 0), nblk) for c in cuts]
        plans = []
        for q in range(R):
            b0, b1 = cuts[q], max(cuts[q + 1], cuts[q])
            steps, i = [], b0
            while i < b1:
                j = i + 1
                while j < b1 and ip_off[j + 1] - ip_off[i] <= max(nbat, big):
                    j += 1
                steps.append((int(ip_off[i]), int(ip_off[j])))
                i = j
            plans.append(steps)
        nsteps = max(len(p) for p in plans)
        cap = max([hi - lo for p in plans for lo, hi in p] + [1])
        Xloc = be.empty((P, max(kw, 1)))
        rows_loc = be.empty((R * cap, ng))           # this step's rows of every rank's batch, on this rank's slice
        pieces = be.empty((cap * G,))
        full = be.empty((cap, G))
        Xb = be.empty((cap, ldx))
        recvX = be.empty((R * cap * max(kw, 1),))

        def views(flat, nrow_of, width_of):
            out, off = [], 0
            for q in range(R):
                n = nrow_of(q) * width_of(q)
                out.append(flat[off:off + n].view(nrow_of(q), width_of(q)))
                off += n
            return out
        first = True
        for t in range(nsteps):
            bat = [p[t] if t < len(p) else (0, 0) for p in plans]
            nrow = [hi - lo for lo, hi in bat]
            send = []
            for q, (lo, hi) in enumerate(bat):
                blockrows = rows_loc[q * cap:q * cap + (hi - lo)]
                if hi > lo:
                    self._bj_rows_range(lo, hi, blockrows)
                    if probe is not None:
                        be.rows_combine(probe[0][:, lo:hi], blockrows, probe[1], accumulate=not first)
                        first = False
                send.append(blockrows)
            mine = nrow[rk]
            recv = views(pieces, lambda q: mine, lambda q: slices[q][1] - slices[q][0])
            comm.all_to_all(recv, send)
            rows_full = full[:mine]
            for (s0, s1), piece in zip(slices, recv):
                rows_full[:, s0:s1] = piece
            if mine:
                be.spectral_rows(rows_full, mesh, plan['idx'], plan['scale'], Xb[:mine], batch=mine)
            got = views(recvX, lambda q: nrow[q], lambda q: kw)
            comm.all_to_all(got, [Xb[:mine, k0:k1].contiguous() for k0, k1 in ks])
            for q in range(R):
                if nrow[q]:
                    Xloc[bat[q][0]:bat[q][1]] = got[q]
        del pieces, full, Xb, recvX, rows_loc
        W.zero_()
        if kw > 0:
            for b0 in range(0, P, 512):
                b1 = min(P, b0 + 512)
                be.gemm_nt(Xloc[b0:b1], Xloc[b0:], W[b0:b1, b0:], alpha=1.0)
        del Xloc
        comm.all_reduce_sum(W)
        be.symmetrize_upper(W)
        self._bj_finish_sharded(st['Afac'], st['Dblk'], st['ip_off'], W)

    def _keep_V_for_robust_k_sharded(self, t0):
        """robust_k on the grid-sharded build: the fit rows on this rank's slice become V = conv(Theta)[:, S_r], in place -
        the same two all-to-alls around the row convolution as S4, without the W product."""
        if not self.robust_k:
            return
        cell, be, comm = self.cell, self.backend, self.comm
        theta = self._fit_state['theta']
        P, ng = theta.shape
        R, rk = comm.size, comm.rank
        mesh = np.asarray(self.mesh, dtype=np.int32)
        G = int(np.prod(mesh))
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        slices = [comm.split_range(G, r) for r in range(R)]
        rows = [comm.split_range(P, r) for r in range(R)]
        nb = self.fft_batch or _default_fft_batch(G, max(1, P // R))
        nsteps = max(-(-(hi - lo) // nb) for lo, hi in rows)
        for t in range(nsteps):
            bat = [(min(lo + t * nb, hi), min(lo + (t + 1) * nb, hi)) for lo, hi in rows]
            nrow = [hi - lo for lo, hi in bat]
            send = [theta[lo:hi] for lo, hi in bat]
            recv = [be.empty((nrow[rk], s1 - s0)) for s0, s1 in slices]
            comm.all_to_all(recv, send)
            full = be.empty((nrow[rk], G))
            for (s0, s1), piece in zip(slices, recv):
                full[:, s0:s1] = piece
            del recv
            if nrow[rk]:
                be.coulomb_rows(full, mesh, a, max(1, nrow[rk]))
            send = [full[:, s0:s1].contiguous() for s0, s1 in slices]
            recv = [be.empty((nrow[q], ng)) for q in range(R)]
            comm.all_to_all(recv, send)
            del full, send
            for q in range(R):
                if nrow[q]:
                    theta[bat[q][0]:bat[q][1]] = recv[q]          # rows bat_q were sent in this step's first all-to-all
            del recv
        self._V = theta
        self._fit_state = None
        self._tick('S5_conv_for_robust_k', t0)

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
            if self.robust_k:
                self._robust_k_correction(d_dm, d_vk)       # K1 over this rank's grid slice, all-reduced inside
            if exxdiv == 'ewald':
                self._add_ewald_exxdiv(d_dm, d_vk)
            t0 = self._tick('S7_get_k', t0)
            vk = be.to_host(d_vk).reshape(out_shape)
        return vj, vk
