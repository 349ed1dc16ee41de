"""A checker backend with the method set of pyscf_isdf_amd.backend.HipBackend, computing every stage
with the CPU oracle on CPU torch tensors.  TEST-ONLY: lets the host driver's orchestration and the
multi-rank sharding logic be exercised without a GPU (gloo, world_size 2).  The product never uses it."""
import numpy as np
import scipy.linalg
import torch
from oracle import ao as oao, isdf as oisdf, kisdf as okisdf, pbc_tools as tools


class OracleBackend:
    name = 'oracle-cpu'
    device = torch.device('cpu')
    omega = None

    rc = None

    def set_coulomb_omega(self, omega):
        self.omega = omega or None

    def set_coulomb_cutoff(self, rc):
        self.rc = rc or None

    ws = None

    def set_coulomb_ws(self, ws):
        self.ws = ws

    def set_option(self, key, value):
        pass

    def empty(self, shape, dtype=torch.float64):
        return torch.zeros(shape, dtype=dtype)

    zeros = empty

    def to_device(self, a, dtype=None):
        t = torch.as_tensor(np.ascontiguousarray(a))
        return t.to(dtype) if dtype is not None else t

    def to_host(self, t):
        return t.detach().numpy().copy()

    def new_stream(self):
        return None

    def on_stream(self, stream):
        import contextlib
        return contextlib.nullcontext()

    def record_event(self):
        return None

    def wait_event(self, ev):
        pass

    def synchronize(self):
        pass

    def free_bytes(self):
        return 1 << 60

    def empty_cache(self):
        pass

    def prof_enable(self, on=True):
        pass

    def prof_reset(self):
        pass

    def prof_results(self):
        return {}

    # ---- stages ----
    def eval_ao(self, atm, bas, env, Ls, rcut, coords_soa, ao):
        v = oao.eval_ao(atm, bas, env, coords_soa.numpy().T, Ls, rcut, rule='point')
        ao[:, :v.shape[0]] = torch.from_numpy(np.ascontiguousarray(v.T))

    # ---- multigrid (half spectra, numpy rfftn / irfftn) ----
    @staticmethod
    def _dense_index(n, N):
        f = np.fft.fftfreq(n, 1. / n).astype(np.int64)
        return np.where(f >= 0, f, f + N)

    def uniform_grid(self, mesh, a):
        from oracle import multigrid as omg
        return torch.from_numpy(np.ascontiguousarray(omg.uniform_grids(np.asarray(a, dtype=float), mesh).T))

    def rho_pair(self, aoA, aoB, ng, dm, rho):
        A, B = aoA.numpy()[:, :ng], aoB.numpy()[:, :ng]
        for i in range(dm.shape[0]):
            rho[i, :ng] = torch.from_numpy(np.einsum('hg,hg->g', A, dm[i].numpy().dot(B)))

    def mg_embed_density(self, field, mesh_sub, scale, spec, mesh, accumulate=True):
        # the reference's placement (multigrid.py:669-673: level spectrum at the fftfreq indices of the dense mesh, Nyquist entries
        # at -n/2 only) followed by the Hermitian part (R(f) + conj(R(-f)))/2 - the spectrum of the real field the reference keeps
        n, N = [int(x) for x in mesh_sub], [int(x) for x in mesh]
        sub = np.fft.fftn(field.numpy().reshape(-1, *n), axes=(1, 2, 3)) * scale
        raw = np.zeros((len(sub), N[0], N[1], N[2]), dtype=np.complex128)
        jx, jy, jz = (self._dense_index(n[i], N[i]) for i in range(3))
        raw[:, jx[:, None, None], jy[:, None], jz] = sub
        neg = raw[np.ix_(np.arange(len(sub)), (-np.arange(N[0])) % N[0], (-np.arange(N[1])) % N[1], (-np.arange(N[2])) % N[2])].conj()
        herm = (0.5 * (raw + neg))[..., :N[2] // 2 + 1]
        full = spec.numpy().reshape(-1, N[0], N[1], N[2] // 2 + 1)
        if accumulate:
            full += herm
        else:
            full[...] = herm

    def mg_restrict_potential(self, spec, mesh, mesh_sub, scale, field):
        # the dense half spectrum completed to the full Hermitian one, the reference's pick of the fftfreq entries, its inverse
        # transform on the level mesh and .real (multigrid.py:905-915)
        n, N = [int(x) for x in mesh_sub], [int(x) for x in mesh]
        half = spec.numpy().reshape(-1, N[0], N[1], N[2] // 2 + 1)
        dense = np.fft.fftn(np.fft.irfftn(half, s=N, axes=(1, 2, 3)), axes=(1, 2, 3))
        jx, jy, jz = (self._dense_index(n[i], N[i]) for i in range(3))
        sub = dense[:, jx[:, None, None], jy[:, None], jz]
        out = np.fft.ifftn(sub, axes=(1, 2, 3)).real * (np.prod(n) * scale)
        field.copy_(torch.from_numpy(np.ascontiguousarray(out.reshape(field.shape))))

    def mg_coulomb_kernel(self, spec, mesh, a):
        N = [int(x) for x in mesh]
        c = tools.get_coulG(np.asarray(a, dtype=float), np.asarray(N), omega=self.omega, rc=self.rc).reshape(N)
        mirror = c[np.ix_((-np.arange(N[0])) % N[0], (-np.arange(N[1])) % N[1], (-np.arange(N[2])) % N[2])]
        half = (0.5 * (c + mirror))[:, :, :N[2] // 2 + 1]
        full = spec.numpy().reshape(-1, N[0], N[1], N[2] // 2 + 1)
        full *= half

    def lda_exchange(self, rho, exc, vxc):
        from oracle import multigrid as omg
        e, v = omg.slater_exchange(rho.numpy())
        exc.copy_(torch.from_numpy(e))
        vxc.copy_(torch.from_numpy(v))

    def lda_vwn_add(self, rho, exc, vxc):
        from oracle import multigrid as omg
        e, v = omg.vwn_correlation(rho.numpy())
        exc += torch.from_numpy(e)
        vxc += torch.from_numpy(v)

    def gga_b88(self, rho, grad, exc, vrho, w):
        from oracle import multigrid as omg
        e, vr, ww = omg.b88_exchange(rho.numpy(), grad.numpy())
        exc.copy_(torch.from_numpy(e))
        vrho.copy_(torch.from_numpy(vr))
        w.copy_(torch.from_numpy(ww))

    def lda_exchange_fxc(self, rho, fxc):
        from oracle import multigrid as omg
        fxc.copy_(torch.from_numpy(omg.slater_exchange_fxc(rho.numpy())))

    def dot(self, x, y=None):
        return float(x.numpy().sum() if y is None else x.numpy().dot(y.numpy()))

    def eval_ao_deriv1(self, atm, bas, env, Ls, rcut, coords_soa, ao4):
        v = oao.eval_ao_deriv1(atm, bas, env, coords_soa.numpy().T, Ls, rcut)           # (4, G, nao)
        ao4[:, :, :v.shape[1]] = torch.from_numpy(np.ascontiguousarray(v.transpose(0, 2, 1)))

    def eval_ao_k_deriv1(self, atm, bas, env, Ls, rcut, kpt, periodic_part, coords_soa, out_re, out_im):
        c = coords_soa.numpy().T
        v = np.asarray(oao.eval_ao_deriv1(atm, bas, env, c, Ls, rcut, kpt=np.asarray(kpt, dtype=float)), dtype=complex)   # (4, G, nao)
        if periodic_part:
            v = v * np.exp(-1j * c.dot(kpt))[None, :, None]
        v = np.ascontiguousarray(v.transpose(0, 2, 1))
        out_re[:, :, :len(c)] = torch.from_numpy(np.ascontiguousarray(v.real))
        out_im[:, :, :len(c)] = torch.from_numpy(np.ascontiguousarray(v.imag))

    def gather_cols(self, src, idx, dst):
        dst[:, :idx.numel()] = src[:, idx]

    def block_row_absmax(self, src, blk_off):
        a = np.abs(src.numpy())
        return np.array([[a[r, blk_off[b]:blk_off[b + 1]].max() if blk_off[b + 1] > blk_off[b] else 0.0 for b in range(len(blk_off) - 1)]
                         for r in range(a.shape[0])])

    def partition_by_atom(self, coords, atom_coords, a, tie_atol=1e-9):
        from pyscf_isdf_amd._common import partition_grid_by_atom
        return partition_grid_by_atom(np.asarray(coords), np.asarray(atom_coords), np.asarray(a), tie_atol)

    def select_ip(self, ao, blk_off, nip, tol, tie_rtol, L, piv):
        rank = np.zeros(len(nip), dtype=np.int32)
        A = ao.numpy()
        for b in range(len(nip)):
            if nip[b] == 0:
                continue
            p, Lb = oisdf.select_ip(A[:, blk_off[b]:blk_off[b + 1]], int(nip[b]), tol=tol, tie_rtol=tie_rtol)
            rank[b] = len(p)
            piv[b, :len(p)] = torch.from_numpy(p)
            L[:len(p), blk_off[b]:blk_off[b + 1]] = torch.from_numpy(Lb)
        return rank

    def select_ip_gram(self, A, nip, tol, tie_rtol, piv, panel=0):
        p, _ = oisdf.pivoted_cholesky_gram(A.numpy(), int(nip), tol=tol, tie_rtol=tie_rtol)
        piv[:len(p)] = torch.from_numpy(p)
        return len(p)

    def fit_from_chol(self, L, k, m, piv):
        th = oisdf.fit_theta(L.numpy()[:k, :m], piv.numpy()[:k])
        L[:k, :m] = torch.from_numpy(th)

    def fit_prepare(self, ao, ip, reg_rel, aoP, chol):
        a = ao.numpy()[:, ip.numpy()]
        aoP.copy_(torch.from_numpy(np.ascontiguousarray(a.T)))
        A = a.T.dot(a) ** 2
        A[np.diag_indices(len(A))] += reg_rel * A.diagonal().max()
        chol.copy_(torch.from_numpy(np.linalg.cholesky(A)))      # lower factor, row-major
        return reg_rel

    def fit_apply(self, chol, aoP, ao, ng, theta, forward_only=False):
        B = aoP.numpy().dot(ao.numpy()[:, :ng]) ** 2
        if forward_only:
            theta[:, :ng] = torch.from_numpy(scipy.linalg.solve_triangular(chol.numpy(), B, lower=True))
        else:
            theta[:, :ng] = torch.from_numpy(scipy.linalg.cho_solve((chol.numpy(), True), B))

    def gather_aoP(self, ao, ip, aoP):
        aoP.copy_(torch.from_numpy(np.ascontiguousarray(ao.numpy()[:, ip.numpy()].T)))

    def gram_sq(self, aoP, A, nh=0):
        ap = aoP.numpy()
        a = ap.dot(ap.T) ** 2
        if nh:
            a += np.hstack([ap[:, nh:], -ap[:, :nh]]).dot(ap.T) ** 2
        A.copy_(torch.from_numpy(a))

    def pair_gram_rows(self, aoP, ao, ng, B, nh=0):
        ap = aoP.numpy()
        x = ao.numpy()[:, :ng]
        b = ap.dot(x) ** 2
        if nh:
            b += np.hstack([ap[:, nh:], -ap[:, :nh]]).dot(x) ** 2
        B[:, :ng] = torch.from_numpy(b)

    def gram_prod(self, aoP, psiP, A):
        ap, pp = aoP.numpy(), psiP.numpy()
        A.copy_(torch.from_numpy(ap.dot(ap.T) * pp.dot(pp.T)))

    def pair_prod_rows(self, aoP, psiP, ao, psi, ng, B):
        B[:, :ng] = torch.from_numpy(aoP.numpy().dot(ao.numpy()[:, :ng]) * psiP.numpy().dot(psi.numpy()[:, :ng]))

    def factor_solve_half(self, fac, backward, X):
        X.copy_(torch.from_numpy(scipy.linalg.solve_triangular(fac.numpy(), X.numpy(), lower=True, trans='T' if backward else 'N')))

    def block_chol(self, A, blk_off, shift_rel, D):
        a = A.numpy()
        d = np.zeros_like(a)
        worst = 0.0
        for b in range(len(blk_off) - 1):
            s = slice(blk_off[b], blk_off[b + 1])
            if s.stop == s.start:
                continue
            reg = shift_rel
            for attempt in range(6):
                try:
                    d[s, s] = np.linalg.cholesky(a[s, s] + reg * a.diagonal().max() * np.eye(s.stop - s.start))
                    worst = max(worst, reg)
                    break
                except np.linalg.LinAlgError:
                    reg = reg * 100.0 if reg > 0 else 1e-14
            else:
                raise np.linalg.LinAlgError('block %d not positive definite' % b)
        D.copy_(torch.from_numpy(d))
        return worst

    def block_solve(self, D, blk_off, side, trans, X):
        d, x = D.numpy(), X.numpy()
        for b in range(len(blk_off) - 1):
            s = slice(blk_off[b], blk_off[b + 1])
            if s.stop == s.start:
                continue
            if side == 0:
                x[s] = scipy.linalg.solve_triangular(d[s, s], x[s], lower=True, trans='T' if trans else 'N')
            else:
                # X op(D)^-1 = (op(D)^-T X^T)^T
                x[:, s] = scipy.linalg.solve_triangular(d[s, s], x[:, s].T, lower=True, trans='N' if trans else 'T').T

    def block_invert(self, D, blk_off, Dinv):
        d = D.numpy()
        out = np.zeros_like(d)
        for b in range(len(blk_off) - 1):
            s = slice(blk_off[b], blk_off[b + 1])
            if s.stop > s.start:
                out[s, s] = scipy.linalg.solve_triangular(d[s, s], np.eye(s.stop - s.start), lower=True)
        Dinv.copy_(torch.from_numpy(out))

    def block_apply(self, Dinv, blk_off, X):
        d, x = Dinv.numpy(), X.numpy()
        for b in range(len(blk_off) - 1):
            s = slice(blk_off[b], blk_off[b + 1])
            if s.stop > s.start:
                x[s] = d[s, s].dot(x[s])

    def pair_rows_block_apply(self, aoP, ao, ng, Dinv, blk_off, B):
        self.pair_gram_rows(aoP, ao, ng, B, 0)
        self.block_apply(Dinv, blk_off, B[:, :ng])

    def shift_diag(self, A, shift_rel):
        a = A.numpy()
        a[np.diag_indices(len(a))] += shift_rel * a.diagonal().max()

    def chol_inplace(self, A, shift_rel, scratch=None):
        a0 = A.numpy().copy()
        reg = shift_rel
        for attempt in range(5):
            a = a0.copy()
            a[np.diag_indices(len(a))] += reg * a.diagonal().max()
            try:
                A.copy_(torch.from_numpy(np.linalg.cholesky(a)))
                return reg
            except np.linalg.LinAlgError:
                reg = reg * 100.0 if reg > 0 else 1e-14
        raise np.linalg.LinAlgError('not positive definite')

    def factor_solve(self, fac, X):
        X.copy_(torch.from_numpy(scipy.linalg.cho_solve((fac.numpy(), True), X.numpy())))

    def bj_probe_rows(self, T, fac, D, blk_off, Yp, ng, F):
        self.block_solve(D, blk_off, 1, 1, T)
        t = scipy.linalg.cho_solve((fac.numpy(), True), T.numpy().T).T
        T.copy_(torch.from_numpy(np.ascontiguousarray(t)))
        F[:, :ng] = torch.from_numpy(t.dot(Yp.numpy()[:, :ng]))

    def bj_probe_vectors(self, T, fac, D, blk_off):
        self.block_solve(D, blk_off, 1, 1, T)
        t = scipy.linalg.cho_solve((fac.numpy(), True), T.numpy().T).T
        T.copy_(torch.from_numpy(np.ascontiguousarray(t)))

    def rows_combine(self, E, Y, F, accumulate=False):
        r = torch.from_numpy(E.numpy().dot(Y.numpy()))
        if accumulate:
            F += r
        else:
            F.copy_(r)

    def gather_T(self, L, k, piv, T):
        T.copy_(torch.from_numpy(np.triu(L.numpy()[:k][:, piv.numpy()[:k]])))

    def W_from_factor(self, F, kind, M):
        if kind == 2:                                          # M <- U^-T M U^-1 with A = U^T U, U = F^T
            Lr = F.numpy()
            Z = scipy.linalg.solve_triangular(Lr, M.numpy(), lower=True)
            M.copy_(torch.from_numpy(scipy.linalg.solve_triangular(Lr, Z.T, lower=True).T))
            return
        S = F.numpy().T if kind == 0 else F.numpy()          # upper triangular S, Theta = S^-1 Y
        Z = scipy.linalg.solve_triangular(S, M.numpy(), lower=False)
        M.copy_(torch.from_numpy(scipy.linalg.solve_triangular(S, Z.T, lower=False).T))

    def fit_global(self, ao, ngrids, ip, reg_rel, theta, aoP):
        chol = torch.zeros((ip.numel(), ip.numel()), dtype=torch.float64)
        self.fit_prepare(ao, ip, reg_rel, aoP, chol)
        self.fit_apply(chol, aoP, ao, ngrids, theta)
        return reg_rel

    def coulomb_rows(self, rows, mesh, a, batch, out=None):
        out = rows if out is None else out
        out.copy_(torch.from_numpy(oisdf.coulomb_V(rows.numpy(), a, mesh, self.omega, self.rc, self.ws)))

    def coulomb_W(self, theta, mesh, a, row0, nrows, batch, W, upper_only=False):
        G = int(np.prod(mesh))
        w = abs(np.linalg.det(a)) / G
        th = theta.numpy()
        V = oisdf.coulomb_V(th[row0:row0 + nrows], a, mesh, self.omega, self.rc, self.ws)
        W[row0:row0 + nrows, :th.shape[0]] = torch.from_numpy(w * V.dot(th.T))

    # spectral form of W (fit_route.FitRouteMixin._spectral_plan / _finish_W_spectral): numpy counterparts
    def spectral_supported(self, mesh, batch=512):
        return True

    def coulG_half(self, mesh, a):
        mesh = [int(x) for x in mesh]
        G = int(np.prod(mesh))
        c = tools.get_coulG(a, mesh, omega=self.omega, rc=self.rc, ws=self.ws).reshape(mesh)
        flip = c[np.ix_(*[(-np.arange(n)) % n for n in mesh])]
        return (0.5 * (c + flip) / G)[:, :, :mesh[2] // 2 + 1].copy()

    def spectral_rows(self, rows, mesh, idx, scale, out, batch=512):
        mesh = [int(x) for x in mesh]
        z = np.fft.rfftn(rows.numpy().reshape(-1, *mesh), axes=(1, 2, 3)).reshape(rows.shape[0], -1)
        v = z[:, idx.numpy()] * scale.numpy()
        o = out.numpy()
        o[:] = 0.0
        o[:, 0:2 * v.shape[1]:2] = v.real
        o[:, 1:2 * v.shape[1]:2] = v.imag

    def symmetrize_upper(self, W):
        w = W.numpy()
        iu = np.triu_indices(len(w), 1)
        w.T[iu] = w[iu]

    def symmetrize_mean(self, W, antisymmetric=False):
        W.copy_((W - W.T) / 2 if antisymmetric else (W + W.T) / 2)

    def gemm_nn(self, A, B, C, alpha=1.0, beta=0.0):
        C.copy_(torch.from_numpy(alpha * A.numpy().dot(B.numpy()) + beta * C.numpy()))

    def hadamard_rows(self, X, Y):
        X.mul_(Y)

    def gemm_nt(self, A, B, C, alpha=1.0, beta=0.0, kscale=None):
        b = B.numpy() if kscale is None else B.numpy() * kscale.numpy()
        C.copy_(torch.from_numpy(alpha * A.numpy().dot(b.T) + beta * C.numpy()))

    def rho(self, ao, ng, dm, rho):
        a = ao.numpy()[:, :ng]
        for i in range(dm.shape[0]):
            rho[i, :ng] = torch.from_numpy(np.einsum('ig,ig->g', dm[i].numpy().dot(a), a))

    def coulomb_potential(self, rho, mesh, a):
        G = int(np.prod(mesh))
        w = abs(np.linalg.det(a)) / G
        rho.copy_(torch.from_numpy(w * oisdf.coulomb_V(rho.numpy(), a, mesh, self.omega)))

    def vj_from_vR(self, ao, ng, vR, vj):
        a = ao.numpy()[:, :ng]
        for i in range(vR.shape[0]):
            vj[i] = torch.from_numpy((a * vR[i, :ng].numpy()).dot(a.T))

    def get_j(self, ao, ngrids, mesh, a, dm, vj):
        vj.copy_(torch.from_numpy(oisdf.get_j(ao.numpy()[:, :ngrids], dm.numpy(), a, mesh, self.omega)))

    def get_k(self, aoP, W, row0, nrows, dm, vk):
        ap, w = aoP.numpy(), W.numpy()
        for i in range(dm.shape[0]):
            M = ap[row0:row0 + nrows].dot(dm[i].numpy()).dot(ap.T) * w[row0:row0 + nrows]
            vk[i] = torch.from_numpy(ap[row0:row0 + nrows].T.dot(M).dot(ap))

    def get_k_exact(self, ao, ngrids, C, mesh, a, i0, ni, max_rows, vk):
        from oracle import fftdf
        aoR = np.ascontiguousarray(ao.numpy()[:, :ngrids].T)
        c = C.numpy()
        k = fftdf.get_k(aoR, c.dot(c.T), a, mesh, mo_coeff=c, mo_occ=np.ones(c.shape[1]))
        vk[i0:i0 + ni] = torch.from_numpy(np.ascontiguousarray(k[i0:i0 + ni]))

    # ---- k-points ----
    def eval_ao_k(self, atm, bas, env, Ls, rcut, kpt, periodic_part, coords_soa, out_re, out_im):
        c = coords_soa.numpy().T
        v = np.asarray(oao.eval_ao(atm, bas, env, c, Ls, rcut, kpts=np.reshape(kpt, (1, 3)), rule='point')[0], dtype=complex)
        if periodic_part:
            v = v * np.exp(-1j * c.dot(kpt))[:, None]
        out_re[:, :len(c)] = torch.from_numpy(np.ascontiguousarray(v.real.T))
        out_im[:, :len(c)] = torch.from_numpy(np.ascontiguousarray(v.imag.T))

    def select_ip_cplx(self, X, nh, blk_off, nip, tol, tie_rtol, L, piv):
        rank = np.zeros(len(nip), dtype=np.int32)
        A = X.numpy()
        for b in range(len(nip)):
            if nip[b] == 0:
                continue
            p, Lb = okisdf.select_ip(A[:, blk_off[b]:blk_off[b + 1]], int(nip[b]), tol=tol, tie_rtol=tie_rtol)
            rank[b] = len(p)
            piv[b, :len(p)] = torch.from_numpy(p)
            L[:len(p), blk_off[b]:blk_off[b + 1]] = torch.from_numpy(Lb)
        return rank

    def fit_prepare_cplx(self, X, nh, ip, reg_rel, aoP, chol):
        x = X.numpy()
        xp = x[:, ip.numpy()]
        xr = np.vstack([x[nh:], -x[:nh]])
        aoP.copy_(torch.from_numpy(np.ascontiguousarray(xp.T)))
        A = xp.T.dot(xp) ** 2 + xr[:, ip.numpy()].T.dot(xp) ** 2
        A[np.diag_indices(len(A))] += reg_rel * A.diagonal().max()
        chol.copy_(torch.from_numpy(np.linalg.cholesky(A)))
        return reg_rel

    def fit_apply_cplx(self, chol, aoP, nh, X, ng, theta, forward_only=False):
        x = X.numpy()[:, :ng]
        xr = np.vstack([x[nh:], -x[:nh]])
        ap = aoP.numpy()
        B = ap.dot(x) ** 2 + ap.dot(xr) ** 2
        if forward_only:
            theta[:, :ng] = torch.from_numpy(scipy.linalg.solve_triangular(chol.numpy(), B, lower=True))
        else:
            theta[:, :ng] = torch.from_numpy(scipy.linalg.cho_solve((chol.numpy(), True), B))

    def coulG_q(self, mesh, a, q, omega=None, wrap_around=True, out=None):
        t = torch.from_numpy(tools.get_coulG(a, mesh, q, wrap_around=wrap_around, omega=omega, rc=self.rc, ws=self.ws))
        if out is None:
            return t
        out.copy_(t)
        return out

    def coulomb_Wq(self, theta, mesh, coulG, weight, row0, nrows, batch, Wre, Wim, upper_only=False):
        th = theta.numpy()
        V = tools.ifft(tools.fft(th[row0:row0 + nrows], mesh) * coulG.numpy(), mesh)
        M = weight * V.dot(th.T)
        Wre[row0:row0 + nrows] = torch.from_numpy(np.ascontiguousarray(M.real))
        Wim[row0:row0 + nrows] = torch.from_numpy(np.ascontiguousarray(M.imag))

    def symmetrize_hermitian(self, Wre, Wim):
        a, b = Wre.numpy(), Wim.numpy()
        iu = np.triu_indices(len(a), 1)
        a.T[iu] = a[iu]
        b.T[iu] = -b[iu]

    def finish_Wq(self, Wre, Wim, phase, Wc):
        ph = phase.numpy()
        Wc.copy_(torch.from_numpy((Wre.numpy() + 1j * Wim.numpy()) * ph[:, None] * ph.conj()[None, :]))

    def get_k_pair(self, A1, A2, D2, Wq, scale, vk):
        a1, a2 = A1.numpy(), A2.numpy()
        X = a2.dot(D2.numpy()).dot(a2.conj().T) * Wq.numpy()
        vk += torch.from_numpy(scale * a1.conj().T.dot(X).dot(a1))

    def coulomb_rows_q(self, rows, mesh, coulG, out_re, out_im):
        v = tools.ifft(tools.fft(rows.numpy(), mesh) * coulG.numpy(), mesh)
        out_re.copy_(torch.from_numpy(np.ascontiguousarray(v.real)))
        out_im.copy_(torch.from_numpy(np.ascontiguousarray(v.imag)))

    def nyquist_spectra(self, rows, mesh, axis, out_re, out_im):
        n = [int(x) for x in mesh]
        f = np.fft.fftn(rows.numpy()[:, :int(np.prod(n))].reshape(-1, *n), axes=(1, 2, 3))
        pl = np.take(f, n[axis] // 2, axis=1 + axis).reshape(len(f), -1)
        out_re.copy_(torch.from_numpy(np.ascontiguousarray(pl.real)))
        out_im.copy_(torch.from_numpy(np.ascontiguousarray(pl.imag)))

    def zhadamard_planes(self, Ar, Ai, Br, Bi):
        z = (Ar.numpy() + 1j * Ai.numpy()) * (Br.numpy() + 1j * Bi.numpy())
        Ar.copy_(torch.from_numpy(np.ascontiguousarray(z.real)))
        Ai.copy_(torch.from_numpy(np.ascontiguousarray(z.imag)))

    def rho_k(self, ur, ui, ng, DTr, DTi, scale, rho):
        u = ur.numpy()[:, :ng] + 1j * ui.numpy()[:, :ng]
        T = (DTr.numpy() + 1j * DTi.numpy()).dot(u)
        rho[0, :ng] += torch.from_numpy(scale * np.einsum('jg,jg->g', T, u.conj()).real)

    def vj_k(self, ur, ui, ng, vR, vj_re, vj_im):
        u = ur.numpy()[:, :ng] + 1j * ui.numpy()[:, :ng]
        v = (u.conj() * vR.numpy().reshape(-1)[:ng]).dot(u.T)
        vj_re.copy_(torch.from_numpy(np.ascontiguousarray(v.real)))
        vj_im.copy_(torch.from_numpy(np.ascontiguousarray(v.imag)))
