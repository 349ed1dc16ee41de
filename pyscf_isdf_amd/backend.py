"""Device backend: the stage functions of include/mi355_isdf.h on torch-owned HBM buffers.

torch is plumbing here (device memory, streams, torch.distributed); all arithmetic happens inside
libmi355_isdf.so.  The method set below is the contract between the host driver
(pyscf_isdf_amd/isdf.py) and a backend; tests exercise the host driver's sharding logic on CPU by
passing a checker backend with the same methods (tests/oracle_backend.py) — the product never does.
"""
import ctypes
import os
import numpy as np
import torch
from . import lib as _lib

_vp = ctypes.c_void_p


def _np_ptr(a):
    return a.ctypes.data_as(_vp)


class HipBackend:
    name = 'hip-gfx950'

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise _lib.IsdfError('no GPU visible to torch; the ISDF path has no CPU fallback')
        self.device = torch.device('cuda', device)
        torch.cuda.set_device(self.device)
        self.handle = _lib.Handle(device)
        if os.environ.get('ISDF_TRSM') == 'subst':        # cross-check path (include/mi355_isdf.h)
            self.set_option('trsm_substitution', 1)
        if os.environ.get('ISDF_OWN_FFT') in ('0', '1', '2'):   # A/B runs: 0 hipFFT, 1 the five-pass own FFT, 2 the three-pass plane form
            self.set_option('own_fft', int(os.environ['ISDF_OWN_FFT']))
        if os.environ.get('ISDF_GEMM_NN') == '1':         # A/B runs: own MFMA NN kernel for the pair-density rows instead of rocBLAS
            self.set_option('gemm_nn_own', 1)

    # ---- memory -------------------------------------------------------------------------------
    def empty(self, shape, dtype=torch.float64):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype=torch.float64):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def to_device(self, a, dtype=None):
        a = np.ascontiguousarray(a)
        if not a.flags.writeable:                  # torch refuses to wrap read-only memory silently
            a = a.copy()
        t = torch.as_tensor(a)
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device)

    def to_host(self, t):
        return t.detach().cpu().numpy()

    kernel_epoch = 0          # bumped whenever the Coulomb kernel state of the handle changes (plans derived from the table key on it)

    def set_coulomb_omega(self, omega):
        self.kernel_epoch += 1
        self.handle.call('isdf_set_coulomb_omega', float(omega or 0.0))

    def set_coulomb_cutoff(self, rc):
        self.kernel_epoch += 1
        self.handle.call('isdf_set_coulomb_cutoff', float(rc or 0.0))

    def set_coulomb_ws(self, ws):
        """ws: dict(alpha, a, mesh, maxq, vq) from pbc_tools.wigner_seitz_kernel, or None to switch the kernel off."""
        self.kernel_epoch += 1
        if ws is None:
            self.handle.call('isdf_set_coulomb_ws', 0.0, None, None, None, None)
            self._ws_table = None
            return
        ak = np.ascontiguousarray(ws['a'], dtype=np.float64)
        mesh = np.ascontiguousarray(ws['mesh'], dtype=np.int32)
        maxq = np.ascontiguousarray(ws['maxq'], dtype=np.float64)
        self._ws_table = self.to_device(np.ascontiguousarray(ws['vq'], dtype=np.float64))     # kept alive while set
        self.handle.call('isdf_set_coulomb_ws', float(ws['alpha']), _np_ptr(ak), _np_ptr(mesh), _np_ptr(maxq),
                         self._p(self._ws_table))

    def set_option(self, key, value):
        self.kernel_epoch += 1
        self.handle.call('isdf_set_option', key.encode(), int(value))

    def free_bytes(self):
        """Device memory obtainable right now: free on the device + free inside torch's cache."""
        free, _ = torch.cuda.mem_get_info(self.device)
        return int(free + torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device))

    def empty_cache(self):
        torch.cuda.empty_cache()

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    # ---- streams (overlap of the exchange / FFT pipeline with the MFMA products in the grid-sharded build) ----
    def new_stream(self):
        return torch.cuda.Stream(device=self.device)

    def on_stream(self, stream):
        """Context: library calls, torch ops and torch.distributed collectives inside are ordered on ``stream``."""
        return torch.cuda.stream(stream)

    def record_event(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def wait_event(self, ev):
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def release_workspace(self):
        self.handle.call('isdf_release_workspace')

    # ---- in-library profiling ------------------------------------------------------------------
    def prof_enable(self, on=True):
        self.handle.call('isdf_prof_enable', int(bool(on)))

    def prof_reset(self):
        self.handle.call('isdf_prof_reset')

    def prof_results(self):
        """{kernel name: dict(launches, ms, work)} accumulated since the last reset."""
        out = {}
        n = self.handle.lib.isdf_prof_count(self.handle.h)
        for i in range(n):
            name = ctypes.create_string_buffer(128)
            launches = ctypes.c_int64(0)
            ms = ctypes.c_double(0)
            work = ctypes.c_double(0)
            self.handle.call('isdf_prof_get', i, name, 128, ctypes.byref(launches), ctypes.byref(ms), ctypes.byref(work))
            out[name.value.decode()] = dict(launches=launches.value, ms=ms.value, work=work.value)
        return out

    def _stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.handle.call('isdf_set_stream', _vp(s))

    @staticmethod
    def _p(t):
        assert t.is_contiguous() or t.stride(-1) == 1
        return _vp(t.data_ptr())          # data_ptr() of a view already points at its first element

    # ---- stages -------------------------------------------------------------------------------
    def eval_ao(self, atm, bas, env, Ls, rcut, coords_soa, ao):
        """ao (nao, ld) <- collocation on coords_soa (3, G)."""
        self._stream()
        atm = np.ascontiguousarray(atm, dtype=np.int32)
        bas = np.ascontiguousarray(bas, dtype=np.int32)
        env = np.ascontiguousarray(env, dtype=np.float64)
        Ls = np.ascontiguousarray(Ls, dtype=np.float64)
        rcut = np.ascontiguousarray(rcut, dtype=np.float64)
        G = coords_soa.shape[1]
        assert coords_soa.is_contiguous() and ao.stride(1) == 1 and ao.shape[1] >= G
        self.handle.call('isdf_eval_ao', _np_ptr(atm), len(atm), _np_ptr(bas), len(bas), _np_ptr(env), len(env),
                         _np_ptr(Ls), len(Ls), _np_ptr(rcut), self._p(coords_soa), G, self._p(ao), ao.stride(0))

    def eval_ao_deriv1(self, atm, bas, env, Ls, rcut, coords_soa, ao4):
        """ao4 (4, nao, ld) <- values and x, y, z derivatives on coords_soa (3, G)."""
        self._stream()
        atm = np.ascontiguousarray(atm, dtype=np.int32)
        bas = np.ascontiguousarray(bas, dtype=np.int32)
        env = np.ascontiguousarray(env, dtype=np.float64)
        Ls = np.ascontiguousarray(Ls, dtype=np.float64)
        rcut = np.ascontiguousarray(rcut, dtype=np.float64)
        G = coords_soa.shape[1]
        assert coords_soa.is_contiguous() and ao4.dim() == 3 and ao4.shape[0] == 4 and ao4.stride(2) == 1 and ao4.shape[2] >= G
        self.handle.call('isdf_eval_ao_deriv1', _np_ptr(atm), len(atm), _np_ptr(bas), len(bas), _np_ptr(env), len(env),
                         _np_ptr(Ls), len(Ls), _np_ptr(rcut), self._p(coords_soa), G, self._p(ao4), ao4.stride(1), ao4.stride(0))

    def eval_ao_k_deriv1(self, atm, bas, env, Ls, rcut, kpt, periodic_part, coords_soa, out_re, out_im):
        """out_re / out_im (4, nao, ld) views of one buffer <- Bloch sums of values and x, y, z derivatives at ``kpt``."""
        self._stream()
        atm = np.ascontiguousarray(atm, dtype=np.int32)
        bas = np.ascontiguousarray(bas, dtype=np.int32)
        env = np.ascontiguousarray(env, dtype=np.float64)
        Ls = np.ascontiguousarray(Ls, dtype=np.float64)
        rcut = np.ascontiguousarray(rcut, dtype=np.float64)
        kpt = np.ascontiguousarray(kpt, dtype=np.float64)
        G = coords_soa.shape[1]
        assert out_re.dim() == 3 and out_re.shape[0] == 4 and tuple(out_re.shape) == tuple(out_im.shape)
        assert out_re.stride() == out_im.stride() and out_re.stride(2) == 1 and out_re.shape[2] >= G
        self.handle.call('isdf_eval_ao_k_deriv1', _np_ptr(atm), len(atm), _np_ptr(bas), len(bas), _np_ptr(env), len(env),
                         _np_ptr(Ls), len(Ls), _np_ptr(rcut), _np_ptr(kpt), int(bool(periodic_part)), self._p(coords_soa), G,
                         self._p(out_re), self._p(out_im), out_re.stride(1), out_re.stride(0))

    def gather_cols(self, src, idx, dst):
        self._stream()
        assert idx.dtype == torch.int64 and src.stride(1) == 1 and dst.stride(1) == 1
        self.handle.call('isdf_gather_cols', self._p(src), src.shape[0], src.stride(0), self._p(idx), idx.numel(),
                         self._p(dst), dst.stride(0))

    def block_row_absmax(self, src, blk_off):
        """(nrow, nblk) host array: max |src[row, block b's columns]|."""
        self._stream()
        d_off = self.to_device(np.ascontiguousarray(blk_off, dtype=np.int64))
        out = self.empty((src.shape[0], len(blk_off) - 1))
        self.handle.call('isdf_block_row_absmax', self._p(src), src.shape[0], src.stride(0), len(blk_off) - 1, self._p(d_off), self._p(out))
        return self.to_host(out)

    def partition_by_atom(self, coords, atom_coords, a, tie_atol=1e-9):
        """owner (G,) int32 on the host: nearest atom of every grid point (coords: host (G, 3)); the grid goes to the device
        once, the partition runs there (isdf_partition_by_atom)."""
        self._stream()
        coords_soa = self.to_device(np.ascontiguousarray(np.asarray(coords, dtype=np.float64).T))
        atoms = np.ascontiguousarray(atom_coords, dtype=np.float64)
        a = np.ascontiguousarray(a, dtype=np.float64)
        G = coords_soa.shape[1]
        owner = self.empty((G,), dtype=torch.int32)
        self.handle.call('isdf_partition_by_atom', self._p(coords_soa), G, _np_ptr(atoms), len(atoms), _np_ptr(a), float(tie_atol),
                         self._p(owner))
        return self.to_host(owner)

    def select_ip(self, ao, blk_off, nip, tol, tie_rtol, L, piv):
        """Returns rank (np.int32[nblk]); fills L (kmax, ldL) and piv (nblk, kmax) int64 (local indices)."""
        self._stream()
        blk_off = np.ascontiguousarray(blk_off, dtype=np.int64)
        nip = np.ascontiguousarray(nip, dtype=np.int32)
        rank = np.zeros(len(nip), dtype=np.int32)
        assert piv.dtype == torch.int64 and piv.is_contiguous() and ao.stride(1) == 1 and L.stride(1) == 1
        self.handle.call('isdf_select_ip', self._p(ao), ao.shape[0], ao.stride(0), len(nip), _np_ptr(blk_off),
                         _np_ptr(nip), float(tol), float(tie_rtol), self._p(L), L.stride(0), self._p(piv), _np_ptr(rank))
        return rank

    def select_ip_gram(self, A, nip, tol, tie_rtol, piv, panel=0):
        """Pivoted Cholesky of the explicit matrix A (m, m) (destroyed); fills piv (nip,) int64; returns the rank."""
        self._stream()
        assert piv.dtype == torch.int64 and piv.is_contiguous() and A.stride(1) == 1 and A.shape[0] == A.shape[1]
        rank = ctypes.c_int32(0)
        self.handle.call('isdf_select_ip_gram', self._p(A), A.shape[0], A.stride(0), int(nip), float(tol), float(tie_rtol),
                         int(panel), self._p(piv), ctypes.byref(rank))
        return int(rank.value)

    def fit_from_chol(self, L, k, m, piv):
        self._stream()
        self.handle.call('isdf_fit_from_chol', self._p(L), int(k), int(m), L.stride(0), self._p(piv))

    def fit_prepare(self, ao, ip, reg_rel, aoP, chol):
        """Returns the diagonal shift actually used."""
        self._stream()
        assert ip.dtype == torch.int64 and aoP.is_contiguous() and chol.is_contiguous()
        reg = ctypes.c_double(0.0)
        self.handle.call('isdf_fit_prepare', self._p(ao), ao.shape[0], ao.stride(0), self._p(ip), ip.numel(),
                         float(reg_rel), self._p(aoP), self._p(chol), ctypes.byref(reg))
        return reg.value

    def fit_apply(self, chol, aoP, ao, ng, theta, forward_only=False):
        """theta (P, >=ng) <- fit on the ng grid columns ``ao`` (nao, >=ng) starts at
        (forward_only: Y = Lr^-1 B instead of Theta)."""
        self._stream()
        self.handle.call('isdf_fit_apply', self._p(chol), self._p(aoP), aoP.shape[0], aoP.shape[1], self._p(ao),
                         int(ng), ao.stride(0), int(bool(forward_only)), self._p(theta), theta.stride(0))

    # ---- block-Jacobi route (S3c) ----
    def gather_aoP(self, ao, ip, aoP):
        self._stream()
        self.handle.call('isdf_gather_aoP', self._p(ao), ao.shape[0], ao.stride(0), self._p(ip), ip.numel(), self._p(aoP))

    def gram_sq(self, aoP, A, nh=0):
        self._stream()
        self.handle.call('isdf_gram_sq', self._p(aoP), aoP.shape[0], aoP.shape[1], int(nh), self._p(A))

    def pair_gram_rows(self, aoP, ao, ng, B, nh=0):
        self._stream()
        self.handle.call('isdf_pair_gram_rows', self._p(aoP), aoP.shape[0], aoP.shape[1], int(nh), self._p(ao), int(ng),
                         ao.stride(0), self._p(B), B.stride(0))

    # ---- (AO x occupied orbital) pair space: products of two Gram matrices instead of squares ----
    def gram_prod(self, aoP, psiP, A):
        """A (P, P) <- (aoP aoP^T) o (psiP psiP^T)."""
        self._stream()
        assert aoP.is_contiguous() and psiP.is_contiguous() and A.is_contiguous() and aoP.shape[0] == psiP.shape[0]
        self.handle.call('isdf_gram_prod', self._p(aoP), aoP.shape[0], aoP.shape[1], self._p(psiP), psiP.shape[1], self._p(A))

    def pair_prod_rows(self, aoP, psiP, ao, psi, ng, B):
        """B (P, ng) <- (aoP ao) o (psiP psi)."""
        self._stream()
        assert aoP.is_contiguous() and psiP.is_contiguous() and aoP.shape[0] == psiP.shape[0]
        assert ao.stride(1) == 1 and psi.stride(1) == 1 and B.stride(1) == 1
        self.handle.call('isdf_pair_prod_rows', self._p(aoP), aoP.shape[0], aoP.shape[1], self._p(psiP), psiP.shape[1],
                         self._p(ao), ao.stride(0), self._p(psi), psi.stride(0), int(ng), self._p(B), B.stride(0))

    def factor_solve_half(self, fac, backward, X):
        """X (P, n) <- L^-1 X (backward False) or L^-T X (backward True), A = L L^T (fac as stored by chol_inplace)."""
        self._stream()
        self.handle.call('isdf_factor_solve_half', self._p(fac), fac.shape[0], int(bool(backward)), self._p(X), X.shape[1],
                         X.stride(0))

    def block_chol(self, A, blk_off, shift_rel, D):
        self._stream()
        blk_off = np.ascontiguousarray(blk_off, dtype=np.int32)
        used = ctypes.c_double(0.0)
        self.handle.call('isdf_block_chol', self._p(A), A.shape[0], len(blk_off) - 1, _np_ptr(blk_off), float(shift_rel),
                         self._p(D), ctypes.byref(used))
        return used.value

    def block_solve(self, D, blk_off, side, trans, X):
        """side 0: X (P, n) <- op(D)^-1 X; side 1: X (n, P) <- X op(D)^-1."""
        self._stream()
        blk_off = np.ascontiguousarray(blk_off, dtype=np.int32)
        n = X.shape[1] if side == 0 else X.shape[0]
        # D may be the (r0:r1, r0:r1) view of the full block factor: its leading dimension is what the library needs
        self.handle.call('isdf_block_solve', self._p(D), D.stride(0), len(blk_off) - 1, _np_ptr(blk_off), int(side), int(trans),
                         self._p(X), int(n), X.stride(0))

    def block_invert(self, D, blk_off, Dinv):
        """Dinv (P, P) <- blockdiag(D_b^-1)."""
        self._stream()
        blk_off = np.ascontiguousarray(blk_off, dtype=np.int32)
        assert D.is_contiguous() and Dinv.is_contiguous() and Dinv.shape == D.shape
        self.handle.call('isdf_block_invert', self._p(D), D.shape[0], len(blk_off) - 1, _np_ptr(blk_off), self._p(Dinv))

    def block_apply(self, Dinv, blk_off, X):
        """X (rows of the blocks, n) <- Dinv_b X_b in place (MFMA); Dinv may be a diagonal sub-block view."""
        self._stream()
        blk_off = np.ascontiguousarray(blk_off, dtype=np.int32)
        self.handle.call('isdf_block_apply', self._p(Dinv), Dinv.stride(0), len(blk_off) - 1, _np_ptr(blk_off), self._p(X),
                         X.shape[1], X.stride(0))

    def pair_rows_block_apply(self, aoP, ao, ng, Dinv, blk_off, B):
        """B (P, ng) <- Dinv_b (aoP ao)^2 (real mode); Dinv may be a diagonal sub-block view matching aoP's rows."""
        self._stream()
        blk_off = np.ascontiguousarray(blk_off, dtype=np.int32)
        assert aoP.stride(1) == 1 and aoP.stride(0) == aoP.shape[1]
        self.handle.call('isdf_pair_rows_block_apply', self._p(aoP), aoP.shape[0], aoP.shape[1], self._p(ao), int(ng), ao.stride(0),
                         self._p(Dinv), Dinv.stride(0), len(blk_off) - 1, _np_ptr(blk_off), self._p(B), B.stride(0))

    def shift_diag(self, A, shift_rel):
        self._stream()
        self.handle.call('isdf_shift_diag', self._p(A), A.shape[0], float(shift_rel))

    def chol_inplace(self, A, shift_rel, scratch=None):
        self._stream()
        reg = ctypes.c_double(0.0)
        assert scratch is None or scratch.numel() >= A.numel()
        self.handle.call('isdf_chol_inplace', self._p(A), A.shape[0], float(shift_rel),
                         self._p(scratch) if scratch is not None else None, ctypes.byref(reg))
        return reg.value

    def factor_solve(self, fac, X):
        """X (P, n) <- A^-1 X, A = U^T U (fac as stored by fit_prepare / chol_inplace)."""
        self._stream()
        self.handle.call('isdf_factor_solve', self._p(fac), fac.shape[0], self._p(X), X.shape[1], X.stride(0))

    def bj_probe_rows(self, T, fac, D, blk_off, Yp, ng, F):
        """T (n, P) <- A'^-1 D^-1 t_j in place; F (n, ng) <- T Yp."""
        self._stream()
        blk_off = np.ascontiguousarray(blk_off, dtype=np.int32)
        self.handle.call('isdf_bj_probe_rows', self._p(T), T.shape[0], self._p(fac), self._p(D), D.shape[0],
                         len(blk_off) - 1, _np_ptr(blk_off), self._p(Yp), int(ng), Yp.stride(0), self._p(F), F.stride(0))

    def bj_probe_vectors(self, T, fac, D, blk_off):
        """T (n, P) <- rows (A'^-1 D^-1 t_j)^T in place."""
        self._stream()
        blk_off = np.ascontiguousarray(blk_off, dtype=np.int32)
        assert T.is_contiguous()
        self.handle.call('isdf_bj_probe_vectors', self._p(T), T.shape[0], self._p(fac), self._p(D), D.shape[0],
                         len(blk_off) - 1, _np_ptr(blk_off))

    def rows_combine(self, E, Y, F, accumulate=False):
        """F (n, ng) (+)= E (n, rows) Y (rows, ng); E may be a column slice of a wider matrix."""
        self._stream()
        assert E.stride(1) == 1 and Y.stride(1) == 1 and F.stride(1) == 1 and E.shape[1] == Y.shape[0]
        self.handle.call('isdf_rows_combine', self._p(E), E.shape[0], E.stride(0), E.shape[1], self._p(Y), Y.shape[1],
                         Y.stride(0), self._p(F), F.stride(0), int(bool(accumulate)))

    def gather_T(self, L, k, piv, T):
        self._stream()
        assert T.is_contiguous() and T.shape == (k, k)
        self.handle.call('isdf_gather_T', self._p(L), int(k), L.stride(0), self._p(piv), self._p(T))

    def W_from_factor(self, F, kind, M):
        """M <- S^-1 M S^-T (kind 0: F = Cholesky factor from fit_prepare; kind 1: F = T)."""
        self._stream()
        assert F.is_contiguous()
        self.handle.call('isdf_W_from_factor', self._p(F), M.shape[0], int(kind), self._p(M), M.stride(0))

    def fit_global(self, ao, ngrids, ip, reg_rel, theta, aoP):
        self._stream()
        assert ip.dtype == torch.int64 and aoP.is_contiguous()
        reg = ctypes.c_double(0.0)
        self.handle.call('isdf_fit_global', self._p(ao), ao.shape[0], int(ngrids), ao.stride(0), self._p(ip),
                         ip.numel(), float(reg_rel), self._p(theta), theta.stride(0), self._p(aoP), ctypes.byref(reg))
        return reg.value

    def coulomb_W(self, theta, mesh, a, row0, nrows, batch, W, upper_only=False):
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.handle.call('isdf_coulomb_W', self._p(theta), theta.shape[0], theta.stride(0), _np_ptr(mesh), _np_ptr(a),
                         int(row0), int(nrows), int(batch), int(bool(upper_only)), self._p(W), W.stride(0))

    def coulomb_rows(self, rows, mesh, a, batch, out=None):
        """out rows = ifft(coulG fft(rows)).real; rows (n, G) contiguous; in place when out is None."""
        self._stream()
        out = rows if out is None else out
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.handle.call('isdf_coulomb_rows', self._p(rows), rows.shape[0], rows.stride(0), _np_ptr(mesh), _np_ptr(a),
                         int(batch), self._p(out), out.stride(0))

    # ---- spectral form of W: W = X X^T with X the scaled half spectra of the fit rows inside a sphere ----
    def spectral_supported(self, mesh, batch=512):
        ok = ctypes.c_int(0)
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        self.handle.call('isdf_spectral_supported', _np_ptr(mesh), int(batch), ctypes.byref(ok))
        return bool(ok.value)

    def coulG_half(self, mesh, a):
        """Host copy (n0, n1, n2/2+1) of the symmetrised half-spectrum kernel table of the Gamma-point convolution (1/G inside)."""
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = self.empty((int(mesh[0]) * int(mesh[1]) * (int(mesh[2]) // 2 + 1),))
        self.handle.call('isdf_coulG_half', _np_ptr(mesh), _np_ptr(a), self._p(out))
        return self.to_host(out).reshape(int(mesh[0]), int(mesh[1]), int(mesh[2]) // 2 + 1)

    def spectral_rows(self, rows, mesh, idx, scale, out, batch=512):
        """out (n, ldx) <- scale_j (Re, Im) fft(rows)[idx_j] packed as consecutive pairs; rows (n, G) contiguous, idx int32."""
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        assert rows.is_contiguous() and out.stride(1) == 1 and idx.dtype == torch.int32 and out.stride(0) % 2 == 0
        self.handle.call('isdf_spectral_rows', self._p(rows), rows.shape[0], rows.stride(0), _np_ptr(mesh), self._p(idx),
                         self._p(scale), int(idx.numel()), int(batch), self._p(out), out.stride(0))

    def symmetrize_upper(self, W):
        self._stream()
        self.handle.call('isdf_symmetrize_upper', self._p(W), W.shape[0], W.stride(0))

    def symmetrize_mean(self, W, antisymmetric=False):
        self._stream()
        self.handle.call('isdf_symmetrize_mean', self._p(W), W.shape[0], W.stride(0), int(bool(antisymmetric)))

    def get_j(self, ao, ngrids, mesh, a, dm, vj):
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert dm.is_contiguous() and vj.is_contiguous()
        self.handle.call('isdf_get_j', self._p(ao), ao.shape[0], int(ngrids), ao.stride(0), _np_ptr(mesh), _np_ptr(a),
                         self._p(dm), dm.shape[0], self._p(vj))

    def rho(self, ao, ng, dm, rho):
        self._stream()
        self.handle.call('isdf_rho', self._p(ao), ao.shape[0], int(ng), ao.stride(0), self._p(dm), dm.shape[0],
                         self._p(rho), rho.stride(0))

    def coulomb_potential(self, rho, mesh, a):
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.handle.call('isdf_coulomb_potential', self._p(rho), rho.shape[0], rho.stride(0), _np_ptr(mesh), _np_ptr(a))

    def vj_from_vR(self, ao, ng, vR, vj):
        self._stream()
        self.handle.call('isdf_vj_from_vR', self._p(ao), ao.shape[0], int(ng), ao.stride(0), self._p(vR), vR.shape[0],
                         vR.stride(0), self._p(vj))

    def get_k(self, aoP, W, row0, nrows, dm, vk):
        self._stream()
        assert aoP.is_contiguous() and dm.is_contiguous() and vk.is_contiguous()
        self.handle.call('isdf_get_k', self._p(aoP), aoP.shape[0], aoP.shape[1], self._p(W), W.stride(0), int(row0),
                         int(nrows), self._p(dm), dm.shape[0], self._p(vk))

    def get_k_exact(self, ao, ngrids, C, mesh, a, i0, ni, max_rows, vk):
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert C.is_contiguous() and vk.is_contiguous()
        self.handle.call('isdf_get_k_exact', self._p(ao), ao.shape[0], int(ngrids), ao.stride(0), self._p(C), C.shape[1],
                         _np_ptr(mesh), _np_ptr(a), int(i0), int(ni), int(max_rows), self._p(vk))

    def gemm_nn(self, A, B, C, alpha=1.0, beta=0.0):
        """C (M, N) = alpha A (M, K) B (K, N) + beta C, row-major."""
        self._stream()
        self.handle.call('isdf_gemm_nn', A.shape[0], B.shape[1], A.shape[1], float(alpha), self._p(A), A.stride(0), self._p(B),
                         B.stride(0), float(beta), self._p(C), C.stride(0))

    def hadamard_rows(self, X, Y):
        """X .*= Y (same shape, rows may be strided)."""
        self._stream()
        self.handle.call('isdf_hadamard_rows', self._p(X), X.stride(0), self._p(Y), Y.stride(0), X.shape[0], X.shape[1])

    def gemm_nt(self, A, B, C, alpha=1.0, beta=0.0, kscale=None):
        """C = alpha * A (B .* kscale)^T + beta * C; A (M,K), B (N,K) row-major, K contiguous."""
        self._stream()
        M, K = A.shape
        N = B.shape[0]
        assert B.shape[1] == K and tuple(C.shape) == (M, N) and A.stride(1) == 1 and B.stride(1) == 1 and C.stride(1) == 1
        ks = self._p(kscale) if kscale is not None else _vp(0)
        self.handle.call('isdf_gemm_nt', M, N, K, float(alpha), self._p(A), A.stride(0), self._p(B), B.stride(0), ks,
                         float(beta), self._p(C), C.stride(0))

    # ---- multigrid ----------------------------------------------------------------------------
    def uniform_grid(self, mesh, a):
        """(3, G) device coordinates of the uniform grid, cell.get_uniform_grids' order and folding."""
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = self.empty((3, int(np.prod(mesh))))
        self.handle.call('isdf_uniform_grid', _np_ptr(mesh), _np_ptr(a), self._p(out))
        return out

    def rho_pair(self, aoA, aoB, ng, dm, rho):
        """rho (nset, ng) = sum_(mu, nu) aoA[mu] dm[:, mu, nu] aoB[nu]; aoA / aoB row blocks of one AO buffer."""
        self._stream()
        assert aoA.stride(0) == aoB.stride(0) and aoA.stride(1) == 1 and aoB.stride(1) == 1 and dm.is_contiguous()
        assert tuple(dm.shape[1:]) == (aoA.shape[0], aoB.shape[0])
        self.handle.call('isdf_rho_pair', self._p(aoA), aoA.shape[0], self._p(aoB), aoB.shape[0], int(ng), aoA.stride(0),
                         self._p(dm), dm.shape[0], self._p(rho), rho.stride(0))

    def mg_embed_density(self, field, mesh_sub, scale, spec, mesh, accumulate=True):
        """spec (nset, gc) complex128 (+)= scale * fft(field (nset, prod(mesh_sub))) at the matching frequencies of ``mesh``."""
        self._stream()
        mesh_sub = np.ascontiguousarray(mesh_sub, dtype=np.int32)
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        assert field.is_contiguous() and spec.is_contiguous() and spec.dtype == torch.complex128
        assert field.shape[1] == int(np.prod(mesh_sub)) and spec.shape[1] == int(mesh[0]) * int(mesh[1]) * (int(mesh[2]) // 2 + 1)
        self.handle.call('isdf_mg_embed_density', self._p(field), field.shape[0], _np_ptr(mesh_sub), float(scale), self._p(spec),
                         _np_ptr(mesh), int(bool(accumulate)))

    def mg_restrict_potential(self, spec, mesh, mesh_sub, scale, field):
        """field (nset, prod(mesh_sub)) = scale * unnormalised inverse FFT of spec cut back to mesh_sub."""
        self._stream()
        mesh_sub = np.ascontiguousarray(mesh_sub, dtype=np.int32)
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        assert field.is_contiguous() and spec.is_contiguous() and spec.dtype == torch.complex128
        assert field.shape[1] == int(np.prod(mesh_sub)) and spec.shape[1] == int(mesh[0]) * int(mesh[1]) * (int(mesh[2]) // 2 + 1)
        self.handle.call('isdf_mg_restrict_potential', self._p(spec), spec.shape[0], _np_ptr(mesh), _np_ptr(mesh_sub), float(scale),
                         self._p(field))

    def mg_coulomb_kernel(self, spec, mesh, a):
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert spec.is_contiguous() and spec.dtype == torch.complex128
        self.handle.call('isdf_mg_coulomb_kernel', self._p(spec), spec.shape[0], _np_ptr(mesh), _np_ptr(a))

    def lda_exchange(self, rho, exc, vxc):
        self._stream()
        assert rho.is_contiguous() and exc.is_contiguous() and vxc.is_contiguous()
        self.handle.call('isdf_lda_exchange', self._p(rho), rho.numel(), self._p(exc), self._p(vxc))

    def lda_vwn_add(self, rho, exc, vxc):
        """exc += eps_c, vxc += v_c (VWN5 correlation, closed shell)."""
        self._stream()
        assert rho.is_contiguous() and exc.is_contiguous() and vxc.is_contiguous()
        self.handle.call('isdf_lda_vwn_add', self._p(rho), rho.numel(), self._p(exc), self._p(vxc))

    def gga_b88(self, rho, grad, exc, vrho, w):
        """rho (G,), grad (3, G) -> exc, vrho (G,), w (3, G) = de/d(grad rho)."""
        self._stream()
        assert rho.is_contiguous() and exc.is_contiguous() and vrho.is_contiguous() and grad.stride(1) == 1 and w.stride(1) == 1
        self.handle.call('isdf_gga_b88', self._p(rho), self._p(grad), grad.stride(0), rho.numel(), self._p(exc), self._p(vrho),
                         self._p(w), w.stride(0))

    def lda_exchange_fxc(self, rho, fxc):
        self._stream()
        assert rho.is_contiguous() and fxc.is_contiguous() and rho.numel() == fxc.numel()
        self.handle.call('isdf_lda_exchange_fxc', self._p(rho), rho.numel(), self._p(fxc))

    def dot(self, x, y=None):
        self._stream()
        assert x.is_contiguous() and (y is None or (y.is_contiguous() and y.numel() == x.numel()))
        out = ctypes.c_double(0.0)
        self.handle.call('isdf_dot', self._p(x), self._p(y) if y is not None else _vp(0), x.numel(), ctypes.byref(out))
        return out.value

    # ---- k-points -----------------------------------------------------------------------------
    def eval_ao_k(self, atm, bas, env, Ls, rcut, kpt, periodic_part, coords_soa, out_re, out_im):
        self._stream()
        atm = np.ascontiguousarray(atm, dtype=np.int32)
        bas = np.ascontiguousarray(bas, dtype=np.int32)
        env = np.ascontiguousarray(env, dtype=np.float64)
        Ls = np.ascontiguousarray(Ls, dtype=np.float64)
        rcut = np.ascontiguousarray(rcut, dtype=np.float64)
        kpt = np.ascontiguousarray(kpt, dtype=np.float64)
        G = coords_soa.shape[1]
        assert out_re.stride(0) == out_im.stride(0) and out_re.stride(1) == 1
        self.handle.call('isdf_eval_ao_k', _np_ptr(atm), len(atm), _np_ptr(bas), len(bas), _np_ptr(env), len(env),
                         _np_ptr(Ls), len(Ls), _np_ptr(rcut), _np_ptr(kpt), int(bool(periodic_part)), self._p(coords_soa), G,
                         self._p(out_re), self._p(out_im), out_re.stride(0))

    def select_ip_cplx(self, X, nh, blk_off, nip, tol, tie_rtol, L, piv):
        self._stream()
        blk_off = np.ascontiguousarray(blk_off, dtype=np.int64)
        nip = np.ascontiguousarray(nip, dtype=np.int32)
        rank = np.zeros(len(nip), dtype=np.int32)
        self.handle.call('isdf_select_ip_cplx', self._p(X), X.shape[0], int(nh), X.stride(0), len(nip), _np_ptr(blk_off),
                         _np_ptr(nip), float(tol), float(tie_rtol), self._p(L), L.stride(0), self._p(piv), _np_ptr(rank))
        return rank

    def fit_prepare_cplx(self, X, nh, ip, reg_rel, aoP, chol):
        self._stream()
        reg = ctypes.c_double(0.0)
        self.handle.call('isdf_fit_prepare_cplx', self._p(X), X.shape[0], int(nh), X.stride(0), self._p(ip), ip.numel(),
                         float(reg_rel), self._p(aoP), self._p(chol), ctypes.byref(reg))
        return reg.value

    def fit_apply_cplx(self, chol, aoP, nh, X, ng, theta, forward_only=False):
        self._stream()
        self.handle.call('isdf_fit_apply_cplx', self._p(chol), self._p(aoP), aoP.shape[0], aoP.shape[1], int(nh),
                         self._p(X), int(ng), X.stride(0), int(bool(forward_only)), self._p(theta), theta.stride(0))

    def coulG_q(self, mesh, a, q, omega=None, wrap_around=True, out=None):
        """Device table (G,) of the Coulomb kernel for the difference vector q (include/mi355_isdf.h isdf_coulG_q)."""
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(3)
        if out is None:
            out = self.empty((int(np.prod(mesh)),))
        self.handle.call('isdf_coulG_q', _np_ptr(mesh), _np_ptr(a), _np_ptr(q), int(bool(wrap_around)), float(omega or 0.0),
                         self._p(out))
        return out

    def coulomb_Wq(self, theta, mesh, coulG, weight, row0, nrows, batch, Wre, Wim, upper_only=False):
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        self.handle.call('isdf_coulomb_Wq', self._p(theta), theta.shape[0], theta.stride(0), _np_ptr(mesh), self._p(coulG),
                         float(weight), int(row0), int(nrows), int(batch), int(bool(upper_only)), self._p(Wre),
                         self._p(Wim), Wre.stride(0))

    def symmetrize_hermitian(self, Wre, Wim):
        self._stream()
        self.handle.call('isdf_symmetrize_hermitian', self._p(Wre), self._p(Wim), Wre.shape[0], Wre.stride(0))

    def finish_Wq(self, Wre, Wim, phase, Wc):
        """Wc (P, P) complex128 <- (Wre + i Wim) * ph[p] * conj(ph[q]); phase (P,) complex128."""
        self._stream()
        assert Wc.dtype == torch.complex128 and Wc.is_contiguous() and phase.dtype == torch.complex128
        self.handle.call('isdf_finish_Wq', self._p(Wre), self._p(Wim), Wre.shape[0], Wre.stride(0), self._p(phase),
                         self._p(Wc))

    def get_k_pair(self, A1, A2, D2, Wq, scale, vk):
        self._stream()
        for t in (A1, A2, D2, Wq, vk):
            assert t.dtype == torch.complex128 and t.is_contiguous()
        self.handle.call('isdf_get_k_pair', self._p(A1), self._p(A2), self._p(D2), self._p(Wq), A1.shape[0], A1.shape[1],
                         float(scale), self._p(vk))

    def get_k_exact_kpt(self, u1r, u1i, m2r, m2i, mesh, coulG, weight, i0, ni, max_rows, vk_re, vk_im):
        """vk (ni, nao) += the exact exchange rows i0..i0+ni of one (k1, k2) pair (include/mi355_isdf.h isdf_get_k_exact_kpt)."""
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        assert u1r.stride(1) == 1 and u1r.stride() == u1i.stride() and m2r.stride(1) == 1 and m2r.stride() == m2i.stride()
        assert vk_re.is_contiguous() and vk_im.is_contiguous() and tuple(vk_re.shape) == (ni, u1r.shape[0])
        self.handle.call('isdf_get_k_exact_kpt', self._p(u1r), self._p(u1i), u1r.shape[0], u1r.stride(0), self._p(m2r),
                         self._p(m2i), m2r.shape[0], m2r.stride(0), _np_ptr(mesh), self._p(coulG), float(weight), int(i0), int(ni),
                         int(max_rows), self._p(vk_re), self._p(vk_im))

    def coulomb_rows_q(self, rows, mesh, coulG, out_re, out_im):
        """out_re + i out_im <- ifft(coulG fft(rows)) for real rows (n, G) contiguous."""
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        assert rows.is_contiguous() and out_re.is_contiguous() and out_im.is_contiguous()
        self.handle.call('isdf_coulomb_rows_q', self._p(rows), rows.shape[0], rows.stride(0), _np_ptr(mesh), self._p(coulG),
                         self._p(out_re), self._p(out_im))

    def nyquist_spectra(self, rows, mesh, axis, out_re, out_im):
        """out_re + i out_im (n, na * nb) <- the 3-D DFT of the real rows on the Nyquist plane of ``axis`` (mesh[axis] even)."""
        self._stream()
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        assert rows.stride(1) == 1 and out_re.is_contiguous() and out_im.is_contiguous()
        self.handle.call('isdf_nyquist_spectra', self._p(rows), rows.shape[0], rows.stride(0), _np_ptr(mesh), int(axis),
                         self._p(out_re), self._p(out_im))

    def zhadamard_planes(self, Ar, Ai, Br, Bi):
        """(Ar + i Ai) .*= (Br + i Bi)."""
        self._stream()
        assert Ar.stride() == Ai.stride() and Br.stride() == Bi.stride() and Ar.stride(1) == 1 and Br.stride(1) == 1
        self.handle.call('isdf_zhadamard_planes', self._p(Ar), self._p(Ai), Ar.stride(0), self._p(Br), self._p(Bi), Br.stride(0),
                         Ar.shape[0], Ar.shape[1])

    def rho_k(self, ur, ui, ng, DTr, DTi, scale, rho):
        self._stream()
        self.handle.call('isdf_rho_k', self._p(ur), self._p(ui), ur.shape[0], int(ng), ur.stride(0), self._p(DTr),
                         self._p(DTi), float(scale), self._p(rho))

    def vj_k(self, ur, ui, ng, vR, vj_re, vj_im):
        self._stream()
        self.handle.call('isdf_vj_k', self._p(ur), self._p(ui), ur.shape[0], int(ng), ur.stride(0), self._p(vR),
                         self._p(vj_re), self._p(vj_im))

    # ---- GTH pseudopotential pieces -----------------------------------------------------------------
    def pp_local_potential(self, coords, pp_par, mesh, a, vlocR):
        self._stream()
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        pp_par = np.ascontiguousarray(pp_par, dtype=np.float64)
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        self.handle.call('isdf_pp_local_potential', len(coords), _np_ptr(coords), _np_ptr(pp_par), _np_ptr(mesh), _np_ptr(a),
                         self._p(vlocR))

    def pp_projector_overlaps(self, atm, bas, env, coords, kpt, proj_tab, proj_rl, mesh, a, out):
        self._stream()
        atm = np.ascontiguousarray(atm, dtype=np.int32)
        bas = np.ascontiguousarray(bas, dtype=np.int32)
        env = np.ascontiguousarray(env, dtype=np.float64)
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        kpt = np.ascontiguousarray(kpt, dtype=np.float64)
        proj_tab = np.ascontiguousarray(proj_tab, dtype=np.int32)
        proj_rl = np.ascontiguousarray(proj_rl, dtype=np.float64)
        mesh = np.ascontiguousarray(mesh, dtype=np.int32)
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert out.dtype == torch.complex128 and out.is_contiguous()
        self.handle.call('isdf_pp_projector_overlaps', _np_ptr(atm), len(atm), _np_ptr(bas), len(bas), _np_ptr(env), len(env),
                         _np_ptr(coords), _np_ptr(kpt), _np_ptr(proj_tab), _np_ptr(proj_rl), len(proj_rl), _np_ptr(mesh),
                         _np_ptr(a), self._p(out))
