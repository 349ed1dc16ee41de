"""The rest of FFTDF's ``with_df`` surface that post-HF callers reach (pyscf/pbc/df/fft.py:317-345): ``get_ao_pairs_G``,
``get_mo_pairs_G`` (exact pair-density transforms, pyscf/pbc/df/fft_ao2mo.py:219-340) and ``loop`` (three-index blocks for the
molecular DF code, fft.py:326-345).  Small-system helpers: AO values come from the device collocation, the pair products and
their transforms are assembled on the host like the ERI helpers of isdf.py.  ``ao2mo_7d`` is the exact transform of the reference (not the ISDF factorisation)."""
import numpy as np
from . import gto


class EriSurfaceMixin:
    def _ao_values_host(self, kpt):
        """(G, nao) AO values at one k-point on the FFT grid (device collocation; complex Bloch AOs for k != 0)."""
        be, cell = self.backend, self.cell
        coords = self.grids.coords
        G = len(coords)
        nao = cell.nao_nr()
        rcut = gto.estimate_rcut_per_shell(cell)
        Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
        args = (np.asarray(cell._atm), np.asarray(cell._bas), np.asarray(cell._env), Ls, rcut)
        d_coords = be.to_device(np.ascontiguousarray(coords.T))
        if abs(np.asarray(kpt)).sum() < 1e-9:
            ao = be.empty((nao, G))
            be.eval_ao(*args, d_coords, ao)
            return be.to_host(ao).T
        re, im = be.empty((nao, G)), be.empty((nao, G))
        be.eval_ao_k(*args, np.asarray(kpt, dtype=float), False, d_coords, re, im)
        return (be.to_host(re) + 1j * be.to_host(im)).T

    def _pairs_G(self, fi, fj, kpts, q, compact):
        """FFT of conj(f_i) f_j exp(-i q.r) over all pairs; fi, fj: (G, n) orbital values at kpts[0], kpts[1]."""
        mesh = [int(x) for x in self.mesh]
        G = int(np.prod(mesh))
        gamma = abs(np.asarray(kpts)).sum() < 1e-9
        same = fi is fj
        if compact and gamma and same:
            n = fi.shape[1]
            i, j = np.tril_indices(n)
            prod = (fi[:, i].conj() * fi[:, j]).T                      # (npair, G), pairs (i >= j) in tril order
        else:
            prod = (fi.conj()[:, :, None] * fj[:, None, :]).reshape(G, -1).T
            if not (gamma or abs(kpts[0] - kpts[1]).max() < 1e-9):
                qv = kpts[1] - kpts[0] if q is None else np.asarray(q)
                prod = prod * np.exp(-1j * self.grids.coords.dot(qv))[None, :]
        out = np.fft.fftn(prod.reshape(-1, *mesh), axes=(1, 2, 3)).reshape(-1, G)
        return np.ascontiguousarray(out.T)

    def get_ao_pairs_G(self, kpts=np.zeros((2, 3)), q=None, shls_slice=None, compact=False):
        """Forward transforms (G|ij) of all AO pair densities, FFTDF.get_ao_pairs_G (fft_ao2mo.py:219-280): complex
        (ngrids, nao*(nao+1)/2) for compact Gamma-point input, else (ngrids, nao_i * nao_j)."""
        kpts = np.zeros((2, 3)) if kpts is None else np.asarray(kpts, dtype=float).reshape(2, 3)
        ao_loc = np.asarray(self.cell.ao_loc_nr())
        if shls_slice is None:
            i0, i1, j0, j1 = 0, self.cell.nao_nr(), 0, self.cell.nao_nr()
        else:
            i0, i1, j0, j1 = (int(ao_loc[s]) for s in shls_slice)
        aoi = self._ao_values_host(kpts[0])
        if abs(kpts[0] - kpts[1]).max() < 1e-9:
            if compact and abs(kpts).sum() < 1e-9 and (i0, i1) == (j0, j1) and i0 == 0:
                sub = aoi[:, :i1]
                return self._pairs_G(sub, sub, kpts, q, True)
            return self._pairs_G(aoi[:, i0:i1], aoi[:, j0:j1], kpts, q, False)
        aoj = self._ao_values_host(kpts[1])
        return self._pairs_G(aoi[:, i0:i1], aoj[:, j0:j1], kpts, q, False)

    get_ao_pairs = get_ao_pairs_G

    def get_mo_pairs_G(self, mo_coeffs, kpts=np.zeros((2, 3)), q=None, compact=False):
        """Forward transforms (G|ij) of all MO pair densities, FFTDF.get_mo_pairs_G (fft_ao2mo.py:282-340)."""
        kpts = np.zeros((2, 3)) if kpts is None else np.asarray(kpts, dtype=float).reshape(2, 3)
        ci, cj = np.asarray(mo_coeffs[0]), np.asarray(mo_coeffs[1])
        aoi = self._ao_values_host(kpts[0])
        aoj = aoi if abs(kpts[0] - kpts[1]).max() < 1e-9 else self._ao_values_host(kpts[1])
        moi = aoi.dot(ci)
        if aoj is aoi and ci.shape == cj.shape and abs(ci - cj).max() < 1e-15:
            return self._pairs_G(moi, moi, kpts, q, compact)
        return self._pairs_G(moi, aoj.dot(cj), kpts, q, False)

    get_mo_pairs = get_mo_pairs_G

    def loop(self, blksize=None):
        """Three-index blocks L (naux_blk, nao*(nao+1)/2) with sum_L L_pq L_rs = (pq|rs), the contract of FFTDF.loop
        (fft.py:326-345) that lets the molecular DF code drive a Gamma-point periodic object.  Here the auxiliary index is
        the ISDF one (get_naoaux() = number of interpolation points, against 2 ngrids for FFTDF):
        L = W^{1/2} X, X_{P,pq} = phi_p(r_P) phi_q(r_P), W^{1/2} from W's eigendecomposition (negative eigenvalues of the
        fitted W, rounding-sized, are dropped)."""
        if getattr(self.cell, 'dimension', 3) < 3:
            raise RuntimeError('ERIs of 1D and 2D systems are not positive definite')
        if not self._is_gamma(self.kpts):
            raise NotImplementedError('loop() is the Gamma-point interface of the molecular DF code')
        if not self._built:
            self.build()
        if blksize is None:
            blksize = self.blockdim
        aoP = self.backend.to_host(self.aoP)
        W = self.backend.to_host(self.W)
        ev, U = np.linalg.eigh((W + W.T) * .5)
        keep = ev > 1e-14 * ev.max()
        half = (U[:, keep] * np.sqrt(ev[keep])).T                      # (naux, P): half^T half = W on the kept space
        i, j = np.tril_indices(aoP.shape[1])
        X = aoP[:, i] * aoP[:, j]
        L = half.dot(X)
        for p0 in range(0, len(L), blksize):
            yield L[p0:p0 + blksize]

    def get_kconserv(self, kpts):
        """kconserv[k, l, m] = n with k_k - k_l + k_m - k_n a reciprocal lattice vector (pyscf/pbc/lib/kpts_helper.py:260-283)."""
        kpts = np.asarray(kpts, dtype=float).reshape(-1, 3)
        nk = len(kpts)
        a = np.asarray(self.cell.lattice_vectors(), dtype=float) / (2 * np.pi)
        out = np.zeros((nk, nk, nk), dtype=int)
        for k in range(nk):
            for l in range(nk):
                for m in range(nk):
                    t = (kpts[k] - kpts[l] + kpts[m])[None, :] - kpts                  # (nk, 3): candidates for zero mod G
                    s = t.dot(a.T)
                    hit = np.nonzero(abs(s - np.rint(s)).sum(axis=1) < 1e-9)[0]
                    if len(hit) == 0:
                        raise ValueError('k-points are not closed under momentum conservation')
                    out[k, l, m] = hit[0]
        return out

    def ao2mo_7d(self, mo_coeff_kpts, kpts=None, factor=1, out=None):
        """All momentum-conserving MO integrals of a k-mesh, (nk, nk, nk, nmo_i, nmo_j, nmo_k, nmo_l) with the fourth k-point
        fixed by conservation modulo a reciprocal lattice vector: FFTDF.ao2mo_7d (pyscf/pbc/df/fft_ao2mo.py:342-425).
        EXACT, like the reference (pair densities on the grid, one FFT pair per (k,l) pair density), not through the ISDF
        factorisation: quartets that conserve momentum only up to a reciprocal lattice vector involve pair products
        exp(i G0.r) conj(u) u outside the fitted span.  A post-HF helper for small cells: AO values from the device
        collocation, transforms and contractions on the host."""
        cell = self.cell
        kpts = self.kpts if kpts is None else np.asarray(kpts, dtype=float).reshape(-1, 3)
        nk = len(kpts)
        if isinstance(mo_coeff_kpts, np.ndarray) and mo_coeff_kpts.ndim == 3:
            mo_coeff_kpts = [mo_coeff_kpts] * 4
        mo_coeff_kpts = [np.asarray(x) for x in mo_coeff_kpts]
        mesh = [int(x) for x in self.mesh]
        G = int(np.prod(mesh))
        coords = self.grids.coords
        aos = [self._ao_values_host(k) for k in kpts]                                # (G, nao) each
        mos = [[aos[k].dot(mo_coeff_kpts[n][k]).T for k in range(nk)] for n in range(4)]     # (nmo, G)
        nmo = [x.shape[2] for x in mo_coeff_kpts]
        shape = (nk, nk, nk) + tuple(nmo)
        gamma = abs(kpts).sum() < 1e-9
        dtype = np.result_type(*mo_coeff_kpts) if gamma else np.complex128
        if out is None:
            out = np.empty(shape, dtype=dtype)
        assert out.shape == shape
        kconserv = self.get_kconserv(kpts)
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        done = set()
        for ki in range(nk):
            for kj in range(nk):
                if (ki, kj) in done:
                    continue
                q = kpts[kj] - kpts[ki]
                same_q = [(i, j) for i in range(nk) for j in range(nk) if abs(kpts[j] - kpts[i] - q).max() < 1e-9]
                coulG = self.backend.to_host(self.backend.coulG_q(np.asarray(mesh, dtype=np.int32), a, q)) * (cell.vol / G) * factor
                phase = np.exp(-1j * coords.dot(q))
                z = []
                for kk in range(nk):
                    kl = kconserv[ki, kj, kk]
                    pairs = (mos[2][kk].conj()[:, None, :] * mos[3][kl][None, :, :]).reshape(-1, G) * phase.conj()
                    v = np.fft.ifftn(pairs.reshape(-1, *mesh), axes=(1, 2, 3)).reshape(-1, G) * coulG
                    v = np.fft.fftn(v.reshape(-1, *mesh), axes=(1, 2, 3)).reshape(-1, G) * phase
                    z.append(v)
                for i, j in same_q:
                    pij = (mos[0][i].conj()[:, None, :] * mos[1][j][None, :, :]).reshape(-1, G)
                    for kk in range(nk):
                        t = pij.dot(z[kk].T)
                        out[i, j, kk] = (t.real if dtype == np.double else t).reshape(shape[3:])
                    done.add((i, j))
        return out
