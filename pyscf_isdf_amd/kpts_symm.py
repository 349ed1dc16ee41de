"""k-point symmetry for the k-point J/K path: the irreducible wedge of a k-mesh and the transformations
between it and the full Brillouin zone.  Host logic only (numpy); the heavy work stays in ``ISDF.get_jk``, which accepts a
``KPoints`` object in place of the k-point array: density matrices come in on the irreducible k-points, are rotated to the full
zone, and J/K come back on the irreducible k-points (``kpts_band``) - the calling convention of
pyscf/pbc/scf/khf_ksymm.py:210-237.

Mirrors the behaviour of
  * pyscf/pbc/symm/geom.py:27-136        lattice point group and space-group search (native backend, no spglib)
  * pyscf/pbc/symm/symmetry.py:32-322    D matrices, atom map + Bloch phases, rotation of MO coefficients / dm / operators
  * pyscf/pbc/lib/kpts.py:32-106         irreducible k-points: the representative of a star is its LAST k-point of the mesh,
                                         irreducible points in ascending order of that index (pinned by the reference's
                                         test_kpts_ksymm.py fingerprints of kpts_ibz)
  * pyscf/pbc/lib/kpts.py:369-405,441-724,772-958   KPoints: symmetrize_density, transform_*, dm_at_ref_cell, check_mo_occ_symmetry
  * pyscf/lib/pbc/symmetry.c             the grid permutation behind symmetrize_density
written from the mathematics, not from those files:

  operation g = {W | t} on fractional coordinates, x' = W x + t (W integer, W^T G W = G for the metric G = a a^T);
  on scaled k-points kappa' = W^-T kappa;  Cartesian rotation R = a^T W a^-T;
  g phi^k_{i m} = exp(i Rk . L) sum_m' D_{m' m}(R) phi^{Rk}_{j m'},   g r_i = r_j - L,   Y_m(R^-1 u) = sum_m' D_{m' m} Y_m'(u)
  => O^{Rk} = U O^k U^H for every g-invariant operator (and for the density matrix), U[j block, i block] = D exp(i Rk . L);
  time reversal: O^{-k} = conj(O^k).
"""
import warnings
import numpy as np

SYMPREC = 1e-6          # pyscf/pbc/symm/geom.py, space_group.py
KPT_DIFF_TOL = 1e-6     # pyscf/pbc/lib/kpts_helper.py


# ---- real solid harmonics in PySCF's AO order (p: x y z; d: xy yz z2 xz x2-y2; f: m = -3 .. 3), common normalisation per l --
def _real_harmonics(l, u):
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    r2 = x * x + y * y + z * z
    if l == 0:
        return np.ones((len(u), 1))
    if l == 1:
        return np.stack([x, y, z], axis=1)
    if l == 2:
        s3 = np.sqrt(3.0)
        return np.stack([s3 * x * y, s3 * y * z, 0.5 * (3 * z * z - r2), s3 * x * z, 0.5 * s3 * (x * x - y * y)], axis=1)
    if l == 3:
        c3, c2, c1 = np.sqrt(5.0 / 8), np.sqrt(15.0), np.sqrt(3.0 / 8)
        return np.stack([c3 * (3 * x * x * y - y ** 3), c2 * x * y * z, c1 * y * (5 * z * z - r2), 0.5 * z * (5 * z * z - 3 * r2),
                         c1 * x * (5 * z * z - r2), 0.5 * c2 * z * (x * x - y * y), c3 * (x ** 3 - 3 * x * y * y)], axis=1)
    raise NotImplementedError('k-point symmetry: shells up to l = 3 (the collocation kernels stop there too)')


_SAMPLE = np.random.default_rng(20240607).standard_normal((64, 3))


def rotation_Dmat(R, l):
    """D (2l+1, 2l+1) of a proper or improper Cartesian rotation R for the real harmonics above:
    Y_m(R^-1 u) = sum_m' D[m', m] Y_m'(u)  (the matrix  < m' | R | m >  of symmetry.py:32-54).  Orthogonal."""
    Y0 = _real_harmonics(l, _SAMPLE)
    Y1 = _real_harmonics(l, _SAMPLE.dot(R))            # rows u^T R = (R^T u)^T = (R^-1 u)^T
    D = np.linalg.lstsq(Y0, Y1, rcond=None)[0]
    D[abs(D) < 1e-13] = 0.0
    return D


class SpaceGroupOp:
    """{rot | trans} on fractional coordinates of the direct lattice (space_group.py:82-205)."""

    def __init__(self, rot=None, trans=None):
        self.rot = np.eye(3, dtype=int) if rot is None else np.asarray(rot, dtype=int)
        self.trans = np.zeros(3) if trans is None else np.asarray(trans, dtype=float)

    @property
    def is_eye(self):
        return bool((self.rot == np.eye(3, dtype=int)).all() and abs(self.trans).max() < SYMPREC)

    @property
    def rot_is_inversion(self):
        return bool((self.rot == -np.eye(3, dtype=int)).all())

    @property
    def trans_is_zero(self):
        return bool(abs(self.trans).max() < SYMPREC)

    def dot_rot(self, x):
        return np.dot(x, self.rot.T)

    def dot(self, x):
        return np.dot(x, self.rot.T) + self.trans

    def inv(self):
        ri = np.rint(np.linalg.inv(self.rot)).astype(int)
        return SpaceGroupOp(ri, -np.dot(self.trans, ri.T))

    def rot_recip(self):
        """The rotation on scaled k-points: kappa' = W^-T kappa (the a2b transform of space_group.py:30-60,207-211)."""
        return np.rint(np.linalg.inv(self.rot).T).astype(int)

    def rot_cart(self, a):
        """Cartesian rotation matrix R = a^T W a^-T (a2r)."""
        a = np.asarray(a, dtype=float)
        return a.T.dot(self.rot).dot(np.linalg.inv(a.T))

    def _key(self):
        t = np.mod(np.round(self.trans, 6), 1.0)
        return tuple(self.rot.ravel().tolist()) + tuple(np.round(t, 6).tolist())

    def __repr__(self):
        return 'SpaceGroupOp(rot=%s, trans=%s)' % (self.rot.tolist(), np.round(self.trans, 6).tolist())


def search_point_group_ops(cell, tol=SYMPREC):
    """Integer matrices W (entries -1, 0, 1 in a reduced basis) that leave the lattice metric invariant (geom.py:27-66)."""
    a = np.asarray(cell.lattice_vectors(), dtype=float)
    G = a.dot(a.T)
    scale = np.sqrt(np.outer(G.diagonal(), G.diagonal()))
    cand = np.array(np.meshgrid(*([[1, 0, -1]] * 9), indexing='ij')).reshape(9, -1).T.reshape(-1, 3, 3)
    Gt = np.einsum('nji,jk,nkl->nil', cand, G, cand)
    ok = (abs(Gt - G) / scale).reshape(len(cand), -1).max(axis=1) < tol
    return cand[ok].astype(int)


def _wrap01(x, tol):
    """Fractional coordinates folded into [0, 1) with values within tol of 1 sent to 0."""
    x = x - np.floor(x)
    x[x > 1.0 - tol] = 0.0
    x[abs(x) < tol] = 0.0
    return x


def _same_point_sets(x, y, a, tol):
    """Do the fractional point sets x and y coincide modulo lattice vectors (Cartesian distance < tol)?"""
    if len(x) != len(y):
        return False
    d = x[:, None, :] - y[None, :, :]
    d -= np.rint(d)
    dist = np.linalg.norm(d.dot(a), axis=2)
    hit = dist < tol
    return bool(hit.any(axis=1).all() and hit.any(axis=0).all())


def search_space_group_ops(cell, rotations=None, tol=SYMPREC):
    """All {W | t} that map the crystal onto itself, atoms distinguished by their symbol (geom.py:68-136; no magnetic moments)."""
    a = np.asarray(cell.lattice_vectors(), dtype=float)
    if rotations is None:
        rotations = search_point_group_ops(cell, tol)
    frac = np.asarray(cell.atom_coords(), dtype=float).dot(np.linalg.inv(a))
    symbols = [cell.atom_symbol(i) for i in range(cell.natm)]
    groups = {}
    for i, s in enumerate(symbols):
        groups.setdefault(s, []).append(i)
    groups = {s: frac[idx] for s, idx in groups.items()}
    smallest = min(groups.values(), key=len)
    ops = []
    for W in rotations:
        # candidate translations: the first atom of the smallest species must land on an atom of that species
        seen = []
        for tgt in smallest:
            t = _wrap01(tgt - smallest[0].dot(W.T), 1e-9)
            if any(np.linalg.norm((t - s - np.rint(t - s)).dot(a)) < tol for s in seen):
                continue
            seen.append(t)
            if all(_same_point_sets(x.dot(W.T) + t, x, a, tol) for x in groups.values()):
                ops.append(SpaceGroupOp(W, _wrap01(np.round(t, 12), 1e-9)))
    ops.sort(key=lambda o: (not o.is_eye, not o.trans_is_zero) + o._key())       # identity first, then the symmorphic ones
    return ops


class KPoints:
    """The k-mesh with its symmetry (pyscf/pbc/lib/kpts.py:815-958).  Attributes as in the reference: kpts, kpts_scaled, nkpts,
    weights; ops, nop, Dmats, time_reversal, has_inversion; kpts_ibz, kpts_scaled_ibz, nkpts_ibz, weights_ibz, ibz2bz, bz2ibz,
    k2opk, stars, stars_ops, stars_ops_bz, time_reversal_symm_bz, little_cogroup_ops."""

    def __init__(self, cell=None, kpts=np.zeros((1, 3))):
        self.cell = cell
        self.kpts = self.kpts_ibz = np.asarray(kpts, dtype=float).reshape(-1, 3)
        n = len(self.kpts)
        self.kpts_scaled = self.kpts_scaled_ibz = None
        self.weights = self.weights_ibz = np.full(n, 1.0 / n)
        self.ibz2bz = self.bz2ibz = np.arange(n)
        self.ops = [SpaceGroupOp()]
        self.nop = 1
        self.symmorphic = True
        self.has_inversion = False
        self.time_reversal = False
        self.Dmats = None
        self.l_max = None
        self.k2opk = None
        self.stars, self.stars_ops = [], []
        self.stars_ops_bz = np.zeros(n, dtype=int)
        self.time_reversal_symm_bz = np.zeros(n, dtype=int)
        self.little_cogroup_ops = []

    nkpts = property(lambda self: len(self.kpts))
    nkpts_ibz = property(lambda self: len(self.kpts_ibz))

    def __len__(self):
        return self.nkpts_ibz

    # ---- construction -------------------------------------------------------------------------------------------------
    def build(self, space_group_symmetry=False, time_reversal_symmetry=False, symmorphic=False, check_mesh_symmetry=True):
        """symmorphic: keep only the operations without a fractional translation (symmetry.py:164-206); otherwise operations whose
        translation is not a multiple of the FFT mesh spacing are dropped (check_mesh_symmetry, symmetry.py:96-129: the density
        symmetrisation permutes grid points)."""
        cell = self.cell
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        if space_group_symmetry:
            ops = search_space_group_ops(cell)
            if symmorphic:
                ops = [o for o in ops if o.trans_is_zero]
            elif check_mesh_symmetry and getattr(cell, 'mesh', None) is not None:
                mesh = np.asarray(cell.mesh, dtype=float)
                keep = [o for o in ops if abs(o.trans * mesh - np.rint(o.trans * mesh)).max() < SYMPREC]
                if len(keep) != len(ops):
                    warnings.warn('k-point symmetry: mesh %s is not compatible with %d of the %d space-group operations; they are '
                                  'left out' % (list(cell.mesh), len(ops) - len(keep), len(ops)))
                ops = keep
            self.ops = ops
            self.symmorphic = bool(symmorphic)
        else:
            self.ops = [SpaceGroupOp()]
        self.nop = len(self.ops)
        self.has_inversion = any(o.rot_is_inversion for o in self.ops)
        self.l_max = int(max(cell.bas_angular(i) for i in range(cell.nbas))) if cell.nbas else 0
        self.Dmats = [[rotation_Dmat(o.rot_cart(a), l) for l in range(self.l_max + 1)] for o in self.ops]
        self.time_reversal = bool(time_reversal_symmetry) and not self.has_inversion
        self.kpts_scaled = self.kpts_scaled_ibz = self.kpts.dot(a.T) / (2 * np.pi)
        self._make_kpts_ibz()
        return self

    def _map_kpts(self, rots, tol=KPT_DIFF_TOL):
        """table[k, s] = index of rots[s] kappa_k in the mesh modulo reciprocal lattice vectors, -1 when it is not a mesh point."""
        from scipy.spatial import cKDTree
        ks = self.kpts_scaled
        box = _wrap01(ks.copy(), tol * 0.5)
        tree = cKDTree(np.clip(box, 0.0, np.nextafter(1.0, 0.0)), boxsize=1.0)
        table = -np.ones((len(ks), len(rots)), dtype=int)
        for s, W in enumerate(rots):
            img = _wrap01(ks.dot(np.asarray(W, dtype=float).T), tol * 0.5)
            dist, idx = tree.query(np.clip(img, 0.0, np.nextafter(1.0, 0.0)), k=1, distance_upper_bound=3 * tol)
            for k in range(len(ks)):
                if np.isfinite(dist[k]):
                    d = img[k] - box[idx[k]]
                    if abs(d - np.rint(d)).max() < tol:
                        table[k, s] = idx[k]
        return table

    def _make_kpts_ibz(self, tol=KPT_DIFF_TOL):
        nk, nop = self.nkpts, self.nop
        rots = [o.rot_recip() for o in self.ops]
        if self.time_reversal:
            rots = rots + [-r for r in rots]
        table = self._map_kpts(rots, tol)
        self.k2opk = table.copy()
        bad = np.unique(np.where(table == -1)[1])
        if len(bad):
            warnings.warn('k-points have lower symmetry than the lattice: %d operations are not used' % len(bad))
            table[:, bad] = -1
        good = [s for s in range(len(rots)) if s not in set(bad.tolist())]
        # the representative of a star is its last k-point; irreducible points in ascending order (kpts.py:58-71)
        rep = -np.ones(nk, dtype=int)
        ibz2bz = []
        for k in range(nk - 1, -1, -1):
            if rep[k] == -1:
                rep[table[k, good]] = k
                ibz2bz.append(k)
        self.ibz2bz = np.array(ibz2bz[::-1], dtype=int)
        pos = np.empty(nk, dtype=int)
        pos[self.ibz2bz] = np.arange(len(self.ibz2bz))
        self.bz2ibz = pos[rep]
        self.weights_ibz = np.bincount(self.bz2ibz) / float(nk)
        self.kpts_scaled_ibz = self.kpts_scaled[self.ibz2bz]
        b = np.asarray(self.cell.reciprocal_vectors(), dtype=float)
        self.kpts_ibz = self.kpts_scaled_ibz.dot(b)
        # for every k-point the first operation that takes its irreducible representative to it
        self.stars_ops_bz = np.zeros(nk, dtype=int)
        self.time_reversal_symm_bz = np.zeros(nk, dtype=int)
        for k in range(nk):
            kr = self.ibz2bz[self.bz2ibz[k]]
            for s in good:
                if table[kr, s] == k:
                    self.time_reversal_symm_bz[k] = s // nop
                    self.stars_ops_bz[k] = s % nop
                    break
            else:
                raise RuntimeError('k-point %d is not an image of its representative' % k)
        self.stars = [np.where(self.bz2ibz == i)[0] for i in range(self.nkpts_ibz)]
        self.stars_ops = [self.stars_ops_bz[idx] for idx in self.stars]
        self.little_cogroup_ops = [np.where(self.k2opk[k] == k)[0] for k in self.ibz2bz]

    # ---- AO rotation matrices -------------------------------------------------------------------------------------------
    def _atom_map_and_phase(self, op, kpt_scaled, tol=SYMPREC):
        """atm_map[i] = j with g r_i = r_j - L, phase_i = exp(2 pi i (W^-T kappa) . L)  (symmetry.py:220-242)."""
        cell = self.cell
        a = np.asarray(cell.lattice_vectors(), dtype=float)
        frac = np.asarray(cell.atom_coords(), dtype=float).dot(np.linalg.inv(a))
        krot = np.dot(kpt_scaled, op.rot_recip().T)
        natm = cell.natm
        amap = np.zeros(natm, dtype=int)
        phase = np.ones(natm, dtype=np.complex128)
        for i in range(natm):
            img = op.dot(frac[i])
            d = frac - img
            L = np.rint(d)
            hit = np.where(np.linalg.norm((d - L).dot(a), axis=1) < tol * 10)[0]
            hit = [j for j in hit if cell.atom_symbol(j) == cell.atom_symbol(i)]
            if len(hit) != 1:
                raise RuntimeError('operation %r does not map atom %d onto an atom' % (op, i))
            amap[i] = hit[0]
            phase[i] = np.exp(2j * np.pi * np.dot(krot, L[hit[0]]))
        return amap, phase

    def rotation_mat(self, k_ibz, iop):
        """U (nao, nao) with  O^{g k} = U O^k U^H  for the irreducible k-point k_ibz (index) and operation iop."""
        cell = self.cell
        op, Dm = self.ops[iop], self.Dmats[iop]
        amap, phase = self._atom_map_and_phase(op, self.kpts_scaled_ibz[k_ibz])
        ao_loc = np.asarray(cell.ao_loc_nr())
        shl_by_atom = [[s for s in range(cell.nbas) if cell.bas_atom(s) == i] for i in range(cell.natm)]
        nao = cell.nao_nr()
        U = np.zeros((nao, nao), dtype=np.complex128)
        for i in range(cell.natm):
            j = amap[i]
            si, sj = shl_by_atom[i], shl_by_atom[j]
            if len(si) != len(sj):
                raise RuntimeError('atoms %d and %d are symmetry equivalent but carry different shells' % (i, j))
            for p, q in zip(si, sj):
                l, nc = cell.bas_angular(p), cell.bas_nctr(p)
                if cell.bas_angular(q) != l or cell.bas_nctr(q) != nc:
                    raise RuntimeError('atoms %d and %d are symmetry equivalent but carry different shells' % (i, j))
                nd = 2 * l + 1
                for c in range(nc):
                    i0, j0 = ao_loc[p] + c * nd, ao_loc[q] + c * nd
                    U[j0:j0 + nd, i0:i0 + nd] = Dm[l] * phase[i]
        return U

    def _to_bz(self, k, x, kind):
        ki, iop, tr = self.bz2ibz[k], self.stars_ops_bz[k], self.time_reversal_symm_bz[k]
        y = np.asarray(x[ki])
        if not self.ops[iop].is_eye:
            U = self.rotation_mat(ki, iop)
            y = U.dot(y) if kind == 'mo' else U.dot(y).dot(U.conj().T)
        return y.conj() if tr else y

    @staticmethod
    def _is_spin_pair(x, inner_ndim):
        """A leading axis of sets ([alpha, beta] of the reference's is_uhf tests, kpts.py:549-553,647-650; any number of sets
        here) in front of the per-k arrays?"""
        first = x[0][0] if not isinstance(x, np.ndarray) else None
        if isinstance(x, np.ndarray):
            return x.ndim == inner_ndim + 2
        return isinstance(first, np.ndarray) and first.ndim == inner_ndim

    def transform_mo_coeff(self, mo_coeff_ibz):
        """MO coefficients on every k-point of the zone from those on the irreducible ones (kpts.py:441-478)."""
        if self._is_spin_pair(mo_coeff_ibz, 2):
            return [[self._to_bz(k, mo_coeff_ibz[s], 'mo') for k in range(self.nkpts)] for s in range(len(mo_coeff_ibz))]
        return [self._to_bz(k, mo_coeff_ibz, 'mo') for k in range(self.nkpts)]

    def transform_mo_occ(self, mo_occ_ibz):
        if self._is_spin_pair(mo_occ_ibz, 1):
            return [[mo_occ_ibz[s][self.bz2ibz[k]] for k in range(self.nkpts)] for s in range(len(mo_occ_ibz))]
        return [mo_occ_ibz[self.bz2ibz[k]] for k in range(self.nkpts)]

    transform_mo_energy = transform_mo_occ

    def transform_dm(self, dm_ibz):
        """(nkpts_ibz, N, N) [or (2, nkpts_ibz, N, N)] -> the density matrices on the full zone (kpts.py:532-588); an
        mo_coeff / mo_occ tag on the input is rotated along."""
        from ._common import tag_array
        if self._is_spin_pair(dm_ibz, 2):
            out = np.array([[self._to_bz(k, dm_ibz[s], 'op') for k in range(self.nkpts)] for s in range(len(dm_ibz))])
        else:
            out = np.array([self._to_bz(k, dm_ibz, 'op') for k in range(self.nkpts)])
        if getattr(dm_ibz, 'mo_coeff', None) is not None:
            out = tag_array(out, mo_coeff=self.transform_mo_coeff(dm_ibz.mo_coeff), mo_occ=self.transform_mo_occ(dm_ibz.mo_occ))
        return out

    def transform_1e_operator(self, fock_ibz):
        if self._is_spin_pair(fock_ibz, 2):
            return np.array([[self._to_bz(k, fock_ibz[s], 'op') for k in range(self.nkpts)] for s in range(len(fock_ibz))])
        return np.array([self._to_bz(k, fock_ibz, 'op') for k in range(self.nkpts)])

    transform_fock = transform_1e_operator

    def dm_at_ref_cell(self, dm_ibz):
        dm0 = self.transform_dm(dm_ibz).sum(axis=-3) / self.nkpts
        if abs(dm0.imag).max() > 1e-10:
            warnings.warn('imaginary density matrix at the reference cell: max |Im| = %g' % abs(dm0.imag).max())
        return dm0

    def check_mo_occ_symmetry(self, mo_occ, tol=1e-5):
        """Occupations on the full zone -> on the irreducible k-points; RuntimeError when a star is not uniformly occupied."""
        for star in self.stars:
            for k in star[1:]:
                if not (abs(np.asarray(mo_occ[star[0]]) - np.asarray(mo_occ[k])) < tol).all():
                    raise RuntimeError('symmetry-broken occupations in the star of k-point %d' % star[0])
        return [mo_occ[k] for k in self.ibz2bz]

    def symmetrize_density(self, rho_k, ibz_k_idx, mesh):
        """sum over the star of irreducible k-point ibz_k_idx of rho_k(g^-1 r) on the uniform grid (kpts.py:369-405 with the grid
        permutations of lib/pbc/symmetry.c); rho_k (..., G) real or complex in C order over the mesh."""
        mesh = np.asarray(mesh, dtype=int)
        rho_k = np.asarray(rho_k)
        flat = rho_k.reshape(-1, int(np.prod(mesh)))
        out = np.zeros_like(flat)
        idx = np.stack(np.meshgrid(*[np.arange(n) for n in mesh], indexing='ij'), axis=-1).reshape(-1, 3)
        for iop in self.stars_ops[ibz_k_idx]:
            op = self.ops[iop]
            if op.is_eye:
                out += flat
                continue
            inv = op.inv()
            src = (idx / mesh).dot(inv.rot.T) + inv.trans
            src = src * mesh
            if abs(src - np.rint(src)).max() > 1e-6:
                raise RuntimeError('mesh %s is not invariant under operation %r' % (mesh.tolist(), op))
            src = np.mod(np.rint(src).astype(int), mesh)
            out += flat[:, np.ravel_multi_index(src.T, mesh)]
        return out.reshape(rho_k.shape)


def make_kpts(cell, kpts=np.zeros((1, 3)), space_group_symmetry=False, time_reversal_symmetry=False, **kwargs):
    """KPoints object of a list of k-points of the full zone (pyscf/pbc/lib/kpts.py:772-813)."""
    if isinstance(kpts, KPoints):
        return kpts.build(space_group_symmetry, time_reversal_symmetry, **kwargs)
    return KPoints(cell, kpts).build(space_group_symmetry, time_reversal_symmetry, **kwargs)
