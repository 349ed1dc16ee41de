"""Host-side periodic cell: the subset of ``pyscf.pbc.gto.Cell`` the ISDF path reads.

PySCF is not importable where this package runs, so ``Cell`` here produces — from the same user
inputs (atom string, lattice, basis name or table, mesh) — exactly the arrays the reference hands to
its native collocation code: the libcint-layout ``_atm/_bas/_env`` tables with normalisation folded
into the contraction coefficients, lattice/reciprocal vectors, the uniform grid, the G vectors, the
per-shell cutoff radii and the lattice-sum translation list.  ``ISDF`` only ever touches the
attribute/method names below, all of which exist with the same meaning on a real PySCF ``Cell``, so
either object can be passed.

Reference conventions followed (nothing is imported from it):
  env/bas/atm slot layout ........ pyscf/gto/mole.py:59-89
  primitive normalisation ........ pyscf/gto/mole.py:116-151 (gto_norm), :980-1023 (make_bas_env,
                                   _nomalize_contracted_ao; primitives sorted by descending exponent)
  CP2K basis format .............. pyscf/gto/basis/parse_cp2k.py:32-129
  shell ordering per element ..... pyscf/gto/mole.py:466-467 (stable sort by l)
  lattice / grids / G vectors .... pyscf/pbc/gto/cell.py:523-603, 874-898, 1571-1591
  rcut estimators ................ pyscf/pbc/gto/cell.py:390-434, pyscf/pbc/gto/eval_gto.py:169-186
  lattice-sum translations ....... pyscf/pbc/gto/eval_gto.py:188-253
"""
import os
import re
import numpy as np
from scipy.special import gamma as _gamma

BOHR = 0.52917721092  # Angstrom, pyscf/data/nist.py:24

# slot names, pyscf/gto/mole.py:59-89
CHARGE_OF, PTR_COORD, NUC_MOD_OF, PTR_ZETA, PTR_FRAC_CHARGE, PTR_RADIUS, ATM_SLOTS = 0, 1, 2, 3, 4, 5, 6
ATOM_OF, ANG_OF, NPRIM_OF, NCTR_OF, KAPPA_OF, PTR_EXP, PTR_COEFF, BAS_SLOTS = 0, 1, 2, 3, 4, 5, 6, 8
PTR_ENV_START = 20

_Z = {'H': 1, 'He': 2, 'Li': 3, 'Be': 4, 'B': 5, 'C': 6, 'N': 7, 'O': 8, 'F': 9, 'Ne': 10,
      'Na': 11, 'Mg': 12, 'Al': 13, 'Si': 14, 'P': 15, 'S': 16, 'Cl': 17, 'Ar': 18}
# valence charges of the GTH-PADE pseudopotentials (pyscf/pbc/gto/pseudo/gth-pade.dat:58,99,113,150)
GTH_PADE_Q = {'H': 1, 'He': 2, 'C': 4, 'N': 5, 'O': 6, 'Mg': 10, 'Si': 4}

_BASIS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'basis')
_BASIS_ALIAS = {'gthszv': 'SZV-GTH', 'gthdzvp': 'DZVP-GTH'}


def cartesian_prod(arrays):
    """Cartesian product with the first array slowest (pyscf/lib/numpy_helper.py:925)."""
    grids = np.meshgrid(*arrays, indexing='ij')
    return np.stack([g.ravel() for g in grids], axis=1)


def parse_cp2k_basis(text, symb, family):
    """Return the shell table ``[[l, [exp, c1, c2, ...], ...], ...]`` of one element.

    CP2K layout: a header ``<symbol> <family>``, the number of sets, then per set a line
    ``n lmin lmax nexp nctr(lmin) ... nctr(lmax)`` followed by ``nexp`` rows of
    ``exp  c(lmin,1..) ... c(lmax,1..)``.  Primitives whose coefficients are all zero for a shell
    are dropped and shells are ordered by l (parse_cp2k.py:72-118, parse_nwchem.py:281-297).
    """
    lines = [ln.split('#')[0].strip() for ln in text.splitlines()]
    lines = [ln for ln in lines if ln]
    it = None
    for i, ln in enumerate(lines):
        tok = ln.split()
        if len(tok) == 2 and tok[0] == symb and tok[1].upper() == family.upper():
            it = iter(lines[i + 1:])
            break
    if it is None:
        raise KeyError('basis %s not found for %s' % (family, symb))
    nsets = int(next(it))
    shells = []
    for _ in range(nsets):
        comp = [int(x) for x in next(it).split()]
        lmin, lmax, nexp, nctr = comp[1], comp[2], comp[3], comp[4:]
        per_l = [[l] for l in range(lmin, lmax + 1)]
        for _ in range(nexp):
            row = [float(x) for x in next(it).split()]
            if len(row) != sum(nctr) + 1:
                raise ValueError('basis data incomplete')
            e, cs = row[0], row[1:]
            off = 0
            for i in range(lmax - lmin + 1):
                per_l[i].append([e] + cs[off:off + nctr[i]])
                off += nctr[i]
        shells.extend(per_l)
    shells = sorted(shells, key=lambda b: b[0])
    out = []
    for b in shells:
        rows = [r for r in b[1:] if any(c != 0 for c in r[1:])]
        if rows:
            out.append([b[0]] + rows)
    return out


def load_basis(name, symb):
    key = re.sub(r'[-_ ]', '', name).lower()
    if key not in _BASIS_ALIAS:
        raise KeyError('basis %r is not bundled (bundled: gth-szv, gth-dzvp); pass an explicit shell table' % name)
    with open(os.path.join(_BASIS_DIR, 'gth_cp2k.dat')) as f:
        return parse_cp2k_basis(f.read(), symb, _BASIS_ALIAS[key])


def load_pseudo(name, symb):
    """GTH pseudopotential parameters of one element in PySCF's internal layout
    ``[nelec_per_l, rloc, nexp, cexp, nproj_types, [r_l, n_l, h_l], ...]`` (pyscf/pbc/gto/pseudo/parse_cp2k.py)."""
    key = re.sub(r'[-_ ]', '', name).lower()
    if key not in ('gthpade', 'gthlda'):
        raise KeyError('pseudopotential %r is not bundled (bundled: gth-pade)' % name)
    with open(os.path.join(_BASIS_DIR, 'gth_pade_pp.dat')) as f:
        lines = [ln.split('#')[0].strip() for ln in f.read().splitlines()]
    lines = [ln for ln in lines if ln]
    for i, ln in enumerate(lines):
        tok = ln.split()
        if tok[0] == symb and any(t.upper().startswith('GTH-PADE') for t in tok[1:]):
            it = iter(lines[i + 1:])
            break
    else:
        raise KeyError('no bundled GTH-PADE pseudopotential for %s' % symb)
    nelec = [int(x) for x in next(it).split()]
    row = next(it).split()
    rloc, nexp = float(row[0]), int(row[1])
    cexp = [float(x) for x in row[2:2 + nexp]]
    nproj_types = int(next(it))
    out = [nelec, rloc, nexp, cexp, nproj_types]
    for _ in range(nproj_types):
        row = next(it).split()
        rl, nl = float(row[0]), int(row[1])
        vals = [float(x) for x in row[2:]]
        while len(vals) < nl * (nl + 1) // 2:
            vals += [float(x) for x in next(it).split()]
        h = np.zeros((nl, nl))
        if nl:
            h[np.triu_indices(nl)] = vals
            h = h + h.T - np.diag(h.diagonal())
        out.append([rl, nl, h.tolist()])
    return out


def gaussian_int(n, alpha):
    """int_0^inf x^n exp(-alpha x^2) dx (mole.py:116-119)."""
    n1 = (n + 1) * .5
    return _gamma(n1) / (2. * alpha ** n1)


def gto_norm(l, expnt):
    return 1. / np.sqrt(gaussian_int(l * 2 + 2, 2 * expnt))


def _normalize_contracted_ao(l, es, cs):
    ee = es.reshape(-1, 1) + es.reshape(1, -1)
    ee = gaussian_int(l * 2 + 2, ee)
    s1 = 1. / np.sqrt(np.einsum('pi,pq,qi->i', cs, ee, cs))
    return cs * s1


def _parse_atoms(atom, unit):
    if isinstance(atom, str):
        items = [a.split() for a in re.split(r'[;\n]', atom) if a.strip()]
        atoms = [(a[0], [float(x) for x in a[1:4]]) for a in items]
    else:
        atoms = [(a[0], [float(x) for x in (a[1] if len(a) == 2 else a[1:4])]) for a in atom]
    scale = 1. if unit.upper().startswith(('B', 'AU')) else 1. / BOHR
    return [(s, np.asarray(c, dtype=float) * scale) for s, c in atoms]


def _std_symbol(s):
    s = re.sub(r'[^A-Za-z]', '', s)
    return s[0].upper() + s[1:].lower()


class Cell:
    """Minimal periodic cell (3-D, Γ or k-points) with PySCF's attribute names."""

    dimension = 3
    omega = 0
    low_dim_ft_type = None

    def __init__(self, atom=None, a=None, basis=None, mesh=None, unit='Angstrom', precision=1e-8,
                 pseudo=None, verbose=0):
        self.atom, self.a, self.basis, self.mesh = atom, a, basis, mesh
        self.unit, self.precision, self.pseudo, self.verbose = unit, precision, pseudo, verbose
        self.stdout = None
        self.max_memory = 4000
        self._built = False
        self._rcut = None
        if atom is not None and a is not None and basis is not None:
            self.build()

    # ------------------------------------------------------------------------------------------
    def build(self):
        self._atom = _parse_atoms(self.atom, self.unit)
        scale = 1. if self.unit.upper().startswith(('B', 'AU')) else 1. / BOHR
        self._a = np.asarray(self.a, dtype=float).reshape(3, 3) * scale
        if self.mesh is None:
            raise ValueError('set cell.mesh explicitly (SURVEY 7.3-4)')
        self.mesh = np.asarray(self.mesh, dtype=int).reshape(3)

        # per-element shell tables
        symbols = []
        for s, _ in self._atom:
            if s not in symbols:
                symbols.append(s)
        self._basis = {}
        for s in symbols:
            std = _std_symbol(s)
            b = self.basis
            if isinstance(b, dict):
                b = b.get(s, b.get(std, b.get('default')))
                if b is None:
                    raise KeyError('no basis for %s' % s)
            if isinstance(b, str):
                b = load_basis(b, std)
            else:
                b = [[x[0]] + [list(np.ravel(r)) for r in x[1:]] for x in b]
                b = sorted(b, key=lambda x: x[0])
            self._basis[s] = b

        self._pseudo = {}
        for s in symbols:
            std = _std_symbol(s)
            if self._has_pseudo(std):
                p = self.pseudo
                pname = p.get(s, p.get(std)) if isinstance(p, dict) else p
                try:
                    self._pseudo[s] = load_pseudo(pname, std)
                except KeyError:
                    pass        # charge table still applies; get_pp will complain if the parameters are needed

        # libcint tables (mole.py:1025-1100)
        env = [np.zeros(PTR_ENV_START)]
        ptr = PTR_ENV_START
        atm = []
        for s, c in self._atom:
            std = _std_symbol(s)
            z = _Z.get(std, 0)
            if self._has_pseudo(std):
                z = sum(self._pseudo[s][0]) if s in self._pseudo else GTH_PADE_Q[std]
            atm.append([z, ptr, 1, ptr + 3, 0, 0])
            env.append(np.append(c, 0.))
            ptr += 4
        basdic = {}
        for s in symbols:
            rows = []
            for b in self._basis[s]:
                l = b[0]
                ec = np.array(sorted(b[1:], reverse=True))   # descending exponents (mole.py:995-1000)
                es, cs = ec[:, 0], ec[:, 1:]
                nprim, nctr = cs.shape
                cs = cs * gto_norm(l, es)[:, None]
                cs = _normalize_contracted_ao(l, es, cs)
                env.append(es)
                env.append(cs.T.reshape(-1))
                rows.append([0, l, nprim, nctr, 0, ptr, ptr + nprim, 0])
                ptr += nprim + nprim * nctr
            basdic[s] = np.array(rows, dtype=np.int32).reshape(-1, BAS_SLOTS)
        bas = []
        for ia, (s, _) in enumerate(self._atom):
            b = basdic[s].copy()
            b[:, ATOM_OF] = ia
            bas.append(b)
        self._atm = np.array(atm, dtype=np.int32).reshape(-1, ATM_SLOTS)
        self._bas = np.vstack(bas).astype(np.int32)
        self._env = np.hstack(env).astype(np.float64)
        self._built = True
        self._rcut = estimate_rcut(self, self.precision)
        return self

    def _has_pseudo(self, std):
        p = self.pseudo
        if p is None:
            return False
        if isinstance(p, dict):
            return std in p
        return True

    # ------------------------------------------------------------------------------------------
    @property
    def natm(self):
        return len(self._atm)

    @property
    def nbas(self):
        return len(self._bas)

    @property
    def rcut(self):
        return self._rcut

    @rcut.setter
    def rcut(self, x):
        self._rcut = x

    @property
    def vol(self):
        return abs(np.linalg.det(self._a))

    @property
    def nelectron(self):
        return int(self._atm[:, CHARGE_OF].sum())

    @property
    def nao(self):
        return self.nao_nr()

    def lattice_vectors(self):
        return self._a

    def reciprocal_vectors(self, norm_to=2 * np.pi):
        return norm_to * np.linalg.inv(self._a.T)

    def atom_coords(self):
        return np.array([c for _, c in self._atom])

    def atom_symbol(self, i):
        return self._atom[i][0]

    def atom_charges(self):
        return self._atm[:, CHARGE_OF].copy()

    def bas_atom(self, i):
        return int(self._bas[i, ATOM_OF])

    def bas_angular(self, i):
        return int(self._bas[i, ANG_OF])

    def bas_nprim(self, i):
        return int(self._bas[i, NPRIM_OF])

    def bas_nctr(self, i):
        return int(self._bas[i, NCTR_OF])

    def bas_exp(self, i):
        p = self._bas[i, PTR_EXP]
        return self._env[p:p + self._bas[i, NPRIM_OF]].copy()

    def _libcint_ctr_coeff(self, i):
        nprim, nctr = self.bas_nprim(i), self.bas_nctr(i)
        p = self._bas[i, PTR_COEFF]
        return self._env[p:p + nprim * nctr].reshape(nctr, nprim).T

    def ao_loc_nr(self):
        dims = (self._bas[:, ANG_OF] * 2 + 1) * self._bas[:, NCTR_OF]
        return np.append(0, np.cumsum(dims)).astype(np.int32)

    def nao_nr(self):
        return int(self.ao_loc_nr()[-1])

    def aoslice_by_atom(self):
        """(natm, 2) array of [ao_start, ao_end) per atom."""
        loc = self.ao_loc_nr()
        out = np.zeros((self.natm, 2), dtype=np.int64)
        for ia in range(self.natm):
            sh = np.where(self._bas[:, ATOM_OF] == ia)[0]
            if len(sh):
                out[ia] = loc[sh[0]], loc[sh[-1] + 1]
        return out

    def get_uniform_grids(self, mesh=None, wrap_around=True):
        """(G,3) grid coordinates, C order over (x,y,z) (cell.py:874-898)."""
        if mesh is None:
            mesh = self.mesh
        if wrap_around:
            qv = cartesian_prod([np.fft.fftfreq(x) for x in mesh])
            return np.dot(qv, self.lattice_vectors())
        qv = cartesian_prod([np.arange(x) for x in mesh])
        a_frac = np.einsum('i,ij->ij', 1. / np.asarray(mesh, dtype=float), self.lattice_vectors())
        return np.dot(qv, a_frac)

    gen_uniform_grids = get_uniform_grids

    def get_Gv(self, mesh=None):
        """(G,3) reciprocal vectors in fftfreq order (cell.py:537-603, lib/pbc/cell.c:122-147)."""
        if mesh is None:
            mesh = self.mesh
        b = self.reciprocal_vectors()
        rx = np.fft.fftfreq(mesh[0], 1. / mesh[0])
        ry = np.fft.fftfreq(mesh[1], 1. / mesh[1])
        rz = np.fft.fftfreq(mesh[2], 1. / mesh[2])
        Gv = (rx[:, None, None, None] * b[0] + ry[None, :, None, None] * b[1]
              + rz[None, None, :, None] * b[2])
        return Gv.reshape(-1, 3)

    def make_kpts(self, nks, wrap_around=False, with_gamma_point=True, scaled_center=None,
                  space_group_symmetry=False, time_reversal_symmetry=False, **kwargs):
        """Monkhorst-Pack mesh (pyscf/pbc/gto/cell.py:815-872): Gamma centred unless with_gamma_point=False; with
        space_group_symmetry / time_reversal_symmetry a kpts_symm.KPoints object (further keywords, e.g. symmorphic=True, go to
        its build)."""
        ks_each_axis = []
        for n in nks:
            if with_gamma_point or scaled_center is not None:
                ks = np.arange(n, dtype=float) / n
            else:
                ks = (np.arange(n) + .5) / n - .5
            if wrap_around:
                ks[ks >= .5] -= 1
            ks_each_axis.append(ks)
        scaled = cartesian_prod(ks_each_axis)
        if scaled_center is not None:
            scaled = scaled + np.asarray(scaled_center, dtype=float)
        kpts = np.dot(scaled, self.reciprocal_vectors())
        if space_group_symmetry or time_reversal_symmetry:
            from . import kpts_symm
            return kpts_symm.make_kpts(self, kpts, space_group_symmetry, time_reversal_symmetry, **kwargs)
        return kpts


# ----------------------------------------------------------------------------------------------
def _estimate_rcut_overlap(alpha, l, c, precision):
    """cell.py:390-406."""
    theta = alpha * .5
    a1 = (alpha * 2) ** -.5
    norm_ang = (2 * l + 1) / (4 * np.pi)
    fac = 2 * np.pi * c ** 2 * norm_ang / theta / precision
    r0 = 20
    fac = fac * 4 * alpha ** 2
    r0 = (np.log(fac * r0 * (r0 * .5 + a1) ** (2 * l + 2) + 1.) / theta) ** .5
    r0 = (np.log(fac * r0 * (r0 * .5 + a1) ** (2 * l + 2) + 1.) / theta) ** .5
    return r0


def estimate_rcut(cell, precision=None):
    """Lattice-sum cutoff of the whole cell: most diffuse primitive of every shell (cell.py:422-434)."""
    if cell.nbas == 0:
        return 0.01
    if precision is None:
        precision = cell.precision
    es, cs = [], []
    for i in range(cell.nbas):
        e = cell.bas_exp(i)
        c = cell._libcint_ctr_coeff(i)
        idx = e.argmin()
        es.append(e[idx])
        cs.append(abs(c[idx]).max())
    ls = np.array([cell.bas_angular(i) for i in range(cell.nbas)])
    return float(_estimate_rcut_overlap(np.array(es), ls, np.array(cs), precision).max())


def estimate_rcut_per_shell(cell):
    """Radius beyond which each shell's value is below precision/vol (eval_gto.py:169-186)."""
    vol = cell.vol
    precision = cell.precision / max(vol, 1)
    rcut = []
    for ib in range(cell.nbas):
        l = cell.bas_angular(ib)
        es = cell.bas_exp(ib)
        cs = abs(cell._libcint_ctr_coeff(ib)).max(axis=1)
        norm_ang = ((2 * l + 1) / (4 * np.pi)) ** .5
        fac = 2 * np.pi / vol * cs * norm_ang / es / precision
        r = cell.rcut
        r = (np.log(fac * r ** (l + 1) + 1.) / es) ** .5
        r = (np.log(fac * r ** (l + 1) + 1.) / es) ** .5
        rcut.append(r.max())
    return np.array(rcut)


def get_lattice_Ls(cell, rcut=None):
    """Lattice translations T = n.a for the AO lattice sum (the role of pyscf/pbc/gto/eval_gto.py:188-253): every T for
    which some atom image R + T can come within ``rcut`` of a point of the grid box, sorted by |T| (stable, so that
    translations of equal length keep their lexicographic order and the summation order is reproducible).

    Own construction, slab test in fractional coordinates: the grid points live in the parallelepiped with fractional
    coordinates in [-1/2, 1] (both the wrap-around and the [0, 1) grid conventions); a point whose fractional coordinate
    along axis i lies e_i outside that interval is at least e_i * h_i away from the box, h_i = 1 / |column i of a^-1| being
    the spacing of the lattice planes of that axis.  An image is kept when max_i e_i h_i < rcut for at least one atom.  The
    list only has to be a superset of the images that contribute: the collocation kernel screens every (point, image)
    pair against the shell's own cutoff, so any such superset gives the same AO values."""
    if rcut is None:
        rcut = cell.rcut
    a = np.asarray(cell.lattice_vectors(), dtype=float)
    ainv = np.linalg.inv(a)
    spacing = 1.0 / np.linalg.norm(ainv, axis=0)
    frac = np.asarray(cell.atom_coords(), dtype=float).dot(ainv)
    lo, hi = -0.5, 1.0
    reach = rcut / spacing
    nlo = np.floor(lo - reach - frac.max(axis=0)).astype(int)
    nhi = np.ceil(hi + reach - frac.min(axis=0)).astype(int)
    n = cartesian_prod([np.arange(nlo[i], nhi[i] + 1) for i in range(3)])
    pos = frac[:, None, :] + n[None, :, :]
    outside = np.maximum(np.maximum(lo - pos, pos - hi), 0.0) * spacing
    keep = (outside.max(axis=2) < rcut).any(axis=0)
    Ls = n[keep].dot(a)
    return np.ascontiguousarray(Ls[np.argsort(np.linalg.norm(Ls, axis=1), kind='stable')])


def super_cell(cell, ncopy, mesh=None):
    """ncopy[0] x ncopy[1] x ncopy[2] supercell, atoms ordered image-major like
    pyscf/pbc/tools/pbc.py:595-700 (translations x atoms).  ``mesh`` must be given explicitly."""
    a = cell.lattice_vectors()
    Ts = cartesian_prod([np.arange(n) for n in ncopy])
    Ls = np.dot(Ts, a)
    coords = cell.atom_coords()
    atoms = []
    for L in Ls:
        for i in range(cell.natm):
            atoms.append((cell.atom_symbol(i), coords[i] + L))
    sup_a = np.einsum('i,ij->ij', np.asarray(ncopy, dtype=float), a)
    if mesh is None:
        mesh = np.asarray(ncopy) * np.asarray(cell.mesh)
    return Cell(atom=atoms, a=sup_a, basis=cell.basis, mesh=mesh, unit='Bohr',
                precision=cell.precision, pseudo=cell.pseudo)


def madelung(cell, nk=(1, 1, 1)):
    """Madelung constant of the (nk-fold) cell: -2 x the Ewald energy of one unit point charge with a
    neutralising background (pyscf/pbc/tools/pbc.py:483-493, Ewald sum of pyscf/pbc/gto/cell.py:692-768).
    Used for the G=0 exchange correction exxdiv='ewald' (pyscf/pbc/df/df_jk.py:1446-1465)."""
    from scipy.special import erfc
    a = np.einsum('xi,x->xi', cell.lattice_vectors(), np.asarray(nk, dtype=float))
    vol = abs(np.linalg.det(a))
    b = 2 * np.pi * np.linalg.inv(a.T)
    eta = np.sqrt(np.pi) / vol ** (1. / 3)
    # real-space and reciprocal-space cutoffs for ~1e-14 relative truncation
    rmax = 6.5 / eta
    gmax = 13.0 * eta
    heights_inv = np.linalg.norm(b, axis=1) / (2 * np.pi)        # 1 / plane spacing of the direct lattice
    nr = np.ceil(rmax * heights_inv).astype(int) + 1
    Ts = cartesian_prod([np.arange(-n, n + 1) for n in nr]).dot(a)
    r = np.linalg.norm(Ts, axis=1)
    r = r[(r > 1e-12) & (r < rmax)]
    e_real = .5 * (erfc(eta * r) / r).sum()
    gheights_inv = np.linalg.norm(a, axis=1) / (2 * np.pi)
    ng = np.ceil(gmax * gheights_inv).astype(int) + 1
    Gs = cartesian_prod([np.arange(-n, n + 1) for n in ng]).dot(b)
    g2 = np.einsum('gi,gi->g', Gs, Gs)
    g2 = g2[(g2 > 1e-12) & (g2 < gmax * gmax)]
    e_recip = .5 * (4 * np.pi / vol) * (np.exp(-g2 / (4 * eta * eta)) / g2).sum()
    e_self = -eta / np.sqrt(np.pi) - .5 * np.pi / (eta * eta * vol)
    return -2. * (e_real + e_recip + e_self)


# ---- benchmark geometries ----------------------------------------------
def diamond_primitive(basis='gth-szv', mesh=(40, 40, 40)):
    """Diamond primitive cell, a0 = 3.5668 A (pyscf/pbc/tools/make_test_cell.py:95-111)."""
    a0 = 3.5668
    a = np.array([[0., a0 / 2, a0 / 2], [a0 / 2, 0., a0 / 2], [a0 / 2, a0 / 2, 0.]])
    atom = [('C', (0., 0., 0.)), ('C', (a0 / 4, a0 / 4, a0 / 4))]
    return Cell(atom=atom, a=a, basis=basis, mesh=mesh, pseudo='gth-pade')


def diamond_supercell(n, basis='gth-dzvp', mesh=None):
    prim = diamond_primitive(basis=basis, mesh=(8, 8, 8))
    return super_cell(prim, [n, n, n], mesh=mesh)
