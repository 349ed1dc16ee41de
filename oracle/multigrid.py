"""Oracle: multigrid J and LDA potential (numpy).  TEST INFRASTRUCTURE ONLY - never imported by the product.

CPU restatement of pyscf/pbc/dft/multigrid/multigrid.py for the Gamma point:
  * primitive cutoffs        multigrid.py:1825-1850 (_primitive_gto_cutoff), pyscf/pbc/gto/cell.py:436-448 (_estimate_ke_cutoff)
  * task ladder              multigrid.py:1696-1822 (multi_grids_tasks_for_ke_cut: windows (ke0, ke1] growing by KE_RATIO = 1.3 from
                             the cutoff of the 12^3 (orthogonal) / 32^3 mesh, odd meshes from pbc/tools/pbc.py:703-727, the last
                             task takes every remaining shell on the FFT mesh; dense shells keep the primitives inside the
                             window, sparse shells those below it)
  * density pass             multigrid.py:531-678 (_eval_rhoG: level density, tools.fft, weight vol/ngrids, _takebak_4d at the
                             numpy.fft.fftfreq indices of the dense mesh)
  * potential pass           multigrid.py:838-935 (_get_j_pass2: _take_4d, tools.ifft, real part, level integrals scattered to
                             [h,h], [h,l] and the transposed [l,h])
  * J, rho, nr_rks (LDA)     multigrid.py:500-529, :1556-1570, :1046-1150
with FULL complex spectra like the reference (the product uses half spectra).  A "task" here is a plain dict
{mesh, bas, env, nH, idx_h, idx_l, Ls, rcut}: the same record the product's planner emits, so tests can run this oracle on
the product's ladder (tight comparison, same arithmetic up to rounding) as well as on the reference's own ladder (both must
then agree with the FFTDF J to the cell precision, which is the reference's own test: test_multigrid.py:112-131).

The lattice-sum inputs of a task's collocation cell (Ls, rcut per shell) come from the caller through ``lattice_fn``: they
are estimates the tests take from the same helpers as every other oracle AO evaluation.

Parity: the reference's multigrid tests compare its J / veff with its own FFTDF J / numint veff at run time (no stored constants
for 'lda,'); tests do the same against oracle/fftdf.py, which is pinned to the reference's constants (tests/test_oracle_pins.py).
The LDA path as a whole is pinned by a reference constant one level up: the RKS 'lda,' total energy of
pyscf/pbc/scf/test/test_newton.py:84-90 (-9.7670882971475663) is reproduced to 5e-9 Eh by an SCF whose J + XC come from the
product's ladder (tests/test_gpu_scf.py, profiles/r02_scf_pins_diamond_prim.log).  The Slater exchange is the closed formula, libxc is absent: densities
at or below 1e-24 give zero here and in the product; what libxc does below its own threshold is not pinned.
"""
import numpy as np
from . import ao as oao, pbc_tools as tools

ATOM_OF, ANG_OF, NPRIM_OF, NCTR_OF, PTR_EXP, PTR_COEFF, BAS_SLOTS = 0, 1, 2, 3, 5, 6, 8
KE_RATIO = 1.3                      # multigrid.py:58
INIT_MESH_ORTH = (12, 12, 12)       # multigrid.py:56
INIT_MESH_NONORTH = (32, 32, 32)    # multigrid.py:57


def _ctr_coeff(bas, env, ib):
    nprim, nctr = bas[ib, NPRIM_OF], bas[ib, NCTR_OF]
    p = bas[ib, PTR_COEFF]
    return env[p:p + nprim * nctr].reshape(nctr, nprim).T


def _exps(bas, env, ib):
    return env[bas[ib, PTR_EXP]:bas[ib, PTR_EXP] + bas[ib, NPRIM_OF]]


def estimate_ke_cutoff(alpha, l, c, precision):
    """cell.py:436-448 with omega = 0."""
    norm_ang = (2 * l + 1) / (4 * np.pi)
    fac = 32 * np.pi ** 2 * (2 * np.pi) ** 1.5 * c ** 2 * norm_ang / (2 * alpha) ** (2 * l + .5) / precision
    Ecut = 20.
    Ecut = np.log(fac * (Ecut * 2) ** (l - .5) + 1.) * 4 * alpha
    Ecut = np.log(fac * (Ecut * 2) ** (l - .5) + 1.) * 4 * alpha
    return Ecut


def primitive_ke_cutoff(bas, env, vol, precision):
    """multigrid.py:1825-1850: per shell, per primitive."""
    precision = precision / max(vol, 1)
    out = []
    for ib in range(len(bas)):
        cs = abs(_ctr_coeff(bas, env, ib)).max(axis=1)
        out.append(estimate_ke_cutoff(_exps(bas, env, ib), bas[ib, ANG_OF], cs, precision))
    return out


def cutoff_to_mesh(a, cutoff):
    """pbc.py:703-727."""
    b = 2 * np.pi * np.linalg.inv(a.T)
    rx = np.linalg.qr(b[[1, 2, 0]].T)[1][2, 2]
    ry = np.linalg.qr(b[[2, 0, 1]].T)[1][2, 2]
    rz = np.linalg.qr(b.T)[1][2, 2]
    Gmax = (2 * cutoff) ** .5 / np.abs([rx, ry, rz])
    return np.ceil(Gmax).astype(int) * 2 + 1


def mesh_to_cutoff(a, mesh):
    """pbc.py:729-742."""
    b = 2 * np.pi * np.linalg.inv(a.T)
    rx = np.linalg.qr(b[[1, 2, 0]].T)[1][2, 2]
    ry = np.linalg.qr(b[[2, 0, 1]].T)[1][2, 2]
    rz = np.linalg.qr(b.T)[1][2, 2]
    gs = (np.asarray(mesh) - 1) // 2
    return (gs * np.array([rx, ry, rz])) ** 2 / 2


def _sub_cell(bas, env, shells, keep_of):
    """Shell rows ``shells`` with the primitives keep_of(ib) (multigrid.py:1706-1757: the contraction matrix keeps its kept
    rows; shells keep all their contractions, as the reference does)."""
    ao_loc = oao.ao_loc(bas)
    env = list(env)
    rows, idx = [], []
    for ib in shells:
        keep = keep_of(ib)
        cs = _ctr_coeff(bas, np.asarray(env), ib)[keep]
        es = _exps(bas, np.asarray(env), ib)[keep]
        ptr = len(env)
        env += list(es) + list(cs.T.ravel())
        row = bas[ib].copy()
        row[NPRIM_OF] = len(keep)
        row[PTR_EXP] = ptr
        row[PTR_COEFF] = ptr + len(keep)
        rows.append(row)
        idx.extend(range(ao_loc[ib], ao_loc[ib + 1]))
    return np.asarray(rows, dtype=np.int32).reshape(-1, BAS_SLOTS), np.asarray(env, dtype=np.float64), idx


def reference_tasks(bas, env, a, fft_mesh, precision, lattice_fn):
    """The reference's ke-cut ladder (multigrid.py:1696-1822).  lattice_fn(bas_sub, env_sub) -> (Ls, rcut per shell)."""
    bas = np.asarray(bas).reshape(-1, BAS_SLOTS)
    env = np.asarray(env, dtype=np.float64)
    a = np.asarray(a, dtype=float)
    fft_mesh = np.asarray(fft_mesh)
    vol = abs(np.linalg.det(a))
    ke_prim = primitive_ke_cutoff(bas, env, vol, precision)
    orth = abs(a - np.diag(a.diagonal())).max() < 1e-12
    ke1 = mesh_to_cutoff(a, INIT_MESH_ORTH if orth else INIT_MESH_NONORTH).min()
    ke_max = max(k.max() for k in ke_prim)
    delim = [0, ke1]
    while ke1 < ke_max:
        ke1 *= KE_RATIO
        delim.append(ke1)
    tasks = []
    for ke0, ke1 in zip(delim[:-1], delim[1:]):
        shls_dense = [ib for ib, ke in enumerate(ke_prim) if np.any((ke0 < ke) & (ke <= ke1))]
        if not shls_dense:
            continue
        mesh = cutoff_to_mesh(a, ke1)
        last = bool(np.all(mesh >= fft_mesh))
        if last:
            shls_dense = [ib for ib, ke in enumerate(ke_prim) if np.any(ke0 < ke)]
            top = ke_max + 1
        else:
            top = ke1
        mesh = np.min([mesh, fft_mesh], axis=0)
        bas_h, env_t, idx_h = _sub_cell(bas, env, shls_dense, lambda ib: np.where((ke0 < ke_prim[ib]) & (ke_prim[ib] <= top))[0])
        shls_sparse = [ib for ib, ke in enumerate(ke_prim) if np.any(ke <= ke0)]
        if shls_sparse:
            bas_l, env_t, idx_l = _sub_cell(bas, env_t, shls_sparse, lambda ib: np.where(ke_prim[ib] <= ke0)[0])
            bas_t = np.vstack([bas_h, bas_l])
        else:
            bas_t, idx_l = bas_h, []
        Ls, rcut = lattice_fn(bas_t, env_t)
        tasks.append(dict(mesh=mesh, bas=bas_t, env=env_t, nH=len(idx_h), idx_h=np.asarray(idx_h, dtype=np.int64),
                          idx_l=np.asarray(idx_l, dtype=np.int64), Ls=Ls, rcut=rcut))
        if last:
            break
    return tasks


def uniform_grids(a, mesh):
    """cell.py:874-898 with its default wrap_around=True: fractional coordinates numpy.fft.fftfreq(n) per axis, C order."""
    fr = [np.fft.fftfreq(int(n)) for n in mesh]
    f = np.stack(np.meshgrid(*fr, indexing='ij'), axis=-1).reshape(-1, 3)
    return f.dot(a)


def _task_ao(task, atm, a):
    coords = uniform_grids(np.asarray(a, dtype=float), task['mesh'])
    return oao.eval_ao(atm, task['bas'], task['env'], coords, task['Ls'], task['rcut'], rule='point')     # (G_t, nT)


def _freq_index(mesh, fft_mesh):
    # numpy.fft.fftfreq integers of the level mesh used as (possibly negative) indices of the dense mesh (multigrid.py:669-673)
    return [np.fft.fftfreq(n, 1. / n).astype(np.int32) % N for n, N in zip(mesh, fft_mesh)]


def eval_rhoG(tasks, atm, dms, a, fft_mesh):
    """(nset, N0, N1, N2) complex density spectrum, integral-normalised (multigrid.py:531-678, hermi = 1, LDA)."""
    a = np.asarray(a, dtype=float)
    vol = abs(np.linalg.det(a))
    dms = np.asarray(dms, dtype=float)
    dms = 0.5 * (dms + dms.transpose(0, 2, 1))
    nset = len(dms)
    rhoG = np.zeros((nset,) + tuple(int(x) for x in fft_mesh), dtype=np.complex128)
    for t in tasks:
        ao = _task_ao(t, atm, a)
        nH, idx_h, idx_l = t['nH'], t['idx_h'], t['idx_l']
        idx_t = np.append(idx_h, idx_l)
        mesh = tuple(int(x) for x in t['mesh'])
        ngrids = int(np.prod(mesh))
        rho = np.empty((nset, ngrids))
        for i in range(nset):
            d = dms[i][idx_h[:, None], idx_t].copy()
            if len(idx_l):
                d[:, nH:] += dms[i][idx_l[:, None], idx_h].T
            rho[i] = np.einsum('gh,ht,gt->g', ao[:, :nH], d, ao, optimize=True)
        rho_freq = tools.fft(rho, mesh) * (vol / ngrids)
        gx, gy, gz = _freq_index(mesh, fft_mesh)
        rhoG[:, gx[:, None, None], gy[:, None], gz] += rho_freq.reshape((nset,) + mesh)
    return rhoG


def integrate(tasks, atm, vG, a, fft_mesh, nao):
    """Matrix of the potential with spectrum vG (nset, N0, N1, N2) (multigrid.py:838-935, hermi = 1)."""
    nset = len(vG)
    out = np.zeros((nset, nao, nao))
    for t in tasks:
        ao = _task_ao(t, atm, a)
        nH, idx_h, idx_l = t['nH'], t['idx_h'], t['idx_l']
        mesh = tuple(int(x) for x in t['mesh'])
        gx, gy, gz = _freq_index(mesh, fft_mesh)
        sub = vG[:, gx[:, None, None], gy[:, None], gz].reshape(nset, -1)
        v = tools.ifft(sub, mesh).real
        for i in range(nset):
            vp = ao[:, :nH].T.dot(v[i][:, None] * ao)
            out[i][idx_h[:, None], idx_h] += vp[:, :nH]
            if len(idx_l):
                out[i][idx_h[:, None], idx_l] += vp[:, nH:]
                out[i][idx_l[:, None], idx_h] += vp[:, nH:].T
    return out


def get_j(tasks, atm, dms, a, fft_mesh):
    """multigrid.py:500-529."""
    dms = np.asarray(dms, dtype=float)
    nao = dms.shape[-1]
    rhoG = eval_rhoG(tasks, atm, dms.reshape(-1, nao, nao), a, fft_mesh)
    coulG = tools.get_coulG(np.asarray(a, dtype=float), np.asarray(fft_mesh)).reshape(rhoG.shape[1:])
    return integrate(tasks, atm, rhoG * coulG, a, fft_mesh, nao).reshape(dms.shape)


def get_rho(tasks, atm, dm, a, fft_mesh):
    """multigrid.py:1556-1570."""
    nao = np.asarray(dm).shape[-1]
    rhoG = eval_rhoG(tasks, atm, np.asarray(dm, dtype=float).reshape(-1, nao, nao), a, fft_mesh)
    ngrids = int(np.prod(fft_mesh))
    weight = abs(np.linalg.det(a)) / ngrids
    return tools.ifft(rhoG.reshape(len(rhoG), ngrids), fft_mesh).real / weight


def slater_exchange(rho):
    """Spin-unpolarised LDA exchange: exc per particle = -(3/4) (3/pi)^(1/3) rho^(1/3), vxc = (4/3) exc; rho <= 1e-24 -> 0."""
    e = np.zeros_like(rho)
    m = rho > 1e-24
    e[m] = -0.75 * (3.0 / np.pi) ** (1.0 / 3.0) * np.cbrt(rho[m])
    return e, 4.0 / 3.0 * e


VWN5 = (0.0310907, 3.72744, 12.9352, -0.10498)          # A (Hartree), b, c, x0: Vosko-Wilk-Nusair parametrisation V, paramagnetic


def vwn_correlation(rho):
    """Spin-unpolarised VWN5 correlation (libxc LDA_C_VWN, what the reference's 'lda,vwn' evaluates through
    pyscf/dft/libxc.py; Vosko, Wilk, Nusair, Can. J. Phys. 58, 1200 (1980), eq. 4.4 with the paramagnetic parameters of fit V):
        eps_c = A { ln(x^2/X) + 2b/Q atan(Q/(2x+b)) - b x0/X(x0) [ ln((x-x0)^2/X) + 2(b+2x0)/Q atan(Q/(2x+b)) ] },
        x = sqrt(rs), X(x) = x^2 + b x + c, Q = sqrt(4c - b^2);   v_c = eps_c - (x/6) d eps_c/dx.
    Returns (eps_c per particle, v_c); rho <= 1e-24 -> 0 like the exchange."""
    A, b, c, x0 = VWN5
    e = np.zeros_like(rho)
    v = np.zeros_like(rho)
    m = rho > 1e-24
    rs = np.cbrt(3.0 / (4.0 * np.pi * rho[m]))
    x = np.sqrt(rs)
    X = x * x + b * x + c
    X0 = x0 * x0 + b * x0 + c
    Q = np.sqrt(4.0 * c - b * b)
    at = np.arctan(Q / (2.0 * x + b))
    ec = A * (np.log(x * x / X) + 2.0 * b / Q * at - b * x0 / X0 * (np.log((x - x0) ** 2 / X) + 2.0 * (b + 2.0 * x0) / Q * at))
    den = Q * Q + (2.0 * x + b) ** 2
    dec = A * (2.0 / x - (2.0 * x + b) / X - 4.0 * b / den
               - b * x0 / X0 * (2.0 / (x - x0) - (2.0 * x + b) / X - 4.0 * (b + 2.0 * x0) / den))
    e[m] = ec
    v[m] = ec - x / 6.0 * dec
    return e, v


def lda_xc(rho, xc='lda,'):
    """(exc per particle, vxc) of 'lda,' (Slater exchange) or 'lda,vwn' (+ VWN5 correlation)."""
    e, v = slater_exchange(rho)
    if xc.replace(' ', '').lower() in ('lda,vwn', 'lda,vwn5', 'svwn', 'slater,vwn', 'slater,vwn5'):
        ec, vc = vwn_correlation(rho)
        e, v = e + ec, v + vc
    return e, v


def nr_rks_lda(tasks, atm, dm, a, fft_mesh, with_j=False, xc='lda,'):
    """(nelec, exc, veff, ecoul) of one density matrix: multigrid.py:1046-1150 with xc = 'lda,' (Slater exchange) or 'lda,vwn'."""
    a = np.asarray(a, dtype=float)
    nao = np.asarray(dm).shape[-1]
    fft_mesh = np.asarray(fft_mesh)
    ngrids = int(np.prod(fft_mesh))
    vol = abs(np.linalg.det(a))
    weight = vol / ngrids
    rhoG = eval_rhoG(tasks, atm, np.asarray(dm, dtype=float).reshape(1, nao, nao), a, fft_mesh)
    coulG = tools.get_coulG(a, fft_mesh).reshape(rhoG.shape[1:])
    vG = rhoG * coulG
    ecoul = .5 * (rhoG.real * vG.real).sum() + .5 * (rhoG.imag * vG.imag).sum()
    ecoul /= vol
    rhoR = tools.ifft(rhoG.reshape(1, ngrids), fft_mesh).real / weight
    nelec = rhoR[0].sum() * weight
    exc, vxc = lda_xc(rhoR[0], xc)
    excsum = (rhoR[0] * exc).sum() * weight
    wv_freq = tools.fft((weight * vxc)[None], fft_mesh).reshape(rhoG.shape)
    if with_j:
        wv_freq = wv_freq + vG
    veff = integrate(tasks, atm, wv_freq, a, fft_mesh, nao)[0]
    return nelec, excsum, veff, ecoul


def nr_rks_lda_dense(aoR, dm, a, fft_mesh, xc='lda,'):
    """The same quantities on the dense grid alone (role of pyscf.pbc.dft.numint.nr_rks for 'lda,' on uniform grids, which the
    reference's multigrid tests use as their answer: test_multigrid.py:133-142).  aoR (G, nao)."""
    ngrids = len(aoR)
    weight = abs(np.linalg.det(a)) / ngrids
    rho = np.einsum('gi,ij,gj->g', aoR, 0.5 * (dm + dm.T), aoR, optimize=True)
    exc, vxc = lda_xc(rho, xc)
    veff = aoR.T.dot((weight * vxc)[:, None] * aoR)
    return rho.sum() * weight, (rho * exc).sum() * weight, veff


# ---- k-points (multigrid.py:531-678 with nkpts > 1, :838-935 with complex AOs) ----------------------------------------------
def _task_ao_kpts(task, atm, a, kpts):
    coords = uniform_grids(np.asarray(a, dtype=float), task['mesh'])
    return [np.asarray(x, dtype=np.complex128)
            for x in oao.eval_ao(atm, task['bas'], task['env'], coords, task['Ls'], task['rcut'], kpts=np.reshape(kpts, (-1, 3)),
                                 rule='point')]


def eval_rhoG_kpts(tasks, atm, dms, a, fft_mesh, kpts):
    """(nset, N0, N1, N2) spectrum of rho = 1/nk sum_k sum_ij ao_i D_ij conj(ao_j) (multigrid.py:590-598), full Bloch functions;
    dms (nset, nk, nao, nao), Hermitian or not (the density is then complex, as in the reference's hermi = 0 branch)."""
    a = np.asarray(a, dtype=float)
    vol = abs(np.linalg.det(a))
    dms = np.asarray(dms, dtype=np.complex128)
    nset, nk = dms.shape[:2]
    rhoG = np.zeros((nset,) + tuple(int(x) for x in fft_mesh), dtype=np.complex128)
    for t in tasks:
        aos = _task_ao_kpts(t, atm, a, kpts)
        nH, idx_h, idx_l = t['nH'], t['idx_h'], t['idx_l']
        idx_t = np.append(idx_h, idx_l)
        mesh = tuple(int(x) for x in t['mesh'])
        ngrids = int(np.prod(mesh))
        rho = np.zeros((nset, ngrids), dtype=np.complex128)
        for i in range(nset):
            for k in range(nk):
                ao = aos[k]
                rho[i] += np.einsum('gh,ht,gt->g', ao[:, :nH], dms[i, k][idx_h[:, None], idx_t], ao.conj(), optimize=True)
                if len(idx_l):
                    rho[i] += np.einsum('gl,lh,gh->g', ao[:, nH:], dms[i, k][idx_l[:, None], idx_h], ao[:, :nH].conj(), optimize=True)
        rho_freq = tools.fft(rho, mesh) * (vol / ngrids / nk)
        gx, gy, gz = _freq_index(mesh, fft_mesh)
        rhoG[:, gx[:, None, None], gy[:, None], gz] += rho_freq.reshape((nset,) + mesh)
    return rhoG


def integrate_kpts(tasks, atm, vG, a, fft_mesh, nao, kpts_band):
    """(nset, nband, nao, nao): conj(ao_i) v ao_j per band k-point for the potential with spectrum vG (complex in general)."""
    nset = len(vG)
    kpts_band = np.reshape(kpts_band, (-1, 3))
    out = np.zeros((nset, len(kpts_band), nao, nao), dtype=np.complex128)
    for t in tasks:
        aos = _task_ao_kpts(t, atm, a, kpts_band)
        nH, idx_h, idx_l = t['nH'], t['idx_h'], t['idx_l']
        mesh = tuple(int(x) for x in t['mesh'])
        gx, gy, gz = _freq_index(mesh, fft_mesh)
        sub = vG[:, gx[:, None, None], gy[:, None], gz].reshape(nset, -1)
        v = tools.ifft(sub, mesh)
        for i in range(nset):
            for ib, ao in enumerate(aos):
                vw = v[i][:, None] * ao
                out[i, ib][idx_h[:, None], idx_h] += ao[:, :nH].conj().T.dot(vw[:, :nH])
                if len(idx_l):
                    out[i, ib][idx_h[:, None], idx_l] += ao[:, :nH].conj().T.dot(vw[:, nH:])
                    out[i, ib][idx_l[:, None], idx_h] += ao[:, nH:].conj().T.dot(vw[:, :nH])
    return out


def get_j_kpts(tasks, atm, dms, a, fft_mesh, kpts, kpts_band=None):
    """multigrid.py:500-529 at k-points; dms (nk, nao, nao) -> (nband, nao, nao)."""
    dms = np.asarray(dms, dtype=np.complex128)
    nao = dms.shape[-1]
    rhoG = eval_rhoG_kpts(tasks, atm, dms[None], a, fft_mesh, kpts)
    coulG = tools.get_coulG(np.asarray(a, dtype=float), np.asarray(fft_mesh)).reshape(rhoG.shape[1:])
    return integrate_kpts(tasks, atm, rhoG * coulG, a, fft_mesh, nao, kpts if kpts_band is None else kpts_band)[0]


def nr_rks_lda_kpts(tasks, atm, dms, a, fft_mesh, kpts, with_j=False):
    """(nelec, exc, veff (nk, nao, nao)) at k-points, 'lda,' (multigrid.py:1046-1150; the functional sees Re rho)."""
    a = np.asarray(a, dtype=float)
    dms = np.asarray(dms, dtype=np.complex128)
    nao = dms.shape[-1]
    fft_mesh = np.asarray(fft_mesh)
    ngrids = int(np.prod(fft_mesh))
    vol = abs(np.linalg.det(a))
    weight = vol / ngrids
    rhoG = eval_rhoG_kpts(tasks, atm, dms[None], a, fft_mesh, kpts)
    coulG = tools.get_coulG(a, fft_mesh).reshape(rhoG.shape[1:])
    rhoR = tools.ifft(rhoG.reshape(1, ngrids), fft_mesh).real / weight
    exc, vxc = slater_exchange(rhoR[0])
    wv_freq = tools.fft((weight * vxc)[None], fft_mesh).reshape(rhoG.shape)
    if with_j:
        wv_freq = wv_freq + rhoG * coulG
    veff = integrate_kpts(tasks, atm, wv_freq, a, fft_mesh, nao, kpts)[0]
    return rhoR[0].sum() * weight, (rhoR[0] * exc).sum() * weight, veff


def nr_uks_lda(tasks, atm, dms, a, fft_mesh, with_j=False, kpts=None):
    """(nelec, exc, veff (2, ...), ecoul) of an (alpha, beta) pair: multigrid.py:1152-1257 with 'lda,'.  Spin-polarised Slater
    exchange in closed form: e = -(3/4)(3/pi)^(1/3) 2^(1/3) (rho_a^(4/3) + rho_b^(4/3)), v_s = -(3/pi)^(1/3) 2^(1/3) rho_s^(1/3)."""
    a = np.asarray(a, dtype=float)
    fft_mesh = np.asarray(fft_mesh)
    ngrids = int(np.prod(fft_mesh))
    vol = abs(np.linalg.det(a))
    weight = vol / ngrids
    nao = np.asarray(dms).shape[-1]
    if kpts is None:
        rhoG = eval_rhoG(tasks, atm, np.asarray(dms, dtype=float), a, fft_mesh)
    else:
        rhoG = eval_rhoG_kpts(tasks, atm, np.asarray(dms), a, fft_mesh, kpts)
    coulG = tools.get_coulG(a, fft_mesh).reshape(rhoG.shape[1:])
    vG = (rhoG[0] + rhoG[1]) * coulG
    tot = rhoG[0] + rhoG[1]
    ecoul = (.5 * (tot.real * vG.real).sum() + .5 * (tot.imag * vG.imag).sum()) / vol
    rhoR = tools.ifft(rhoG.reshape(2, ngrids), fft_mesh).real / weight
    c = -(3.0 / np.pi) ** (1.0 / 3.0) * 2.0 ** (1.0 / 3.0)
    pos = np.where(2 * rhoR > 1e-24, rhoR, 0.0)
    vxc = c * np.cbrt(pos)
    edens = 0.75 * c * (pos[0] ** (4.0 / 3.0) + pos[1] ** (4.0 / 3.0))
    wv_freq = tools.fft(weight * vxc, fft_mesh).reshape(rhoG.shape)
    if with_j:
        wv_freq = wv_freq + vG
    if kpts is None:
        veff = integrate(tasks, atm, wv_freq, a, fft_mesh, nao)
    else:
        veff = integrate_kpts(tasks, atm, wv_freq, a, fft_mesh, nao, kpts)
    return rhoR.sum() * weight, edens.sum() * weight, veff, ecoul


# ---- LDA response (multigrid.py:1259-1452), 'lda,' --------------------------------------------------------------------
def slater_exchange_fxc(rho):
    """d2(rho exc)/d rho2 = -(1/3)(3/pi)^(1/3) rho^(-2/3); rho <= 1e-24 -> 0."""
    f = np.zeros_like(rho)
    m = rho > 1e-24
    f[m] = -(1.0 / 3.0) * (3.0 / np.pi) ** (1.0 / 3.0) * np.cbrt(rho[m]) ** -2
    return f


def _rho_real_space(tasks, atm, dms, a, fft_mesh, kpts):
    """(nset, G) density of a stack of matrices, complex for non-Hermitian k-point matrices; and its spectrum."""
    fft_mesh = np.asarray(fft_mesh)
    ngrids = int(np.prod(fft_mesh))
    weight = abs(np.linalg.det(a)) / ngrids
    if kpts is None:
        rhoG = eval_rhoG(tasks, atm, dms, a, fft_mesh)
        return tools.ifft(rhoG.reshape(len(rhoG), ngrids), fft_mesh).real / weight, rhoG
    rhoG = eval_rhoG_kpts(tasks, atm, dms, a, fft_mesh, kpts)
    return tools.ifft(rhoG.reshape(len(rhoG), ngrids), fft_mesh) / weight, rhoG


def nr_fxc_lda(tasks, atm, dm0, dms, a, fft_mesh, kind='rks', with_j=False, kpts=None):
    """Response matrices of ``dms`` around ``dm0``: kind 'rks' (multigrid.py:1259-1318), 'st' (:1321-1386; singlet = triplet
    for exchange alone), 'uks' (:1389-1452; dm0 = (a, b), dms = (a responses..., b responses...)).  Closed forms of the
    spin-resolved kernel: f_aa(rho_a, rho_b) = -(1/3)(3/pi)^(1/3) 2^(1/3)... = (4/9) C 2^(1/3) rho_a^(-2/3), f_ab = 0."""
    a = np.asarray(a, dtype=float)
    fft_mesh = np.asarray(fft_mesh)
    ngrids = int(np.prod(fft_mesh))
    weight = abs(np.linalg.det(a)) / ngrids
    nao = np.asarray(dms).shape[-1]
    dm0 = np.asarray(dm0)
    dms = np.asarray(dms)
    nk = 0 if kpts is None else len(kpts)
    stack0 = dm0.reshape((-1, nao, nao) if kpts is None else (-1, nk, nao, nao))
    stack1 = dms.reshape((-1, nao, nao) if kpts is None else (-1, nk, nao, nao))
    rho0 = _rho_real_space(tasks, atm, stack0, a, fft_mesh, kpts)[0].real
    rho1, rhoG1 = _rho_real_space(tasks, atm, stack1, a, fft_mesh, kpts)
    c = (4.0 / 9.0) * (-0.75 * (3.0 / np.pi) ** (1.0 / 3.0))
    if kind == 'rks':
        f = slater_exchange_fxc(rho0[0])[None]
    elif kind == 'st':
        ra = 0.5 * rho0[0]
        f = np.where(2 * ra > 1e-24, c * 2.0 ** (1.0 / 3.0) * np.cbrt(np.where(ra > 0, ra, 1.0)) ** -2, 0.0)[None]
    else:
        f = np.where(2 * rho0 > 1e-24, c * 2.0 ** (1.0 / 3.0) * np.cbrt(np.where(rho0 > 0, rho0, 1.0)) ** -2, 0.0)
        f = np.repeat(f, len(stack1) // 2, axis=0)
    wv = tools.fft(weight * f * rho1, fft_mesh).reshape(rhoG1.shape)
    if with_j:
        coulG = tools.get_coulG(a, fft_mesh).reshape(rhoG1.shape[1:])
        if kind == 'uks':
            half = len(stack1) // 2
            tot = rhoG1[:half] + rhoG1[half:]
            wv = wv + np.concatenate([tot, tot]) * coulG
        else:
            wv = wv + rhoG1 * coulG
    if kpts is None:
        return integrate(tasks, atm, wv, a, fft_mesh, nao).reshape(dms.shape)
    return integrate_kpts(tasks, atm, wv, a, fft_mesh, nao, kpts).reshape(dms.shape)


# ---- GGA: Becke-88 exchange ('b88,'), Gamma point ------------------------------------------------------------------------
def b88_energy_density(rho, g):
    """e(rho, |grad rho|) of Becke's 1988 exchange for a spin-unpolarised density (Becke, PRA 38, 3098 eq. 8; libxc GGA_X_B88:
    beta = 0.0042, with the LDA exchange included): e = sum_s [-C_x rho_s^(4/3) - beta rho_s^(4/3) x_s^2 / (1 + 6 beta x_s asinh x_s)],
    x_s = |grad rho_s| / rho_s^(4/3), rho_s = rho / 2."""
    beta = 0.0042
    cx = 1.5 * (3.0 / (4.0 * np.pi)) ** (1.0 / 3.0)
    rs = 0.5 * np.asarray(rho, dtype=float)
    gs = 0.5 * np.asarray(g, dtype=float)
    r43 = rs ** (4.0 / 3.0)
    x = gs / r43
    return 2.0 * r43 * (-cx - beta * x * x / (1.0 + 6.0 * beta * x * np.arcsinh(x)))


def b88_exchange(rho, grad):
    """(exc per particle, vrho, w = de/d(grad rho)) with rho (G,), grad (3, G); derivatives in closed form (checked against
    finite differences of b88_energy_density in tests/test_multigrid.py); rho <= 1e-14 -> 0."""
    rho = np.asarray(rho, dtype=float)
    grad = np.asarray(grad, dtype=float)
    beta = 0.0042
    cx = 1.5 * (3.0 / (4.0 * np.pi)) ** (1.0 / 3.0)
    m = rho > 1e-14
    rs = np.where(m, 0.5 * rho, 1.0)
    gs = 0.5 * np.sqrt((grad ** 2).sum(axis=0))
    r13 = np.cbrt(rs)
    r43 = rs * r13
    x = gs / r43
    a = np.arcsinh(x)
    D = 1.0 + 6.0 * beta * x * a
    Dp = 6.0 * beta * (a + x / np.sqrt(1.0 + x * x))
    G = -cx - beta * x * x / D
    Gp_x = -beta * (2.0 * D - x * Dp) / (D * D)
    exc = np.where(m, 2.0 * r43 * G / np.where(m, rho, 1.0), 0.0)
    vrho = np.where(m, (4.0 / 3.0) * r13 * (G - x * x * Gp_x), 0.0)
    w = np.where(m, Gp_x / (2.0 * r43), 0.0)[None] * grad          # de/d|grad rho| = G'(x); x = |grad rho| / (2 rho_s^(4/3))
    return exc, vrho, w


def _rho4_dense(ao4, dm):
    """(4, G): density and its gradient from AO values and derivatives ao4 (4, G, nao) (numint.eval_rho, GGA, hermi = 1)."""
    dm = 0.5 * (dm + dm.T)
    c0 = ao4[0].dot(dm)
    rho = np.empty((4, ao4.shape[1]))
    rho[0] = np.einsum('gi,gi->g', c0, ao4[0])
    for x in range(1, 4):
        rho[x] = 2.0 * np.einsum('gi,gi->g', c0, ao4[x])
    return rho


def nr_rks_b88_dense(ao4, dm, a, fft_mesh):
    """(nelec, exc, vxc matrix) by quadrature on the dense uniform grid - what pyscf.pbc.dft.numint.nr_rks does for a GGA
    (numint.py:nr_rks via pyscf/dft/numint.py: wv[0] *= .5, V = ao0^T (sum_c wv_c ao_c), V + V^T): the reference's answer for
    'b88,' on a cell (test_newton.py:92-98 runs its SCF on exactly this)."""
    ngrids = ao4.shape[1]
    weight = abs(np.linalg.det(a)) / ngrids
    rho = _rho4_dense(ao4, dm)
    exc, vrho, w = b88_exchange(rho[0], rho[1:])
    wv = weight * np.vstack([0.5 * vrho[None], w])
    aow = sum(wv[c][:, None] * ao4[c] for c in range(4))
    v = ao4[0].T.dot(aow)
    return rho[0].sum() * weight, (rho[0] * exc).sum() * weight, v + v.T


def _task_ao4(task, atm, a):
    coords = uniform_grids(np.asarray(a, dtype=float), task['mesh'])
    return oao.eval_ao_deriv1(atm, task['bas'], task['env'], coords, task['Ls'], task['rcut'])          # (4, G_t, nT)


def eval_rhoG_gga(tasks, atm, dm, a, fft_mesh):
    """(4, N0, N1, N2): spectra of rho and of its three gradient components, every level's gradient taken in REAL space from the
    AO derivatives on the level mesh (the RHOG_HIGH_ORDER = True branch of multigrid.py:545-560,633-658; hermi = 1)."""
    a = np.asarray(a, dtype=float)
    vol = abs(np.linalg.det(a))
    dm = 0.5 * (np.asarray(dm, dtype=float) + np.asarray(dm, dtype=float).T)
    rhoG = np.zeros((4,) + tuple(int(x) for x in fft_mesh), dtype=np.complex128)
    for t in tasks:
        ao4 = _task_ao4(t, atm, a)
        nH, idx_h, idx_l = t['nH'], t['idx_h'], t['idx_l']
        idx_t = np.append(idx_h, idx_l)
        mesh = tuple(int(x) for x in t['mesh'])
        ngrids = int(np.prod(mesh))
        d = dm[idx_h[:, None], idx_t].copy()
        if len(idx_l):
            d[:, nH:] += dm[idx_l[:, None], idx_h].T
        rho = np.empty((4, ngrids))
        c0 = ao4[0].dot(d.T)                                   # (G, nH): sum_t D'_ht phi_t
        rho[0] = np.einsum('gh,gh->g', ao4[0][:, :nH], c0)
        for x in range(1, 4):
            rho[x] = np.einsum('gh,gh->g', ao4[x][:, :nH], c0) + np.einsum('gh,gh->g', ao4[0][:, :nH], ao4[x].dot(d.T))
        rho_freq = tools.fft(rho, mesh) * (vol / ngrids)
        gx, gy, gz = _freq_index(mesh, fft_mesh)
        rhoG[:, gx[:, None, None], gy[:, None], gz] += rho_freq.reshape((4,) + mesh)
    return rhoG


def integrate_gga(tasks, atm, wvG, a, fft_mesh, nao):
    """Potential matrix of a GGA potential given by the spectra wvG (4, N0, N1, N2) of (w vrho, w de/d grad rho): per level
    V_ht = sum_r phi_h [v0 phi_t + v_c d_c phi_t] + (d_c phi_h) v_c phi_t  (role of multigrid.py:936-1043, hermi = 1)."""
    out = np.zeros((nao, nao))
    for t in tasks:
        ao4 = _task_ao4(t, atm, a)
        nH, idx_h, idx_l = t['nH'], t['idx_h'], t['idx_l']
        mesh = tuple(int(x) for x in t['mesh'])
        gx, gy, gz = _freq_index(mesh, fft_mesh)
        sub = wvG[:, gx[:, None, None], gy[:, None], gz].reshape(4, -1)
        v = tools.ifft(sub, mesh).real
        vp = ao4[0][:, :nH].T.dot(v[0][:, None] * ao4[0])
        for x in range(1, 4):
            vp += ao4[0][:, :nH].T.dot(v[x][:, None] * ao4[x]) + ao4[x][:, :nH].T.dot(v[x][:, None] * ao4[0])
        out[idx_h[:, None], idx_h] += vp[:, :nH]
        if len(idx_l):
            out[idx_h[:, None], idx_l] += vp[:, nH:]
            out[idx_l[:, None], idx_h] += vp[:, nH:].T
    return out


def nr_rks_b88(tasks, atm, dm, a, fft_mesh, with_j=False):
    """(nelec, exc, veff, ecoul) through the ladder, 'b88,' (multigrid.py:1046-1150, GGA branch with real-space gradients)."""
    a = np.asarray(a, dtype=float)
    nao = np.asarray(dm).shape[-1]
    fft_mesh = np.asarray(fft_mesh)
    ngrids = int(np.prod(fft_mesh))
    vol = abs(np.linalg.det(a))
    weight = vol / ngrids
    rhoG = eval_rhoG_gga(tasks, atm, dm, a, fft_mesh)
    coulG = tools.get_coulG(a, fft_mesh).reshape(rhoG.shape[1:])
    vG = rhoG[0] * coulG
    ecoul = (.5 * (rhoG[0].real * vG.real).sum() + .5 * (rhoG[0].imag * vG.imag).sum()) / vol
    rhoR = tools.ifft(rhoG.reshape(4, ngrids), fft_mesh).real / weight
    exc, vrho, w = b88_exchange(rhoR[0], rhoR[1:])
    wv = weight * np.vstack([vrho[None], w])
    wvG = tools.fft(wv, fft_mesh).reshape(rhoG.shape)
    if with_j:
        wvG[0] += vG
    veff = integrate_gga(tasks, atm, wvG, a, fft_mesh, nao)
    return rhoR[0].sum() * weight, (rhoR[0] * exc).sum() * weight, veff, ecoul


def nr_rks_b88_dense_kpts(ao4_kpts, dms, a):
    """(nelec, exc, vxc (nk, nao, nao)) of 'b88,' by quadrature on the dense grid at k-points: what KNumInt.nr_rks does for a GGA
    (pyscf/pbc/dft/numint.py:1090-1292 with pyscf/dft/numint.py's GGA kernel: rho = 1/nk sum_k Re[...], wv[0] *= .5,
    V^k = ao0^H (sum_c wv_c ao_c) + h.c.).  ao4_kpts: list over k of (4, G, nao) complex Bloch functions and derivatives."""
    nk = len(ao4_kpts)
    ngrids = ao4_kpts[0].shape[1]
    weight = abs(np.linalg.det(a)) / ngrids
    rho = np.zeros((4, ngrids))
    for k in range(nk):
        ao4, dm = ao4_kpts[k], np.asarray(dms[k])
        c0 = ao4[0].dot(dm.T)                              # sum_j dm_ij ao_j ... rho = sum_ij conj(ao_i) dm_ij ao_j
        rho[0] += np.einsum('gi,gi->g', ao4[0].conj(), ao4[0].dot(dm.T)).real
        for x in range(1, 4):
            rho[x] += 2.0 * np.einsum('gi,gi->g', ao4[0].conj(), ao4[x].dot(dm.T)).real
    rho /= nk
    exc, vrho, w = b88_exchange(rho[0], rho[1:])
    wv = weight * np.vstack([0.5 * vrho[None], w])
    vxc = []
    for k in range(nk):
        ao4 = ao4_kpts[k]
        aow = sum(wv[c][:, None] * ao4[c] for c in range(4))
        v = ao4[0].conj().T.dot(aow)
        vxc.append(v + v.conj().T)
    return rho[0].sum() * weight, (rho[0] * exc).sum() * weight, np.array(vxc)


def _task_ao4_kpts(task, atm, a, kpts):
    coords = uniform_grids(np.asarray(a, dtype=float), task['mesh'])
    return [np.asarray(oao.eval_ao_deriv1(atm, task['bas'], task['env'], coords, task['Ls'], task['rcut'], kpt=np.asarray(k, dtype=float)),
                       dtype=np.complex128) for k in np.reshape(kpts, (-1, 3))]


def nr_rks_b88_kpts(tasks, atm, dms, a, fft_mesh, kpts, with_j=False):
    """(nelec, exc, veff (nk, nao, nao)) of 'b88,' through the ladder at k-points (Hermitian dms (nk, nao, nao)): level densities
    and gradients in real space from the Bloch functions and their derivatives, the GGA potential integrated per level and k."""
    a = np.asarray(a, dtype=float)
    fft_mesh = np.asarray(fft_mesh)
    ngrids = int(np.prod(fft_mesh))
    vol = abs(np.linalg.det(a))
    weight = vol / ngrids
    dms = np.asarray(dms, dtype=np.complex128)
    nk, nao = dms.shape[0], dms.shape[-1]
    rhoG = np.zeros((4,) + tuple(int(x) for x in fft_mesh), dtype=np.complex128)
    for t in tasks:
        aos = _task_ao4_kpts(t, atm, a, kpts)
        nH, idx_h, idx_l = t['nH'], t['idx_h'], t['idx_l']
        mesh = tuple(int(x) for x in t['mesh'])
        ng = int(np.prod(mesh))
        rho = np.zeros((4, ng), dtype=np.complex128)
        for k in range(nk):
            ao4 = aos[k]
            blocks = [(slice(0, nH), slice(0, nH), dms[k][idx_h[:, None], idx_h])]
            if len(idx_l):
                blocks += [(slice(0, nH), slice(nH, None), dms[k][idx_h[:, None], idx_l]),
                           (slice(nH, None), slice(0, nH), dms[k][idx_l[:, None], idx_h])]
            for si, sj, d in blocks:
                c = ao4[0][:, si].dot(d)                                                     # sum_i ao_i d_ij
                rho[0] += np.einsum('gj,gj->g', c, ao4[0][:, sj].conj())
                for x in range(1, 4):
                    rho[x] += np.einsum('gj,gj->g', ao4[x][:, si].dot(d), ao4[0][:, sj].conj()) + \
                        np.einsum('gj,gj->g', c, ao4[x][:, sj].conj())
        rho_freq = tools.fft(rho, mesh) * (vol / ng / nk)
        gx, gy, gz = _freq_index(mesh, fft_mesh)
        rhoG[:, gx[:, None, None], gy[:, None], gz] += rho_freq.reshape((4,) + mesh)
    coulG = tools.get_coulG(a, fft_mesh).reshape(rhoG.shape[1:])
    rhoR = tools.ifft(rhoG.reshape(4, ngrids), fft_mesh).real / weight
    exc, vrho, w = b88_exchange(rhoR[0], rhoR[1:])
    wvG = tools.fft(weight * np.vstack([vrho[None], w]), fft_mesh).reshape(rhoG.shape)
    if with_j:
        wvG[0] += rhoG[0] * coulG
    veff = np.zeros((nk, nao, nao), dtype=np.complex128)
    for t in tasks:
        aos = _task_ao4_kpts(t, atm, a, kpts)
        nH, idx_h, idx_l = t['nH'], t['idx_h'], t['idx_l']
        mesh = tuple(int(x) for x in t['mesh'])
        gx, gy, gz = _freq_index(mesh, fft_mesh)
        v = tools.ifft(wvG[:, gx[:, None, None], gy[:, None], gz].reshape(4, -1), mesh).real
        for k in range(nk):
            ao4 = aos[k]
            vp = ao4[0][:, :nH].conj().T.dot(v[0][:, None] * ao4[0])
            for x in range(1, 4):
                vp += ao4[0][:, :nH].conj().T.dot(v[x][:, None] * ao4[x]) + ao4[x][:, :nH].conj().T.dot(v[x][:, None] * ao4[0])
            veff[k][idx_h[:, None], idx_h] += vp[:, :nH]
            if len(idx_l):
                veff[k][idx_h[:, None], idx_l] += vp[:, nH:]
                veff[k][idx_l[:, None], idx_h] += vp[:, nH:].conj().T
    return rhoR[0].sum() * weight, (rhoR[0] * exc).sum() * weight, veff


def b88_spin_channel(rho_s, grad_s):
    """One spin channel of Becke's exchange in its own variables: f(rho_s, grad rho_s) = rho_s^(4/3) G(x_s), and its derivatives
    df/drho_s and df/d(grad rho_s) (no spin-scaling shortcut); rho_s <= 5e-15 -> 0."""
    beta = 0.0042
    cx = 1.5 * (3.0 / (4.0 * np.pi)) ** (1.0 / 3.0)
    rho_s = np.asarray(rho_s, dtype=float)
    m = rho_s > 5e-15
    rs = np.where(m, rho_s, 1.0)
    r13 = np.cbrt(rs)
    r43 = rs * r13
    x = np.sqrt((np.asarray(grad_s) ** 2).sum(axis=0)) / r43
    a = np.arcsinh(x)
    D = 1.0 + 6.0 * beta * x * a
    Dp = 6.0 * beta * (a + x / np.sqrt(1.0 + x * x))
    G = -cx - beta * x * x / D
    Gp_x = -beta * (2.0 * D - x * Dp) / (D * D)
    f = np.where(m, r43 * G, 0.0)
    vrho = np.where(m, (4.0 / 3.0) * r13 * (G - x * x * Gp_x), 0.0)
    w = np.where(m, Gp_x / r43, 0.0)[None] * np.asarray(grad_s)
    return f, vrho, w


def nr_uks_b88(tasks, atm, dms, a, fft_mesh, with_j=False):
    """(nelec, exc, veff (2, nao, nao), ecoul) of an (alpha, beta) pair at the Gamma point, 'b88,' through the ladder
    (multigrid.py:1152-1257, GGA branch; spin channels evaluated in their own variables)."""
    a = np.asarray(a, dtype=float)
    fft_mesh = np.asarray(fft_mesh)
    ngrids = int(np.prod(fft_mesh))
    vol = abs(np.linalg.det(a))
    weight = vol / ngrids
    nao = np.asarray(dms).shape[-1]
    rhoG = [eval_rhoG_gga(tasks, atm, dms[s], a, fft_mesh) for s in range(2)]
    coulG = tools.get_coulG(a, fft_mesh).reshape(rhoG[0].shape[1:])
    tot = rhoG[0][0] + rhoG[1][0]
    vG = tot * coulG
    ecoul = (.5 * (tot.real * vG.real).sum() + .5 * (tot.imag * vG.imag).sum()) / vol
    nelec = exc = 0.0
    veff = []
    for s in range(2):
        rhoR = tools.ifft(rhoG[s].reshape(4, ngrids), fft_mesh).real / weight
        f, vrho, w = b88_spin_channel(rhoR[0], rhoR[1:])
        nelec += rhoR[0].sum() * weight
        exc += f.sum() * weight
        wvG = tools.fft(weight * np.vstack([vrho[None], w]), fft_mesh).reshape(rhoG[s].shape)
        if with_j:
            wvG[0] += vG
        veff.append(integrate_gga(tasks, atm, wvG, a, fft_mesh, nao))
    return nelec, exc, np.array(veff), ecoul
