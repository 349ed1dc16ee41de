"""Developer demonstration: a Gamma-point RHF on diamond 2x2x2 / gth-dzvp / 80^3 whose Fock matrix comes entirely from the
ISDF object (get_pp for the GTH pseudopotential, get_jk for J and K), once with the ISDF exchange and once with the
reference's exact exchange evaluated on the same GPU (get_k_exact) - the SCF-level error of the ISDF approximation with
physical orbitals.  Kinetic energy and overlap from the AO values on the FFT grid (plane-wave quadrature, host numpy).
The nuclear repulsion is left out: it is the same constant in both runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF

name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-222-dzvp-80'
cs = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [10]
modes = [False, True] if '--both' in sys.argv else [('--robust' in sys.argv)]
cell = workloads.make_cell(name)
nao, nocc = cell.nao_nr(), cell.nelectron // 2
mesh = np.asarray(cell.mesh)
G = int(np.prod(mesh))


def kinetic_and_overlap(df):
    ao = df.backend.to_host(df.ao).reshape(nao, *mesh)
    F = np.fft.fftn(ao, axes=(1, 2, 3)).reshape(nao, G)
    b = 2 * np.pi * np.linalg.inv(cell.lattice_vectors().T)
    fr = [np.fft.fftfreq(n, 1. / n) for n in mesh]
    Gv = (fr[0][:, None, None, None] * b[0] + fr[1][None, :, None, None] * b[1] + fr[2][None, None, :, None] * b[2]).reshape(-1, 3)
    g2 = np.einsum('gi,gi->g', Gv, Gv)
    T = 0.5 * cell.vol / G ** 2 * (F.conj() * g2).dot(F.T).real
    S = cell.vol / G ** 2 * F.conj().dot(F.T).real
    return T, S


def rhf(hcore, S, get_jk, tag, max_cycle=40, conv=1e-9):
    e, c = scipy.linalg.eigh(hcore, S)
    dm = 2 * c[:, :nocc].dot(c[:, :nocc].T)
    errs, focks = [], []
    e_last = 0.0
    for it in range(max_cycle):
        t0 = time.perf_counter()
        vj, vk = get_jk(dm, c[:, :nocc])
        f = hcore + vj - 0.5 * vk
        e_el = 0.5 * np.einsum('ij,ji', hcore + f, dm)
        # Pulay DIIS on F D S - S D F
        err = f.dot(dm).dot(S) - S.dot(dm).dot(f)
        focks.append(f); errs.append(err)
        focks, errs = focks[-8:], errs[-8:]
        n = len(focks)
        if n > 1:
            B = -np.ones((n + 1, n + 1)); B[n, n] = 0
            for i in range(n):
                for j in range(n):
                    B[i, j] = np.vdot(errs[i], errs[j])
            rhs = np.zeros(n + 1); rhs[n] = -1
            coef = np.linalg.lstsq(B, rhs, rcond=None)[0][:n]
            f = sum(ci * fi for ci, fi in zip(coef, focks))
        print('  %s it %2d  E_el %.10f  dE %.2e  |err| %.1e  %.2f s' % (tag, it, e_el, e_el - e_last, abs(err).max(), time.perf_counter() - t0), flush=True)
        if abs(e_el - e_last) < conv and abs(err).max() < 1e-6:
            break
        e_last = e_el
        e, c = scipy.linalg.eigh(f, S)
        dm = 2 * c[:, :nocc].dot(c[:, :nocc].T)
    return e_el, dm, e


ref = None
for c_isdf in cs:
    for robust in modes:
        df = ISDF(cell, c_isdf=c_isdf)
        df.robust_k = robust
        t0 = time.perf_counter()
        df.build()
        print('c_isdf %d%s: build %.2f s, P = %d, route %s' % (c_isdf, ' robust' if robust else '', time.perf_counter() - t0, len(df.ip),
                                                             df.fit_route_used), flush=True)
        if ref is None:
            T, S = kinetic_and_overlap(df)
            hcore = T + df.get_pp()

            def jk_exact(dm, cocc, df=df):
                vj = df.get_jk(dm, with_k=False)[0]
                return vj, df.get_k_exact(mo_coeff=cocc, mo_occ=np.full(cocc.shape[1], 2.0))
            ref = rhf(hcore, S, jk_exact, 'exact K')
            print('exact-K RHF: E_el = %.10f Eh, gap %.4f Eh' % (ref[0], ref[2][nocc] - ref[2][nocc - 1]), flush=True)
        tag = 'ISDF c=%d%s' % (c_isdf, ' robust' if robust else '')
        e_isdf, dm, eps = rhf(hcore, S, lambda dm, cocc: df.get_jk(dm), tag)
        print('%s RHF: E_el = %.10f Eh   E(ISDF) - E(exact K) = %.3e Eh = %.3e Eh/atom   gap %.4f (exact %.4f)' %
              (tag, e_isdf, e_isdf - ref[0], (e_isdf - ref[0]) / cell.natm, eps[nocc] - eps[nocc - 1], ref[2][nocc] - ref[2][nocc - 1]), flush=True)
        df.reset()
