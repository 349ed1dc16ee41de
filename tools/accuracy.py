"""ISDF fitting error at full size: ISDF K vs the reference's exact exchange (both on the GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF

name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-222-dzvp-80'
cs = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [10]
selects = sys.argv[3].split(',') if len(sys.argv) > 3 else ['local']
cell = workloads.make_cell(name)
dm, c, occ = workloads.make_dm(cell)
nocc = int((occ > 0).sum())


class Tagged(np.ndarray):
    pass


tdm = dm.view(Tagged)
tdm.mo_coeff, tdm.mo_occ = c, occ
print(name, 'nao', cell.nao_nr(), 'nocc', nocc, 'ngrids', int(np.prod(cell.mesh)), flush=True)
ref = None
for sel in selects:
    for cc in cs:
        over = None
        cut = None
        space = 'ao'
        sel_full = sel
        sel_ = sel
        if '@' in sel:                      # 'refined@occ' = the (AO x occupied) pair space of the density (pair_space='occ')
            sel_, space = sel.split('@')
        if ':' in sel_:                      # 'refined:3' = refined selection with refine_over = 3; 'refined:2:7': + cand_ao_cutoff 7 Bohr
            parts = sel_.split(':')
            sel_, over = parts[0], float(parts[1])
            cut = float(parts[2]) if len(parts) > 2 else None
        df = ISDF(cell, c_isdf=cc, select=sel_)
        df.pair_space = space
        if over is not None:
            df.refine_over = over
        if cut is not None:
            df.cand_ao_cutoff = cut
        if os.environ.get('ISDF_CAND_CUTOFF'):
            df.cand_ao_cutoff = float(os.environ['ISDF_CAND_CUTOFF'])
        if os.environ.get('ISDF_FIT_ROUTE'):
            df.fit_route = os.environ['ISDF_FIT_ROUTE']
        if os.environ.get('ISDF_BJ_MAX_C'):
            df.bj_max_c = int(os.environ['ISDF_BJ_MAX_C'])
        if os.environ.get('ISDF_W_MAX_C'):
            df.w_spectral_max_c = int(os.environ['ISDF_W_MAX_C'])
        if os.environ.get('ISDF_W_CHECK_TOL'):
            df.w_spectral_check_tol = float(os.environ['ISDF_W_CHECK_TOL'])
        if os.environ.get('ISDF_BJ_GROUP'):
            df.bj_group = int(os.environ['ISDF_BJ_GROUP'])
        if os.environ.get('ISDF_W_FORM'):
            df.w_spectral = os.environ['ISDF_W_FORM'] == 'spectral'
        t0 = time.perf_counter()
        vk = df.get_jk(tdm if space == 'occ' else dm, with_j=False)[1]
        t1 = time.perf_counter()
        nip, route_used = len(df.ip), '%s, %d panel(s), probe %s, W %s' % (df.fit_route_used, df.n_panels, ('%.1e' % df.bj_check) if df.bj_check is not None else '-', ('spectral %.3f' % df.w_spectral_fraction) if df.w_spectral_fraction else 'classic')
        if ref is None:
            df.release_fit_buffers()            # the fit's buffers fill HBM at large c; the exact exchange needs phi only
            ref = df.get_k_exact(mo_coeff=c, mo_occ=occ)
            df.backend.synchronize()
            print('exact K (N*nocc = %d FFT pairs) on the GPU: %.1f s' % (cell.nao_nr() * nocc, time.perf_counter() - t1), flush=True)
        ek, ek0 = np.einsum('ij,ji', vk, dm) / 4, np.einsum('ij,ji', ref, dm) / 4
        print('select=%-14s c=%2d P=%6d  build+K %.2f s   E_K(ISDF) %.8f  E_K(exact) %.8f  dE_K %.2e Eh (%.1e rel)  max|dK| %.2e  route %s  stages %s'
              % (sel_full, cc, nip, t1 - t0, ek, ek0, ek - ek0, abs(ek - ek0) / ek0, abs(vk - ref).max(), route_used, {k: round(v, 2) for k, v in df.timings.items()}), flush=True)
        df.reset()
        del df
