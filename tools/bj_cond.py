"""Developer experiment: conditioning of the block-preconditioned Gram matrix A' = D^-1 A D^-T against the
disagreement of the block-Jacobi and Cholesky fit routes (which one amplifies rounding by cond(A'))."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF

name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-222-dzvp-80'
cs = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [10]
group = int(os.environ.get('BJ_GROUP', '1'))
cell = workloads.make_cell(name)
dm, c, occ = workloads.make_dm(cell)
for c_isdf in cs:
    vk = {}
    for route in ('cholesky', 'blockjacobi'):
        df = ISDF(cell, c_isdf=c_isdf, select='local')
        df.fit_route = route if route == 'cholesky' else 'auto'
        df.bj_check_tol = 1e99
        df.bj_max_c = 10**9
        df.bj_group = group
        if 'BJ_RADIUS' in os.environ:
            df.bj_cluster_radius = float(os.environ['BJ_RADIUS'])
        t0 = time.perf_counter()
        vk[route] = df.get_jk(dm, with_j=False)[1]
        t1 = time.perf_counter()
        print(name, c_isdf, route, 'P', len(df.ip), 'reg', df.reg_used, '%.2f s' % (t1 - t0), 'probe check', df.bj_check,
              {k: round(v, 3) for k, v in df.timings.items()}, flush=True)
        if route == 'blockjacobi' and '--eig' in sys.argv:
            be = df.backend
            P = len(df.ip)
            A = be.empty((P, P)); be.gram_sq(df.aoP, A)
            wA = torch.linalg.eigvalsh(A)
            D = df._buffer('Dblk', (P, P))
            # per-atom block offsets: points are stored atom by atom
            from pyscf_isdf_amd.isdf import partition_grid_by_atom
            owner = partition_grid_by_atom(df.grids.coords[df.ip], cell.atom_coords(), cell.lattice_vectors())
            ip_off = df._bj_blocks(np.bincount(owner, minlength=cell.natm), df._bj_clusters())
            be.block_solve(D, ip_off, 0, 0, A); be.block_solve(D, ip_off, 1, 1, A)
            A = (A + A.T) / 2
            w = torch.linalg.eigvalsh(A)
            print('   cond(A) %.3e   cond(A\') %.3e  lam_min %.3e lam_max %.3e' %
                  (float(wA[-1] / abs(wA[0])), float(w[-1] / abs(w[0])), float(w[0]), float(w[-1])), flush=True)
        del df
    d = vk['cholesky'] - vk['blockjacobi']
    print('   max|dK| %.3e  |K|max %.3e  dEK %.3e' % (abs(d).max(), abs(vk['cholesky']).max(), abs(np.einsum('ij,ji', d, dm)) / 4), flush=True)
