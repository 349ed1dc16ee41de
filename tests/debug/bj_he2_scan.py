"""Debug aid: route disagreement vs conditioning on the He2 test cell, Gamma and k-mode, GPU and checker."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
import test_gpu_kpts as T
from oracle_backend import OracleBackend
from pyscf_isdf_amd.isdf import ISDF

cell, coords, Ls, rcut, kpts, aos, dms = T._setup()
for mode in ('gamma', 'k'):
    for c in (4, 6, 8, 10):
        ks = {}
        for bname in ('hip', 'cpu'):
            for route in ('cholesky', 'blockjacobi'):
                kw = dict(c_isdf=c, select='local')
                if bname == 'cpu':
                    kw['backend'] = OracleBackend()
                if mode == 'k':
                    df = ISDF(cell, kpts=kpts, **kw); df.fit_route = route
                    ks[bname, route] = df.get_jk(dms, kpts=kpts, with_j=False)[1]
                else:
                    df = ISDF(cell, **kw); df.fit_route = route if route == 'cholesky' else 'auto'; df.bj_check_tol = 1e99
                    ks[bname, route] = df.get_jk(dms[0].real.copy(), with_j=False)[1]
                    if route != 'cholesky':
                        print('      ', bname, 'probe check %.2e' % df.bj_check)
        ref = ks['cpu', 'cholesky']
        print(mode, c, 'P', len(df.ip), ' '.join('%s/%s %.2e' % (b, r[:4], abs(ks[b, r] - ref).max()) for b, r in ks), flush=True)
