"""Debug aid: which stage of the block-Jacobi route loses accuracy in k-mode on the GPU?  Swap single stages
for CPU (scipy) versions and look at the route disagreement in K."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch, scipy.linalg
import test_gpu_kpts as T
from pyscf_isdf_amd.isdf import ISDF

cell, coords, Ls, rcut, kpts, aos, dms = T._setup()
c_isdf = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def cpu_block_solve(be, D, ip_off, side, trans, X):
    d, x = be.to_host(D), be.to_host(X)
    for b in range(len(ip_off) - 1):
        s = slice(ip_off[b], ip_off[b + 1])
        Db = np.tril(d[s, s])
        if side == 0:
            x[s] = scipy.linalg.solve_triangular(Db, x[s], lower=True, trans='T' if trans else 'N')
        else:
            x[:, s] = scipy.linalg.solve_triangular(Db, x[:, s].T, lower=True, trans='N' if trans else 'T').T
    X.copy_(be.to_device(x))


class Swap(ISDF):
    swap = ()

    def _bj_rows(self, aoP, nh, ao, ng, Dblk, ip_off, out):
        be = self.backend
        be.pair_gram_rows(aoP, ao, ng, out, nh)
        if 'rows' in self.swap:
            cpu_block_solve(be, Dblk, ip_off, 0, 0, out)
        else:
            be.block_solve(Dblk, ip_off, 0, 0, out)

    def _bj_finish(self, Afac, Dblk, ip_off, W):
        be = self.backend
        if 'finish' in self.swap:
            U = np.tril(be.to_host(Afac))           # row-major lower L, A' = L L^T
            w = be.to_host(W)
            for _ in range(2):
                z = scipy.linalg.cho_solve((U, True), w)
                w = scipy.linalg.cho_solve((U, True), z.T).T
                break
            W.copy_(be.to_device(w))
            cpu_block_solve(be, Dblk, ip_off, 0, 1, W)
            cpu_block_solve(be, Dblk, ip_off, 1, 0, W)
        else:
            ISDF._bj_finish(self, Afac, Dblk, ip_off, W)


ref = ISDF(cell, kpts=kpts, c_isdf=c_isdf, select='local'); ref.fit_route = 'cholesky'
k0 = ref.get_jk(dms, kpts=kpts, with_j=False)[1]
for swap in ((), ('rows',), ('finish',), ('rows', 'finish')):
    df = Swap(cell, kpts=kpts, c_isdf=c_isdf, select='local'); df.swap = swap
    k1 = df.get_jk(dms, kpts=kpts, with_j=False)[1]
    print(swap, 'max|dK| %.3e' % abs(k1 - k0).max(), 'reg', df.reg_used, flush=True)
