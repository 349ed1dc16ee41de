"""Debug aid: run the k-point block-Jacobi build on the HIP backend with chosen stages delegated to the checker
backend (tensors copied to the host and back), to find the stage that loses accuracy."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
import test_gpu_kpts as T
from oracle_backend import OracleBackend
from pyscf_isdf_amd.backend import HipBackend
from pyscf_isdf_amd.isdf import ISDF


class Mix:
    def __init__(self, cpu_stages):
        self.hip, self.cpu, self.cpu_stages = HipBackend(0), OracleBackend(), set(cpu_stages)
        self.device = self.hip.device

    def __getattr__(self, name):
        f = getattr(self.hip, name)
        if name not in self.cpu_stages:
            return f
        g = getattr(self.cpu, name)

        def call(*args, **kw):
            host = [a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args]
            hkw = {k: (v.detach().cpu().clone() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
            out = g(*host, **hkw)
            for a, h in zip(args, host):
                if isinstance(a, torch.Tensor):
                    a.copy_(h)
            for k, v in kw.items():
                if isinstance(v, torch.Tensor):
                    v.copy_(hkw[k])
            return out
        return call


cell, coords, Ls, rcut, kpts, aos, dms = T._setup()
c = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ref = ISDF(cell, kpts=kpts, c_isdf=c, select='local'); ref.fit_route = 'cholesky'
k0 = ref.get_jk(dms, kpts=kpts, with_j=False)[1]
groups = {
    'none': [],
    'prepare': ['gather_aoP', 'gram_sq', 'shift_diag', 'block_chol', 'chol_inplace'],
    'solves': ['block_solve', 'W_from_factor'],
    'rows': ['pair_gram_rows'],
    'conv': ['coulomb_Wq', 'symmetrize_hermitian'],
    'finishq': ['finish_Wq'],
    'kpair': ['get_k_pair'],
    'all-fit': ['gather_aoP', 'gram_sq', 'shift_diag', 'block_chol', 'chol_inplace', 'block_solve', 'W_from_factor', 'pair_gram_rows',
                'coulomb_Wq', 'symmetrize_hermitian', 'finish_Wq'],
}
for name, st in groups.items():
    df = ISDF(cell, kpts=kpts, c_isdf=c, select='local', backend=Mix(st)); df.fit_route = 'blockjacobi'
    k1 = df.get_jk(dms, kpts=kpts, with_j=False)[1]
    print('%-8s on cpu: max|dK| vs cholesky %.3e' % (name, abs(k1 - k0).max()), flush=True)
