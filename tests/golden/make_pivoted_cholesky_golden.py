"""Generate golden pivot lists from the REFERENCE's pivoted Cholesky
(/root/reference/pyscf/lib/scipy_helper.py:71-110, loaded standalone by path — it needs only numpy).

Run in the build container only (the reference does not exist on the GPU box):
    python tests/golden/make_pivoted_cholesky_golden.py
Writes tests/golden/pivoted_cholesky_golden.json: for each case the seed/shape that regenerates the
input AO-like matrix and the pivots + rank + diagonal of the factor the reference returns on the
explicitly formed pair-density Gram matrix A = (ao^T ao)^2.
"""
import importlib.util
import json
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location('ref_scipy_helper', '/root/reference/pyscf/lib/scipy_helper.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def make_ao(seed, nao, m):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((nao, m)) * np.exp(-3.0 * rng.random(m))


cases = []
for seed, nao, m in [(1, 3, 40), (2, 5, 120), (3, 8, 300), (4, 2, 64), (5, 6, 257)]:
    ao = make_ao(seed, nao, m)
    A = ao.T.dot(ao) ** 2
    L, piv, rank = ref.pivoted_cholesky_python(A, tol=-1.0, lower=True)
    cases.append(dict(seed=seed, nao=nao, m=m, rank=int(rank), piv=[int(x) for x in piv[:rank]],
                      diag=[float(x) for x in np.diag(L)[:rank]]))
    print(seed, nao, m, 'rank', rank, '(npair = %d)' % (nao * (nao + 1) // 2))
with open(os.path.join(HERE, 'pivoted_cholesky_golden.json'), 'w') as f:
    json.dump(dict(source='pyscf/lib/scipy_helper.py:71-110 pivoted_cholesky_python(A, tol=-1, lower=True), A=(ao^T ao)^2',
                   make_ao='rng=default_rng(seed); ao=rng.standard_normal((nao,m))*exp(-3*rng.random(m))',
                   cases=cases), f, indent=1)
