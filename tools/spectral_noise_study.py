"""Which form of M' = w conv(Y') Y'^T is closer to the truth?  Rows Y' of a real build (diamond 2x2x2, 80^3, c = 12): a few entries of
M' from (i) the classic product, (ii) X X^T over the whole box, against (iii) numpy in extended precision on the host
(float64 FFT of the same rows, products and sums in longdouble)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF
cell = workloads.make_cell('diamond-222-dzvp-80')
dm = workloads.make_dm(cell)[0]
df = ISDF(cell, c_isdf=12, select='refined')
df.w_spectral = False
df.get_jk(dm, with_j=False)
be = df.backend
st = df._fit_state
Y = st['theta']                                   # (P, G) rows Y'
P, G = Y.shape
mesh = np.asarray(cell.mesh, dtype=np.int32)
a = np.asarray(cell.lattice_vectors())
w = cell.vol / G
rows_i = np.array([0, 1, 5, 40, 41, 300, 777, 1500, 2000, 2495])
sub = Y[be.to_device(rows_i.astype(np.int64))].contiguous()
n = len(rows_i)
Mc = be.empty((n, n)); be.coulomb_W(sub, mesh, a, 0, n, n, Mc)
df.w_spectral, df.w_sphere = True, 0
plan = df._spectral_plan()
X = be.empty((n, plan['ldx'])); be.spectral_rows(sub, mesh, plan['idx'], plan['scale'], X, batch=n)
Ms = be.empty((n, n)); be.gemm_nt(X, X, Ms)
df.w_sphere = 100.0
plan2 = df._spectral_plan()
X2 = be.empty((n, plan2['ldx'])); be.spectral_rows(sub, mesh, plan2['idx'], plan2['scale'], X2, batch=n)
Ms2 = be.empty((n, n)); be.gemm_nt(X2, X2, Ms2)
# extended precision on the host
y = be.to_host(sub)
z = np.fft.rfftn(y.reshape(n, *mesh), axes=(1, 2, 3)).reshape(n, -1)
cg = be.coulG_half(mesh, a).ravel()
mult = np.full((mesh[0], mesh[1], mesh[2] // 2 + 1), 2.0); mult[:, :, 0] = 1.0; mult[:, :, mesh[2] // 2] = 1.0
wt = (mult.ravel() * w * cg).astype(np.longdouble)
zr, zi = z.real.astype(np.longdouble), z.imag.astype(np.longdouble)
Mref = np.array([[np.sum(wt * (zr[i] * zr[j] + zi[i] * zi[j])) for j in range(n)] for i in range(n)], dtype=np.longdouble)
# and the real-space sum in extended precision with the float64 convolution from numpy
v = np.fft.irfftn(z.reshape(n, mesh[0], mesh[1], -1) * (cg * G).reshape(1, mesh[0], mesh[1], -1), s=tuple(mesh), axes=(1, 2, 3)).reshape(n, -1)
Mreal = np.array([[w * np.sum(v[i].astype(np.longdouble) * y[j].astype(np.longdouble)) for j in range(n)] for i in range(n)], dtype=np.longdouble)
mc, ms, ms2 = be.to_host(Mc), be.to_host(Ms), be.to_host(Ms2)
sc = np.sqrt(np.outer(np.diag(mc), np.diag(mc)))
def rel(a, b):
    return float(np.max(np.abs((a - b) / sc)))
print('entries relative to sqrt(M_PP M_QQ); diag of M\' spans %.2e .. %.2e' % (np.diag(mc).min(), np.diag(mc).max()))
print('classic (GPU)        vs extended-precision Fourier sum: %.2e' % rel(mc, Mref.astype(float)))
print('spectral box (GPU)   vs extended-precision Fourier sum: %.2e' % rel(ms, Mref.astype(float)))
print('spectral sphere (GPU) vs extended-precision Fourier sum: %.2e' % rel(ms2, Mref.astype(float)))
print('classic vs spectral box (GPU):                         %.2e' % rel(mc, ms))
print('extended real-space sum (numpy conv) vs Fourier sum:    %.2e' % rel(Mreal.astype(float), Mref.astype(float)))
print('asymmetry classic %.2e  spectral %.2e' % (rel(mc, mc.T), rel(ms, ms.T)))
