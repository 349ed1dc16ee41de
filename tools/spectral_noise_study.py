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

# where does the spectral form's error come from?  (a) the product of the GPU's own X in extended precision on the host against
# the GPU's gemm of the same X: the accumulation;  (b) that host product against the extended-precision sum over numpy's
# spectra: the transform + packing
x = be.to_host(X).astype(np.longdouble)
Mx = x.dot(x.T)
print('GPU gemm of X            vs extended-precision product of the same X: %.2e   (accumulation)' % rel(ms, Mx.astype(float)))
print('extended product of GPU X vs extended-precision Fourier sum (numpy):   %.2e   (transform + packing)' % rel(Mx.astype(float), Mref.astype(float)))
# the same split for the classic form: V from the GPU convolution, product on the host in extended precision
V = sub.clone(); be.coulomb_rows(V, mesh, a, n)
vh = be.to_host(V).astype(np.longdouble)
Mv = w * vh.dot(y.astype(np.longdouble).T)
print('GPU classic product      vs extended product of the GPU V and rows:   %.2e   (accumulation)' % rel(mc, Mv.astype(float)))
print('extended product of GPU V vs extended-precision Fourier sum:           %.2e   (convolution)' % rel(Mv.astype(float), Mref.astype(float)))

# does the ORDER of the packed points matter for the accumulation?  512-row strips (the production kernel variant); points as they
# lie in the half spectrum / sorted by descending |G| (small terms first, the large low-G terms at the end of the chain) / ascending
import numpy as _np
sub512 = Y[:512].contiguous()
pick = _np.array([0, 1, 5, 40, 41, 100, 200, 300, 400, 511])
b = 2 * _np.pi * _np.linalg.inv(a).T
n0, n1, n2 = (int(v) for v in mesh)
f0, f1, f2 = _np.fft.fftfreq(n0, 1.0 / n0), _np.fft.fftfreq(n1, 1.0 / n1), _np.arange(n2 // 2 + 1, dtype=float)
Gv = f0[:, None, None, None] * b[0] + f1[None, :, None, None] * b[1] + f2[None, None, :, None] * b[2]
g2 = _np.einsum('xyzc,xyzc->xyz', Gv, Gv).ravel()
idx0, sc0 = be.to_host(plan2['idx']), be.to_host(plan2['scale'])
for tag, order in (('as stored', _np.arange(len(idx0))), ('descending |G|', _np.argsort(-g2[idx0], kind='stable')), ('ascending |G|', _np.argsort(g2[idx0], kind='stable'))):
    Xo = be.empty((512, plan2['ldx']))
    be.spectral_rows(sub512, mesh, be.to_device(idx0[order].astype(_np.int32)), be.to_device(sc0[order]), Xo, batch=512)
    Mo = be.empty((512, 512)); be.gemm_nt(Xo, Xo, Mo)
    xo = be.to_host(Xo)[pick].astype(_np.longdouble)
    ref = xo.dot(xo.T).astype(float)
    got = be.to_host(Mo)[_np.ix_(pick, pick)]
    scl = _np.sqrt(_np.outer(_np.diag(ref), _np.diag(ref)))
    print('512-row strip, sphere, points %-16s: gemm vs extended product of the same X  %.2e' % (tag, float(_np.max(_np.abs((got - ref) / scl)))))
Vc = sub512.clone(); be.coulomb_rows(Vc, mesh, a, 512)
Mc5 = be.empty((512, 512)); be.gemm_nt(Vc, sub512, Mc5, alpha=w)
vh = be.to_host(Vc)[pick].astype(_np.longdouble); yh = be.to_host(sub512)[pick].astype(_np.longdouble)
ref = (w * vh.dot(yh.T)).astype(float)
scl = _np.sqrt(_np.outer(_np.diag(ref), _np.diag(ref)))
print('512-row strip, classic product: gemm vs extended product of the same V, rows           %.2e' % float(_np.max(_np.abs((be.to_host(Mc5)[_np.ix_(pick, pick)] - ref) / scl))))
