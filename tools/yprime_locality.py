"""Developer experiment: how local are the rows of Y' = D^-1 (aoP ao)^2 ?  For the grid in atom-major (Voronoi)
order, the largest |Y'| of each (point-block Q, grid cell A) pair, relative to the row-block's largest value."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF, partition_grid_by_atom

name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-222-dzvp-80'
cell = workloads.make_cell(name)
df = ISDF(cell, c_isdf=10, select='local')
df.fit_route = 'blockjacobi'
# stop after the fit: monkeypatch coulomb_W to capture Y'
be = df.backend
cap = {}
orig = be.coulomb_W
def grab(theta, *a, **k):
    cap['Y'] = theta
    raise StopIteration
be.coulomb_W = grab
try:
    df.build()
except StopIteration:
    pass
Y = cap['Y']
P, G = Y.shape
coords = df.grids.coords
owner = partition_grid_by_atom(coords, cell.atom_coords(), cell.lattice_vectors())
natm = cell.natm
own_d = torch.from_numpy(owner).to(Y.device)
blk = P // natm
mx = torch.zeros((natm, natm), dtype=torch.float64, device=Y.device)
l2 = torch.zeros((natm, natm), dtype=torch.float64, device=Y.device)
for q in range(natm):
    rows = Y[q * blk:(q + 1) * blk]
    colmax = rows.abs().amax(dim=0)                       # (G,)
    col2 = (rows * rows).sum(dim=0)
    mx[q].scatter_reduce_(0, own_d, colmax, reduce='amax')
    l2[q].scatter_add_(0, own_d, col2)
mx = mx.cpu().numpy(); l2 = l2.cpu().numpy()
rel = mx / mx.max(axis=1, keepdims=True)
print(name, 'P', P, 'G', G, 'natm', natm)
for thr in (1e-4, 1e-6, 1e-8, 1e-10, 1e-12, 1e-14):
    frac = (rel > thr).mean()
    tail = np.sqrt((l2 * (rel <= thr)).sum() / l2.sum())
    print('  threshold %.0e: %.3f of the (block, cell) pairs kept; dropped L2 mass (relative) %.2e' % (thr, frac, tail))
