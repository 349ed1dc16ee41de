"""Does dropping the Fourier components of the Coulomb kernel outside the sphere inscribed in the FFT box change the ISDF K?
(option coul_sphere = percent of the inscribed radius; 0 = the whole box, the reference's sum)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF
name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-444-dzvp-120'
cc = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pcts = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else '0,100,90,80').split(',')]
cell = workloads.make_cell(name)
dm = workloads.make_dm(cell)[0]
df = ISDF(cell, c_isdf=cc, select='refined')
ref = None
for pct in pcts:
    df.backend.set_option('coul_sphere', pct)
    df._built = False
    t0 = time.perf_counter()
    vj, vk = df.get_jk(dm)
    df.backend.synchronize()
    ek, ej = np.einsum('ij,ji', vk, dm) / 4, np.einsum('ij,ji', vj, dm) / 2
    if ref is None:
        ref = (vk.copy(), ek, vj.copy(), ej)
    print('coul_sphere %3d%%: E_K %.12f (%+.3e vs whole box)  max|dK| %.2e   E_J %.12f (%+.3e)  max|dJ| %.2e  %.1f s' % (
        pct, ek, ek - ref[1], abs(vk - ref[0]).max(), ej, ej - ref[3], abs(vj - ref[2]).max(), time.perf_counter() - t0), flush=True)
