"""Spectral form of W against the classic one on the same build: whole box (must agree to rounding), sphere (must agree with the
classic product made with the sphere-truncated kernel table, option coul_sphere)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF
name = sys.argv[1] if len(sys.argv) > 1 else 'diamond-222-dzvp-80'
cc = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cell = workloads.make_cell(name)
dm = workloads.make_dm(cell)[0]
ref = None
for tag, spectral, sphere, ksphere in (('classic', False, 0, 0), ('classic, kernel table cut to the sphere', False, 0, 100),
                                       ('spectral, whole box', True, 0, 0), ('spectral, sphere', True, 100.0, 0), ('spectral, auto', True, 'auto', 0)):
    df = ISDF(cell, c_isdf=cc, select='refined')
    df.w_spectral, df.w_sphere = spectral, sphere
    df.backend.set_option('coul_sphere', ksphere)
    t0 = time.perf_counter()
    vk = df.get_jk(dm, with_j=False)[1]
    df.backend.synchronize()
    dt = time.perf_counter() - t0
    ek = np.einsum('ij,ji', vk, dm) / 4
    if ref is None:
        ref = (vk.copy(), ek)
    print('%-42s E_K %.12f (%+.3e)  max|dK| vs classic %.2e  fraction %s  share %s  probe %.2e  panels %d  %.1f s' % (
        tag, ek, ek - ref[1], abs(vk - ref[0]).max(), df.w_spectral_fraction, getattr(df, '_sphere_share', (None, None))[1], df.bj_check or 0,
        df.n_panels, dt), flush=True)
    df.backend.set_option('coul_sphere', 0)
    df.reset() if hasattr(df, 'reset') else None
    del df
