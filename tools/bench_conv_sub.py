"""The plane convolution with cache-resident sub-batches (option conv_sub_rows): does keeping a sub-batch's half spectra in the
Infinity Cache between the three passes pay?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_isdf_amd.backend import HipBackend
be = HipBackend(0)
a = np.eye(3) * 26.96
for mesh, nrow in (((120, 120, 120), 512), ((96, 96, 96), 512), ((160, 160, 160), 256)):
    G = int(np.prod(mesh))
    rows = torch.randn(nrow, G, dtype=torch.float64, device=be.device)
    out = be.empty((nrow, G))
    ref = None
    for sub in (0, 4, 8, 12, 16, 24, 32, 64):
        be.set_option('conv_sub_rows', sub)
        be.coulomb_rows(rows, np.asarray(mesh), a, nrow, out=out); be.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            be.coulomb_rows(rows, np.asarray(mesh), a, nrow, out=out)
        e1.record(); be.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if ref is None:
            ref = out.clone()
        err = (out - ref).abs().max().item()
        print('mesh %s rows %d conv_sub_rows %3d: %.2f ms  %.2f TB/s algorithmic  (max diff vs whole batch %.1e)' % (mesh, nrow, sub, ms, 32.0 * G * nrow / ms / 1e9, err), flush=True)
    del rows, out, ref
    torch.cuda.empty_cache()
