"""Coulomb convolution of a batch of rows: hipFFT (own_fft 0), the five-pass own FFT (1), the three-pass plane form (2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_isdf_amd.backend import HipBackend
be = HipBackend(0)
a = np.eye(3) * 26.96
for mesh, nrow in (((120, 120, 120), 512), ((108, 108, 108), 512), ((128, 128, 128), 256), ((96, 96, 96), 512)):
    G = int(np.prod(mesh))
    rows = torch.randn(nrow, G, dtype=torch.float64, device=be.device)
    out = be.empty((nrow, G))
    ref = None
    for own in (0, 1, 2):
        be.set_option('own_fft', own)
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
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        print('mesh %s rows %d own_fft %d: %.2f ms  %.2f TB/s algorithmic  (max rel diff vs hipFFT %.1e)' % (mesh, nrow, own, ms, 32.0 * G * nrow / ms / 1e9, err), flush=True)
    del rows, out, ref
    torch.cuda.empty_cache()
be.set_option('own_fft', 2)
