"""Plane passes of the convolution: one workgroup per plane (conv_pipe = 0) against persistent workgroups with the next plane
prefetched under the current plane's stages (conv_pipe = 1).  Per-kernel times from rocprofv3 --kernel-trace --stats of this."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pyscf_isdf_amd.backend import HipBackend
be = HipBackend(0)
a = np.eye(3) * 26.96
meshes = [int(x) for x in os.environ.get('MESHES', '120,96,108,64').split(',')]
for n in meshes:
    mesh, nrow = (n, n, n), 512
    G = n ** 3
    rows = torch.randn(nrow, G, dtype=torch.float64, device=be.device)
    out = be.empty((nrow, G))
    ref = None
    for pipe in (0, 1, 0, 1):
        be.set_option('conv_pipe', pipe)
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
        print('mesh %d^3 rows %d conv_pipe %d: %.2f ms  %.2f TB/s algorithmic  (max diff vs first %.1e)' % (n, nrow, pipe, ms, 32.0 * G * nrow / ms / 1e9, err), flush=True)
    del rows, out, ref
    torch.cuda.empty_cache()

# k-point form: real rows in, complex rows out, full real kernel table (the W^q builds of the k-point path)
for n in [int(x) for x in os.environ.get('QMESHES', '96,64').split(',') if x]:
    mesh, nrow = (n, n, n), 256
    G = n ** 3
    rows = torch.randn(nrow, G, dtype=torch.float64, device=be.device)
    tab = torch.rand(G, dtype=torch.float64, device=be.device)
    ore, oim = be.empty((nrow, G)), be.empty((nrow, G))
    ref = None
    for pipe in (0, 1, 0, 1):
        be.set_option('conv_pipe', pipe)
        be.coulomb_rows_q(rows, np.asarray(mesh), tab, ore, oim); be.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            be.coulomb_rows_q(rows, np.asarray(mesh), tab, ore, oim)
        e1.record(); be.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if ref is None:
            ref = (ore.clone(), oim.clone())
        err = max((ore - ref[0]).abs().max().item(), (oim - ref[1]).abs().max().item())
        print('k-point form, mesh %d^3 rows %d conv_pipe %d: %.2f ms  %.2f TB/s (64 G bytes per row)  (max diff vs first %.1e)' % (n, nrow, pipe, ms, 64.0 * G * nrow / ms / 1e9, err), flush=True)
    del rows, ore, oim, ref, tab
    torch.cuda.empty_cache()
