#!/usr/bin/env python
"""bench.py — ISDF build + 1x get_jk on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--no-cpu-baseline]

One "step" = ISDF.build() (collocation, interpolation-point selection, fit, Coulomb W) followed by
one get_jk() on a synthetic density matrix; the cell tables and the density matrix are already
resident when the timed region starts.  N > 1 is launched by torch.distributed.run (one rank per
GPU, RCCL).  Rank 0 prints ONE JSON line (see the contract in the task description) with two extra
objects: "roofline" for the dominant hand-written kernel (measured live with HIP events on the work
stream by the library's profiling hooks) and "cpu_baseline" (the numpy/scipy oracle timed on the
host on a bounded sample of the same workload, extrapolated stage by stage).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

T_PROCESS_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD datasheet, MI355X FP64 matrix (the guide's table has no f64 row);
                               # best rocBLAS dgemm measured on the box: 73.8 (profiles/r01_probe_*.log)
PMC_FILE = 'r03_pmc_bench_cfg3_spectral_fetch_write.json'
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default=os.environ.get('ISDF_BENCH_WORKLOAD', 'diamond-444-dzvp-120'))
    ap.add_argument('--select', default='refined', choices=['local', 'refined', 'global'])
    ap.add_argument('--refine-over', type=float, default=2.0)
    ap.add_argument('--cand-ao-cutoff', type=float, default=None,
                    help="Bohr: the candidate stage of the refined selection sees only the AOs of atoms this close to a block's atom (default: all)")
    ap.add_argument('--no-accuracy', action='store_true',
                    help="skip the exact-exchange comparison (config.dE_K_vs_exact; the reference's algorithm on the same GPU, "
                         "about 40 s at configs[2], evaluated BEFORE the warm-up and timed steps)")
    ap.add_argument('--pair-space', default='ao', choices=['ao', 'occ'],
                    help="'occ': fit the (AO x occupied orbital) pairs of the benchmark density (its mo_coeff/mo_occ tag, "
                         "fft_jk.py:206-210); the fit then runs inside get_jk, still inside the timed step")
    ap.add_argument('--c-isdf', type=int, default=None,
                    help='interpolation points per AO; default: 12 for the headline workload (the accuracy scan of DESIGN.md section 2), else 10')
    ap.add_argument('--fit-route', default=None, choices=['auto', 'cholesky', 'blockjacobi'])
    ap.add_argument('--w-form', default='spectral', choices=['spectral', 'classic'],
                    help="spectral (default): W = X X^T from the half spectra inside the sphere; classic: w conv(rows) rows^T")
    ap.add_argument('--robust-k', action='store_true', help='time the build + get_jk with the robust exchange (not the headline)')
    ap.add_argument('--density', default='random', choices=['random', 'scf'],
                    help="'random': BASELINE.json's benchmark density (random orthogonal orbitals; the headline); 'scf': physical "
                         "orbitals - an RHF converged with the ISDF object itself before anything is timed (hcore from the device, S and T "
                         "by plane-wave quadrature of the AO values); Gamma-point workloads, one GPU")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--stage-report', default=None, help='write the per-kernel table to this file')
    return ap.parse_args()


def cpu_baseline(cell, c_isdf, gpu_stage_sizes):
    """Time the numpy/scipy oracle on a bounded sample of the workload and extrapolate each stage
    linearly in its sampled dimension.  Returns (seconds for the full workload, description)."""
    import scipy.linalg
    from pyscf_isdf_amd import gto
    from oracle import ao as oao, isdf as oisdf, pbc_tools as tools
    rng = np.random.default_rng(0)
    t_all = time.perf_counter()
    mesh = np.asarray(cell.mesh)
    G = int(np.prod(mesh))
    nao = cell.nao_nr()
    natm = cell.natm
    P = gpu_stage_sizes['P']
    a = cell.lattice_vectors()
    coords = cell.get_uniform_grids()
    rcut = gto.estimate_rcut_per_shell(cell)
    Ls = gto.get_lattice_Ls(cell, rcut=rcut.max())
    est = {}
    # S1 collocation: a contiguous run of grid points
    n1 = min(G, 16384)
    t = time.perf_counter()
    aoT_s = np.ascontiguousarray(oao.eval_ao(cell._atm, cell._bas, cell._env, coords[:n1], Ls, rcut, rule='point').T)
    est['S1_eval_ao'] = (time.perf_counter() - t) * G / n1
    # S2 selection: one atom block of average size with its share of points, synthetic AO columns of the right shape
    m_b = G // natm
    k_b = max(1, P // natm)
    reps = max(1, m_b // n1 + 1)
    ao_blk = np.tile(aoT_s, (1, reps))[:, :m_b] * (1.0 + 0.01 * rng.standard_normal(m_b))
    t = time.perf_counter()
    oisdf.select_ip(ao_blk, k_b)
    est['S2_select_ip'] = (time.perf_counter() - t) * natm
    # S3 fit: Cholesky of A_PP (a leading block of at most 20480 points, extrapolated with its cube: the sample has to stay
    # bounded at P = 30 000) + solves on a slice of grid columns (extrapolated with the square of the point count)
    aoP = rng.standard_normal((P, nao)) / np.sqrt(nao)
    Pc = min(P, 20480)
    t = time.perf_counter()
    A = aoP[:Pc].dot(aoP[:Pc].T) ** 2
    A[np.diag_indices(Pc)] += 1e-3 * A.diagonal().max()
    cf = scipy.linalg.cho_factor(A, overwrite_a=True)
    t_chol = (time.perf_counter() - t) * (P / Pc) ** 3
    n3 = min(G, 16384)
    t = time.perf_counter()
    B = aoP[:Pc].dot(aoT_s[:, :n3]) ** 2
    scipy.linalg.cho_solve(cf, B)
    est['S3_fit'] = t_chol + (time.perf_counter() - t) * (P / Pc) ** 2 * G / n3
    del A, cf
    # S4 Coulomb convolution of a few full-grid rows; S5 W rows
    n4 = 4
    rows = rng.standard_normal((n4, G))
    t = time.perf_counter()
    oisdf.coulomb_V(rows, a, mesh)
    est['S4_coulomb_fft'] = (time.perf_counter() - t) * P / n4
    del rows
    r5, n5 = min(P, 512), min(G, 16384)                  # a 512-row batch of V against all P rows of Theta: BLAS at full rate
    Vs = rng.standard_normal((r5, n5))
    th = rng.standard_normal((P, n5))
    t = time.perf_counter()
    Vs.dot(th.T)
    est['S5_W_gemm'] = (time.perf_counter() - t) * (P / r5) * (G / n5) * 0.5   # symmetric half, like the GPU path
    del th, Vs
    # S6 J (two N x N x G contractions on a slice) and S7 K (exact size if affordable, else row slice)
    dm = rng.standard_normal((nao, nao))
    t = time.perf_counter()
    tmp = dm.dot(aoT_s)
    np.einsum('ig,ig->g', tmp, aoT_s)
    (aoT_s * tmp[0]).dot(aoT_s.T)
    est['S6_get_j'] = (time.perf_counter() - t) * G / n1
    n7 = min(P, 1024)
    Wr = rng.standard_normal((n7, P))
    t = time.perf_counter()
    X = aoP.dot(dm)
    M = X[:n7].dot(aoP.T) * Wr
    aoP[:n7].T.dot(M.dot(aoP))
    est['S7_get_k'] = (time.perf_counter() - t) * P / n7
    total = sum(est.values())
    sample = ('numpy/scipy oracle (Cholesky fit route), stage samples extrapolated linearly: S1 %d of %d grid points; S2 1 of %d atom blocks '
              '(%d pts, %d pivots); S3 Cholesky of a %dx%d block of the %dx%d matrix (cubic extrapolation) + %d of %d grid columns; S4 %d of %d FFT rows; S5 %d rows x %d grid columns against all P rows; '
              'S6 %d grid points; S7 %d of %d rows; measured %.1f s of CPU work; per-stage estimate (s): %s'
              % (n1, G, natm, m_b, k_b, Pc, Pc, P, P, n3, G, n4, P, r5, n5, n1, n7, P, time.perf_counter() - t_all,
                 {k: round(v, 1) for k, v in est.items()}))
    return total, sample


def cpu_baseline_exact_fftdf(cell, nocc):
    """The reference's own algorithm for K (pyscf/pbc/df/fft_jk.py:276-287: one FFT pair per (occupied orbital, AO) pair
    density + the contraction with the AOs), timed on a few pairs with the oracle's FFT and extrapolated to the N*nocc
    pairs of the workload.  Returns (seconds for one exact get_k of the full workload, description)."""
    from oracle import pbc_tools as tools
    rng = np.random.default_rng(1)
    mesh = np.asarray(cell.mesh)
    G = int(np.prod(mesh))
    nao = cell.nao_nr()
    a = cell.lattice_vectors()
    coulG = tools.get_coulG(a, mesh)
    npair, nrow = 16, 256
    ao_rows = rng.standard_normal((nrow, G))                 # stand-ins for AO values on the grid (timing only)
    rho = rng.standard_normal((npair, G))
    t = time.perf_counter()
    v = tools.ifft(tools.fft(rho, mesh) * coulG, mesh).real   # fft_jk.py:281-284
    t_fft = (time.perf_counter() - t) / npair
    t = time.perf_counter()
    (ao_rows * v[0]).dot(ao_rows.T)                           # fft_jk.py:286: vk += einsum over the grid, one occupied orbital
    t_dot = (time.perf_counter() - t) * (nao / nrow) ** 2 / nao   # per pair: an (nao x G x nao) product serves nao pairs
    pairs = nao * nocc
    total = pairs * (t_fft + t_dot)
    return total, ('exact FFTDF exchange (fft_jk.py:276-287) extrapolated from %d FFT pairs (%.3f s each) and a %d-row slice of the '
                   'grid contraction (%.4f s per pair) to N*nocc = %d pairs' % (npair, t_fft, nrow, t_dot, pairs))


def scf_density(cell, c_isdf=10):
    """Physical orbitals for the accuracy entry: a closed-shell RHF driven by an AO-pair ISDF object (get_pp, J, K from the device;
    S and T by plane-wave quadrature of the collocated AOs with torch.fft - plumbing, outside every timed region).  Returns
    (dm, C_occ, occ)."""
    import scipy.linalg
    import torch
    from pyscf_isdf_amd.isdf import ISDF
    nao, nocc = cell.nao_nr(), cell.nelectron // 2
    mesh = [int(x) for x in cell.mesh]
    G = int(np.prod(mesh))
    df = ISDF(cell, c_isdf=c_isdf, select='refined')
    df.collocate()
    b = 2 * np.pi * np.linalg.inv(cell.lattice_vectors().T)
    fr = [np.fft.fftfreq(n, 1. / n) for n in mesh]
    Gv = (fr[0][:, None, None, None] * b[0] + fr[1][None, :, None, None] * b[1] + fr[2][None, None, :, None] * b[2]).reshape(-1, 3)
    g2 = df.backend.to_device(np.einsum('gi,gi->g', Gv, Gv))
    F = torch.empty((nao, G), dtype=torch.complex128, device=df.backend.device)
    for r0 in range(0, nao, 64):
        r1 = min(nao, r0 + 64)
        F[r0:r1] = torch.fft.fftn(df.ao[r0:r1].reshape(r1 - r0, *mesh), dim=(1, 2, 3)).reshape(r1 - r0, G)
    T = ((0.5 * cell.vol / G ** 2) * torch.matmul(F.conj() * g2, F.T).real).cpu().numpy()
    S = ((cell.vol / G ** 2) * torch.matmul(F.conj(), F.T).real).cpu().numpy()
    del F, g2
    df.reset()
    torch.cuda.empty_cache()
    hcore = T + df.get_pp()
    torch.cuda.empty_cache()
    e, c = scipy.linalg.eigh(hcore, S)
    dm = 2 * c[:, :nocc].dot(c[:, :nocc].T)
    errs, focks, e_last = [], [], 0.0
    for it in range(30):
        vj, vk = df.get_jk(dm)
        f = hcore + vj - 0.5 * vk
        e_el = 0.5 * np.einsum('ij,ji', hcore + f, dm)
        err = f.dot(dm).dot(S) - S.dot(dm).dot(f)
        focks.append(f); errs.append(err); focks, errs = focks[-8:], errs[-8:]
        n = len(focks)
        if n > 1:
            B = -np.ones((n + 1, n + 1)); B[n, n] = 0
            for i in range(n):
                for j in range(n):
                    B[i, j] = np.vdot(errs[i], errs[j])
            rhs = np.zeros(n + 1); rhs[n] = -1
            f = sum(ci * fi for ci, fi in zip(np.linalg.lstsq(B, rhs, rcond=None)[0][:n], focks))
        if abs(e_el - e_last) < 1e-8 and abs(err).max() < 1e-5:
            break
        e_last = e_el
        e, c = scipy.linalg.eigh(f, S)
        dm = 2 * c[:, :nocc].dot(c[:, :nocc].T)
    print('[bench] SCF orbitals ready after %d iterations (E_el %.8f)' % (it + 1, e_el), file=sys.stderr, flush=True)
    df.reset()
    del df
    torch.cuda.empty_cache()
    occ = np.zeros(nao)
    occ[:nocc] = 2
    return dm, np.ascontiguousarray(c), occ


def spawn_ranks(n):
    """python bench.py --gpus N with no launcher around it: start the N ranks as a fresh child (torch.distributed.run, one
    rank per GPU) BEFORE this process touches torch or the GPU, relay its output and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.c_isdf is None:
        # configs[2]: c = 18 is where the measured |dE_K| falls under the north star's 1e-6 Eh (3.8e-7; c = 17 / 19 read +1.6e-6 /
        # -2.0e-6: DESIGN.md section 2) - inside the 30 s budget since the spectral form of W; c = 12 (--c-isdf 12) is the fast
        # variant (9.3 s, -3.6e-5 Eh).  The same c at every rank count (strong scaling): the sharded spectral build holds a K slice of
        # X per rank (74 GB at 2 ranks), not the fit rows.
        args.c_isdf = 18 if args.workload == 'diamond-444-dzvp-120' else 10
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    import torch
    import torch.distributed as dist
    from pyscf_isdf_amd import workloads
    from pyscf_isdf_amd.isdf import ISDF
    from pyscf_isdf_amd.parallel import Comm

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        # production: one rank per GPU over RCCL.  ISDF_ONE_GPU=1 ISDF_DIST_BACKEND=gloo rehearses the N > 1 code
        # path with all ranks on device 0 (tools/rehearse_ranks_one_gpu.sh) - never a benchmark number.
        torch.cuda.set_device(0 if os.environ.get('ISDF_ONE_GPU') else local_rank)
        dist.init_process_group(os.environ.get('ISDF_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)
    comm = Comm.from_env()

    cell = workloads.make_cell(args.workload)
    kpts = workloads.make_kpts(args.workload, cell)
    if kpts is None:
        if args.density == 'scf':
            if world != 1:
                raise SystemExit('--density scf is a one-GPU option')
            dm, c_mo, occ_mo = scf_density(cell)
        else:
            dm, c_mo, occ_mo = workloads.make_dm(cell)
        if args.pair_space == 'occ':
            class _Tagged(np.ndarray):          # numpy_helper.tag_array's role: the density carries its orbitals
                pass
            dm = dm.view(_Tagged)
            dm.mo_coeff, dm.mo_occ = c_mo, occ_mo
        df = ISDF(cell, c_isdf=args.c_isdf, select=args.select, comm=comm)
        df.pair_space = args.pair_space
    else:
        # k-point workload (configs[3]): Hermitian D^k = C^k occ C^k^H with random unitary C^k
        nao = cell.nao_nr()
        rng = np.random.default_rng(20240203)
        occ = np.zeros(nao)
        occ[:cell.nelectron // 2] = 2
        dm = []
        for _ in range(len(kpts)):
            c = np.linalg.qr(rng.standard_normal((nao, nao)) + 1j * rng.standard_normal((nao, nao)))[0]
            dm.append((c * occ).dot(c.conj().T))
        dm = np.array(dm)
        df = ISDF(cell, kpts=kpts, c_isdf=args.c_isdf, select=args.select, comm=comm)
    df.refine_over = args.refine_over
    df.cand_ao_cutoff = args.cand_ao_cutoff
    if args.fit_route:
        df.fit_route = args.fit_route
    if args.robust_k:
        df.robust_k = True
    df.w_spectral = args.w_form == 'spectral'
    be = df.backend

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        df.build()
        return df.get_jk(dm) if kpts is None else df.get_jk(dm, kpts=kpts)

    # a progress line on stderr about once a minute: a long run (--steps 20) is otherwise silent for minutes, and a silent
    # GPU command is taken for a hung one by the box's watchdog; stdout carries the one JSON line only
    last_note = [time.perf_counter()]

    def progress(what, i, n):
        now = time.perf_counter()
        if rank == 0 and now - last_note[0] > 45.0:
            last_note[0] = now
            print('[bench] %s step %d of %d done, %.0f s since start' % (what, i + 1, n, now - T_PROCESS_START), file=sys.stderr, flush=True)

    # accuracy reference FIRST (outside every timed region): the reference's exact exchange (fft_jk.py:177-302, N*nocc FFT
    # pairs) for the benchmark density on this GPU - measured in this run, never read from a file
    vk_exact = None
    t_exact = 0.0
    if kpts is None and not args.no_accuracy:
        print('[bench] exact exchange on the GPU for the accuracy entry (about 40 s)', file=sys.stderr, flush=True)
        ta = time.perf_counter()
        vk_exact = df.get_k_exact(mo_coeff=c_mo, mo_occ=occ_mo)      # N > 1: the AO rows of K are shared out over the ranks
        torch.cuda.synchronize()
        t_exact = time.perf_counter() - ta
        df.release_fit_buffers()
        be.release_workspace()
        be.empty_cache()

    cold_first_step = None
    for i in range(args.warmup):
        torch.cuda.synchronize()
        tc = time.perf_counter()
        step()
        torch.cuda.synchronize()
        if i == 0:
            cold_first_step = time.perf_counter() - tc     # allocations, FFT plans, rocBLAS kernel loading: what a one-shot caller pays
        progress('warmup', i, args.warmup)
    be.prof_reset()
    be.prof_enable(True)
    comm.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        vj, vk = step()
        progress('timed', i, args.steps)
    barrier()
    t1 = time.perf_counter()
    be.prof_enable(False)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=be.device)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    sec_per_step = elapsed.item() / args.steps
    # what every rank spent where in its last step and what it put on the fabric in the timed steps (bytes always; seconds
    # when ISDF_COMM_TIMING=1: event pairs around each collective)
    mine = {'rank': rank, 'stage_seconds_last_step': {k: round(v, 4) for k, v in df.timings.items()},
            'comm': {k: {'calls': v['calls'], 'GB': round(v['bytes'] / 1e9, 3), 'seconds': round(v['seconds'], 4)}
                     for k, v in comm.collect_stats().items()}}
    per_rank = comm.all_gather_object(mine) if world > 1 else [mine]

    if rank == 0:
        prof = be.prof_results()
        rows = []
        for name, r in prof.items():
            kind = 'flop' if name.endswith('[flop]') else 'byte'
            per_launch_ms = r['ms'] / max(r['launches'], 1)
            rate = r['work'] / (r['ms'] * 1e-3) if r['ms'] > 0 else 0.0
            rows.append(dict(kernel=name, launches=r['launches'], total_ms=r['ms'], avg_ms=per_launch_ms,
                             achieved=(rate / 1e12 if kind == 'flop' else rate / 1e9),
                             unit=('TFLOP/s' if kind == 'flop' else 'GB/s'),
                             work_per_launch=r['work'] / max(r['launches'], 1)))
        rows.sort(key=lambda x: -x['total_ms'])
        # dominant HAND-WRITTEN kernel (library calls are listed in the stage report, not used for the roofline)
        own = [r for r in rows if not r['kernel'].startswith(('rocblas', 'rocsolver'))]
        dom = own[0] if own else rows[0]
        # HBM-side bytes per launch of that kernel come from rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950
        # correction + WRITE_SIZE, separate passes), which cannot run inside this process: the committed summary of the
        # same command (PMC_FILE, written by tools/pmc_summarise.py) is used ONLY when it describes this run - same
        # kernel, same launches per step, average launch duration within 5 % of the live HIP-event figure; else null.
        traffic = None
        traffic_note = None
        try:
            with open(os.path.join(ROOT, 'profiles', PMC_FILE)) as f:
                pmc = json.load(f)
            # algorithmic bytes of the launches that kernel handles: W batch r reads V (nb x G) and the rows r.. of Y' ((P - r) x G)
            # once and writes nb x (P - r); the 256x128 kernel takes every batch of more than 128 rows (gemm_f64.hip)
            Gn, Pn = int(np.prod(cell.mesh)), len(df.ip)
            if getattr(df, 'w_spectral_fraction', None):
                Gn = int(getattr(df, '_last_spectral_ldx', Gn))          # spectral form: strips of X (512 x ldx) against X ((P - r) x ldx)
            nbat = int(getattr(df, '_last_fft_batch', 0) or 512)
            algs = [8.0 * ((min(nbat, Pn - r) + (Pn - r)) * Gn + min(nbat, Pn - r) * (Pn - r))
                    for r in range(0, Pn, nbat) if min(nbat, Pn - r) > 128]
            base = dom['kernel'].split('[')[0]
            v = pmc.get(base)
            if v is None:
                traffic_note = 'no entry for %s in profiles/%s' % (base, PMC_FILE)
            else:
                fetch, write = v.get('FETCH_SIZE_per_launch'), v.get('WRITE_SIZE_per_launch')     # counter unit: KB
                per_step = dom['launches'] / max(args.steps, 1)
                if fetch is None or write is None:
                    traffic_note = 'profiles/%s lacks FETCH_SIZE or WRITE_SIZE for %s' % (PMC_FILE, base)
                elif v['launches'] != per_step or abs(v['avg_ms'] - dom['avg_ms']) > 0.05 * dom['avg_ms'] or world != 1 or not algs:
                    traffic_note = ('profiles/%s is for another run (%d launches of %.1f ms there, %.1f per step of %.1f ms here): the PMC '
                                    'passes were collected on the c = 12 variant of this build (--c-isdf 12: %.1f GB per launch of this '
                                    'kernel, 1.19 x its algorithmic bytes); at c = 18 they did not finish inside the GPU box\'s limit'
                                    % (PMC_FILE, v['launches'], v['avg_ms'], per_step, dom['avg_ms'], (2 * fetch + write) * 1024 / 1e9))
                else:
                    traffic = dict(bytes_per_launch=round((2 * fetch + write) * 1024),
                                   algorithmic_bytes_per_launch=round(sum(algs) / len(algs)),
                                   source='profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command, '
                                          'tools/pmc_summarise.py; FETCH x2 per MI355X_MICROARCH.md; counts fabric requests incl. '
                                          'Infinity-Cache hits)' % PMC_FILE)
        except (OSError, KeyError, ValueError) as e:
            traffic_note = 'profiles/%s unusable: %s' % (PMC_FILE, e)
        if dom['unit'] == 'TFLOP/s':
            roof = dict(bound='mfma', kernel=dom['kernel'], achieved=round(dom['achieved'], 2), peak=FP64_MFMA_PEAK_TFLOPS,
                        unit='TFLOP/s', frac=round(dom['achieved'] / FP64_MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                        avg_launch_ms=round(dom['avg_ms'], 3), launches=dom['launches'], traffic_note=traffic_note)
        else:
            roof = dict(bound='hbm', kernel=dom['kernel'], achieved=round(dom['achieved'], 1), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(dom['achieved'] / HBM_PEAK_GBS, 4), traffic=None, avg_launch_ms=round(dom['avg_ms'], 3),
                        launches=dom['launches'])
        # the HBM-bound hand-written kernels next to it (algorithmic bytes / HIP-event time), largest first
        hbm = [dict(kernel=r['kernel'], achieved=round(r['achieved'], 1), unit='GB/s', peak=HBM_PEAK_GBS,
                    frac=round(r['achieved'] / HBM_PEAK_GBS, 4), seconds_per_step=round(r['total_ms'] / 1e3 / max(args.steps, 1), 3))
               for r in own if r['unit'] == 'GB/s' and r['total_ms'] / max(args.steps, 1) >= 50.0]
        G = int(np.prod(cell.mesh))
        nao = cell.nao_nr()
        P = len(df.ip)
        alg_bytes = 40.0 * G * nao + 40.0 * G * P       # BASELINE.md section 3 (whole-path algorithmic bytes)
        out = {
            'metric': 'isdf_build_plus_get_jk_wall_time', 'value': round(sec_per_step, 4), 'unit': 's',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(sec_per_step * 1e3, 2),
            'higher_is_better': False, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': workloads.WORKLOADS[args.workload][1], 'natm': cell.natm, 'nao': nao, 'ngrids': G,
                       'nip': P, 'c_isdf': args.c_isdf, 'select': args.select, 'parallelism': ('grid-shard x%d' if kpts is None else 'q-shard x%d') % world,
                       'nkpts': (1 if kpts is None else len(kpts)),
                       'fit_route': df.fit_route, 'fit_route_used': df.fit_route_used, 'robust_k': bool(getattr(df, 'robust_k', False)),
                       'route_probe_mismatch': df.bj_check, 'route_probe_tol': df.bj_check_tol},
            'whole_path_algorithmic_GBps': round(alg_bytes / sec_per_step / 1e9, 1),
            'stage_seconds_last_step': {k: round(v, 4) for k, v in df.timings.items()},
            'energies': ({'EJ': float(np.einsum('ij,ji', vj, dm) / 2), 'EK': float(np.einsum('ij,ji', vk, dm) / 4)} if kpts is None else
                         {'EJ': float(np.einsum('kij,kji', vj, dm).real / 2 / len(kpts)),
                          'EK': float(np.einsum('kij,kji', vk, dm).real / 4 / len(kpts))}),
            'roofline': roof,
            'roofline_hbm_kernels': hbm,
        }
        out['cold_first_step_s'] = None if cold_first_step is None else round(cold_first_step, 3)
        if world > 1:
            out['per_rank'] = per_rank
            out['comm_timing'] = ('event pairs around every collective (ISDF_COMM_TIMING=1)' if comm.timing else
                                  'bytes only; set ISDF_COMM_TIMING=1 for seconds')
        out['config']['refine_over'] = args.refine_over if args.select == 'refined' else None
        out['config']['cand_ao_cutoff'] = args.cand_ao_cutoff if args.select == 'refined' else None
        out['config']['fit_row_panels'] = int(getattr(df, 'n_panels', 1))
        frac = getattr(df, 'w_spectral_fraction', None)
        out['config']['w_form'] = ('spectral: X X^T over %.3f G half-spectrum terms (sphere inscribed in the reciprocal FFT box)' % frac) if frac else 'classic: w conv(rows) rows^T over G grid points'
        out['config']['w_spectral_fraction'] = frac
        # accuracy of the timed configuration against the reference's exact exchange (fft_jk.py:177-302 on the GPU,
        # isdf_get_k_exact; outside the timed region).  J is the reference's own formula (fft_jk.py:33-109): no fit error.
        out['config']['dE_K_vs_exact'] = None
        out['config']['pair_space'] = args.pair_space
        out['config']['density'] = ('random orthogonal orbitals (BASELINE.json)' if args.density == 'random' else
                                    'SCF orbitals (RHF converged with the ISDF object before the timed steps)')
        if vk_exact is not None:
            ek_ex = float(np.einsum('ij,ji', vk_exact, np.asarray(dm)) / 4)
            out['config']['dE_K_vs_exact'] = float(out['energies']['EK'] - ek_ex)
            out['accuracy'] = {'E_K_exact': ek_ex, 'dE_K_Eh': out['config']['dE_K_vs_exact'],
                               'dE_K_Eh_per_atom': out['config']['dE_K_vs_exact'] / cell.natm,
                               'max_abs_dK': float(abs(vk - vk_exact).max()), 'source': 'measured in this run',
                               'north_star_tol_Eh': 1e-6, 'meets_north_star': bool(abs(out['config']['dE_K_vs_exact']) <= 1e-6),
                               'exact_K_seconds_on_this_gpu': round(t_exact, 1), 'dE_J_Eh': 0.0,
                               'note': 'exact = the reference algorithm (N*nocc FFT pairs) on the same GPU and grid, evaluated before '
                                       'the warm-up steps; J uses the reference formula itself; DESIGN.md section 2 has the scan over '
                                       'c, the selection and the pair space'}
        if os.environ.get('ISDF_ONE_GPU'):
            out['data'] = 'synthetic; REHEARSAL: %d ranks on one GPU over gloo, not a benchmark' % world
        if world == 1 and not args.no_cpu_baseline and kpts is None:
            # threads the numpy/scipy oracle really uses = the BLAS pool of this interpreter (not os.cpu_count())
            try:
                from threadpoolctl import threadpool_info
                pools = [p_.get('num_threads', 1) for p_ in threadpool_info() if p_.get('user_api') == 'blas']
                ncores = max(pools) if pools else 1
            except Exception:                      # noqa: BLE001 - threadpoolctl missing: report the host's cores
                ncores = os.cpu_count() or 1
            print('[bench] cpu_baseline sample on the host cores (about 30 s)', file=sys.stderr, flush=True)
            val, sample = cpu_baseline(cell, args.c_isdf, dict(P=P))
            out['cpu_baseline'] = {'value': round(val, 1), 'unit': 's', 'cores': ncores, 'kind': 'port',
                                   'sample': sample + '; host has %d logical CPUs' % (os.cpu_count() or 1)}
            # second entry: the reference's EXACT algorithm on the same host (what ISDF replaces), extrapolated per FFT pair
            xv, xs = cpu_baseline_exact_fftdf(cell, cell.nelectron // 2)
            out['cpu_baseline']['exact_fftdf'] = {'value': round(xv, 1), 'unit': 's per get_k', 'cores': ncores, 'kind': 'port', 'sample': xs}
        else:
            out['cpu_baseline'] = None
        if args.stage_report:
            with open(args.stage_report, 'w') as f:
                f.write('per-kernel table, %d timed steps (HIP events on the work stream)\n' % args.steps)
                for r in rows:
                    f.write('%-36s launches %6d  total %10.2f ms  avg %9.3f ms  %8.2f %s\n' %
                            (r['kernel'], r['launches'], r['total_ms'], r['avg_ms'], r['achieved'], r['unit']))
                f.write('stage wall times of the last step (s): %s\n' % out['stage_seconds_last_step'])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
