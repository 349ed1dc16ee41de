"""The N > 1 path on CPU: two gloo ranks run the host driver's grid-sharded build and row-sharded K
with the checker backend, and must reproduce the single-rank result (same backend) — this tests the
sharding, the two all-to-alls and the all-reduces, not the kernels."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _setup_path():
    for p in (os.path.dirname(HERE), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run_case(comm, fft_batch, route='cholesky'):
    _setup_path()
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_diamond_prim('gth-szv', (10, 9, 8))
    nao = cell.nao_nr()
    rng = np.random.default_rng(5)
    dm = rng.standard_normal((2, nao, nao))
    dm = dm + dm.transpose(0, 2, 1)
    occ_space = route in ('occ', 'occ-bj')
    if occ_space:
        # (AO x occupied) pair space: a density with occupied orbitals (mo_coeff / mo_occ tag), the fit made inside get_jk
        c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
        occ = np.zeros(nao); occ[:3] = 2

        class Tagged(np.ndarray):
            pass
        dm = (c * occ).dot(c.T).view(Tagged)
        dm.mo_coeff, dm.mo_occ = c, occ
    df = ISDF(cell, c_isdf=3, select='local' if route in ('cholesky', 'blockjacobi', 'auto', 'robust') else 'refined',
              backend=OracleBackend(), comm=comm)
    if occ_space:
        df.pair_space = 'occ'
    df.fft_batch = fft_batch
    df.fit_route = 'cholesky' if route in ('robust', 'refined', 'occ') else ('blockjacobi' if route in ('occ-bj', 'spectral') else ('auto' if route == 'spectral-auto' else route))
    if route == 'spectral-auto' and comm.size > 1:
        df.bj_max_c = 2                        # above bj_max_c the sharded build still tries the spectral form first (route 'auto')
    if route in ('spectral', 'spectral-auto'):
        df.w_spectral_check_tol = 1e-6
        df.w_sphere = 0                        # W = X X^T over the whole half spectrum: exact, K slices exchanged instead of V slices
    df.robust_k = route == 'robust'          # Dunlap's correction: V slices through the same all-to-alls, K1 all-reduced
    df.bj_check_tol = 1e-6                     # c_isdf=3 on 8 AOs: the check value is ~1e-8, far from the decision
    df.build()
    if occ_space:
        assert df._fit_pending and df.W is None
    vj, vk = df.get_jk(dm)
    # range separation through the sharded S4/S5: long range + short range = full, on every layout
    if route != 'robust':
        vjl, vkl = df.get_jk(dm, omega=0.4)
        vjs, vks = df.get_jk(dm, omega=-0.4)
        assert abs(vjl + vjs - vj).max() < 1e-11 and abs(vkl + vks - vk).max() < 1e-7 * abs(vk).max()
    if route == 'refined':
        # the verification path shards too: every rank the whole grid, a share of the AO rows, all-reduce
        kx = df.get_k_exact(dm[0].dot(dm[0].T))
        if comm.size > 1:
            ref = ISDF(cell, c_isdf=3, backend=OracleBackend()).get_k_exact(dm[0].dot(dm[0].T))
            assert abs(kx - ref).max() < 1e-12 * abs(ref).max()
    if route in ('spectral', 'spectral-auto'):
        assert df.w_spectral_fraction is not None and df.w_spectral_fraction > 1.0
        if comm.size > 1:
            assert df._fit_state['kind'] == 'blockjacobi-spectral' and df._fit_state['theta'] is None      # no resident rows
    else:
        assert df.w_spectral_fraction is None          # (this coarse mesh does not resolve the pair products: 'auto' declines)
    return df.ip.copy(), df.W.numpy().copy(), vj, vk, df.fit_route_used, df.bj_check


def _worker(rank, world, port, q, route):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _setup_path()
    from pyscf_isdf_amd.parallel import Comm
    out = _run_case(Comm.from_env(), 5, route)                   # ragged batches: 12 rows per rank, 5 per step
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('route', ['cholesky', 'blockjacobi', 'auto', 'robust', 'refined', 'occ', 'occ-bj', 'spectral', 'spectral-auto'])
def test_two_ranks_match_one_rank(route):
    _setup_path()
    from pyscf_isdf_amd.parallel import Comm
    ip1, W1, vj1, vk1, used1, chk1 = _run_case(Comm(), None, route)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, route)) for r in range(2)]
    for p in procs:
        p.start()
    ip2, W2, vj2, vk2, used2, chk2 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(ip1, ip2)
    assert used1 == used2 == ('cholesky' if route in ('cholesky', 'robust', 'refined', 'occ') else 'blockjacobi')
    if route in ('auto', 'spectral-auto'):
        # the probe energies are all-reduced over the grid slices: both layouts measure the same mismatch
        assert chk1 is not None and abs(chk1 - chk2) <= 0.5 * chk1 + 1e-13
    assert abs(W1 - W2).max() < (1e-9 if route in ('cholesky', 'robust', 'refined') else 1e-6) * abs(W1).max() or route in ('occ', 'occ-bj')
    assert abs(vj1 - vj2).max() < 1e-10 and abs(vk1 - vk2).max() < 1e-8 * abs(vk1).max()


@pytest.mark.timeout(300)
def test_three_ranks_ragged_shares_match_one_rank():
    """world_size 3: nothing divides evenly (720 grid points -> 240 each but 2 atoms over 3 ranks, 24 points -> 8 rows per rank in
    batches of 5, one rank without an atom block): the refined-selection build on the block-Jacobi route must reproduce the
    single-rank result."""
    _setup_path()
    from pyscf_isdf_amd.parallel import Comm
    ip1, W1, vj1, vk1, used1, chk1 = _run_case(Comm(), None, 'refined')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q, 'refined')) for r in range(3)]
    for p in procs:
        p.start()
    ip3, W3, vj3, vk3, used3, chk3 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(ip1, ip3) and used1 == used3
    assert abs(W1 - W3).max() < 1e-9 * abs(W1).max()
    assert abs(vj1 - vj3).max() < 1e-10 and abs(vk1 - vk3).max() < 1e-8 * abs(vk1).max()


@pytest.mark.timeout(300)
def test_three_ranks_spectral_form_block_aligned_shares():
    """The spectral form of W on three ranks: the points are dealt in whole preconditioner blocks (2 atoms -> one rank gets no
    block at all), the rows of every rank's batch are made on the fly on each grid slice, K slices of X are exchanged; with the
    probe check alongside.  Same points, same K as one rank."""
    _setup_path()
    from pyscf_isdf_amd.parallel import Comm
    ip1, W1, vj1, vk1, used1, chk1 = _run_case(Comm(), None, 'spectral-auto')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 3, port, q, 'spectral-auto')) for r in range(3)]
    for p in procs:
        p.start()
    ip3, W3, vj3, vk3, used3, chk3 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(ip1, ip3) and used1 == used3 == 'blockjacobi'
    assert chk1 is not None and chk3 is not None
    assert abs(W1 - W3).max() < 1e-6 * abs(W1).max()
    assert abs(vj1 - vj3).max() < 1e-10 and abs(vk1 - vk3).max() < 1e-8 * abs(vk1).max()


def _run_kcase(comm):
    _setup_path()
    import cells
    from oracle_backend import OracleBackend
    from pyscf_isdf_amd.isdf import ISDF
    cell = cells.cell_he2_triclinic()
    cell.mesh = np.array([9, 9, 9])
    kpts = cell.make_kpts([2, 1, 1])
    nao = cell.nao_nr()
    rng = np.random.default_rng(2)
    c = rng.standard_normal((2, nao, nao)) + 1j * rng.standard_normal((2, nao, nao))
    dms = np.einsum('kpi,kqi->kpq', c[:, :, :2], c[:, :, :2].conj())
    df = ISDF(cell, kpts=kpts, c_isdf=4, select='refined', backend=OracleBackend(), comm=comm)
    vj, vk = df.get_jk(dms, kpts=kpts)
    return df.ip.copy(), vj, vk


def _kworker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _setup_path()
    from pyscf_isdf_amd.parallel import Comm
    out = _run_kcase(Comm.from_env())
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_kpoints_two_ranks_share_the_q_list():
    """k-point path: the W^q builds and the K pair terms are split over the ranks by q; two gloo ranks
    must reproduce the one-rank J and K (all-reduce of the K partial sums)."""
    _setup_path()
    from pyscf_isdf_amd.parallel import Comm
    ip1, vj1, vk1 = _run_kcase(Comm())
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_kworker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ip2, vj2, vk2 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(ip1, ip2)
    assert abs(vj1 - vj2).max() < 1e-10 and abs(vk1 - vk2).max() < 1e-9 * abs(vk1).max()
    assert abs(vk1 - vk1.conj().transpose(0, 2, 1)).max() < 1e-8 * abs(vk1).max()


def test_split_range_covers_everything():
    _setup_path()
    from pyscf_isdf_amd.parallel import Comm
    for n in (0, 1, 7, 8, 1001):
        for size in (1, 2, 3, 8):
            parts = [Comm(r, size).split_range(n) for r in range(size)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(size - 1))
            assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= 1


def _root_fail_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    _setup_path()
    from pyscf_isdf_amd.parallel import Comm
    comm = Comm.from_env()
    out = []
    out.append(comm.run_on_root(lambda: 42))                       # value on rank 0, None elsewhere

    def boom():
        raise ValueError('root-only stage failed')
    try:
        comm.run_on_root(boom)
        out.append('no error')
    except ValueError as e:
        out.append('ValueError: %s' % e)
    except RuntimeError as e:
        out.append('RuntimeError: %s' % e)
    t = torch.full((3,), float(rank + 1), dtype=torch.float64)
    comm.broadcast(t, 0)
    out.append(t.tolist())
    out.append(comm.agree_max(0.5 + rank))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_root_only_stage_failure_reaches_every_rank():
    """A root-only stage (the P x P factorisations of the sharded build) that fails must fail on every rank instead of
    leaving the others in the next collective; broadcast and agree_max give every rank the root's bits / one decision."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_root_fail_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == 42 and res[1][0] is None
    assert res[0][1].startswith('ValueError') and res[1][1].startswith('RuntimeError')
    assert res[0][2] == res[1][2] == [1.0, 1.0, 1.0]
    assert res[0][3] == res[1][3] == 1.5
