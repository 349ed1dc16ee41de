"""Run by tests/test_gpu_parity.py::test_rccl_code_paths_execute_on_one_gpu as a child process: ONE rank,
torch.distributed backend "nccl" (= RCCL), Comm(always=True) so that every collective of the grid-sharded build and of the
q-sharded k-point build is really issued on the device (broadcast, all_reduce SUM / MAX, list all_to_all,
all_gather_object).  The results must equal the plain single-GPU path.  Prints 'RCCL-ONE-RANK OK' on success."""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import torch
    import torch.distributed as dist
    import cells
    from pyscf_isdf_amd.isdf import ISDF
    from pyscf_isdf_amd.parallel import Comm
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%s' % sys.argv[1], rank=0, world_size=1)
    assert dist.get_backend() == 'nccl'
    live = Comm(0, 1, 0, always=True)
    cell = cells.cell_diamond_prim('gth-szv', (12, 12, 12))
    nao = cell.nao_nr()
    rng = np.random.default_rng(5)
    dm = rng.standard_normal((2, nao, nao))
    dm = dm + dm.transpose(0, 2, 1)
    for select, route, robust in (('local', 'auto', False), ('refined', 'cholesky', False), ('local', 'cholesky', True)):
        ref = ISDF(cell, c_isdf=4, select=select, comm=Comm())
        ref.fit_route, ref.robust_k, ref.bj_check_tol = route, robust, 1e-6
        vj0, vk0 = ref.get_jk(dm)
        df = ISDF(cell, c_isdf=4, select=select, comm=live)
        df.fit_route, df.robust_k, df.bj_check_tol = route, robust, 1e-6
        df.fft_batch = 7                                            # several pipelined exchange steps, ragged last batch
        vj, vk = df.get_jk(dm)
        assert df._fit_state is None or df._fit_state.get('sharded')
        assert np.array_equal(df.ip, ref.ip), (select, route)
        assert abs(vj - vj0).max() < 1e-10 and abs(vk - vk0).max() < 1e-7 * abs(vk0).max(), (select, route, abs(vk - vk0).max())
        vkl = df.get_jk(dm, omega=0.3, with_j=False)[1] if not robust else None
        if vkl is not None:
            assert abs(vkl - ref.get_jk(dm, omega=0.3, with_j=False)[1]).max() < 1e-7 * abs(vk0).max()
    # the (AO x occupied) pair space through the sharded code path (fit inside get_jk from the MO-tagged density) and the
    # headline's setting (refined selection, c = 12, block-Jacobi behind its probe check)
    c0 = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:3] = 2

    class Tagged(np.ndarray):
        pass
    tdm = (c0 * occ).dot(c0.T).view(Tagged)
    tdm.mo_coeff, tdm.mo_occ = c0, occ
    for space, c_isdf, route in (('occ', 3, 'cholesky'), ('ao', 12, 'auto')):
        ref = ISDF(cell, c_isdf=c_isdf, select='refined', comm=Comm())
        ref.pair_space, ref.fit_route, ref.bj_check_tol = space, route, 1e-6
        k0 = ref.get_jk(tdm, with_j=False)[1]
        df = ISDF(cell, c_isdf=c_isdf, select='refined', comm=live)
        df.pair_space, df.fit_route, df.bj_check_tol, df.fft_batch = space, route, 1e-6, 7
        k1 = df.get_jk(tdm, with_j=False)[1]
        assert df._fit_state.get('sharded') and np.array_equal(df.ip, ref.ip) and df.fit_route_used == ref.fit_route_used
        assert abs(k1 - k0).max() < 1e-7 * abs(k0).max(), (space, abs(k1 - k0).max())
    # q-sharded k-point build
    cellk = cells.cell_he2_triclinic()
    cellk.mesh = np.array([10, 10, 10])
    kpts = cellk.make_kpts([2, 1, 1])
    n = cellk.nao_nr()
    c = rng.standard_normal((2, n, n)) + 1j * rng.standard_normal((2, n, n))
    dms = np.einsum('kpi,kqi->kpq', c[:, :, :2], c[:, :, :2].conj())
    a = ISDF(cellk, kpts=kpts, c_isdf=6, select='local', comm=Comm())
    b = ISDF(cellk, kpts=kpts, c_isdf=6, select='local', comm=live)
    ja, ka = a.get_jk(dms, kpts=kpts)
    jb, kb = b.get_jk(dms, kpts=kpts)
    assert abs(ja - jb).max() < 1e-10 and abs(ka - kb).max() < 1e-8 * abs(ka).max()
    dist.barrier()
    dist.destroy_process_group()
    print('RCCL-ONE-RANK OK', flush=True)


if __name__ == '__main__':
    main()
