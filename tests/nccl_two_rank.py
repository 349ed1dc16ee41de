"""Two ranks over RCCL (backend "nccl"), one GPU each: the grid-sharded build and get_jk against the single-GPU path.  Run by
tests/test_gpu_parity.py::test_two_ranks_over_rccl_when_two_gpus_are_visible through torch.distributed.run; needs two GPUs
(the gpurun boxes have one: the test skips there; the driver's multi-GPU node runs it)."""
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
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local)
    dist.init_process_group('nccl', rank=rank, world_size=world)
    comm = Comm.from_env()
    cell = cells.cell_diamond_prim('gth-szv', (12, 12, 12))
    nao = cell.nao_nr()
    rng = np.random.default_rng(5)
    dm = rng.standard_normal((2, nao, nao))
    dm = dm + dm.transpose(0, 2, 1)
    ok = True
    # local + auto, refined + Cholesky, and what the headline runs: refined selection, c = 12, block-Jacobi route behind its probe check
    for select, route, c_isdf in (('local', 'auto', 4), ('refined', 'cholesky', 4), ('refined', 'auto', 12)):
        df = ISDF(cell, c_isdf=c_isdf, select=select, comm=comm)
        df.fit_route, df.bj_check_tol, df.fft_batch = route, 1e-6, 7
        vj, vk = df.get_jk(dm)
        if rank == 0:
            ref = ISDF(cell, c_isdf=c_isdf, select=select, comm=Comm())
            ref.fit_route, ref.bj_check_tol = route, 1e-6
            vj0, vk0 = ref.get_jk(dm)
            ok = ok and np.array_equal(df.ip, ref.ip) and abs(vj - vj0).max() < 1e-10 and abs(vk - vk0).max() < 1e-7 * abs(vk0).max()
            ok = ok and df.fit_route_used == ref.fit_route_used
    # the (AO x occupied) pair space through the sharded build: the fit is made inside get_jk from the tagged density
    c = np.linalg.qr(rng.standard_normal((nao, nao)))[0]
    occ = np.zeros(nao); occ[:3] = 2

    class Tagged(np.ndarray):
        pass
    tdm = (c * occ).dot(c.T).view(Tagged)
    tdm.mo_coeff, tdm.mo_occ = c, occ
    df = ISDF(cell, c_isdf=3, select='refined', comm=comm)
    df.pair_space, df.fit_route, df.fft_batch = 'occ', 'cholesky', 7
    vk = df.get_jk(tdm, with_j=False)[1]
    if rank == 0:
        ref = ISDF(cell, c_isdf=3, select='refined', comm=Comm())
        ref.pair_space, ref.fit_route = 'occ', 'cholesky'
        vk0 = ref.get_jk(tdm, with_j=False)[1]
        ok = ok and np.array_equal(df.ip, ref.ip) and abs(vk - vk0).max() < 1e-7 * abs(vk0).max()
    dist.barrier()
    if rank == 0:
        print('RCCL-TWO-RANK OK' if ok else 'RCCL-TWO-RANK MISMATCH', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
