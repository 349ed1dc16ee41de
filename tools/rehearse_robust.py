"""Rehearsal helper (run under tools/rehearse_ranks_one_gpu.sh-style env): robust_k on R ranks sharing one GPU vs one rank."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from pyscf_isdf_amd import workloads
from pyscf_isdf_amd.isdf import ISDF
from pyscf_isdf_amd.parallel import Comm
world = int(os.environ.get('WORLD_SIZE', '1'))
if world > 1:
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=world)
comm = Comm.from_env()
cell = workloads.make_cell('diamond-222-dzvp-80')
dm = workloads.make_dm(cell)[0]
df = ISDF(cell, c_isdf=10, comm=comm)
df.robust_k = True
vj, vk = df.get_jk(dm)
if comm.rank == 0:
    print('ranks %d  EJ %.12f  EK(robust) %.12f' % (world, np.einsum('ij,ji', vj, dm) / 2, np.einsum('ij,ji', vk, dm) / 4), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
