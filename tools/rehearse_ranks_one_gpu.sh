#!/bin/bash
# Rehearse the N > 1 code path of bench.py on a ONE-GPU box: R ranks share device 0, collectives over gloo
# (staged through the host).  Checks the sharded orchestration with the real kernels; it is not a benchmark.
# usage: tools/rehearse_ranks_one_gpu.sh R WORKLOAD [extra bench.py args]
R=${1:-2}; W=${2:-diamond-222-dzvp-80}; shift 2
export ISDF_ONE_GPU=1 ISDF_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$R" --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus "$R" --steps 1 --warmup 1 --workload "$W" --no-cpu-baseline "$@"
