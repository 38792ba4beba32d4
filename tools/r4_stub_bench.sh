#!/bin/bash
# Runs ON THE GPU BOX: bench.py --gpus N as the driver launches it, N processes on the ONE GPU of the box over the RCCL stand-in of tests/_rccl_stub (plumbing, not
# performance: the ranks share a device and every exchange is staged through the host).  usage: tools/r4_stub_bench.sh N [bench args]
N=$1; shift
R=$GRAFT_REPO_ROOT
g++ -O2 -std=c++17 -fPIC -shared -o /tmp/librccl_stub.so $R/tests/_rccl_stub/rccl_stub.cpp -ldl -lrt -lpthread || exit 1
LD_PRELOAD=/tmp/librccl_stub.so WT_BENCH_FORCE_DEVICE=0 WT_BENCH_TORCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 \
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29760 $R/bench.py --gpus $N --cpu-steps 0 "$@"
