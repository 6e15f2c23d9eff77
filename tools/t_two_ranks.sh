#!/bin/bash
# Exercise the N > 1 path of bench.py on a 1-GPU box: two ranks share cuda:0, collectives over gloo (control flow only).
APX_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
    bench.py --gpus 2 --steps 2 --warmup 1 --no_cpu_baseline
