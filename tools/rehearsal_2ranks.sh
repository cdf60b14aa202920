#!/bin/bash
# Two ranks on ONE GPU over gloo: launcher, rank set-up, StepPipeline and the packed gather of `bench.py --gpus 2` (not a measurement).
# Usage (GPU box): bash tools/rehearsal_2ranks.sh gpurun_out/r04
OUT=${1:-gpurun_out/r04}; mkdir -p $OUT
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 \
    --cpu-sample 0 --host-steps 0 --rehearse-on-one-gpu > $OUT/bench_rehearsal_2ranks.out 2> $OUT/bench_rehearsal_2ranks.err; echo "rehearsal rc $?"
grep '^{' $OUT/bench_rehearsal_2ranks.out > $OUT/bench_rehearsal_2ranks.json   # (gloo prints its connection lines on stdout; RCCL runs do not)
