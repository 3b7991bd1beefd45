#!/bin/bash
# Everything a round's last GPU call should leave behind, in one gpurun call (run from the repository root on the GPU box):
#   scripts/round_end.sh r05_f            -> gpurun_out/ (copy what should be kept into profiles/)
#   1. the GPU test suite                                     (${TAG}_gpu_tests.txt)
#   2. scripts/profile_round.sh: PMC traffic passes, the default bench line, the rocprofv3 kernel trace of the same command
#   3. bench.py through torch.distributed.run with 2 and 3 ranks on this one device (ZLNG_BENCH_ONE_DEVICE=1: the N > 1 control flow
#      over gloo -- RCCL refuses two ranks on one device), every line with parity / roofline / cpu_baseline
#   4. config 4's per-GPU share (8 GiB at e4), one 4 GiB stream at e0 (longer than a context), K = 2 / 8 streams at once on the GPU, the decode line
#   5. scripts/ring_fix_ab.sh ring-only: the GPU suite and the e1/e4 lines with ZLNG_RING_FIX=1 against the same without it
#   scripts/round_end.sh r06_a quick      -> steps 1 and 2 only (~20 min): the suite and the profile set, for a GPU slot that opens late
set -u
TAG=${1:-r05_x}
QUICK=${2:-}
OUT=$PWD/gpurun_out; mkdir -p $OUT
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $OUT/${TAG}_gpu_tests.txt 2>&1; tail -3 $OUT/${TAG}_gpu_tests.txt
bash scripts/profile_round.sh $TAG
if [ "$QUICK" = quick ]; then exit 0; fi
export MASTER_ADDR=127.0.0.1
ZLNG_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
    bench.py --gpus 2 --steps 2 --warmup 1 --size 402653184 --no-multistream 2> $OUT/${TAG}_two_ranks.err | grep '^{' > $OUT/${TAG}_two_ranks_one_device.json
ZLNG_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29542 \
    bench.py --gpus 3 --steps 2 --warmup 1 --size 1000000000 --strong --no-multistream 2> $OUT/${TAG}_three_ranks.err | grep '^{' > $OUT/${TAG}_three_ranks_strong_one_device.json
python - <<PY
import json
for f in ("two_ranks_one_device", "three_ranks_strong_one_device"):
    try:
        d = json.load(open("$OUT/${TAG}_%s.json" % f))
        print(f, d["n_gpus"], d["value"], "parity", d["parity"], d["parity_ranges"]["source"][:40], "alt identical", d["alt_host_rank_chains"]["identical_bytes"])
    except Exception as e:
        print(f, "FAILED", e)
PY
timeout 900 python bench.py --level 4 --size 8589934592 --steps 1 --warmup 0 --no-multistream > $OUT/${TAG}_config4_share_e4_8GiB_1gpu.json 2> $OUT/${TAG}_config4.err
# one e0 stream longer than a context: 4 GiB through 2 contexts of 128 blocks, the second parse beside the first rank stage
timeout 900 python bench.py --size 4294967296 --steps 2 --warmup 1 --no-multistream --no-realtext > $OUT/${TAG}_long_stream_e0_4GiB_1gpu.json 2> $OUT/${TAG}_long.err
tail -c 200 $OUT/${TAG}_long_stream_e0_4GiB_1gpu.json; echo
# K independent streams at once on this one GPU (gpu_multistream; K = 4 is in the default bench line above): K = 2 and 8
for K in 2 8; do
  timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-realtext --gpu-multistream $K > $OUT/${TAG}_gpu_multistream_k$K.json 2> $OUT/${TAG}_gpu_multistream_k$K.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_gpu_multistream_k$K.json")); g = d.get("gpu_multistream") or d.get("extras_failed")
    print("gpu_multistream K=$K:", {k: g[k] for k in ("value", "seconds_per_round", "speedup_over_one_stream", "parity") if k in g} if "value" in g else g)
except Exception as e:
    print("gpu_multistream K=$K FAILED", e)
PY
done
timeout 900 python bench.py --decode --size 100000000 > $OUT/${TAG}_decode.json 2> $OUT/${TAG}_decode.err
tail -c 300 $OUT/${TAG}_config4_share_e4_8GiB_1gpu.json; echo; tail -c 300 $OUT/${TAG}_decode.json
echo
RING_ONLY=1 bash scripts/ring_fix_ab.sh $TAG
