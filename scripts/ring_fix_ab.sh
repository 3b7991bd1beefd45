#!/bin/bash
# Round 5, first GPU call after the pool re-opened: the GPU suite as shipped, the same suite with the ring rule on
# (ZLNG_RING_FIX=1), then config 4's per-GPU share and the 10^9 B text at e1..e4 with and without the rule.
#   scripts/ring_fix_ab.sh r05_f   -> gpurun_out/${TAG}_*
set -u
TAG=${1:-r05_x}
OUT=$PWD/gpurun_out; mkdir -p $OUT
# RING_ONLY=1 (from scripts/round_end.sh, which has run the suite as shipped already): skip the default suite
if [ "${RING_ONLY:-0}" != 1 ]; then
  (time timeout 1500 python -m pytest tests -m gpu -x -q) > $OUT/${TAG}_gpu_tests.txt 2>&1; tail -3 $OUT/${TAG}_gpu_tests.txt
fi
(time ZLNG_RING_FIX=1 timeout 1500 python -m pytest tests -m gpu -x -q) > $OUT/${TAG}_gpu_tests_ring_fix.txt 2>&1; tail -3 $OUT/${TAG}_gpu_tests_ring_fix.txt
for RF in 0 1; do
  ZLNG_RING_FIX=$RF timeout 900 python bench.py --level 4 --size 8589934592 --steps 1 --warmup 0 --no-multistream --no-realtext \
      > $OUT/${TAG}_config4_share_ring${RF}.json 2> $OUT/${TAG}_config4_ring${RF}.err
  tail -c 400 $OUT/${TAG}_config4_share_ring${RF}.json; echo
  for L in 1 4; do
    ZLNG_RING_FIX=$RF timeout 600 python bench.py --level $L --steps 3 --warmup 1 --no-multistream --no-cpu-baseline \
        > $OUT/${TAG}_e${L}_1e9_ring${RF}.json 2> $OUT/${TAG}_e${L}_ring${RF}.err
  done
done
python - <<PY
import json
for rf in (0, 1):
    for f in ("config4_share", "e1_1e9", "e4_1e9"):
        try:
            d = json.load(open("$OUT/${TAG}_%s_ring%d.json" % (f, rf)))
            print(f, "ring", rf, "ms_per_step", d["ms_per_step"], "value", d["value"], "parity", d.get("parity"),
                  "parse", d.get("stage_ms", {}).get("rolz_parse"), "realtext", d.get("value_realtext"))
        except Exception as e:
            print(f, rf, "FAILED", e)
PY
