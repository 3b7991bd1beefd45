#!/bin/bash
# A/B of the LDS-resident hot bucket (ZLNG_WG_HOT): parse time (perf_probe, same box) and HBM traffic of the parser (two PMC passes each).
# Usage (GPU box): scripts/hot_bucket_ab.sh > gpurun_out/hot_bucket_ab.txt
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for H in 0 1; do
  export ZLNG_WG_HOT=$H
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/hb_${H}_$C -o p -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /tmp/hb_${H}_$C.log 2>&1
    cp $(find /tmp/hb_${H}_$C -name "*_results.db" | head -1) /tmp/hb_${H}_$C.db
  done
  echo "== ZLNG_WG_HOT=$H"
  python $R/scripts/pmc_traffic.py /tmp/hb_${H}_FETCH_SIZE.db /tmp/hb_${H}_WRITE_SIZE.db 1000000000 0 60 | python -c "
import sys, json
k = json.load(sys.stdin)['kernels']['k_rolz_parse_wg']
print('k_rolz_parse_wg: FETCH_SIZE %.1f GB raw, WRITE_SIZE %.1f GB, hbm_bytes (2*F+W) %.1f GB, %.1f ms under the counters' % (k['FETCH_SIZE_KB'] * 1024 / 1e9, k['WRITE_SIZE_KB'] * 1024 / 1e9, k['hbm_bytes_corrected'] / 1e9, k['dur_ms_fetch']))"
  for i in 1 2; do python $R/scripts/perf_probe.py 1024 0 2>&1 | grep -A3 "iter 1" | grep rolz_parse; done
done
