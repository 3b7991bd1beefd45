#!/bin/bash
# Round-end measurement set on the GPU box (one gpurun call): the two PMC passes for HBM traffic first (so that the bench line
# of the same run can quote them: bench.py reads the newest profiles/r*_pmc_traffic.json whose kernel-source hash matches),
# then the default bench with the CPU baseline, then the rocprofv3 kernel trace of the same command.
# Usage: scripts/profile_round.sh r02_a
set -u
TAG=${1:-r01_x}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_${TAG}_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /tmp/pmc_${TAG}_$C.log 2>&1
  cp $(find /tmp/pmc_${TAG}_$C -name "*_results.db" | head -1) /tmp/${TAG}_$C.db
done
python $GRAFT_REPO_ROOT/scripts/pmc_traffic.py /tmp/${TAG}_FETCH_SIZE.db /tmp/${TAG}_WRITE_SIZE.db 1000000000 0 60 > $OUT/${TAG}_pmc_traffic.json
cp $OUT/${TAG}_pmc_traffic.json $GRAFT_REPO_ROOT/profiles/${TAG}_pmc_traffic.json
python $GRAFT_REPO_ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -1 $OUT/${TAG}_bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/${TAG}_prof_bench.json 2> /tmp/prof_$TAG.err
cp $(find /tmp/prof_$TAG -name "*_results.db" | head -1) $OUT/${TAG}_results.db
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $OUT/${TAG}_results.db > $OUT/${TAG}_kernel_stats.txt
head -c 600 $OUT/${TAG}_kernel_stats.txt
