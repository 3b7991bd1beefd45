#!/bin/bash
# Everything that can be verified WITHOUT a GPU, in one go (round 6 ran with the GPU pool closed; this is its evidence, reproducible):
#   1. build: every .hip for gfx950, the shim, the CLI, the checker, the real reference (oracle/_ref, when /root/reference is there)
#   2. the CPU suite (oracle vs reference and goldens, hostile streams small and big, host logic, the shim on the stand-in ABI,
#      ISA hygiene incl. "every kernel of the last GPU run is instruction-identical at HEAD")
#   3. the same host code under ASan + UBSan on the stand-in ABI; the CPU suite on the ASan build of the checker
#   4. kernel machine code against the last commit before the pool closed; the ring rule's registers
#   5. (with `soak`) the checker against the real reference on 2,400 big hostile mutants and 3,000 small ones
# Usage: scripts/verify_cpu.sh [soak]      -> gpurun_out/verify_cpu_*.log
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
OUT=$ROOT/gpurun_out; mkdir -p $OUT
python __graft_entry__.py > $OUT/verify_cpu_build.log 2>&1 && echo "build ok" || { echo "BUILD FAILED"; tail -5 $OUT/verify_cpu_build.log; exit 1; }
python -m pytest tests -q -m "not gpu" -p no:cacheprovider 2>&1 | tee $OUT/verify_cpu_suite.log | tail -2
bash scripts/sanitize.sh host-cpu 2>&1 | tail -2
bash scripts/sanitize.sh cpu 2>&1 | tail -2
if git rev-parse b703640 > /dev/null 2>&1; then python scripts/isa_diff.py b703640 2>&1 | tee $OUT/verify_cpu_isa.log | tail -1; fi
python scripts/kernel_resources.py rolz_wg.hip 2>&1 | grep "false, false, false, false" | tee $OUT/verify_cpu_registers.log
python scripts/experiments/replay_split_model.py 2>&1 | tee $OUT/verify_cpu_replay_model.log | grep -E "ceiling|verdict" | head -3
if [ "${1:-}" = soak ]; then
  python scripts/oracle_ref_hostile_soak.py 41000 4 600 big 2>&1 | tee $OUT/verify_cpu_soak_big.log | head -1
  python scripts/oracle_ref_hostile_soak.py 31000 2 1500 2>&1 | tee $OUT/verify_cpu_soak_small.log | head -1
fi
