#!/bin/bash
# The GPU suite's HARNESS code executed on the CPU: `pytest -m gpu` with the library, the CLI and the protocol driver replaced by the
# stand-in stack of tests/stub_build.py (tests/cxx/zlng_stub.c: the C-ABI's contract on the checker) and bench.py in its stand-in mode.
# It says nothing about the kernels -- the codec under test is the checker itself -- but every test that is NOT about a device-only
# facility (the zlng_debug_* hooks, torch CUDA tensors) runs to the end, so that a test written while the GPU pool was closed
# (tests/test_gpu_zz_*.py, the per-call-traits case, the ten-block CLI decode) does not meet a GPU with a harness error in it.
#   scripts/gpu_suite_on_stub.sh [-n 4]      -> gpurun_out/gpu_suite_on_stub.log
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
OUT=$ROOT/gpurun_out; mkdir -p $OUT
python tests/stub_build.py > /dev/null || exit 1
S=$ROOT/tests/cxx/_stub
ZLNG_HIP_SO=$S/libzlng_hip.so ZLNG_DEMO=$S/zling_demo ZLNG_PROTOCOL_TEST=$S/protocol_test ZLNG_BENCH_STANDIN=1 \
  python -m pytest tests -m gpu -q -p no:cacheprovider "$@" 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | tee $OUT/gpu_suite_on_stub.log
