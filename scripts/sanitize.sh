#!/bin/bash
# Sanitizer pass ("Race detection / sanitizers").
#   scripts/sanitize.sh cpu     ASan + UBSan build of the CPU restatement, the whole `-m "not gpu"` suite on it (no GPU needed)
#   scripts/sanitize.sh device  on the GPU box: every kernel source except mtf_rank.hip (its inline-asm scalar loads do not assemble under the
#                               instrumentation) built with -fsanitize=address for gfx950:xnack+ (libzlng_hip_asan.so, built in the container by
#                               `scripts/sanitize.sh device-build`), the small golden streams through it at e0 and e4.  This image has no ASan
#                               build of the HIP runtime, so a device report cannot be printed: a detected error shows up as
#                               "Hostcall: no handler found for service ID 4" (checked with scripts/ubench/asan_probe.hip, which writes out of
#                               bounds on purpose); the pass is clean when that line is absent and the outputs are bit-exact.
#   scripts/sanitize.sh host    on the GPU box: the C++ shim and zling_demo built with ASan + UBSan, the CLI tests through them
#                               and the callback-protocol tests (tests/cxx/protocol_test.cpp) through them
#                               (host side of the product path: Inputter/Outputter loops, batching, the helper thread, the exact-pull Decode)
#   scripts/sanitize.sh host-tsan no GPU needed: the same host code under ThreadSanitizer on the stand-in ABI (helper thread of the shim's pipeline, copy-out threads of the group driver)
#   scripts/sanitize.sh host-cpu  no GPU needed: the same host code (shim, group driver, zling_demo, protocol_test) under ASan + UBSan on the CPU
#                               stand-in of the C-ABI (tests/cxx/zlng_stub.c, tests/stub_build.py --sanitize), the CLI / protocol / error-order tests through it
# Logs go to gpurun_out/ (copy what should be kept into profiles/).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
LIBASAN=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
case "${1:-cpu}" in
cpu)
  make -s -C $ROOT/oracle asan || exit 1
  # (tests that drive the REAL reference are deselected: it is not instrumented, and its C++ exceptions do not survive a preloaded ASan runtime)
  LD_PRELOAD=$LIBASAN ZLNG_NO_REF=1 ZLNG_ORACLE_SO=$ROOT/oracle/_asan/liboracle.so python -m pytest $ROOT/tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_oracle_vs_ref.py 2>&1 | tee $OUT/sanitize_cpu.log | tail -5
  ;;
device-build)
  B=/tmp/hipasan; mkdir -p $B; cd $B
  for f in $ROOT/libzling_amd/csrc/*.hip; do
    b=$(basename $f .hip); san="-fsanitize=address -shared-libsan -O1"; [ $b = mtf_rank ] && san="-O3"
    /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack+ $san -g -std=c++17 -fPIC -munsafe-fp-atomics -c $f -o $b.o || exit 1
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -shared -fPIC -o $ROOT/libzling_amd/libzlng_hip_asan.so *.o
  ;;
device)
  export LD_LIBRARY_PATH=$(dirname $(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)):${LD_LIBRARY_PATH:-}
  ( echo "== the probe (writes out of bounds on purpose): must show the hostcall line"; HSA_XNACK=1 timeout 60 $ROOT/scripts/ubench/asan_t 2>&1 | tail -2
    echo "== small golden streams e0 / e4 through the instrumented kernels"
    LD_PRELOAD=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1) HSA_XNACK=1 ZLNG_HIP_SO=$ROOT/libzling_amd/libzlng_hip_asan.so \
      timeout 900 python $ROOT/scripts/wg_probe.py small 0,4 2>&1 | tail -30 ) | tee $OUT/sanitize_device.log
  grep -c "no handler found for service ID 4" $OUT/sanitize_device.log
  ;;
host)
  B=/tmp/zlng_asan; mkdir -p $B
  g++ -std=c++14 -O1 -g -fPIC -shared -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -I $ROOT/include -I $ROOT/include/libzling \
      -o $B/libzling_amd.so $ROOT/libzling_amd/cxx/*.cpp -L $ROOT/libzling_amd -lzlng_hip -Wl,-rpath,$ROOT/libzling_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib || exit 1
  g++ -std=c++14 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I $ROOT/include/libzling -I $ROOT/include -o $B/zling_demo $ROOT/tools/zling_demo.cpp \
      -L $B -lzling_amd -L $ROOT/libzling_amd -lzlng_hip -Wl,-rpath,$B -Wl,-rpath,$ROOT/libzling_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib || exit 1
  g++ -std=c++14 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I $ROOT/include/libzling -I $ROOT/include -o $B/protocol_test $ROOT/tests/cxx/protocol_test.cpp \
      -L $B -lzling_amd -L $ROOT/libzling_amd -lzlng_hip -Wl,-rpath,$B -Wl,-rpath,$ROOT/libzling_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
  ZLNG_DEMO=$B/zling_demo ZLNG_PROTOCOL_TEST=$B/protocol_test python -m pytest $ROOT/tests/test_gpu_cli.py $ROOT/tests/test_gpu_protocol.py -q -m gpu -p no:cacheprovider 2>&1 | tee $OUT/sanitize_host.log | tail -5
  ;;
host-tsan)
  # ThreadSanitizer over the host threads of the product path: the shim's helper thread (two-slot pipeline) and the group driver's copy-out threads
  python $ROOT/tests/stub_build.py --tsan || exit 1
  B=$ROOT/tests/cxx/_stub_tsan
  TSAN_OPTIONS=halt_on_error=1:exitcode=66 ZLNG_DEMO=$B/zling_demo ZLNG_PROTOCOL_TEST=$B/protocol_test python -m pytest $ROOT/tests/test_gpu_cli.py $ROOT/tests/test_gpu_protocol.py \
      -q -m gpu -p no:cacheprovider -k "not python_stream and not split_host_api" 2>&1 | tee $OUT/sanitize_host_tsan.log | tail -5
  ;;
host-cpu)
  python $ROOT/tests/stub_build.py --sanitize || exit 1
  B=$ROOT/tests/cxx/_stub_asan
  ZLNG_DEMO=$B/zling_demo ZLNG_PROTOCOL_TEST=$B/protocol_test python -m pytest $ROOT/tests/test_gpu_cli.py $ROOT/tests/test_gpu_protocol.py $ROOT/tests/test_gpu_zz_cli_error_order.py \
      -q -m gpu -p no:cacheprovider -k "not python_stream and not split_host_api" 2>&1 | tee $OUT/sanitize_host_cpu.log | tail -5
  ;;
esac
