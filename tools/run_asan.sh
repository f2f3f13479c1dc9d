#!/bin/bash
# Host-logic tests of the CPU suite on the AddressSanitizer build of the library (SURVEY 5 / VERDICT r4 item 9): the tracker (Kalman,
# Hungarian, DeepSORT / SORT lifecycle), the byte tracker, the person-stream / cascade host logic and the ABI checks run their C++
# through libposepipe_hip_asan.so.  No GPU needed (the device code is untouched; nothing here launches a kernel).
#   make -C posepipeline_amd/csrc asan && bash tools/run_asan.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export POSEPIPE_LIB=$R/posepipeline_amd/libposepipe_hip_asan.so
# python itself is not instrumented: its arena allocator looks like leaks to the sanitizer
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
cd $R
LD_PRELOAD=$RT python -m pytest -q -m "not gpu" tests/test_oracle_golden.py tests/test_tracker_sort.py tests/test_bytetrack.py \
    tests/test_sort_reid.py tests/test_person_stream.py tests/test_abi.py tests/test_arch_configs.py "$@"
