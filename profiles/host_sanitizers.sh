#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the host side of the stage (sedef_amd/csrc/host/*.cc, g++): the host
# library is rebuilt with -fsanitize=address,undefined in place, the CPU tests that drive it are run with the sanitizer
# runtime preloaded into python (libstdc++ along with it, so that __cxa_throw is intercepted), and the library is rebuilt
# as it was.  Run in the build container (no GPU needed): bash profiles/host_sanitizers.sh
# End of round 3: test_host_pipeline, test_pinning, test_stats_generate, test_chain_oracle, test_stage_scale (the whole
# host pipeline on the reference kernel, 52 Mb and chr1-sized genomes): 30 passed, no report.
cd "$(dirname "$0")/.." || exit 1
SRCS=$(python3 -c "import sedef_amd.host as h; print(' '.join(h.HOST_SRC + '/' + f for f in h._SOURCES))")
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -Wall -shared -o sedef_amd/lib/libsedef_host.so \
    $SRCS -Lsedef_amd/lib -lsedef_hip -lpthread '-Wl,-rpath,$ORIGIN' || exit 1
ASAN=$(g++ -print-file-name=libasan.so); STD=$(g++ -print-file-name=libstdc++.so)
rm -f /tmp/sdf_asan.* /tmp/sdf_ubsan.*
LD_PRELOAD="$ASAN $STD" ASAN_OPTIONS=detect_leaks=0:log_path=/tmp/sdf_asan UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/sdf_ubsan \
  python3 -m pytest tests/test_host_pipeline.py tests/test_pinning.py tests/test_stats_generate.py tests/test_chain_oracle.py \
  tests/test_stage_scale.py -q -m "not gpu"
rc=$?
ls /tmp/sdf_asan.* /tmp/sdf_ubsan.* 2>/dev/null && echo "sanitizer reports above" || echo "no sanitizer report"
python3 -c "from sedef_amd.host import build_host; build_host(force=True)"
# ... and the host side of the HIP library itself (batch cut and chunk planner, sdf_plan.hip; device code not instrumented):
# the planner tests through the sdf_debug_plan hook -- one- and two-pass cut, planning on several threads.  End of round 3:
# 9 passed, no report.
cp sedef_amd/lib/libsedef_hip.so /tmp/sdf_libsedef_hip.keep
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -shared -fsanitize=address,undefined -fno-gpu-sanitize \
    -fno-omit-frame-pointer -Wno-unused-function -o sedef_amd/lib/libsedef_hip.so sedef_amd/csrc/sdf_unity.hip || { cp /tmp/sdf_libsedef_hip.keep sedef_amd/lib/libsedef_hip.so; exit 1; }
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD="$RT $STD" ASAN_OPTIONS=detect_leaks=0:log_path=/tmp/sdf_asan UBSAN_OPTIONS=print_stacktrace=1:log_path=/tmp/sdf_ubsan \
  python3 -m pytest tests/test_planner.py tests/test_cabi_exports.py -q
rc2=$?
ls /tmp/sdf_asan.* /tmp/sdf_ubsan.* 2>/dev/null && echo "sanitizer reports above" || echo "no sanitizer report"
# ... and ThreadSanitizer over the same planner tests (the cut's scan threads, the two passes with the early start next to
# the second, the chunk planner's workers).  End of round 3: 5 passed, no report.
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -shared -fsanitize=thread -fno-gpu-sanitize \
    -fno-omit-frame-pointer -Wno-unused-function -o sedef_amd/lib/libsedef_hip.so sedef_amd/csrc/sdf_unity.hip || { cp /tmp/sdf_libsedef_hip.keep sedef_amd/lib/libsedef_hip.so; exit 1; }
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
rm -f /tmp/sdf_tsan.*
LD_PRELOAD="$RT" TSAN_OPTIONS="log_path=/tmp/sdf_tsan report_signal_unsafe=0" python3 -m pytest tests/test_planner.py -q
rc3=$?
ls /tmp/sdf_tsan.* 2>/dev/null && echo "thread sanitizer reports above" || echo "no thread sanitizer report"
cp /tmp/sdf_libsedef_hip.keep sedef_amd/lib/libsedef_hip.so
exit $((rc | rc2 | rc3))
