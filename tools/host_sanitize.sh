#!/bin/bash
# AddressSanitizer + UBSan and ThreadSanitizer runs of the native host side (magellanmapper_amd/csrc/mmx_host.cpp):
# CPU only -- the GPU pool offers no sanitizers.  Builds libmmx_asan.so / libmmx_tsan.so (the stock device objects +
# an instrumented mmx_host.o), preloads the matching runtime into Python and runs the CPU tests that drive the host
# entry points (tests/test_host_logic.py, tests/test_dist_gloo.py) through MMX_LIB_PATH.
#   tools/host_sanitize.sh [asan|tsan|both] [log directory] [round tag]     (default: both, profiles/, r06)
# Logs: <dir>/<tag>_host_asan.log, <dir>/<tag>_host_tsan.log; exit status 0 only when pytest passed and no report
# mentions mmx_host.cpp / libmmx (reports from inside CPython / NumPy / torch, which are not instrumented, are
# counted separately and listed).
set -u
cd "$(dirname "$0")/.."
which=${1:-both}
out=${2:-profiles}
tag=${3:-r06}
mkdir -p "$out"
make -s -C magellanmapper_amd/csrc -j8 all || exit 1
status=0
run() {   # name, runtime libraries, environment, tests
    local name=$1 libs=$2 envs=$3 tests=$4
    make -s -C magellanmapper_amd/csrc host-$name || exit 1
    local log="$out/${tag}_host_$name.log"
    echo "# tools/host_sanitize.sh $name: $(date -u +%FT%TZ), $(gcc --version | head -1)" > "$log"
    echo "# LD_PRELOAD=$libs $envs MMX_LIB_PATH=magellanmapper_amd/libmmx_$name.so python -m pytest $tests -q -m 'not gpu' -p no:cacheprovider" >> "$log"
    timeout 1800 env LD_PRELOAD="$libs" $envs MMX_LIB_PATH="$PWD/magellanmapper_amd/libmmx_$name.so" \
        python -m pytest $tests -q -m "not gpu" -p no:cacheprovider --timeout=300 >> "$log" 2>&1
    local rc=$?
    local ours
    ours=$(grep -c -E "mmx_host\.cpp|libmmx_$name" "$log" | head -1)
    # (the command line above names the library once)
    ours=$((ours - 1))
    local reports
    reports=$(grep -c -E "^(==[0-9]+==ERROR|WARNING: ThreadSanitizer|.*runtime error:)" "$log")
    echo "# pytest exit status $rc; sanitizer reports $reports; lines naming mmx_host.cpp / libmmx_$name.so: $ours" >> "$log"
    tail -3 "$log"
    if [ $rc -ne 0 ] || [ "$ours" -gt 0 ]; then status=1; fi
}
asan_rt="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
tsan_rt="$(gcc -print-file-name=libtsan.so)"
if [ "$which" = asan ] || [ "$which" = both ]; then
    run asan "$asan_rt" "ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1" \
        "tests/test_host_logic.py tests/test_dist_gloo.py tests/test_oracle_golden.py"
fi
if [ "$which" = tsan ] || [ "$which" = both ]; then
    # (gloo ranks are separate processes: what TSan can see is the pool inside one process -- test_host_logic drives
    #  every parallel section; MMX_HOST_THREADS=8 makes sure the sections really fan out on a small container and
    #  MMX_HOST_SPIN_US=150 that the workers poll before they sleep as on a machine with cores to spare.
    #  TSan cannot follow fork() from a process that already has threads -- it deadlocks in the child -- so the
    #  tests that fork are left to the ASan run)
    run tsan "$tsan_rt" "TSAN_OPTIONS=report_signal_unsafe=0:history_size=4 MMX_HOST_THREADS=8 MMX_HOST_SPIN_US=150" \
        "tests/test_host_logic.py --deselect tests/test_host_logic.py::test_native_prune_works_in_a_forked_child --deselect tests/test_host_logic.py::test_get_mp_pool_follows_config --deselect tests/test_host_logic.py::test_region_threads_work_in_a_forked_child"
fi
exit $status
