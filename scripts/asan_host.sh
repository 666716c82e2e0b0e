#!/bin/bash
# CPU-only sanitizer run of the pybind11 host layer (GPU ASan is not available on this pool): builds
# cbioseq_module.cpp with -fsanitize=address,undefined into a scratch copy of the package and runs the
# host-only test modules against it.  Usage: scripts/asan_host.sh
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
cp -r "$REPO/bioseq_amd" "$REPO/tests" "$REPO/oracle" "$REPO/include" "$W/"
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -fvisibility=hidden \
    -I"$REPO/include" -I"$(python3 -c 'import pybind11;print(pybind11.get_include())')" \
    -I"$(python3 -c 'import sysconfig;print(sysconfig.get_paths()["include"])')" \
    -o "$W/bioseq_amd/cbioseq$(python3 -c 'import sysconfig;print(sysconfig.get_config_var("EXT_SUFFIX"))')" \
    "$REPO/bioseq_amd/csrc/cbioseq_module.cpp" -L"$REPO/bioseq_amd" -lbsq_hip -Wl,-rpath,"$REPO/bioseq_amd"
cd "$W"
touch DESIGN.md INTEGRATION.md
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
    python3 -m pytest tests/test_host_surface.py tests/test_flatfile.py -q -m "not gpu" -p no:cacheprovider
echo "sanitizer run clean ($W)"
