#!/bin/bash
# CPU-only sanitizer run of ALL host code of the product (GPU ASan is not available on this pool):
#   * bsq_host.cpp (staging ring, host entry points), bsq_alphabet.cpp (LUT builder, descriptors), bsq_fastx.cpp (the streaming
#     FASTA / FASTQ / gzip parser -- untrusted text) are rebuilt with g++ -fsanitize=address,undefined and linked with the hipcc-built
#     kernel objects (exactly build.py's LIB_SRCS, uninstrumented) into a scratch libbsq_hip.so;
#   * cbioseq_module.cpp (pybind11 layer) is rebuilt the same way against it;
#   * the host-only test modules run against the scratch copy: surface + error paths, FlatFile / FASTX differentials against the
#     compiled reference (400 adversarial texts, 16 383 ... 131 072-byte lines, truncated gzip members), ABI + alphabets.
# Without a device every compute entry point stops at BSQ_ERR_NO_DEVICE, so the staging ring itself is exercised by
# tests/native/host_ring_stress.cpp (ThreadSanitizer, a mock HIP runtime) instead -- tests/test_sanitizers.py runs both.
# Usage: scripts/asan_host.sh
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
trap 'rm -rf "$W"' EXIT
cp -r "$REPO/bioseq_amd" "$REPO/tests" "$REPO/oracle" "$REPO/include" "$REPO/bench.py" "$REPO/__graft_entry__.py" "$W/"
# always: the build is incremental, and objects of an older tree must never be linked against fresh host code
python3 "$REPO/bioseq_amd/build.py" > /dev/null
KOBJS=$(python3 "$REPO/bioseq_amd/build.py" --kernel-objects)
SAN="-O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
for f in bsq_host bsq_alphabet bsq_fastx; do
  g++ $SAN -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$REPO/include" -I"$REPO/bioseq_amd/csrc" -c "$REPO/bioseq_amd/csrc/$f.cpp" -o "$W/$f.o" &
done
wait
g++ -shared -fPIC -fsanitize=address,undefined -pthread -o "$W/bioseq_amd/libbsq_hip.so" "$W"/bsq_host.o "$W"/bsq_alphabet.o "$W"/bsq_fastx.o \
    $KOBJS -L/opt/rocm/lib -lamdhip64 -lz -Wl,-rpath,/opt/rocm/lib
g++ $SAN -shared -fvisibility=hidden \
    -I"$REPO/include" -I"$(python3 -c 'import pybind11;print(pybind11.get_include())')" \
    -I"$(python3 -c 'import sysconfig;print(sysconfig.get_paths()["include"])')" \
    -o "$W/bioseq_amd/cbioseq$(python3 -c 'import sysconfig;print(sysconfig.get_config_var("EXT_SUFFIX"))')" \
    "$REPO/bioseq_amd/csrc/cbioseq_module.cpp" -L"$W/bioseq_amd" -lbsq_hip -Wl,-rpath,"$W/bioseq_amd"
cd "$W"
touch DESIGN.md INTEGRATION.md
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
    python3 -m pytest tests/test_host_surface.py tests/test_flatfile.py tests/test_dropin_surface.py tests/test_abi_and_layout.py tests/test_oracle_golden.py \
        -q -m "not gpu" -p no:cacheprovider -x --deselect tests/test_abi_and_layout.py::test_repo_layout
echo "sanitizer run clean"
