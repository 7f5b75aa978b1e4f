#!/bin/bash
# CPU-side AddressSanitizer + UBSan pass (GPU sanitizers are not available on the pool): the oracle (gcc) and the
# library's host code (trm_capi.cc, trm_setup.cc, trm_io.cc; clang runtime of the ROCm toolchain) are rebuilt
# instrumented into /tmp and the CPU test suite runs against them.  Leaves the tree as it was.
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
trap 'cp $T/oracle_orig.so oracle/libtrm_oracle.so; touch oracle/libtrm_oracle.so; rm -rf $T' EXIT
make -s -C oracle libtrm_oracle.so && make -s -C gnuspeech_amd/csrc
cp oracle/libtrm_oracle.so $T/oracle_orig.so
gcc -O1 -g -ffp-contract=off -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -std=c99 -D_XOPEN_SOURCE=600 \
    -shared -o oracle/libtrm_oracle.so oracle/trm_oracle.c oracle/evt_oracle.c -lm
echo "== oracle under gcc ASan+UBSan"
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
cp $T/oracle_orig.so oracle/libtrm_oracle.so; touch oracle/libtrm_oracle.so
cd gnuspeech_amd/csrc
for f in trm_capi trm_setup trm_io; do
    hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -c $f.cc -o $T/$f.o
done
hipcc --offload-arch=gfx950 -shared -fsanitize=address,undefined -o $T/libtrm_hip_san.so $T/trm_capi.o $T/trm_setup.o $T/trm_io.o \
    build/trm_kernels.o build/trm_quad.o build/trm_oct.o build/trm_tracks.o
cd ../..
echo "== library host code under clang ASan+UBSan"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
TRM_LIB=$T/libtrm_hip_san.so ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD="$RT" \
    python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
