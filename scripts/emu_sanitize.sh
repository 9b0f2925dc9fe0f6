#!/bin/bash
# Build the kernel simulator (tests/emu: every HIP kernel compiled for the host, device memory = heap blocks) with
# AddressSanitizer or UndefinedBehaviorSanitizer and run the CPU suite against it: out-of-bounds accesses of device buffers
# by kernels, and of host tables by the bookkeeping, stop the run.
#   scripts/emu_sanitize.sh address|undefined [pytest args...]
set -e
KIND=${1:-address}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CL=/opt/rocm/lib/llvm/bin/clang++
OUT=/tmp/emu_$KIND
mkdir -p $OUT
SRC=$ROOT/holoagent_amd/csrc
FLAGS="-fsanitize=$KIND -fno-omit-frame-pointer"
[ $KIND = address ] && FLAGS="$FLAGS -shared-libasan"
[ $KIND = undefined ] && FLAGS="$FLAGS -fno-sanitize-recover=undefined -fno-sanitize=vptr,function"
for f in $SRC/*.hip; do
  b=$(basename $f .hip)
  ( $CL -x c++ -O1 -g -std=c++17 -fPIC -ffp-contract=off -mf16c -mavx2 $FLAGS -I$ROOT/tests/emu/include -Wno-unused-value -c $f -o $OUT/$b.o ) &
done
wait
$CL -shared -fPIC $FLAGS -o $OUT/libhmsg_emu.so $OUT/*.o
RT=$(dirname $($CL -print-libgcc-file-name))
export HMSG_EMU_PATH=$OUT/libhmsg_emu.so
if [ $KIND = address ]; then
  export LD_PRELOAD=$RT/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 HMSG_DEBUG_EXACT_ALLOC=1
else
  export LD_PRELOAD=$RT/libclang_rt.ubsan_standalone-x86_64.so UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
fi
cd $ROOT
# (test files / pytest options may follow the kind; without any: the whole CPU suite)
TARGETS="tests/"
for a in "$@"; do case "$a" in tests/*) TARGETS="";; esac; done
exec python -m pytest $TARGETS -q -m "not gpu" -p no:cacheprovider "$@"
