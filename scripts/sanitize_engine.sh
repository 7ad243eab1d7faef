#!/bin/bash
# Builds the engine's host code under AddressSanitizer (+ leak detection) and ThreadSanitizer: the product's sources compiled with
# -Xarch_host -fsanitize=... (device code objects unchanged) and linked with tests/host_emul/engine_sanitize_main.cpp into one
# executable per sanitizer: curdleproofs_amd/_lib/engine_asan, engine_tsan (they travel to the GPU box with the snapshot;
# tests/test_sanitizers.py runs them there).  hipcc cross-compiles without a GPU.   Usage: scripts/sanitize_engine.sh [asan|tsan]...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/curdleproofs_amd/csrc
O=$R/curdleproofs_amd/_lib
mkdir -p $O
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SRCS="kernels.hip late.hip protocol.hip round.hip engine.cpp engine_device.cpp whisk.cpp capi.cpp"
build_one() {   # name, sanitizer
  local W=$O/san_$1.tmp
  rm -rf $W && mkdir -p $W
  local FL="-O1 -g1 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -Xarch_host -march=x86-64-v3 -Xarch_host -madx -Xarch_host -fsanitize=$2 -Xarch_host -fno-omit-frame-pointer -Xarch_device -O3"
  local pids=""
  for s in $SRCS; do
    $HIPCC $FL -x hip -c $C/$s -o $W/$s.o > $W/$s.log 2>&1 &
    pids="$pids $!"
  done
  $HIPCC $FL -I $R/include -x hip -c $R/tests/host_emul/engine_sanitize_main.cpp -o $W/main.o > $W/main.log 2>&1 &
  pids="$pids $!"
  for p in $pids; do wait $p || { cat $W/*.log | grep -v warning | tail -20; echo "sanitize_engine: compile failed ($1)"; exit 1; }; done
  $HIPCC --offload-arch=gfx950 -fsanitize=$2 $W/*.o -o $O/engine_$1 -lpthread > $W/link.log 2>&1 || { tail -20 $W/link.log; exit 1; }
  rm -rf $W
  echo $O/engine_$1
}
for which in ${@:-asan tsan}; do
  case $which in
    asan) build_one asan address ;;
    tsan) build_one tsan thread ;;
  esac
done
