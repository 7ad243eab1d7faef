#!/bin/sh
# Sanitizer builds of the product's host-side headers (see sanitize_main.cpp).  Usage: tests/host_emul/sanitize.sh [outdir]
# Exit code 0 = both builds ran clean (no sanitizer report, all checks passed).
set -e
HERE=$(dirname "$0")
OUT=${1:-/tmp/cpx_sanitize}
mkdir -p "$OUT"
# (-g1: line numbers in reports without the variable-tracking pass, which takes minutes on the unrolled field code; both builds at once)
g++ -std=c++17 -O1 -g1 -fno-omit-frame-pointer -pthread -fsanitize=thread -o "$OUT/san_tsan" "$HERE/sanitize_main.cpp" &
g++ -std=c++17 -O1 -g1 -fno-omit-frame-pointer -pthread -fsanitize=address,undefined -fno-sanitize-recover=all -o "$OUT/san_asan_ubsan" "$HERE/sanitize_main.cpp"
wait
echo "== ThreadSanitizer"
TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1" "$OUT/san_tsan"
echo "== AddressSanitizer + UndefinedBehaviorSanitizer"
ASAN_OPTIONS="detect_leaks=1" UBSAN_OPTIONS="print_stacktrace=1" "$OUT/san_asan_ubsan"
echo "== clean"
