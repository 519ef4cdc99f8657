#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer (and, with TSAN=1, ThreadSanitizer) over the library's HOST-ONLY units --
# csrc/mesh_host.hip (Triangle / TetGen reader and writer, face graph, permutation, partition, halo plans) and
# csrc/ordering.hip -- built alone with g++ for the CPU (GPU sanitizers are not available on this pool):
#   tools/sanitize/run.sh [work dir] [fuzz cases] [seed]
# 1. drive: the reference's 2-D mesh and a tetrahedral box through reader, orderings, permutation, RCB / slab partition;
# 2. fuzz_reader: mutated file sets (make_fuzz_files.py) -- accepted or rejected, never a crash, an overflow or a leak;
# 3. fuzz_partition: random permutations and partitions (empty ranks, more ranks than cells): what rank r sends to q is, in
#    order, what q's halo group expects; invalid permutations and rank numbers are refused.
set -eu
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
WORK=${1:-$(mktemp -d /tmp/storm_sanitize.XXXXXX)}
CASES=${2:-300}
SEED=${3:-1}
mkdir -p "$WORK"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"
[ "${TSAN:-0}" = "1" ] && SAN="-fsanitize=thread"
CXX="g++ -std=c++17 -O1 -g $SAN -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$ROOT/include -I$ROOT/stormruler_amd/csrc"
$CXX -x c++ -c "$ROOT/stormruler_amd/csrc/mesh_host.hip" -o "$WORK/mesh_host.o" &
$CXX -x c++ -c "$ROOT/stormruler_amd/csrc/ordering.hip" -o "$WORK/ordering.o" &
$CXX -c "$HERE/stubs.cpp" -o "$WORK/stubs.o" &
wait
for d in drive fuzz_reader fuzz_partition; do
  $CXX "$HERE/$d.cpp" "$WORK/stubs.o" "$WORK/mesh_host.o" "$WORK/ordering.o" -o "$WORK/$d" -lpthread
done
export STORM_HIP_BUILD_THREADS=${STORM_HIP_BUILD_THREADS:-4}
# (ThreadSanitizer: a box large enough for the threaded parse / sort / bisection paths -- 384 000 tetrahedra)
EDGE=${BOX_EDGE:-9}
[ "${TSAN:-0}" = "1" ] && EDGE=${BOX_EDGE:-40}
"$WORK/drive" "$ROOT" "$WORK" "$EDGE"
python3 "$HERE/make_fuzz_files.py" "$ROOT" "$WORK" "$SEED" "$CASES"
"$WORK/fuzz_reader" "$WORK" | tail -3
"$WORK/fuzz_partition" "$ROOT" | tail -2
echo "sanitizers: clean"
