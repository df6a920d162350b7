#!/bin/bash
# Builds libspfm_hip.so (gfx950) in-tree: sparsepoly_amd/lib/libspfm_hip.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
"$HIPCC" $FLAGS -c "$HERE/spfm_engine.hip" -o "$OUT/spfm_engine.o"
g++ -O2 -std=c++17 -fPIC -Wall -pthread -c "$HERE/spfm_schedule.cpp" -o "$OUT/spfm_schedule.o"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libspfm_hip.so" "$OUT/spfm_engine.o" "$OUT/spfm_schedule.o" -ldl -lpthread
rm -f "$OUT/spfm_engine.o" "$OUT/spfm_schedule.o"
echo "built $OUT/libspfm_hip.so"
