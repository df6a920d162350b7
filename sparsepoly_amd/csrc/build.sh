#!/bin/bash
# Builds libspfm_hip.so (gfx950) in-tree: sparsepoly_amd/lib/libspfm_hip.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# engine tag = hash of every source that goes into the library: read back through
# spfm_build_tag() so that measurements (profiles/*_traffic.json) name the code they describe
TAG="$(cat "$HERE"/*.hip "$HERE"/*.h "$HERE"/*.cpp "$HERE/../../include/spfm.h" | sha256sum | cut -c1-12)"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DSPFM_BUILD_TAG=\"$TAG\""
"$HIPCC" $FLAGS -c "$HERE/spfm_engine.hip" -o "$OUT/spfm_engine.o" &
p1=$!
"$HIPCC" $FLAGS -c "$HERE/spfm_ingest.hip" -o "$OUT/spfm_ingest.o" &
p2=$!
"$HIPCC" $FLAGS -c "$HERE/spfm_colour.hip" -o "$OUT/spfm_colour.o" &
p3=$!
wait $p1
wait $p2
wait $p3
g++ -O2 -std=c++17 -fPIC -Wall -pthread -c "$HERE/spfm_schedule.cpp" -o "$OUT/spfm_schedule.o"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libspfm_hip.so" "$OUT/spfm_engine.o" "$OUT/spfm_ingest.o" "$OUT/spfm_colour.o" "$OUT/spfm_schedule.o" -ldl -lpthread
rm -f "$OUT/spfm_engine.o" "$OUT/spfm_ingest.o" "$OUT/spfm_colour.o" "$OUT/spfm_schedule.o"
echo "built $OUT/libspfm_hip.so (engine tag $TAG)"
