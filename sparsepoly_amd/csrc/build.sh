#!/bin/bash
# Builds libspfm_hip.so (gfx950) in-tree: sparsepoly_amd/lib/libspfm_hip.so
# One translation unit per seam of the engine (spfm_engine.hip.h), compiled in parallel.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
OBJ="$OUT/obj"
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
JOBS="${SPFM_BUILD_JOBS:-$(nproc)}"
# engine tag = hash of every source that goes into the library: read back through
# spfm_build_tag() so that measurements (profiles/*_traffic.json) name the code they describe
TAG="$(cat "$HERE"/*.hip "$HERE"/*.h "$HERE"/*.cpp "$HERE/../../include/spfm.h" | sha256sum | cut -c1-12)"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DSPFM_BUILD_TAG=\"$TAG\""
# heaviest units first
UNITS="spfm_engine_prb_f32 spfm_engine_prb_f64 spfm_engine_pbprb_f32 spfm_engine_pbprb_f64 spfm_engine_pcd spfm_engine_pbcd spfm_engine_wide spfm_engine_psgd spfm_engine_core spfm_ingest spfm_colour"
# a unit is rebuilt when its source, any header or the flags changed (hash kept beside the object)
HDRHASH="$(cat "$HERE"/*.h "$HERE/../../include/spfm.h" | sha256sum | cut -c1-16)"
compile() {
    local u="$1"
    local want
    want="$(cat "$HERE/$u.hip" | sha256sum | cut -c1-16)-$HDRHASH"
    # the build tag is compiled into the core unit only
    local flags="$FLAGS"
    if [ "$u" != "spfm_engine_core" ]; then
        flags="${FLAGS/-DSPFM_BUILD_TAG=\"$TAG\"/}"
    else
        want="$want-$TAG"
    fi
    if [ -f "$OBJ/$u.o" ] && [ "$(cat "$OBJ/$u.stamp" 2>/dev/null)" = "$want" ]; then return 0; fi
    rm -f "$OBJ/$u.stamp" "$OBJ/$u.o"  # a failed compile must not leave a stale object behind
    "$HIPCC" $flags -c "$HERE/$u.hip" -o "$OBJ/$u.o"
    echo "$want" > "$OBJ/$u.stamp"
}
export -f compile
export HERE OBJ HIPCC FLAGS TAG HDRHASH
if ! printf '%s\n' $UNITS | xargs -P "$JOBS" -I{} bash -c 'compile {}'; then
    echo "build failed" >&2
    exit 1
fi
for u in $UNITS; do
    [ -f "$OBJ/$u.o" ] || { echo "build failed: $u" >&2; exit 1; }
done
if [ ! -f "$OBJ/spfm_schedule.o" ] || [ "$HERE/spfm_schedule.cpp" -nt "$OBJ/spfm_schedule.o" ]; then
    g++ -O2 -std=c++17 -fPIC -Wall -pthread -c "$HERE/spfm_schedule.cpp" -o "$OBJ/spfm_schedule.o"
fi
OBJS=""
for u in $UNITS; do OBJS="$OBJS $OBJ/$u.o"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o "$OUT/libspfm_hip.so" $OBJS "$OBJ/spfm_schedule.o" -ldl -lpthread
echo "built $OUT/libspfm_hip.so (engine tag $TAG)"
