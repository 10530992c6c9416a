#!/bin/bash
# Builds oracle/_ref/libblubref.so: the reference's own compute shaders (read from the reference checkout where it
# lies, never copied into the repository), turned into C++ by glsl2cpp.py and compiled against glsl_shim.h.
# TEST INFRASTRUCTURE.  Usage: oracle/glsl/build_ref.sh [reference root, default /root/reference]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${1:-/root/reference}"
SHADERS="$REF/shader"
OUT="$HERE/../_ref"
if [ ! -d "$SHADERS/simulation" ]; then echo "build_ref.sh: $SHADERS/simulation not found (the reference is absent: nothing to build)"; exit 3; fi
mkdir -p "$OUT/gen" "$OUT/obj"
CXX="${CXX:-g++}"
FLAGS="-std=c++17 -O2 -fPIC -ffp-contract=off -fno-fast-math -fno-strict-aliasing -Wno-attributes -I$HERE"
objs=()
for f in "$SHADERS"/simulation/*.comp "$SHADERS"/simulation/pressure_solver/*.comp; do
  rel="${f#$SHADERS/}"
  name="$(basename "$f" .comp)"
  [ "$name" = "pressure_reduce" ] && continue   # a header for pressure_reduce_sum / _max (no #version of its own)
  python3 "$HERE/glsl2cpp.py" "$SHADERS" "$rel" "$OUT/gen/$name.cpp"
  $CXX $FLAGS -c "$OUT/gen/$name.cpp" -o "$OUT/obj/$name.o" &
  objs+=("$OUT/obj/$name.o")
done
$CXX $FLAGS -c "$HERE/ref_runtime.cpp" -o "$OUT/obj/ref_runtime.o" &
wait
$CXX -shared -o "$OUT/libblubref.so" "$OUT/obj/ref_runtime.o" "${objs[@]}"
( cd "$SHADERS" && find simulation -type f | sort | xargs sha256sum ) > "$OUT/shader_sha256.txt"
echo "built $OUT/libblubref.so from $(ls "$OUT"/gen/*.cpp | wc -l) shaders"
