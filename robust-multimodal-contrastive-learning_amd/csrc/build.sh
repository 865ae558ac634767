#!/bin/bash
# Builds librmcl_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [-j N]
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=fast $RMCL_EXTRA_FLAGS"   # (developer builds: e.g. RMCL_EXTRA_FLAGS=-DST_TRACE, tools/st_trace.py)
SRCS="gemm_exact.hip gemm_fast.hip gemm_big.hip gemm_pp.hip gemm_st.hip gemm_sw.hip gemm_dp.hip norm_softmax.hip embed_misc.hip infonce.hip ipot.hip itm.hip attention.hip barlow.hip encoder.cpp api.cpp"
pids=()
objs=()
for f in $SRCS; do
  [ -f "$f" ] || { echo "build.sh: missing source $f" >&2; exit 1; }
  o=obj/${f%.*}.o
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer "$o")" ] || [ ../../include/rmcl.h -nt "$o" ]; then
    ( hipcc $FLAGS -x hip -c "$f" -o "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p || { echo "build.sh: compile failed" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT/librmcl_hip.so"
echo "built $OUT/librmcl_hip.so"
