#!/bin/bash
# Host-side sanitizer build (SURVEY section 5 "Race detection / sanitizers"): the two pure-host translation units (encoder.cpp: pass
# orchestration, workspace / stash carving, event protocol; api.cpp: the C ABI, argument checks, heads) compiled with AddressSanitizer +
# UndefinedBehaviorSanitizer (host pass only: GPU sanitizers are not available on this pool) and linked with the normal kernel objects
# into lib/librmcl_hip_asan.so.  tests/test_host_sanitized.py drives the entry points that need no GPU through it (layout and size
# queries, routing, every argument-validation path) in a child process with the ASan runtime preloaded.  Run csrc/build.sh first.
set -e
cd "$(dirname "$0")"
OUT=../lib
[ -f obj/gemm_st.o ] || { echo "build_host_asan.sh: run build.sh first (kernel objects missing)" >&2; exit 1; }
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
mkdir -p obj/asan
for f in encoder.cpp api.cpp; do
  hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wall -Wno-unused-function --cuda-host-only $SAN -x hip -c "$f" -o obj/asan/${f%.*}.o
done
objs=$(grep '^SRCS=' build.sh | sed 's/SRCS="//; s/"//' | tr ' ' '\n' | grep -v -e encoder.cpp -e api.cpp | sed 's/\.[a-z]*$/.o/; s/^/obj\//')
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan $objs obj/asan/encoder.o obj/asan/api.o -o "$OUT/librmcl_hip_asan.so"
echo "built $OUT/librmcl_hip_asan.so"
