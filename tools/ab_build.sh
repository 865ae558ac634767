#!/bin/bash
# A/B helper: builds the library of git revision $1 (default HEAD) next to the working-tree build as lib/librmcl_hip_ref.so,
# so that one gpurun call can time both on the same box:  RMCL_LIB=$PWD/robust-*/lib/librmcl_hip_ref.so python bench.py ...
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG="$ROOT/robust-multimodal-contrastive-learning_amd"
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" robust-multimodal-contrastive-learning_amd/csrc include | tar -x -C "$TMP"
bash "$TMP/robust-multimodal-contrastive-learning_amd/csrc/build.sh" > /dev/null 2>&1
cp "$TMP/robust-multimodal-contrastive-learning_amd/lib/librmcl_hip.so" "$PKG/lib/librmcl_hip_ref.so"
rm -rf "$TMP"
echo "built $PKG/lib/librmcl_hip_ref.so from $REV"
