#!/bin/bash
# A/B of two whole TREES (Python + library): extracts git revision $1 (default HEAD) into ab_ref/ (git-ignored, travels with gpurun) and
# builds its library there.  On the GPU box:  tools/ab_bench.sh 3 --tree   alternates ab_ref/bench.py and ./bench.py.
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$ROOT/ab_ref" && mkdir -p "$ROOT/ab_ref"
git -C "$ROOT" archive "$REV" bench.py rmcl_pkg.py robust-multimodal-contrastive-learning_amd include profiles/roofline_traffic.json | tar -x -C "$ROOT/ab_ref"
bash "$ROOT/ab_ref/robust-multimodal-contrastive-learning_amd/csrc/build.sh" > /dev/null 2>&1
rm -rf "$ROOT/ab_ref/robust-multimodal-contrastive-learning_amd/csrc/obj"
echo "built ab_ref/ from $REV"
