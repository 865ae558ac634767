#!/bin/bash
# Alternating A/B of two library builds on the same box: `rounds` x (ref, new) bench.py runs; prints ms per step of each.
# Usage (inside gpurun): tools/ab_bench.sh [rounds] [bench.py args...]
# With --tree as the second argument the reference is the whole tree ab_ref/ (tools/ab_tree.sh) instead of a second library.
R=${1:-3}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF="$ROOT/robust-multimodal-contrastive-learning_amd/lib/librmcl_hip_ref.so"
REFBENCH="$ROOT/bench.py"
if [ "$1" == "--tree" ]; then shift; REF=""; REFBENCH="$ROOT/ab_ref/bench.py"; fi
for i in $(seq 1 $R); do
  a=$(RMCL_LIB=$REF python "$REFBENCH" --no-cpu-baseline --steps 30 --warmup 5 "$@" | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(python "$ROOT/bench.py" --no-cpu-baseline --steps 30 --warmup 5 "$@" | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "round $i: ref $a ms   new $b ms"
done
