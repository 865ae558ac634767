#!/bin/bash
# Alternating A/B of rmcl_tune_set settings on ONE box and ONE build: `rounds` x (base, variant 1, variant 2, ...) bench.py runs; prints ms per step.
# Usage (inside gpurun): tools/ab_tune.sh rounds "label=key:value,key:value" ["label2=..."] [-- bench.py args]
# The base run has no RMCL_BENCH_TUNE; every variant is given to bench.py through that variable (bench.py applies it with rmcl_tune_set).
R=$1; shift
VARS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done
[ "$1" == "--" ] && shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
run() { RMCL_BENCH_TUNE="$1" python "$ROOT/bench.py" --no-cpu-baseline --no-feed-bench --no-realistic --steps 30 --warmup 5 "${@:2}" | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in $(seq 1 $R); do
  line="round $i: base $(run "" "$@")"
  for v in "${VARS[@]}"; do line="$line | ${v%%=*} $(run "${v#*=}" "$@")"; done
  echo "$line"
done
