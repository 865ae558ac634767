set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-feed-bench > $O/bench_trace.json 2> $O/trace.err
python tools/step_timeline.py "$O/trace/**/*_kernel_trace.csv" --md $O/lanes_step_table.md > $O/timeline.txt
python tools/dump_step.py "$O/trace/**/*_kernel_trace.csv" > $O/step_dump.txt
rm -rf $O/trace
