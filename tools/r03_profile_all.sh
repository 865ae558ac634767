# Round-3 profiles in two GPU calls (each well inside gpurun's limit):  tools/r03_profile_all.sh a  (rocprofv3 kernel trace + the three PMC
# passes of the step) and  tools/r03_profile_all.sh b  (un-profiled benches of every configuration + the microbenchmarks); outputs under
# gpurun_out/r03p/, the summaries to keep are copied to profiles/ by hand.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p
mkdir -p $O
if [ "${1:-a}" == "a" ]; then
# the one-chain step (RMCL_LANES=0: comparable with round 2's table), then the default step (half-batch lanes: 1378 dispatches, host-bound under the profiler)
RMCL_LANES=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-feed-bench > $O/bench_trace.json 2> $O/trace.err
python tools/step_timeline.py "$O/trace/**/*_kernel_trace.csv" --md $O/r03_step_table.md > /dev/null
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/r03_step_kernel_stats.csv
python tools/dump_step.py "$O/trace/**/*_kernel_trace.csv" > $O/step_dump.txt
rm -rf $O/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tracel -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-feed-bench > $O/bench_trace_lanes.json 2> $O/tracel.err
python tools/step_timeline.py "$O/tracel/**/*_kernel_trace.csv" --md $O/r03_step_table_lanes.md > /dev/null
cp $(ls $O/tracel/*/*kernel_stats.csv | head -1) $O/r03_step_kernel_stats_lanes.csv
rm -rf $O/tracel
rocprofv3 --kernel-trace --stats --output-format csv -d $O/traceb -- python3 bench.py --config barlowtwins --steps 3 --warmup 1 --no-cpu-baseline --no-feed-bench > $O/bench_barlow.json 2> $O/traceb.err
cp $(ls $O/traceb/*/*kernel_stats.csv | head -1) $O/r03_barlowtwins_kernel_stats.csv
rm -rf $O/traceb
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-feed-bench > /dev/null 2> $O/pmc_sq.err
python tools/pmc_sq.py $O/pmc_sq --out $O/r03_pmc_sq_summary.csv > $O/pmc_sq.txt
rm -rf $O/pmc_sq
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-feed-bench > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-feed-bench > /dev/null 2> $O/pmc_w.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write --out $O/roofline_traffic.json > $O/pmc_traffic.txt
rm -rf $O/pmc_fetch $O/pmc_write
else
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python tools/gemm_vs_blaslt.py > $O/r03_gemm_vs_blaslt.txt 2>&1 || true
python tools/host_overhead.py > $O/r03_host_overhead.txt 2>&1 || true
python tools/dp_bench.py -1,80 > $O/r03_dp_bench_final.txt 2>&1 || true
python tools/infonce_bench.py > $O/r03_infonce_bench.txt 2>&1 || true
python bench.py --padded-images --no-cpu-baseline --no-feed-bench > $O/bench_padded_images.json 2>/dev/null || true
python bench.py --config itm_clean --no-cpu-baseline --no-feed-bench > $O/bench_itm_clean.json 2>/dev/null
python bench.py --config full_rmcl --steps 10 --warmup 2 --no-cpu-baseline --no-feed-bench > $O/bench_full_rmcl.json 2>/dev/null
python bench.py --config barlowtwins --no-cpu-baseline --no-feed-bench > $O/bench_barlowtwins.json 2>/dev/null
fi
tail -c 400 $O/bench_default.json 2>/dev/null || tail -c 300 $O/bench_trace.json
