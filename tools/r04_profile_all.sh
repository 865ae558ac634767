# Round-4 profiles of the FINAL build, two GPU calls (each inside gpurun's limit):
#   tools/r04_profile_all.sh a   rocprofv3 kernel traces (one-chain step, default step with the lanes, training-realistic step) + the PMC passes
#   tools/r04_profile_all.sh b   un-profiled benches of every configuration + the microbenchmarks that the round's decisions rest on
# Outputs under gpurun_out/r04p/; the summaries to keep are copied to profiles/ by hand (profiles/README.md lists them).
# Every PMC pass runs with RMCL_LANES=0 (ONE chain): with the half-batch lanes two half-size launches co-run and a per-kernel counter divided by a
# whole-chip GRBM_GUI_ACTIVE is not a per-kernel number (round-3 review, "what's weak" 2) - the lanes get a kernel TRACE only, labelled as such.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-feed-bench --no-realistic"
if [ "${1:-a}" == "a" ]; then
RMCL_LANES=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 3 --warmup 1 > $O/bench_trace_onechain.json 2> $O/trace.err
python tools/step_timeline.py "$O/trace/**/*_kernel_trace.csv" --md $O/r04_step_table.md > /dev/null
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/r04_step_kernel_stats.csv
rm -rf $O/trace
echo "one-chain trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tracel -- $B --steps 3 --warmup 1 > $O/bench_trace_lanes.json 2> $O/tracel.err
python tools/step_timeline.py "$O/tracel/**/*_kernel_trace.csv" --md $O/r04_step_table_lanes.md > /dev/null
rm -rf $O/tracel
echo "lanes trace done"
RMCL_LANES=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/traced -- $B --drop-rate 0.1 --steps 3 --warmup 1 > $O/bench_trace_dropout.json 2> $O/traced.err
python tools/step_timeline.py "$O/traced/**/*_kernel_trace.csv" --md $O/r04_step_table_dropout.md > /dev/null
rm -rf $O/traced
echo "dropout trace done"
RMCL_LANES=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- $B --steps 2 --warmup 1 > /dev/null 2> $O/pmc_sq.err
python tools/pmc_sq.py $O/pmc_sq --out $O/r04_pmc_sq_summary.csv > $O/pmc_sq.txt
rm -rf $O/pmc_sq
echo "pmc sq done"
RMCL_LANES=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B --steps 2 --warmup 1 > /dev/null 2> $O/pmc_f.err
RMCL_LANES=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B --steps 2 --warmup 1 > /dev/null 2> $O/pmc_w.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write --out $O/roofline_traffic.json > $O/pmc_traffic.txt
rm -rf $O/pmc_fetch $O/pmc_write
echo "pmc traffic done"
tail -c 300 $O/bench_trace_onechain.json
else
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default bench done"
python tools/chain_bench.py > $O/r04_chain_bench.txt 2>&1 || true
python tools/infonce_bench.py > $O/r04_infonce_bench.txt 2>&1 || true
python bench.py --drop-rate 0.1 --no-cpu-baseline --no-feed-bench > $O/bench_dropout.json 2>/dev/null || true
RMCL_LANES=0 python bench.py --no-cpu-baseline --no-feed-bench --no-realistic > $O/bench_onechain.json 2>/dev/null || true
python bench.py --config itm_clean --no-cpu-baseline --no-feed-bench > $O/bench_itm_clean.json 2>/dev/null
python bench.py --config full_rmcl --steps 10 --warmup 2 --no-cpu-baseline --no-feed-bench > $O/bench_full_rmcl.json 2>/dev/null
python bench.py --config barlowtwins --no-cpu-baseline --no-feed-bench > $O/bench_barlowtwins.json 2>/dev/null
tail -c 600 $O/bench_default.json
fi
