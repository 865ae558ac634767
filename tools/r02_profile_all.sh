set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_trace.json 2> $O/trace.err
python tools/step_timeline.py "$O/trace/**/*_kernel_trace.csv" --md $O/r02_step_table.md > /dev/null
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/r02_step_kernel_stats.csv
python tools/dump_step.py "$O/trace/**/*_kernel_trace.csv" > $O/step_dump.txt
rm -rf $O/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/traceb -- python3 bench.py --config barlowtwins --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_barlow.json 2> $O/traceb.err
cp $(ls $O/traceb/*/*kernel_stats.csv | head -1) $O/r02_barlowtwins_kernel_stats.csv
rm -rf $O/traceb
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_sq.err
python tools/pmc_sq.py $O/pmc_sq --out $O/r02_pmc_sq_summary.csv > $O/pmc_sq.txt
rm -rf $O/pmc_sq
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_w.err
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write --out $O/roofline_traffic.json > $O/pmc_traffic.txt
rm -rf $O/pmc_fetch $O/pmc_write
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --config itm_clean --no-cpu-baseline > $O/bench_itm_clean.json 2>/dev/null
python bench.py --config full_rmcl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_full_rmcl.json 2>/dev/null
python bench.py --config barlowtwins --no-cpu-baseline > $O/bench_barlowtwins.json 2>/dev/null
tail -c 400 $O/bench_default.json
