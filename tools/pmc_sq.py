#!/usr/bin/env python3
"""MFMA-busy and wait fractions of the step's kernel classes from one rocprofv3 PMC pass (SQ + GRBM counters only).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace \
        --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_sq.py gpurun_out/pmc_sq --out profiles/r02_pmc_sq_summary.csv

MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) as in round 1's summary (the SQ counter sums
over all SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs)."""
import argparse
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--out", default="profiles/r02_pmc_sq_summary.csv")
    args = ap.parse_args()
    files = glob.glob(os.path.join(args.dir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit("no counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("unsigned short", "bf16")
            if not ("gemm_" in name or "attn_" in name or "ln_" in name or "infonce" in name):
                continue
            a = acc[name][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    rows = []
    for name, cs in acc.items():
        m = {k: v[1] / v[0] for k, v in cs.items()}
        n = next(iter(cs.values()))[0]
        gui = m.get("GRBM_GUI_ACTIVE", 0.0)
        mfma = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        wave = m.get("SQ_WAVE_CYCLES", 0.0)
        rows.append((name, n, mfma / (gui / 8 * 1024) if gui else 0.0, m.get("SQ_WAIT_INST_ANY", 0.0) / wave if wave else 0.0,
                     m.get("SQ_WAIT_ANY", 0.0) / wave if wave else 0.0, m.get("SQ_ACTIVE_INST_ANY", 0.0) / wave if wave else 0.0, gui))
    rows.sort(key=lambda r: -r[1] * r[6])
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        f.write("kernel,dispatches,mfma_busy_frac,wait_inst_any_over_wave_cycles,wait_any_over_wave_cycles,active_inst_over_wave_cycles,mean_GRBM_GUI_ACTIVE\n")
        for r in rows:
            f.write(f"\"{r[0]}\",{r[1]},{r[2]:.4f},{r[3]:.4f},{r[4]:.4f},{r[5]:.4f},{r[6]:.0f}\n")
    for r in rows[:16]:
        print(f"{r[0][:64]:64s} n={r[1]:4d} mfma_busy {r[2]:.3f} wait_inst {r[3]:.3f} wait_any {r[4]:.3f} active {r[5]:.3f}")


if __name__ == "__main__":
    main()
