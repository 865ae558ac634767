#!/usr/bin/env python3
"""LDS utilisation of the step's kernel classes from one rocprofv3 PMC pass.

    rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace \
        --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-feed-bench
    python tools/pmc_lds.py gpurun_out/pmc_lds --out profiles/r03_pmc_lds_summary.csv

lds_active = SQ_LDS_IDX_ACTIVE (LDS-array cycles, summed over the CUs) / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs): the fraction of the launch
during which a CU's LDS array was serving an access; conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import argparse, csv, glob, os, re, sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--out", default="profiles/r03_pmc_lds_summary.csv")
args = ap.parse_args()
files = glob.glob(os.path.join(args.dir, "**", "*counter_collection.csv"), recursive=True)
if not files:
    sys.exit("no counter_collection.csv")
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
with open(files[0]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("unsigned short", "bf16")
        if not ("gemm_" in name or "attn_" in name or "infonce" in name):
            continue
        a = acc[name][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
rows = []
for name, cs in acc.items():
    m = {k: v[1] / v[0] for k, v in cs.items()}
    n = next(iter(cs.values()))[0]
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    cu_cycles = gui / 8 * 256
    idx = m.get("SQ_LDS_IDX_ACTIVE", 0.0)
    rows.append((name, n, idx / cu_cycles if gui else 0.0, m.get("SQ_LDS_BANK_CONFLICT", 0.0) / idx if idx else 0.0, m.get("SQ_INSTS_LDS", 0.0),
                 m.get("SQ_WAIT_INST_LDS", 0.0) / (gui / 8 * 1024) if gui else 0.0, m.get("SQ_ACTIVE_INST_LDS", 0.0) / (gui / 8 * 1024) if gui else 0.0, gui))
rows.sort(key=lambda r: -r[1] * r[7])
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
with open(args.out, "w") as f:
    f.write("kernel,dispatches,lds_active_frac,bank_conflict_over_active,insts_lds_per_launch,wait_inst_lds_per_simd_cycle,active_inst_lds_per_simd_cycle,mean_GRBM_GUI_ACTIVE\n")
    for r in rows:
        f.write(f"\"{r[0]}\",{r[1]},{r[2]:.4f},{r[3]:.4f},{r[4]:.0f},{r[5]:.4f},{r[6]:.4f},{r[7]:.0f}\n")
for r in rows[:14]:
    print(f"{r[0][:60]:60s} n={r[1]:4d} lds_active {r[2]:.3f} conflict {r[3]:.3f} wait_lds {r[5]:.3f} active_lds {r[6]:.3f}")
