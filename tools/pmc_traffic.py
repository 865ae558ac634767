#!/usr/bin/env python3
"""HBM-side traffic of bench.py's roofline kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cost 3 + 2 of
the 4 TCC slots: they cannot share a pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --out profiles/roofline_traffic.json

Only the REPLAY launches of bench.py's roofline leg are read (the dispatches after the last adamw_kernel: fc1 + fc2 forward
GEMMs back to back, the launches `roofline.achieved` is measured on).  Correction as the guide's HBM section prescribes for
gfx950: FETCH_SIZE counts 64 B per 128-B request -> x2; WRITE_SIZE is exact; both are reported in KiB."""
import argparse
import csv
import glob
import json
import os
import sys


def per_kernel(dirpath, counter):
    files = glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under {dirpath}")
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    last_adam = max((i for i, r in enumerate(rows) if r[1].startswith("adamw_kernel")), default=-1)
    replay = rows[last_adam + 1:]
    out = {}
    for _, name, v in replay:
        if "gemm_sw_kernel" in name or "gemm_st_kernel" in name:
            key = "fc1" if "gemm_sw_kernel" in name else "fc2"
            out.setdefault(key, []).append(v)
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--out", default="profiles/roofline_traffic.json")
    ap.add_argument("--M", type=int, default=11840)
    args = ap.parse_args()
    fetch, nf = per_kernel(args.fetch_dir, "FETCH_SIZE")
    write, nw = per_kernel(args.write_dir, "WRITE_SIZE")
    M, D, H = args.M, 768, 3072
    alg = {"fc1": M * D * 2 + H * D * 2 + 2 * M * H * 2, "fc2": M * H * 2 + D * H * 2 + M * D * 4 + M * D * 4}   # A + W + outputs (+ fp32 residual)
    res = {}
    for k in ("fc1", "fc2"):
        b = (2.0 * fetch[k] + write[k]) * 1024.0
        res[k] = {"fetch_kib_raw": fetch[k], "write_kib": write[k], "bytes_corrected": b, "bytes_algorithmic": alg[k],
                  "ratio": b / alg[k], "launches": [nf[k], nw[k]]}
    pair = 0.5 * (res["fc1"]["bytes_corrected"] + res["fc2"]["bytes_corrected"])
    pair_alg = 0.5 * (alg["fc1"] + alg["fc2"])
    rec = {"mlp_fwd_pair": {"bytes_per_launch": round(pair), "algorithmic_bytes_per_launch": round(pair_alg), "ratio": round(pair / pair_alg, 3),
                            "note": "mean over the replayed fc1 / fc2 forward launches of (2 x FETCH_SIZE + WRITE_SIZE) KiB from separate rocprofv3 "
                                    "--pmc passes (gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE x 2); algorithmic = operands + outputs once",
                            "detail": res}}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench                                             # the identity of the kernel sources this record was measured on
    rec["kernel_build_id"] = bench.roofline_kernel_build_id()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(rec, open(args.out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
