#!/usr/bin/env python3
"""Per-phase wall time and per-class kernel table of the TIMED steps of bench.py from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/step_timeline.py gpurun_out/trace/**/**_kernel_trace.csv [--md profiles/r02_step_table.md]

The trace has one row per dispatch with start/end timestamps and the stream (queue) it ran on.  Steps are cut at the
adamw_kernel dispatches; bench.py's replay launches (the roofline leg, after the last optimizer step) are reported
separately so they never pollute the in-step averages.  Inside a step, encoder passes are cut at their first kernel
(text_embed_fwd_kernel = a forward; scatter_rows_kernel / the final fp32 LayerNorm backward = a backward), giving the wall
time of: key forward, each PGD forward / data-gradient backward, the attacked forward, the full backward, optimizer."""
import argparse
import csv
import glob
import re
import sys
from collections import defaultdict

FLOPS = {  # algorithmic FLOPs of one launch at M = 64*185 (2*M*N*K), by (class)
}


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("unsigned short", "bf16")
    return name.strip()


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append({"name": r["Kernel_Name"], "start": int(r["Start_Timestamp"]), "end": int(r["End_Timestamp"]),
                         "queue": r.get("Queue_Id", "0"), "grid": r.get("Grid_Size", ""), "wg": r.get("Workgroup_Size", "")})
    rows.sort(key=lambda r: r["start"])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--md", default=None)
    ap.add_argument("--timed-steps", type=int, default=3, help="how many optimizer steps at the END of the run were timed")
    args = ap.parse_args()
    paths = glob.glob(args.trace, recursive=True)
    if not paths:
        sys.exit(f"no trace matches {args.trace}")
    rows = load(paths[0])
    adam = [i for i, r in enumerate(rows) if r["name"].startswith("adamw_kernel")]
    if len(adam) < args.timed_steps + 1:
        sys.exit(f"only {len(adam)} optimizer steps in the trace")
    first = adam[-args.timed_steps - 1] + 1          # first dispatch after the optimizer step that precedes the timed region
    last = adam[-1]
    step_rows = rows[first:last + 1]
    replay = rows[last + 1:]
    n = args.timed_steps
    wall = (rows[last]["end"] - rows[first]["start"]) / n / 1e6
    out = []
    out.append(f"timed steps: {n}; wall per step (first dispatch -> end of adamw): {wall:.3f} ms; dispatches per step: {len(step_rows) / n:.0f}")

    # ---- per-class table (in-step) -------------------------------------------------------------------------------
    agg = defaultdict(lambda: [0, 0])
    for r in step_rows:
        a = agg[short(r["name"])]
        a[0] += 1
        a[1] += r["end"] - r["start"]
    tot = sum(a[1] for a in agg.values())
    out.append("")
    out.append("| kernel (in-step) | calls/step | avg us | ms/step | % of kernel time |")
    out.append("|---|---|---|---|---|")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        out.append(f"| `{k[:100]}` | {c / n:.1f} | {t / c / 1e3:.1f} | {t / n / 1e6:.3f} | {100 * t / tot:.1f} |")
    out.append(f"| **sum of kernel time** (streams overlap) | | | {tot / n / 1e6:.3f} | |")
    if replay:
        agg2 = defaultdict(lambda: [0, 0])
        for r in replay:
            a = agg2[short(r["name"])]
            a[0] += 1
            a[1] += r["end"] - r["start"]
        out.append("")
        out.append("| kernel (replay launches AFTER the timed steps: bench.py roofline leg) | calls | avg us |")
        out.append("|---|---|---|")
        for k, (c, t) in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:8]:
            out.append(f"| `{k[:100]}` | {c} | {t / c / 1e3:.1f} |")

    # ---- phases of the LAST timed step ---------------------------------------------------------------------------
    s0 = adam[-2] + 1
    one = rows[s0:last + 1]
    t0 = one[0]["start"]
    cuts = []
    for r in one:
        nm = r["name"]
        if nm.startswith("text_embed_fwd_kernel"):
            cuts.append((r["start"], "forward"))
        elif nm.startswith("scatter_rows_kernel"):
            cuts.append((r["start"], "backward"))
        elif nm.startswith("adamw_kernel"):
            cuts.append((r["start"], "adamw"))
    cuts.sort()
    out.append("")
    out.append("| phase of the last timed step (cut at the first kernel of each encoder pass) | start ms | length ms |")
    out.append("|---|---|---|")
    end = one[-1]["end"]
    for i, (ts, nm) in enumerate(cuts):
        te = cuts[i + 1][0] if i + 1 < len(cuts) else end
        out.append(f"| {i}: {nm} | {(ts - t0) / 1e6:.3f} | {(te - ts) / 1e6:.3f} |")
    text = "\n".join(out)
    print(text)
    if args.md:
        with open(args.md, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
