#!/usr/bin/env python3
"""Dump every dispatch of the LAST timed step of a rocprofv3 kernel trace (see tools/step_timeline.py for the command):
    index, queue, start (us from the step's first dispatch), duration (us), short kernel name.
Used to read per-SHAPE in-step GEMM times off the launch order (the class table averages proj with fc2 etc.)."""
import sys, glob
sys.path.insert(0, __import__("os").path.dirname(__file__))
from step_timeline import load, short

rows = load(glob.glob(sys.argv[1], recursive=True)[0])
adam = [i for i, r in enumerate(rows) if r["name"].startswith("adamw_kernel")]
first, last = adam[-2] + 1, adam[-1]
t0 = rows[first]["start"]
for i, r in enumerate(rows[first:last + 1]):
    print(f"{i:4d} q{r['queue']:>3s} {(r['start'] - t0) / 1e3:9.1f} {(r['end'] - r['start']) / 1e3:7.1f}  {short(r['name'])[:90]}")
