#!/usr/bin/env python3
"""Per-pass timeline of the pipelined IQ->bits steps from a rocprofv3 --kernel-trace (and optionally --memory-copy-trace) csv:
(--all: every kernel, also RCCL's and torch's) for the last `--passes` hot kernels: duration, gap to the next hot kernel, and when each tail kernel of the same pass started /
ended relative to the hot kernel's start.  usage: tools/timeline.py <dir with *_kernel_trace.csv> [--passes 12]"""
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name[:30]


def main():
    d = sys.argv[1]
    passes = int(sys.argv[sys.argv.index("--passes") + 1]) if "--passes" in sys.argv else 12
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "urh::" in r["Kernel_Name"] or "--all" in sys.argv:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "")))
    copies = []
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", ""), r.get("Size", "")))
    rows.sort()
    hot = [r for r in rows if r[2].startswith("k_demod_runs")]
    if len(hot) < passes + 2:
        print("not enough hot kernels:", len(hot))
        return
    sel = hot[-(passes + 1):-1]
    print(f"{len(hot)} hot kernels; last {passes}:")
    for k, h in enumerate(sel):
        nxt = hot[hot.index(h) + 1]
        line = f"hot {h[1] - h[0]:7d} ns  gap->next {nxt[0] - h[1]:6d} ns | "
        # tail kernels that start after this hot kernel's end and before the next hot kernel's end, on another queue
        tail = [r for r in rows if not r[2].startswith("k_demod_runs") and h[1] <= r[0] < nxt[1] + 400000 and r[0] < nxt[1]]
        line += " ".join(f"{r[2].replace('k_', '')}@{(r[0] - h[1]) // 1000}+{(r[1] - r[0]) // 1000}" for r in tail[:(30 if "--all" in sys.argv else 9)])
        cp = [c for c in copies if h[0] <= c[0] < nxt[0]]
        if cp:
            line += " | copies " + " ".join(f"{c[3]}B@{(c[0] - h[0]) // 1000}+{(c[1] - c[0]) // 1000}" for c in cp[:4])
        print(line)
    durs = [h[1] - h[0] for h in sel]
    gaps = [hot[hot.index(h) + 1][0] - h[1] for h in sel]
    print(f"mean hot {sum(durs) / len(durs) / 1000:.1f} us, mean gap {sum(gaps) / len(gaps) / 1000:.1f} us, period {(sel[-1][0] - sel[0][0]) / (len(sel) - 1) / 1000:.1f} us")


if __name__ == "__main__":
    main()
