#!/usr/bin/env python
"""profiles/traffic.json from the two PMC summaries of a round (tools/rocprof_summary.py output of the FETCH_SIZE and the
WRITE_SIZE pass): KiB per dispatch and dispatches per kernel, as rocprofv3 reports them (bench.py doubles FETCH_SIZE when it
converts to bytes, MI355X_MICROARCH.md HBM section).

usage: tools/traffic_from_pmc.py profiles/r02a_pmc_fetch.txt profiles/r02a_pmc_write.txt WIDTH HEIGHT SEED "<command>" [profiles/r02a_full_pipeline_kernel_trace.txt] > profiles/traffic.json
With the trace summary of the same command: rocprofv3's average execution time per kernel as well (bench.py: roofline.rocprof_avg_launch_us)."""
import json
import re
import sys


def read(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"\s*(\w+)\s+([0-9.eE+-]+)\s+(\d+)\s+(.*)$", line)
        if m and m.group(1) == counter:
            name = re.search(r"(k_\w+)", m.group(4))
            if name:
                k = name.group(1) + ("<1>" if "<1>" in m.group(4) else "<0>" if "<0>" in m.group(4) else "")
                out[k] = (float(m.group(2)), int(m.group(3)))
    return out


def read_trace(path):
    """avg_us per kernel from the `calls total_us avg_us pct name` table of the --kernel-trace --stats summary"""
    out = {}
    for line in open(path):
        m = re.match(r"\s*(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+(.*)$", line)
        if m:
            name = re.search(r"(k_\w+)", m.group(5))
            if name:
                k = name.group(1) + ("<1>" if "<1>" in m.group(5) else "<0>" if "<0>" in m.group(5) else "")
                out[k] = (float(m.group(3)), int(m.group(1)))
    return out


def main(fp, wp, w, h, seed, cmd, trace=None):
    f, wr = read(fp, "FETCH_SIZE"), read(wp, "WRITE_SIZE")
    tr = read_trace(trace) if trace else {}
    kernels = {}
    for k in sorted(set(f) | set(wr)):
        fs, fn = f.get(k, (0.0, 1))
        ws, wn = wr.get(k, (0.0, 1))
        kernels[k] = {"fetch_kib": round(fs / max(fn, 1), 1), "write_kib": round(ws / max(wn, 1), 1), "dispatches": fn}
        if k in tr:
            kernels[k]["trace_avg_us"], kernels[k]["trace_dispatches"] = tr[k]
    json.dump({"source": f"{fp} + {wp} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, `{cmd}`)",
               "workload": {"width": int(w), "height": int(h), "seed": int(seed)},
               "unit": "KiB per dispatch, as reported (FETCH_SIZE is doubled when converted to bytes: MI355X_MICROARCH.md, HBM section)",
               "kernels": kernels}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:8])
