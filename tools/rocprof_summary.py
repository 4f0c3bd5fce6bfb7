#!/usr/bin/env python
"""Turn a rocprofv3 `*_results.db` (rocpd SQLite, ROCm 7.2 default output) into the text summary
committed under profiles/: per-kernel calls / total / average duration (the --stats view) plus
launch geometry and register/LDS usage of each kernel, and PMC counter sums when present.

usage: tools/rocprof_summary.py gpurun_out/prof_xxx/yyy_results.db > profiles/r01_xxx.txt"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    print(f"# rocprofv3 summary of {path}")
    print("## kernel stats (durations in us)")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  name")
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{calls:6d} {total:12.2f} {avg:10.3f} {pct:6.2f}  {name}")
    print("\n## launch geometry / resources (first dispatch of each kernel)")
    q = ("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, "
         "min(duration), max(duration), count(*) from kernels group by name")
    for r in c.execute(q):
        print(f"{r[0]}\n    grid={r[1]} wg={r[2]} lds={r[3]}B scratch={r[4]}B vgpr={r[5]} agpr={r[6]} sgpr={r[7]} "
              f"min={r[8] / 1e3:.2f}us max={r[9] / 1e3:.2f}us n={r[10]}")
    try:
        rows = list(c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                              "group by kernel_name, counter_name"))
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n## PMC counters (sum over dispatches, n dispatches)")
        for k, n, v, cnt in rows:
            print(f"{n:>28} {v:18.1f} {cnt:6d}  {k}")


if __name__ == "__main__":
    main(sys.argv[1])
