#!/usr/bin/env python
"""Summarises a rocprofv3 --kernel-trace results .db (rocpd sqlite) into a per-kernel CSV
(name, calls, total_ms, avg_us, min_us, max_us, pct) -- the same figures `--stats` prints.
Usage: python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.csv"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, "
         f"max(d.end-d.start)/1e3, max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
         f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    # (rocprofv3's arch_vgpr_count / accum_vgpr_count read HALF of the code object's register counts on this ROCm -- igemm 256x256: 128 against .vgpr_count 254,
    #  the 4-wave loop 216 against 432, attention 104 against 207: profiles/r06_a_igemm_register_budget.txt has the real table -- hence the column names)
    print("kernel,calls,total_ms,avg_us,min_us,max_us,pct,rocprof_arch_vgpr(=half_of_code_object),rocprof_accum_vgpr,sgpr,lds_bytes")
    for r in rows:
        print(f"\"{r[0]}\",{r[1]},{r[2]:.3f},{r[3]:.1f},{r[4]:.1f},{r[5]:.1f},{100 * r[2] / tot:.2f},{r[6]},{r[7]},{r[8]},{r[9]}")


if __name__ == "__main__":
    main(sys.argv[1])
