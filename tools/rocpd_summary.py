#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel launch statistics and per-kernel PMC counter sums.

    python tools/rocpd_summary.py gpurun_out/prof_x/stats/bench_results.db [more.db ...] > profiles/xxx.txt

PMC values are stored per counter *instance* (per XCD / shader engine); this script sums instances per dispatch
and then averages over dispatches, which is the per-launch figure DESIGN.md quotes.
"""
import sqlite3
import sys


def summarise(db):
    c = sqlite3.connect(db)
    q = lambda s: c.execute(s).fetchall()
    tabs = [r[0] for r in q("select name from sqlite_master where type='table'")]
    suf = [t for t in tabs if t.startswith("rocpd_metadata")][0][len("rocpd_metadata"):]
    kd, ks = "rocpd_kernel_dispatch" + suf, "rocpd_info_kernel_symbol" + suf
    pe, pi = "rocpd_pmc_event" + suf, "rocpd_info_pmc" + suf
    print(f"== {db}")
    print(f"{'kernel':90s} {'calls':>6s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'total_ms':>10s}")
    rows = q(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, "
             f"sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc")
    for r in rows:
        print(f"{r[0][:90]:90s} {r[1]:6d} {r[2]:12.3f} {r[3]:12.3f} {r[4]:12.3f} {r[5]:10.3f}")
    # bench.py launches its pre-roll (clock ramp: the first launches of a process run 5-30 % slow) and warm-up before the K timed steps and the
    # K launches of the event pass: the average of the LAST 2K dispatches is the figure to hold against the bench line's ms_per_step
    for name, n in q(f"select s.kernel_name, count(*) from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name like '%flux_kernel%' "
                     f"group by s.kernel_name having count(*) >= 30"):
        durs = [r[0] for r in q(f"select (d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id where s.kernel_name='{name}' order by d.start")]
        tail = durs[-20:]
        print(f"{name[:90]:90s} last {len(tail)} dispatches (timed steps + event pass): avg_us {sum(tail) / len(tail):.3f}  "
              f"first {len(durs) - len(tail)} (pre-roll, warm-up): avg_us {sum(durs[:-20]) / max(len(durs) - 20, 1):.3f}")
    rows = q(f"select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from {pe} e join {pi} p on e.pmc_id=p.id "
             f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, p.name")
    if rows:
        print(f"-- PMC (sum over instances, average per dispatch)")
        for r in rows:
            print(f"{r[0][:90]:90s} {r[1]:24s} {r[2] / max(r[3], 1):20.1f}  (dispatches {r[3]})")
    print()


if __name__ == "__main__":
    for db in sys.argv[1:]:
        summarise(db)
