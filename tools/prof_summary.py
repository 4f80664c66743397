"""Dump the per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd sqlite DB or
*_kernel_stats.csv) to a text table for profiles/.   python tools/prof_summary.py <db-or-dir> <out.txt>"""
import csv
import glob
import os
import sqlite3
import sys


def rows_from(path):
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    if dbs:
        con = sqlite3.connect(dbs[0])
        return [tuple(r) for r in con.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels")], "us"
    csvs = glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
    if not csvs:
        raise SystemExit("no rocpd .db or kernel_stats.csv under " + path)
    out = []
    for r in csv.DictReader(open(csvs[0])):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                    float(r["Percentage"])))
    return out, "us"


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows, unit = rows_from(src)
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(src)}); durations in {unit}\n")
        f.write(f"{'calls':>7} {'total':>12} {'avg':>10} {'pct':>6}  kernel\n")
        for name, calls, total, avg, pct in rows:
            f.write(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name}\n")
    print(open(dst).read()[:3000])


if __name__ == "__main__":
    main()
