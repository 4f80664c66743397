"""Per-dispatch durations of the kernels matching a pattern in a rocprofv3 --kernel-trace run (rocpd sqlite DB): the stats table
averages calls with different grids (the decoder kernels run once per point set and step).
  python tools/prof_calls.py <db-or-dir> <substring> [<substring> ...]"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    con = sqlite3.connect(dbs[0])
    try:
        cur = con.execute("select * from kernels limit 1")
    except sqlite3.Error:
        print([r for r in con.execute("select type, name from sqlite_master")])
        raise
    cols = [d[0] for d in cur.description]
    name_c = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gcols = [c for c in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z", "workgroup_x", "workgroup_size_x") if c in cols]
    groups = defaultdict(list)
    for row in con.execute(f"select {name_c}, start, end, {', '.join(gcols) if gcols else '0'} from kernels order by start"):
        if any(p in row[0] for p in pats):
            groups[(row[0][:70],) + tuple(row[3:])].append((row[2] - row[1]) / 1e3)
    print("columns:", gcols)
    for k, v in sorted(groups.items()):
        v2 = sorted(v)
        print(f"{len(v):4d} calls  median {v2[len(v2) // 2]:8.2f} us  min {v2[0]:8.2f}  max {v2[-1]:8.2f}   {k}")


if __name__ == "__main__":
    main()
