"""Summarise a rocprofv3 (ROCm 7 rocpd sqlite) kernel trace as a --stats style table.
    python tests/tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/<name>.kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute(
        f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        f"from kernels group by {name} order by 3 desc"))
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>12s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'pct':>7s}")
    for n, c, tot, avg, mn, mx in rows:
        print(f"{n[:70]:70s} {c:6d} {tot / 1e6:12.3f} {avg / 1e3:12.2f} {mn / 1e3:12.2f} {mx / 1e3:12.2f} {100.0 * tot / total:7.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
