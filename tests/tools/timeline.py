"""Print the kernel timeline of the last analysis step in a rocprofv3 (rocpd sqlite) kernel trace: start / end offsets
in ms relative to the step's first fft512 launch, to see which tails sit on the critical path.
    python tests/tools/timeline.py gpurun_out/tl/x_results.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute(f"select {name}, start, end from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if "fft512_kernel" in r[0]]
    if not starts:
        print("no fft512 launches in the trace")
        return
    for k, first in enumerate(starts[-2:]):
        t0 = rows[first][1]
        last = starts[starts.index(first) + 1] if first != starts[-1] else len(rows)
        print(f"# step {k}")
        for n, s, e in rows[first:last]:
            short = n.split("(")[0].replace("bg::", "")[:28]
            print(f"{short:28s} {(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f}   ({(e - s) / 1e6:7.3f} ms)")


if __name__ == "__main__":
    main(sys.argv[1])
