"""Turn a rocprofv3 results .db (default output of `rocprofv3 --kernel-trace --stats`) into a per-kernel stats CSV
(name, calls, total_us, avg_us, pct) -- the same numbers rocprofv3's *_kernel_stats.csv carries."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for name, calls, total, avg, pct in rows:
            short = name if len(name) < 160 else name[:157] + "..."
            w.writerow([short, calls, round(total, 3), round(avg, 3), round(pct, 3)])
    print(f"wrote {out_csv}: {len(rows)} kernels, {sum(r[2] for r in rows) / 1e3:.2f} ms of kernel time")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
