#!/usr/bin/env python
"""Turn a rocprofv3 results .db (rocpd sqlite, `--kernel-trace --stats`) into the per-kernel summary table kept under profiles/."""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [f"# rocprofv3 --kernel-trace --stats  ({db_path})", f"{'kernel':120s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}"]
    for name, calls, tot, avg, pct in rows:
        lines.append(f"{name[:120]:120s} {calls:6d} {tot / 1e3:10.3f} {avg:10.1f} {pct:6.2f}")
    lines.append(f"# total kernel time {sum(r[2] for r in rows) / 1e3:.3f} ms")
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
