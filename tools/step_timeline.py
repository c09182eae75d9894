#!/usr/bin/env python
"""Timeline of the LAST step in a rocprofv3 --kernel-trace results .db (rocpd sqlite): per-queue busy time, time with no kernel running anywhere, time in which
only the side queues run (the step's stream waits at a join), and the longest such intervals with the kernels around them.
    tools/step_timeline.py <results.db> [k [marker]]      (the step = from launch k of the marker kernel to launch k + 1; default k = -2, marker rtfs::stft_kernel.
    bench.py --mode train ends with three steps whose weight-gradient launches are in line: a timed step of `--steps 4 --warmup 2` is k = 4)"""
import sqlite3
import sys


def main(db_path, k=-2, marker="rtfs::stft_kernel"):
    db = sqlite3.connect(db_path)
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    view = next((n for n in names if n == "kernels"), None) or next(n for n in names if "kernel" in n.lower() and "dispatch" in n.lower())
    cols = [r[1] for r in db.execute(f"pragma table_info('{view}')")]
    print("# view", view, cols)
    import os
    qcol = os.environ.get("TL_COL", "stream_id")
    combos = list(db.execute(f"select queue_id, stream_id, count(*), sum(end - start) / 1e6 from {view} group by queue_id, stream_id"))
    print("# (queue_id, stream_id, kernels, summed ms) over the whole trace:", combos)
    rows = list(db.execute(f"select name, start, end, {qcol} from {view} order by start"))
    marks = [r[1] for r in rows if marker in r[0]]
    k = int(k)
    t0, t_end = marks[k], marks[k + 1]
    print(f'# {len(marks)} launches of the marker; step = launches {k} .. {k + 1}')
    rows = [r for r in rows if t0 <= r[1] < t_end]
    queues = {}
    for n, s, e, q in rows:
        queues.setdefault(q, []).append((s, e, n))
    main_q = max(queues, key=lambda q: sum(e - s for s, e, _ in queues[q]))
    print(f"# step {(t_end - t0) / 1e6:.2f} ms, {len(rows)} kernels, queues: " + ", ".join(f"{q}: {len(v)} kernels {sum(e - s for s, e, _ in v) / 1e6:.2f} ms" for q, v in queues.items()))

    def union(iv):
        out = []
        for s, e in sorted(iv):
            if out and s <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e)
            else:
                out.append([s, e])
        return out

    allu = union([(s, e) for _, s, e, _ in rows])
    mainu = union([(s, e) for s, e, _ in queues[main_q]])
    busy_all = sum(e - s for s, e in allu) / 1e6
    busy_main = sum(e - s for s, e in mainu) / 1e6
    span = (t_end - rows[0][1]) / 1e6
    print(f"span {span:.2f} ms; some kernel running {busy_all:.2f} ms (idle {span - busy_all:.2f}); main queue {main_q} running {busy_main:.2f} ms (not running {span - busy_main:.2f})")
    # gaps of the main queue, with what runs elsewhere meanwhile
    gaps = [(mainu[i + 1][0] - mainu[i][1], mainu[i][1], mainu[i + 1][0]) for i in range(len(mainu) - 1)]
    gaps.sort(reverse=True)
    print("# longest intervals in which the main queue runs nothing:")
    for g, a, b in gaps[:14]:
        before = max((r for r in queues[main_q] if r[1] <= a), key=lambda r: r[1])[2][:60]
        after = min((r for r in queues[main_q] if r[0] >= b), key=lambda r: r[0])[2][:60]
        others = sorted({n[:50] for n, s, e, q in rows if q != main_q and s < b and e > a})
        print(f"  {g / 1e3:8.1f} us at {(a - rows[0][1]) / 1e6:7.2f} ms  after [{before}] before [{after}]  meanwhile: {others[:4]}")
    for q, v in queues.items():
        if q == main_q:
            continue
        qu = union([(s, e) for s, e, _ in v])
        ov, j = 0, 0
        for s, e in qu:  # overlap of this queue's busy intervals with the main queue's
            for ms, me in mainu:
                if me <= s:
                    continue
                if ms >= e:
                    break
                ov += min(e, me) - max(s, ms)
        print(f"# queue {q}: busy {sum(e - s for s, e in qu) / 1e6:.2f} ms, of which {ov / 1e6:.2f} ms while the main queue runs a kernel")
    small = sum(g for g, _, _ in gaps if g < 20000) / 1e6
    print(f"# gaps < 20 us: {small:.2f} ms in {sum(1 for g, _, _ in gaps if g < 20000)} gaps; >= 20 us: {sum(g for g, _, _ in gaps if g >= 20000) / 1e6:.2f} ms in {sum(1 for g, _, _ in gaps if g >= 20000)}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
