"""Where is the GPU idle?  Reads a rocprofv3 `--kernel-trace --output-format csv` trace and reports, per
hardware queue, the busy time and the largest gaps between consecutive kernels (development aid for the
pipelined `apply_batch`: the kernel time of a file's front end sums to less than the measured step).

CAVEAT (learned in round 2): under rocprofv3 the dispatches of different HIP streams are largely
serialised, so the gaps of a two-stream run (apply_batch) are NOT the gaps of the un-profiled run -- compare
per-kernel durations (tools/trace_table.py) and host timestamps (`batch_timeline_s` of the bench line) instead.

usage: python tools/gap_report.py <dir-or-kernel_trace.csv> [min_gap_ms]"""
import csv
import os
import sys
from collections import defaultdict


def find_trace(path):
    if os.path.isfile(path):
        return path
    for root, _, files in os.walk(path):
        for f in files:
            if f.endswith("kernel_trace.csv"):
                return os.path.join(root, f)
    raise SystemExit(f"no *kernel_trace.csv under {path}")


def short(name):
    name = name.replace("void ", "")
    for cut in ("(", "<"):
        i = name.find(cut)
        if i > 0:
            name = name[:i]
    return name.split("::")[-1][:40]


def main():
    path = find_trace(sys.argv[1])
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = []
    with open(path) as fp:
        rd = csv.DictReader(fp)
        cols = {c.lower(): c for c in rd.fieldnames}
        cs, ce, cn = cols["start_timestamp"], cols["end_timestamp"], cols["kernel_name"]
        cq = cols.get("queue_id") or cols.get("stream_id")
        for r in rd:
            rows.append((int(r[cs]), int(r[ce]), r[cq] if cq else "0", short(r[cn])))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    print(f"{len(rows)} kernels over {(t1 - t0) / 1e6:.1f} ms ({path})")
    # union of busy intervals over all queues
    busy, cur_s, cur_e = 0, None, None
    for s, e, _, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"any-queue busy {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms")
    per_q = defaultdict(list)
    for r in rows:
        per_q[r[2]].append(r)
    for q, rs in sorted(per_q.items(), key=lambda kv: -len(kv[1])):
        b = sum(e - s for s, e, _, _ in rs)
        print(f"\nqueue {q}: {len(rs)} kernels, busy {b / 1e6:.1f} ms, span {(rs[-1][1] - rs[0][0]) / 1e6:.1f} ms")
        gaps = []
        for a, c in zip(rs[:-1], rs[1:]):
            g = c[0] - a[1]
            if g > min_gap * 1e6:
                gaps.append((g, a, c))
        tot = sum(g for g, _, _ in gaps)
        print(f"  gaps > {min_gap} ms: {len(gaps)}, total {tot / 1e6:.1f} ms")
        by_pair = defaultdict(lambda: [0, 0])
        for g, a, c in gaps:
            k = (a[3], c[3])
            by_pair[k][0] += g
            by_pair[k][1] += 1
        for (ka, kc), (g, n) in sorted(by_pair.items(), key=lambda kv: -kv[1][0])[:14]:
            print(f"    {g / 1e6:8.1f} ms in {n:3d} gaps  after {ka:32s} before {kc}")
        small = sum(c[0] - a[1] for a, c in zip(rs[:-1], rs[1:]) if 0 < c[0] - a[1] <= min_gap * 1e6)
        print(f"  gaps <= {min_gap} ms (launch gaps): total {small / 1e6:.1f} ms")


if __name__ == "__main__":
    main()
