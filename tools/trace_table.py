"""rocprofv3 kernel-trace csv -> compact npz (name ids, queue, start, end) for offline analysis (development aid).
usage: python tools/trace_table.py <trace dir> out.npz"""
import csv, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gap_report import find_trace, short
path = find_trace(sys.argv[1])
names, rows = {}, []
with open(path) as fp:
    rd = csv.DictReader(fp)
    cols = {c.lower(): c for c in rd.fieldnames}
    for r in rd:
        n = short(r[cols["kernel_name"]])
        full = r[cols["kernel_name"]]
        if "k_conv3x3_wino" in full:
            n = "wino" + full[full.find("<"):full.find(">") + 1].replace(" ", "")
        rows.append((names.setdefault(n, len(names)), int(r[cols.get("queue_id", cols.get("stream_id"))]),
                     int(r[cols["start_timestamp"]]), int(r[cols["end_timestamp"]]),
                     int(r[cols["grid_size"]]) if "grid_size" in cols else 0))
a = np.array(rows, dtype=np.int64)
np.savez_compressed(sys.argv[2], rows=a, names=np.array(sorted(names, key=names.get)))
print(len(rows), "kernels", len(names), "names")
