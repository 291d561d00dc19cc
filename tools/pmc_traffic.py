"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Each dir holds *_counter_collection.csv of one pass (counters collected in their own runs, as the
MI355X guide prescribes).  FETCH_SIZE / WRITE_SIZE are in KiB units; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads -> doubled (both raw and corrected numbers are written)."""
import collections, csv, glob, json, sys

# kernels that one launcher (= one profiler scope / bench.py roofline entry) dispatches between: folded into the
# launcher's name, the per-kernel split kept under "_split"
ALIASES = {"k_conv3x3_wino32": "k_conv3x3_wino", "k_linkage_centroid_mw": "k_linkage_centroid",
           "k_linkage_fast": "k_linkage_centroid", "k_lf_square": "k_linkage_centroid",
           "k_lf_row_nearest": "k_linkage_centroid"}

def load(d, name):
    acc = collections.defaultdict(lambda: [0.0, set()])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("pa::", "")
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1].add(r["Dispatch_Id"])
    return {k: (v[0], len(v[1])) for k, v in acc.items()}

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    n = max(nf, nw, 1)
    out[k] = {"launches": n, "fetch_size_kib_per_launch_raw": f / max(nf, 1),
              "write_size_kib_per_launch_raw": w / max(nw, 1),
              "hbm_bytes_per_launch": round((2.0 * f / max(nf, 1) + w / max(nw, 1)) * 1024)}
for src, dst in ALIASES.items():
    if src not in out:
        continue
    a, b = out.pop(src), out.get(dst)
    if b is None:
        out[dst] = a
        continue
    n = a["launches"] + b["launches"]
    merged = {"launches": n}
    for key in ("fetch_size_kib_per_launch_raw", "write_size_kib_per_launch_raw", "hbm_bytes_per_launch"):
        merged[key] = (a[key] * a["launches"] + b[key] * b["launches"]) / n
    merged["hbm_bytes_per_launch"] = round(merged["hbm_bytes_per_launch"])
    merged["_split"] = {**b.pop("_split", {}), dst: b, src: a}
    out[dst] = merged
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
