"""Summarise one rocprofv3 --pmc SQ pass into per-kernel matrix-pipe / wait / LDS-conflict shares.

usage: python tools/pmc_sq.py <dir> > profiles/rN_pipeline_pmc_sq.txt
pass:  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
       SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -- <command>
gpu_cycles = GRBM_GUI_ACTIVE / 8 XCDs; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x gpu_cycles);
wait / active shares are relative to SQ_WAVE_CYCLES."""
import collections, csv, glob, sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("pa::", "")
        if not k.startswith("k_"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
rows = []
for k, c in acc.items():
    gpu = c["GRBM_GUI_ACTIVE"] / 8.0
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    rows.append((gpu, f"{k:24s} launches={len(disp[k]):5d} gpu_cycles={gpu / 1e6:9.2f}M "
                      f"mfma_util={100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1024 * gpu, 1):5.1f}% "
                      f"wait_any={100 * c['SQ_WAIT_ANY'] / wc:3.0f}% "
                      f"wait_inst={100 * c['SQ_WAIT_INST_ANY'] / wc:3.0f}% "
                      f"active={100 * c['SQ_ACTIVE_INST_ANY'] / wc:3.0f}% "
                      f"lds_conflict_cycles={int(c['SQ_LDS_BANK_CONFLICT'])}"))
for _, line in sorted(rows, reverse=True):
    print(line)
