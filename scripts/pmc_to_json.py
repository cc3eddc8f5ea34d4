#!/usr/bin/env python3
"""rocprofv3 --pmc pass directories (scripts/gpu_profiles.sh) -> profiles/<name>.json with per-launch means and the
gfx950 HBM-byte corrections of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE in KB, x2 on gfx950; WRITE_SIZE in KB).
usage: pmc_to_json.py <pmc_dir> <out.json> [kernel-substring ...]"""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffsheg_amd.buildid import kernel_build_id

base, out = sys.argv[1], sys.argv[2]
filters = sys.argv[3:] or ["tl_linear_kernel", "linear_attention_tiled"]
acc = collections.OrderedDict()
for d in sorted(glob.glob(base + "/*/")):
    try:
        rows = list(csv.DictReader(open(d + "p_counter_collection.csv")))
    except Exception as e:                                   # noqa: BLE001
        print(d, "ERR", e); continue
    for r in rows:
        name = r["Kernel_Name"]
        if not any(f in name for f in filters):
            continue
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(dsh::TlArgs\)$|\(.*\)$", "", name)
        name = re.sub(r"^dsh::", "", name)
        name = re.sub(r", 0>$", ">", name)                   # drop the (default) ablation template argument
        acc.setdefault(name, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {"kernel_build_id": kernel_build_id(), "source": "rocprofv3 --kernel-trace --pmc <set> (separate passes, scripts/gpu_profiles.sh) over ONE single-stream step of the default bench (SHOW B=950 T=88 CFG ddim25 bf16): per-launch means of every kernel at its real shape",
       "correction": "FETCH_SIZE is reported in KB and on gfx950 counts 64 B per 128 B request for wide coalesced reads: bytes = FETCH_SIZE*1024*2 "
                     "(MI355X_MICROARCH.md HBM section); WRITE_SIZE*1024 uncorrected", "kernels": {}}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"counters_mean_per_launch": m}
    if "FETCH_SIZE" in m: e["hbm_read_bytes"] = m["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in m: e["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m: e["hbm_traffic_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    if "TCC_HIT_sum" in m: e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        e["mfma_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)     # GUI_ACTIVE summed over 8 XCDs; 1024 SIMDs
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        e["wave_cycles_breakdown"] = {k2: m[c] / w for k2, c in (("wait_any", "SQ_WAIT_ANY"), ("wait_inst_any", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY")) if c in m}
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, list(res["kernels"]))
