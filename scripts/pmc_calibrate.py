#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this part against launches whose HBM bytes are known (plain 16 B-per-lane
copies), cold (4 GiB moved: far beyond the 256 MiB Infinity Cache) and hot (the source written by the launch in front, 32 / 96 MiB —
the situation of every layer kernel of the bench, whose inputs are the previous launch's outputs).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d D/f -o p --output-format csv -- python scripts/pmc_calibrate.py run
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d D/w -o p --output-format csv -- python scripts/pmc_calibrate.py run
    python scripts/pmc_calibrate.py parse D
"""
import csv, glob, sys

CASES = [("cold", 2048), ("hot", 32), ("hot", 96), ("hot", 192)]          # MiB per buffer
REPS = 3


def run():
    import torch
    d = "cuda:0"
    for kind, mib in CASES:
        n = mib * (1 << 20) // 4
        src = torch.empty(n, device=d, dtype=torch.float32); dst = torch.empty_like(src)
        src.fill_(1.0); dst.fill_(2.0); torch.cuda.synchronize()
        for _ in range(REPS):
            if kind == "hot":
                src.fill_(3.0)                       # the producer: the copy's input was just written
            dst.copy_(src)
            torch.cuda.synchronize()
        del src, dst
    print("calibration launches done")


def parse(base):
    vals = {}
    for f in glob.glob(base + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = int(r.get("Dispatch_Id") or r.get("Correlation_Id") or 0)
            vals.setdefault(key, {"name": r["Kernel_Name"], "grid": r.get("Grid_Size")})[r["Counter_Name"]] = float(r["Counter_Value"])
    # the copies: elementwise kernels whose grid covers a whole buffer; listed in dispatch order with the bytes one launch moves
    print("%-8s %-60s %12s %12s %12s   %s" % ("dispatch", "kernel", "grid", "FETCH_KB", "WRITE_KB", "bytes per launch: read (copy only) / written"))
    for k in sorted(vals):
        v = vals[k]
        nm = v["name"]
        short = ("copy" if "copy" in nm.lower() or "direct_copy" in nm else "fill" if "Fill" in nm else nm)[:60]
        g = int(v["grid"] or 0)
        print("%-8d %-60s %12d %12.0f %12.0f" % (k, short, g, v.get("FETCH_SIZE", float("nan")), v.get("WRITE_SIZE", float("nan"))))
    print("\nexpected KiB per copy launch (read = written):", ", ".join("%s %d MiB: %d" % (k, m, m * 1024) for k, m in CASES))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2])
