#!/usr/bin/env python
"""Per-kernel-family summary of a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE pass
(counter_collection.csv) -> JSON on stdout.  mfma_pipe_util = MFMA busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs);
clock = GRBM_GUI_ACTIVE / 8 / duration."""
import collections
import csv
import json
import sys

FAMILIES = [("igemm 256x256 tile (qkv, proj, fc1, fc2, wide convs)", lambda n: "igemm_kernel<256, 256" in n or "igemm_kernelILi256ELi256" in n),
            ("attention_kernel_mix", lambda n: "attention_kernel_mix" in n),
            ("igemm other tiles", lambda n: "igemm_kernel" in n),
            ("layernorm", lambda n: "layernorm_kernel" in n),
            ("bilinear", lambda n: "bilinear" in n)]


def family(name):
    for fam, pred in FAMILIES:
        if pred(name):
            return fam
    return None


def main(path):
    disp = {}
    for r in csv.DictReader(open(path)):
        fam = family(r["Kernel_Name"])
        if fam is None:
            continue
        d = disp.setdefault(r["Dispatch_Id"], dict(fam=fam, us=(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, c={}))
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = collections.OrderedDict()
    for fam, _ in FAMILIES:
        ds = [d for d in disp.values() if d["fam"] == fam]
        if not ds:
            continue
        e = {}
        for k in sorted(ds[0]["c"]):
            e[k] = sum(d["c"].get(k, 0.0) for d in ds) / len(ds)
        e["launches"] = len(ds)
        e["avg_us_profiled"] = sum(d["us"] for d in ds) / len(ds)
        if "GRBM_GUI_ACTIVE" in e:
            e["clock_GHz_est"] = e["GRBM_GUI_ACTIVE"] / 8 / e["avg_us_profiled"] / 1e3
            if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
                e["mfma_pipe_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * e["GRBM_GUI_ACTIVE"] / 8)
        out[fam] = e
    print(json.dumps({"note": __doc__.strip(), "per_kernel_family": out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
