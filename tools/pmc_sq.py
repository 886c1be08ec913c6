#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes of SQ counters (tools/gpu_pass.sh) into profiles/pmc_sq.json: per instrumented kernel the
counters per launch and what bench.py derives from them --
  kernel_cycles      = SQ_BUSY_CYCLES / 32 (the counter is summed over the 32 shader engines)
  mfma_busy_counted  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel_cycles): matrix-pipe time, recomputation included
  issue_busy         = 4 x SQ_ACTIVE_INST_ANY / (1024 x kernel_cycles): share of the SIMDs' time in which an instruction of some
                       wave is being issued (SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units")
  valu_issue / lds_issue = the same for vector and LDS instructions; cycles_per_valu = 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU
usage: pmc_sq.py <pass1_dir> <pass2_dir> <out.json> [tag]"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import NAMES  # noqa: E402


def collect(d):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(set))
    for r in csv.DictReader(open(f)):
        for pat, short in NAMES:
            if pat in r["Kernel_Name"]:
                tot[short][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[short][r["Counter_Name"]].add(r["Dispatch_Id"])
                break
    return {k: {c: v / max(len(cnt[k][c]), 1) for c, v in cs.items()} for k, cs in tot.items()}


if __name__ == "__main__":
    a, b = collect(sys.argv[1]), collect(sys.argv[2])
    out = {}
    for k in sorted(set(a) | set(b)):
        c = dict(a.get(k, {}))
        c.update(b.get(k, {}))
        d = {}
        cyc = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0
        if cyc > 0:
            d["kernel_cycles"] = round(cyc)
            simd = 1024.0 * cyc
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                d["mfma_busy_counted"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4)
            for name, ctr in (("issue_busy", "SQ_ACTIVE_INST_ANY"), ("valu_issue", "SQ_ACTIVE_INST_VALU"), ("lds_issue", "SQ_ACTIVE_INST_LDS")):
                if ctr in c:
                    d[name] = round(4.0 * c[ctr] / simd, 4)
        if c.get("SQ_INSTS_VALU"):
            d["cycles_per_valu"] = round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_INSTS_VALU"], 3)
            d["valu_per_mfma"] = round(c["SQ_INSTS_VALU"] / max(c.get("SQ_INSTS_MFMA", 0.0), 1.0), 2)
        out[k] = dict(derived=d, counters={kk: round(v) for kk, v in sorted(c.items())})
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["_source"] = dict(tag=sys.argv[4] if len(sys.argv) > 4 else None, csrc_sha16=bench.csrc_sha16(),
                          how="rocprofv3 --pmc, two passes of eight SQ counters over one eager bench.py step (tools/gpu_pass.sh)")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: v.get("derived") for k, v in out.items() if k != "_source"}, indent=1))
