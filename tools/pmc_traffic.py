#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection.csv) into profiles/pmc_traffic.json:
HBM-side bytes per launch for the kernels bench.py instruments.  Corrections per MI355X_MICROARCH.md §HBM:
counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled (the raw
value is kept alongside).  usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [tag]"""
import collections, csv, glob, json, sys

# (the forward's main pass only: k_attn_fwd_asm or the compiled bound-based kernel <4, 4, ...>; the adaptive fallback launch <3, 4, ...> that
# follows it finds no flagged workgroup in the benchmark and would halve the per-launch average)
NAMES = [("k_attn_fwd_asm", "attn_fwd"), ("k_attn_fwd_bf16<4", "attn_fwd"), ("k_attn_bwd_dkv_kb", "attn_bwd_dkv"), ("k_attn_bwd_dkv_bf16", "attn_bwd_dkv"), ("k_attn_bwd_dq_kb", "attn_bwd_dq"), ("k_attn_bwd_dq_bf16", "attn_bwd_dq"),
         ("k_attn_fwd_f32", "attn_fwd"), ("k_attn_bwd_dkv_f32", "attn_bwd_dkv"), ("k_attn_bwd_dq_f32", "attn_bwd_dq"),
         ("k_gno_fwd_bf16<3", "gno_fwd_nh3"), ("k_gno_fwd_bf16<2", "gno_fwd_nh2"), ("k_gno_bwd3_bf16<3", "gno_bwd_nh3"),
         ("k_gno_bwd3_bf16<2", "gno_bwd_nh2"), ("k_attn_bwd_fused", "attn_bwd"), ("k_attn_bwd_asm", "attn_bwd"),
         ("k_gno_fwd<3", "gno_fwd_nh3"), ("k_gno_fwd<2", "gno_fwd_nh2"), ("k_gno_bwd<3", "gno_bwd_nh3"),
         ("k_gno_bwd<2", "gno_bwd_nh2")]


def collect(d, counter):
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    tot, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for pat, short in NAMES:
            if pat in r["Kernel_Name"]:
                tot[short] += float(r["Counter_Value"])
                cnt[short].add(r["Dispatch_Id"])
    return {k: tot[k] / max(len(cnt[k]), 1) for k in tot}


if __name__ == "__main__":
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f_raw, w_raw = fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        out[k] = dict(bytes_per_launch=2 * f_raw + w_raw, fetch_raw_bytes=f_raw, fetch_corrected_bytes=2 * f_raw,
                      write_bytes=w_raw, note="FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes")
    # provenance: bench.py reports these bytes only while the kernel sources still hash to this value
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["_source"] = dict(tag=sys.argv[4] if len(sys.argv) > 4 else None, csrc_sha16=bench.csrc_sha16())
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))
