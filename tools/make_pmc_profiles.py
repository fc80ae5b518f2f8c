#!/usr/bin/env python3
"""Distils tools/pmc_summary.py output (per-kernel means of rocprofv3 --pmc passes over a batch-encoder run, tools/probe_enc.py) into profiles/r03_search_pmc.json
(the integer motion search k_me_epzs = the SAD kernel of the path: what bench.py's roofline.traffic quotes) and profiles/r03_cu_bits_pmc.json (CABAC bit counting,
the class with the largest share of the GPU time).  usage: make_pmc_profiles.py pmc.json "<what was run>" [round prefix, default r03]; a k_walk row (the fused walk) goes to profiles/<prefix>_walk_pmc.json """
import json
import sys

ser, what = json.load(open(sys.argv[1])), sys.argv[2]
RND = sys.argv[3] if len(sys.argv) > 3 else "r03"


def mean(v, c):
    return v[c]["mean_per_launch"] if c in v else None


def rows(prefix):
    tot, per = {}, {}
    for k, v in ser.items():
        if not k.startswith(prefix):
            continue
        n = v["_duration_ns"]["launches"]
        d = v["_duration_ns"]["mean"] * 1e-9
        row = {"launches_in_run": n, "avg_launch_us": round(d * 1e6, 2)}
        if mean(v, "FETCH_SIZE") is not None:
            row["hbm_bytes_x2"] = int(mean(v, "FETCH_SIZE") * 1024 * 2)
        for c, name in (("SQ_WAVES", "waves"), ("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_SALU", "salu_insts"), ("SQ_WAVE_CYCLES", "wave_quad_cycles"),
                        ("SQ_BUSY_CYCLES", "sq_busy_cycles"), ("SQ_WAIT_ANY", "wait_any_quad_cycles"), ("SQ_WAIT_INST_ANY", "wait_inst_any_quad_cycles"),
                        ("SQ_ACTIVE_INST_VALU", "active_inst_valu_quad_cycles")):
            if mean(v, c) is not None:
                row[name] = int(mean(v, c))
        per[k] = row
        tot["launches"] = tot.get("launches", 0) + n
        tot["seconds"] = tot.get("seconds", 0.0) + d * n
        for key in ("hbm_bytes_x2", "valu_insts", "salu_insts", "wave_quad_cycles"):
            if key in row:
                tot[key] = tot.get(key, 0) + row[key] * n
    return tot, per


units = ("FETCH_SIZE is reported in KiB and under-counts wide reads 2x on gfx950 (MI355X_MICROARCH.md, HBM): x 1024 x 2 = an upper bound of the HBM bytes; SQ_*_CYCLES in "
         "quad-cycles; every --pmc pass in its own run with --kernel-trace only")
tot, per = rows("k_me_epzs")
if tot:
    out = {"what": what, "units": units, "launches_in_run": tot["launches"], "avg_launch_s": tot["seconds"] / tot["launches"],
           "hbm_bytes_per_launch_x2": int(tot.get("hbm_bytes_x2", 0) / tot["launches"]) if "hbm_bytes_x2" in tot else None,
           "valu_insts_per_launch": int(tot.get("valu_insts", 0) / tot["launches"]) if "valu_insts" in tot else None,
           "salu_insts_per_launch": int(tot.get("salu_insts", 0) / tot["launches"]) if "salu_insts" in tot else None, "per_kernel": per}
    json.dump(out, open("profiles/%s_search_pmc.json" % RND, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
tot, per = rows("k_cu_bits")
if tot:
    out = {"what": what, "units": units, "launches_in_run": tot["launches"], "avg_launch_s": tot["seconds"] / tot["launches"],
           "valu_insts_per_launch": int(tot.get("valu_insts", 0) / tot["launches"]) if "valu_insts" in tot else None,
           "wave_quad_cycles_per_launch": int(tot.get("wave_quad_cycles", 0) / tot["launches"]) if "wave_quad_cycles" in tot else None, "per_kernel": per}
    json.dump(out, open("profiles/%s_cu_bits_pmc.json" % RND, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
tot, per = rows("k_walk")
if tot:
    out = {"what": what, "units": units, "launches_in_run": tot["launches"], "avg_launch_s": tot["seconds"] / tot["launches"],
           "hbm_bytes_per_launch": int(tot.get("hbm_bytes_x2", 0) / tot["launches"]) if "hbm_bytes_x2" in tot else None,
           "valu_insts_per_launch": int(tot.get("valu_insts", 0) / tot["launches"]) if "valu_insts" in tot else None, "per_kernel": per}
    json.dump(out, open("profiles/%s_walk_pmc.json" % RND, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
