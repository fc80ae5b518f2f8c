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
                        ("SQ_ACTIVE_INST_VALU", "active_inst_valu_quad_cycles"), ("SQ_THREAD_CYCLES_VALU", "thread_cycles_valu"),
                        ("SQ_INSTS_VALU_MFMA_I8", "mfma_i8_insts"), ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_cycles")):
            if mean(v, c) is not None:
                row[name] = int(mean(v, c))
        per[k] = row
        tot["launches"] = tot.get("launches", 0) + n
        tot["seconds"] = tot.get("seconds", 0.0) + d * n
        for key in ("hbm_bytes_x2", "valu_insts", "salu_insts", "wave_quad_cycles", "wait_any_quad_cycles", "active_inst_valu_quad_cycles", "thread_cycles_valu", "waves",
                    "mfma_i8_insts", "mfma_busy_cycles"):
            if key in row:
                tot[key] = tot.get(key, 0) + row[key] * n
    return tot, per


def ratios(tot):
    """what the wave cycles were spent on: waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES), issuing VALU work (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES), and how many of a wave's 64
    lanes that VALU work kept busy (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); the thread-cycle counter is in cycles, the other in quad-cycles on some parts:
    a ratio above 1 is divided by 4 and the key says so)"""
    r = {}
    wc = tot.get("wave_quad_cycles")
    if wc:
        if "wait_any_quad_cycles" in tot:
            r["wait_any_frac"] = round(tot["wait_any_quad_cycles"] / wc, 4)
        if "active_inst_valu_quad_cycles" in tot:
            r["valu_issue_frac"] = round(tot["active_inst_valu_quad_cycles"] / wc, 4)
    if tot.get("active_inst_valu_quad_cycles") and "thread_cycles_valu" in tot:
        a = tot["thread_cycles_valu"] / (64.0 * tot["active_inst_valu_quad_cycles"])
        r["active_lane_frac"], r["active_lane_frac_note"] = (round(a / 4, 4), "thread cycles / (64 x 4 x quad-cycles)") if a > 1.0 else (round(a, 4), "thread cycles / (64 x cycles)")
    if "waves" in tot and tot.get("launches"):
        r["waves_per_launch"] = round(tot["waves"] / tot["launches"], 1)
    return r


units = ("FETCH_SIZE is reported in KiB and under-counts wide reads 2x on gfx950 (MI355X_MICROARCH.md, HBM): x 1024 x 2 = an upper bound of the HBM bytes; SQ_*_CYCLES in "
         "quad-cycles; every --pmc pass in its own run with --kernel-trace only")
tot, per = rows("k_me_epzs")
if tot:
    out = {"what": what, "units": units, "launches_in_run": tot["launches"], "avg_launch_s": tot["seconds"] / tot["launches"],
           "hbm_bytes_per_launch_x2": int(tot.get("hbm_bytes_x2", 0) / tot["launches"]) if "hbm_bytes_x2" in tot else None,
           "valu_insts_per_launch": int(tot.get("valu_insts", 0) / tot["launches"]) if "valu_insts" in tot else None,
           "salu_insts_per_launch": int(tot.get("salu_insts", 0) / tot["launches"]) if "salu_insts" in tot else None, **ratios(tot), "per_kernel": per}
    json.dump(out, open("profiles/%s_search_pmc.json" % RND, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
tot, per = rows("k_cu_bits")
if tot:
    out = {"what": what, "units": units, "launches_in_run": tot["launches"], "avg_launch_s": tot["seconds"] / tot["launches"],
           "valu_insts_per_launch": int(tot.get("valu_insts", 0) / tot["launches"]) if "valu_insts" in tot else None,
           "wave_quad_cycles_per_launch": int(tot.get("wave_quad_cycles", 0) / tot["launches"]) if "wave_quad_cycles" in tot else None, **ratios(tot), "per_kernel": per}
    json.dump(out, open("profiles/%s_cu_bits_pmc.json" % RND, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
tot, per = rows("k_walk")
if tot:
    out = {"what": what, "units": units, "launches_in_run": tot["launches"], "avg_launch_s": tot["seconds"] / tot["launches"],
           "hbm_bytes_per_launch": int(tot.get("hbm_bytes_x2", 0) / tot["launches"]) if "hbm_bytes_x2" in tot else None,
           "valu_insts_per_launch": int(tot.get("valu_insts", 0) / tot["launches"]) if "valu_insts" in tot else None, "per_kernel": per}
    json.dump(out, open("profiles/%s_walk_pmc.json" % RND, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))

# the matrix-core kernels of the residual chain (32x32 / 64x64 luma blocks): instructions, busy cycles, utilisation against the dense int8 peak
CUS, SIMDS, CLOCK = 256, 4, 2.4e9
PEAK_I8_TOPS = 3944.0  # MI355X_MICROARCH.md: I8 dense >= 3944 TOPS (never the 2:1-sparsity figure)
mt, mper = {}, {}
for pref in ("k_rdo_mfma", "k_dct_mfma"):
    t, pk = rows(pref)
    for k, v in pk.items():
        mper[k] = v
    for k, v in t.items():
        mt[k] = mt.get(k, 0) + v
if mt.get("launches") and "mfma_i8_insts" in mt:
    secs = mt["seconds"]
    # v_mfma_i32_32x32x32_i8: 32 x 32 x 32 x 2 = 65536 int8 operations per wave instruction
    ops = mt["mfma_i8_insts"] * 65536.0
    out = {"what": what, "units": units, "launches_in_run": mt["launches"], "avg_launch_s": secs / mt["launches"], "mfma_i8_insts_per_launch": int(mt["mfma_i8_insts"] / mt["launches"]),
           "achieved_TOPS": round(ops / secs / 1e12, 3), "peak_TOPS_i8_dense": PEAK_I8_TOPS, "mfma_utilisation": round(ops / secs / 1e12 / PEAK_I8_TOPS, 5),
           "mfma_busy_over_all_simd_cycles": round(mt.get("mfma_busy_cycles", 0) / (secs * CLOCK * CUS * SIMDS), 5) if mt.get("mfma_busy_cycles") else None,
           "share_of_kernel_time_note": "the 32x32 / 64x64 luma blocks' residual chain is < 1 % of a step's kernel time: the matrix cores are exact and idle", "per_kernel": mper}
    json.dump(out, open("profiles/%s_mfma_pmc.json" % RND, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
