#!/usr/bin/env python3
"""Distils tools/pmc_summary.py output (per-kernel means of the rocprofv3 --pmc passes of tools/gpu/r02_pmc.sh) into the two small files bench.py and DESIGN.md
quote: profiles/r02_search_pmc.json (the integer motion search = the SAD kernel of the path) and profiles/r02_mfma_pmc.json (the fused residual chain on MFMA).
Usage: make_pmc_profiles.py gpurun_out/r02pmc/pmc_serial.json gpurun_out/r02pmc/pmc_mfma.json"""
import json
import sys

ser, mf = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))


def mean(v, c):
    return v[c]["mean_per_launch"] if c in v else None


# ---- search: every k_me_epzs instantiation, weighted by launches of one step
ks = {k: v for k, v in ser.items() if k.startswith("k_me_epzs")}
tot = {}
per = {}
for k, v in ks.items():
    n = v["FETCH_SIZE"]["launches"]  # launches of this kernel in one step (one pass, one rep)
    d = v["_duration_ns"]["mean"] * 1e-9
    row = {"launches_in_run": n, "avg_launch_us": round(d * 1e6, 1),
           "hbm_bytes_x2": int(mean(v, "FETCH_SIZE") * 1024 * 2), "l2_read_bytes": int(mean(v, "TCP_TCC_READ_REQ_sum") * 64),
           "l2_hit_rate": round(mean(v, "TCC_HIT_sum") / max(1.0, mean(v, "TCC_HIT_sum") + mean(v, "TCC_MISS_sum")), 4),
           "waves": int(mean(v, "SQ_WAVES")), "valu_insts": int(mean(v, "SQ_INSTS_VALU")), "salu_insts": int(mean(v, "SQ_INSTS_SALU")),
           "wave_quad_cycles": int(mean(v, "SQ_WAVE_CYCLES")), "wait_any_quad_cycles": int(mean(v, "SQ_WAIT_ANY")), "wait_inst_any_quad_cycles": int(mean(v, "SQ_WAIT_INST_ANY")),
           "lds_idx_active": int(mean(v, "SQ_LDS_IDX_ACTIVE")), "lds_bank_conflict": int(mean(v, "SQ_LDS_BANK_CONFLICT"))}
    per[k] = row
    for key in ("hbm_bytes_x2", "l2_read_bytes"):
        tot[key] = tot.get(key, 0) + row[key] * n
    tot["launches"] = tot.get("launches", 0) + n
    tot["seconds"] = tot.get("seconds", 0.0) + d * n
out = {"what": "rocprofv3 --pmc passes (FETCH_SIZE | TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum | SQ_* | SQ_LDS_*; each in its own run with --kernel-trace only) over one bench step "
               "(tools/probe_step.py 1 --serial: 3840x2160 i.i.d. picture, the four levels one after the other on one stream); per-launch means of every k_me_epzs instantiation",
       "units": "FETCH_SIZE is reported in KiB and under-counts wide reads 2x on gfx950 (MI355X_MICROARCH.md, HBM): x 1024 x 2 = an upper bound of the HBM bytes; TCP_TCC_READ_REQ x 64 B = bytes the "
                "vector L1s asked L2 for; SQ_*_CYCLES in quad-cycles",
       "launches_in_run": tot["launches"], "avg_launch_s": tot["seconds"] / tot["launches"],
       "hbm_bytes_per_launch_x2": int(tot["hbm_bytes_x2"] / tot["launches"]), "l2_read_bytes_per_launch": int(tot["l2_read_bytes"] / tot["launches"]),
       "per_kernel": per}
json.dump(out, open("profiles/r02_search_pmc.json", "w"), indent=1)

# ---- MFMA: k_rdo_mfma<32|64>
res = {"what": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE (and FETCH_SIZE in its own "
               "run) over tools/probe_step.py 3 --mfma: the fused residual chain (DIFF, SSD, DCT, quant, dequant, IDCT, recon, SSD) of every 32x32 / 64x64 luma block of a 3840x2160 picture",
       "per_kernel": {}}
for k, v in mf.items():
    if "mfma" not in k:
        continue
    d = v["_duration_ns"]["mean"] * 1e-9
    busy, gui = mean(v, "SQ_VALU_MFMA_BUSY_CYCLES"), mean(v, "GRBM_GUI_ACTIVE")
    insts = mean(v, "SQ_INSTS_VALU_MFMA_I8")
    n = 32 if "<32>" in k else 64
    # v_mfma_i32_32x32x32_i8: 32*32*32 MACs = 65536 ops x 2; dense I8 peak from the guide's table (>= 3944 TOPS at 16x16x64; quoted against ~5 PFLOP/s-class fp8 rate / 2 = 2x bf16)
    ops = insts * 2 * 32 * 32 * 32 if insts else None
    res["per_kernel"][k] = {"avg_launch_us": round(d * 1e6, 1), "mfma_i8_insts": int(insts) if insts else None, "mfma_busy_cycles": int(busy) if busy else None,
                            "grbm_gui_active_cycles": int(gui) if gui else None,
                            "mfma_busy_over_all_simd_cycles": round(busy / (gui * 1024), 5) if busy and gui else None,
                            "achieved_TOPS": round(ops / d / 1e12, 2) if ops else None, "peak_TOPS_i8_dense": 3944, "mfma_utilisation": round(ops / d / 3944e12, 5) if ops else None,
                            "hbm_bytes_x2": int(mean(v, "FETCH_SIZE") * 1024 * 2) if mean(v, "FETCH_SIZE") else None, "valu_insts": int(mean(v, "SQ_INSTS_VALU")) if mean(v, "SQ_INSTS_VALU") else None}
json.dump(res, open("profiles/r02_mfma_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
print(json.dumps(res, indent=1))
