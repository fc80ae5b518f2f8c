"""Seeded cases for the integer-pel diamond search, shared by the reference-pinning test, the golden generator and
the GPU test."""
import ctypes as C

import numpy as np

from _libs import MeJob, MeParams, MeResult, oracle_me, ptr

PAD = 144
REFI_BITS_2_0 = 1  # xeve_tbl_refi_bits[2][0] (xeve_tbl.c:498-517)


def make_planes(r, textured, W=256, H=192):
    s = W + 2 * PAD
    if textured:  # smooth moving texture: the search actually walks, the far rings get used
        yy, xx = np.mgrid[0:H + 2 * PAD, 0:s]
        base = (512 + 300 * np.sin(xx / 23.0) * np.cos(yy / 17.0) + 150 * np.sin((xx + yy) / 41.0))
        org = np.clip(base + r.integers(-6, 7, size=base.shape), 0, 1023).astype(np.int16)
        shift = (int(r.integers(-30, 31)), int(r.integers(-30, 31)))
        ref = np.roll(org, shift, axis=(0, 1))
        ref = np.clip(ref + r.integers(-6, 7, size=base.shape), 0, 1023).astype(np.int16)
    else:
        org = r.integers(0, 1024, size=(H + 2 * PAD, s), dtype=np.int16)
        ref = r.integers(0, 1024, size=(H + 2 * PAD, s), dtype=np.int16)
    return dict(org=org, ref=ref, s=s, W=W, H=H)


def make_job(r, pl, S, bi):
    W, H = pl["W"], pl["H"]
    x, y = int(r.integers(0, (W - S) // 8 + 1)) * 8, int(r.integers(0, (H - S) // 8 + 1)) * 8
    min_clip, max_clip = (-127, -127), (W - 1, H - 1)  # -MAX_CU_SIZE + 1 .. pic - 1 (xeve_pinter.c:2124-2127): every block read stays inside the 144-sample padding
    mvp = (int(r.integers(-160, 161)), int(r.integers(-160, 161)))  # quarter pel
    msr = int(r.choice([32, 64, 128]))
    sr = 5 if bi == 1 else int(r.choice([msr // 4, msr // 2, msr]))
    cx = min(max(x + (mvp[0] >> 2), min_clip[0]), max_clip[0])
    cy = min(max(y + (mvp[1] >> 2), min_clip[1]), max_clip[1])
    rng = [min(max(cx - sr, min_clip[0]), max_clip[0]), min(max(cy - sr, min_clip[1]), max_clip[1]),
           min(max(cx + sr, min_clip[0]), max_clip[0]), min(max(cy + sr, min_clip[1]), max_clip[1])]
    org_bi = (2 * r.integers(0, 1024, size=S * S) - r.integers(0, 1024, size=S * S)).astype(np.int16)
    return dict(org=pl["org"], ref=pl["ref"], s=pl["s"], x=x, y=y, S=S, bi=bi, min_clip=min_clip, max_clip=max_clip, range=rng,
                gmvp=(mvp[0] + (x << 2), mvp[1] + (y << 2)), mvi=(mvp[0] + (x << 2), mvp[1] + (y << 2)), msr=msr, sr=sr,
                lambda_mv=int(r.integers(1 << 16, 1 << 23)), faststep=int(r.choice([2, 3])), mot_other=int(r.integers(2, 30)),
                org_bi=org_bi, beststep_in=int(r.integers(0, 3)))


def make_case(r, S, bi, textured):
    return make_job(r, make_planes(r, textured), S, bi)


def params_of(c):
    return MeParams(c["lambda_mv"], REFI_BITS_2_0, c["mot_other"], c["bi"], c["faststep"], c["msr"], c["sr"],
                    (C.c_int32 * 2)(*c["min_clip"]), (C.c_int32 * 2)(*c["max_clip"]), 0)


def job_of(c, org_off=0):
    return MeJob(c["x"], c["y"], org_off, (C.c_int16 * 4)(*c["range"]), (C.c_int16 * 2)(*c["gmvp"]), (C.c_int16 * 2)(*c["mvi"]), c["beststep_in"])


def run_oracle(c):
    O = oracle_me()
    lg = c["S"].bit_length() - 1
    p, j, res = params_of(c), job_of(c), MeResult()
    O.xo_me_ipel_diamond(ptr(c["org"], PAD * c["s"] + PAD), c["s"], ptr(c["org_bi"]), ptr(c["ref"], PAD * c["s"] + PAD), c["s"], C.byref(j), lg, lg,
                         10, C.byref(p), C.byref(res))
    return res


# ---- sub-pel pattern search (me_spel_pattern) ----------------------------------------------------------------------
def make_spel_job(r, pl, S, bi):
    from _libs import oracle

    W, H = pl["W"], pl["H"]
    x, y = int(r.integers(0, (W - S) // 8 + 1)) * 8, int(r.integers(0, (H - S) // 8 + 1)) * 8
    mvp = (int(r.integers(-160, 161)), int(r.integers(-160, 161)))
    mvi = (int(r.integers(-30, 31)) * 4, int(r.integers(-30, 31)) * 4)  # an integer-pel MV, quarter-pel units
    org_bi = (2 * r.integers(0, 1024, size=S * S) - r.integers(0, 1024, size=S * S)).astype(np.int16)
    return dict(org=pl["org"], ref=pl["ref"], s=pl["s"], x=x, y=y, S=S, bi=bi, gmvp=(mvp[0] + (x << 2), mvp[1] + (y << 2)), mvi=mvi,
                lambda_mv=int(r.integers(1 << 16, 1 << 23)), mot_other=int(r.integers(2, 30)), org_bi=org_bi,
                hpel_cnt=int(r.choice([2, 4, 8])), qpel_cnt=int(r.choice([0, 4, 8])))


def run_oracle_spel(c):
    from _libs import SpelJob, SpelParams, oracle_spel

    O = oracle_spel()
    lg = c["S"].bit_length() - 1
    p = SpelParams(c["lambda_mv"], REFI_BITS_2_0, c["mot_other"], c["bi"], c["hpel_cnt"], c["qpel_cnt"])
    j = SpelJob(c["x"], c["y"], 0, (C.c_int16 * 2)(*c["gmvp"]), (C.c_int16 * 2)(*c["mvi"]))
    res = MeResult()
    O.xo_me_spel_pattern(ptr(c["org"], PAD * c["s"] + PAD), c["s"], ptr(c["org_bi"]), ptr(c["ref"], PAD * c["s"] + PAD), c["s"], C.byref(j), lg, lg, 10,
                         O.mc_l_coeff, C.byref(p), C.byref(res))
    return res


# ---- EPZS driver (pinter_me_epzs) ------------------------------------------------------------------------------------
def make_epzs_job(r, pl, S, bi):
    W, H = pl["W"], pl["H"]
    x, y = int(r.integers(0, (W - S) // 8 + 1)) * 8, int(r.integers(0, (H - S) // 8 + 1)) * 8
    msr = int(r.choice([32, 64]))
    return dict(org=pl["org"], ref=pl["ref"], s=pl["s"], x=x, y=y, S=S, bi=bi, min_clip=(-127, -127), max_clip=(W - 1, H - 1),
                mvp=(int(r.integers(-100, 101)), int(r.integers(-100, 101))), mv0=(int(r.integers(-25, 26)) * 4, int(r.integers(-25, 26)) * 4),
                msr=msr, sr=int(r.choice([msr // 4, msr // 2, msr])), lambda_mv=int(r.integers(1 << 16, 1 << 22)), mot_other=int(r.integers(2, 30)),
                org_bi=(2 * r.integers(0, 1024, size=S * S) - r.integers(0, 1024, size=S * S)).astype(np.int16),
                hpel_cnt=int(r.choice([2, 4, 8])), qpel_cnt=int(r.choice([0, 4, 8])))


def epzs_params_of(c):
    from _libs import EpzsParams, SpelParams

    # reserved: bit 0 = raster search on (me_complexity > 1), bits 8.. = refi (the raster step scales with it); refi_bits [2][0] == [2][1]
    me = MeParams(c["lambda_mv"], REFI_BITS_2_0, c["mot_other"], c["bi"], 3, c["msr"], c["sr"], (C.c_int32 * 2)(*c["min_clip"]),
                  (C.c_int32 * 2)(*c["max_clip"]), int(c.get("raster", 0)) | (int(c.get("refi", 0)) << 8))
    return EpzsParams(me, SpelParams(0, 0, 0, 0, c["hpel_cnt"], c["qpel_cnt"]))


def run_oracle_epzs(c, with_mot=False):
    """(cost, mv) of pinter_me_epzs; with_mot: also what it leaves in pi->mot_bits[lidx] (-1 = untouched)"""
    from _libs import oracle_epzs

    O = oracle_epzs()
    lg = c["S"].bit_length() - 1
    mvp, mv = np.array(c["mvp"], np.int16), np.array(c["mv0"], np.int16)
    p = epzs_params_of(c)
    mot = C.c_int(-1)
    O.xo_me_epzs_mot.restype = C.c_uint32
    O.xo_me_epzs_mot.argtypes = O.xo_me_epzs.argtypes + [C.POINTER(C.c_int)]
    cost = O.xo_me_epzs_mot(ptr(c["org"], PAD * c["s"] + PAD), c["s"], ptr(c["org_bi"]), ptr(c["ref"], PAD * c["s"] + PAD), c["s"], c["x"], c["y"], ptr(mvp),
                            ptr(mv), lg, lg, 10, O.mc_l_coeff, C.byref(p), C.byref(mot))
    return (cost, int(mv[0]), int(mv[1]), mot.value) if with_mot else (cost, int(mv[0]), int(mv[1]))
