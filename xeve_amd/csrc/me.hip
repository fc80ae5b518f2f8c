// xeve_amd/csrc/me.hip -- integer-pel diamond motion search on the GPU: the consumer of the SAD kernel.
//
// reference: me_ipel_diamond  src_base/xeve_pinter.c:363-551, with get_mv_bits (:74-120), MV_COST (:47) and the
// re-centring of get_range_ipel (:122-140).  ONE WAVE PER JOB runs the whole data-dependent search: round 0 is the
// dense (2d+1)^2 grid around the clipped start, later rounds are 4 / 8 / 16-point diamonds of doubling radius around
// the INITIAL centre, until `faststep` rounds pass without improvement.  Per candidate the block SAD is a DPP
// butterfly over the candidate's lane group (same lane layout as k_sad_sq); per round the winner is a wave-wide
// minimum over 64-bit keys (cost << 32 | evaluation order), which reproduces the reference's
// "first strictly smaller cost wins" tie-break exactly.
#include <cstdlib>
#include <hip/hip_ext.h>
#include "xh_common.h"

// developer switch (measurement): 0 = every candidate row through the vector L1 (default: measured fastest), 1 = dense round from an LDS window
static const int g_me_lds = getenv("XEVE_HIP_ME_LDS") ? atoi(getenv("XEVE_HIP_ME_LDS")) : 0;
// developer switch (measurement): 1 = one candidate per lane (cpl_*), 0 = the block spread over the lanes (me_*)
static const int g_me_cpl = getenv("XEVE_HIP_ME_CPL") ? atoi(getenv("XEVE_HIP_ME_CPL")) : 1;

template <int S> struct MGeo {
    static constexpr int LPR = S / 8, RPP = XH_WAVE / LPR, CPP = RPP >= S ? RPP / S : 1, NP = RPP >= S ? 1 : S / RPP, GROUP = XH_WAVE / CPP;
};

// The dense round's window in LDS (LDSM != 0): every candidate of the (2d+1)^2 grid reads its rows out of ONE staged copy of the
// (S + 2d) x (S + 2d) samples the grid covers -- staged with coalesced dword loads (a row of the window is contiguous in the plane),
// read back as 16-byte row segments at any 2-byte offset (unaligned ds_read_b128).  d = 2 (uni) / 5 (bi-prediction refinement),
// xeve_pinter.c:409-416.  One window per wave.  MEASURED (profiles/r02_lds_search.md): on one MI355X the window does not pay -- the search class of a
// 3840x2160 picture takes 16.4 ms through the vector L1, 17.3 ms with the window, 18.4 ms with a second, dword-aligned copy of it (two ds_read2_b32 per
// segment; since removed) -- so the L1 path stays the default and this form is kept behind XEVE_HIP_ME_LDS=1 for reproduction.
template <int S, bool BI> struct MWin {
    static constexpr int DMAX = BI ? 5 : 2, H = S + 2 * DMAX, NDW = S / 2 + DMAX + 1, PITCH = 2 * NDW; // NDW dwords = S + 2 DMAX + 2 samples: + parity, + round-up
    static constexpr int PELS = H * PITCH;
};

// 16-point diamond of L1 radius 4 (xeve_pinter.c:57-65); the 8-point ring is every other point halved
__device__ __constant__ int8_t c_dia16[16][2] = {{-4, 0}, {-3, 1}, {-2, 2}, {-1, 3}, {0, 4}, {1, 3}, {2, 2}, {3, 1},
                                                 {4, 0}, {3, -1}, {2, -2}, {1, -3}, {0, -4}, {-1, -3}, {-2, -2}, {-3, -1}};

__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- per-block minima by DPP alone (wave64): quad exchanges and the two row mirrors make every row of 16 lanes uniform, row_bcast:15 / :31 carry the rows' minima up to lane
// 63, v_readlane brings the wave's minimum back as a scalar.  No LDS crossbar (ds_bpermute) on the search's critical path.
template <int CTRL, int ROWS> __device__ __forceinline__ int dpp_rows(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false); } // (rows not named keep v)
#define XH_DPP_ROW_BCAST15 0x142 // lane 15 of a row -> every lane of the next row
#define XH_DPP_ROW_BCAST31 0x143 // lane 31 -> every lane of rows 2 and 3
// the minimum of a 64-bit key over the wave and the payload of the lane that holds it; both wave-uniform on return (keys are distinct, or ~0 with no payload of interest)
__device__ __forceinline__ void wave_min_key(unsigned long long &key, int &pay)
{
#define XH_KEY_STEP(EX)                                                                      \
    {                                                                                        \
        const unsigned lo = (unsigned)EX((int)(unsigned)key), hi = (unsigned)EX((int)(unsigned)(key >> 32)); \
        const int      ob = EX(pay);                                                         \
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;                   \
        if(o < key) key = o, pay = ob;                                                       \
    }
    XH_KEY_STEP(xh_dpp<XH_DPP_QUAD_XOR1>)
    XH_KEY_STEP(xh_dpp<XH_DPP_QUAD_XOR2>)
    XH_KEY_STEP(xh_dpp<XH_DPP_ROW_HALF_MIRROR>)
    XH_KEY_STEP(xh_dpp<XH_DPP_ROW_MIRROR>)
    XH_KEY_STEP((dpp_rows<XH_DPP_ROW_BCAST15, 0xA>))
    XH_KEY_STEP((dpp_rows<XH_DPP_ROW_BCAST31, 0xC>))
#undef XH_KEY_STEP
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), 63);
    key = ((unsigned long long)hi << 32) | lo, pay = __builtin_amdgcn_readlane(pay, 63);
}

// the original block of a job (org_bi for bi-prediction refinement: 2*org - pred, may be negative): loaded once, kept in registers
template <int S, bool BI>
__device__ __forceinline__ void me_load_org(const pel *__restrict__ org0, int s_org, const pel *__restrict__ org_bi, int x, int y, int org_off, int lane,
                                            u32x4 (&org)[MGeo<S>::NP])
{
    using G = MGeo<S>;
    const int gl = lane % G::GROUP, row0 = gl / G::LPR, col = (gl % G::LPR) * 8;
    const pel *o  = BI ? org_bi + org_off : org0 + (long)y * s_org + x;
    const int  so = BI ? S : s_org;
#pragma unroll
    for(int p = 0; p < G::NP; p++) {
        org[p] = xh_ld8(o + (row0 + p * G::RPP) * so + col);
        if(BI) org[p] ^= 0x80008000u; // bias once: v_sad_u16 is unsigned
    }
}

// one complete me_ipel_diamond by the calling wave; the result is wave-uniform
template <int S, bool BI, int LDSM = 0>
__device__ __forceinline__ xeve_hip_me_result me_diamond(const u32x4 (&org)[MGeo<S>::NP], const pel *__restrict__ ref0, int s_ref, const xeve_hip_me_job &jb, int shift,
                                                         const xeve_hip_me_params &P, int lane, int *range_out = nullptr, unsigned *evals = nullptr,
                                                         pel *win = nullptr)
{
    using W = MWin<S, BI>;
    using G = MGeo<S>;
    const int slot = lane / G::GROUP, gl = lane % G::GROUP, row0 = gl / G::LPR, col = (gl % G::LPR) * 8;
    unsigned nev = 0; // block SADs evaluated (measurement only: xeve_hip_prof_*)
    int r0 = jb.range[0], r1 = jb.range[1], r2 = jb.range[2], r3 = jb.range[3];
    int bx = clip3(P.min_clip[0], P.max_clip[0], jb.mvi[0] >> 2), by = clip3(P.min_clip[1], P.max_clip[1], jb.mvi[1] >> 2);
    const int ix = bx, iy = by;
    unsigned long long best_key = 0xFFFFFFFF00000000ull; // cost_best = UINT32_MAX, order 0 (nothing evaluated beats it on a tie)
    int best_bits = 0, beststep = jb.beststep_in, step = 0, not_found = 0;
    unsigned order = 1;
    const int d = P.bi == 1 ? 5 : 2; // BI_STEP : 2 (xeve_pinter.c:409-416)

    for(;;) {
        not_found++;
        // ---- candidate set of this round
        const bool dense = step <= 2, coarse = step > 8;
        int x0 = 0, y0 = 0, wd = 1, nc;
        if(dense) {
            x0 = bx <= r0 ? bx : bx - d, y0 = by <= r1 ? by : by - d;
            const int x1 = bx >= r2 ? bx : bx + d, y1 = by >= r3 ? by : by + d;
            wd = x1 - x0 + 1;
            nc = wd * (y1 - y0 + 1);
        }
        else nc = coarse ? 16 : (step == 4 ? 5 : 9);
        unsigned long long round_key = ~0ull;
        int round_bits = 0;
        // ---- the dense grid out of the LDS window
        bool staged = false;
        int  wpar = 0;
        if(LDSM && dense && !(s_ref & 1) && wd <= 2 * W::DMAX + 1 && nc <= (2 * W::DMAX + 1) * wd) {
            const pel      *g0   = ref0 + (long)y0 * s_ref + x0;
            wpar                 = (int)(((uintptr_t)g0 >> 1) & 1);
            const uint32_t *ga   = reinterpret_cast<const uint32_t *>(g0 - wpar); // dword-aligned; row r starts s_ref / 2 dwords further
            const int       nrow = nc / wd + S - 1, tot = nrow * W::NDW, sdw = s_ref >> 1;
            uint32_t       *wa   = reinterpret_cast<uint32_t *>(win);
            for(int i = lane; i < tot; i += 64) {
                const int r = i / W::NDW, k = i - r * W::NDW;
                const uint32_t *gp = ga + (long)r * sdw + k;
                wa[i] = gp[0];
            }
            __builtin_amdgcn_wave_barrier(); // (LDS operations of one wave execute in order; this only pins the compiler's schedule)
            staged = true;
        }
        for(int c0 = 0; c0 < nc; c0 += G::CPP) {
            const int k = c0 + slot;
            int mx, my;
            if(dense) {
                const int q = k / wd;
                mx = x0 + (k - q * wd), my = y0 + q;
            }
            else if(coarse) mx = ix + (step >> 2) * c_dia16[k & 15][0], my = iy + (step >> 2) * c_dia16[k & 15][1];
            else {
                const int i = step == 4 ? 2 * k : k; // 4-point ring skips the odd points; i == 8 is the centre
                const int dx = i < 8 ? c_dia16[(2 * i) & 15][0] / 2 : 0, dy = i < 8 ? c_dia16[(2 * i) & 15][1] / 2 : 0;
                mx = ix + (step >> 1) * dx, my = iy + (step >> 1) * dy;
            }
            const bool valid = k < nc && mx <= r2 && mx >= r0 && my <= r3 && my >= r1;
            if(evals) nev += (unsigned)__popcll(__ballot(valid && gl == 0));
            int acc = 0;
            if(LDSM && staged) {
                if(valid) {
                    const int wx = mx - x0 + wpar + col; // sample offset inside a window row
                    // (explicit LDS address space: with a generic pointer the compiler folds this load and the global one below into one flat load)
                    typedef __attribute__((address_space(3))) const pel lds_pel;
                    lds_pel *r = (lds_pel *)win + (my - y0 + row0) * W::PITCH + wx;
#pragma unroll
                    for(int p = 0; p < G::NP; p++) {
                        u32x4 v = *(__attribute__((address_space(3))) const u32x4_a2 *)(r + p * G::RPP * W::PITCH);
                        if(BI) v ^= 0x80008000u;
                        acc = __builtin_amdgcn_sad_u16(org[p].x, v.x, acc);
                        acc = __builtin_amdgcn_sad_u16(org[p].y, v.y, acc);
                        acc = __builtin_amdgcn_sad_u16(org[p].z, v.z, acc);
                        acc = __builtin_amdgcn_sad_u16(org[p].w, v.w, acc);
                    }
                }
            }
            else if(valid) {
                const pel *r = ref0 + (long)(my + row0) * s_ref + mx + col;
#pragma unroll
                for(int p = 0; p < G::NP; p++) {
                    u32x4 v = xh_ld8(r + (long)p * G::RPP * s_ref);
                    if(BI) v ^= 0x80008000u;
                    acc = __builtin_amdgcn_sad_u16(org[p].x, v.x, acc);
                    acc = __builtin_amdgcn_sad_u16(org[p].y, v.y, acc);
                    acc = __builtin_amdgcn_sad_u16(org[p].z, v.z, acc);
                    acc = __builtin_amdgcn_sad_u16(org[p].w, v.w, acc);
                }
            }
            acc = xh_group_sum<G::GROUP>(acc);
            int bits = xh_mvd_bits((mx << 2) - jb.gmvp[0]) + xh_mvd_bits((my << 2) - jb.gmvp[1]) + P.refi_bits;
            if(BI) bits += P.extra_bits;
            const int sad = acc >> shift;
            const unsigned cost = ((P.lambda_mv * (unsigned)bits + (1u << 15)) >> 16) + (unsigned)(BI ? sad >> 1 : sad);
            unsigned long long key = valid ? ((unsigned long long)cost << 32) | (order + (unsigned)k) : ~0ull;
            int kb = bits;
            // minimum over the CPP candidate groups of this pass (every lane of a group holds the same key)
            wave_min_key(key, kb);
            if(key < round_key) round_key = key, round_bits = kb;
        }
        // ---- wave-uniform bookkeeping (everything below is identical in all lanes; force it into SGPRs)
        const unsigned rk_hi = (unsigned)uni((int)(round_key >> 32)), rk_lo = (unsigned)uni((int)round_key);
        round_key  = ((unsigned long long)rk_hi << 32) | rk_lo;
        round_bits = uni(round_bits);
        if(round_key < best_key) { // cost < cost_best, earliest candidate on ties
            best_key = round_key, best_bits = round_bits, not_found = 0;
            const int k = (int)(rk_lo - order);
            if(dense) {
                const int q = k / wd;
                bx = x0 + (k - q * wd), by = y0 + q, beststep = 2;
            }
            else {
                int dx, dy, mul;
                if(coarse) dx = c_dia16[k][0], dy = c_dia16[k][1], mul = step >> 2;
                else {
                    const int i = step == 4 ? 2 * k : k;
                    dx = i < 8 ? c_dia16[(2 * i) & 15][0] / 2 : 0, dy = i < 8 ? c_dia16[(2 * i) & 15][1] / 2 : 0, mul = step >> 1;
                }
                bx = ix + mul * dx, by = iy + mul * dy, beststep = step;
            }
            bx = uni(bx), by = uni(by);
        }
        order += (unsigned)nc;
        if(dense) { // get_range_ipel around the best position so far (xeve_pinter.c:463-468, 122-140)
            const int sr = P.bi == 1 ? 5 : P.range_recentre;
            r0 = clip3(P.min_clip[0], P.max_clip[0], bx - sr), r2 = clip3(P.min_clip[0], P.max_clip[0], bx + sr);
            r1 = clip3(P.min_clip[1], P.max_clip[1], by - sr), r3 = clip3(P.min_clip[1], P.max_clip[1], by + sr);
            step += 2;
        }
        if(not_found == P.faststep) break;
        if(P.bi == 1) break;
        step <<= 1;
        if(step > P.max_search_range) break;
    }
    xeve_hip_me_result res;
    res.mv[0] = (int16_t)((bx - jb.x) << 2), res.mv[1] = (int16_t)((by - jb.y) << 2);
    res.cost = (uint32_t)(best_key >> 32), res.beststep = beststep, res.best_mv_bits = best_bits;
    if(range_out) range_out[0] = r0, range_out[1] = r1, range_out[2] = r2, range_out[3] = r3; // the caller's `range`, re-centred in place (:463-468)
    if(evals) *evals += nev;
    return res;
}

// A list of integer positions evaluated by the calling wave in the list's order: cost = MV_COST + SAD, a position wins only with a strictly
// smaller cost than `cost_best`, the earliest on ties (me_raster, its 3x3 refinement grids, me_ipel_refinement).  gen(k, mx, my) -> valid.
template <int S, bool BI, class Gen>
__device__ __forceinline__ void me_eval(const u32x4 (&org)[MGeo<S>::NP], const pel *__restrict__ ref0, int s_ref, int nc, Gen gen, int gmvp_x, int gmvp_y, int shift,
                                        const xeve_hip_me_params &P, int lane, unsigned &cost_best, int &best_bits, int &bx, int &by)
{
    using G = MGeo<S>;
    const int slot = lane / G::GROUP, gl = lane % G::GROUP, row0 = gl / G::LPR, col = (gl % G::LPR) * 8;
    unsigned long long best_key = (unsigned long long)cost_best << 32; // order 0: an equal cost never beats it
    int win = -1, win_bits = 0;
    for(int c0 = 0; c0 < nc; c0 += G::CPP) {
        const int k = c0 + slot;
        int mx = 0, my = 0;
        const bool valid = k < nc && gen(k, mx, my);
        int acc = 0;
        if(valid) {
            const pel *r = ref0 + (long)(my + row0) * s_ref + mx + col;
#pragma unroll
            for(int p = 0; p < G::NP; p++) {
                u32x4 v = xh_ld8(r + (long)p * G::RPP * s_ref);
                if(BI) v ^= 0x80008000u;
                acc = __builtin_amdgcn_sad_u16(org[p].x, v.x, acc);
                acc = __builtin_amdgcn_sad_u16(org[p].y, v.y, acc);
                acc = __builtin_amdgcn_sad_u16(org[p].z, v.z, acc);
                acc = __builtin_amdgcn_sad_u16(org[p].w, v.w, acc);
            }
        }
        acc = xh_group_sum<G::GROUP>(acc);
        int bits = xh_mvd_bits((mx << 2) - gmvp_x) + xh_mvd_bits((my << 2) - gmvp_y) + P.refi_bits;
        if(BI) bits += P.extra_bits;
        const int sad = acc >> shift;
        const unsigned cost = ((P.lambda_mv * (unsigned)bits + (1u << 15)) >> 16) + (unsigned)(BI ? sad >> 1 : sad);
        unsigned long long key = valid ? ((unsigned long long)cost << 32) | (unsigned)(k + 1) : ~0ull;
        int kb = bits;
        wave_min_key(key, kb);
        if(key < best_key) best_key = key, win = (int)(unsigned)key - 1, win_bits = kb;
    }
    win = uni(win);
    if(win >= 0) {
        cost_best = (unsigned)uni((int)(best_key >> 32)), best_bits = uni(win_bits);
        int mx = 0, my = 0;
        (void)gen(win, mx, my);
        bx = uni(mx), by = uni(my);
    }
}

// =========================================================================================================
// One CANDIDATE per lane.  The mapping above spreads a block over the lanes and pays for it per candidate: a cross-lane sum, the vector cost and a
// 64-bit cross-group minimum for every 1 .. 8 candidates -- counters (profiles/r02_search_pmc.json): 2 700 VALU + 2 000 SALU instructions per 8x8 job of
// which ~5 % are v_sad_u16; the kernel is bound by its own bookkeeping, not by any memory level.  Here a lane owns a candidate: it walks the block's
// rows itself (reference row segment from the plane, original row segment broadcast from a per-wave LDS copy), so a SAD needs no reduction, the vector
// cost is computed once per candidate, and a round's winner is one 32-bit wave minimum plus a ballot (lanes are in evaluation order, so the lowest
// lane among the minima is the reference's "first strictly smaller").  The dense (2d+1)^2 round is one pass (two for the 121 candidates of the
// bi-prediction refinement); the rings that are certain to be evaluated -- the search only stops after `faststep` rounds without improvement -- share
// one pass as well, and the bookkeeping then consumes their minima ring by ring, in the reference's order.
// =========================================================================================================
typedef __attribute__((address_space(3))) const pel lds_cpel;

// the original block (org_bi for the bi-prediction refinement, biased once for the unsigned SAD) into the wave's LDS copy, dense S x S
template <int S, bool BI>
__device__ __forceinline__ void cpl_load_org(const pel *__restrict__ org0, int s_org, const pel *__restrict__ org_bi, int x, int y, int org_off, int lane, pel *lorg)
{
    const pel *o  = BI ? org_bi + org_off : org0 + (long)y * s_org + x;
    const int  so = BI ? S : s_org;
    for(int i = lane; i < S * S / 8; i += 64) {
        const int r = i / (S / 8), c = (i % (S / 8)) * 8;
        u32x4 v = xh_ld8(o + (long)r * so + c);
        if(BI) v ^= 0x80008000u;
        *reinterpret_cast<u32x4 *>(lorg + r * S + c) = v;
    }
    __builtin_amdgcn_wave_barrier();
}

template <int S, bool BI> __device__ __forceinline__ int cpl_sad(const pel *lorg, const pel *__restrict__ r, int s_ref)
{
    int acc = 0;
    lds_cpel *o = (lds_cpel *)lorg;
#pragma unroll 2
    for(int y = 0; y < S; y++) {
#pragma unroll
        for(int x = 0; x < S; x += 8) {
            u32x4 v = xh_ld8(r + x);
            if(BI) v ^= 0x80008000u;
            const u32x4 q = *(__attribute__((address_space(3))) const u32x4 *)(o + x);
            acc = __builtin_amdgcn_sad_u16(q.x, v.x, acc);
            acc = __builtin_amdgcn_sad_u16(q.y, v.y, acc);
            acc = __builtin_amdgcn_sad_u16(q.z, v.z, acc);
            acc = __builtin_amdgcn_sad_u16(q.w, v.w, acc);
        }
        r += s_ref, o += S;
    }
    return acc;
}

__device__ __forceinline__ unsigned cpl_wave_min(unsigned v)
{
    v = min(v, (unsigned)xh_dpp<XH_DPP_QUAD_XOR1>((int)v));
    v = min(v, (unsigned)xh_dpp<XH_DPP_QUAD_XOR2>((int)v));
    v = min(v, (unsigned)xh_dpp<XH_DPP_ROW_HALF_MIRROR>((int)v));
    v = min(v, (unsigned)xh_dpp<XH_DPP_ROW_MIRROR>((int)v));
    v = min(v, (unsigned)dpp_rows<XH_DPP_ROW_BCAST15, 0xA>((int)v));
    v = min(v, (unsigned)dpp_rows<XH_DPP_ROW_BCAST31, 0xC>((int)v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63); // (a scalar: the comparisons against it and the ballot that follow are wave-uniform by construction)
}

template <bool BI> __device__ __forceinline__ unsigned cpl_cost(int mx, int my, int sad_raw, int shift, int gx, int gy, const xeve_hip_me_params &P, int &bits)
{
    bits = xh_mvd_bits((mx << 2) - gx) + xh_mvd_bits((my << 2) - gy) + P.refi_bits;
    if(BI) bits += P.extra_bits;
    const int sad = sad_raw >> shift;
    return ((P.lambda_mv * (unsigned)bits + (1u << 15)) >> 16) + (unsigned)(BI ? sad >> 1 : sad);
}

// one complete me_ipel_diamond by the calling wave (candidate per lane); the result is wave-uniform
template <int S, bool BI>
__device__ __forceinline__ xeve_hip_me_result cpl_diamond(const pel *lorg, const pel *__restrict__ ref0, int s_ref, const xeve_hip_me_job &jb, int shift,
                                                          const xeve_hip_me_params &P, int lane, int *range_out = nullptr, unsigned *evals = nullptr)
{
    int r0 = jb.range[0], r1 = jb.range[1], r2 = jb.range[2], r3 = jb.range[3];
    int bx = clip3(P.min_clip[0], P.max_clip[0], jb.mvi[0] >> 2), by = clip3(P.min_clip[1], P.max_clip[1], jb.mvi[1] >> 2);
    const int ix = bx, iy = by;
    unsigned cost_best = 0xFFFFFFFFu, nev = 0;
    int best_bits = 0, beststep = jb.beststep_in, not_found = 0;
    const int d = P.bi == 1 ? 5 : 2; // BI_STEP : 2 (xeve_pinter.c:409-416)
    // ---- round 0: the dense grid around the clipped start
    {
        not_found++;
        const int x0 = bx <= r0 ? bx : bx - d, y0 = by <= r1 ? by : by - d;
        const int x1 = bx >= r2 ? bx : bx + d, y1 = by >= r3 ? by : by + d;
        const int wd = x1 - x0 + 1, nc = wd * (y1 - y0 + 1);
        unsigned rcost = 0xFFFFFFFFu;
        int rk = -1, rbits = 0;
        for(int c0 = 0; c0 < nc; c0 += 64) {
            const int k = c0 + lane, q = k / wd, mx = x0 + (k - q * wd), my = y0 + q;
            const bool valid = k < nc && mx <= r2 && mx >= r0 && my <= r3 && my >= r1;
            if(evals) nev += (unsigned)__popcll(__ballot(valid));
            int bits = 0;
            unsigned cost = 0xFFFFFFFFu;
            if(valid) cost = cpl_cost<BI>(mx, my, cpl_sad<S, BI>(lorg, ref0 + (long)my * s_ref + mx, s_ref), shift, jb.gmvp[0], jb.gmvp[1], P, bits);
            // (a real cost never reaches 0xFFFFFFFF: 15 bits of SAD scale + the vector cost)
            const unsigned mn = cpl_wave_min(cost);
            const unsigned long long at = __ballot(valid && cost == mn);
            if(at && mn < rcost) { // strictly smaller than an earlier pass's minimum; inside the pass the lowest lane = the earliest candidate
                const int wl = (int)__ffsll((long long)at) - 1;
                rcost = mn, rk = c0 + wl, rbits = __builtin_amdgcn_readlane(bits, wl);
            }
        }
        if(rk >= 0 && rcost < cost_best) {
            const int q = rk / wd;
            cost_best = rcost, best_bits = rbits, not_found = 0, bx = x0 + (rk - q * wd), by = y0 + q, beststep = 2;
        }
        // get_range_ipel around the best position so far (xeve_pinter.c:463-468, 122-140)
        const int sr = P.bi == 1 ? 5 : P.range_recentre;
        r0 = clip3(P.min_clip[0], P.max_clip[0], bx - sr), r2 = clip3(P.min_clip[0], P.max_clip[0], bx + sr);
        r1 = clip3(P.min_clip[1], P.max_clip[1], by - sr), r3 = clip3(P.min_clip[1], P.max_clip[1], by + sr);
    }
    // ---- the rings around the INITIAL centre: steps 4 (4 points + centre), 8 (8 + centre), 16, 32, ... (16 points), xeve_pinter.c:470-540
    if(not_found != P.faststep && P.bi != 1) {
        int step = 4;
        while(step <= P.max_search_range) {
            // the next (faststep - not_found) rings are evaluated whatever they find: one pass for all of them (at most three: 5 + 9 + 16 candidates, or 48)
            // (ring g of the batch has step `step << g` and 5 / 9 / 16 candidates for step 4 / 8 / larger; no arrays: they would be indexed dynamically)
            auto ring_n = [](int s2) { return s2 > 8 ? 16 : (s2 == 4 ? 5 : 9); };
            const int nr = P.faststep - not_found;
            int tot = 0, m = 0;
            for(int s2 = step; m < nr && m < 3 && s2 <= P.max_search_range; s2 <<= 1, m++) tot += ring_n(s2);
            // lane -> (ring, index inside the ring), in evaluation order
            const int c0n = ring_n(step), c1n = ring_n(step << 1);
            int ri = 0, k = lane;
            if(m > 1 && k >= c0n) k -= c0n, ri = 1;
            if(m > 2 && ri == 1 && k >= c1n) k -= c1n, ri = 2;
            const int st = step << ri;
            int mx, my;
            if(st > 8) mx = ix + (st >> 2) * c_dia16[k & 15][0], my = iy + (st >> 2) * c_dia16[k & 15][1];
            else {
                const int i = st == 4 ? 2 * k : k; // the 4-point ring skips the odd points; i == 8 is the centre
                const int dx = i < 8 ? c_dia16[(2 * i) & 15][0] / 2 : 0, dy = i < 8 ? c_dia16[(2 * i) & 15][1] / 2 : 0;
                mx = ix + (st >> 1) * dx, my = iy + (st >> 1) * dy;
            }
            const bool valid = lane < tot && mx <= r2 && mx >= r0 && my <= r3 && my >= r1;
            if(evals) nev += (unsigned)__popcll(__ballot(valid));
            int bits = 0;
            unsigned cost = 0xFFFFFFFFu;
            if(valid) cost = cpl_cost<BI>(mx, my, cpl_sad<S, BI>(lorg, ref0 + (long)my * s_ref + mx, s_ref), shift, jb.gmvp[0], jb.gmvp[1], P, bits);
            bool stop = false;
            for(int g = 0; g < m; g++) { // the rings' minima, consumed in the reference's order
                not_found++;
                const unsigned cg = ri == g ? cost : 0xFFFFFFFFu, mn = cpl_wave_min(cg);
                const unsigned long long at = __ballot(valid && ri == g && cost == mn);
                if(at && mn < cost_best) {
                    const int wl = (int)__ffsll((long long)at) - 1;
                    cost_best = mn, best_bits = __builtin_amdgcn_readlane(bits, wl), not_found = 0;
                    bx = __builtin_amdgcn_readlane(mx, wl), by = __builtin_amdgcn_readlane(my, wl), beststep = step << g;
                }
                if(not_found == P.faststep) {
                    stop = true;
                    break;
                }
            }
            if(stop || m == 0) break;
            step <<= m;
        }
    }
    xeve_hip_me_result res;
    res.mv[0] = (int16_t)((bx - jb.x) << 2), res.mv[1] = (int16_t)((by - jb.y) << 2);
    res.cost = cost_best, res.beststep = beststep, res.best_mv_bits = best_bits;
    if(range_out) range_out[0] = r0, range_out[1] = r1, range_out[2] = r2, range_out[3] = r3;
    if(evals) *evals += nev;
    return res;
}

// A list of integer positions in the list's order (me_raster, its 3x3 grids, me_ipel_refinement): a position wins only with a strictly smaller cost
// than `cost_best`, the earliest on ties.  gen(k, mx, my) -> valid.
template <int S, bool BI, class Gen>
__device__ __forceinline__ void cpl_eval(const pel *lorg, const pel *__restrict__ ref0, int s_ref, int nc, Gen gen, int gmvp_x, int gmvp_y, int shift,
                                         const xeve_hip_me_params &P, int lane, unsigned &cost_best, int &best_bits, int &bx, int &by)
{
    for(int c0 = 0; c0 < nc; c0 += 64) {
        const int k = c0 + lane;
        int mx = 0, my = 0, bits = 0;
        const bool valid = k < nc && gen(k, mx, my);
        unsigned cost = 0xFFFFFFFFu;
        if(valid) cost = cpl_cost<BI>(mx, my, cpl_sad<S, BI>(lorg, ref0 + (long)my * s_ref + mx, s_ref), shift, gmvp_x, gmvp_y, P, bits);
        const unsigned mn = cpl_wave_min(cost);
        const unsigned long long at = __ballot(valid && cost == mn);
        if(at && mn < cost_best) {
            const int wl = (int)__ffsll((long long)at) - 1;
            cost_best = mn, best_bits = __builtin_amdgcn_readlane(bits, wl), bx = __builtin_amdgcn_readlane(mx, wl), by = __builtin_amdgcn_readlane(my, wl);
        }
    }
}

template <int S, bool BI, bool CPL>
__global__ __launch_bounds__(256) void k_me_diamond(const pel *__restrict__ org0, int s_org, const pel *__restrict__ org_bi,
                                                    const pel *__restrict__ ref0, int s_ref, const xeve_hip_me_job *__restrict__ jobs,
                                                    int njobs, int shift, xeve_hip_me_params P, xeve_hip_me_result *__restrict__ out,
                                                    unsigned long long *__restrict__ units)
{
    __shared__ __attribute__((aligned(16))) pel s_lorg[CPL ? 4 * S * S : 8];
    const int lane = threadIdx.x & 63;
    const int j    = xh_xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(j >= njobs) return;
    const xeve_hip_me_job jb = jobs[j];
    if(jb.range[0] > jb.range[2]) return; // empty range = parked job (its result slot is left untouched)
    unsigned nev = 0;
    xeve_hip_me_result res;
    if constexpr(CPL) {
        pel *lorg = s_lorg + (threadIdx.x >> 6) * (S * S);
        cpl_load_org<S, BI>(org0, s_org, org_bi, jb.x, jb.y, jb.org_off, lane, lorg);
        res = cpl_diamond<S, BI>(lorg, ref0, s_ref, jb, shift, P, lane, nullptr, units ? &nev : nullptr);
    }
    else {
        u32x4 org[MGeo<S>::NP];
        me_load_org<S, BI>(org0, s_org, org_bi, jb.x, jb.y, jb.org_off, lane, org);
        res = me_diamond<S, BI>(org, ref0, s_ref, jb, shift, P, lane, nullptr, units ? &nev : nullptr);
    }
    if(lane == 0) out[j] = res;
    if(units && lane == 0) atomicAdd(XH_PROF_SLOT(units), (unsigned long long)nev * (S * S / 64));
}

extern "C" int xeve_hip_me_ipel_diamond_jobs(const pel *org0, int s_org, const pel *org_bi, const pel *ref0, int s_ref,
                                             const xeve_hip_me_job *jobs, int njobs, int log2w, int log2h, int bit_depth,
                                             const xeve_hip_me_params *params, xeve_hip_me_result *results, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(org0 && ref0 && jobs && params && results && njobs >= 0);
    XH_REQUIRE(log2w == log2h && log2w >= 3 && log2w <= 6); // Baseline inter CUs are square 8..64 (xeve_enc.c:2440-2443)
    XH_REQUIRE(bit_depth >= 8 && bit_depth <= 14 && params->bi >= 0 && params->bi <= 3 && params->faststep >= 1);
    XH_REQUIRE(params->bi == 0 || org_bi != nullptr);
    if(njobs == 0) return XEVE_HIP_OK;
    hipStream_t st = (hipStream_t)stream;
    const dim3  grid((njobs + 3) / 4);
    const int   shift = bit_depth - 8;
    const xeve_hip_me_params P = *params;
    XhProf prof(XH_PROF_SEARCH, st);
    unsigned long long *units = xh_prof_units(XH_PROF_SEARCH);
#define ME_LAUNCH(S)                                                                                                          \
    do {                                                                                                                      \
        if(g_me_cpl) {                                                                                                        \
            if(P.bi) k_me_diamond<S, true, true><<<grid, 256, 0, st>>>(org0, s_org, org_bi, ref0, s_ref, jobs, njobs, shift, P, results, units);  \
            else k_me_diamond<S, false, true><<<grid, 256, 0, st>>>(org0, s_org, org_bi, ref0, s_ref, jobs, njobs, shift, P, results, units);     \
        }                                                                                                                     \
        else if(P.bi) k_me_diamond<S, true, false><<<grid, 256, 0, st>>>(org0, s_org, org_bi, ref0, s_ref, jobs, njobs, shift, P, results, units);  \
        else k_me_diamond<S, false, false><<<grid, 256, 0, st>>>(org0, s_org, org_bi, ref0, s_ref, jobs, njobs, shift, P, results, units);     \
    } while(0)
    if(log2w == 3) ME_LAUNCH(8);
    else if(log2w == 4) ME_LAUNCH(16);
    else if(log2w == 5) ME_LAUNCH(32);
    else ME_LAUNCH(64);
#undef ME_LAUNCH
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}


// =========================================================================================================
// pinter_me_epzs per job (xeve_hip_me_epzs_jobs)
// =========================================================================================================
__device__ __forceinline__ void epzs_range(const xeve_hip_me_params &P, int cx, int cy, int16_t (&range)[4])
{
    const int sr = P.bi == 1 ? 5 : P.range_recentre; // get_range_ipel, xeve_pinter.c:122-140
    range[0] = (int16_t)clip3(P.min_clip[0], P.max_clip[0], cx - sr), range[1] = (int16_t)clip3(P.min_clip[1], P.max_clip[1], cy - sr);
    range[2] = (int16_t)clip3(P.min_clip[0], P.max_clip[0], cx + sr), range[3] = (int16_t)clip3(P.min_clip[1], P.max_clip[1], cy + sr);
}

// The whole integer stage of pinter_me_epzs by ONE WAVE PER JOB: first search, then refinement searches from the running best while the
// reference's rule asks for one (xeve_pinter.c:757-822) without leaving the kernel: no
// launch, no host round trip between the searches, the original block stays in registers across them.
// EXTRA: compiled with the branches presets fast / medium never take (me_raster, me_ipel_refinement); the plain form keeps its registers
template <int S, bool BI, bool EXTRA, int LDSM, bool CPL>
__global__ __launch_bounds__(256) void k_me_epzs(const pel *__restrict__ org0, int s_org, const pel *__restrict__ org_bi, const pel *__restrict__ ref0, int s_ref,
                                                 const xeve_hip_epzs_job *__restrict__ jobs, int njobs, int shift, xeve_hip_me_params P,
                                                 const int32_t *__restrict__ extra, EpzsState *__restrict__ st, XhSearchPlanes pl, int ipel_only,
                                                 unsigned long long *__restrict__ units, xeve_hip_spel_job *__restrict__ sj)
{
    __shared__ __attribute__((aligned(16))) pel s_win[LDSM ? 4 * LDSM * MWin<S, BI>::PELS : 8];
    __shared__ __attribute__((aligned(16))) pel s_lorg[CPL ? 4 * S * S : 8];
    pel *win = s_win + (threadIdx.x >> 6) * (LDSM * MWin<S, BI>::PELS);
    pel *lorg = s_lorg + (threadIdx.x >> 6) * (CPL ? S * S : 0);
    const int lane = threadIdx.x & 63;
    const int j    = xh_xcd_block(blockIdx.x, gridDim.x) * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if(j >= njobs) return;
    if(pl.n) { // several reference pictures in one launch: the job's plane supplies picture, index bits and range
        const int q = uni(xh_plane_of_job(pl.job_plane, pl.per_plane, j));
        ref0 = pl.ref[q], P.refi_bits = pl.refi_bits[q], P.range_recentre = pl.range[q], P.reserved = (P.reserved & 0xFF) | (pl.refi[q] << 8);
    }
    xeve_hip_epzs_job e = jobs[j];
    int yb = 0;
    if(pl.vh) { // a batch of pictures stacked vertically (xh_common.h): from here on the job's own picture, in its own coordinates
        yb = uni(xh_vh_base(e.y, pl.vh));
        org0 += (long)yb * s_org, ref0 += (long)yb * s_ref, e.y -= yb;
    }
    // the job of the sub-pel stage that follows (round 6: was a launch of its own, k_epzs_spel_jobs): y keeps the picture of a stacked batch, the predictor is in the
    // picture's own coordinates (16 bits), the centre is where the integer searches ended
    auto put_spel_job = [&](const EpzsState &z) {
        if(!sj || lane != 0) return;
        xeve_hip_spel_job q;
        q.x = e.x, q.y = e.y + yb, q.org_off = e.org_off;
        q.gmvp[0] = (int16_t)(e.mvp[0] + (e.x << 2)), q.gmvp[1] = (int16_t)(e.mvp[1] + (e.y << 2));
        q.mvi[0] = z.mv[0], q.mvi[1] = z.mv[1];
        sj[j] = q;
    };
    if(e.x < 0) { // job switched off
        EpzsState z;
        z.cost = 0xFFFFFFFFu, z.mv[0] = e.mv_start[0], z.mv[1] = e.mv_start[1], z.tmpstep = 0, z.searches = 0, z.mot_bits = 0;
        if(lane == 0) st[j] = z;
        put_spel_job(z);
        return;
    }
    if(BI && extra) P.extra_bits = uni(extra[j]); // pi->mot_bits[other list] of this CU
    const int sx = P.bi == 1 ? e.mv_start[0] : e.mvp[0], sy = P.bi == 1 ? e.mv_start[1] : e.mvp[1];
    xeve_hip_me_job m;
    m.x = e.x, m.y = e.y, m.org_off = e.org_off, m.beststep_in = 0;
    m.gmvp[0] = (int16_t)(e.mvp[0] + (e.x << 2)), m.gmvp[1] = (int16_t)(e.mvp[1] + (e.y << 2));
    m.mvi[0] = (int16_t)(sx + (e.x << 2)), m.mvi[1] = (int16_t)(sy + (e.y << 2));
    epzs_range(P, clip3(P.min_clip[0], P.max_clip[0], e.x + (sx >> 2)), clip3(P.min_clip[1], P.max_clip[1], e.y + (sy >> 2)), m.range); // clipped centre (:738-741)
    EpzsState s;
    s.cost = 0xFFFFFFFFu, s.mv[0] = e.mv_start[0], s.mv[1] = e.mv_start[1], s.tmpstep = 0, s.searches = 0, s.mot_bits = 0;
    u32x4 org[MGeo<S>::NP];
    if constexpr(CPL) cpl_load_org<S, BI>(org0, s_org, org_bi, e.x, e.y, e.org_off, lane, lorg);
    else me_load_org<S, BI>(org0, s_org, org_bi, e.x, e.y, e.org_off, lane, org);
    // the two lane mappings behind one face
    auto diamond = [&](const xeve_hip_me_job &jb, const xeve_hip_me_params &pp, int *rng, unsigned *ne) {
        if constexpr(CPL) return cpl_diamond<S, BI>(lorg, ref0, s_ref, jb, shift, pp, lane, rng, ne);
        else return me_diamond<S, BI, LDSM>(org, ref0, s_ref, jb, shift, pp, lane, rng, ne, win);
    };
    auto eval = [&](int nc, auto gen, int gx, int gy, const xeve_hip_me_params &pp, unsigned &cb, int &bb, int &px, int &py) {
        if constexpr(CPL) cpl_eval<S, BI>(lorg, ref0, s_ref, nc, gen, gx, gy, shift, pp, lane, cb, bb, px, py);
        else me_eval<S, BI>(org, ref0, s_ref, nc, gen, gx, gy, shift, pp, lane, cb, bb, px, py);
    };
    unsigned nev = 0; // (measurement only; the raster / integer-refinement branches are not counted)
    xeve_hip_me_params Q = P;
    for(int it = 0; it < 64; it++) { // (the reference's loop ends when a search no longer improves; 64 is a safety bound)
        Q.faststep = it == 0 ? 3 : 2; // MAX_FIRST_SEARCH_STEP / MAX_REFINE_SEARCH_STEP
        int rng[4];
        const xeve_hip_me_result r = diamond(m, Q, rng, units ? &nev : nullptr);
        s.tmpstep = r.beststep, s.searches++;
        if(P.bi != 1 && r.best_mv_bits > 0) s.mot_bits = r.best_mv_bits; // me_ipel_diamond's side effect on pi->mot_bits (:546-548)
        int beststep = 0;
        if(r.cost < s.cost) {
            s.cost = r.cost, s.mv[0] = r.mv[0], s.mv[1] = r.mv[1];
            const int dx = e.mvp[0] - s.mv[0], dy = e.mvp[1] - s.mv[1];
            beststep = ((dx < 0 ? -dx : dx) < 2 && (dy < 0 ? -dy : dy) < 2) ? 0 : s.tmpstep;
        }
        if(EXTRA && !BI && it == 0 && (P.reserved & 1) && beststep > 5) {
            // me_raster (:158-268; me_complexity > 1, beststep > RASTER_SEARCH_THD): a grid of step max(5, S / 2) * (refi + 1) over the range as the first
            // search left it, then 3x3 grids of halving step around the best
            const int mult = ((P.reserved >> 8) & 0xFF) + 1, st0 = (S / 2 > 5 ? S / 2 : 5), stp = st0 * mult;
            const int nx = (rng[2] - rng[0]) / stp + 1, ny = (rng[3] - rng[1]) / stp + 1;
            unsigned rc = 0xFFFFFFFFu;
            int rbits = 0, rx = (r.mv[0] >> 2) + e.x, ry = (r.mv[1] >> 2) + e.y; // (`mv` as the diamond search left it)
            eval(nx * ny, [&](int k, int &mx, int &my) { mx = rng[0] + (k % nx) * stp, my = rng[1] + (k / nx) * stp; return true; }, m.gmvp[0], m.gmvp[1], Q, rc, rbits,
                 rx, ry);
            for(int ss = (mult * st0) >> 1; ss > 0; ss >>= 1) {
                const int cx = rx, cy = ry;
                eval(9,
                     [&](int k, int &mx, int &my) {
                         mx = cx + (k % 3 - 1) * ss, my = cy + (k / 3 - 1) * ss;
                         return mx >= rng[0] && mx <= rng[2] && my >= rng[1] && my <= rng[3];
                     },
                     m.gmvp[0], m.gmvp[1], Q, rc, rbits, rx, ry);
            }
            if(rbits > 0) s.mot_bits = rbits;
            if(rc < s.cost) beststep = 5, s.cost = rc, s.mv[0] = (int16_t)((rx - e.x) << 2), s.mv[1] = (int16_t)((ry - e.y) << 2); // (:760-767)
        }
        if(P.bi == 1 || beststep <= 0) break;
        epzs_range(P, e.x + (s.mv[0] >> 2), e.y + (s.mv[1] >> 2), m.range); // the refinement centre is NOT clipped (:785-788)
        m.mvi[0] = (int16_t)(s.mv[0] + (e.x << 2)), m.mvi[1] = (int16_t)(s.mv[1] + (e.y << 2));
        m.beststep_in = s.tmpstep;
    }
    if(EXTRA && ipel_only) { // me_level <= ME_LEV_IPEL: me_ipel_refinement instead of the sub-pel pattern (:835-866, 270-361)
        int16_t rg[4];
        epzs_range(P, e.x + (s.mv[0] >> 2), e.y + (s.mv[1] >> 2), rg);
        const int ix = clip3(P.min_clip[0], P.max_clip[0], (s.mv[0] + (e.x << 2)) >> 2), iy = clip3(P.min_clip[1], P.max_clip[1], (s.mv[1] + (e.y << 2)) >> 2);
        unsigned rc = 0xFFFFFFFFu;
        int rbits = 0, rx = ix, ry = iy;
        eval(9,
             [&](int k, int &mx, int &my) {
                 // test_pos (:311): the centre, then x = -1, 0, 1 with y = -1, 0, 1 (the centre left out)
                 const int q = k == 0 ? 4 : (k <= 4 ? k - 1 : k);
                 mx = ix + (q / 3 - 1), my = iy + (q % 3 - 1);
                 return mx >= rg[0] && mx <= rg[2] && my >= rg[1] && my <= rg[3];
             },
             m.gmvp[0], m.gmvp[1], P, rc, rbits, rx, ry);
        if(P.bi != 1 && rbits > 0) s.mot_bits = rbits;
        if(rc < s.cost) s.cost = rc, s.mv[0] = (int16_t)((rx - e.x) << 2), s.mv[1] = (int16_t)((ry - e.y) << 2);
    }
    if(lane == 0) st[j] = s;
    put_spel_job(s);
    if(units && lane == 0) atomicAdd(XH_PROF_SLOT(units), (unsigned long long)nev * (S * S / 64));
}

__global__ void k_epzs_finish(int n, int bi, const EpzsState *__restrict__ st, const xeve_hip_me_result *__restrict__ spel, xeve_hip_me_result *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if(j >= n) return;
    xeve_hip_me_result r;
    r.cost = st[j].cost, r.mv[0] = st[j].mv[0], r.mv[1] = st[j].mv[1], r.beststep = 0;
    r.best_mv_bits = st[j].mot_bits;
    if(spel) { // (NULL: the integer refinement of the ME_LEV_IPEL branch already ran inside the search kernel)
        if(!bi && spel[j].best_mv_bits > 0) r.best_mv_bits = spel[j].best_mv_bits; // me_spel_pattern's side effect (:690-692)
        if(spel[j].cost < r.cost) r.cost = spel[j].cost, r.mv[0] = spel[j].mv[0], r.mv[1] = spel[j].mv[1]; // xeve_pinter.c:828-833
    }
    out[j] = r;
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t xeve_hip_me_epzs_workspace(int njobs)
{
    const size_t n = njobs > 0 ? njobs : 0;
    return al256(n * sizeof(xeve_hip_me_result)) + al256(n * sizeof(EpzsState)) + al256(n * sizeof(xeve_hip_spel_job)) +
           al256(xeve_hip_me_spel_workspace(njobs)) + 256;
}

extern "C" int xeve_hip_me_epzs_jobs(const pel *org0, int s_org, const pel *org_bi, const pel *ref0, int s_ref, const xeve_hip_epzs_job *jobs,
                                     int njobs, int log2w, int log2h, int bit_depth, const int16_t (*coef)[8],
                                     const xeve_hip_epzs_params *params, xeve_hip_me_result *results, void *workspace, size_t workspace_bytes,
                                     void *stream)
{
    return xeve_hip_me_epzs_jobs_x(org0, s_org, org_bi, ref0, s_ref, jobs, njobs, log2w, log2h, bit_depth, coef, params, nullptr, results, workspace,
                                   workspace_bytes, stream);
}

extern "C" int xeve_hip_me_epzs_jobs_x(const pel *org0, int s_org, const pel *org_bi, const pel *ref0, int s_ref, const xeve_hip_epzs_job *jobs,
                                       int njobs, int log2w, int log2h, int bit_depth, const int16_t (*coef)[8],
                                       const xeve_hip_epzs_params *params, const int32_t *extra_bits, xeve_hip_me_result *results, void *workspace,
                                       size_t workspace_bytes, void *stream)
{
    return xh_me_epzs_jobs_planes(org0, s_org, org_bi, ref0, s_ref, jobs, njobs, log2w, log2h, bit_depth, coef, params, extra_bits, results, workspace,
                                  workspace_bytes, stream, nullptr);
}

int xh_me_epzs_jobs_planes(const pel *org0, int s_org, const pel *org_bi, const pel *ref0, int s_ref, const xeve_hip_epzs_job *jobs, int njobs, int log2w, int log2h,
                           int bit_depth, const int16_t (*coef)[8], const xeve_hip_epzs_params *params, const int32_t *extra_bits, xeve_hip_me_result *results,
                           void *workspace, size_t workspace_bytes, void *stream, const XhSearchPlanes *planes)
{
    XH_ENTER();
    XH_REQUIRE(org0 && (ref0 || (planes && planes->n > 0)) && jobs && coef && params && results && workspace && njobs >= 0);
    XhSearchPlanes pl;
    pl.n = 0, pl.per_plane = 1, pl.job_plane = nullptr;
    if(planes && planes->n > 0) {
        XH_REQUIRE(planes->n <= XH_MAX_PLANES && planes->per_plane > 0 && (planes->job_plane || (long)planes->n * planes->per_plane >= njobs));
        pl = *planes;
        for(int i = pl.n; i < XH_MAX_PLANES; i++) pl.ref[i] = pl.ref[0], pl.refi_bits[i] = pl.refi_bits[0], pl.range[i] = pl.range[0], pl.refi[i] = pl.refi[0];
    }
    pl.vh = xh_vh();
    XH_REQUIRE(workspace_bytes >= xeve_hip_me_epzs_workspace(njobs) && ((uintptr_t)workspace & 15) == 0);
    XH_REQUIRE(params->me.bi == 0 || params->me.bi == 1);
    XH_REQUIRE(log2w == log2h && log2w >= 3 && log2w <= 6 && bit_depth >= 8 && bit_depth <= 14 && (params->me.bi == 0 || org_bi != nullptr));
    if(njobs == 0) return XEVE_HIP_OK;
    hipStream_t st = (hipStream_t)stream;
    char *w = static_cast<char *>(workspace);
    const size_t n = njobs;
    xeve_hip_me_result *sres = reinterpret_cast<xeve_hip_me_result *>(w);   w += al256(n * sizeof(xeve_hip_me_result));
    EpzsState          *state = reinterpret_cast<EpzsState *>(w);           w += al256(n * sizeof(EpzsState));
    xeve_hip_spel_job  *sj   = reinterpret_cast<xeve_hip_spel_job *>(w);    w += al256(n * sizeof(xeve_hip_spel_job));
    void               *sws  = w;
    const dim3 g((njobs + 255) / 256);
    xeve_hip_me_params P = params->me;
    const int ipel_only = params->hpel_cnt == 0; // me_level <= ME_LEV_IPEL: no sub-pel stage, an integer refinement inside the search kernel
    const bool extra_branches = ipel_only || (P.reserved & 1);
    xeve_hip_spel_job *sj_out = ipel_only ? nullptr : sj; // (the integer searches' kernel writes the sub-pel stage's jobs)
    {
        const dim3 grid((njobs + 3) / 4);
        const int  shift = bit_depth - 8;
        // (class timer on: the launch records its own begin and end -- the roofline kernel's duration as rocprofv3 sees it, not the stream's wait in front of it)
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        void      *ptok = xh_prof_on(XH_PROF_SEARCH) ? xh_prof_begin_kernel(XH_PROF_SEARCH, &ev0, &ev1) : nullptr;
        unsigned long long *units = xh_prof_units(XH_PROF_SEARCH);
#define EPZS_ARGS org0, s_org, org_bi, ref0, s_ref, jobs, njobs, shift, P, extra_bits, state, pl
#define EPZS_GO(K, ...)                                                                                            \
    do {                                                                                                           \
        if(ptok) hipExtLaunchKernelGGL((K), grid, dim3(256), 0, st, ev0, ev1, 0, __VA_ARGS__);                     \
        else (K)<<<grid, 256, 0, st>>>(__VA_ARGS__);                                                               \
    } while(0)
#define EPZS_LAUNCH_M(S, M, C)                                                                                     \
    do {                                                                                                           \
        if(extra_branches) {                                                                                       \
            if(P.bi) EPZS_GO((k_me_epzs<S, true, true, 0, C>), EPZS_ARGS, ipel_only, units, sj_out);               \
            else EPZS_GO((k_me_epzs<S, false, true, 0, C>), EPZS_ARGS, ipel_only, units, sj_out);                  \
        }                                                                                                          \
        else if(P.bi) EPZS_GO((k_me_epzs<S, true, false, M, C>), EPZS_ARGS, 0, units, sj_out);                     \
        else EPZS_GO((k_me_epzs<S, false, false, M, C>), EPZS_ARGS, 0, units, sj_out);                             \
    } while(0)
#define EPZS_LAUNCH(S)                                                                                             \
    do {                                                                                                           \
        if(g_me_cpl) EPZS_LAUNCH_M(S, 0, true);                                                                    \
        else if(g_me_lds == 0) EPZS_LAUNCH_M(S, 0, false);                                                         \
        else EPZS_LAUNCH_M(S, 1, false);                                                                           \
    } while(0)
        if(log2w == 3) EPZS_LAUNCH(8);
        else if(log2w == 4) EPZS_LAUNCH(16);
        else if(log2w == 5) EPZS_LAUNCH(32);
        else EPZS_LAUNCH(64);
#undef EPZS_LAUNCH
#undef EPZS_LAUNCH_M
#undef EPZS_GO
#undef EPZS_ARGS
        if(ptok) xh_prof_end_kernel(ptok);
        XH_HIP(hipGetLastError());
    }
    if(ipel_only) {
        k_epzs_finish<<<g, 256, 0, st>>>(njobs, P.bi, state, nullptr, results);
        XH_HIP(hipGetLastError());
        return XEVE_HIP_OK;
    }
    xeve_hip_spel_params SP;
    SP.lambda_mv = P.lambda_mv, SP.refi_bits = P.refi_bits, SP.extra_bits = P.extra_bits, SP.bi = P.bi;
    SP.hpel_cnt = params->hpel_cnt, SP.qpel_cnt = params->qpel_cnt;
    int rc;
    {
        XhProf prof(XH_PROF_SPEL, st);
        const XhSpelFinish fin = {state, results}; // (round 6: k_epzs_finish's merge rides on the sub-pel stage's last selection)
        rc = xh_me_spel_pattern_jobs_x(org0, s_org, org_bi, ref0, s_ref, sj, njobs, log2w, log2h, bit_depth, coef, &SP, extra_bits, sres, sws,
                                       xeve_hip_me_spel_workspace(njobs), st, pl.n ? &pl : nullptr, &fin);
    }
    return rc;
}


// ---- host-memory form of one pinter_me_epzs call (the table layer's style: synchronous, planes staged per call) ------------------
// What pi->fn_me can be pointed at (tests/test_integration_ref.py does, through shim/xeve_hip_shim.c).  org0 / ref0: sample (0, 0) of
// the original luma plane (rows 0 .. pic_h - 1 are read) and of the padded reference luma plane (pad samples around the picture).
extern "C" int xeve_hip_me_epzs_host(const pel *org0, int s_org, const pel *org_bi, const pel *ref0, int s_ref, int pad, int pic_h,
                                     const xeve_hip_epzs_job *job, int log2w, int log2h, int bit_depth, const int16_t (*coef)[8],
                                     const xeve_hip_epzs_params *params, xeve_hip_me_result *result)
{
    XH_ENTER();
    XH_REQUIRE(org0 && ref0 && job && params && result && pad >= 0 && pic_h > 0);
    const size_t eo = (size_t)s_org * pic_h, er = (size_t)s_ref * (pic_h + 2 * pad), ro = (size_t)pad * s_ref + pad, nb = (size_t)1 << (log2w + log2h);
    const size_t wsb = xeve_hip_me_epzs_workspace(1);
    pel *d_org = nullptr, *d_ref = nullptr, *d_bi = nullptr;
    char *d_misc = nullptr; // job | result | workspace
    int rc = XEVE_HIP_OK;
    auto ok = [&](hipError_t e) { if(e != hipSuccess && rc == XEVE_HIP_OK) xh_set_error("xeve_hip_me_epzs_host: %s", hipGetErrorString(e)), rc = XEVE_HIP_ERR_DEVICE; return e == hipSuccess; };
    ok(hipMalloc((void **)&d_org, eo * sizeof(pel))) && ok(hipMemcpy(d_org, org0, eo * sizeof(pel), hipMemcpyHostToDevice));
    ok(hipMalloc((void **)&d_ref, er * sizeof(pel))) && ok(hipMemcpy(d_ref, ref0 - ro, er * sizeof(pel), hipMemcpyHostToDevice));
    if(org_bi) ok(hipMalloc((void **)&d_bi, nb * sizeof(pel))) && ok(hipMemcpy(d_bi, org_bi, nb * sizeof(pel), hipMemcpyHostToDevice));
    ok(hipMalloc((void **)&d_misc, 512 + wsb)) && ok(hipMemcpy(d_misc, job, sizeof(*job), hipMemcpyHostToDevice));
    if(rc == XEVE_HIP_OK) {
        xeve_hip_epzs_job j0 = *job;
        j0.org_off = 0; // one job: its org_bi block is the whole buffer
        ok(hipMemcpy(d_misc, &j0, sizeof(j0), hipMemcpyHostToDevice));
        rc = xeve_hip_me_epzs_jobs(d_org, s_org, d_bi, d_ref + ro, s_ref, (const xeve_hip_epzs_job *)d_misc, 1, log2w, log2h, bit_depth, coef, params,
                                   (xeve_hip_me_result *)(d_misc + 256), d_misc + 512, wsb, nullptr);
        if(rc == XEVE_HIP_OK) ok(hipMemcpy(result, d_misc + 256, sizeof(*result), hipMemcpyDeviceToHost));
    }
    (void)hipFree(d_org), (void)hipFree(d_ref), (void)hipFree(d_bi), (void)hipFree(d_misc);
    return rc;
}
