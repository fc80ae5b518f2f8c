// xeve_amd/csrc/eco_lane.h -- the bitstream writer's side of one decided CTU, written by ONE lane: xeve_eco_tree (src_base/xeve_enc.c:35-100) -> xeve_eco_split_mode
// (xeve_eco.c:1377-1429) and xeve_eco_unit (:1431-1640) for every CU of the tree, on the WRITER's arithmetic coder (never reset inside a tile).
//
// Why it exists: the coder state after writing CTU n is the state CTU n + 1 starts its mode decision from (xeve_enc.c:139).  With the writer on the device a chain
// of the CTU walk (tree.hip) runs from CTU to CTU without the host, and what comes out is the picture's slice data.  Why a lane: writing is one pass over the
// decided syntax -- a few hundred to a few thousand bins per CTU, against the ~ 10^5 the decision counted -- and strictly serial; the parallel axis is the chains.
// The syntax is the writer's, not the rate estimate's: in P slices neither direct_mode_flag nor inter_pred_idc is coded (xeve_rdo_bit_cnt_cu_inter codes both).
// As the CUs are written their units get what xeve_eco_unit stores (coded flag, skip flag, luma cbf flag, CU size), after the CTU's coded flags have been reset as
// mode_analyze_lcu's tail does (xeve_mode.c:2591-2607).  Baseline (tool_admvp 0, sps_cm_init_flag 0), no delta QP, one transform block per CU (<= 64x64).
// __host__ __device__ like cu_lane.h: tests/native builds the host side, `pytest -m "not gpu"` compares it with the oracle's xo_eco_ctu, which is pinned beside the
// reference's writer in the live encoder.
#pragma once
#include "cu_lane.h"

namespace xl {
struct EcoParams {
    int idc, slice_type, log2_ctu, pic_w, pic_h, w_scu, num_refp[2];
    const uint16_t *scan[7]; // zig-zag scans of the square blocks 16x16, 32x32, 64x64 at [4], [5], [6] (xeve_tbl_scan); smaller blocks use the built-in tables
};

// What the writer reads of a decided CTU -- a third of the walk's record (no reconstruction, no maps, no motion vectors, the chroma levels dense): the form in which
// a picture's CTUs wait for the second writer pass (encode.hip keeps 2040 of them per 3840x2160 picture and GOP).  4:2:0.  Field names as in xeve_hip_ctu_data, so
// the lane code below is written once for both (xl_coef / xl_ipm0 hide the two differences).
struct CtuSyntax {
    int8_t  split_mode[XEVE_HIP_CU_DEPTHS][256];
    uint8_t pred_mode[256];
    int8_t  ipm0[256];
    int32_t nnz[3][256];
    int16_t coef_y[64 * 64], coef_u[32 * 32], coef_v[32 * 32];
    int16_t mvd[256][2][2];
    int8_t  refi[256][2];
    uint8_t mvp_idx[256][2];
};
XL const int16_t *xl_coef(const xeve_hip_ctu_data &d, int c) { return d.coef[c]; }
XL const int16_t *xl_coef(const CtuSyntax &d, int c) { return c == 0 ? d.coef_y : c == 1 ? d.coef_u : d.coef_v; }
XL int xl_ipm0(const xeve_hip_ctu_data &d, int u) { return d.ipm[0][u]; }
XL int xl_ipm0(const CtuSyntax &d, int u) { return d.ipm0[u]; }

// xeve_eco_abs_mvd + sign (xeve_eco.c:1205-1270): exp-Golomb, the first two bins on the mvd model
XL void eco_mvd1(Sbac &s, int v, Sink *o)
{
    const unsigned a = (unsigned)(v < 0 ? -v : v);
    unsigned nn = (a + 1) >> 1;
    int len = 0;
    for(; len < 16 && nn; len++) nn >>= 1;
    const unsigned code = (1u << len) | ((a + 1 - (1u << len)) & ((1u << len) - 1));
    const int nbin = 2 * len + 1;
    for(int i = 0; i < nbin; i++) {
        const unsigned b = (code >> (nbin - 1 - i)) & 1;
        if(i <= 1) sb_bin(s, XEVE_HIP_CTX_MVD, b, o);
        else sb_bin_ep(s, b, o);
    }
    if(a) sb_bin_ep(s, v < 0, o);
}
// xeve_eco_mvp_idx = a truncated unary symbol over three models (xeve_eco.c:492-511, 1190-1203)
XL void eco_mvp_idx(Sbac &s, int idx, Sink *o)
{
    for(int i = 0; i < 3; i++) {
        const int sym = i == idx ? 0 : 1;
        sb_bin(s, XEVE_HIP_CTX_MVP_IDX + i, sym, o);
        if(!sym) break;
    }
}
// xeve_eco_refi (xeve_eco.c:1158-1188)
XL void eco_refi(Sbac &s, int num_refp, int refi, Sink *o)
{
    if(num_refp <= 1) return;
    if(refi == 0) {
        sb_bin(s, XEVE_HIP_CTX_REFI, 0, o);
        return;
    }
    sb_bin(s, XEVE_HIP_CTX_REFI, 1, o);
    for(int i = 2; i < num_refp; i++) {
        const int bin = i == refi + 1 ? 0 : 1;
        if(i == 2) sb_bin(s, XEVE_HIP_CTX_REFI + 1, bin, o);
        else sb_bin_ep(s, bin, o);
        if(!bin) break;
    }
}
// xeve_eco_run_length_cc on a block that sits inside the CTU's coefficient image (pitch = the CTU's width in that component)
// WAVE (device only): the kernel runs the writer on ALL 64 lanes of its wave in step -- same state, same bins, same bytes in every lane -- so that this scan can be
// shared: the lanes fetch 64 scan positions at a time (the next 64 already in flight), a ballot finds the non-zero ones, and the coder only ever visits those.  The
// lone lane of the serial form waits for one dependent global load per POSITION (4096 of them in a 64x64 block), which is where the second writer pass spent its time.
template <bool WAVE = false> XL void eco_levels(Sbac &s, const EcoParams &E, const int16_t *blk, int pitch, int log2n, int num_sig, int ch, Sink *o)
{
    const int n = 1 << log2n, nn = n * n, t0 = ch ? 2 : 0;
    const uint16_t *scan = log2n >= 4 ? E.scan[log2n] : nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr(WAVE) {
        const int lane = threadIdx.x & 63;
        auto fetch = [&](int base) -> int {
            const int pos = base + lane;
            if(pos >= nn) return 0;
            const int at = scan ? scan[pos] : zigzag(n, pos);
            return blk[(at >> log2n) * pitch + (at & (n - 1))];
        };
        int c = fetch(0), prev = -1;
        for(int base = 0; base < nn; base += 64) {
            const int cn = base + 64 < nn ? fetch(base + 64) : 0;
            unsigned long long mask = __ballot(c != 0);
            while(mask) {
                const int b = __builtin_ctzll(mask);
                mask &= mask - 1;
                const int pos = base + b, v = __builtin_amdgcn_readlane(c, b);
                const unsigned level = (unsigned)(v < 0 ? -v : v) & 0xFFFF;
                sb_unary2(s, (unsigned)(pos - prev - 1), XEVE_HIP_CTX_RUN + t0, o);
                sb_unary2(s, level - 1, XEVE_HIP_CTX_LEVEL + t0, o);
                sb_bin_ep(s, v < 0, o);
                if(pos == nn - 1) return;
                prev = pos, num_sig--;
                sb_bin(s, XEVE_HIP_CTX_LAST + (ch ? 1 : 0), num_sig == 0, o);
                if(num_sig == 0) return;
            }
            c = cn;
        }
        return;
    }
#endif
    unsigned run = 0;
    for(int pos = 0; pos < nn; pos++) {
        const int at = scan ? scan[pos] : zigzag(n, pos), c = blk[(at >> log2n) * pitch + (at & (n - 1))];
        if(!c) {
            run++;
            continue;
        }
        const unsigned level = (unsigned)(c < 0 ? -c : c) & 0xFFFF;
        sb_unary2(s, run, XEVE_HIP_CTX_RUN + t0, o);
        sb_unary2(s, level - 1, XEVE_HIP_CTX_LEVEL + t0, o);
        sb_bin_ep(s, c < 0, o);
        if(pos == nn - 1) break;
        run = 0, num_sig--;
        sb_bin(s, XEVE_HIP_CTX_LAST + (ch ? 1 : 0), num_sig == 0, o);
        if(num_sig == 0) break;
    }
}

// xeve_eco_unit (xeve_eco.c:1431-1640) of the CU at (x, y), unit index cup inside the CTU
template <bool WAVE = false, class D = xeve_hip_ctu_data> XL void eco_unit(const EcoParams &E, Sbac &s, const D &d, uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, int x0, int y0,
                 int x, int y, int log2, int cup, Sink *o)
{
    const int idc = E.idc, ws = idc <= 2, hs = idc <= 1, cu = 1 << log2, st = E.slice_type, mode = d.pred_mode[cup], skip = mode == 2 /* MODE_SKIP */;
    const int ctu = 1 << E.log2_ctu, lx = x - x0, ly = y - y0;
    if(st != 2) {
        sb_bin(s, XEVE_HIP_CTX_SKIP_FLAG, skip, o); // (ctx_flags: 0 without sps_cm_init_flag, xeve_util.c:1181-1288)
        if(skip) {
            eco_mvp_idx(s, d.mvp_idx[cup][0], o);
            if(st == 0) eco_mvp_idx(s, d.mvp_idx[cup][1], o);
        }
        else {
            sb_bin(s, XEVE_HIP_CTX_PRED_MODE, mode == 0, o);
            if(mode != 0) {
                if(st == 0) sb_bin(s, XEVE_HIP_CTX_DIRECT, mode == 3 /* MODE_DIR */, o);
                if(mode != 3) {
                    const int r0 = d.refi[cup][0], r1 = d.refi[cup][1];
                    if(st == 0) { // xeve_eco_inter_pred_idc (xeve_eco.c:1123-1156)
                        if(r0 >= 0 && r1 >= 0) sb_bin(s, XEVE_HIP_CTX_INTER_DIR, 0, o);
                        else sb_bin(s, XEVE_HIP_CTX_INTER_DIR, 1, o), sb_bin(s, XEVE_HIP_CTX_INTER_DIR + 1, r0 >= 0 ? 0 : 1, o);
                    }
                    if(r0 >= 0) eco_refi(s, E.num_refp[0], r0, o), eco_mvp_idx(s, d.mvp_idx[cup][0], o), eco_mvd1(s, d.mvd[cup][0][0], o), eco_mvd1(s, d.mvd[cup][0][1], o);
                    if(st == 0 && r1 >= 0) eco_refi(s, E.num_refp[1], r1, o), eco_mvp_idx(s, d.mvp_idx[cup][1], o), eco_mvd1(s, d.mvd[cup][1][0], o), eco_mvd1(s, d.mvd[cup][1][1], o);
                }
            }
        }
    }
    if(mode == 0) { // xeve_get_mpm from the live maps (the units written so far are coded), xeve_eco_intra_dir
        const int x_scu = x >> 2, y_scu = y >> 2, scup = y_scu * E.w_scu + x_scu;
        int l = 0, u = 0;
        if(x_scu > 0 && XL_IF(map_scu[scup - 1]) && XL_COD(map_scu[scup - 1]) && map_tidx[scup] == map_tidx[scup - 1]) l = map_ipm[scup - 1] + 1;
        if(y_scu > 0 && XL_IF(map_scu[scup - E.w_scu]) && XL_COD(map_scu[scup - E.w_scu]) && map_tidx[scup] == map_tidx[scup - E.w_scu]) u = map_ipm[scup - E.w_scu] + 1;
        sb_unary2(s, (unsigned)mpm_rank(l, u, xl_ipm0(d, cup)), XEVE_HIP_CTX_INTRA_DIR, o);
    }
    int nnz[3] = {0, 0, 0};
    if(!skip) { // xeve_eco_coef(RUN_L | RUN_CB | RUN_CR): xeve_eco_cbf (xeve_eco.c:793-894), then the levels of the coded components
        for(int c = 0; c < 3; c++) nnz[c] = d.nnz[c][cup];
        const int cb = nnz[1] != 0, cr = nnz[2] != 0, cy = nnz[0] != 0;
        int coded = 1;
        if(mode != 0) {
            const int cbf_all = cy + cb + cr;
            sb_bin(s, XEVE_HIP_CTX_CBF_ALL, cbf_all != 0, o);
            if(!cbf_all) coded = 0;
            else {
                if(idc) sb_bin(s, XEVE_HIP_CTX_CBF_CB, cb, o), sb_bin(s, XEVE_HIP_CTX_CBF_CR, cr, o);
                if(cb + cr != 0) sb_bin(s, XEVE_HIP_CTX_CBF_LUMA, cy, o);
            }
        }
        else {
            if(idc) sb_bin(s, XEVE_HIP_CTX_CBF_CB, cb, o), sb_bin(s, XEVE_HIP_CTX_CBF_CR, cr, o);
            sb_bin(s, XEVE_HIP_CTX_CBF_LUMA, cy, o);
        }
        if(coded) {
            if(nnz[0]) eco_levels<WAVE>(s, E, xl_coef(d, 0) + ly * ctu + lx, ctu, log2, nnz[0], 0, o);
            for(int c = 1; c < 3; c++)
                if(nnz[c]) eco_levels<WAVE>(s, E, xl_coef(d, c) + (ly >> hs) * (ctu >> ws) + (lx >> ws), ctu >> ws, log2 - ws, nnz[c], 1, o);
        }
    }
    for(int j = 0; j < cu >> 2; j++)
        for(int i = 0; i < cu >> 2; i++) {
            const int g = ((y >> 2) + j) * E.w_scu + (x >> 2) + i;
            uint32_t  m = map_scu[g];
            m = skip ? m | (1u << 23) : m & ~(1u << 23);       // MCU_SET_SF / CLR_SF
            m = nnz[0] > 0 ? m | (1u << 24) : m & ~(1u << 24); // MCU_SET_CBFL / CLR_CBFL: core->nnz_sub[Y_C][0]
            map_scu[g] = m | (1u << 31);                       // MCU_SET_COD
            map_cu_mode[g] = (map_cu_mode[g] & 0x00FFFFFFu) | ((uint32_t)log2 << 24) | ((uint32_t)log2 << 28);
        }
}

// xeve_eco_tree of the CTU at (x0, y0): the coded flags of its units reset first, then the tree in z order (an explicit stack: the walk is five levels deep at most)
template <bool WAVE = false, class D = xeve_hip_ctu_data> XL void eco_ctu(const EcoParams &E, Sbac &s, const D &d, uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, int x0, int y0, Sink *o)
{
    const int pitch = 1 << (E.log2_ctu - 2), ctu = 1 << E.log2_ctu;
    {
        const int w = ((x0 + ctu > E.pic_w ? E.pic_w - x0 : ctu) >> 2), h = ((y0 + ctu > E.pic_h ? E.pic_h - y0 : ctu) >> 2);
        for(int j = 0; j < h; j++)
            for(int i = 0; i < w; i++) map_scu[((y0 >> 2) + j) * E.w_scu + (x0 >> 2) + i] &= 0x7FFFFFFFu; // MCU_CLR_COD
    }
    struct Frame {
        int x, y, log2, cud, cup, part;
    } st[6];
    int sp = 0;
    st[0].x = x0, st[0].y = y0, st[0].log2 = E.log2_ctu, st[0].cud = 0, st[0].cup = 0, st[0].part = -1;
    while(sp >= 0) {
        Frame &f = st[sp];
        const int cu = 1 << f.log2, half = cu >> 1;
        if(f.part < 0) {
            const int split = cu >= 8 ? d.split_mode[f.cud][f.cup + (half >> 2) * pitch + (half >> 2)] : 0; // xeve_get_split_mode (xeve_util.c:1125-1144)
            if(!split) {
                if(cu > 4) sb_bin(s, XEVE_HIP_CTX_SPLIT_CU, 0, o);
                eco_unit<WAVE, D>(E, s, d, map_scu, map_ipm, map_tidx, map_cu_mode, x0, y0, f.x, f.y, f.log2, f.cup, o);
                sp--;
                continue;
            }
            sb_bin(s, XEVE_HIP_CTX_SPLIT_CU, 1, o); // (always coded without sps_btt_flag, also where the picture edge implies the split)
            f.part = 0;
        }
        if(f.part == 4) {
            sp--;
            continue;
        }
        const int part = f.part++, xp = f.x + (part & 1) * half, yp = f.y + (part >> 1) * half;
        if(xp < E.pic_w && yp < E.pic_h) {
            Frame &c = st[sp + 1];
            c.x = xp, c.y = yp, c.log2 = f.log2 - 1, c.cud = f.cud + 2, c.cup = f.cup + (part & 1) * (half >> 2) + (part >> 1) * (half >> 2) * pitch, c.part = -1;
            sp++;
        }
    }
}
// The end of a tile on the writer's coder: xeve_eco_tile_end_flag(bs, 1) = xeve_sbac_encode_bin_trm (xeve_eco.c:577-595), then xeve_sbac_finish (:622-672): what the coder
// still held, and -- where no pending byte is left and fewer than four code bits remain -- the zero bits up to the byte boundary (one zero byte)
XL void eco_tile_end(Sbac &s, Sink *o)
{
    s.bin_counter++;
    s.range--;
    s.code += s.range, s.range = 1; // the terminating bin is 1
    while(s.range < 8192) s.range <<= 1, sb_shift(s, o);
    uint32_t tmp = (s.code + s.range - 1) & (0xFFFFFFFFu << 14);
    if(tmp < s.code) tmp += 8192;
    s.code = tmp << s.code_bits;
    for(int pass = 0; pass < 2; pass++) { // sbac_carry_propagate, twice (the second time after code <<= 8)
        if(pass) s.code <<= 8;
        const unsigned out = s.code >> 17;
        s.code &= (1u << 17) - 1;
        if(out < 0xFF) {
            for(; s.stacked_ff; s.stacked_ff--) sb_byte(s, 0xFF, o);
            sb_byte(s, out, o);
        }
        else if(out > 0xFF) {
            s.pending_byte++;
            for(; s.stacked_ff; s.stacked_ff--) sb_byte(s, 0, o);
            sb_byte(s, out, o);
        }
        else s.stacked_ff++;
    }
    auto put = [&](unsigned b) {
        if(o->n < o->cap) o->p[o->n] = (uint8_t)b;
        o->n++;
    };
    for(; s.stacked_zero; s.stacked_zero--) put(0);
    if(s.pending_byte != 0) put(s.pending_byte);
    else if(s.code_bits < 4) put(0);
}
} // namespace xl
