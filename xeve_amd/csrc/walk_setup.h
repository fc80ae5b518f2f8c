// xeve_amd/csrc/walk_setup.h -- host side of walk.h: the tables (DCT-II matrices from the cosine formula, zig-zag scans, entropy_bits), the static schedule of one
// CTU and the parameter record, from the arguments of xeve_hip_mode_analyze_ctu_jobs.  Plain C++; used by walk.hip (device tables) and by the CPU test harness.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include "walk.h"

namespace xw {

struct Tables {
    std::vector<int8_t>   dct;     // [k][x] of sizes 2 .. 64, then [x][k] of the same
    std::vector<uint16_t> scan;    // zig-zag of the square blocks 2 .. 64
    std::vector<int32_t>  entropy; // entropy_bits[1024]
};
inline void make_tables(Tables &T)
{
    // EVC integer DCT-II: M_N[k][x] = +-g[fold((2x+1) * k * 64/N mod 256)], g[j] = round(64*sqrt(2)*cos(j*pi/128)) (xeve_tbl_tm2 .. tm64, xeve_tbl.c:83-236)
    int g[65];
    g[0] = 64, g[64] = 0;
    for(int j = 1; j < 64; j++) g[j] = (int)floor(64.0 * 1.4142135623730951 * cos(j * 3.14159265358979323846 / 128.0) + 0.5);
    T.dct.assign(2 * XW_DCT_ELEMS, 0), T.scan.assign(XW_DCT_ELEMS, 0);
    for(int l = 1; l <= 6; l++) {
        const int n = 1 << l;
        for(int k = 0; k < n; k++)
            for(int x = 0; x < n; x++) {
                int th = ((2 * x + 1) * k * (64 / n)) % 256, sg = 1;
                if(th > 128) th = 256 - th;
                if(th > 64) sg = -1, th = 128 - th;
                const int8_t v = (int8_t)(sg * g[th]);
                T.dct[dct_off(l) + k * n + x] = v, T.dct[XW_DCT_ELEMS + dct_off(l) + x * n + k] = v;
            }
        // anti-diagonals x + y = d; odd d run down-left, even d run up-right (xeve_util.c:1301-1325)
        uint16_t *s = T.scan.data() + dct_off(l);
        int pos = 0;
        for(int d = 0; d < 2 * n - 1; d++) {
            if(d & 1)
                for(int x = d < n - 1 ? d : n - 1, y = d - x; x >= 0 && y < n; x--, y++) s[pos++] = (uint16_t)(y * n + x);
            else
                for(int y = d < n - 1 ? d : n - 1, x = d - y; y >= 0 && x < n; x++, y--) s[pos++] = (uint16_t)(y * n + x);
        }
    }
    T.entropy.resize(1026);
    for(int i = 0; i < 1026; i++) { // xeve_init_bits_est (xeve_mode.c:304-313)
        const double pr = (512 * (i + 0.5)) / 1024;
        T.entropy[i] = (int32_t)(-32768 * (log(pr) / log(2.0) - 9));
    }
}

// a node of this size can be a CU at all: within max_cu and no larger than the picture
inline bool level_has_cu(const xeve_hip_tree_params *p, int log2) { return (1 << log2) <= p->max_cu && (1 << log2) <= p->pic_w && (1 << log2) <= p->pic_h; }
inline void ops_node(std::vector<Op> &v, const xeve_hip_tree_params *p, bool inter, int L, int part)
{
    auto add = [&](int op, int lvl, int pt) { Op o; o.op = (unsigned char)op, o.lvl = (unsigned char)lvl, o.part = (signed char)pt, o.pad_ = 0; v.push_back(o); };
    const int cu = 1 << (L + 2);
    add(OP_ENTER, L, part);
    if(level_has_cu(p, L + 2)) {
        if(inter) add(OP_INTER, L, 0), add(OP_MID, L, 0);
        add(OP_INTRA, L, 0);
    }
    add(OP_LEAF, L, 0);
    if(cu > 4 && cu > p->min_cu && cu > p->min_cuwh)
        for(int q = 0; q < 4; q++) {
            ops_node(v, p, inter, L - 1, q);
            add(OP_CHILD_DONE, L, q);
        }
    add(OP_EXIT, L, 0);
}
inline std::vector<Op> make_ops(const xeve_hip_tree_params *p, bool inter)
{
    std::vector<Op> v;
    ops_node(v, p, inter, p->log2_ctu - 2, -1);
    Op o;
    o.op = OP_ROOT_DONE, o.lvl = (unsigned char)(p->log2_ctu - 2), o.part = 0, o.pad_ = 0;
    v.push_back(o);
    return v;
}

// everything of P that follows from the call's arguments (pointers to tables, schedule and workspace are the caller's to fill); refp: host table [refi * 2 + list]
inline void fill_params(P &q, const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu,
                        int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, const int64_t *pic_elems, const xeve_hip_sbac *states,
                        const xeve_hip_tree_params *p, const xeve_hip_tree_inter *I, const xeve_hip_ctu_job *jobs, int nchains, xeve_hip_ctu_data *out,
                        xeve_hip_sbac *next_best, double *cost, int vh)
{
    static const int q_scale[6] = {26214, 23302, 20560, 18396, 16384, 14764}, dq_scale[6] = {40, 45, 51, 57, 64, 71}; // xeve_quant_scale[0] (xeve_tq.c:37), xeve_tbl_dq_scale_b (xeve_tbl.c:237)
    memset(&q, 0, sizeof(q));
    const xeve_hip_intra_params &ip = p->ip;
    const int idc = ip.chroma_format_idc, bd = ip.bit_depth;
    q.nchains = nchains, q.log2_ctu = p->log2_ctu, q.pic_w = p->pic_w, q.pic_h = p->pic_h, q.w_scu = ip.w_scu, q.h_scu = ip.h_scu, q.max_cu = p->max_cu, q.min_cu = p->min_cu;
    q.min_cuwh = p->min_cuwh, q.idc = idc, q.ws = idc <= 2, q.hs = idc <= 1, q.bd = bd, q.slice_type = ip.slice_type, q.slice_qp = p->slice_qp, q.slice_num = p->slice_num;
    q.cip = ip.constrained_intra_pred != 0, q.rdo_dbk = p->rdo_dbk != 0;
    q.s_org_l = s_org_l, q.s_org_c = s_org_c, q.s_mod_l = s_mod_l, q.s_mod_c = s_mod_c;
    q.org_pic_l = pic_elems ? pic_elems[0] : 0, q.org_pic_c = pic_elems ? pic_elems[1] : 0, q.mod_pic_l = pic_elems ? pic_elems[2] : 0;
    q.mod_pic_c = pic_elems ? pic_elems[3] : 0, q.map_pic = pic_elems ? pic_elems[4] : 0;
    for(int c = 0; c < 3; c++) {
        const int v = ip.qp[c];
        q.qp[c] = v, q.q_scale[c] = q_scale[v % 6], q.dq_scale[c] = dq_scale[v % 6] << (v / 6), q.lambda[c] = ip.lambda[c];
        for(int l = 1; l <= 6; l++) { // ctx->err_scale[qp % 6][log2_size - 1], xeve_init_err_scale (xeve_tq.c:406-423)
            const int tr_shift = 15 - bd - l;
            double e = (double)(1 << 15) * pow(2.0, -tr_shift);
            e = e / q_scale[v % 6] / (1 << (bd - 8));
            q.err_scale[c][l] = (int64_t)(e * (double)(1 << 20));
        }
    }
    q.sqrt_lambda0 = ip.sqrt_lambda0, q.wgt[0] = ip.dist_chroma_weight[0], q.wgt[1] = ip.dist_chroma_weight[1];
    for(int c = 0; c < 3; c++) q.org[c] = org[c], q.mod[c] = mod[c];
    q.map_scu = map_scu, q.map_cu_mode = map_cu_mode, q.map_ipm = map_ipm, q.map_tidx = map_tidx, q.states = states, q.jobs = jobs, q.out = out, q.out_next = next_best;
    q.out_cost = cost;
    if(I) {
        const xeve_hip_inter_params &ep = I->ipar;
        q.inter = 1, q.isb = ip.slice_type == 0, q.ecu_depth = I->ecu_depth, q.vh = vh, q.nref[0] = ep.rdo.num_refp[0], q.nref[1] = q.isb ? ep.rdo.num_refp[1] : 0;
        q.max_cand = ep.max_cand, q.poc = ep.poc, q.col_list_poc0 = ep.col_list_poc0, q.s_ref_l = I->s_ref_l, q.s_ref_c = I->s_ref_c, q.me = ep.me, q.skip_th = ep.skip_th;
        memcpy(q.refi_bits, ep.refi_bits, sizeof(q.refi_bits)), memcpy(q.range_recentre, ep.range_recentre, sizeof(q.range_recentre));
        const int nr = q.nref[0] > q.nref[1] ? q.nref[0] : q.nref[1];
        for(int r = 0; r < nr; r++)
            for(int l = 0; l < 2; l++) {
                const xeve_hip_refpic &e = I->refp[r * 2 + l];
                q.refp[r * 2 + l].y = e.y, q.refp[r * 2 + l].u = e.u, q.refp[r * 2 + l].v = e.v, q.refp[r * 2 + l].poc = e.poc;
            }
        q.map_mv = (int16_t(*)[2][2])I->map_mv, q.map_refi = (int8_t(*)[2])I->map_refi;
        q.col0 = (const int16_t(*)[2][2])I->col_mv0, q.col1 = (const int16_t(*)[2][2])(I->col_mv1 ? I->col_mv1 : I->col_mv0);
    }
}

} // namespace xw
