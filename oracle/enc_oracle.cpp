// oracle/enc_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// The product's frame loop (xeve_amd/csrc/enc_host.h + enc_plan.h: which picture when, reference lists, QPs and lambdas, row chains, NAL units) instantiated with a
// CPU engine that keeps the pictures in host memory and lets the oracle (xeve_oracle.c, pinned to the reference) decide and write every CTU.  What comes out is held
// against bitstreams of the unmodified reference application (tests/test_enc_host.py): that pins the host side of the batch encoder without a GPU.  The product never
// links this file; libxeve_hip.so instantiates the same template with its HIP engine (xeve_amd/csrc/encode.cpp).
#include <pthread.h>
#include <cstdlib>
#include <memory>
#include <thread>
#define XENC_TEST_OVERRIDES 1 // (enc_plan.h: XO_PIN_* take a preset apart)
#include "../xeve_amd/csrc/enc_host.h"
#include "../xeve_amd/csrc/walk_setup.h" // the fused CTU walk's host side (XO_ENC_WALK=1: every CTU decided by it instead of the oracle -- pins walk.h end to end)
extern "C" {
#include "xeve_oracle.h"
}
static int g_use_walk = -1;
static bool use_walk()
{
    if(g_use_walk < 0) g_use_walk = getenv("XO_ENC_WALK") ? atoi(getenv("XO_ENC_WALK")) : 0;
    return g_use_walk != 0;
}

using namespace xenc;

static_assert(sizeof(xo_tree_params) == sizeof(xeve_hip_tree_params) && sizeof(xo_ctu_data) == sizeof(xeve_hip_ctu_data) && sizeof(xo_sbac) == sizeof(xeve_hip_sbac) &&
                  sizeof(xo_rdo_params) == sizeof(xeve_hip_rdo_params) && sizeof(xo_me_params) == sizeof(xeve_hip_me_params) &&
                  sizeof(xo_deblock_params) == sizeof(xeve_hip_deblock_params) && sizeof(xo_refpic) == sizeof(xeve_hip_refpic),
              "the oracle's records mirror the library's");
static void to_oracle(xo_inter_params &o, const xeve_hip_inter_params &h) // (the oracle's search parameters carry the sub-pel stage as a record of its own)
{
    memset(&o, 0, sizeof(o));
    memcpy(&o.rdo, &h.rdo, sizeof(o.rdo)), memcpy(&o.me.me, &h.me.me, sizeof(o.me.me));
    o.me.spel.lambda_mv = h.me.me.lambda_mv, o.me.spel.hpel_cnt = h.me.hpel_cnt, o.me.spel.qpel_cnt = h.me.qpel_cnt;
    memcpy(o.refi_bits, h.refi_bits, sizeof(o.refi_bits)), memcpy(o.range_recentre, h.range_recentre, sizeof(o.range_recentre));
    o.max_cand = h.max_cand, o.poc = h.poc, o.col_list_poc0 = h.col_list_poc0, o.skip_th = h.skip_th;
}

namespace {
struct Store { // one picture store: padded planes + the motion maps kept with the picture
    std::vector<xo_pel> y, u, v;
    std::vector<int16_t> mv;   // [unit][list][x, y]
    std::vector<int8_t>  refi; // [unit][list]
};
struct Gop {
    std::vector<xo_pel>   org[3];
    std::vector<Store>    st;
    std::vector<uint32_t> scu, cum;
    std::vector<int8_t>   ipm;
    std::vector<uint8_t>  tidx;
    std::vector<xo_ctu_data> ctus;
    std::vector<xo_sbac>  chain;
    std::vector<uint8_t>  first_pass;
};
struct CpuEngine {
    Param P;
    int   G, F, nslots, w_scu, h_scu, nscu, s_l, s_c, w_lcu, h_lcu;
    const uint8_t *const *yuv; // [G]: F frames each
    std::vector<Gop> gop;
    PicSetup S;

    CpuEngine(const Param &p, int g, int f, int slots, const uint8_t *const *in) : P(p), G(g), F(f), nslots(slots), yuv(in)
    {
        w_scu = P.w >> 2, h_scu = P.h >> 2, nscu = w_scu * h_scu, s_l = P.w + 2 * PAD_L, s_c = P.w / 2 + 2 * PAD_C, w_lcu = (P.w + 63) / 64, h_lcu = (P.h + 63) / 64;
        gop.resize(G);
        for(Gop &q : gop) {
            q.org[0].resize((size_t)P.w * P.h), q.org[1].resize((size_t)P.w * P.h / 4), q.org[2].resize((size_t)P.w * P.h / 4);
            q.st.resize(nslots);
            for(Store &s : q.st) {
                s.y.assign((size_t)s_l * (P.h + 2 * PAD_L), 0), s.u.assign((size_t)s_c * (P.h / 2 + 2 * PAD_C), 0), s.v = s.u;
                s.mv.assign((size_t)nscu * 4, 0), s.refi.assign((size_t)nscu * 2, -1);
            }
            q.scu.assign(nscu, 0), q.cum.assign(nscu, 0), q.ipm.assign(nscu, 0), q.tidx.assign(nscu, 0);
            q.ctus.resize((size_t)w_lcu * h_lcu);
            q.chain.resize(8);
        }
    }
    xo_pel *plane0(Store &s, int c) { return c == 0 ? s.y.data() + (size_t)PAD_L * s_l + PAD_L : (c == 1 ? s.u.data() : s.v.data()) + (size_t)PAD_C * s_c + PAD_C; }

    void begin_picture(const PicSetup &setup)
    {
        S = setup;
        for(int g = 0; g < G; g++) {
            Gop &q = gop[g];
            const uint8_t *f = yuv[g] + (size_t)S.frame * P.frame_bytes();
            const bool     wide = P.input_depth > 8; // -d 10: 16-bit little-endian samples, copied as they are; -d 8: the application's 8 -> 10 bit conversion (imgb_cpy_conv_8b_to_16b)
            size_t         at = 0;
            auto sample = [&](size_t i) { return wide ? (xo_pel)(f[2 * i] | (f[2 * i + 1] << 8)) : (xo_pel)(f[i] << (BIT_DEPTH - 8)); };
            for(int c = 0; c < 3; c++) {
                for(size_t i = 0; i < q.org[c].size(); i++) q.org[c][i] = sample(at + i);
                at += q.org[c].size();
            }
            std::fill(q.scu.begin(), q.scu.end(), 0u), std::fill(q.cum.begin(), q.cum.end(), 0u); // xeve_pic_prepare (:1236-1237)
            Store &cur = q.st[S.cur_slot];
            std::fill(cur.mv.begin(), cur.mv.end(), (int16_t)0), std::fill(cur.refi.begin(), cur.refi.end(), (int8_t)-1); // (:1220-1225)
            q.first_pass.clear();
        }
    }
    void reset_chain(int t)
    {
        for(Gop &q : gop) xo_sbac_reset(&q.chain[t]);
    }
    void step(const ChainCtu *c, int n)
    {
        std::vector<uint8_t> buf(1 << 18);
        for(int g = 0; g < G; g++) {
            Gop   &q   = gop[g];
            Store &cur = q.st[S.cur_slot];
            const xo_pel *org[3] = {q.org[0].data(), q.org[1].data(), q.org[2].data()};
            xo_pel       *mod[3] = {plane0(cur, 0), plane0(cur, 1), plane0(cur, 2)};
            xo_refpic tab[2 * MAX_ACTIVE_REF];
            memset(tab, 0, sizeof(tab));
            xo_tree_inter TI;
            const bool inter = S.slice_type != ST_I;
            if(inter) {
                to_oracle(TI.ipar, S.ti.ipar);
                for(int l = 0; l < 2; l++)
                    for(int r = 0; r < S.nref[l]; r++) {
                        Store &rs = q.st[S.ref[r][l].slot];
                        tab[r * 2 + l].y = plane0(rs, 0), tab[r * 2 + l].u = plane0(rs, 1), tab[r * 2 + l].v = plane0(rs, 2), tab[r * 2 + l].poc = S.ref[r][l].poc;
                    }
                if(S.slice_type == ST_P) tab[1] = tab[0];
                TI.refp = tab, TI.s_ref_l = s_l, TI.s_ref_c = s_c, TI.ecu_depth = S.ti.ecu_depth, TI.pad_ = 0;
                TI.map_mv = (int16_t(*)[2][2])cur.mv.data(), TI.map_refi = (int8_t(*)[2])cur.refi.data();
                TI.col0 = (const int16_t(*)[2][2])q.st[S.ref[0][0].slot].mv.data();
                TI.col1 = S.slice_type == ST_B ? (const int16_t(*)[2][2])q.st[S.ref[0][1].slot].mv.data() : TI.col0;
            }
            const int num_refp[2] = {S.ep.num_refp[0], S.ep.num_refp[1]};
            std::vector<xeve_hip_ctu_data> wout;
            if(use_walk()) { // the step's chains of this picture in ONE call of the fused walk (team-local lockstep, the chains share the picture and the maps)
                static xw::Tables T;
                if(T.dct.empty()) xw::make_tables(T);
                xeve_hip_tree_inter HI = S.ti;
                xeve_hip_refpic htab[2 * MAX_ACTIVE_REF];
                memcpy(htab, tab, sizeof(htab));
                if(inter) {
                    HI.refp = htab, HI.s_ref_l = s_l, HI.s_ref_c = s_c, HI.map_mv = cur.mv.data(), HI.map_refi = cur.refi.data();
                    HI.col_mv0 = q.st[S.ref[0][0].slot].mv.data(), HI.col_mv1 = S.slice_type == ST_B ? q.st[S.ref[0][1].slot].mv.data() : HI.col_mv0;
                    HI.coef_l = xo_mc_l_coeff, HI.coef_c = xo_mc_c_coeff;
                }
                std::vector<xeve_hip_ctu_job> jobs(n);
                for(int i = 0; i < n; i++) jobs[i].x = c[i].x * CTU, jobs[i].y = c[i].y * CTU, jobs[i].sbac = c[i].t, jobs[i].pic = 0;
                wout.resize(n);
                std::vector<xeve_hip_sbac> nxt(n);
                std::vector<double> cost(n);
                xw::P wp;
                xw::fill_params(wp, (const xeve_hip_pel *const *)org, P.w, P.w / 2, (xeve_hip_pel *const *)mod, s_l, s_c, q.scu.data(), q.ipm.data(), q.tidx.data(), q.cum.data(), nullptr,
                                (const xeve_hip_sbac *)q.chain.data(), &S.tp, inter ? &HI : nullptr, jobs.data(), n, wout.data(), nxt.data(), cost.data(), 0);
                const std::vector<xw::Op> ops = xw::make_ops(&S.tp, inter);
                std::vector<xw::Cw> cw((size_t)n);
                wp.C = n < XW_MAXC ? n : XW_MAXC, wp.full = 0;
                wp.entropy = T.entropy.data(), wp.dct = T.dct.data(), wp.scan = T.scan.data(), wp.ops = ops.data(), wp.nops = (int)ops.size(), wp.cw = cw.data();
                if(inter) wp.mc_l = &xo_mc_l_coeff[0][0], wp.mc_c = &xo_mc_c_coeff[0][0];
                std::unique_ptr<xw::Lds> lds(new xw::Lds());
                static const int nt = getenv("XO_ENC_WALK_THREADS") ? atoi(getenv("XO_ENC_WALK_THREADS")) : 1; // (tests/test_walk_race.py: the team as real threads)
                if(nt <= 1) {
                    const xw::Tm tm = {0, 1};
                    for(int team = 0; team * wp.C < n; team++) xw::walk_team<false>(tm, wp, *lds, team);
                }
                else {
                    pthread_barrier_t bar;
                    pthread_barrier_init(&bar, nullptr, (unsigned)nt);
                    std::vector<std::thread> th;
                    for(int t = 0; t < nt; t++)
                        th.emplace_back([&, t] {
                            xw::host_team() = {[](void *b) { pthread_barrier_wait((pthread_barrier_t *)b); }, &bar};
                            const xw::Tm tm = {t, nt};
                            for(int team = 0; team * wp.C < n; team++) {
                                xw::walk_team<false>(tm, wp, *lds, team);
                                pthread_barrier_wait(&bar);
                            }
                            xw::host_team() = {nullptr, nullptr};
                        });
                    for(auto &t : th) t.join();
                    pthread_barrier_destroy(&bar);
                }
            }
            for(int i = 0; i < n; i++) {
                const int x0 = c[i].x * CTU, y0 = c[i].y * CTU;
                xo_sbac   next;
                xo_ctu_data &out = q.ctus[c[i].lcu];
                if(use_walk()) memcpy(&out, &wout[i], sizeof(out));
                else {
                    if(S.tp.rdo_dbk) { // rdo_dbk_switch: the candidates' distortions include the loop filter's share, read off the reconstruction so far and the unit maps
                        xo_dbk_ctx D;
                        memset(&D, 0, sizeof(D));
                        D.mod[0] = mod[0], D.mod[1] = mod[1], D.mod[2] = mod[2], D.s_mod_l = s_l, D.s_mod_c = s_c, D.qp = S.tp.slice_qp;
                        D.map_scu = q.scu.data(), D.map_refi = cur.refi.data(), D.map_mv = cur.mv.data(), D.map_tidx = q.tidx.data(), D.dp = (const xo_deblock_params *)&S.dp;
                        xo_rdo_dbk_begin(&D);
                    }
                    (void)xo_mode_analyze_ctu(org, P.w, P.w / 2, mod, s_l, s_c, q.scu.data(), q.ipm.data(), q.tidx.data(), q.cum.data(), &q.chain[c[i].t],
                                              (const xo_tree_params *)&S.tp, inter ? &TI : nullptr, x0, y0, &out, &next);
                    if(S.tp.rdo_dbk) xo_rdo_dbk_end();
                }
                for(int j = 0; j < std::min((int)CTU, P.h - y0) >> 2; j++) // mode_analyze_lcu's tail: the CTU's coded flags reset (xeve_mode.c:2591-2607)
                    for(int k = 0; k < std::min((int)CTU, P.w - x0) >> 2; k++) q.scu[(size_t)((y0 >> 2) + j) * w_scu + (x0 >> 2) + k] &= 0x7FFFFFFFu;
                const int nb = xo_eco_ctu(&q.chain[c[i].t], &out, (const xo_tree_params *)&S.tp, num_refp, q.scu.data(), q.ipm.data(), q.tidx.data(), q.cum.data(), x0, y0,
                                          buf.data(), (int)buf.size());
                if(c[i].t == 0) q.first_pass.insert(q.first_pass.end(), buf.begin(), buf.begin() + std::min<size_t>(nb, buf.size()));
            }
        }
    }
    std::vector<std::vector<uint8_t>> slice;
    std::vector<uint32_t> bins;
    void collect(std::vector<std::vector<uint8_t>> &s, std::vector<uint32_t> &b) { s = slice, b = bins; }
    void end_picture(bool rewrite)
    {
        slice.assign(G, std::vector<uint8_t>()), bins.assign(G, 0);
        std::vector<uint8_t> buf(1 << 18);
        const int num_refp[2] = {S.ep.num_refp[0], S.ep.num_refp[1]};
        for(int g = 0; g < G; g++) {
            Gop   &q   = gop[g];
            Store &cur = q.st[S.cur_slot];
            xo_deblock_picture(plane0(cur, 0), plane0(cur, 1), plane0(cur, 2), s_l, s_c, q.scu.data(), q.cum.data(), cur.refi.data(), cur.mv.data(), (const xo_deblock_params *)&S.dp);
            xo_sbac w;
            if(rewrite) {
                for(uint32_t &m : q.scu) m &= 0x7FFFFFFFu; // MCU_CLR_COD over the picture (xeve_enc.c:466-468)
                xo_sbac_reset(&w);
                for(int lcu = 0; lcu < w_lcu * h_lcu; lcu++) {
                    const int nb = xo_eco_ctu(&w, &q.ctus[lcu], (const xo_tree_params *)&S.tp, num_refp, q.scu.data(), q.ipm.data(), q.tidx.data(), q.cum.data(),
                                              (lcu % w_lcu) * CTU, (lcu / w_lcu) * CTU, buf.data(), (int)buf.size());
                    slice[g].insert(slice[g].end(), buf.begin(), buf.begin() + nb);
                }
            }
            else w = q.chain[0], slice[g] = q.first_pass;
            const int nb = xo_eco_tile_end(&w, buf.data(), (int)buf.size());
            slice[g].insert(slice[g].end(), buf.begin(), buf.begin() + nb);
            bins[g] = w.bin_counter;
            xo_picbuf_expand(plane0(cur, 0), s_l, P.w, P.h, PAD_L), xo_picbuf_expand(plane0(cur, 1), s_c, P.w / 2, P.h / 2, PAD_C);
            xo_picbuf_expand(plane0(cur, 2), s_c, P.w / 2, P.h / 2, PAD_C);
        }
    }
};
} // namespace

extern "C" {
// 1: every CTU decided by the fused walk's host side (xeve_amd/csrc/walk.h) instead of the oracle; 0: the oracle; returns the previous setting
int xo_encode_use_walk(int on)
{
    const int was = use_walk();
    g_use_walk = on != 0;
    return was;
}
// out[g] is malloc'ed (release with xo_encode_free); returns 0, or -1 with a message in err
int xo_encode_gops(const xeve_hip_enc_config *cfg, const uint8_t *const *yuv, int ngops, int frames, int always_rewrite, uint8_t **out, size_t *out_bytes, char *err, int err_cap)
{
    Param P;
    auto  say = [&](const std::string &m) { if(err && err_cap > 0) snprintf(err, (size_t)err_cap, "%s", m.c_str()); return -1; };
    if(!P.finish(*cfg)) return say(P.error);
    CpuEngine               E(P, ngops, frames, BatchEncoder<CpuEngine>::slots_needed(P, frames), yuv);
    BatchEncoder<CpuEngine> enc(E, P, ngops, frames);
    enc.always_rewrite = always_rewrite != 0;
    std::vector<std::vector<uint8_t>> o;
    if(enc.run(o) != 0) return say(enc.error);
    for(int g = 0; g < ngops; g++) {
        out[g] = (uint8_t *)malloc(o[g].size() ? o[g].size() : 1);
        memcpy(out[g], o[g].data(), o[g].size()), out_bytes[g] = o[g].size();
    }
    return 0;
}
// the same run cut after every `pictures_per_slice` pictures with BatchEncoder::flush() at each cut: after[k] = bytes of GOP 0's bitstream after the k-th cut (cap entries,
// *ncuts of them used); out / out_bytes as xo_encode_gops.  What the product's xeve_hip_enc_flush does to the frame loop, pinned without a GPU: the final bytes must be
// the un-flushed run's and every cut a prefix of them.
int xo_encode_gops_flushed(const xeve_hip_enc_config *cfg, const uint8_t *const *yuv, int ngops, int frames, int pictures_per_slice, uint8_t **out, size_t *out_bytes,
                           size_t *after, int cap, int *ncuts, char *err, int err_cap)
{
    Param P;
    auto  say = [&](const std::string &m) { if(err && err_cap > 0) snprintf(err, (size_t)err_cap, "%s", m.c_str()); return -1; };
    if(!P.finish(*cfg)) return say(P.error);
    CpuEngine               E(P, ngops, frames, BatchEncoder<CpuEngine>::slots_needed(P, frames), yuv);
    BatchEncoder<CpuEngine> enc(E, P, ngops, frames);
    std::vector<std::vector<uint8_t>> o;
    if(enc.begin(o) != 0) return say(enc.error);
    const long per_picture = enc.total_steps() / frames;
    int n = 0;
    for(long left = enc.total_steps(); left > 0;) {
        left = enc.advance(per_picture * std::max(1, pictures_per_slice));
        if(left < 0 || enc.flush() != 0) return say(enc.error);
        if(n < cap) after[n++] = o[0].size();
        if(enc.flush() != 0) return say(enc.error); // (a second flush has nothing to do)
    }
    *ncuts = n;
    for(int g = 0; g < ngops; g++) {
        out[g] = (uint8_t *)malloc(o[g].size() ? o[g].size() : 1);
        memcpy(out[g], o[g].data(), o[g].size()), out_bytes[g] = o[g].size();
    }
    return 0;
}
void xo_encode_free(uint8_t *p) { free(p); }
// the frame loop alone: plan[i] = {frame, poc, slice type, temporal id, slice QP, idr, L0 POC or -1, L1 POC or -1} of the i-th coded picture; returns their number
int xo_encode_plan(const xeve_hip_enc_config *cfg, int frames, int32_t *plan, int cap)
{
    Param P;
    if(!P.finish(*cfg)) return -1;
    const std::vector<PicPlan> pics = Planner(P, frames).run();
    Dpb dpb(64);
    int last_intra = 0, n = 0;
    for(const PicPlan &pp : pics) {
        if(n >= cap) return -2;
        if(pp.slice_type == ST_I) last_intra = pp.poc;
        if(!dpb.refp_init(P.max_num_ref_pics(), pp.slice_type, pp.poc, pp.tid, last_intra)) return -3;
        const int slot = dpb.get_empty();
        if(slot < 0) return -4;
        int32_t *r = plan + 8 * n++;
        r[0] = pp.frame, r[1] = pp.poc, r[2] = pp.slice_type, r[3] = pp.tid, r[4] = slice_qp(P, pp.depth), r[5] = pp.idr;
        r[6] = pp.slice_type != ST_I ? dpb.refp[0][0].poc : -1, r[7] = pp.slice_type == ST_B ? dpb.refp[0][1].poc : -1;
        dpb.put(slot, pp.idr != 0, pp.poc, pp.tid, pp.ref_flag != 0, P.ref_pic_gap_length);
    }
    return n;
}
}
