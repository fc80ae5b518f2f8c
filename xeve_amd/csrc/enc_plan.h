// xeve_amd/csrc/enc_plan.h -- the host side of the closed-GOP batch encoder that needs no device: the frame loop's decisions
// (which input frame is coded next, as what), the reference-picture bookkeeping, the per-picture parameters of the CTU walk, and the
// high-level syntax (NAL units, SPS / PPS / SEI, slice header).  Plain C++, compiled into libxeve_hip.so and -- with the CPU engine of
// oracle/enc_oracle.cpp -- into the test harness that pins all of it against bitstreams of the unmodified reference.
//
// reference (src_base/): the application's push / encode / bump loop (app/xeve_app.c:1180-1355), xeve_encode / xeve_push (xeve.c:111-146),
// xeve_enc (xeve_enc.c:602-640), decide_slice_type / decide_normal_gop (:989-1182), xeve_poc_derivation (xeve_util.c:250-281),
// xeve_picman_refp_init / xeve_picman_put_pic / pic_marking (xeve_picman.c:58-98, 271-392, 428-474), xeve_set_sh (xeve_enc.c:1463-1527),
// set_lambda (xeve_mode.c:660-677), pinter_set_complexity / xeve_pinter_init_lcu (xeve_pinter.c:1759-1771, 2087-2111),
// xeve_eco_nalu / _sps / _pps / _sh / _emitsei (xeve_eco.c:45-376), xeve_param2string (xeve_enc.c:2533-2673).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/xeve_hip.h"

namespace xenc __attribute__((visibility("hidden"))) { // (hidden: the test harness instantiates the same inline code in its own library, and the two must not bind to each other)

enum { ST_B = 0, ST_P = 1, ST_I = 2 };                                     // XEVE_ST_* (inc/xeve.h:168-172)
enum { NUT_NONIDR = 0, NUT_IDR = 1, NUT_SPS = 24, NUT_PPS = 25, NUT_SEI = 28 }; // inc/xeve.h:158-164
enum { PAD_L = 144, PAD_C = 72, BIT_DEPTH = 10, LOG2_CTU = 6, CTU = 64, MAX_ACTIVE_REF = 5, MAX_INBUF = 70 };

// ---- encoder parameters: xeve_param_init (xeve_enc.c:2290-2324) + xeve_param_apply_ppt_baseline (:2431-2531) + xeve_set_init_param (:2210-2288) -----------------------
struct Param {
    int w = 0, h = 0, fps_num = 30, fps_den = 1, qp = 32, keyint = 0, bframes = 15, closed_gop = 0, threads = 1, inter_slice_type = 0, ref = 0;
    int preset = 1; // 0 fast, 1 medium, 2 slow, 3 placebo
    int qp_cb_offset = 0, qp_cr_offset = 0; // --qp-cb-offset / --qp-cr-offset (sh->qp_u_offset / qp_v_offset, xeve_enc.c:1509-1510)
    int level_idc = 40, sei_info = 1; // --level-idc (sps->level_idc = 3 x, xeve_enc.c:1402) and --info (the SEI that lists the options, :1989)
    int input_depth = 8; // the application's -d: 8 = one byte per sample, 10 = 16-bit little-endian samples (both go to the codec's 10 bits, xeve_app.c:1153-1158)
    // derived
    int gop_size = 16, ref_pic_gap_length = 0, me_ref_num = 1, me_range = 64, me_sub = 2, me_sub_pos = 4, me_sub_range = 1, merge_num = 3, me_algo = 1, rdo_dbk = 0;
    int max_cu_intra = 32, min_cu_intra = 4, max_cu_inter = 64, min_cu_inter = 8, lookahead = 17;
    std::string error;

    bool finish(const xeve_hip_enc_config &c)
    {
        w = c.w, h = c.h, fps_num = c.fps_num, fps_den = c.fps_den, qp = c.qp, keyint = c.keyint, bframes = c.bframes, closed_gop = c.closed_gop != 0;
        threads = c.threads, inter_slice_type = c.inter_slice_type, ref = c.ref, preset = c.preset, input_depth = c.reserved[1] ? c.reserved[1] : 8;
        qp_cb_offset = c.reserved[2], qp_cr_offset = c.reserved[3];
        level_idc = ((c.reserved[0] >> 8) & 0xFF) ? ((c.reserved[0] >> 8) & 0xFF) : 40, sei_info = (c.reserved[0] & 2) ? 0 : 1;
        auto bad = [&](const char *m) { error = m; return false; };
        if(w <= 0 || h <= 0 || (w & 7) || (h & 7)) return bad("picture size must be a positive multiple of 8 in both directions");
        if(w > 8192 || h > 4320) return bad("picture larger than 8192x4320");
        if(qp < 0 || qp > 51 || keyint < 0 || threads < 1 || threads > 8 || fps_num <= 0 || fps_den <= 0) return bad("qp / keyint / threads / fps out of range");
        if(!(bframes == 0 || bframes == 1 || bframes == 3 || bframes == 7 || bframes == 15)) return bad("bframes must be 0, 1, 3, 7 or 15");
        if(bframes && !closed_gop && keyint % (bframes + 1) != 0) return bad("an open GOP needs keyint to be a multiple of bframes + 1");
        if(input_depth != 8 && input_depth != 10) return bad("input depth must be 8 or 10 bits");
        // (xeve_ctu_mt_core waits for the CTU up-right only in front of a row's last column, xeve_enc.c:130-133: in a picture ONE CTU wide no row waits for the row
        // above at all, and the reference's own output changes from run to run -- 5 different bitstreams in 12 runs of 64x200 -m 3 -- so there is nothing to reproduce)
        if(threads > 1 && w <= CTU) return bad("a picture one CTU wide must be coded with threads = 1: the reference's row threads race there");
        // P slices (--inter-slice-type 1) and chroma qp offsets (--qp-cb-offset / --qp-cr-offset): options the reference application lists and fails to parse.  The
        // host logic for both follows the reference's sources and is held to the reference LIBRARY run with those parameters (oracle/ref_param_pin.c sets them on the
        // way into xeve_create): tests/test_enc_host.py on the CPU harness, tests/test_walk_host.py with every CTU decided by the fused walk's host side, and -- since
        // round 5 -- tests/test_enc_gpu.py on the device with both walks.
        if(inter_slice_type != 0 && inter_slice_type != 1) return bad("inter_slice_type must be 0 (B) or 1 (P)");
        if(qp_cb_offset < -12 || qp_cb_offset > 12 || qp_cr_offset < -12 || qp_cr_offset > 12) return bad("chroma qp offsets must lie in -12 .. 12");
        if(preset == 0) me_range = 32, me_sub_pos = 2, merge_num = 2;
        else if(preset == 1) me_range = 64, me_sub_pos = 4, merge_num = 3;
        else if(preset == 2) me_range = 128, me_sub = 3, me_sub_pos = 4, me_sub_range = 2, merge_num = 3, rdo_dbk = 1; // slow (xeve_enc.c:2473-2489): quarter-pel search, rdo_dbk_switch
        else if(preset == 3) // placebo (xeve_enc.c:2490-2506): 64x64 intra CUs in I slices, 4x4 CUs in inter slices, two reference pictures per list, the raster search
            max_cu_intra = 64, min_cu_inter = 4, me_ref_num = 2, me_algo = 2, me_range = 384, me_sub = 3, me_sub_pos = 8, me_sub_range = 3, merge_num = 4, rdo_dbk = 1;
        else return bad("preset must be 0 (fast), 1 (medium), 2 (slow) or 3 (placebo)");
        if(rdo_dbk && (qp_cb_offset || qp_cr_offset)) return bad("presets slow / placebo with chroma qp offsets: the loop filter's share of the chroma distortions is coded for offsets of 0 only (walk_dbk.h)");
#ifdef XENC_TEST_OVERRIDES // (the CPU harness only: a preset taken apart, against the reference library pinned the same way -- oracle/ref_param_pin.c)
        if(getenv("XO_PIN_RDO_DBK")) rdo_dbk = atoi(getenv("XO_PIN_RDO_DBK"));
        if(getenv("XO_PIN_ME_SUB")) me_sub = atoi(getenv("XO_PIN_ME_SUB"));
        if(getenv("XO_PIN_ME_RANGE")) me_range = atoi(getenv("XO_PIN_ME_RANGE"));
#endif
        // (the reference picture tables of the frame loop and of the device path hold 5 / 4 entries per list -- Dpb::refp, PicSetup::ref, the refi_bits table of
        // fill_inter_params, XEVE_HIP_MAX_REFP planes per search: more reference pictures than that are refused here, before anything indexes them)
        if(ref < 0 || ref > 4) return bad("ref must lie in 0 .. 4 (0: the preset's one reference picture per list)");
        if(closed_gop && keyint == 0 && bframes == 0) return bad("low-delay closed GOPs need keyint > 0 (the reference divides by the intra period there)");
        if(ref) me_ref_num = bframes == 0 ? std::min(5, ref) : std::min(ref, bframes);
        if(bframes == 0) ref_pic_gap_length = 1;
        gop_size  = bframes + 1;
        lookahead = std::min(std::max(0, 17), MAX_INBUF >> 1);
        return true;
    }
    long frame_bytes() const { return (long)w * h * 3 / 2 * (input_depth > 8 ? 2 : 1); } // one planar 4:2:0 input frame
    int max_num_ref_pics() const { return bframes > 0 ? me_ref_num : ref_pic_gap_length; } // xeve_set_sps (xeve_enc.c:1413-1419)
};

// ---- which picture is coded when, and as what -------------------------------------------------------------------------------------------------------------------------
struct PicPlan {
    int frame;      // input frame (of this encoder run) the picture codes: the frame PIC_ORIG points at
    int poc;        // ctx->poc.poc_val
    int slice_type; // ST_*
    int depth;      // ctx->slice_depth
    int tid;        // nuh_temporal_id
    int ref_flag;   // ctx->slice_ref_flag
    int idr;        // nal unit type IDR; parameter sets go in front of it (xeve_header, xeve_enc.c:1975-1993)
};

// The frame loop replayed: the application pushes a frame and asks for a picture until the frame delay is filled, then bumps (xeve_app.c:1180-1355); every call of
// xeve_enc decides the next picture from the counters below (decide_slice_type).  The names are the reference's context fields.
class Planner {
  public:
    Planner(const Param &p, int nframes) : P(p), N(nframes)
    {
        pico_max_cnt = (P.gop_size == 1 && P.keyint != 1) ? 2 : MAX_INBUF; // xeve_ready (xeve_enc.c:1693-1700)
        frm_rnum     = P.bframes ? P.bframes + 1 : 0;                      // (:1702-1707; use_fcst is 0 without rate control / AQ)
        slot_frame.assign(pico_max_cnt, -1), slot_used.assign(pico_max_cnt, 0);
    }
    std::vector<PicPlan> run()
    {
        std::vector<PicPlan> out;
        int pushed = 0;
        bool encoding = true;
        for(long guard = 0; guard < 100000; guard++) {
            if(encoding) {
                if(pushed >= N) { encoding = false, bump(); continue; }
                pic_icnt++; // xeve_push_frm (xeve_enc.c:658-663)
                slot_frame[pic_icnt % pico_max_cnt] = pic_icnt, slot_used[pic_icnt % pico_max_cnt] = 1;
                pushed++;
            }
            if(force_output) { // xeve_check_more_frames (:965-987)
                pic_icnt++;
                if(std::find(slot_used.begin(), slot_used.end(), 1) == slot_used.end()) break;
            }
            if(pic_icnt < frm_rnum) continue; // xeve_check_frame_delay: XEVE_OK_OUT_NOT_AVAILABLE (the application goes round without its end-of-loop test)
            out.push_back(encode_one());
            if(pushed >= N && encoding) encoding = false, bump();
        }
        return out;
    }

  private:
    const Param &P;
    int N;
    int pic_icnt = -1, pic_cnt = 0, ip_cnt = 0, pic_ticnt = 0, force_output = 0, force_slice = 0, force_ignored_cnt = 0, frm_rnum = 0, pico_max_cnt = 0;
    int poc_val = 0, prev_doc_offset = 0, prev_poc_val = 0;
    int slice_type = ST_I, slice_depth = 0, slice_ref_flag = 1, pico = 0;
    std::vector<int>  slot_frame;
    std::vector<char> slot_used;

    void bump() { force_output = 1, pic_ticnt = pic_icnt; } // setup_bumping -> XEVE_CFG_SET_FORCE_OUT (xeve.c:158-164)
    void pick(int idx) { pico = ((idx % pico_max_cnt) + pico_max_cnt) % pico_max_cnt; }

    void poc_derivation(int tid) // xeve_poc_derivation
    {
        const int log2_sub = (int)(std::log2((double)P.gop_size) + .5), sub = 1 << log2_sub;
        if(tid == 0) {
            poc_val = prev_poc_val + sub, prev_doc_offset = 0, prev_poc_val = poc_val;
            return;
        }
        int doc = (prev_doc_offset + 1) % sub, expected = 0;
        if(doc == 0) prev_poc_val += sub;
        else expected = 1 + (int)std::log2((double)doc);
        while(tid != expected) {
            doc      = (doc + 1) % sub;
            expected = doc == 0 ? 0 : 1 + (int)std::log2((double)doc);
        }
        poc_val         = prev_poc_val + (int)(sub * ((2.0 * doc + 1) / (double)(1 << tid) - 2));
        prev_doc_offset = doc;
    }
    static int depth_b(int gop_size, int pos) // xeve_tbl_slice_depth (xeve_tbl.c:546-563): the hierarchy level of the pos-th B picture of a sub-GOP in coding order
    {
        static const signed char d16[15] = {2, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5};
        return pos < gop_size - 1 && pos >= 0 ? d16[pos] : -1; // (the rows for 2, 4, 8 are prefixes of the row for 16)
    }
    static int depth_p(int gap, int i) // xeve_tbl_slice_depth_P (xeve_tbl.c:528-544), row gap >> 2
    {
        static const signed char r0[2] = {2, 1}, r1[4] = {3, 2, 3, 1}, r2[8] = {4, 3, 4, 2, 4, 3, 4, 1}, r4[16] = {5, 4, 5, 3, 5, 4, 5, 2, 5, 4, 5, 3, 5, 4, 5, 1};
        switch(gap >> 2) {
        case 0: return r0[i & 1];
        case 1: return r1[i & 3];
        case 2: return r2[i & 7];
        default: return r4[i & 15];
        }
    }
    void normal_gop(int pic_imcnt) // decide_normal_gop
    {
        const int i_period = P.keyint, gop = P.gop_size;
        auto anchor = [&](int type, int depth, int poc) { slice_type = type, slice_depth = depth, poc_val = poc, prev_doc_offset = 0, prev_poc_val = poc, slice_ref_flag = 1; };
        if(i_period == 0 && pic_imcnt == 0) anchor(ST_I, 0, pic_imcnt);
        else if(i_period != 0 && pic_imcnt % i_period == 0 && !P.closed_gop) anchor(ST_I, 0, pic_imcnt), ip_cnt++;
        else if(i_period != 0 && pic_cnt % i_period == 0 && P.closed_gop) anchor(ST_I, 0, pic_cnt), ip_cnt++;
        else if(pic_imcnt % gop == 0) anchor(P.inter_slice_type, 1, pic_imcnt);
        else {
            slice_type    = P.inter_slice_type;
            const int pos = (pic_imcnt % gop) - 1;
            slice_depth   = depth_b(gop, pos);
            poc_derivation(slice_depth - (slice_depth > 0));
            slice_ref_flag = gop >= 2 ? (slice_depth == depth_b(gop, gop - 2) ? 0 : 1) : 1;
        }
        poc_val += P.closed_gop ? (ip_cnt - 1) * i_period : 0;
        pick(poc_val);
    }
    PicPlan encode_one()
    {
        { // xeve_enc (xeve_enc.c:607-620)
            const int cnt = pic_icnt - frm_rnum, gop = P.gop_size;
            if(P.keyint == 0) force_slice = (pic_ticnt % gop >= pic_ticnt - cnt + 1) && force_output;
            else force_slice = ((pic_ticnt % P.keyint) % gop >= (pic_ticnt % P.keyint) - cnt + 1) && force_output;
        }
        // decide_slice_type
        const int i_period = P.keyint, gop = P.gop_size;
        const int ip_pic_cnt = P.closed_gop && i_period > 0 ? pic_cnt % i_period : pic_cnt;
        int icnt = ip_pic_cnt + P.bframes;
        pick(icnt);
        const bool aligned = !(P.closed_gop && i_period > 0 && ((ip_pic_cnt + gop - 1) / gop) > ((i_period - 1) / gop));
        if(gop == 1) {
            if(i_period == 1) slice_type = ST_I, slice_depth = 0, poc_val = icnt, slice_ref_flag = 0;
            else {
                const int imcnt = i_period > 0 ? icnt % i_period : icnt;
                if(imcnt == 0) slice_type = ST_I, slice_depth = 0, slice_ref_flag = 1;
                else slice_type = P.inter_slice_type, slice_depth = depth_p(P.ref_pic_gap_length, (imcnt - 1) % P.ref_pic_gap_length), slice_ref_flag = 1;
                poc_val = P.closed_gop && i_period > 0 && (pic_cnt % i_period) == 0 ? 0 : (P.closed_gop ? pic_cnt % i_period : pic_cnt);
            }
        }
        else if(icnt == gop - 1) { // the first picture of the sequence
            slice_type = ST_I, slice_depth = 0, poc_val = P.closed_gop ? ip_cnt * i_period : 0, prev_doc_offset = 0, prev_poc_val = poc_val, slice_ref_flag = 1;
            pick(poc_val);
            ip_cnt++, force_ignored_cnt = 0;
        }
        else if(force_slice) {
            int f = force_ignored_cnt;
            for(; f < gop; f++) {
                normal_gop(ip_pic_cnt + P.bframes + f);
                if(poc_val <= pic_ticnt && (P.keyint == 0 || poc_val < P.keyint * ip_cnt)) break;
            }
            force_ignored_cnt = f;
        }
        else if(!aligned) {
            int f = force_ignored_cnt;
            for(; f < gop; f++) {
                normal_gop(ip_pic_cnt + P.bframes + f);
                if(poc_val < P.keyint * ip_cnt && poc_val == slot_frame[pico]) break; // (the frame's time stamp is its number)
            }
            force_ignored_cnt = f;
        }
        else normal_gop(icnt);
        PicPlan pp;
        pp.tid = gop > 1 ? slice_depth - (slice_depth > 0) : 0;
        pp.frame = slot_frame[pico], pp.poc = poc_val, pp.slice_type = slice_type, pp.depth = slice_depth, pp.ref_flag = slice_ref_flag;
        pp.idr = pic_cnt == 0 || (slice_type == ST_I && P.closed_gop);
        pic_cnt++, slot_used[pico] = 0; // xeve_pic_finish (:1320-1322)
        return pp;
    }
};

// ---- the reference-picture buffer (xeve_picman.c): which earlier pictures a picture may use, in which order --------------------------------------------------------------
struct RefPic {
    int poc, tid, slot; // slot: the engine's picture store
    int list_poc0;      // pic->list_poc[0]: the POC of reference 0 of list 0 the picture itself was coded with (xeve_get_mv_dir reads it from the collocated picture)
};
class Dpb {
  public:
    explicit Dpb(int nslots) : used(nslots, 0) {}
    std::vector<RefPic> refs; // pm->pic[0 .. cur_num_ref_pics): pictures marked as reference, in the order they were put
    int num_refp[2] = {0, 0};
    RefPic refp[MAX_ACTIVE_REF][2] = {}; // ctx->refp (kept between pictures like the reference's: an I picture leaves it as it was)

    int get_empty() // xeve_picman_get_empty_pic: any store that holds no reference picture
    {
        for(size_t i = 0; i < used.size(); i++)
            if(!used[i]) return (int)i;
        return -1;
    }
    // xeve_picman_refp_init
    bool refp_init(int max_num_ref_pics, int slice_type, int poc, int layer_id, int last_intra)
    {
        if(slice_type == ST_I) return true;
        std::vector<RefPic> s = refs; // xeve_picman_update_pic_ref: descending POC
        std::stable_sort(s.begin(), s.end(), [](const RefPic &a, const RefPic &b) { return a.poc > b.poc; });
        const int n = (int)s.size();
        if(n == 0) return false;
        num_refp[0] = num_refp[1] = 0;
        auto before_intra = [&](const RefPic &r) { return poc >= last_intra && r.poc < last_intra; };
        int cnt = 0;
        if(slice_type == ST_P) {
            if(layer_id > 0) {
                for(int i = 0; i < n && cnt < max_num_ref_pics; i++) {
                    if(layer_id == 1) {
                        if(s[i].poc < poc && s[i].tid <= layer_id) refp[cnt++][0] = s[i];
                    }
                    else if(s[i].poc < poc && cnt == 0) refp[cnt++][0] = s[i];
                    else if(cnt != 0 && s[i].poc < poc && s[i].tid <= 1) refp[cnt++][0] = s[i];
                }
            }
            else
                for(int i = 0; i < n && cnt < max_num_ref_pics; i++) {
                    if(before_intra(s[i])) continue;
                    if(s[i].poc < poc) refp[cnt++][0] = s[i];
                }
        }
        else {
            int next = std::max(layer_id - 1, 0);
            for(int i = 0; i < n && cnt < max_num_ref_pics; i++) {
                if(before_intra(s[i])) continue;
                if(s[i].poc < poc && s[i].tid <= next) refp[cnt++][0] = s[i], next = std::max(s[i].tid - 1, 0);
            }
            if(cnt < max_num_ref_pics) {
                next = std::max(layer_id - 1, 0);
                for(int i = n - 1; i >= 0 && cnt < max_num_ref_pics; i--) {
                    if(before_intra(s[i])) continue;
                    if(s[i].poc > poc && s[i].tid <= next) refp[cnt++][0] = s[i], next = std::max(s[i].tid - 1, 0);
                }
            }
        }
        if(cnt == 0) return false;
        num_refp[0] = cnt;
        if(slice_type == ST_B) {
            int next = std::max(layer_id - 1, 0);
            cnt = 0;
            for(int i = n - 1; i >= 0 && cnt < max_num_ref_pics; i--) {
                if(before_intra(s[i])) continue;
                if(s[i].poc > poc && s[i].tid <= next) refp[cnt++][1] = s[i], next = std::max(s[i].tid - 1, 0);
            }
            if(cnt < max_num_ref_pics) {
                next = std::max(layer_id - 1, 0);
                for(int i = 0; i < n && cnt < max_num_ref_pics; i++) {
                    if(before_intra(s[i])) continue;
                    if(s[i].poc < poc && s[i].tid <= next) refp[cnt++][1] = s[i], next = std::max(s[i].tid - 1, 0);
                }
            }
            if(cnt == 0) return false;
            num_refp[1] = cnt;
            num_refp[0] = std::min(num_refp[0], max_num_ref_pics), num_refp[1] = std::min(num_refp[1], max_num_ref_pics);
        }
        return true;
    }
    // xeve_picman_put_pic (tool_rpl 0)
    void put(int slot, bool is_idr, int poc, int tid, bool ref_pic, int ref_pic_gap_length)
    {
        if(is_idr) drop_all();
        else if(tid == 0) { // pic_marking
            for(int i = 0; i < (int)refs.size(); i++)
                if(refs[i].tid > 0 || (i > 0 && ref_pic_gap_length > 0 && refs[i].poc % ref_pic_gap_length != 0)) drop(i), i--;
            while((int)refs.size() >= MAX_ACTIVE_REF) drop(0);
        }
        if(ref_pic) {
            RefPic r;
            r.poc = poc, r.tid = tid, r.slot = slot, r.list_poc0 = num_refp[0] > 0 ? refp[0][0].poc : 0; // picman_set_pic_to_pb (:188-213)
            refs.push_back(r), used[slot] = 1;
        }
    }

  private:
    std::vector<char> used;
    void drop(int i) { used[refs[i].slot] = 0, refs.erase(refs.begin() + i); }
    void drop_all() { while(!refs.empty()) drop(0); }
};

// ---- bit writer (xeve_bsw.c) and the high-level syntax ------------------------------------------------------------------------------------------------------------------
struct Bits {
    std::vector<uint8_t> b;
    uint32_t acc = 0;
    int      n   = 0;
    void put(uint32_t v, int len)
    {
        for(int i = len - 1; i >= 0; i--) {
            acc = (acc << 1) | ((v >> i) & 1);
            if(++n == 8) b.push_back((uint8_t)acc), acc = 0, n = 0;
        }
    }
    void ue(uint32_t v)
    {
        int len = 0;
        for(uint32_t t = (v + 1) >> 1; t; t >>= 1) len++;
        put(0, len), put(v + 1, len + 1);
    }
    void se(int v) { ue(v <= 0 ? (uint32_t)(-v * 2) : (uint32_t)(v * 2 - 1)); }
    void align() { while(n) put(0, 1); }
};
inline void nal_open(Bits &bs, int nut, int tid) // xeve_eco_nalu: the size is patched by nal_close
{
    bs.put(0, 32), bs.put(0, 1), bs.put((uint32_t)nut + 1, 6), bs.put((uint32_t)tid, 3), bs.put(0, 5), bs.put(0, 1);
}
inline void nal_close(std::vector<uint8_t> &nal) // xeve_eco_nal_unit_len
{
    const uint32_t size = (uint32_t)nal.size() - 4;
    for(int i = 0; i < 4; i++) nal[i] = (uint8_t)(size >> (24 - 8 * i));
}
inline std::vector<uint8_t> make_sps(const Param &P) // xeve_set_sps + xeve_eco_sps, Baseline: every tool flag 0
{
    Bits bs;
    nal_open(bs, NUT_SPS, 0);
    bs.ue(0), bs.put(0, 8) /* profile_idc: baseline */, bs.put((uint32_t)(P.level_idc * 3) & 0xFF, 8) /* level_idc */, bs.put(0, 32), bs.put(0, 32) /* toolset_idc_h / _l */;
    bs.ue(1) /* 4:2:0 */, bs.ue((uint32_t)P.w), bs.ue((uint32_t)P.h), bs.ue(BIT_DEPTH - 8), bs.ue(BIT_DEPTH - 8);
    bs.put(0, 13); // btt, suco, admvp, eipd, cm_init, iqt, addb, alf, htdf, rpl, pocs, dquant, dra
    const int log2_sub_gop = (int)(std::log2((double)P.gop_size) + .5);
    bs.ue((uint32_t)log2_sub_gop);
    if(log2_sub_gop == 0) bs.ue((uint32_t)(int)(std::log2((double)P.ref_pic_gap_length) + .5));
    bs.ue((uint32_t)P.max_num_ref_pics()), bs.put(0, 1) /* cropping */, bs.put(0, 1) /* chroma_qp_table_present_flag */, bs.put(0, 1) /* vui */;
    bs.align();
    nal_close(bs.b);
    return bs.b;
}
inline std::vector<uint8_t> make_pps(int tid) // xeve_set_pps + xeve_eco_pps
{
    Bits bs;
    nal_open(bs, NUT_PPS, tid);
    bs.ue(0), bs.ue(0), bs.ue(0), bs.ue(0), bs.ue(0);
    bs.put(0, 1) /* rpl1_idx_present */, bs.put(1, 1) /* single_tile_in_pic */, bs.ue(0) /* tile_id_len_minus1 */;
    bs.put(0, 5); // explicit_tile_id, pic_dra_enabled, arbitrary_slice_present, constrained_intra_pred, cu_qp_delta_enabled
    bs.align();
    nal_close(bs.b);
    return bs.b;
}
inline std::string fmt(const char *f, ...)
{
    char    t[256];
    va_list ap;
    va_start(ap, f);
    vsnprintf(t, sizeof(t), f, ap);
    va_end(ap);
    return t;
}
inline std::string sei_text(const Param &P) // xeve_eco_emitsei's banner + xeve_param2string: the encoder's settings as text, in the reference's spelling
{
    const int cs = (BIT_DEPTH << 8) | 11; // XEVE_CS_SET(XEVE_CF_YCBCR420, 10, 0): the application hands the codec 10-bit pictures
    std::string s = " xeve - MPEG-5 EVC codec - ESSENTIAL VIDEO CODING https://github.com/mpeg5/xeve - options: ";
    struct KV { const char *k; int v; };
    s += fmt("profile=%d threads=%d input-res=%dx%d fps=%.3f keyint=%d color-space=%d rc-type=CQP", 0, P.threads, P.w, P.h, (float)P.fps_num / P.fps_den, P.keyint, cs);
    const KV a[] = {{"qp", P.qp}, {"qp_cb_offset", P.qp_cb_offset}, {"qp_cr_offset", P.qp_cr_offset}, {"info", P.sei_info}, {"hash", 0}, {"bframes", P.bframes}, {"aq-mode", 0}, {"lookahead", P.lookahead},
                    {"closed-gop", P.closed_gop}, {"disable-hgop", 0}, {"ref_pic_gap_length", P.ref_pic_gap_length}, {"codec-bit-depth", BIT_DEPTH}, {"level-idc", P.level_idc},
                    {"cu-tree", 0}, {"constrained-ip", 0}, {"use-deblock", 1}, {"inter-slice-type", P.inter_slice_type}, {"rdo-deblk-switch", P.rdo_dbk},
                    {"qp-increased-frame", 0}, {"forced-idr-frame-flag", 0}, {"qp-increased-frame", 0}};
    for(const KV &e : a) s += fmt(" %s=%d", e.k, e.v);
    s += fmt(" qp-max=%d qp-min=%d gop-size=%d use-fcst=%d chroma-format-idc=%d cs-w-shift=%d cs-h-shift=%d", 51, 0, P.gop_size, 0, 1, 1, 1);
    s += fmt(" max-cu-intra=%d min-cu-intra=%d max-cu-inter=%d min-cu-inter=%d ", P.max_cu_intra, P.min_cu_intra, P.max_cu_inter, P.min_cu_inter);
    s += fmt(" max-num-ref=%d", P.ref);
    s += fmt(" me-ref-num=%d me-algo=%d me-range=%d me-sub=%d me-sub-pos=%d me-sub-range=%d ", P.me_ref_num, P.me_algo, P.me_range, P.me_sub, P.me_sub_pos, P.me_sub_range);
    const KV b[] = {{"rdoq", 1}, {"cabac-refine", 1}, {"intra-block-copy", 0}, {"btt", 0}, {"suco", 0}, {"amvr", 0}, {"vd", 0}, {"affine", 0}, {"dmvr", 0}, {"addb", 0},
                    {"alf", 0}, {"htdf", 0}, {"admvp", 0}, {"hmvp", 0}, {"eipd", 0}, {"iqt", 0}, {"cm-init", 0}, {"adcc", 0}, {"rpl", 0}, {"pocs", 0}, {"ats", 0}, {"pocs", 0},
                    {"deblock-alpha-offset", 0}, {"deblock-beta-offset", 0}, {"dra", 0}, {"aspect-ration-info-flag", 0}, {"overscan", 0}, {"videoformat", 2}, {"range", 0},
                    {"colorprim", 2}, {"transfer", 2}, {"colormatrix", 2}, {"master-display", 2}, {"chromaloc", 0}, {"field-seq-flag", 0}, {"vui-timing-info-flag", 0},
                    {"fixed-pic-rate-flag", 0}, {"nal-hrd-params-present-flag", 0}, {"vcl-hrd-params-present-flag", 0}, {"num-reorder-pics", 21}};
    for(const KV &e : b) s += fmt(" %s=%d", e.k, e.v);
    return s;
}
inline std::vector<uint8_t> make_sei(const Param &P, int tid) // xeve_encode_sei -> xeve_eco_emitsei -> write_sei_userdata_unregistered
{
    static const uint8_t uuid[16] = {0x2C, 0xA2, 0xDE, 0x09, 0xB5, 0x17, 0x47, 0xDB, 0xBB, 0x55, 0xA4, 0xFE, 0x7F, 0xC2, 0xFC, 0x4E};
    const std::string    t = sei_text(P);
    Bits bs;
    nal_open(bs, NUT_SEI, tid);
    bs.put(5, 8); // USER_DATA_UNREGISTERED
    uint32_t size = (uint32_t)(16 + t.size()) << 3; // (the reference writes the payload size in bits, xeve_eco.c:335)
    for(; size >= 0xff; size -= 0xff) bs.put(0xff, 8);
    bs.put(size, 8);
    for(uint8_t u : uuid) bs.put(u, 8);
    for(char c : t) bs.put((uint8_t)c, 8);
    bs.align();
    nal_close(bs.b);
    return bs.b;
}
// the slice NAL unit's head: NAL header + xeve_eco_sh (Baseline: no POC, no reference picture lists)
inline void slice_head(Bits &bs, bool idr, int tid, int slice_type, int qp, int qp_u_offset = 0, int qp_v_offset = 0)
{
    nal_open(bs, idr ? NUT_IDR : NUT_NONIDR, tid);
    bs.ue(0), bs.ue((uint32_t)slice_type);
    if(idr) bs.put(0, 1);                    // no_output_of_prior_pics_flag
    if(slice_type != ST_I) bs.put(0, 1);     // num_ref_idx_active_override_flag
    bs.put(1, 1), bs.put((uint32_t)qp, 6), bs.se(qp_u_offset), bs.se(qp_v_offset); // deblocking_filter_on, qp, qp_u_offset, qp_v_offset (xeve_eco.c:275-278)
    bs.align();
}

// ---- per-picture numbers ------------------------------------------------------------------------------------------------------------------------------------------------
inline int chroma_qp(int q) // ctx->qp_chroma_dynamic[c][q]: xeve_tbl_qp_chroma_ajudst (xeve_tbl.c:259-267) for q >= 0, the identity below (xeve_util.c:1841-1846)
{
    static const int t[58] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28,
                              29, 29, 29, 30, 31, 32, 32, 33, 33, 34, 34, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 39, 39, 40, 40, 40, 41, 41, 41};
    return q < 0 ? q : t[std::min(q, 57)];
}
inline int slice_qp(const Param &P, int depth) // xeve_set_sh: the hierarchy's QP offsets (xeve_qp_adapt_param_*, xeve_tbl.c:564-623)
{
    struct A { int layer; double offset, scale; };
    static const A ra8[8]  = {{0, 0, 0}, {1, 0, 0.4420}, {2, 0, 0.3536}, {3, 0, 0.3536}, {4, 0, 0.68}, {5, 0, 0.68}, {6, 0, 0.68}, {7, 0, 0.68}};
    static const A ra16[8] = {{-3, 0, 0}, {1, 0, 0}, {1, -4.8848, 0.2061}, {4, -5.7476, 0.2286}, {5, -5.9, 0.2333}, {6, -7.1444, 0.3}, {7, -7.1444, 0.3}, {8, -7.1444, 0.3}};
    static const A ld[8]   = {{-1, 0, 0}, {1, 0, 0}, {4, -6.5, 0.259}, {4, -6.5, 0.259}, {5, -6.5, 0.259}, {5, -6.5, 0.259}, {5, -6.5, 0.259}, {5, -6.5, 0.259}};
    static const A ai[8]   = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    const A *tab = P.bframes == 0 ? (P.keyint == 1 ? ai : ld) : (P.gop_size == 16 ? ra16 : ra8);
    double   qp  = std::min(51.0, std::max(0.0, (double)P.qp));
    qp += tab[depth].layer;
    const double dqp = qp * tab[depth].scale + tab[depth].offset + 0.5;
    qp += (int)std::floor(std::min(3.0, std::max(0.0, dqp)));
    return (int)(uint8_t)std::min(51.0, std::max(0.0, qp));
}
struct PicNumbers {
    int    qp, qp_y, qp_u, qp_v;
    double lambda[3], sqrt_lambda0, dcw[2];
    uint32_t lambda_mv;
};
inline PicNumbers pic_numbers(int qp, int qp_u_offset = 0, int qp_v_offset = 0) // set_lambda with the tile's QP (xeve_mode.c:660-679); mode_cu_init's QPs (:780-784)
{
    PicNumbers n;
    const int  off = 6 * (BIT_DEPTH - 8);
    const int  qu = chroma_qp(std::min(57, std::max(-off, qp + qp_u_offset))), qv = chroma_qp(std::min(57, std::max(-off, qp + qp_v_offset)));
    n.qp = qp, n.qp_y = qp + off, n.qp_u = qu + off, n.qp_v = qv + off;
    n.lambda[0] = 0.57 * std::pow(2.0, (qp - 12.0) / 3.0);
    n.dcw[0] = std::pow(2.0, (qp - qu) / 3.0), n.dcw[1] = std::pow(2.0, (qp - qv) / 3.0);
    n.lambda[1] = n.lambda[0] / n.dcw[0], n.lambda[2] = n.lambda[0] / n.dcw[1];
    n.sqrt_lambda0 = std::sqrt(n.lambda[0]);
    n.lambda_mv    = (uint32_t)std::floor(65536.0 * n.sqrt_lambda0);
    return n;
}
inline void fill_deblock_params(xeve_hip_deblock_params &d, const Param &P)
{
    memset(&d, 0, sizeof(d));
    d.w = P.w, d.h = P.h, d.w_scu = P.w >> 2, d.h_scu = P.h >> 2, d.log2_max_cuwh = LOG2_CTU, d.bit_depth_luma = d.bit_depth_chroma = BIT_DEPTH, d.chroma_format_idc = 1;
    d.qp_u_offset = P.qp_cb_offset, d.qp_v_offset = P.qp_cr_offset;
    const int off = 6 * (BIT_DEPTH - 8);
    for(int c = 0; c < 2; c++)
        for(int q = -off; q <= 57; q++) d.qp_chroma[c][q + off] = chroma_qp(q);
}
// what the CTU walk of a picture is given (the fields shim/xeve_hip_shim.c reads out of the live encoder's context, computed here)
inline void fill_tree_params(xeve_hip_tree_params &t, const Param &P, int slice_type, const PicNumbers &n)
{
    memset(&t, 0, sizeof(t));
    t.ip.w_scu = P.w >> 2, t.ip.h_scu = P.h >> 2, t.ip.slice_type = slice_type, t.ip.chroma_format_idc = 1, t.ip.bit_depth = BIT_DEPTH;
    t.ip.qp[0] = n.qp_y, t.ip.qp[1] = n.qp_u, t.ip.qp[2] = n.qp_v;
    for(int c = 0; c < 3; c++) t.ip.lambda[c] = n.lambda[c];
    t.ip.sqrt_lambda0 = n.sqrt_lambda0, t.ip.dist_chroma_weight[0] = n.dcw[0], t.ip.dist_chroma_weight[1] = n.dcw[1];
    t.pic_w = P.w, t.pic_h = P.h, t.log2_ctu = LOG2_CTU, t.min_cuwh = 4;
    t.max_cu = slice_type == ST_I ? P.max_cu_intra : P.max_cu_inter, t.min_cu = slice_type == ST_I ? P.min_cu_intra : P.min_cu_inter;
    t.slice_qp = n.qp, t.slice_num = 0, t.rdo_dbk = P.rdo_dbk;
}
inline void fill_inter_params(xeve_hip_tree_inter &I, const Param &P, int slice_type, int poc, const PicNumbers &n, const Dpb &dpb)
{
    static const int refi_bits[5][4] = {{0}, {0}, {1, 1}, {1, 2, 2}, {1, 2, 3, 3}}; // xeve_tbl_refi_bits (xeve_tbl.c:519-526)
    memset(&I, 0, sizeof(I));
    xeve_hip_inter_params &p = I.ipar;
    const int isb = slice_type == ST_B, nr[2] = {dpb.num_refp[0], isb ? dpb.num_refp[1] : 0};
    p.rdo.pic_w = P.w, p.rdo.pic_h = P.h, p.rdo.slice_type = slice_type, p.rdo.num_refp[0] = nr[0], p.rdo.num_refp[1] = nr[1], p.rdo.chroma_format_idc = 1;
    p.rdo.bit_depth = BIT_DEPTH, p.rdo.qp[0] = n.qp_y, p.rdo.qp[1] = n.qp_u, p.rdo.qp[2] = n.qp_v;
    for(int c = 0; c < 3; c++) p.rdo.lambda[c] = n.lambda[c];
    p.rdo.dist_chroma_weight[0] = n.dcw[0], p.rdo.dist_chroma_weight[1] = n.dcw[1];
    const int range = P.bframes == 0 ? 64 : P.me_range; // SEARCH_RANGE_IPEL_LD
    p.me.me.lambda_mv = n.lambda_mv, p.me.me.faststep = 3, p.me.me.max_search_range = range;
    p.me.me.min_clip[0] = p.me.me.min_clip[1] = -128 + 1, p.me.me.max_clip[0] = P.w - 1, p.me.me.max_clip[1] = P.h - 1; // xeve_pinter_create (xeve_pinter.c:2124-2127)
    p.me.hpel_cnt = P.me_sub > 1 ? P.me_sub_pos : 0, p.me.qpel_cnt = P.me_sub > 2 ? P.me_sub_pos : 0;
    p.me.me.reserved = P.me_algo > 1 ? 1 : 0;
    for(int l = 0; l < 2; l++)
        for(int r = 0; r < nr[l]; r++) {
            p.refi_bits[l][r]      = refi_bits[nr[l]][r];
            const int scaled       = (range * std::abs(poc - dpb.refp[r][l].poc) + (P.gop_size >> 1)) / P.gop_size; // get_range_ipel (xeve_pinter.c:122-129)
            p.range_recentre[l][r] = std::min(range, std::max(range >> 2, scaled));
        }
    p.max_cand = P.merge_num, p.poc = poc, p.col_list_poc0 = isb ? dpb.refp[0][1].list_poc0 : 0, p.skip_th = 0;
    I.ecu_depth = (poc % 2) ? 4 - 2 : 4; // ENC_ECU_DEPTH_B, ENC_ECU_ADAPTIVE (xeve_mode.c:2162-2166)
}

// the CTU order of a picture coded by T row chains (xeve_ctu_mt_core, xeve_enc.c:103-175): chain t owns CTU rows t, t + T, ...; a CTU waits for the one up-right of
// it (:130-133).  One entry per lockstep step: the CTUs that can be decided side by side.
struct ChainCtu { int t, x, y, lcu; };
inline std::vector<std::vector<ChainCtu>> wavefront(int w_lcu, int h_lcu, int threads)
{
    const int T = std::min(threads, h_lcu);
    std::vector<int> row(T), col(T, 0);
    std::vector<char> done((size_t)w_lcu * h_lcu, 0);
    for(int t = 0; t < T; t++) row[t] = t;
    std::vector<std::vector<ChainCtu>> steps;
    for(;;) {
        std::vector<ChainCtu> now;
        for(int t = 0; t < T; t++) {
            if(row[t] >= h_lcu) continue;
            const int x = col[t], y = row[t];
            if(y != 0 && x < w_lcu - 1 && !done[(size_t)(y - 1) * w_lcu + x + 1]) continue;
            now.push_back(ChainCtu{t, x, y, y * w_lcu + x});
        }
        if(now.empty()) break;
        for(const ChainCtu &c : now) {
            done[c.lcu] = 1;
            if(++col[c.t] == w_lcu) col[c.t] = 0, row[c.t] += T;
        }
        steps.push_back(now);
    }
    return steps;
}

} // namespace xenc
