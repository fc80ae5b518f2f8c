// xeve_amd/csrc/enc_host.h -- the closed-GOP batch encoder: G independent encoder runs (one closed GOP of F frames each, coded as the reference codes
// `--seek g*F --frames F`) advance in LOCKSTEP -- the same picture of every GOP at the same time, the same CTU of every picture in the same step -- so that one batched
// device call decides a CTU of every GOP.  This file is the frame loop above the engine: xeve_enc -> xeve_pic_prepare / xeve_header / xeve_pic / xeve_pic_finish
// (src_base/xeve_enc.c:602-640, 226-600, 1184-1337) with xeve_ctu_mt_core's row chains (:103-175) as the lockstep axis inside a picture.
//
// Engine = where the pictures live and who decides a CTU.  The product instantiates it with the HIP engine (encode.cpp: everything resident in HBM); the test
// harness under oracle/ instantiates the SAME template with a CPU engine built on the oracle, which pins this file and enc_plan.h against bitstreams of the
// unmodified reference without a GPU.  There is no run-time switch between the two.
//
//   struct Engine {
//       void begin_picture(const PicSetup &);     // every GOP: load the original of frame `frame`, clear the unit maps, take store `cur_slot` for the reconstruction
//       void reset_chain(int t);                  // every GOP: row chain t's writer = a freshly reset coder (fn_eco_sbac_reset, xeve_enc.c:114-118)
//       void step(const ChainCtu *c, int n);      // every GOP: decide CTU c[i] from chain c[i].t's writer state (:138-142), write it on that writer (xeve_eco_tree, :152), keep it
//       void end_picture(bool rewrite);           // every GOP: loop filter (:462); the slice data = all CTUs written again in raster order on a fresh coder + the tile's end
//                                                 // (:466-560; rewrite == false: chain 0's own bytes, which are the same when there is one chain); padding (xeve_pic_finish).
//                                                 // ISSUES the work: the next picture may begin while the second writer pass is still running
//       void collect(std::vector<std::vector<uint8_t>> &slice_data, std::vector<uint32_t> &bins);
//                                                 // the slice data + bin count of the picture last ended (waits for it); called before the next end_picture
//   };
#pragma once
#include "enc_plan.h"

namespace xenc __attribute__((visibility("hidden"))) { // (hidden: the test harness instantiates the same inline code in its own library, and the two must not bind to each other)

struct PicSetup {
    int frame, poc, slice_type, cur_slot, nchains;
    int nref[2];
    RefPic ref[MAX_ACTIVE_REF][2]; // [refi][list]
    xeve_hip_tree_params    tp;
    xeve_hip_tree_inter     ti;    // (pointers are the engine's to fill)
    xeve_hip_eco_params     ep;
    xeve_hip_deblock_params dp;
};

template <class Engine> class BatchEncoder {
  public:
    BatchEncoder(Engine &e, const Param &p, int ngops, int nframes) : E(e), P(p), G(ngops), F(nframes), dpb(1) {}
    std::vector<PicPlan> plan() const { return Planner(P, F).run(); }
    // picture stores the run needs at once: the reference pictures alive at some point plus the picture being coded (a dry run of the bookkeeping)
    static int slots_needed(const Param &P, int nframes)
    {
        Dpb dpb(64);
        int need = 1, last_intra = 0;
        for(const PicPlan &pp : Planner(P, nframes).run()) {
            if(pp.slice_type == ST_I) last_intra = pp.poc;
            (void)dpb.refp_init(P.max_num_ref_pics(), pp.slice_type, pp.poc, pp.tid, last_intra);
            const int slot = dpb.get_empty();
            if(slot < 0) return -1;
            need = std::max(need, slot + 1);
            dpb.put(slot, pp.idr != 0, pp.poc, pp.tid, pp.ref_flag != 0, P.ref_pic_gap_length);
        }
        return need;
    }

    // The run as a resumable sequence of LOCKSTEP STEPS (one CTU of every row chain of every GOP each): begin(), then advance() until it returns 0.  A picture's
    // set-up rides on its first step, its end (loop filter, slice data, NAL units, reference bookkeeping) on its last.  out[g] = the bitstream of GOP g (what the
    // reference application writes to its output file for that run), complete when advance() has returned 0.
    int begin(std::vector<std::vector<uint8_t>> &out_)
    {
        out = &out_;
        out->assign(G, std::vector<uint8_t>());
        // low-delay closed GOPs with an odd keyint over more than one GOP: the reference keeps two input slots there (xeve_enc.c:1693-1695), files a frame under its count
        // in the SEQUENCE (:661) and fetches it under its count in the GOP (:1080) -- with an odd keyint the two part ways after the first GOP and it codes stale slots
        if(P.bframes == 0 && P.closed_gop && P.keyint > 1 && (P.keyint & 1) && F > P.keyint)
            return fail("low-delay closed GOPs need an even keyint when a run holds more than one GOP: the reference codes stale input slots otherwise");
        pics = plan();
        if((int)pics.size() != F) // (e.g. low-delay closed GOPs with an odd keyint and more frames than one GOP: the reference picks its input slot from the picture's
                                  // count inside the GOP while the frames sit at their count in the sequence, xeve_enc.c:1080 vs :661 -- its own output there codes
                                  // stale slots and its application does not terminate cleanly; nothing to reproduce)
            return fail("the frame loop did not code every frame (the reference has no defined output for this combination of options and frame count)");
        dpb = Dpb(slots_needed(P, F));
        const int w_lcu = (P.w + CTU - 1) / CTU, h_lcu = (P.h + CTU - 1) / CTU;
        steps = wavefront(w_lcu, h_lcu, P.threads);
        T = std::min(P.threads, h_lcu), last_intra_poc = 0, pic = 0, step = 0, pending = -1;
        return 0;
    }
    long total_steps() const { return (long)pics.size() * (long)steps.size(); }
    long remaining() const { return total_steps() - ((long)pic * (long)steps.size() + step); }
    // up to max_steps further steps; returns the steps still to do (0: the run is complete), -1 on error
    long advance(long max_steps)
    {
        for(long done = 0; done < max_steps && pic < (int)pics.size(); done++) {
            if(step == 0 && begin_picture() != 0) return -1;
            E.step(steps[step].data(), (int)steps[step].size());
            if(++step == (int)steps.size()) {
                if(pending >= 0 && finish_picture() != 0) return -1; // (the picture before this one: its second writer pass ran beside this picture's steps)
                E.end_picture(T > 1 || always_rewrite);
                const PicPlan &pp = pics[pic];
                dpb.put(S.cur_slot, pp.idr != 0, pp.poc, pp.tid, pp.ref_flag != 0, P.ref_pic_gap_length); // xeve_pic_finish -> xeve_picman_put_pic
                pending = pic, pending_qp = qp;
                step = 0, pic++;
                if(pic == (int)pics.size() && finish_picture() != 0) return -1;
            }
        }
        return remaining();
    }
    // the access unit of the picture whose end was issued last, NOW instead of at the next picture's end (waits for its second writer pass): afterwards out[g] holds every
    // picture whose steps have all been issued -- what a caller that stops a run part way (a bounded measurement, a test against the reference's per-picture prefixes)
    // reads.  The run can go on after it.
    int flush() { return pending >= 0 ? finish_picture() : 0; }
    int run(std::vector<std::vector<uint8_t>> &out_)
    {
        if(begin(out_) != 0) return -1;
        return advance(total_steps() + 1) == 0 ? 0 : -1;
    }
    // every picture's set-up, in coding order, from a dry run of the bookkeeping: what an engine sizes its buffers by before the first step (empty: the run is refused)
    std::vector<PicSetup> dry_setups() const
    {
        std::vector<PicSetup> v;
        Dpb d(slots_needed(P, F));
        int last_intra = 0, q = 0;
        for(const PicPlan &pp : plan()) {
            PicSetup s;
            if(make_setup(pp, d, last_intra, s, q)) return {};
            v.push_back(s);
            d.put(s.cur_slot, pp.idr != 0, pp.poc, pp.tid, pp.ref_flag != 0, P.ref_pic_gap_length);
        }
        return v;
    }
    std::string error;
    bool always_rewrite = false; // (tests: one chain through the second pass too)

  private:
    Engine     &E;
    const Param P;
    const int   G, F;
    std::vector<std::vector<uint8_t>> *out = nullptr;
    std::vector<PicPlan> pics;
    std::vector<std::vector<ChainCtu>> steps;
    Dpb dpb;
    int T = 1, last_intra_poc = 0, pic = 0, step = 0, qp = 0, pending = -1, pending_qp = 0;
    PicSetup S;
    std::vector<std::vector<uint8_t>> slice;
    std::vector<uint32_t> bins;
    int fail(const char *m) { error = m; return -1; }

    // the set-up of picture pp against the bookkeeping d (xeve_pic_prepare, xeve_set_sh, xeve_picman_refp_init); nullptr: fine, else what is wrong
    const char *make_setup(const PicPlan &pp, Dpb &d, int &last_intra, PicSetup &S_, int &qp_) const
    {
        if(pp.frame < 0 || pp.frame >= F) return "the frame loop asked for a frame that was never pushed";
        if(pp.slice_type == ST_I) last_intra = pp.poc; // xeve_pic_prepare (:1217-1218)
        qp_ = slice_qp(P, pp.depth);
        const PicNumbers num = pic_numbers(qp_, P.qp_cb_offset, P.qp_cr_offset);
        if(!d.refp_init(P.max_num_ref_pics(), pp.slice_type, pp.poc, pp.tid, last_intra)) return "no reference picture for an inter picture";
        memset(&S_, 0, sizeof(S_));
        S_.frame = pp.frame, S_.poc = pp.poc, S_.slice_type = pp.slice_type, S_.nchains = std::min(P.threads, (P.h + CTU - 1) / CTU);
        if((S_.cur_slot = d.get_empty()) < 0) return "no free picture store";
        fill_tree_params(S_.tp, P, pp.slice_type, num);
        if(pp.slice_type != ST_I) {
            fill_inter_params(S_.ti, P, pp.slice_type, pp.poc, num, d);
            S_.nref[0] = d.num_refp[0], S_.nref[1] = pp.slice_type == ST_B ? d.num_refp[1] : 0;
            for(int l = 0; l < 2; l++)
                for(int r = 0; r < S_.nref[l]; r++) S_.ref[r][l] = d.refp[r][l];
            if(pp.slice_type == ST_B && S_.nref[1] > S_.nref[0]) return "list 1 longer than list 0: outside what the inter analysis takes";
        }
        S_.ep.chroma_format_idc = 1, S_.ep.slice_type = pp.slice_type, S_.ep.log2_ctu = LOG2_CTU, S_.ep.pic_w = P.w, S_.ep.pic_h = P.h, S_.ep.w_scu = P.w >> 2, S_.ep.h_scu = P.h >> 2;
        S_.ep.num_refp[0] = d.num_refp[0], S_.ep.num_refp[1] = d.num_refp[1];
        fill_deblock_params(S_.dp, P);
        return nullptr;
    }
    int begin_picture()
    {
        const char *bad = make_setup(pics[pic], dpb, last_intra_poc, S, qp);
        if(bad) return fail(bad);
        E.begin_picture(S);
        for(int t = 0; t < T; t++) E.reset_chain(t);
        return 0;
    }
    int finish_picture() // the access unit of the picture whose end was issued last
    {
        const PicPlan &pp = pics[pending];
        const int qp = pending_qp;
        pending = -1;
        E.collect(slice, bins);
        if((int)slice.size() != G || (int)bins.size() != G) return fail("the engine returned no slice data");
        // the access unit: parameter sets in front of an IDR picture (xeve_header), then the slice NAL unit (xeve_pic :466-590)
        for(int g = 0; g < G; g++) {
            std::vector<uint8_t> &o = (*out)[g];
            if(pp.idr) {
                const std::vector<uint8_t> sps = make_sps(P), pps = make_pps(pp.tid), sei = make_sei(P, pp.tid);
                o.insert(o.end(), sps.begin(), sps.end()), o.insert(o.end(), pps.begin(), pps.end());
                if(P.sei_info) o.insert(o.end(), sei.begin(), sei.end()); // (--info 0: no SEI with the options, xeve_enc.c:1989)
            }
            Bits bs;
            slice_head(bs, pp.idr != 0, pp.tid, pp.slice_type, qp, P.qp_cb_offset, P.qp_cr_offset);
            std::vector<uint8_t> nal = bs.b;
            nal.insert(nal.end(), slice[g].begin(), slice[g].end());
            // cabac_zero_words when the slice's bins outrun its bytes (:562-583)
            const uint32_t num_bytes = ((uint32_t)nal.size() & ~3u) - 4; // (bs->cur counts whole flushed words, xeve_bsw.c:32-44)
            const int      pw = ((P.w + 3) / 4) * 4, ph = ((P.h + 3) / 4) * 4, raw_bits = pw * ph * (BIT_DEPTH + 2 * (BIT_DEPTH >> 2));
            const uint32_t threshold = (32 / 3) * num_bytes + (uint32_t)(raw_bits / 32);
            if(bins[g] >= threshold) {
                const uint32_t target = ((bins[g] - (uint32_t)(raw_bits / 32)) * 3 + 31) / 32;
                if(target > num_bytes)
                    for(uint32_t i = 0, words = (target - num_bytes + 2) / 3; i < words; i++) nal.push_back(0), nal.push_back(0);
            }
            nal_close(nal);
            o.insert(o.end(), nal.begin(), nal.end());
        }
        return 0;
    }
};

} // namespace xenc
