// xeve_amd/csrc/sbac.hip -- CABAC (SBAC) bit counting of an inter CU: the rate term of the inter RDO.
//
// reference: src_base/xeve_mode.c:39-295 (xeve_sbac_bit_reset, xeve_get_bit_number, xeve_rdo_bit_cnt_cu_inter,
// _cu_inter_comp, _cu_skip) over src_base/xeve_eco.c (xeve_sbac_encode_bin :521-575, sbac_encode_bin_ep :455-472,
// sbac_carry_propagate :429-453, sbac_put_byte :397-427, xeve_eco_run_length_cc :707-771, xeve_eco_cbf :793-894, ...).
//
// An adaptive binary arithmetic coder is a serial chain per CU (range and context states feed forward bin by bin), so
// the parallel axis is the JOB: one lane per job, and the lever is instructions per bin.  Which model a bin uses and what value it has
// depend on the coefficient block alone, never on the coder state, so:
//   * k_coef_events (wave-cooperative): every coded coefficient block is compacted, in zig-zag order, into a list of (zero-run, |level| - 1,
//     sign, is-last-position) events by ballot + popcount AND expanded into its BIN STRING -- one byte per bin: (model << 1) | value;
//   * k_cu_bits (one lane per job): the few header bins (skip / pred_mode / direct / inter_dir / refi / mvp_idx / mvd / cbf) are queued per lane
//     in LDS and drained by one loop; the coefficient bins are streamed from the blocks' strings (fetch byte, model, coder step, model write-back);
//     blocks without a usable string are coded from their event lists (code_events);
//   * k_cu_bits_chain: the per-component cbf tests of pinter_residue_rdo under an assumed outcome, one lane per assumption.
// The coder state is carried field for field (code register, pending / stacked bytes, bit counter), so the exit state
// is what SBAC_STORE would keep and xeve_get_bit_number's formula applies unchanged.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "xh_common.h"

#define NCTX XEVE_HIP_SBAC_NCTX
// Header queue in LDS: QCAP entries per lane.  A header has up to 5 + 2 * (20 refi + 3 mvp + 2 * 34 mvd) + 4 cbf = 191 bins, but more than a few dozen only with
// vector differences of thousands of samples: the queue holds a WINDOW of the header, and a job whose header is longer generates it again for the next window
// (q_header is a pure function of the job).  48 entries = 3 KB per wave instead of 12.5: LDS per workgroup 22 -> 12.5 KB, 7 -> 12 waves per CU.
#define QCAP 48

struct CuBitsK {
    int n[3], log2n[3];
    int slice_type, num_refp[2], cm_init, idc;
    const uint16_t *scan[3];
};

// ---- event list ---------------------------------------------------------------------------------------------------
// bits 0..14 |level| - 1, 15 sign, 16..27 zero run before it, 28 the coefficient sits at the last scan position
__device__ __forceinline__ unsigned ev_pack(int v, int run, int at_end)
{
    const unsigned a = (unsigned)(v < 0 ? -v : v) & 0xFFFFu; // XEVE_ABS16
    return ((a - 1) & 0x7FFFu) | ((unsigned)(v < 0) << 15) | ((unsigned)run << 16) | ((unsigned)at_end << 28);
}

// which components of a job are coded (xeve_eco_coefficient: nnz_sub[c] && run[c]; cbf_all == 0 codes nothing)
__device__ __forceinline__ unsigned coded_mask(const xeve_hip_cu_bits_job &j)
{
    if(j.mode == XEVE_HIP_BITS_CU_SKIP || j.mode == XEVE_HIP_BITS_MVP || j.mode == XEVE_HIP_BITS_INTRA_DIR) return 0;
    unsigned m = (j.nnz[0] ? 1u : 0u) | (j.nnz[1] ? 2u : 0u) | (j.nnz[2] ? 4u : 0u);
    if(j.mode == XEVE_HIP_BITS_ECO_COEF) return m & ((j.dir_flag >> 2) & 7u);
    if(j.mode == XEVE_HIP_BITS_INTRA_LUMA) return m & 1u;
    if(j.mode != XEVE_HIP_BITS_CU_INTER && j.mode != XEVE_HIP_BITS_CU_INTRA) m &= 1u << (j.mode - 1);
    return m;
}

// ---- bin streams ---------------------------------------------------------------------------------------------------------
// The serial part of the coder only ever needs, per bin, WHICH model and WHAT value: both are functions of the coefficient block alone (the
// models of run / level / last depend on the component type -- and with sps_cm_init_flag 1 on the previous level -- never on the coder state).
// So the event pass also expands every coded block, in parallel, into its bin string: one byte per bin, (model index << 1) | value, model index
// BYP for the bypass-coded sign.  The block's string lives in the workspace at byte offset coef_off * BINK behind a 16-byte header {bins, events};
// a block whose string does not fit its BINK bytes per coefficient is marked OVF and its jobs take the event automaton below (exact, slower).
// The serial kernel (k_cu_bits_s) is then a loop of "fetch byte, model, encode, write model back": a third of the instructions of the automaton.
#define BINK 32          // bytes of bin-stream space per coefficient (header included)
#define BIN_HDR 16
#define BIN_OVF 0xFFFFFFFFu
#define BYP NCTX         // model index of bypass bins (a dummy row of the model table)

// LPB lanes per (job, component) block
template <int LPB> __global__ __launch_bounds__(256) void k_coef_events(const int16_t *__restrict__ coef, const xeve_hip_cu_bits_job *__restrict__ jobs,
                                                                        int njobs, CuBitsK P, unsigned *__restrict__ ev, int *__restrict__ nev,
                                                                        unsigned char *__restrict__ bins)
{
    constexpr int GPW = 64 / LPB;
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int g = lane / LPB, l = lane % LPB;
    const int item = wave * GPW + g, job = item / 3, c = item % 3;
    bool on = job < njobs;
    int off = 0;
    if(on) {
        const xeve_hip_cu_bits_job &J = jobs[job];
        on = (coded_mask(J) >> c) & 1;
        off = J.coef_off[c];
    }
    const int n = P.n[c];
    const uint16_t *scan = P.scan[c];
    int count = 0, prev = -1;
    // bin stream of this block
    unsigned char *bo = bins ? bins + (size_t)off * BINK + BIN_HDR : nullptr;
    const unsigned cap = (unsigned)n * BINK - BIN_HDR;
    unsigned base = 0;        // bins of the events before this chunk
    int      lfp = -1;        // where the most recent event's last-position flag sits (-1: it has none)
    unsigned plev = 5;        // min(previous level - 1, 5); 5 at the start of a block (xeve_eco.c:722,731-733)
    const int ch = c != 0;
    // the trip count is uniform per wave only when all its groups work on equally sized blocks: use the largest
    const int nmax = P.n[0];
    for(int chunk = 0; chunk < nmax; chunk += LPB) {
        const int pos = chunk + l;
        int v = 0;
        if(on && pos < n) v = coef[off + scan[pos]];
        const unsigned long long ball = __ballot(v != 0);
        unsigned long long mask = ball;
        if constexpr(LPB < 64) mask = (ball >> (g * LPB)) & ((1ull << LPB) - 1);
        const unsigned long long below = mask & ((1ull << l) - 1);
        const unsigned a = (unsigned)(v < 0 ? -v : v) & 0xFFFFu, lev1 = (a - 1) & 0x7FFFu;
        int run = 0;
        if(v != 0) {
            const int before = below ? chunk + 63 - __clzll((long long)below) : prev;
            run = pos - before - 1;
            ev[off + count + __popcll(below)] = ev_pack(v, run, pos == n - 1);
        }
        if(bins) { // (uniform)
            const bool at_end = pos == n - 1;
            unsigned cntb = v != 0 ? (run ? run + 1 : 1) + (lev1 ? lev1 + 1 : 1) + 1 + (at_end ? 0 : 1) : 0;
            unsigned incl = cntb;
#pragma unroll
            for(int d = 1; d < LPB; d <<= 1) {
                const unsigned t = __shfl_up(incl, d, LPB);
                if(l >= d) incl += t;
            }
            const unsigned tot = __shfl(incl, LPB - 1, LPB);
            // the level before this event (sps_cm_init_flag 1 picks the models by it): the previous non-zero lane's, or the last chunk's
            const int pl = below ? 63 - __clzll((long long)below) : 0;
            const unsigned lev_prev = __shfl(lev1, pl, LPB);
            if(v != 0) {
                const unsigned pv = below ? (lev_prev < 5 ? lev_prev : 5u) : plev;
                const unsigned t0 = P.cm_init == 1 ? (pv << 1) + ch * 12 : ch * 2;
                const unsigned r0 = (XEVE_HIP_CTX_RUN + t0) << 1, l0 = (XEVE_HIP_CTX_LEVEL + t0) << 1;
                unsigned q = base + incl - cntb;
                if(q + cntb <= cap) { // (a string that overflows is not used at all: no point in writing part of it)
                    bo[q++] = (unsigned char)(r0 | (run != 0));
                    if(run) {
                        for(int i = 1; i < run; i++) bo[q++] = (unsigned char)((r0 + 2) | 1);
                        bo[q++] = (unsigned char)(r0 + 2);
                    }
                    bo[q++] = (unsigned char)(l0 | (lev1 != 0));
                    if(lev1) {
                        for(unsigned i = 1; i < lev1; i++) bo[q++] = (unsigned char)((l0 + 2) | 1);
                        bo[q++] = (unsigned char)(l0 + 2);
                    }
                    bo[q++] = (unsigned char)((BYP << 1) | (unsigned)(v < 0));
                    if(!at_end) bo[q] = (unsigned char)((XEVE_HIP_CTX_LAST + ch) << 1); // "not the last coefficient" until the patch below says otherwise
                }
            }
            if(mask) { // the chunk's last event: remember where its last-position flag sits and its level
                const int hi = 63 - __clzll((long long)mask);
                const unsigned ie = __shfl(incl, hi, LPB), le = __shfl(lev1, hi, LPB);
                lfp = (chunk + hi == n - 1) ? -1 : (int)(base + ie - 1);
                plev = le < 5 ? le : 5u;
            }
            base += tot;
        }
        if(mask) prev = chunk + 63 - __clzll((long long)mask);
        count += __popcll(mask);
    }
    if(job < njobs && l == 0) {
        if(nev) nev[job * 3 + c] = on ? count : 0;
        if(bins && on) {
            const bool fits = base <= cap;
            if(fits && lfp >= 0) bo[lfp] |= 1; // the final event's flag: this was the last coefficient
            unsigned *hdr = reinterpret_cast<unsigned *>(bo - BIN_HDR);
            hdr[0] = fits ? base : BIN_OVF, hdr[1] = (unsigned)count;
        }
    }
}

// ---- the coder (count mode), state in registers --------------------------------------------------------------------
// FULL carries every field of XEVE_SBAC (code register, pending / stacked bytes, bit counter) so that the exit state is
// what SBAC_STORE would keep.  Without it only what feeds forward is kept: range, the models, and the number of
// renormalisation shifts -- which IS xeve_get_bit_number after xeve_sbac_bit_reset: every shift moves one bit out of the
// 11 + 8k bit window the formula measures (bitcounter + 8 * (stacked + pending) + 8 - code_bits + 3 == total shifts;
// tests/test_sbac_golden.py::test_bit_count_is_the_number_of_renormalisation_shifts).
// THE MODEL TRANSITIONS AS A TABLE.  xeve_sbac_encode_bin's update of a context model (xeve_eco.c:540-557) is a function of the model and of "was it the LPS": both
// outcomes of all 1024 models sit in one 32-bit word each (low half: after an MPS, high half: after an LPS; k_model_tab below, built at compile time from the
// reference's arithmetic), copied into LDS when a kernel starts.  The word is fetched as soon as the model is known, the range arithmetic of the bin runs while it is
// on its way, and one select replaces the twelve instructions that computed both successors -- on a coder whose speed IS its instruction count per bin (one wave
// per SIMD, every lane at a different bin), a fifth of them.
struct ModelTab {
    uint32_t v[1024];
};
static constexpr ModelTab make_model_tab()
{
    ModelTab t{};
    for(unsigned m = 0; m < 1024; m++) {
        const unsigned state = m >> 1, mps = m & 1;
        unsigned       sl = state + ((528 - state) >> 5); // LPS: towards 1/2, swapping the MPS past it
        const bool     flip = sl > 256;
        sl = flip ? 512 - sl : sl;
        const unsigned sm = state - ((state + 16) >> 5);  // MPS
        const unsigned after_mps = (sm << 1) | mps, after_lps = (sl << 1) | (flip ? mps ^ 1 : mps);
        t.v[m] = (after_mps & 0xFFFFu) | ((after_lps & 0xFFFFu) << 16);
    }
    return t;
}
__device__ const ModelTab k_model_tab = make_model_tab();
// into LDS by the whole wave (call before any lane leaves); one wave per workgroup
__device__ __forceinline__ void load_model_tab(uint32_t *s_tab)
{
    const uint4 *g = reinterpret_cast<const uint4 *>(k_model_tab.v);
    uint4       *d = reinterpret_cast<uint4 *>(s_tab);
    for(int i = threadIdx.x; i < 256; i += 64) d[i] = g[i];
    __syncthreads();
}

struct Sbac {
    unsigned range, shifts, bins;
    const uint32_t *tab; // k_model_tab in LDS
    // FULL only.  code / cb are the reference's code register and code_bits, bin for bin (the byte that leaves at a boundary is cut out branch-free).  What is
    // DEFERRED is the bookkeeping of the bytes that left (sbac_carry_propagate / sbac_put_byte: pending byte, stacked 0xFF / 0x00, bit counter): they queue up in
    // `fifo` (16 bits each, oldest on top) and sb_drain() works them off in order every four bins.  64 lanes cross byte boundaries at 64 different bins, so a
    // per-bin "a byte leaves" branch is taken by some lane on nearly every bin and the whole wave pays the branchy bookkeeping each time; a drain every four bins
    // (at most one byte leaves per bin: n <= 7) costs a fraction of it.  (Keeping the register 64 bits wide and cutting the bytes out late is cheaper still but
    // NOT field-exact: a carry that arrives after its byte has left stays in `code` until the next boundary in the reference -- measured and rejected.)
    unsigned long long fifo;
    unsigned code, cb, cnt, sff, sz, pb, ipb, bc;
};

// sbac_put_byte with is_bitcount set: written bytes only advance the bit counter
__device__ __forceinline__ void sb_byte(Sbac &s, unsigned b)
{
    if(s.ipb) {
        if(s.pb == 0) s.sz++;
        else s.bc += 8 * s.sz + 8, s.sz = 0;
    }
    s.pb = b, s.ipb = 1;
}

// sbac_carry_propagate with its while loops in closed form; out = the reference's code >> 17 at the byte boundary
__device__ __forceinline__ void sb_carry(Sbac &s, unsigned out)
{
    if(out == 0xFF) {
        s.sff++;
        return;
    }
    if(out > 0xFF) {
        s.pb++;
        if(s.sff) { // the first 0x00 flushes the pending byte, the others stack up as zeros
            sb_byte(s, 0);
            s.sz += s.sff - 1, s.sff = 0;
        }
    }
    else if(s.sff) { // the first 0xFF flushes the pending byte, each further one writes an 0xFF
        sb_byte(s, 0xFF);
        if(s.sff > 1) s.bc += 8 * s.sz + 8 * (s.sff - 1), s.sz = 0;
        s.sff = 0;
    }
    sb_byte(s, out & 0xFF);
}
// the queued bytes, oldest first
__device__ __forceinline__ void sb_drain(Sbac &s)
{
    while(s.cnt) {
        s.cnt--;
        sb_carry(s, (unsigned)(s.fifo >> (16 * s.cnt)) & 0xFFFFu);
    }
}

// xeve_sbac_encode_bin on model m (returns the updated model) / sbac_encode_bin_ep (ep; m passes through).  Branch-free
// in the part every bin executes: 64 lanes are at 64 different places of 64 different bin strings.
template <bool FULL> __device__ __forceinline__ unsigned sb_encode(Sbac &s, unsigned m, unsigned bin, bool ep)
{
    const unsigned R = s.range & 0xFFFFu, state = (m >> 1) & 511u, mps = m & 1; // (the masks cost nothing where the values are: they let the 24-bit multiply be chosen)
    const unsigned both = s.tab[m & 1023u];           // the model after an MPS (low half) / after an LPS (high half): on its way while the range is worked out
    __builtin_amdgcn_sched_barrier(0);                // (issued HERE, in front of the range arithmetic ...)
    unsigned lps = (state * R) >> 9;                  // (state < 2^9, range < 2^16: the full-rate 24-bit multiply)
    lps = lps < 437 ? 437 : lps;
    const unsigned rm = R - lps;                      // range after taking the MPS branch
    const bool     isl = bin != mps, cut = isl && rm >= lps;
    const unsigned r = cut ? lps : rm;
    const int      lz = __builtin_clz(r) - 18;        // 437 <= r < 2^16: at most 5 shifts back to >= 8192
    unsigned n = lz > 0 ? (unsigned)lz : 0;
    unsigned rn = r << n;
    // (the bin's arithmetic stays unconditional: the lanes of a wave are at different bins of different strings, so a branch around it for the bypass bins is taken
    // both ways by every wave and only adds its own cost)
    asm volatile("" : "+v"(rn), "+v"(n));
    __builtin_amdgcn_sched_barrier(0);                // (... and waited for here, behind it)
    const unsigned m1 = isl ? both >> 16 : both & 0xFFFFu;
    const unsigned half = R >> 1;                     // bypass: the range loses its LSB (xeve_eco.c:459-467), one shift
    s.range = ep ? R & ~1u : rn;
    n = ep ? 1 : n;
    s.shifts += n, s.bins++;
    if(FULL) { // the code register as the reference moves it; a byte that leaves (n >= cb; at most one: n <= 7) is queued for sb_drain
        const unsigned c0 = s.code + (ep ? (bin ? half : 0) : (cut ? rm : 0));
        const bool     crossed = n >= s.cb;
        const unsigned n1 = crossed ? s.cb : n, c1 = c0 << n1; // up to the boundary (or all the shifts)
        s.fifo = crossed ? (s.fifo << 16) | (c1 >> 17) : s.fifo; // (the byte and what the reference finds above it: 9 bits, 13 right after a bit reset)
        s.cnt += crossed ? 1u : 0u;
        s.code = (crossed ? c1 & 0x1FFFFu : c1) << (n - n1);
        s.cb   = crossed ? 8 - (n - n1) : s.cb - n;
    }
    return ep ? m : m1;
}

// ---- header queue ----------------------------------------------------------------------------------------------------
// entry: the byte of the bin strings -- (model << 1) | bin, bypass bins as model BYP (the dummy row of the model table)
struct Queue {
    uint8_t *q; // &s_q[0][lane], stride 64
    int      n; // bins generated so far
    int      base; // first bin of the window the LDS queue holds
    __device__ __forceinline__ void put(unsigned e)
    {
        if((unsigned)(n - base) < (unsigned)QCAP) q[64 * (n - base)] = (uint8_t)e;
        n++;
    }
    __device__ __forceinline__ void ctx(int ci, unsigned bin) { put(((unsigned)ci << 1) | (bin & 1)); }
    __device__ __forceinline__ void ep(unsigned bin) { put((BYP << 1) | (bin & 1)); }
};
__device__ __forceinline__ void q_intra_dir(Queue &Q, unsigned sym)
{ // xeve_eco_intra_dir (xeve_eco.c:1104-1121): sbac_write_unary_sym(mpm[ipm], 2 models) (:474-490)
    Q.ctx(XEVE_HIP_CTX_INTRA_DIR, sym != 0);
    while(sym) {
        sym--;
        Q.ctx(XEVE_HIP_CTX_INTRA_DIR + 1, sym != 0);
    }
}

__device__ __forceinline__ void q_mvd1(Queue &Q, int v)
{ // xeve_eco_abs_mvd + sign (xeve_eco.c:1205-1270)
    const unsigned a = (unsigned)(v < 0 ? -v : v);
    unsigned nn = (a + 1) >> 1;
    int len = 0;
    for(; len < 16 && nn; len++) nn >>= 1;
    const unsigned code = (1u << len) | ((a + 1 - (1u << len)) & ((1u << len) - 1));
    const int nbin = 2 * len + 1;
    for(int i = 0; i < nbin; i++) {
        const unsigned b = (code >> (nbin - 1 - i)) & 1;
        if(i <= 1) Q.ctx(XEVE_HIP_CTX_MVD, b);
        else Q.ep(b);
    }
    if(a) Q.ep(v < 0);
}
__device__ __forceinline__ void q_mvp_idx(Queue &Q, int idx)
{ // sbac_write_truncate_unary_sym(idx, 3, 4) (xeve_eco.c:492-511)
    for(int i = 0; i < 3; i++) {
        Q.ctx(XEVE_HIP_CTX_MVP_IDX + i, i != idx);
        if(i == idx) break;
    }
}
__device__ __forceinline__ void q_refi(Queue &Q, int num_refp, int refi)
{ // xeve_eco_refi (xeve_eco.c:1158-1188)
    if(num_refp <= 1) return;
    Q.ctx(XEVE_HIP_CTX_REFI, refi != 0);
    if(refi == 0) return;
    for(int i = 2; i < num_refp; i++) {
        const unsigned bin = i != refi + 1;
        if(i == 2) Q.ctx(XEVE_HIP_CTX_REFI + 1, bin);
        else Q.ep(bin);
        if(!bin) break;
    }
}

// returns the mask of components whose coefficients follow
__device__ __forceinline__ unsigned q_header(Queue &Q, const xeve_hip_cu_bits_job &J, const CuBitsK &P)
{
    const int st = P.slice_type;
    if(J.mode == XEVE_HIP_BITS_CU_SKIP) { // xeve_mode.c:276-295
        if(st != 2) {
            Q.ctx(XEVE_HIP_CTX_SKIP_FLAG + J.ctx_skip, 1);
            q_mvp_idx(Q, J.mvp_idx[0]);
            if(st == 0) q_mvp_idx(Q, J.mvp_idx[1]);
        }
        return 0;
    }
    if(J.mode == XEVE_HIP_BITS_MVP) { // xeve_rdo_bit_cnt_mvp (xeve_mode.c:57-79): what check_best_mvp prices
        if(st != 2 && J.refi[0] >= 0) q_mvp_idx(Q, J.mvp_idx[0]), q_mvd1(Q, J.mvd[0][0]), q_mvd1(Q, J.mvd[0][1]);
        if(st == 0 && J.refi[1] >= 0) q_mvp_idx(Q, J.mvp_idx[1]), q_mvd1(Q, J.mvd[1][0]), q_mvd1(Q, J.mvd[1][1]);
        return 0;
    }
    if(J.mode == XEVE_HIP_BITS_ECO_COEF) { // ctx->fn_eco_coef on its own: xeve_eco_cbf (xeve_eco.c:793-894) for an inter or an intra CU
        const unsigned f = J.dir_flag, run = (f >> 2) & 7, cbf = (J.nnz[0] ? 1u : 0u) | (J.nnz[1] ? 2u : 0u) | (J.nnz[2] ? 4u : 0u);
        if(!(f & XEVE_HIP_ECO_INTRA)) {
            if(!(f & XEVE_HIP_ECO_NO_CBF) && run == 7) {
                Q.ctx(XEVE_HIP_CTX_CBF_ALL, (cbf & run) != 0);
                if(!(cbf & run)) return 0;
            }
            if((run & 2) && P.idc) Q.ctx(XEVE_HIP_CTX_CBF_CB, (cbf >> 1) & 1);
            if((run & 4) && P.idc) Q.ctx(XEVE_HIP_CTX_CBF_CR, (cbf >> 2) & 1);
            if((run & 1) && (cbf & 6)) Q.ctx(XEVE_HIP_CTX_CBF_LUMA, cbf & 1);
        }
        else {
            if((run & 2) && P.idc) Q.ctx(XEVE_HIP_CTX_CBF_CB, (cbf >> 1) & 1);
            if((run & 4) && P.idc) Q.ctx(XEVE_HIP_CTX_CBF_CR, (cbf >> 2) & 1);
            if(run & 1) Q.ctx(XEVE_HIP_CTX_CBF_LUMA, cbf & 1);
        }
        return cbf & run;
    }
    if(J.mode == XEVE_HIP_BITS_INTRA_DIR) { // xeve_rdo_bit_cnt_intra_dir (xeve_mode.c:136-139)
        q_intra_dir(Q, J.mvp_idx[0]);
        return 0;
    }
    if(J.mode == XEVE_HIP_BITS_CU_INTRA || J.mode == XEVE_HIP_BITS_INTRA_LUMA) { // xeve_rdo_bit_cnt_cu_intra (xeve_mode.c:141-175) / _cu_intra_luma (:81-117), Baseline
        if(st != 2) {
            Q.ctx(XEVE_HIP_CTX_SKIP_FLAG + J.ctx_skip, 0);
            Q.ctx(XEVE_HIP_CTX_PRED_MODE + J.ctx_pred_mode, 1); // xeve_eco_pred_mode(MODE_INTRA)
        }
        q_intra_dir(Q, J.mvp_idx[0]);
        const unsigned runi = J.mode == XEVE_HIP_BITS_CU_INTRA ? 7u : 1u, cbfi = (J.nnz[0] ? 1u : 0u) | (J.nnz[1] ? 2u : 0u) | (J.nnz[2] ? 4u : 0u);
        if((runi & 2) && P.idc) Q.ctx(XEVE_HIP_CTX_CBF_CB, (cbfi >> 1) & 1); // xeve_eco_cbf, intra branch (xeve_eco.c:864-890)
        if((runi & 4) && P.idc) Q.ctx(XEVE_HIP_CTX_CBF_CR, (cbfi >> 2) & 1);
        Q.ctx(XEVE_HIP_CTX_CBF_LUMA, cbfi & 1);
        return cbfi & runi;
    }
    unsigned run = 7;
    if(J.mode == XEVE_HIP_BITS_CU_INTER) { // xeve_mode.c:201-274
        if(st != 2) {
            Q.ctx(XEVE_HIP_CTX_SKIP_FLAG + J.ctx_skip, 0);
            Q.ctx(XEVE_HIP_CTX_PRED_MODE + J.ctx_pred_mode, 0);
            Q.ctx(XEVE_HIP_CTX_DIRECT, J.dir_flag != 0);
            if(!J.dir_flag) {
                const bool v0 = J.refi[0] >= 0, v1 = J.refi[1] >= 0;
                if(v0 && v1) Q.ctx(XEVE_HIP_CTX_INTER_DIR, 0); // xeve_eco_inter_pred_idc (xeve_eco.c:1123-1156)
                else {
                    if(st == 0) Q.ctx(XEVE_HIP_CTX_INTER_DIR, 1);
                    Q.ctx(XEVE_HIP_CTX_INTER_DIR + 1, !v0);
                }
                if(v0) {
                    q_refi(Q, P.num_refp[0], J.refi[0]);
                    q_mvp_idx(Q, J.mvp_idx[0]);
                    q_mvd1(Q, J.mvd[0][0]), q_mvd1(Q, J.mvd[0][1]);
                }
                if(st == 0 && v1) {
                    q_refi(Q, P.num_refp[1], J.refi[1]);
                    q_mvp_idx(Q, J.mvp_idx[1]);
                    q_mvd1(Q, J.mvd[1][0]), q_mvd1(Q, J.mvd[1][1]);
                }
            }
        }
    }
    else run = 1u << (J.mode - 1);
    // xeve_eco_cbf (xeve_eco.c:793-894), inter branch, one transform block
    const unsigned cbf = (J.nnz[0] ? 1u : 0u) | (J.nnz[1] ? 2u : 0u) | (J.nnz[2] ? 4u : 0u);
    if(run == 7) {
        Q.ctx(XEVE_HIP_CTX_CBF_ALL, cbf != 0);
        if(!cbf) return 0;
    }
    if((run & 2) && P.idc) Q.ctx(XEVE_HIP_CTX_CBF_CB, (cbf >> 1) & 1);
    if((run & 4) && P.idc) Q.ctx(XEVE_HIP_CTX_CBF_CR, (cbf >> 2) & 1);
    if((run & 1) && (cbf & 6)) Q.ctx(XEVE_HIP_CTX_CBF_LUMA, cbf & 1);
    return cbf & run;
}

// ---- one lane per job ------------------------------------------------------------------------------------------------
// The models live in LDS ([model][lane], one wave per workgroup), the header bins are queued per lane and drained by one loop, and the coefficient bins
// of every coded block come from the block's bin string (k_coef_events): per bin the byte, its model (read from LDS one bin AHEAD, with the just-written
// model forwarded when two consecutive bins share it), the coder step, the model write-back.  The string is fetched 16 bytes at a time, the next chunk
// while the current one is coded, so neither the global nor the LDS latency sits on the serial chain.  A block without a usable string (BIN_OVF: levels so
// large that the string does not fit; or a job whose coefficient count is not the block's true count -- the reference then stops after nnz events) goes
// through code_events: the same bins generated on the fly from the event list, one at a time (exact, slow, rare).
typedef u32x4 u32x4_a4g __attribute__((aligned(4)));
typedef uint16_t (*CtxTab)[64];

template <bool FULL> __device__ __forceinline__ void code_queue(Sbac &s, CtxTab s_ctx, const uint8_t (*s_q)[64], int lane, int n)
{
    for(int i = 0; i < n; i++) {
        const unsigned e = s_q[i][lane], ci = e >> 1;
        const unsigned m = sb_encode<FULL>(s, s_ctx[ci][lane], e & 1, ci == BYP); // (bypass entries pass the dummy row through)
        s_ctx[ci][lane] = (uint16_t)m;
        if(FULL && (i & 3) == 3) sb_drain(s);
    }
    if(FULL) sb_drain(s);
}

template <bool FULL> __device__ __forceinline__ void code_string(Sbac &s, CtxTab s_ctx, int lane, const unsigned char *sp, int rem)
{
    if(rem <= 0) return;
    const u32x4_a4g *gp = reinterpret_cast<const u32x4_a4g *>(sp);
    u32x4 cur = gp[0];
    unsigned ci = (cur.x & 0xFFu) >> 1;
    ci = ci < BYP ? ci : BYP;
    unsigned m = s_ctx[ci][lane];
    auto step = [&](unsigned b, unsigned b1) { // code bin b on model ci (value m); b1: the byte after it
        unsigned ci1 = (b1 & 0xFFu) >> 1;
        ci1 = ci1 < BYP ? ci1 : BYP; // (past the end of the string: any byte)
        const unsigned mp = s_ctx[ci1][lane];
        const unsigned m1 = sb_encode<FULL>(s, m, b & 1u, ci == BYP);
        s_ctx[ci][lane] = (uint16_t)m1;
        m = ci1 == ci ? m1 : mp, ci = ci1;
    };
    while(rem > 0) {
        gp++;
        const u32x4 nxt = gp[0]; // (reads up to 31 bytes past the string: inside the block's region, the next one, or the slack behind the last)
        if(rem >= 16) {
#pragma unroll
            for(int i = 0; i < 16; i++) {
                const unsigned wv = cur[i >> 2], wn = i < 15 ? cur[(i + 1) >> 2] : nxt.x;
                step(wv >> (8 * (i & 3)), wn >> (8 * ((i + 1) & 3)));
                if(FULL && (i & 3) == 3) sb_drain(s);
            }
        }
        else {
            for(int i = 0; i < rem; i++) {
                const int i1 = i + 1;
                const unsigned wv = (i >> 2) == 0 ? cur.x : (i >> 2) == 1 ? cur.y : (i >> 2) == 2 ? cur.z : cur.w;
                const unsigned wn = (i1 >> 2) == 0 ? cur.x : (i1 >> 2) == 1 ? cur.y : (i1 >> 2) == 2 ? cur.z : cur.w; // (i1 <= 15 here)
                step(wv >> (8 * (i & 3)), wn >> (8 * (i1 & 3)));
                if(FULL && (i & 3) == 3) sb_drain(s);
            }
            if(FULL) sb_drain(s);
        }
        cur = nxt, rem -= 16;
    }
}

// xeve_eco_run_length_cc (xeve_eco.c:707-771) straight from the event list: `nevents` events, the last-position flag driven by num_sig as the reference's
template <bool FULL>
__device__ __forceinline__ void code_events(Sbac &s, CtxTab s_ctx, int lane, const unsigned *__restrict__ ev, int nevents, int num_sig, int ch, int cm_init)
{
    auto bin = [&](int ci, unsigned b) {
        const unsigned m = sb_encode<FULL>(s, s_ctx[ci][lane], b, false);
        s_ctx[ci][lane] = (uint16_t)m;
        if(FULL) sb_drain(s); // (the slow path: every bin)
    };
    unsigned plev = 5; // min(previous level - 1, 5); 5 at the start of a block
    for(int e = 0; e < nevents; e++) {
        const unsigned cur = ev[e], run = (cur >> 16) & 0xFFF, lev1 = cur & 0x7FFF, sign = (cur >> 15) & 1, at_end = (cur >> 28) & 1;
        const int t0 = cm_init == 1 ? (int)(plev << 1) + ch * 12 : ch * 2;
        bin(XEVE_HIP_CTX_RUN + t0, run != 0);
        if(run) {
            for(unsigned i = 1; i < run; i++) bin(XEVE_HIP_CTX_RUN + t0 + 1, 1);
            bin(XEVE_HIP_CTX_RUN + t0 + 1, 0);
        }
        bin(XEVE_HIP_CTX_LEVEL + t0, lev1 != 0);
        if(lev1) {
            for(unsigned i = 1; i < lev1; i++) bin(XEVE_HIP_CTX_LEVEL + t0 + 1, 1);
            bin(XEVE_HIP_CTX_LEVEL + t0 + 1, 0);
        }
        (void)sb_encode<FULL>(s, 0, sign, true);
        if(FULL) sb_drain(s);
        num_sig--;
        if(!at_end) bin(XEVE_HIP_CTX_LAST + ch, num_sig == 0);
        plev = lev1 < 5 ? lev1 : 5u;
    }
}

// the coefficients of component c of a job: from the block's bin string when it has one and the job's count is the block's, else from the events
template <bool FULL>
__device__ __forceinline__ void code_block(Sbac &s, CtxTab s_ctx, int lane, const unsigned char *__restrict__ bins, const unsigned *__restrict__ ev, int coef_off, int nnz,
                                           int c, int cm_init, unsigned long long *slow)
{
    const unsigned *hdr = reinterpret_cast<const unsigned *>(bins + (size_t)coef_off * BINK);
    const unsigned nb = hdr[0], nev = hdr[1];
    if(nb != BIN_OVF && nev == (unsigned)nnz) code_string<FULL>(s, s_ctx, lane, reinterpret_cast<const unsigned char *>(hdr) + BIN_HDR, (int)nb);
    else {
        if(slow) atomicAdd(XH_PROF_SLOT(slow), 1ull); // measurement only: blocks coded from their event lists
        code_events<FULL>(s, s_ctx, lane, ev + coef_off, (int)(nev < (unsigned)nnz ? nev : (unsigned)nnz), nnz, c != 0, cm_init);
    }
}

__device__ __forceinline__ void sbac_load(Sbac &s, const xeve_hip_sbac &in, CtxTab s_ctx, int lane, bool continue_coder, const uint32_t *s_tab)
{
    s.tab = s_tab;
    s.range = in.range, s.shifts = s.bins = 0;
    s.code = in.code & 0x7FFFF, s.fifo = 0, s.cnt = 0, s.cb = 11, s.sff = s.sz = s.pb = s.ipb = s.bc = 0; // SBAC_LOAD + xeve_sbac_bit_reset (xeve_mode.c:39-49)
    if(continue_coder) { // continue the coder where the state stands
        s.code = in.code, s.cb = in.code_bits, s.sff = in.stacked_ff, s.sz = in.stacked_zero, s.pb = in.pending_byte, s.ipb = in.is_pending_byte;
        s.bc = in.bitcounter, s.bins = in.bin_counter;
    }
    for(int i = 0; i < NCTX; i++) s_ctx[i][lane] = in.ctx[i];
    s_ctx[BYP][lane] = 0;
}

__device__ unsigned long long g_bins_hist[16 * 32]; // (16 stripes of 32 buckets: same-address atomics of thousands of lanes serialise)
// bins-per-job histogram of every k_cu_bits launch that ran with the class timer on (xeve_hip_prof_enable): out[b] = jobs whose bin count has bit length b (b = 0: no bin);
// reset != 0 clears it.  Measurement only.
extern "C" int xeve_hip_prof_cu_bits_hist(unsigned long long *out, int reset)
{
    XH_ENTER();
    unsigned long long h[16 * 32];
    XH_HIP(hipDeviceSynchronize());
    XH_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bins_hist), sizeof(h)));
    if(out)
        for(int b = 0; b < 32; b++) {
            out[b] = 0;
            for(int s = 0; s < 16; s++) out[b] += h[s * 32 + b];
        }
    if(reset) {
        memset(h, 0, sizeof(h));
        XH_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_bins_hist), h, sizeof(h)));
    }
    return XEVE_HIP_OK;
}

template <bool FULL>
__global__ __launch_bounds__(64) void k_cu_bits(const xeve_hip_sbac *__restrict__ sin, const xeve_hip_cu_bits_job *__restrict__ jobs, int njobs, CuBitsK P,
                                                const unsigned char *__restrict__ bins, const unsigned *__restrict__ ev, unsigned *__restrict__ bits,
                                                xeve_hip_sbac *__restrict__ sout, unsigned long long *__restrict__ units, unsigned long long *__restrict__ slow)
{
    __shared__ uint16_t s_ctx[NCTX + 1][64]; // (+ the dummy row of the bypass bins)
    __shared__ uint8_t  s_q[QCAP][64];
    __shared__ uint32_t s_tab[1024];
    load_model_tab(s_tab);
    const int lane = threadIdx.x, j = blockIdx.x * 64 + lane;
    if(j >= njobs) return;
    const xeve_hip_cu_bits_job J = jobs[j];
    const xeve_hip_sbac &in = sin[J.sbac];
    const bool cont = FULL && J.mode == XEVE_HIP_BITS_ECO_COEF && (J.dir_flag & XEVE_HIP_ECO_NO_RESET);
    Sbac s;
    sbac_load(s, in, s_ctx, lane, cont, s_tab);
    Queue Q{&s_q[0][lane], 0, 0};
    const unsigned coded = q_header(Q, J, P);
    code_queue<FULL>(s, s_ctx, s_q, lane, Q.n < QCAP ? Q.n : QCAP);
    for(int base = QCAP; base < Q.n; base += QCAP) { // (a header longer than the window: rare)
        Queue W{&s_q[0][lane], 0, base};
        (void)q_header(W, J, P);
        code_queue<FULL>(s, s_ctx, s_q, lane, Q.n - base < QCAP ? Q.n - base : QCAP);
    }
    for(int c = 0; c < 3; c++)
        if((coded >> c) & 1) code_block<FULL>(s, s_ctx, lane, bins, ev, J.coef_off[c], J.nnz[c], c, P.cm_init, slow);
    bits[j] = s.shifts;
    if(units) { // measurement only: the lane's bins into the histogram of bins per job (bucket = bit length of the count: 0, 1, 2-3, 4-7, ...; xeve_hip_prof_cu_bits_hist)
        const unsigned nb = s.bins - (cont ? in.bin_counter : 0u);
        atomicAdd(&g_bins_hist[((blockIdx.x & 15) << 5) + (nb ? 32 - __clz(nb) : 0)], 1ull);
    }
    if(units) { // measurement only (xeve_hip_prof_*): bins coded by this wave (lanes may have left: the sum goes through one LDS word)
        unsigned *cnt = reinterpret_cast<unsigned *>(&s_q[0][0]);
        const bool first = lane == (int)(__ffsll((long long)__ballot(true)) - 1);
        if(first) *cnt = 0;
        atomicAdd(cnt, s.bins - (cont ? in.bin_counter : 0u));
        if(first) atomicAdd(XH_PROF_SLOT(units), (unsigned long long)*cnt);
    }
    if(!FULL && sout) { // what feeds forward into later bit counts: the range and the models (xeve_sbac_bit_reset discards the rest but the
                        // low bits of the code register, and those never reach a bit count)
        xeve_hip_sbac &o = sout[j];
        o.range = s.range, o.code = 0, o.code_bits = 11, o.stacked_ff = o.stacked_zero = o.pending_byte = o.is_pending_byte = o.bitcounter = 0;
        o.bin_counter = s.bins;
        for(int i = 0; i < NCTX; i++) o.ctx[i] = s_ctx[i][lane];
    }
    if(FULL) {
        sb_drain(s);
        // xeve_get_bit_number (xeve_mode.c:51-55) -- equal to s.shifts, kept as the reference computes it
        bits[j] = s.bc + 8 * (s.sz + s.sff) + 8 * (s.ipb ? 1 : 0) + 8 - s.cb + 3;
        xeve_hip_sbac &o = sout[j];
        o.range = s.range, o.code = s.code, o.code_bits = s.cb, o.stacked_ff = s.sff, o.stacked_zero = s.sz;
        o.pending_byte = s.pb, o.is_pending_byte = s.ipb, o.bitcounter = s.bc, o.bin_counter = s.bins;
        for(int i = 0; i < NCTX; i++) o.ctx[i] = s_ctx[i][lane];
    }
}

// ---- the per-component cbf tests of pinter_residue_rdo as ONE job per (Y choice, U choice) ---------------------------------------------
// xeve_pinter.c:1180-1218 tests every component with and without its coefficients, handing the coder state of the cheaper alternative to the next
// component: three dependent bit-count rounds.  A lane here ASSUMES the outcome of the Y and of the U test (job.dir_flag bit 0 / bit 1: coefficients kept)
// and codes the whole chain under that assumption -- Y alternative, bit reset, U alternative, bit reset, then both V alternatives (the one-bin "without"
// on a copy of range and model) -- so that all four assumptions run side by side in one launch and the decision afterwards reads the lane that matches
// what the costs picked.  out[4 * j + {0, 1, 2, 3}] = bits of the Y segment, the U segment, V without, V with.  job.nnz = the stored counts; a component
// whose stored count is 0 has no test and no segment (the state passes through).  Count-only: these states are only ever loaded into further counts.
__global__ __launch_bounds__(64) void k_cu_bits_chain(const xeve_hip_sbac *__restrict__ sin, const xeve_hip_cu_bits_job *__restrict__ jobs, int njobs, CuBitsK P,
                                                      const unsigned char *__restrict__ bins, const unsigned *__restrict__ ev, unsigned *__restrict__ out,
                                                      unsigned long long *__restrict__ units, unsigned long long *__restrict__ slow)
{
    __shared__ uint16_t s_ctx[NCTX + 1][64];
    __shared__ uint8_t  s_q[QCAP][64];
    __shared__ uint32_t s_tab[1024];
    load_model_tab(s_tab);
    const int lane = threadIdx.x, j = blockIdx.x * 64 + lane;
    if(j >= njobs) return;
    const xeve_hip_cu_bits_job J = jobs[j];
    if(J.mode == XEVE_HIP_BITS_CU_SKIP) return; // lane switched off (an assumption that cannot occur: the component has no coefficients)
    Sbac s;
    sbac_load(s, sin[J.sbac], s_ctx, lane, false, s_tab);
    const int keep[3] = {J.dir_flag & 1, (J.dir_flag >> 1) & 1, 1};
    unsigned total_bins = 0;
    for(int c = 0; c < 3; c++) {
        if(J.nnz[c] <= 0) continue; // no test for a component without coefficients (:1182)
        xeve_hip_cu_bits_job T = J;
        T.mode = (uint8_t)(XEVE_HIP_BITS_COMP_Y + c);
        if(c == 2) { // V without: one cbf bin, on a copy (range + that model; nothing is written back)
            T.nnz[2] = 0;
            Queue Q0{&s_q[0][lane], 0, 0};
            (void)q_header(Q0, T, P);
            Sbac t = s;
            for(int i = 0; i < Q0.n; i++) {
                const unsigned e = s_q[i][lane];
                (void)sb_encode<false>(t, s_ctx[e >> 1][lane], e & 1, (e >> 1) == BYP); // (a single header bin: no model is used twice)
            }
            out[4 * j + 2] = t.shifts;
            T.nnz[2] = J.nnz[2];
        }
        else if(!keep[c]) T.nnz[c] = 0;
        Queue Q{&s_q[0][lane], 0, 0}; // (a component test's header is the cbf flags: a few bins)
        const unsigned coded = q_header(Q, T, P);
        code_queue<false>(s, s_ctx, s_q, lane, Q.n);
        if((coded >> c) & 1) code_block<false>(s, s_ctx, lane, bins, ev, J.coef_off[c], J.nnz[c], c, P.cm_init, slow);
        out[4 * j + (c == 2 ? 3 : c)] = s.shifts;
        total_bins += s.bins;
        s.shifts = 0, s.bins = 0; // xeve_sbac_bit_reset before the next component's test
    }
    if(units) {
        unsigned *cnt = reinterpret_cast<unsigned *>(&s_q[0][0]);
        const bool first = lane == (int)(__ffsll((long long)__ballot(true)) - 1);
        if(first) *cnt = 0;
        atomicAdd(cnt, total_bins);
        if(first) atomicAdd(XH_PROF_SLOT(units), (unsigned long long)*cnt);
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------------
// layout: event lists [4 B x coef_elems] | bin strings [BINK B x coef_elems + slack] | event counts [3 x njobs ints] | done flags [njobs]
// (the first two do not move with njobs: the rounds of one RDO batch pass different job counts over the same coefficient buffer and workspace)
static size_t ws_bins_off(size_t coef_elems) { return (sizeof(unsigned) * coef_elems + 255) & ~(size_t)255; }
static size_t ws_nev_off(size_t coef_elems) { return ws_bins_off(coef_elems) + (((size_t)BINK * coef_elems + 64 + 255) & ~(size_t)255); }
extern "C" size_t xeve_hip_cu_bits_workspace(int njobs, size_t coef_elems)
{
    const size_t n = njobs > 0 ? njobs : 0;
    return ws_nev_off(coef_elems) + ((sizeof(int) * 3 * n + 255) & ~(size_t)255) + n + 256;
}

static int cu_bits_launch(const int16_t *coef, size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs,
                          const xeve_hip_cu_bits_params *p, void *workspace, size_t workspace_bytes, uint32_t *bits, xeve_hip_sbac *sbac_out, bool full,
                          void *stream, bool reuse_events = false, int ev_first = 0, int ev_count = -1);

extern "C" int xeve_hip_cu_bits_jobs(const int16_t *coef, size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs,
                                     const xeve_hip_cu_bits_params *p, void *workspace, size_t workspace_bytes, uint32_t *bits,
                                     xeve_hip_sbac *sbac_out, void *stream)
{
    return cu_bits_launch(coef, coef_elems, sbac_in, jobs, njobs, p, workspace, workspace_bytes, bits, sbac_out, sbac_out != nullptr, stream);
}

// count-only kernel, but handing on what later bit counts depend on (range + context models; every other field of state_out is reset)
extern "C" int xeve_hip_cu_bits_jobs_chain(const int16_t *coef, size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs,
                                           const xeve_hip_cu_bits_params *p, void *workspace, size_t workspace_bytes, uint32_t *bits,
                                           xeve_hip_sbac *state_out, void *stream)
{
    return cu_bits_launch(coef, coef_elems, sbac_in, jobs, njobs, p, workspace, workspace_bytes, bits, state_out, false, stream);
}

// A chain of bit-count rounds over ONE coefficient buffer (pinter_residue_rdo's four) needs the event lists only once: later rounds pass
// reuse = 1 with the same workspace (jobs' nnz must then be the exact non-zero counts, as RDOQ returns them).
int xh_cu_bits_jobs_round(const int16_t *coef, size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs,
                          const xeve_hip_cu_bits_params *p, void *workspace, size_t workspace_bytes, uint32_t *bits, xeve_hip_sbac *state_out, int full, int reuse,
                          void *stream, int ev_first, int ev_count)
{
    return cu_bits_launch(coef, coef_elems, sbac_in, jobs, njobs, p, workspace, workspace_bytes, bits, state_out, full != 0, stream, reuse != 0, ev_first, ev_count);
}

static int fill_k(CuBitsK &P, const xeve_hip_cu_bits_params *p)
{
    const int ws = p->chroma_format_idc <= 2, hs = p->chroma_format_idc <= 1; // XEVE_GET_CHROMA_{W,H}_SHIFT (xeve_util.h:92-94)
    for(int c = 0; c < 3; c++) {
        const int lw = p->log2_cuw - (c ? ws : 0), lh = p->log2_cuh - (c ? hs : 0);
        P.log2n[c] = lw + lh, P.n[c] = 1 << (lw + lh);
        const int rc = xh_get_scan(lw, lh, &P.scan[c]);
        if(rc != XEVE_HIP_OK) return rc;
    }
    P.slice_type = p->slice_type, P.num_refp[0] = p->num_refp[0], P.num_refp[1] = p->num_refp[1];
    P.cm_init = p->cm_init, P.idc = p->chroma_format_idc;
    return XEVE_HIP_OK;
}

// the chain jobs of one RDO batch (k_cu_bits_chain) over the event lists / bin strings an earlier xh_cu_bits_jobs_round over the SAME coefficient buffer and
// workspace left behind; out: 4 counts per job
int xh_cu_bits_chain_round(size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs, const xeve_hip_cu_bits_params *p,
                           void *workspace, size_t workspace_bytes, uint32_t *out, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(p && njobs >= 0 && sbac_in && jobs && out && workspace && workspace_bytes >= xeve_hip_cu_bits_workspace(njobs, coef_elems));
    if(njobs == 0) return XEVE_HIP_OK;
    CuBitsK P;
    const int rc = fill_k(P, p);
    if(rc != XEVE_HIP_OK) return rc;
    char *W = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    XhProf prof(XH_PROF_CU_BITS, st);
    k_cu_bits_chain<<<(njobs + 63) / 64, 64, 0, st>>>(sbac_in, jobs, njobs, P, (const unsigned char *)(W + ws_bins_off(coef_elems)), (const unsigned *)W, out,
                                                     xh_prof_units(XH_PROF_CU_BITS), xh_prof_units(XH_PROF_CU_BITS_SLOW));
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

static int cu_bits_launch(const int16_t *coef, size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs,
                          const xeve_hip_cu_bits_params *p, void *workspace, size_t workspace_bytes, uint32_t *bits, xeve_hip_sbac *sbac_out, bool full,
                          void *stream, bool reuse_events, int ev_first, int ev_count)
{
    XH_ENTER();
    XH_REQUIRE(p && njobs >= 0);
    XH_REQUIRE(p->log2_cuw >= 2 && p->log2_cuw <= 6 && p->log2_cuh >= 2 && p->log2_cuh <= 6);
    XH_REQUIRE(p->slice_type >= 0 && p->slice_type <= 2 && p->chroma_format_idc >= 0 && p->chroma_format_idc <= 3);
    XH_REQUIRE(p->num_refp[0] >= 0 && p->num_refp[0] <= 21 && p->num_refp[1] >= 0 && p->num_refp[1] <= 21);
    if(njobs == 0) return XEVE_HIP_OK;
    if(xh_count_states()) full = false; // (xh_common.h: the caller only ever loads these states into further counts)
    XH_REQUIRE(sbac_in && jobs && bits && workspace); // coef == NULL: no job codes coefficients (skip / mvp jobs only) -- the event pass is left out
    XH_REQUIRE(workspace_bytes >= xeve_hip_cu_bits_workspace(njobs, coef_elems));
    CuBitsK P;
    {
        const int rc = fill_k(P, p);
        if(rc != XEVE_HIP_OK) return rc;
    }
    char          *W    = (char *)workspace;
    unsigned      *ev   = (unsigned *)W;
    unsigned char *bins = (unsigned char *)(W + ws_bins_off(coef_elems));
    int           *nev  = (int *)(W + ws_nev_off(coef_elems));
    XH_REQUIRE(((uintptr_t)workspace & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    // the event / bin pass over jobs [ev_first, ev_first + ev_count) only: a caller whose jobs share blocks names a sub-range that covers every coded block once
    if(ev_count < 0) ev_first = 0, ev_count = njobs;
    XH_REQUIRE(ev_first >= 0 && ev_first + ev_count <= njobs);
    const long items = 3L * ev_count;
    if(!coef || reuse_events || ev_count == 0) {} // (no coefficient buffer: header-only jobs; reuse: the lists and strings of an earlier round over the same buffer)
    else if(P.n[0] <= 64) {
        const long waves = (items + 3) / 4;
        k_coef_events<16><<<(unsigned)((waves + 3) / 4), 256, 0, st>>>(coef, jobs + ev_first, ev_count, P, ev, nev, bins);
    }
    else k_coef_events<64><<<(unsigned)((items + 3) / 4), 256, 0, st>>>(coef, jobs + ev_first, ev_count, P, ev, nev, bins);
    {
        XhProf prof(XH_PROF_CU_BITS, st);
        unsigned long long *units = xh_prof_units(XH_PROF_CU_BITS), *slow = xh_prof_units(XH_PROF_CU_BITS_SLOW);
        if(full) k_cu_bits<true><<<(njobs + 63) / 64, 64, 0, st>>>(sbac_in, jobs, njobs, P, bins, ev, bits, sbac_out, units, slow);
        else k_cu_bits<false><<<(njobs + 63) / 64, 64, 0, st>>>(sbac_in, jobs, njobs, P, bins, ev, bits, sbac_out, units, slow);
    }
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}


// ---- host-memory form of one xeve_eco_coef call in bit-count mode (the table layer's style) -----------------------------------------
// What ctx->fn_eco_coef can be pointed at while the encoder is counting bits (sbac->is_bitcount): the cbf flags and the coefficients of one
// CU go through the GPU coder, continuing from *state exactly where it stands, and *state is left as the reference's coder would leave it.
extern "C" int xeve_hip_eco_coef_host(xeve_hip_sbac *state, const int16_t *coef_y, const int16_t *coef_u, const int16_t *coef_v, int log2_cuw, int log2_cuh,
                                      const int32_t nnz[3], int flags, int chroma_format_idc, int cm_init)
{
    XH_ENTER();
    XH_REQUIRE(state && coef_y && nnz && log2_cuw >= 2 && log2_cuw <= 6 && log2_cuh >= 2 && log2_cuh <= 6 && chroma_format_idc >= 0 && chroma_format_idc <= 3);
    XH_REQUIRE(chroma_format_idc == 0 || (coef_u && coef_v));
    const int ws = chroma_format_idc <= 2, hs = chroma_format_idc <= 1;
    const size_t n0 = (size_t)1 << (log2_cuw + log2_cuh), n1 = chroma_format_idc ? n0 >> (ws + hs) : 0, ne = n0 + 2 * n1;
    xeve_hip_cu_bits_params p;
    p.log2_cuw = log2_cuw, p.log2_cuh = log2_cuh, p.slice_type = 0, p.num_refp[0] = p.num_refp[1] = 0, p.cm_init = cm_init, p.chroma_format_idc = chroma_format_idc;
    xeve_hip_cu_bits_job j;
    memset(&j, 0, sizeof(j));
    j.coef_off[0] = 0, j.coef_off[1] = (int)n0, j.coef_off[2] = (int)(n0 + n1), j.nnz[0] = nnz[0], j.nnz[1] = nnz[1], j.nnz[2] = nnz[2];
    j.mode = XEVE_HIP_BITS_ECO_COEF, j.dir_flag = (uint8_t)(flags | XEVE_HIP_ECO_NO_RESET);
    const size_t o_job = (ne * 2 + 255) & ~(size_t)255, o_st = o_job + 256, o_bits = o_st + 2 * 256, o_ws = o_bits + 256;
    const size_t wsb = xeve_hip_cu_bits_workspace(1, ne);
    char *d = nullptr;
    XH_HIP(hipMalloc((void **)&d, o_ws + wsb));
    int rc = XEVE_HIP_OK;
    auto up = [&](size_t off, const void *src, size_t bytes) { if(rc == XEVE_HIP_OK && hipMemcpy(d + off, src, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = XEVE_HIP_ERR_DEVICE; };
    up(0, coef_y, n0 * 2);
    if(n1) up(n0 * 2, coef_u, n1 * 2), up((n0 + n1) * 2, coef_v, n1 * 2);
    up(o_job, &j, sizeof(j)), up(o_st, state, sizeof(*state));
    if(rc == XEVE_HIP_OK)
        rc = xeve_hip_cu_bits_jobs((const int16_t *)d, ne, (const xeve_hip_sbac *)(d + o_st), (const xeve_hip_cu_bits_job *)(d + o_job), 1, &p, d + o_ws, wsb,
                                   (uint32_t *)(d + o_bits), (xeve_hip_sbac *)(d + o_st + 256), nullptr);
    else xh_set_error("xeve_hip_eco_coef_host: staging failed");
    if(rc == XEVE_HIP_OK && hipMemcpy(state, d + o_st + 256, sizeof(*state), hipMemcpyDeviceToHost) != hipSuccess) {
        xh_set_error("xeve_hip_eco_coef_host: copy back failed");
        rc = XEVE_HIP_ERR_DEVICE;
    }
    (void)hipFree(d);
    return rc;
}
