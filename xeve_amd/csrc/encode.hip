// xeve_amd/csrc/encode.hip -- the closed-GOP batch encoder on the device: the engine behind enc_host.h's frame loop + the C-ABI entry points xeve_hip_enc_*.
//
// What the reference does per picture (xeve_pic, src_base/xeve_enc.c:226-600) happens here for G pictures at once -- the same picture of G independent closed GOPs:
//   begin_picture   8-bit original -> 10-bit planes (the application's conversion), maps cleared (xeve_pic_prepare :1220-1237)
//   step            one CTU of every row chain of every picture: xeve_hip_mode_analyze_ctu_jobs (mode_analyze_lcu), then the writer on the chain's coder
//                   (xeve_eco_tree, :152), whose state the chain's next CTU starts from (:138-139)
//   end_picture     xeve_hip_deblock (:462), the slice data -- every CTU written again in raster order on a fresh coder + the tile's end (:466-560), or chain 0's own
//                   bytes when the picture has one chain --, xeve_hip_picbuf_expand (xeve_pic_finish)
// Everything a GOP needs stays in HBM: the frames (8 bit), the current original, the picture stores with their motion maps, the unit maps, the decided CTUs.  The
// pictures of the batch are STACKED VERTICALLY (xh_common.h): GOP g's planes and maps lie g * vh luma rows below GOP 0's, so the intra analysis / the tree operations
// address them with a picture index and element distances (pic_elems) and the inter analysis as one tall picture.
#include <chrono>
#include <cmath>
#include <optional>
#include <cstdlib>
#include <memory>
#include <string>
#include "xh_common.h"
// The writer kernels of this file run eco_lane.h's lane code with the coder's 72 context models in LDS ([model][lane]: no bank conflicts, no scratch) and, every
// function forced inline, the coder core (range, code, pending bytes) and the byte sink in registers.  The lane-serial form with the whole record in scratch
// (round 2: 13 ms per CTU of noise for one chain, 346 ms per step for 2048 chains) was bound by the scratch round trip of every bin's model.
#define XL_NCTX 72
static __shared__ uint16_t xl_lds_ctx[XL_NCTX * 64];
#if defined(__HIP_DEVICE_COMPILE__)
#define XL __host__ __device__ static inline __attribute__((always_inline))
#define XL_CTX(s, ci) xl_lds_ctx[(ci) * 64 + (threadIdx.x & 63)]
#define XL_SINK(o) true // (every coder of this file writes: see cu_lane.h)
#endif
#include "eco_lane.h"
#include "enc_host.h"

using namespace xenc;

namespace {
struct StepDesc { // the CTUs of one lockstep step (the same for every GOP): at most one per row chain
    int n;
    int t[8], x[8], y[8], lcu[8];
};

// frames: [G][F][w * h * 3 / 2] samples of one byte (the application's -d 8) or of two (-d 10, little-endian); one thread per luma sample and its share of the chroma;
// blockIdx.y = GOP.  Both depths land in the codec's 10 bits (imgb_cpy_conv_8b_to_16b / the plain copy of xeve_app.c's imgb_cpy)
template <class SAMPLE, int SHIFT>
__global__ void k_enc_load(const uint8_t *__restrict__ frames, long gop_bytes, long frame_off, pel *__restrict__ y, pel *__restrict__ u, pel *__restrict__ v, int w, int h,
                           long pic_l, long pic_c)
{
    const int     g = blockIdx.y;
    const SAMPLE *f = reinterpret_cast<const SAMPLE *>(frames + g * gop_bytes + frame_off);
    const long    nl = (long)w * h, nc = nl >> 2, i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < nl) y[g * pic_l + i] = (pel)(f[i] << SHIFT); // (the original planes have no padding: stride = width)
    if(i < nc) u[g * pic_c + i] = (pel)(f[nl + i] << SHIFT), v[g * pic_c + i] = (pel)(f[nl + nc + i] << SHIFT);
}
__global__ void k_enc_reset_chain(xeve_hip_sbac *__restrict__ states, int stride, int at, int G)
{ // xeve_sbac_reset (xeve_eco.c:597-620) without sps_cm_init_flag: every model at 1/2
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if(g >= G) return;
    xeve_hip_sbac s;
    memset(&s, 0, sizeof(s));
    s.range = 16384, s.code_bits = 11;
    for(int i = 0; i < (int)(sizeof(s.ctx) / sizeof(s.ctx[0])); i++) s.ctx[i] = 512;
    states[(long)g * stride + at] = s;
}
__global__ void k_enc_jobs(StepDesc D, int G, int T, xeve_hip_ctu_job *__restrict__ jobs)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if(c >= D.n * G) return;
    const int i = c / G, g = c - i * G;
    xeve_hip_ctu_job j;
    j.x = D.x[i] * CTU, j.y = D.y[i] * CTU, j.sbac = g * T + D.t[i], j.pic = g;
    jobs[c] = j;
}
// the decided CTUs of the step into the picture's store, in the writer's form (eco_lane.h CtuSyntax: the ten pieces of the walk's record the writer reads, 21 of its
// 62 KB): 16 bytes per thread and turn; blockIdx.y = chain
struct KeepSeg {
    int src, dst, n16; // byte offsets in xeve_hip_ctu_data / CtuSyntax, length in 16-byte units
};
#define KEEP_SEG(SRC_FIELD, SRC_EXTRA, DST_FIELD, BYTES) \
    { (int)offsetof(xeve_hip_ctu_data, SRC_FIELD) + (SRC_EXTRA), (int)offsetof(xl::CtuSyntax, DST_FIELD), (int)((BYTES) / 16) }
__constant__ const KeepSeg k_keep_segs[10] = {
    KEEP_SEG(split_mode, 0, split_mode, sizeof(((xl::CtuSyntax *)0)->split_mode)),
    KEEP_SEG(pred_mode, 0, pred_mode, 256),
    KEEP_SEG(ipm, 0, ipm0, 256),
    KEEP_SEG(nnz, 0, nnz, 3 * 256 * 4),
    KEEP_SEG(coef, 0, coef_y, 64 * 64 * 2),
    KEEP_SEG(coef, 64 * 64 * 2, coef_u, 32 * 32 * 2), // (a 4:2:0 CTU's chroma levels: the first 32 x 32 of their 64 x 64 planes, pitch 32)
    KEEP_SEG(coef, 2 * 64 * 64 * 2, coef_v, 32 * 32 * 2),
    KEEP_SEG(mvd, 0, mvd, 256 * 8),
    KEEP_SEG(refi, 0, refi, 512),
    KEEP_SEG(mvp_idx, 0, mvp_idx, 512),
};
__global__ void k_enc_keep(const xeve_hip_ctu_data *__restrict__ out, xl::CtuSyntax *__restrict__ store, StepDesc D, int G, int f_lcu)
{
    const int   c = blockIdx.y, i = c / G, g = c - i * G;
    const char *s = reinterpret_cast<const char *>(out + c);
    char       *d = reinterpret_cast<char *>(store + (long)g * f_lcu + D.lcu[i]);
    for(int k = 0; k < 10; k++) {
        const KeepSeg  sg = k_keep_segs[k];
        const uint4   *sp = reinterpret_cast<const uint4 *>(s + sg.src);
        uint4         *dp = reinterpret_cast<uint4 *>(d + sg.dst);
        for(int t = blockIdx.x * blockDim.x + threadIdx.x; t < sg.n16; t += gridDim.x * blockDim.x) dp[t] = sp[t];
    }
}
static_assert(sizeof(xl::CtuSyntax) % 16 == 0 && offsetof(xl::CtuSyntax, pred_mode) % 16 == 0 && offsetof(xl::CtuSyntax, ipm0) % 16 == 0 && offsetof(xl::CtuSyntax, nnz) % 16 == 0 &&
                  offsetof(xl::CtuSyntax, coef_y) % 16 == 0 && offsetof(xl::CtuSyntax, coef_u) % 16 == 0 && offsetof(xl::CtuSyntax, coef_v) % 16 == 0 &&
                  offsetof(xl::CtuSyntax, mvd) % 16 == 0 && offsetof(xl::CtuSyntax, refi) % 16 == 0 && offsetof(xl::CtuSyntax, mvp_idx) % 16 == 0,
              "the writer's record is moved 16 bytes at a time");
static_assert(offsetof(xeve_hip_ctu_data, pred_mode) % 16 == 0 && offsetof(xeve_hip_ctu_data, ipm) % 16 == 0 && offsetof(xeve_hip_ctu_data, nnz) % 16 == 0 &&
                  offsetof(xeve_hip_ctu_data, coef) % 16 == 0 && offsetof(xeve_hip_ctu_data, mvd) % 16 == 0 && offsetof(xeve_hip_ctu_data, refi) % 16 == 0 &&
                  offsetof(xeve_hip_ctu_data, mvp_idx) % 16 == 0 && sizeof(((xeve_hip_ctu_data *)0)->split_mode) == sizeof(((xl::CtuSyntax *)0)->split_mode),
              "the walk's record: the pieces the writer takes start on 16-byte boundaries");
static_assert(sizeof(xeve_hip_ctu_data) % 16 == 0, "CTU records are moved 16 bytes at a time");
static_assert(sizeof(((xeve_hip_sbac *)0)->ctx) == XL_NCTX * sizeof(uint16_t), "the models' LDS image");
__device__ __forceinline__ void sbac_in(xl::Sbac &s, const xeve_hip_sbac *__restrict__ g)
{
    s = *g;
#pragma unroll
    for(int i = 0; i < XL_NCTX; i++) xl_lds_ctx[i * 64 + (threadIdx.x & 63)] = s.ctx[i];
}
__device__ __forceinline__ void sbac_out(xeve_hip_sbac *__restrict__ g, xl::Sbac &s)
{
#pragma unroll
    for(int i = 0; i < XL_NCTX; i++) s.ctx[i] = xl_lds_ctx[i * 64 + (threadIdx.x & 63)];
    *g = s;
}
// the writer of the first pass: chain c writes the CTU it has just decided on its own coder; the bytes are kept only where they are the slice data (one chain per
// picture: cap > 0), appended at pos[g]
template <bool WAVE> __global__ void __launch_bounds__(64) k_enc_write(const xeve_hip_ctu_data *__restrict__ ctus, xeve_hip_sbac *__restrict__ states, xl::EcoParams E, uint32_t *map_scu,
                                                  const int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, long map_pic,
                                                  const xeve_hip_ctu_job *__restrict__ jobs, int nchains, uint8_t *__restrict__ bytes, long cap, int32_t *__restrict__ pos)
{
    // ONE CHAIN PER WAVE, one lane working: an arithmetic coder's control flow follows its data bin by bin, so chains packed into the lanes of a wave run one after
    // the other (measured: 16 chains per wave, 78 ms per CTU of noise; a lone chain, 13 ms) -- a wave per chain keeps every chain at the speed of a lone one, and the
    // chip holds a thousand waves.  WAVE: the 64 lanes run the chain's writer in step (identical state, bins and bytes) and share the scan of the coefficient
    // blocks (eco_lane.h eco_levels); without it lane 0 works alone
    const int c = blockIdx.x;
    if(c >= nchains || (!WAVE && threadIdx.x != 0)) return;
    const xeve_hip_ctu_job J = jobs[c];
    xl::Sbac s;
    sbac_in(s, states + J.sbac);
    // cap == 0 (a second writer pass follows: the bytes of this one are dropped): what the pass leaves is the next CTU's ENTRY STATE, and every reader of that -- the walk's
    // bit counts (xeve_sbac_bit_reset, xeve_mode.c:39-49), this writer's next CTU -- keeps the range and the context models of it and nothing else.  A code-bit count that no
    // shift can exhaust switches the coder's byte output off (sb_shift / sb_shift_n, cu_lane.h: the boundary branch is never taken), the register fields are then
    // normalised so that the record does not depend on how the bins fell
    if(!cap) s.code_bits = 0x3FFFFFFFu;
    const int at = cap ? pos[J.pic] : 0;
    xl::Sink o = {cap ? bytes + (long)J.pic * cap + at : nullptr, cap ? (int)(cap - at) : 0, 0};
    xl::eco_ctu<WAVE>(E, s, ctus[c], map_scu + J.pic * map_pic, map_ipm + J.pic * map_pic, map_tidx + J.pic * map_pic, map_cu_mode + J.pic * map_pic, J.x, J.y, &o);
    if(!cap) s.code = 0, s.code_bits = 11, s.stacked_ff = s.stacked_zero = s.pending_byte = s.is_pending_byte = 0, s.bitcounter = 0;
    sbac_out(states + J.sbac, s);
    if(cap) pos[J.pic] = at + o.n;
}
// the second pass (xeve_enc.c:466-560): GOP g's CTUs [lcu0, lcu1) in raster order on the picture's own coder
template <bool WAVE> __global__ void __launch_bounds__(64) k_enc_rewrite(const xl::CtuSyntax *__restrict__ store, xeve_hip_sbac *__restrict__ states, xl::EcoParams E, uint32_t *map_scu,
                                                    const int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, long map_pic, int G, int f_lcu, int w_lcu,
                                                    int lcu0, int lcu1, uint8_t *__restrict__ bytes, long cap, int32_t *__restrict__ pos)
{
    const int g = blockIdx.x; // (one GOP per wave, one lane working: see k_enc_write)
    if(g >= G || (!WAVE && threadIdx.x != 0)) return;
    xl::Sbac s;
    sbac_in(s, states + g);
    int at = pos[g];
    for(int lcu = lcu0; lcu < lcu1; lcu++) {
        xl::Sink o = {bytes + (long)g * cap + at, (int)(cap - at), 0};
        xl::eco_ctu<WAVE>(E, s, store[(long)g * f_lcu + lcu], map_scu + g * map_pic, map_ipm + g * map_pic, map_tidx + g * map_pic, map_cu_mode + g * map_pic, (lcu % w_lcu) * CTU,
                    (lcu / w_lcu) * CTU, &o);
        at += o.n;
    }
    sbac_out(states + g, s);
    pos[g] = at;
}
__global__ void __launch_bounds__(64) k_enc_tile_end(xeve_hip_sbac *__restrict__ states, int stride, int G, uint8_t *__restrict__ bytes, long cap, int32_t *__restrict__ pos)
{
    const int g = blockIdx.x;
    if(g >= G || threadIdx.x != 0) return;
    xl::Sbac s = states[(long)g * stride]; // (no context-coded bin here: the models stay where they are)
    const int at = pos[g];
    xl::Sink o = {bytes + (long)g * cap + at, (int)(cap - at), 0};
    xl::eco_tile_end(s, &o);
    states[(long)g * stride] = s, pos[g] = at + o.n;
}
__global__ void k_enc_clear_cod(uint32_t *__restrict__ map_scu, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) map_scu[i] &= 0x7FFFFFFFu;
}

static const int16_t k_coef_l[16][8] = {{0, 0, 0, 64, 0, 0, 0, 0}, {0}, {0}, {0}, {0, 1, -5, 52, 20, -5, 1, 0}, {0}, {0}, {0}, {0, 2, -10, 40, 40, -10, 2, 0}, {0}, {0}, {0},
                                        {0, 1, -5, 20, 52, -5, 1, 0}, {0}, {0}, {0}}; // xeve_tbl_mc_l_coeff (xeve_mc.c:39-57)
static const int16_t k_coef_c8[8][4] = {{0, 64, 0, 0}, {-2, 58, 10, -2}, {-4, 52, 20, -4}, {-6, 46, 30, -6}, {-8, 40, 40, -8}, {-6, 30, 46, -6}, {-4, 20, 52, -4}, {-2, 10, 58, -2}};

struct DevBuf {
    void  *p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if(p) (void)hipFree(p); }
    bool need(size_t n)
    {
        if(bytes >= n) return true;
        if(p) (void)hipFree(p), p = nullptr, bytes = 0;
        if(hipMalloc(&p, n) != hipSuccess) { p = nullptr, (void)hipGetLastError(); return false; } // (the caller reports the failure: the runtime's sticky error is cleared)
        bytes = n;
        return true;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};
} // namespace

struct xeve_hip_enc {
    // ---- the engine (enc_host.h) --------------------------------------------------------------------------------------------------------------------------------
    Param P;
    int   G = 0, F = 0, T = 1, nslots = 0, w_scu = 0, h_scu = 0, w_lcu = 0, h_lcu = 0, f_lcu = 0, vh = 0, s_l = 0, s_c = 0;
    long  org_l = 0, org_c = 0, pic_l = 0, pic_c = 0, map_pic = 0, frame_bytes = 0, slice_cap = 0;
    hipStream_t st = nullptr, st2 = nullptr, collect_stream = nullptr; // st2: the second writer pass, beside the next picture's steps
    hipEvent_t  ev_ready = nullptr, ev_done = nullptr;
    bool rows_pending = false, two_stores = false;
    // The stores of later pictures live in the memory of input frames already coded: the frames are kept in CODING order ([position][GOP]), a picture's frame is dead once
    // begin_picture has widened it into `org`, and store s is first written while the picture at position >= s is coded (the frame loop takes the lowest free store) -- so
    // the a-th store placed over the frames' buffer may take [a, a + 1) store sizes of it as soon as that ends inside the first s + 1 frames.  At 3840x2160 two of the five
    // stores of an 8-frame GOP (62 MB of 452 MB per GOP).  A run consumes its frames: push them again before the next begin().
    std::vector<int>  pos_of_frame;           // input frame -> coding position
    std::vector<long> slot_in_frames;         // per store: its index among the stores placed over the frames' buffer, or -1 (a store of slot_planes)
    std::vector<int>  slot_own;               // per store: its index inside slot_planes (stores that are not placed over the frames)
    std::vector<char> slot_clean;             // per store placed over the frames: zeroed in this run (the stores start from zero, as create() leaves slot_planes)
    int  n_own = 0;
    long pushed = 0;                          // frames pushed since the last run began
    bool ran = false;
    int  cur_store = 0; // the store the picture being decided fills
    xeve_hip_sbac *fin = nullptr;
    int fin_stride = 1;
    DevBuf frames, org[3], slot_planes, slot_mv, slot_refi, scu, cum, ipm, tidx, rw_scu, rw_cum, rw_ipm, store2[2], states, rw_states, jobs, out, next_best, cost, ws, slice, pos;
    xeve_hip_sbac *h_states = nullptr; // pinned: the copies behind the second writer pass must not stall the host
    int32_t       *h_pos = nullptr;
    bool           rewrite_mode = false; // the slice data comes from the second writer pass (several chains per picture, or asked for)
    std::vector<uint8_t>       h_bytes;
    int16_t coef_c[32][4];
    PicSetup S;
    xeve_hip_refpic tab[2 * XEVE_HIP_MAX_REFP];
    xl::EcoParams   E;
    std::string     error;
    int64_t n_steps = 0;
    double  t_steps = 0, t_ends = 0;
    std::vector<std::vector<uint8_t>> bitstreams;
    std::unique_ptr<BatchEncoder<xeve_hip_enc>> loop; // the frame loop of the run in progress (xeve_hip_enc_begin .. the advance that returns 0)

    bool fail(const std::string &m)
    {
        if(error.empty()) error = m;
        return false;
    }
    bool hip_ok(hipError_t e, const char *what) { return e == hipSuccess ? true : fail(std::string(what) + ": " + hipGetErrorString(e)); }
    bool rc_ok(int rc, const char *what) { return rc == XEVE_HIP_OK ? true : fail(std::string(what) + ": " + xeve_hip_last_error()); }
    pel *slot_plane(int slot, int c) const // sample (0, 0) of GOP 0's picture in store `slot`
    {
        const size_t store = (size_t)G * (pic_l + 2 * pic_c);
        pel *base = slot_in_frames[slot] >= 0 ? frames.as<pel>() + (size_t)slot_in_frames[slot] * store : slot_planes.as<pel>() + (size_t)slot_own[slot] * store;
        return c == 0 ? base + (size_t)PAD_L * s_l + PAD_L : base + (size_t)G * pic_l + (size_t)(c - 1) * G * pic_c + (size_t)PAD_C * s_c + PAD_C;
    }
    int16_t *slot_map_mv(int slot) const { return slot_mv.as<int16_t>() + (size_t)slot * G * map_pic * 4; }
    int8_t  *slot_map_refi(int slot) const { return slot_refi.as<int8_t>() + (size_t)slot * G * map_pic * 2; }

    // the batch's dimensions (no device call)
    bool dims(const Param &p, int ngops, int nframes)
    {
        P = p, G = ngops, F = nframes;
        w_scu = P.w >> 2, h_scu = P.h >> 2, w_lcu = (P.w + CTU - 1) / CTU, h_lcu = (P.h + CTU - 1) / CTU, f_lcu = w_lcu * h_lcu, T = std::min(P.threads, h_lcu);
        vh = (P.h + 2 * PAD_L + 63) & ~63, s_l = P.w + 2 * PAD_L, s_c = P.w / 2 + 2 * PAD_C;
        org_l = (long)vh * P.w, org_c = (long)(vh / 2) * (P.w / 2), pic_l = (long)vh * s_l, pic_c = (long)(vh / 2) * s_c, map_pic = (long)(vh / 4) * w_scu;
        frame_bytes = P.frame_bytes(), slice_cap = (long)P.w * P.h * 3 / 2 + 4096;
        if((double)G * org_l >= 8589934592.0) return fail("too many GOPs for one batch at this picture size: the stacked originals must stay below 2^33 samples (xh_common.h: halved 32-bit offsets)");
        if((long)G * T > 65535) return fail("too many GOPs for one batch: GOPs x row chains is a grid dimension (at most 65535)");
        if((double)G * vh * 32 >= 2147483648.0) return fail("too many GOPs for one batch at this picture size: the tall picture's rows in 1/16 sample units must fit 31 bits");
        rewrite_mode = T > 1 || (P_reserved0 & 1);
        nslots = BatchEncoder<xeve_hip_enc>::slots_needed(P, F);
        if(nslots < 1) return fail("the frame loop needs more picture stores than there are");
        const std::vector<PicPlan> plan = Planner(P, F).run();
        pos_of_frame.assign(F, -1);
        for(size_t i = 0; i < plan.size(); i++)
            if(plan[i].frame >= 0 && plan[i].frame < F && pos_of_frame[plan[i].frame] < 0) pos_of_frame[plan[i].frame] = (int)i;
        bool in_order = (int)plan.size() == F;
        for(int f = 0; f < F; f++) in_order = in_order && pos_of_frame[f] >= 0;
        static const bool share = !(getenv("XEVE_HIP_ENC_SHARE") && atoi(getenv("XEVE_HIP_ENC_SHARE")) == 0); // developer switch: 0 = every store in slot_planes
        slot_in_frames.assign(nslots, -1), slot_own.assign(nslots, 0), slot_clean.assign(nslots, 0), n_own = 0;
        if(!in_order) // (a run whose plan does not code every frame exactly once keeps the frames where they were pushed)
            for(int f = 0; f < F; f++) pos_of_frame[f] = f;
        const double store = (double)(pic_l + 2 * pic_c) * 2, frame = (double)frame_bytes;
        long a = 0;
        for(int sl = 0; sl < nslots; sl++) {
            if(share && in_order && (a + 1) * store <= (sl + 1) * frame && (a + 1) * store <= (double)F * frame) slot_in_frames[sl] = a++;
            else slot_own[sl] = n_own++;
        }
        return true;
    }
    // every device buffer of the batch with its size (the second CTU store apart: the batch runs without it)
    std::vector<std::pair<DevBuf *, size_t>> buffers()
    {
        const size_t nst = (size_t)G * T, m = (size_t)G * map_pic;
        std::vector<std::pair<DevBuf *, size_t>> v = {
            {&frames, (size_t)G * F * frame_bytes}, {&org[0], (size_t)G * org_l * 2}, {&org[1], (size_t)G * org_c * 2}, {&org[2], (size_t)G * org_c * 2},
            {&slot_planes, std::max<size_t>(1, (size_t)n_own) * G * (pic_l + 2 * pic_c) * 2}, {&slot_mv, (size_t)nslots * m * 8}, {&slot_refi, (size_t)nslots * m * 2},
            {&scu, m * 4}, {&cum, m * 4}, {&ipm, m}, {&tidx, m}, {&states, nst * sizeof(xeve_hip_sbac)}, {&rw_states, (size_t)G * sizeof(xeve_hip_sbac)},
            {&jobs, nst * sizeof(xeve_hip_ctu_job)}, {&out, nst * sizeof(xeve_hip_ctu_data)}, {&next_best, nst * sizeof(xeve_hip_sbac)}, {&cost, nst * 8},
            {&slice, (size_t)G * slice_cap}, {&pos, (size_t)G * 4}};
        if(rewrite_mode) v.insert(v.end(), {{&store2[0], store_bytes()}, {&rw_scu, m * 4}, {&rw_cum, m * 4}, {&rw_ipm, m}});
        return v;
    }
    size_t store_bytes() const { return (size_t)G * f_lcu * sizeof(xl::CtuSyntax); }
    // the walk's workspace for the run's most demanding picture (no device call: the tables of an inter picture are bound to placeholders)
    size_t workspace_bytes(const std::vector<PicSetup> &setups)
    {
        size_t most = 0;
        for(PicSetup s : setups) {
            bind_inter(s);
            if(s.slice_type != ST_I) { // (a size query dereferences none of them; a batch that has no buffers yet has no addresses to give)
                static int16_t dummy[4];
                if(!s.ti.map_mv) s.ti.map_mv = dummy;
                if(!s.ti.map_refi) s.ti.map_refi = reinterpret_cast<int8_t *>(dummy);
                if(!s.ti.col_mv0) s.ti.col_mv0 = dummy;
                if(!s.ti.col_mv1) s.ti.col_mv1 = dummy;
            }
            const size_t need = xeve_hip_mode_analyze_ctu_workspace(G * T, &s.tp, s.slice_type == ST_I ? nullptr : &s.ti, P.w, P.w / 2);
            if(need == 0) { fail(std::string("the CTU walk refuses a picture's parameters: ") + xeve_hip_last_error()); return 0; }
            most = std::max(most, need);
        }
        return most;
    }
    bool create(const Param &p, int ngops, int nframes)
    {
        if(!dims(p, ngops, nframes)) return false;
        memset(coef_c, 0, sizeof(coef_c));
        for(int i = 0; i < 8; i++) memcpy(coef_c[4 * i], k_coef_c8[i], sizeof(k_coef_c8[i])); // xeve_tbl_mc_c_coeff (xeve_mc.c:59-93)
        int pr_low = 0, pr_high = 0;
        (void)hipDeviceGetStreamPriorityRange(&pr_low, &pr_high);
        const char *pe = getenv("XEVE_HIP_ENC_PRIO"); // developer switch: 1 = the walk's stream above the second pass's
        if(!(pe && atoi(pe) == 1)) pr_low = pr_high = 0;
        if(!hip_ok(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr_high), "hipStreamCreate") || !hip_ok(hipStreamCreateWithPriority(&st2, hipStreamNonBlocking, pr_low), "hipStreamCreate") ||
           !hip_ok(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming), "hipEventCreate") || !hip_ok(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming), "hipEventCreate"))
            return false;

        bool ok = true;
        for(auto &b : buffers()) ok = ok && b.first->need(b.second);
        // a second CTU store lets the second writer pass of a picture run beside the next picture's steps (which fill the other store); without the memory for it the
        // next picture waits for the pass
        // (XEVE_HIP_ENC_TWO_STORES=0: developer switch, the one-store form)
        const char *ts = getenv("XEVE_HIP_ENC_TWO_STORES");
        if(ok && rewrite_mode && !(ts && atoi(ts) == 0)) two_stores = store2[1].need(store_bytes());
        if(!two_stores) (void)hipGetLastError(); // (the second store is optional: a failed allocation must not be reported by the next error check)
        if(!ok) return fail("not enough device memory for this batch (hipMalloc failed)");
        // everything starts from zero: the stores' padding and the rows between the stacked pictures are read by nobody before they are written, the maps' rows between
        // the pictures must say "not coded"
        for(DevBuf *b : {&org[0], &org[1], &org[2], &slot_planes, &slot_mv, &scu, &cum, &ipm, &tidx})
            if(!hip_ok(hipMemsetAsync(b->p, 0, b->bytes, st), "hipMemset")) return false;
        if(!hip_ok(hipMemsetAsync(slot_refi.p, 0xFF, slot_refi.bytes, st), "hipMemset")) return false;
        memset(&E, 0, sizeof(E));
        for(int l = 4; l <= LOG2_CTU; l++)
            if(!rc_ok(xh_get_scan(l, l, &E.scan[l]), "scan tables")) return false;
        if(!hip_ok(hipHostMalloc((void **)&h_states, (size_t)G * sizeof(xeve_hip_sbac), hipHostMallocDefault), "hipHostMalloc") ||
           !hip_ok(hipHostMalloc((void **)&h_pos, (size_t)G * 4, hipHostMallocDefault), "hipHostMalloc"))
            return false;
        return hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
    }
    int P_reserved0 = 0; // bit 0: always run the second writer pass (tests)
    long walk_load = 0; // chains announced to the fused walk while a run is in progress (walk.hip: xh_walk_load)
    void announce(long chains)
    {
        xh_walk_load(chains - walk_load);
        walk_load = chains;
    }
    ~xeve_hip_enc()
    {
        announce(0);
        if(st2) (void)hipStreamSynchronize(st2), (void)hipStreamDestroy(st2);
        if(st) (void)hipStreamSynchronize(st), (void)hipStreamDestroy(st);
        if(ev_ready) (void)hipEventDestroy(ev_ready);
        if(ev_done) (void)hipEventDestroy(ev_done);
        if(h_states) (void)hipHostFree(h_states);
        if(h_pos) (void)hipHostFree(h_pos);
    }

    // the device side of an inter picture's set-up: the reference table and the motion maps of the stores the frame loop named
    void bind_inter(PicSetup &s)
    {
        memset(tab, 0, sizeof(tab));
        if(s.slice_type == ST_I) return;
        for(int l = 0; l < 2; l++)
            for(int r = 0; r < s.nref[l]; r++) {
                xeve_hip_refpic &e = tab[r * 2 + l];
                e.y = slot_plane(s.ref[r][l].slot, 0), e.u = slot_plane(s.ref[r][l].slot, 1), e.v = slot_plane(s.ref[r][l].slot, 2), e.poc = s.ref[r][l].poc;
            }
        if(s.slice_type == ST_P) tab[1] = tab[0]; // (P slices never read list 1; the table stays addressable)
        s.ti.refp = tab, s.ti.s_ref_l = s_l, s.ti.s_ref_c = s_c, s.ti.map_mv = slot_map_mv(s.cur_slot), s.ti.map_refi = slot_map_refi(s.cur_slot);
        s.ti.col_mv0 = slot_map_mv(s.ref[0][0].slot), s.ti.col_mv1 = s.slice_type == ST_B ? slot_map_mv(s.ref[0][1].slot) : s.ti.col_mv0;
        s.ti.coef_l = k_coef_l, s.ti.coef_c = coef_c;
    }
    // The walk's workspace at the size of the run's most demanding picture, BEFORE the first step: growing it between two pictures means hipFree + hipMalloc, and hipFree
    // waits for the whole device -- for the second writer pass of the picture before, which is meant to run beside the next picture's steps.
    bool reserve(const std::vector<PicSetup> &setups)
    {
        const size_t most = workspace_bytes(setups);
        if(most == 0) return false;
        return ws.need(most) || fail("not enough device memory for the CTU walk's workspace");
    }
    void begin_picture(const PicSetup &setup)
    {
        S = setup;
        if(!error.empty()) return;
        const long n = (long)P.w * P.h;
        const dim3 lg((unsigned)((n + 255) / 256), G);
        if(P.input_depth > 8)
            k_enc_load<uint16_t, 0><<<lg, 256, 0, st>>>(frames.as<uint8_t>(), frame_bytes, (long)pos_of_frame[S.frame] * G * frame_bytes, org[0].as<pel>(), org[1].as<pel>(),
                                                        org[2].as<pel>(), P.w, P.h, org_l, org_c);
        else
            k_enc_load<uint8_t, BIT_DEPTH - 8><<<lg, 256, 0, st>>>(frames.as<uint8_t>(), frame_bytes, (long)pos_of_frame[S.frame] * G * frame_bytes, org[0].as<pel>(),
                                                                   org[1].as<pel>(), org[2].as<pel>(), P.w, P.h, org_l, org_c);
        if(slot_in_frames[S.cur_slot] >= 0 && !slot_clean[S.cur_slot]) { // (behind the load on the same stream: this picture's own frame may lie inside the store)
            if(pos_of_frame[S.frame] < S.cur_slot) fail("a picture store over the frames' buffer is taken before the frames it covers are coded");
            hip_ok(hipMemsetAsync(slot_plane(S.cur_slot, 0) - ((size_t)PAD_L * s_l + PAD_L), 0, (size_t)G * (pic_l + 2 * pic_c) * 2, st), "hipMemset");
            slot_clean[S.cur_slot] = 1;
        }
        hip_ok(hipMemsetAsync(scu.p, 0, scu.bytes, st), "hipMemset"), hip_ok(hipMemsetAsync(cum.p, 0, cum.bytes, st), "hipMemset"); // xeve_pic_prepare (:1236-1237)
        hip_ok(hipMemsetAsync(slot_map_mv(S.cur_slot), 0, (size_t)G * map_pic * 8, st), "hipMemset"); // (:1220-1225)
        hip_ok(hipMemsetAsync(slot_map_refi(S.cur_slot), 0xFF, (size_t)G * map_pic * 2, st), "hipMemset");
        if(!rewrite_mode) hip_ok(hipMemsetAsync(pos.p, 0, pos.bytes, st), "hipMemset"); // (the first pass's bytes are the slice data; otherwise the second pass owns the buffers)
        bind_inter(S);
        const size_t need = xeve_hip_mode_analyze_ctu_workspace(G * T, &S.tp, S.slice_type == ST_I ? nullptr : &S.ti, P.w, P.w / 2);
        if(need == 0) fail(std::string("the CTU walk refuses the picture's parameters: ") + xeve_hip_last_error());
        else if(!ws.need(need)) fail("not enough device memory for the CTU walk's workspace"); // (reserve() has sized it: no allocation here in a run that began with begin())
        E.idc = 1, E.slice_type = S.slice_type, E.log2_ctu = LOG2_CTU, E.pic_w = P.w, E.pic_h = P.h, E.w_scu = w_scu, E.num_refp[0] = S.ep.num_refp[0], E.num_refp[1] = S.ep.num_refp[1];
    }
    void reset_chain(int t)
    {
        if(error.empty()) k_enc_reset_chain<<<(G + 63) / 64, 64, 0, st>>>(states.as<xeve_hip_sbac>(), T, t, G);
    }
    bool keeps_store() const { return store2[0].p != nullptr; }
    // the writer kernels on whole waves (eco_lane.h eco_levels) unless XEVE_HIP_WRITER_WAVE=0 (developer switch: the lone-lane form, for comparison)
    static bool writer_wave()
    {
        static const bool v = !getenv("XEVE_HIP_WRITER_WAVE") || atoi(getenv("XEVE_HIP_WRITER_WAVE")) != 0;
        return v;
    }
    xl::CtuSyntax *store_now() const { return store2[cur_store].as<xl::CtuSyntax>(); }
    void step(const ChainCtu *c, int n)
    {
        if(!error.empty()) return;
        const auto t0 = std::chrono::steady_clock::now();
        StepDesc D;
        memset(&D, 0, sizeof(D));
        D.n = n;
        for(int i = 0; i < n; i++) D.t[i] = c[i].t, D.x[i] = c[i].x, D.y[i] = c[i].y, D.lcu[i] = c[i].lcu;
        const int nch = n * G;
        if(rows_pending && !two_stores) { // (one store: it is about to be overwritten -- the pass must be through)
            if(!hip_ok(hipStreamWaitEvent(st, ev_done, 0), "event")) return;
            rows_pending = false;
        }
        k_enc_jobs<<<(nch + 255) / 256, 256, 0, st>>>(D, G, T, jobs.as<xeve_hip_ctu_job>());
        const pel *o[3] = {org[0].as<pel>(), org[1].as<pel>(), org[2].as<pel>()};
        pel       *m[3] = {slot_plane(S.cur_slot, 0), slot_plane(S.cur_slot, 1), slot_plane(S.cur_slot, 2)};
        const int64_t pe[5] = {org_l, org_c, pic_l, pic_c, map_pic};
        // (the walk's exit states are never loaded by anything but further counts -- the next CTU starts from the writer's state: count-only states, xh_common.h)
        static const bool full_states = getenv("XEVE_HIP_ENC_FULL_STATES") && atoi(getenv("XEVE_HIP_ENC_FULL_STATES"));
        std::optional<XhCountStatesScope> count_only;
        if(!full_states) count_only.emplace();
        if(!rc_ok(xeve_hip_mode_analyze_ctu_jobs(o, P.w, P.w / 2, m, s_l, s_c, scu.as<uint32_t>(), ipm.as<int8_t>(), tidx.as<uint8_t>(), cum.as<uint32_t>(), pe,
                                                 states.as<xeve_hip_sbac>(), G * T, &S.tp, S.slice_type == ST_I ? nullptr : &S.ti, jobs.as<xeve_hip_ctu_job>(), nch,
                                                 out.as<xeve_hip_ctu_data>(), next_best.as<xeve_hip_sbac>(), cost.as<double>(), ws.p, ws.bytes, st),
                  "xeve_hip_mode_analyze_ctu_jobs"))
            return;
        if(keeps_store()) k_enc_keep<<<dim3(4, nch), 256, 0, st>>>(out.as<xeve_hip_ctu_data>(), store_now(), D, G, f_lcu);
        if(writer_wave())
            k_enc_write<true><<<nch, 64, 0, st>>>(out.as<xeve_hip_ctu_data>(), states.as<xeve_hip_sbac>(), E, scu.as<uint32_t>(), ipm.as<int8_t>(), tidx.as<uint8_t>(), cum.as<uint32_t>(),
                                                  map_pic, jobs.as<xeve_hip_ctu_job>(), nch, slice.as<uint8_t>(), rewrite_mode ? 0 : slice_cap, pos.as<int32_t>());
        else
            k_enc_write<false><<<nch, 64, 0, st>>>(out.as<xeve_hip_ctu_data>(), states.as<xeve_hip_sbac>(), E, scu.as<uint32_t>(), ipm.as<int8_t>(), tidx.as<uint8_t>(), cum.as<uint32_t>(),
                                                   map_pic, jobs.as<xeve_hip_ctu_job>(), nch, slice.as<uint8_t>(), rewrite_mode ? 0 : slice_cap, pos.as<int32_t>());
        hip_ok(hipGetLastError(), "step kernels");
        n_steps++, t_steps += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // The picture's end, ISSUED: loop filter and padding on the main stream (the next picture may start behind them); the slice data on the second stream, from private
    // copies of the unit maps and from the picture's own CTU store (the next picture fills the other one) -- the second writer pass (2040 serial CTUs per 3840x2160
    // picture: seconds) runs beside the next picture's steps, and nothing on the main stream waits for it.  (A wait on an event of the second stream turned out to
    // cover everything queued there: measured, the walk stood still for the whole pass.)
    void end_picture(bool rewrite)
    {
        if(!error.empty()) return;
        const auto t0 = std::chrono::steady_clock::now();
        for(int g = 0; g < G; g++) // xeve_loop_filter
            if(!rc_ok(xeve_hip_deblock(slot_plane(S.cur_slot, 0) + (size_t)g * pic_l, slot_plane(S.cur_slot, 1) + (size_t)g * pic_c, slot_plane(S.cur_slot, 2) + (size_t)g * pic_c, s_l,
                                       s_c, scu.as<uint32_t>() + (size_t)g * map_pic, cum.as<uint32_t>() + (size_t)g * map_pic, tidx.as<uint8_t>() + (size_t)g * map_pic,
                                       slot_map_refi(S.cur_slot) + (size_t)g * map_pic * 2, slot_map_mv(S.cur_slot) + (size_t)g * map_pic * 4, &S.dp, st),
                      "xeve_hip_deblock"))
                return;
        fin = states.as<xeve_hip_sbac>(), fin_stride = T;
        hipStream_t ws_ = st;
        if(rewrite) {
            if(!keeps_store()) { fail("the second writer pass needs the CTU store"); return; }
            const size_t n = (size_t)G * map_pic;
            if(!hip_ok(hipMemcpyAsync(rw_scu.p, scu.p, n * 4, hipMemcpyDeviceToDevice, st), "copy maps") || !hip_ok(hipMemcpyAsync(rw_cum.p, cum.p, n * 4, hipMemcpyDeviceToDevice, st), "copy maps") ||
               !hip_ok(hipMemcpyAsync(rw_ipm.p, ipm.p, n, hipMemcpyDeviceToDevice, st), "copy maps") || !hip_ok(hipEventRecord(ev_ready, st), "event") ||
               !hip_ok(hipStreamWaitEvent(st2, ev_ready, 0), "event"))
                return;
            ws_ = st2;
            k_enc_clear_cod<<<(unsigned)((n + 255) / 256), 256, 0, st2>>>(rw_scu.as<uint32_t>(), (long)n); // MCU_CLR_COD over the picture (:466-468)
            hip_ok(hipMemsetAsync(pos.p, 0, pos.bytes, st2), "hipMemset");
            fin = rw_states.as<xeve_hip_sbac>(), fin_stride = 1;
            k_enc_reset_chain<<<(G + 63) / 64, 64, 0, st2>>>(fin, 1, 0, G);
            xl::EcoParams Ew = E; // (the pass's own copy: E follows the next picture)
            for(int row = 0; row < h_lcu; row++) {
                if(writer_wave())
                    k_enc_rewrite<true><<<G, 64, 0, st2>>>(store_now(), fin, Ew, rw_scu.as<uint32_t>(), rw_ipm.as<int8_t>(), tidx.as<uint8_t>(), rw_cum.as<uint32_t>(), map_pic, G,
                                                           f_lcu, w_lcu, row * w_lcu, (row + 1) * w_lcu, slice.as<uint8_t>(), slice_cap, pos.as<int32_t>());
                else
                    k_enc_rewrite<false><<<G, 64, 0, st2>>>(store_now(), fin, Ew, rw_scu.as<uint32_t>(), rw_ipm.as<int8_t>(), tidx.as<uint8_t>(), rw_cum.as<uint32_t>(), map_pic, G,
                                                            f_lcu, w_lcu, row * w_lcu, (row + 1) * w_lcu, slice.as<uint8_t>(), slice_cap, pos.as<int32_t>());
            }
            rows_pending = true;
            if(two_stores) cur_store ^= 1;
        }
        k_enc_tile_end<<<G, 64, 0, ws_>>>(fin, fin_stride, G, slice.as<uint8_t>(), slice_cap, pos.as<int32_t>());
        for(int g = 0; g < G; g++) // xeve_pic_finish: the picture becomes a reference
            if(!rc_ok(xeve_hip_picbuf_expand(slot_plane(S.cur_slot, 0) + (size_t)g * pic_l, slot_plane(S.cur_slot, 1) + (size_t)g * pic_c, slot_plane(S.cur_slot, 2) + (size_t)g * pic_c,
                                             s_l, s_c, P.w, P.h, P.w / 2, P.h / 2, PAD_L, PAD_C, 1, st),
                      "xeve_hip_picbuf_expand"))
                return;
        if(!hip_ok(hipMemcpy2DAsync(h_states, sizeof(xeve_hip_sbac), fin, sizeof(xeve_hip_sbac) * (size_t)fin_stride, sizeof(xeve_hip_sbac), G, hipMemcpyDeviceToHost, ws_), "copy states") ||
           !hip_ok(hipMemcpyAsync(h_pos, pos.p, (size_t)G * 4, hipMemcpyDeviceToHost, ws_), "copy sizes") || !hip_ok(hipEventRecord(ev_done, ws_), "event"))
            return;
        collect_stream = ws_, have_held = false;
        t_ends += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if(ws_ == st) fetch(held, held_bins), have_held = true; // (one chain per picture: the next picture's writer appends to the same buffers, so they are read out now)
    }
    std::vector<std::vector<uint8_t>> held;
    std::vector<uint32_t> held_bins;
    bool have_held = false;
    void collect(std::vector<std::vector<uint8_t>> &slice_data, std::vector<uint32_t> &bins)
    {
        if(have_held) slice_data.swap(held), bins.swap(held_bins), have_held = false;
        else fetch(slice_data, bins);
    }
    void fetch(std::vector<std::vector<uint8_t>> &slice_data, std::vector<uint32_t> &bins)
    {
        slice_data.clear(), bins.clear();
        if(!error.empty()) return;
        const auto t0 = std::chrono::steady_clock::now();
        if(!hip_ok(hipEventSynchronize(ev_done), "hipEventSynchronize")) return;
        slice_data.resize(G), bins.resize(G);
        for(int g = 0; g < G; g++) {
            if(h_pos[g] < 0 || h_pos[g] > slice_cap) { fail("a picture's slice data outgrew its buffer"); slice_data.clear(), bins.clear(); return; }
            slice_data[g].resize(h_pos[g]);
            if(h_pos[g] && !hip_ok(hipMemcpyAsync(slice_data[g].data(), slice.as<uint8_t>() + (size_t)g * slice_cap, h_pos[g], hipMemcpyDeviceToHost, collect_stream), "copy slice data")) {
                slice_data.clear(), bins.clear();
                return;
            }
            bins[g] = h_states[g].bin_counter;
        }
        if(!hip_ok(hipStreamSynchronize(collect_stream), "hipStreamSynchronize")) slice_data.clear(), bins.clear();
        rows_pending = false;
        t_ends += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
};

// ---- C-ABI ---------------------------------------------------------------------------------------------------------------------------------------------------------
extern "C" xeve_hip_enc *xeve_hip_enc_create(const xeve_hip_enc_config *cfg, int ngops, int frames)
{
    if(!xh_ready()) { xh_set_error("xeve_hip_init() has not been called"); return nullptr; }
    if(!cfg || ngops < 1 || frames < 1) { xh_set_error("xeve_hip_enc_create: invalid argument"); return nullptr; }
    Param P;
    if(!P.finish(*cfg)) { xh_set_error("xeve_hip_enc_create: %s", P.error.c_str()); return nullptr; }
    std::unique_ptr<xeve_hip_enc> e(new xeve_hip_enc);
    e->P_reserved0 = cfg->reserved[0];
    if(!e->create(P, ngops, frames)) { xh_set_error("xeve_hip_enc_create: %s", e->error.c_str()); return nullptr; }
    return e.release();
}
extern "C" void xeve_hip_enc_delete(xeve_hip_enc *e) { delete e; }
extern "C" int xeve_hip_enc_footprint(const xeve_hip_enc_config *cfg, int ngops, int frames, uint64_t *device_bytes, int32_t *max_gops)
{
    XH_REQUIRE(cfg && ngops >= 1 && frames >= 1);
    Param P;
    if(!P.finish(*cfg)) { xh_set_error("xeve_hip_enc_footprint: %s", P.error.c_str()); return XEVE_HIP_ERR_ARG; }
    const long vh = (P.h + 2 * PAD_L + 63) & ~63;
    const int  T_ = std::min(P.threads, (P.h + CTU - 1) / CTU);
    // (the batch's limits: stacked originals below 2^33 samples, GOPs x row chains a grid dimension, the tall picture's rows in 1/16 sample units in 31 bits -- dims())
    const int  most = (int)std::min<double>(std::min<double>(65535 / std::max(1, T_), std::floor((2147483648.0 - 1) / ((double)vh * 32))), std::floor((8589934592.0 - 1) / ((double)vh * P.w)));
    if(max_gops) *max_gops = most;
    if(!device_bytes) return XEVE_HIP_OK;
    xeve_hip_enc e; // (dimensions and sizes only: nothing of it touches the device)
    e.P_reserved0 = cfg->reserved[0];
    if(!e.dims(P, ngops, frames)) { xh_set_error("xeve_hip_enc_footprint: %s", e.error.c_str()); return XEVE_HIP_ERR_ARG; }
    size_t total = 0;
    for(auto &b : e.buffers()) total += (b.second + 255) & ~(size_t)255;
    if(e.rewrite_mode) total += e.store_bytes(); // the second CTU store
    BatchEncoder<xeve_hip_enc> loop(e, P, ngops, frames);
    const std::vector<PicSetup> setups = loop.dry_setups();
    const size_t ws = setups.empty() ? 0 : e.workspace_bytes(setups);
    if(ws == 0) { xh_set_error("xeve_hip_enc_footprint: %s", e.error.empty() ? "the frame loop refuses the run" : e.error.c_str()); return XEVE_HIP_ERR_ARG; }
    *device_bytes = (uint64_t)(total + ws);
    return XEVE_HIP_OK;
}
extern "C" int xeve_hip_enc_push(xeve_hip_enc *e, int gop, int frame, const uint8_t *yuv, int on_device)
{
    XH_ENTER();
    XH_REQUIRE(e && yuv && gop >= 0 && gop < e->G && frame >= 0 && frame < e->F);
    XH_HIP(hipMemcpyAsync(e->frames.as<uint8_t>() + ((size_t)e->pos_of_frame[frame] * e->G + gop) * e->frame_bytes, yuv, e->frame_bytes,
                          on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e->st));
    XH_HIP(hipStreamSynchronize(e->st));
    e->pushed++;
    return XEVE_HIP_OK;
}
extern "C" int xeve_hip_enc_begin(xeve_hip_enc *e)
{
    XH_ENTER();
    XH_REQUIRE(e);
    e->error.clear(), e->n_steps = 0, e->t_steps = e->t_ends = 0;
    bool shares = false;
    for(long v : e->slot_in_frames) shares = shares || v >= 0;
    if(shares && e->ran && e->pushed < (long)e->G * e->F) {
        xh_set_error("xeve_hip_enc_begin: the last run consumed its frames (picture stores reuse the memory of frames already coded): push every frame again before the next run");
        return XEVE_HIP_ERR_ARG;
    }
    e->ran = true, e->pushed = 0;
    std::fill(e->slot_clean.begin(), e->slot_clean.end(), 0);
    e->loop.reset(new BatchEncoder<xeve_hip_enc>(*e, e->P, e->G, e->F));
    e->loop->always_rewrite = (e->P_reserved0 & 1) != 0;
    if(e->loop->begin(e->bitstreams) != 0) { xh_set_error("xeve_hip_enc_begin: %s", e->loop->error.c_str()); return XEVE_HIP_ERR_ARG; }
    const std::vector<PicSetup> setups = e->loop->dry_setups();
    if(setups.empty() || !e->reserve(setups)) { xh_set_error("xeve_hip_enc_begin: %s", e->error.empty() ? "the frame loop refuses the run" : e->error.c_str()); return XEVE_HIP_ERR_ARG; }
    e->announce(xeve_hip_walk_fused(e->G * e->T) ? (long)e->G * e->T : 0); // (only the fused kernel's teams share the workgroup slots)
    return XEVE_HIP_OK;
}
extern "C" int xeve_hip_enc_advance(xeve_hip_enc *e, int64_t max_steps, int64_t *remaining)
{
    XH_ENTER();
    XH_REQUIRE(e && e->loop && max_steps >= 0);
    const long left = e->loop->advance((long)max_steps);
    if(left <= 0) e->announce(0);
    if(!e->error.empty()) { xh_set_error("xeve_hip_enc_advance: %s", e->error.c_str()); return XEVE_HIP_ERR_DEVICE; }
    if(left < 0) { xh_set_error("xeve_hip_enc_advance: %s", e->loop->error.c_str()); return XEVE_HIP_ERR_ARG; }
    if(remaining) *remaining = left;
    return XEVE_HIP_OK;
}
extern "C" int xeve_hip_enc_sync(xeve_hip_enc *e)
{
    XH_ENTER();
    XH_REQUIRE(e);
    XH_HIP(hipStreamSynchronize(e->st));
    XH_HIP(hipStreamSynchronize(e->st2)); // (the second writer pass, the tile end and the read-back of a picture run there: a fence covers both)
    return XEVE_HIP_OK;
}
extern "C" int xeve_hip_enc_flush(xeve_hip_enc *e)
{
    XH_ENTER();
    XH_REQUIRE(e && e->loop);
    const int rc = e->loop->flush();
    if(!e->error.empty()) { xh_set_error("xeve_hip_enc_flush: %s", e->error.c_str()); return XEVE_HIP_ERR_DEVICE; }
    if(rc != 0) { xh_set_error("xeve_hip_enc_flush: %s", e->loop->error.c_str()); return XEVE_HIP_ERR_ARG; }
    return XEVE_HIP_OK;
}
extern "C" int xeve_hip_enc_encode(xeve_hip_enc *e)
{
    int rc = xeve_hip_enc_begin(e);
    int64_t left = 1;
    while(rc == XEVE_HIP_OK && left > 0) rc = xeve_hip_enc_advance(e, 1 << 20, &left);
    return rc == XEVE_HIP_OK ? xeve_hip_enc_sync(e) : rc;
}
extern "C" int xeve_hip_enc_bitstream(xeve_hip_enc *e, int gop, const uint8_t **data, size_t *bytes)
{
    XH_REQUIRE(e && data && bytes && gop >= 0 && gop < (int)e->bitstreams.size());
    *data = e->bitstreams[gop].data(), *bytes = e->bitstreams[gop].size();
    return XEVE_HIP_OK;
}
extern "C" int xeve_hip_enc_stats(xeve_hip_enc *e, int64_t *ctu_steps, double *step_seconds, double *picture_end_seconds)
{
    XH_REQUIRE(e);
    if(ctu_steps) *ctu_steps = e->n_steps;
    if(step_seconds) *step_seconds = e->t_steps;
    if(picture_end_seconds) *picture_end_seconds = e->t_ends;
    return XEVE_HIP_OK;
}
