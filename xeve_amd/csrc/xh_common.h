// xeve_amd/csrc/xh_common.h -- shared by the HIP translation units of libxeve_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/xeve_hip.h"

typedef int16_t pel;

#define XH_WAVE 64

// ---- DEVELOPER SWITCHES (environment, read once per process; results never depend on them: tests/test_dev_switches_gpu.py runs a reference-bitstream case under each) ------
//   XEVE_HIP_WALK = 0 | 1 | auto, XEVE_HIP_WALK_AUTO_MAX = n       which CTU walk (composed / fused / by width: fused up to n chains, 0 since round 6); xeve_hip_walk_select
//   XEVE_HIP_TREE_SIDE = 0 | 1 | 2                                 the composed walk on one stream / with its side stream (default) / a side stream per node size; xeve_hip_walk_side
//   XEVE_HIP_TREE_GRAPH = 1, XEVE_HIP_TREE_LANE = 1                the one-stream walk replayed from a HIP graph; 4x4 / 8x8 intra nodes by the lane-serial kernel (both measured slower)
//   XEVE_HIP_RDO_SPEC = n                                          pinter_residue_rdo's four bit-count rounds as one speculative round for batches of up to n candidates (256)
//   XEVE_HIP_ME_CPL = 0, XEVE_HIP_ME_LDS = 1                       the search kernel's older lane mappings (rows per lane; the LDS-staged window)
//   XEVE_HIP_DCT = valu                                            32x32 / 64x64 transforms on the VALU path instead of the matrix cores
//   XEVE_HIP_WRITER_WAVE = 0, XEVE_HIP_ENC_TWO_STORES = 0          the entropy writer on a lone lane; one CTU store instead of two
//   XEVE_HIP_ENC_PRIO = 1, XEVE_HIP_ENC_FULL_STATES = 1            the encoder's main stream above its second-pass stream; complete coder states through the walk
//   XEVE_HIP_ENC_SHARE = 0                                         every picture store in memory of its own (default: later stores over the frames already coded, encode.hip)
//   XEVE_HIP_HOST_GRAPH = 0                                        the host-memory form of the inter analysis without its per-CU graph replay
//   XEVE_HIP_WALK_C / _NT / _SPREAD / _DEAL / _INTER / _COUNT / _PROF / _DBG   the fused kernel's team shape, wave placement, stage profile and debug counters (walk.hip)
// ---- error plumbing (abi.cpp) -------------------------------------------------------------
void xh_set_error(const char *fmt, ...);
bool xh_ready();

#define XH_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if(e_ != hipSuccess) {                                                                     \
            xh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return XEVE_HIP_ERR_DEVICE;                                                            \
        }                                                                                          \
    } while(0)

#define XH_REQUIRE(cond)                                                         \
    do {                                                                         \
        if(!(cond)) {                                                            \
            xh_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
            return XEVE_HIP_ERR_ARG;                                             \
        }                                                                        \
    } while(0)

#define XH_ENTER()                                                   \
    do {                                                             \
        if(!xh_ready()) {                                            \
            xh_set_error("xeve_hip_init() has not been called");     \
            return XEVE_HIP_ERR_UNINIT;                              \
        }                                                            \
    } while(0)

// ---- kernel-class timers (abi.cpp; include/xeve_hip.h "xeve_hip_prof_*") ----------------------
enum { XH_PROF_SEARCH = 0, XH_PROF_SPEL = 1, XH_PROF_CU_BITS = 2, XH_PROF_MC = 3, XH_PROF_RESID = 4, XH_PROF_RDOQ = 5, XH_PROF_CU_BITS_SLOW = 6, XH_PROF_WALK = 7 };
bool  xh_prof_on(int cls);
void *xh_prof_begin(int cls, hipStream_t st);
void  xh_prof_end(void *tok, hipStream_t st);
// device counters of the class, NULL while the timers are off: XH_PROF_STRIPES words, a wave adds to word (its workgroup index & (XH_PROF_STRIPES - 1)) -- one
// word for everybody serialised 259 000 same-address atomics per search launch (+5 ms per 4K step, measured)
#define XH_PROF_STRIPES 256
unsigned long long *xh_prof_units(int cls);
#define XH_PROF_SLOT(units) ((units) + (blockIdx.x & (XH_PROF_STRIPES - 1)))
// the same for ONE kernel whose launch records the events itself (hipExtLaunchKernelGGL's start / stop events = the dispatch's own begin and end on the device): events
// recorded around a launch on the stream include what the stream waited for between the previous kernel's end and this one's start -- the host's enqueue gap, other
// streams' kernels holding the CUs -- and do not agree with rocprofv3's per-kernel durations; these do.
void *xh_prof_begin_kernel(int cls, hipEvent_t *start, hipEvent_t *stop);
void  xh_prof_end_kernel(void *tok);
struct XhProf {
    void       *tok;
    hipStream_t st;
    XhProf(int cls, hipStream_t s) : tok(xh_prof_on(cls) ? xh_prof_begin(cls, s) : nullptr), st(s) {}
    ~XhProf() { if(tok) xh_prof_end(tok, st); }
};

// ---- resident pictures (resident.cpp) and the library's generation counter (abi.cpp) -----------
bool     xh_resident_on();                              // xeve_hip_picture_begin() has been called: host planes are cached per picture
void    *xh_resident(const void *host, size_t bytes);   // device copy of host[0 .. bytes) for the current picture (uploaded on first sight)
void     xh_resident_free_all();
uint32_t xh_generation();                               // bumped by every xeve_hip_init / _shutdown: per-thread device state re-creates itself

// parameters of the fused residual chain (tq.hip: k_rdo_valu / k_rdo_rows, dct_mfma.hip: k_rdo_mfma)
struct RdoParams {
    int  shift_fwd, shift_inv;        // transform rounding shifts (xeve_util.c:34-35, xeve_itdq.h:38-39)
    int  q_scale, q_shift, q_offset;  // plain quant (xeve_tq.c:704-727)
    long z_scale, z_thr;              // RDOQ zero pre-test (xeve_tq.c:666-699); z_thr < 0 disables it
    long dq_scale;                    // dequant (xeve_itdq.c:442-475)
    int  dq_shift, dq_offset;
    int  ssd_shift, maxv;
    int  stage; // 0: whole chain (plain quant); 1: front half, stops after the DCT and writes the coefficients + SSD(pred);
                // 2: back half, reads quantised levels from `coef` (e.g. left there by xeve_hip_rdoq) and reconstructs
};

// zig-zag scan of a (1 << log2w) x (1 << log2h) block (xeve_tbl_scan), device memory, built once (rdoq.hip)
int xh_get_scan(int log2w, int log2h, const uint16_t **out);
const int *xh_entropy_table(); // entropy_bits[1024] of xeve_init_bits_est (device memory, built by xeve_hip_init; rdoq.hip)
int xh_cu_bits_jobs_round(const int16_t *coef, size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs,
                          const xeve_hip_cu_bits_params *p, void *workspace, size_t workspace_bytes, uint32_t *bits, xeve_hip_sbac *state_out, int full, int reuse,
                          void *stream, int ev_first = 0, int ev_count = -1); // sbac.hip
int xh_cu_bits_chain_round(size_t coef_elems, const xeve_hip_sbac *sbac_in, const xeve_hip_cu_bits_job *jobs, int njobs, const xeve_hip_cu_bits_params *p,
                           void *workspace, size_t workspace_bytes, uint32_t *out, void *stream); // sbac.hip
// One launch over the searches of several reference pictures: job j belongs to plane j / per_plane; the plane supplies the picture, the
// reference index bits and (integer stage) the re-centred range.  n == 0: the single-picture form.
#define XH_MAX_PLANES 16
struct XhSearchPlanes {
    const pel *ref[XH_MAX_PLANES];
    int        refi_bits[XH_MAX_PLANES], range[XH_MAX_PLANES], refi[XH_MAX_PLANES]; // refi: me_raster's step scales with it
    int        n, per_plane;
    const unsigned char *job_plane; // device array or NULL: the plane of job j when the jobs are not laid out plane by plane
    int        vh;                  // the batch's virtual picture height (below); 0: one picture
};

// PICTURES OF A BATCH STACKED VERTICALLY.  The CTU walk of a closed-GOP batch (encode.cpp) decides CTUs of many pictures in one call.  The intra analysis and the tree
// operations carry a picture index (job.pic + pic_elems); the inter analysis, whose job records have no room for one, sees the batch as ONE TALL PICTURE: every plane
// and unit map of picture p lies p * vh luma rows below picture 0's (vh = the virtual picture height, a multiple of 64 that covers the padded picture), a job's y
// is pic * vh + its row in the picture, and `plane + y * stride + x` addresses the right picture with no further help.  Only the few kernels that compare y with the
// picture's bounds (search ranges, vector clipping, neighbour availability, picture coordinates kept in 16 bits) split y again with xh_vh_base.  Reference planes
// are addressed through (x, y) with 64-bit row arithmetic; offsets into the stacked ORIGINALS are unsigned 32-bit (xh_u below), so a batch's originals stay below 2^32
// samples (the caller's duty).  The height travels per host thread: the walk sets it around the inter calls.
int xh_vh();
struct XhVhScope {
    int prev;
    explicit XhVhScope(int vh);
    ~XhVhScope();
};
// COUNT-ONLY CODER STATES.  A bit count starts with xeve_sbac_bit_reset (xeve_mode.c:39-49), which keeps of the state it was loaded with the range and the context
// models (and low bits of the code register that never reach a count).  A caller whose exit states are only ever loaded into further counts -- the encoder's CTU walk:
// the next CTU's counts start from the WRITER's state (xeve_enc.c:138-139), never from the walk's -- asks for exactly that much: within the scope every bit-count
// launch runs the count-only kernel, whose states carry range + models and a reset remainder.  The C-ABI's own callers get the complete state (the default).
int xh_count_states();
void xh_walk_load(long chains); // walk.hip: a batch encoder starts (+) / ends (-) a run of that many lockstep chains
struct XhCountStatesScope {
    int prev;
    XhCountStatesScope();
    ~XhCountStatesScope();
};
__host__ __device__ __forceinline__ int xh_vh_base(int y, int vh) { return vh > 0 ? (y / vh) * vh : 0; }
// Offsets into the ORIGINAL (xeve_hip_job.off1, the fused comparison's pred_off) are 32-bit element counts read as UNSIGNED: 2^32 samples = 448 stacked pictures of
// 3840x2160.  The job lists the library builds ITSELF go further: every offset it produces is EVEN (CUs start on multiples of 4 luma / 2 chroma samples, strides and
// picture distances are multiples of 4), so it travels HALVED -- 2^33 samples, 896 pictures of 3840x2160, all a GPU's HBM holds in ONE batch -- and says so in the record:
// bit 30 of off2 (XH_OFF2_HALF; the dense second operand's offsets stay far below it) resp. bit 8 of an interpolation job's frac (XH_FRAC_HALF).  A caller's own records
// (any offset, odd ones too) carry no mark and mean what include/xeve_hip.h says.  Consumers decode a record once (xh_job) and use the decoded fields.
#define XH_OFF2_HALF 0x40000000
#define XH_FRAC_HALF 0x100
// what only the fused CTU walk (walk.hip) codes: rdo_dbk_switch (the loop filter's share of the distortions, walk_dbk.h) and inter CUs of 4x4 (min_cu_inter 4) --
// presets slow and placebo.  The composed walk's stage kernels cover square inter CUs of 8 .. 64 without the filter estimate.
inline bool xh_walk_only(const xeve_hip_tree_params *p) { return p->rdo_dbk != 0 || (p->ip.slice_type != 2 && p->min_cu < 8); }

struct XhJob {
    size_t off1; // element offset into plane 1 (the original)
    int    off2; // element offset into plane 2
};
__host__ __device__ __forceinline__ XhJob xh_job(const xeve_hip_job &j)
{
    const bool half = ((uint32_t)j.off2 & 0xC0000000u) == (uint32_t)XH_OFF2_HALF; // (a negative off2 of a caller's record has bit 31 set: never taken for the mark)
    XhJob r;
    r.off1 = half ? (size_t)(uint32_t)j.off1 << 1 : (size_t)(uint32_t)j.off1, r.off2 = half ? (int)((uint32_t)j.off2 & 0x3FFFFFFFu) : j.off2;
    return r;
}
__host__ __device__ __forceinline__ size_t xh_u(size_t off) { return off; } // (a decoded record's offset)
__host__ __device__ __forceinline__ size_t xh_u(int off) { return (size_t)(uint32_t)off; }
// a job record of the library's own making for the block at element offset o of plane 1: halved and marked when o is even (always, for the encoder's pictures); an odd
// offset (a C-ABI caller's CU at an odd position or with an odd stride) stays as it is, below 2^32
__host__ __device__ __forceinline__ xeve_hip_job xh_make_job(size_t o, int off2)
{
    xeve_hip_job j;
    // (the halved form keeps bit 30 of off2 for its mark: an off2 outside 0 .. 2^30 - 1 -- a dense offset beyond a gigasample, ADVICE r05 -- stays unhalved)
    if((o & 1) || off2 < 0 || off2 >= XH_OFF2_HALF) j.off1 = (int)(uint32_t)o, j.off2 = off2;
    else j.off1 = (int)(uint32_t)(o >> 1), j.off2 = off2 | XH_OFF2_HALF;
    return j;
}
__host__ __device__ __forceinline__ xeve_hip_job xh_make_job(long y, long stride, long x, int off2) { return xh_make_job((size_t)(y * stride + x), off2); }
__device__ __forceinline__ int xh_plane_of_job(const unsigned char *job_plane, int per_plane, int j) { return job_plane ? job_plane[j] : j / per_plane; }
// the encoder's per-unit maps a CU's merge / MVP candidates are derived from (xeve_hip_inter_candidates' arguments; inter.hip)
struct XhInterCand {
    const uint32_t *map_scu;
    const uint8_t  *map_tidx;
    const int16_t  *map_mv, *col0, *col1;
    int             w_scu, scuw, scuh, isb, vh;
};
int xh_pinter_analyze_cu_jobs_x(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c, const xeve_hip_sbac *states,
                                int nstates, const xeve_hip_inter_params *p, xeve_hip_inter_job *jobs, int njobs, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4],
                                xeve_hip_inter_result *results, int16_t *coef, xeve_hip_pel *rec_y, xeve_hip_pel *rec_u, xeve_hip_pel *rec_v, xeve_hip_pel *pred_y,
                                xeve_hip_sbac *next_best, void *workspace, size_t workspace_bytes, void *stream, const XhInterCand *cand, const void *est_shared); // inter.hip
int xh_pintra_analyze_cu_jobs_x(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c, const uint32_t *map_scu,
                                const int8_t *map_ipm, const uint8_t *map_tidx, const int64_t *pic_elems, const xeve_hip_sbac *states, int nstates,
                                const xeve_hip_intra_params *p, const xeve_hip_intra_job *jobs, int njobs, xeve_hip_intra_result *results, int16_t *coef, xeve_hip_pel *rec,
                                xeve_hip_sbac *best, void *workspace, size_t workspace_bytes, void *stream, const void *est_shared); // intra.hip
int xh_residue_rdo_jobs_x(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, const xeve_hip_refpic *refp, int s_l, int s_c, const xeve_hip_sbac *states, int nstates,
                          const xeve_hip_rdo_params *p, const xeve_hip_rdo_job *jobs, int njobs, const int16_t (*coef_l)[8], const int16_t (*coef_c)[4],
                          xeve_hip_rdo_result *results, int16_t *coef, xeve_hip_sbac *best, void *workspace, size_t workspace_bytes, void *stream,
                          const void *est_shared, int keep_dropped); // rdo.hip
int xh_residual_back(const pel *org, int s_org, const pel *pred, int s_pred, const xeve_hip_job *jobs, int njobs, int log2w, int log2h, int bit_depth, int qp, int dqscale,
                     const int16_t *coef, pel *rec, int s_rec, int32_t *nnz, int64_t *ssd, hipStream_t st); // tq.hip
// what the integer searches of pinter_me_epzs leave per job (me.hip) and the merge of the sub-pel stage's result into it (xeve_pinter.c:828-833, 690-692): with `finish`
// the sub-pel stage's last selection kernel writes the search's final result itself
struct EpzsState {
    uint32_t cost;
    int16_t  mv[2];
    int32_t  tmpstep, searches;
    int32_t  mot_bits; // what the searches leave in pi->mot_bits[lidx] (0 = untouched; xeve_pinter.c:546-548,690-692)
};
struct XhSpelFinish {
    const EpzsState    *state;
    xeve_hip_me_result *out;
};
int xh_me_spel_pattern_jobs_x(const pel *org0, int s_org, const pel *org_bi, const pel *ref0, int s_ref, const xeve_hip_spel_job *jobs, int njobs, int log2w,
                              int log2h, int bit_depth, const int16_t (*coef)[8], const xeve_hip_spel_params *params, const int32_t *extra,
                              xeve_hip_me_result *results, void *workspace, size_t workspace_bytes, void *stream, const XhSearchPlanes *planes = nullptr, const struct XhSpelFinish *finish = nullptr); // mc.hip
int xh_me_epzs_jobs_planes(const pel *org0, int s_org, const pel *org_bi, const pel *ref0, int s_ref, const xeve_hip_epzs_job *jobs, int njobs, int log2w, int log2h,
                           int bit_depth, const int16_t (*coef)[8], const xeve_hip_epzs_params *params, const int32_t *extra_bits, xeve_hip_me_result *results,
                           void *workspace, size_t workspace_bytes, void *stream, const XhSearchPlanes *planes); // me.hip

// per-thread staging of the host-memory forms that run a batched entry point on ONE unit (intra.hip, tree.hip): a stream, a pinned host buffer and a device arena
// that grow on demand and re-create themselves after xeve_hip_shutdown / _init
struct XhHostArena {
    uint32_t    gen = 0;
    hipStream_t st  = nullptr;
    char       *dev = nullptr, *pin = nullptr;
    size_t      dev_bytes = 0, pin_bytes = 0;
    void release()
    {
        if(st) (void)hipStreamDestroy(st);
        if(dev) (void)hipFree(dev);
        if(pin) (void)hipHostFree(pin);
        st = nullptr, dev = pin = nullptr, dev_bytes = pin_bytes = 0;
    }
    int ensure(size_t io_bytes, size_t ws_bytes)
    {
        if(gen != xh_generation()) release(), gen = xh_generation(); // the library was shut down or re-bound since
        if(!st) XH_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        if(pin_bytes < io_bytes) {
            if(pin) (void)hipHostFree(pin);
            pin = nullptr, pin_bytes = 0;
            XH_HIP(hipHostMalloc((void **)&pin, io_bytes + (io_bytes >> 1), hipHostMallocDefault));
            pin_bytes = io_bytes + (io_bytes >> 1);
        }
        const size_t need = io_bytes + 256 + ws_bytes;
        if(dev_bytes < need) {
            if(dev) {
                XH_HIP(hipStreamSynchronize(st));
                (void)hipFree(dev);
                dev = nullptr, dev_bytes = 0;
            }
            XH_HIP(hipMalloc((void **)&dev, need + (need >> 2)));
            dev_bytes = need + (need >> 2);
        }
        return XEVE_HIP_OK;
    }
    ~XhHostArena() { release(); }
};

static inline int xh_ilog2(int v) { int l = 0; while((1 << l) < v) l++; return l; }
static inline bool xh_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

#ifdef __HIPCC__
// ---- device helpers -------------------------------------------------------------------------
// 16-byte vector of 8 pels that may sit at any 2-byte-aligned address (block rows inside a plane
// start at arbitrary x).  gfx950 under amdhsa runs in unaligned-access mode, so this is ONE
// global_load_dwordx4.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a2 __attribute__((aligned(2)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_a2 __attribute__((aligned(2)));

__device__ __forceinline__ u32x4 xh_ld8(const pel *p) { return *reinterpret_cast<const u32x4_a2 *>(p); }
__device__ __forceinline__ u32x2 xh_ld4(const pel *p) { return *reinterpret_cast<const u32x2_a2 *>(p); }
__device__ __forceinline__ void  xh_st8(pel *p, u32x4 v) { *reinterpret_cast<u32x4_a2 *>(p) = v; }
__device__ __forceinline__ void  xh_st4(pel *p, u32x2 v) { *reinterpret_cast<u32x2_a2 *>(p) = v; }

__device__ __forceinline__ int xh_lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int xh_hi16(uint32_t v) { return ((int)v) >> 16; }
__device__ __forceinline__ uint32_t xh_pack16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }

// XCD-aware workgroup index: the dispatcher places workgroup b on XCD b % 8 (observed, speed only).  Job lists are
// in picture raster order, so handing XCD k the k-th CONTIGUOUS eighth of the grid keeps each XCD's private 4 MB L2
// on one horizontal band of the planes instead of the whole picture.  Bijective for any grid size.
__device__ __forceinline__ unsigned xh_xcd_block(unsigned bid, unsigned nblk)
{
    const unsigned per = nblk >> 3, main = per << 3;
    return bid < main ? (bid & 7u) * per + (bid >> 3) : bid;
}

// one component of get_mv_bits: xeve_tbl_mv_bits in closed form (incl. its -2047 entry) / exp-Golomb beyond +-2048
__device__ __forceinline__ int xh_mvd_bits(int mvd)
{
    const unsigned a = (unsigned)(mvd < 0 ? -mvd : mvd);
    if(mvd > 2048 || mvd <= -2048) {
        unsigned nn = (a + 1) >> 12;
        int len_i = 11;
        while(len_i < 16 && nn != 0) nn >>= 1, len_i++;
        return (len_i << 1) + 2;
    }
    if(mvd == 0) return 1;
    if(mvd == -2047) return 22;
    return 2 * (31 - __clz((int)(a + 1))) + 2;
}


// DPP lane exchanges (wave64): the cross-lane step of every per-block reduction.
#define XH_DPP_QUAD_XOR1 0xB1        // quad_perm:[1,0,3,2]
#define XH_DPP_QUAD_XOR2 0x4E        // quad_perm:[2,3,0,1]
#define XH_DPP_ROW_HALF_MIRROR 0x141 // lane i <-> 7-i inside each 8-lane group
#define XH_DPP_ROW_MIRROR 0x140      // lane i <-> 15-i inside each 16-lane row
template <int CTRL> __device__ __forceinline__ int xh_dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// Sum over aligned groups of N lanes; every lane of the group receives the total.
template <int N> __device__ __forceinline__ int xh_group_sum(int v)
{
    static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16 || N == 32 || N == 64, "group size");
    if(N >= 2) v += xh_dpp<XH_DPP_QUAD_XOR1>(v);
    if(N >= 4) v += xh_dpp<XH_DPP_QUAD_XOR2>(v);
    if(N >= 8) v += xh_dpp<XH_DPP_ROW_HALF_MIRROR>(v);
    if(N >= 16) v += xh_dpp<XH_DPP_ROW_MIRROR>(v);
    if(N >= 32) v += __shfl_xor(v, 16, 64);
    if(N >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
#endif
