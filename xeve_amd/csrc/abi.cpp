// xeve_amd/csrc/abi.cpp -- lifecycle, error state and the DROP-IN DISPATCH TABLES of libxeve_hip.so.
//
// The table functions have exactly the reference's signatures (include/xeve_hip.h) and therefore
// receive borrowed HOST pointers, one small block per call.  Each call stages its operands in a
// per-thread pinned, device-mapped buffer, launches the same batched kernel the device API uses
// with njobs = 1, waits, and copies the result back.  This is the literal drop-in (and what the
// parity tests drive); it pays a launch + sync per call, so production callers are expected to
// move to the batched API (DESIGN.md "granularity").  There is deliberately NO CPU fallback: a
// HIP failure here aborts the process with a message, because the reference's table signatures have
// no error channel and silently answering from the CPU would void the parity claim.
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <vector>

#include "xh_common.h"

int xh_tq_init();
int xh_tx1d(bool fwd, const void *src, void *dst, int log2n, int shift, int line, int step, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// state
// ------------------------------------------------------------------------------------------------
static std::atomic<int>      g_device{-1};
static std::atomic<uint64_t> g_table_calls{0};
static std::atomic<uint32_t> g_generation{0}; // bumped by every init / shutdown: per-thread staging re-creates itself when it changes
static std::mutex            g_init_mu, g_err_mu;
static thread_local char     t_err[512];
static char                  g_err[512];

void xh_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_err, sizeof(t_err), fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(g_err_mu);
    memcpy(g_err, t_err, sizeof(g_err));
}
bool xh_ready() { return g_device.load() >= 0; }
uint32_t xh_generation() { return g_generation.load(); }

// the calling thread's last message; a thread that has none gets a copy of the process-wide last one (taken under the lock)
extern "C" const char *xeve_hip_last_error(void)
{
    if(!t_err[0]) {
        std::lock_guard<std::mutex> lk(g_err_mu);
        memcpy(t_err, g_err, sizeof(t_err));
    }
    return t_err;
}
extern "C" uint64_t    xeve_hip_table_calls(void) { return g_table_calls.load(); }
extern "C" int xeve_hip_sizeof(int i)
{
    static const int sz[] = {(int)sizeof(xeve_hip_job), (int)sizeof(xeve_hip_mc_job), (int)sizeof(xeve_hip_me_params), (int)sizeof(xeve_hip_me_job),
                             (int)sizeof(xeve_hip_me_result), (int)sizeof(xeve_hip_spel_params), (int)sizeof(xeve_hip_spel_job), (int)sizeof(xeve_hip_epzs_job),
                             (int)sizeof(xeve_hip_epzs_params), (int)sizeof(xeve_hip_sbac), (int)sizeof(xeve_hip_cu_bits_params), (int)sizeof(xeve_hip_cu_bits_job),
                             (int)sizeof(xeve_hip_rdoq_est_full), (int)sizeof(xeve_hip_deblock_params), (int)sizeof(xeve_hip_refpic), (int)sizeof(xeve_hip_cu_mc_job),
                             (int)sizeof(xeve_hip_rdo_params), (int)sizeof(xeve_hip_rdo_job), (int)sizeof(xeve_hip_rdo_result), (int)sizeof(xeve_hip_skip_job),
                             (int)sizeof(xeve_hip_skip_result), (int)sizeof(xeve_hip_inter_params), (int)sizeof(xeve_hip_inter_job), (int)sizeof(xeve_hip_inter_result)};
    return i >= 0 && i < (int)(sizeof(sz) / sizeof(sz[0])) ? sz[i] : -1;
}

int  xh_rdoq_tables_init(); // rdoq.hip: zig-zag scans + entropy table, built once per device binding
void xh_rdoq_tables_free();
static void prof_reset_locked();

extern "C" int xeve_hip_init(int device_ordinal)
{
    std::lock_guard<std::mutex> lk(g_init_mu); // the whole initialisation is one critical section (concurrent first calls)
    if(g_device.load() == device_ordinal && device_ordinal >= 0) return XEVE_HIP_OK;
    int n = 0;
    XH_HIP(hipGetDeviceCount(&n));
    if(device_ordinal < 0 || device_ordinal >= n) {
        xh_set_error("xeve_hip_init: device %d not present (%d HIP devices visible)", device_ordinal, n);
        return XEVE_HIP_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    XH_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    if(strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        xh_set_error("xeve_hip_init: device %d is %s; this library is built for gfx950 (MI355X) only", device_ordinal, prop.gcnArchName);
        return XEVE_HIP_ERR_DEVICE;
    }
    if(g_device.load() >= 0) { // re-binding to another device: everything built for the old one goes first
        (void)hipSetDevice(g_device.load());
        (void)hipDeviceSynchronize();
        xh_rdoq_tables_free();
        xh_resident_free_all();
        g_device.store(-1);
    }
    XH_HIP(hipSetDevice(device_ordinal));
    int rc = xh_tq_init();
    if(rc != XEVE_HIP_OK) return rc;
    g_device.store(device_ordinal); // (the table builders below go through entry-point checks that want a bound device)
    rc = xh_rdoq_tables_init();
    if(rc != XEVE_HIP_OK) {
        g_device.store(-1);
        return rc;
    }
    g_generation++;
    return XEVE_HIP_OK;
}

extern "C" void xeve_hip_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_init_mu);
    if(g_device.load() >= 0) {
        (void)hipSetDevice(g_device.load());
        (void)hipDeviceSynchronize();
        prof_reset_locked();
        xh_rdoq_tables_free();
        xh_resident_free_all();
    }
    g_device.store(-1);
    g_generation++;
}

// ------------------------------------------------------------------------------------------------
// kernel-class timers (include/xeve_hip.h: xeve_hip_prof_enable / _read)
// ------------------------------------------------------------------------------------------------
namespace {
struct ProfTok {
    int        cls;
    hipEvent_t e0, e1;
};
std::mutex               g_prof_mu;
std::atomic<unsigned>    g_prof_mask{0}; // bit per class
std::vector<ProfTok *>   g_prof_done;
unsigned long long      *g_prof_units = nullptr; // device, XEVE_HIP_PROF_CLASSES counters
} // namespace
bool xh_prof_on(int cls) { return (g_prof_mask.load(std::memory_order_relaxed) >> cls) & 1u; }
unsigned long long *xh_prof_units(int cls) { return xh_prof_on(cls) && g_prof_units ? g_prof_units + cls : nullptr; }
void *xh_prof_begin(int cls, hipStream_t st)
{
    ProfTok *t = new ProfTok{cls, nullptr, nullptr};
    if(hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess || hipEventRecord(t->e0, st) != hipSuccess) {
        if(t->e0) (void)hipEventDestroy(t->e0);
        if(t->e1) (void)hipEventDestroy(t->e1);
        delete t;
        return nullptr;
    }
    return t;
}
void xh_prof_end(void *tok, hipStream_t st)
{
    ProfTok *t = static_cast<ProfTok *>(tok);
    (void)hipEventRecord(t->e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_done.push_back(t);
}
static void prof_reset_locked()
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for(ProfTok *t : g_prof_done) {
        (void)hipEventDestroy(t->e0), (void)hipEventDestroy(t->e1);
        delete t;
    }
    g_prof_done.clear();
    g_prof_mask.store(0);
    if(g_prof_units) (void)hipFree(g_prof_units), g_prof_units = nullptr;
}
extern "C" int xeve_hip_prof_enable(int class_mask)
{
    XH_ENTER();
    if(class_mask && !g_prof_units) {
        XH_HIP(hipMalloc((void **)&g_prof_units, sizeof(unsigned long long) * XEVE_HIP_PROF_CLASSES));
        XH_HIP(hipMemset(g_prof_units, 0, sizeof(unsigned long long) * XEVE_HIP_PROF_CLASSES));
    }
    g_prof_mask.store((unsigned)class_mask & ((1u << XEVE_HIP_PROF_CLASSES) - 1));
    return XEVE_HIP_OK;
}
extern "C" int xeve_hip_prof_read(double *ms, uint64_t *launches, uint64_t *units, int n)
{
    XH_ENTER();
    XH_REQUIRE(n >= 0 && n <= XEVE_HIP_PROF_CLASSES);
    XH_HIP(hipDeviceSynchronize());
    double   t[XEVE_HIP_PROF_CLASSES] = {};
    uint64_t c[XEVE_HIP_PROF_CLASSES] = {}, u[XEVE_HIP_PROF_CLASSES] = {};
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        for(ProfTok *k : g_prof_done) {
            float f = 0.f;
            if(hipEventElapsedTime(&f, k->e0, k->e1) == hipSuccess && k->cls >= 0 && k->cls < XEVE_HIP_PROF_CLASSES) t[k->cls] += f, c[k->cls]++;
            (void)hipEventDestroy(k->e0), (void)hipEventDestroy(k->e1);
            delete k;
        }
        g_prof_done.clear();
    }
    if(g_prof_units) {
        XH_HIP(hipMemcpy(u, g_prof_units, sizeof(u), hipMemcpyDeviceToHost));
        XH_HIP(hipMemset(g_prof_units, 0, sizeof(u)));
    }
    for(int i = 0; i < n; i++) {
        if(ms) ms[i] = t[i];
        if(launches) launches[i] = c[i];
        if(units) units[i] = u[i];
    }
    return XEVE_HIP_OK;
}

// ------------------------------------------------------------------------------------------------
// per-thread staging for the table layer (the reference calls the tables from up to 8 pool threads,
// lock-free: src_base/xeve_enc.c:336-365)
// ------------------------------------------------------------------------------------------------
[[noreturn]] static void die(const char *what)
{
    fprintf(stderr, "libxeve_hip: FATAL in dispatch-table call (%s): %s\n", what, xeve_hip_last_error());
    fflush(stderr);
    abort();
}
#define TBL_HIP(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if(e_ != hipSuccess) {                                                 \
            xh_set_error("%s failed: %s", #expr, hipGetErrorString(e_));       \
            die(__func__);                                                     \
        }                                                                      \
    } while(0)
#define TBL_RC(expr)                  \
    do {                              \
        if((expr) != XEVE_HIP_OK) die(__func__); \
    } while(0)

namespace {
constexpr size_t REG_A = 0, REG_B = 64 << 10, REG_OUT = 128 << 10, REG_MISC = 192 << 10, REG_TOTAL = 196 << 10;

struct Stage {
    hipStream_t st   = nullptr;
    char       *host = nullptr; // pinned, device-mapped
    char       *dev  = nullptr;
    uint32_t    gen  = 0;       // g_generation this staging was created under
    void create()
    {
        if(!xh_ready()) {
            xh_set_error("dispatch table used before xeve_hip_init()");
            die("Stage");
        }
        TBL_HIP(hipSetDevice(g_device.load()));
        TBL_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        TBL_HIP(hipHostMalloc((void **)&host, REG_TOTAL, hipHostMallocMapped));
        TBL_HIP(hipHostGetDevicePointer((void **)&dev, host, 0));
        memset(host, 0, REG_TOTAL);
        gen = g_generation.load();
    }
    void destroy()
    { // (errors ignored: after a shutdown the old context may be gone already)
        if(st) (void)hipStreamDestroy(st);
        if(host) (void)hipHostFree(host);
        st = nullptr, host = dev = nullptr;
    }
    Stage() { create(); }
    ~Stage() { destroy(); }
    template <typename T> T *h(size_t off) { return reinterpret_cast<T *>(host + off); }
    template <typename T> T *d(size_t off) { return reinterpret_cast<T *>(dev + off); }
    void sync() { TBL_HIP(hipStreamSynchronize(st)); }
};
Stage &stage()
{
    static thread_local Stage s;
    if(s.gen != g_generation.load()) s.destroy(), s.create(); // the library was shut down or re-bound since: fresh stream + buffer
    return s;
}

// copy a w x h block with row stride `s` (elements) into a dense row-major buffer with stride `ds`
inline void gather(int16_t *dst, int ds, const int16_t *src, int s, int w, int h)
{
    for(int y = 0; y < h; y++) memcpy(dst + (size_t)y * ds, src + (size_t)y * s, sizeof(int16_t) * w);
}

struct PairCall {
    Stage         &S;
    xeve_hip_job  *job;
    int32_t       *cand;
    PairCall(int w, int h, void *src1, void *src2, int s1, int s2) : S(stage())
    {
        g_table_calls++;
        gather(S.h<int16_t>(REG_A), w, (const int16_t *)src1, s1, w, h);
        gather(S.h<int16_t>(REG_B), w, (const int16_t *)src2, s2, w, h);
        job       = S.h<xeve_hip_job>(REG_MISC);
        cand      = S.h<int32_t>(REG_MISC + 64);
        job->off1 = job->off2 = 0;
        cand[0]   = 0;
    }
};
} // namespace

// ---- SAD / SSD / SATD / DIFF -------------------------------------------------------------------
static int tbl_sad(int w, int h, void *src1, void *src2, int s_src1, int s_src2, int bit_depth)
{
    PairCall c(w, h, src1, src2, s_src1, s_src2);
    Stage   &S = c.S;
    // operands may be org_bi (negative samples, xeve_pinter.c:143-156): always take the signed kernel
    TBL_RC(xeve_hip_sad_jobs(S.d<pel>(REG_A), w, S.d<pel>(REG_B), w, S.d<xeve_hip_job>(REG_MISC), 1, S.d<int32_t>(REG_MISC + 64), 1,
                             w, h, bit_depth, XEVE_HIP_SRC1_SIGNED, S.d<int32_t>(REG_OUT), S.st));
    S.sync();
    return *S.h<int32_t>(REG_OUT);
}
static int64_t tbl_ssd(int w, int h, void *src1, void *src2, int s_src1, int s_src2, int bit_depth)
{
    PairCall c(w, h, src1, src2, s_src1, s_src2);
    Stage   &S = c.S;
    TBL_RC(xeve_hip_ssd_jobs(S.d<pel>(REG_A), w, S.d<pel>(REG_B), w, S.d<xeve_hip_job>(REG_MISC), 1, S.d<int32_t>(REG_MISC + 64), 1,
                             w, h, bit_depth, S.d<int64_t>(REG_OUT), S.st));
    S.sync();
    return *S.h<int64_t>(REG_OUT);
}
static int tbl_satd(int w, int h, void *src1, void *src2, int s_src1, int s_src2, int bit_depth)
{
    PairCall c(w, h, src1, src2, s_src1, s_src2);
    Stage   &S = c.S;
    TBL_RC(xeve_hip_satd_jobs(S.d<pel>(REG_A), w, S.d<pel>(REG_B), w, S.d<xeve_hip_job>(REG_MISC), 1, S.d<int32_t>(REG_MISC + 64), 1,
                              w, h, bit_depth, S.d<int32_t>(REG_OUT), S.st));
    S.sync();
    return *S.h<int32_t>(REG_OUT);
}
static void tbl_diff(int w, int h, void *src1, void *src2, int s_src1, int s_src2, int s_diff, int16_t *diff, int bit_depth)
{
    (void)bit_depth;
    PairCall c(w, h, src1, src2, s_src1, s_src2);
    Stage   &S = c.S;
    TBL_RC(xeve_hip_diff_jobs(S.d<pel>(REG_A), w, S.d<pel>(REG_B), w, S.d<xeve_hip_job>(REG_MISC), 1, w, h, S.d<int16_t>(REG_OUT), S.st));
    S.sync();
    gather(diff, s_diff, S.h<int16_t>(REG_OUT), w, w, h);
}

#define ROW8(f) {f, f, f, f, f, f, f, f}
extern "C" {
const XEVE_HIP_FN_SAD  xeve_tbl_sad_16b_hip[8][8]  = {ROW8(tbl_sad), ROW8(tbl_sad), ROW8(tbl_sad), ROW8(tbl_sad), ROW8(tbl_sad), ROW8(tbl_sad), ROW8(tbl_sad), ROW8(tbl_sad)};
const XEVE_HIP_FN_SSD  xeve_tbl_ssd_16b_hip[8][8]  = {ROW8(tbl_ssd), ROW8(tbl_ssd), ROW8(tbl_ssd), ROW8(tbl_ssd), ROW8(tbl_ssd), ROW8(tbl_ssd), ROW8(tbl_ssd), ROW8(tbl_ssd)};
const XEVE_HIP_FN_DIFF xeve_tbl_diff_16b_hip[8][8] = {ROW8(tbl_diff), ROW8(tbl_diff), ROW8(tbl_diff), ROW8(tbl_diff), ROW8(tbl_diff), ROW8(tbl_diff), ROW8(tbl_diff), ROW8(tbl_diff)};
const XEVE_HIP_FN_SATD xeve_tbl_satd_16b_hip[1]    = {tbl_satd};
}

// ---- MC ---------------------------------------------------------------------------------------------
// Stages exactly the footprint the reference variant reads (never more host memory than the
// reference touches) into a zero-padded tile whose origin is (ix - BACK, iy - BACK).
template <int TAPS, bool HX, bool VY>
static void tbl_mc(pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, pel *pred, int w, int h, int bit_depth, const int16_t *coef)
{
    constexpr int FS = TAPS == 8 ? 4 : 5, FM = (1 << FS) - 1, BACK = TAPS / 2 - 1;
    Stage &S = stage();
    g_table_calls++;
    const int ix = gmv_x >> FS, iy = gmv_y >> FS;
    const int sw = w + 16, sh = h + TAPS - 1;
    if((size_t)sw * sh * 2 > (64 << 10) || (size_t)w * h * 2 > (64 << 10)) {
        xh_set_error("mc table call %dx%d exceeds the staging tile", w, h);
        die(__func__);
    }
    int16_t *tile = S.h<int16_t>(REG_A);
    memset(tile, 0, sizeof(int16_t) * (size_t)sw * sh);
    const int x_lo = HX ? 0 : BACK, cw = HX ? w + TAPS - 1 : w;
    const int y_lo = VY ? 0 : BACK, ch = VY ? h + TAPS - 1 : h;
    gather(tile + y_lo * sw + x_lo, sw, ref + (long)(iy - BACK + y_lo) * s_ref + (ix - BACK + x_lo), s_ref, cw, ch);
    xeve_hip_mc_job *job = S.h<xeve_hip_mc_job>(REG_MISC);
    job->gmv_x    = (BACK << FS) | (gmv_x & FM);
    job->gmv_y    = (BACK << FS) | (gmv_y & FM);
    job->pred_off = 0;
    job->frac     = (HX ? 1 : 0) | (VY ? 2 : 0);
    if(TAPS == 8) TBL_RC(xeve_hip_mc_l_jobs(S.d<pel>(REG_A), sw, S.d<pel>(REG_OUT), w, S.d<xeve_hip_mc_job>(REG_MISC), 1, w, h, bit_depth, (const int16_t(*)[8])coef, S.st));
    else TBL_RC(xeve_hip_mc_c_jobs(S.d<pel>(REG_A), sw, S.d<pel>(REG_OUT), w, S.d<xeve_hip_mc_job>(REG_MISC), 1, w, h, bit_depth, (const int16_t(*)[4])coef, S.st));
    S.sync();
    gather(pred, s_pred, S.h<int16_t>(REG_OUT), w, w, h);
}
template <bool HX, bool VY>
static void tbl_mc_l(pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, pel *pred, int w, int h, int bit_depth, const int16_t (*c)[8])
{
    tbl_mc<8, HX, VY>(ref, gmv_x, gmv_y, s_ref, s_pred, pred, w, h, bit_depth, &c[0][0]);
}
template <bool HX, bool VY>
static void tbl_mc_c(pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, pel *pred, int w, int h, int bit_depth, const int16_t (*c)[4])
{
    tbl_mc<4, HX, VY>(ref, gmv_x, gmv_y, s_ref, s_pred, pred, w, h, bit_depth, &c[0][0]);
}
extern "C" {
// index [dx != 0][dy != 0]  (xeve_mc.c:383-399)
const XEVE_HIP_MC_L xeve_tbl_mc_l_hip[2][2] = {{tbl_mc_l<false, false>, tbl_mc_l<false, true>}, {tbl_mc_l<true, false>, tbl_mc_l<true, true>}};
const XEVE_HIP_MC_C xeve_tbl_mc_c_hip[2][2] = {{tbl_mc_c<false, false>, tbl_mc_c<false, true>}, {tbl_mc_c<true, false>, tbl_mc_c<true, true>}};

void xeve_average_16b_no_clip_hip(int16_t *src, int16_t *ref, int16_t *dst, int s_src, int s_ref, int s_dst, int wd, int ht)
{
    Stage &S = stage();
    g_table_calls++;
    gather(S.h<int16_t>(REG_A), wd, src, s_src, wd, ht);
    gather(S.h<int16_t>(REG_B), wd, ref, s_ref, wd, ht);
    TBL_RC(xeve_hip_avg(S.d<int16_t>(REG_A), S.d<int16_t>(REG_B), S.d<int16_t>(REG_OUT), (int64_t)wd * ht, S.st));
    S.sync();
    gather(dst, s_dst, S.h<int16_t>(REG_OUT), wd, wd, ht);
}

void xeve_recon_blk_hip(int16_t *coef, pel *pred, int is_coef, int cuw, int cuh, int s_rec, pel *rec, int bit_depth)
{
    Stage &S = stage();
    g_table_calls++;
    const size_t n = (size_t)cuw * cuh;
    if(is_coef) memcpy(S.h<int16_t>(REG_A), coef, 2 * n); // the reference does not read coef when is_coef == 0
    memcpy(S.h<int16_t>(REG_B), pred, 2 * n);
    *S.h<int32_t>(REG_MISC)      = 0;
    *S.h<uint8_t>(REG_MISC + 64) = (uint8_t)(is_coef != 0);
    TBL_RC(xeve_hip_recon(S.d<int16_t>(REG_A), S.d<pel>(REG_B), S.d<uint8_t>(REG_MISC + 64), 1, cuw, cuh, S.d<int32_t>(REG_MISC), cuw,
                          S.d<pel>(REG_OUT), bit_depth, S.st));
    S.sync();
    gather(rec, s_rec, S.h<int16_t>(REG_OUT), cuw, cuw, cuh);
}
}

// ---- 1-D transforms ---------------------------------------------------------------------------------
template <bool FWD, int LOG2N> static void tbl_tx(void *src, void *dst, int shift, int line, int step)
{
    Stage &S = stage();
    g_table_calls++;
    const size_t n = (size_t)(1 << LOG2N) * line, in_b = n * (step == 0 ? 2 : 4), out_b = n * (step == 0 ? 4 : 2);
    if(in_b > (64 << 10) || out_b > (64 << 10)) {
        xh_set_error("transform table call N=%d line=%d exceeds the staging tile", 1 << LOG2N, line);
        die(__func__);
    }
    memcpy(S.h<char>(REG_A), src, in_b);
    TBL_RC(xh_tx1d(FWD, S.d<char>(REG_A), S.d<char>(REG_OUT), LOG2N, shift, line, step, S.st));
    S.sync();
    memcpy(dst, S.h<char>(REG_OUT), out_b);
}
extern "C" {
const XEVE_HIP_TXB  xeve_tbl_txb_hip[6]  = {tbl_tx<true, 1>, tbl_tx<true, 2>, tbl_tx<true, 3>, tbl_tx<true, 4>, tbl_tx<true, 5>, tbl_tx<true, 6>};
const XEVE_HIP_ITXB xeve_tbl_itxb_hip[6] = {tbl_tx<false, 1>, tbl_tx<false, 2>, tbl_tx<false, 3>, tbl_tx<false, 4>, tbl_tx<false, 5>, tbl_tx<false, 6>};
}

// ---- zero-edit installation into a loaded reference library ---------------------------------------------
extern "C" int xeve_hip_install_tables(void *fn_itxb_slot)
{
    XH_ENTER();
    struct {
        const char *name;
        const void *value;
    } pats[] = {
        {"xeve_func_sad", xeve_tbl_sad_16b_hip},   {"xeve_func_ssd", xeve_tbl_ssd_16b_hip},
        {"xeve_func_diff", xeve_tbl_diff_16b_hip}, {"xeve_func_satd", xeve_tbl_satd_16b_hip},
        {"xeve_func_mc_l", xeve_tbl_mc_l_hip},     {"xeve_func_mc_c", xeve_tbl_mc_c_hip},
        {"xeve_func_average_no_clip", (const void *)xeve_average_16b_no_clip_hip},
        {"xeve_func_txb", &xeve_tbl_txb_hip},
    };
    int n = 0;
    for(auto &p : pats) {
        void **slot = (void **)dlsym(RTLD_DEFAULT, p.name);
        if(!slot) {
            xh_set_error("xeve_hip_install_tables: symbol %s not found (is the reference library loaded RTLD_GLOBAL?)", p.name);
            return XEVE_HIP_ERR_ARG;
        }
        *slot = const_cast<void *>(p.value);
        n++;
    }
    if(fn_itxb_slot) {
        *(const void **)fn_itxb_slot = &xeve_tbl_itxb_hip;
        n++;
    }
    return n;
}
