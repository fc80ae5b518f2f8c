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
int xh_main_tools_init(); // main_tools.hip: the ATS matrices
int xh_itrans_ats(int type, int log2n, const int16_t *coef, int16_t *block, int shift, int line, int skip_line, int skip_line_2, hipStream_t st);
int xh_trans_ats(int type, int log2n, const int16_t *block, int16_t *coef, int shift, int line, int skip_line, int skip_line_2, hipStream_t st);
int xh_sobel(int vertical, const pel *pred, int s_pred, int32_t *der, int s_der, int w, int h, hipStream_t st);
int xh_ipred_ang(int group, int right, const pel *lines, pel *dst, int w, int h, int ipm, int bit_depth, hipStream_t st);
int xh_equal_coeff(const pel *residue, const int32_t *d0, const int32_t *d1, int s_der, long *eq, int w, int h, int vertex_num, hipStream_t st);
int xh_tx1d(bool fwd, const void *src, void *dst, int log2n, int shift, int line, int step, hipStream_t st);

// ------------------------------------------------------------------------------------------------
// state
// ------------------------------------------------------------------------------------------------
static std::atomic<int>      g_device{-1};
static std::atomic<uint64_t> g_table_calls{0}, g_table_calls_main{0};
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
// Every entry point runs XH_ENTER -> xh_ready(): a thread that has not yet worked for this binding of the library (a worker thread of the reference encoder starts
// on HIP device 0) is bound to the library's device first, so that its streams, arenas and resident planes land where the constant tables live.
bool xh_ready()
{
    const int dev = g_device.load();
    if(dev < 0) return false;
    static thread_local uint32_t t_bound_gen = 0xFFFFFFFFu;
    const uint32_t gen = g_generation.load();
    if(t_bound_gen != gen) {
        if(hipSetDevice(dev) != hipSuccess) return false;
        t_bound_gen = gen;
    }
    return true;
}
uint32_t xh_generation() { return g_generation.load(); }
static thread_local int t_vh = 0; // the virtual picture height of the batch this thread is walking (xh_common.h)
int xh_vh() { return t_vh; }
XhVhScope::XhVhScope(int vh) : prev(t_vh) { t_vh = vh; }
XhVhScope::~XhVhScope() { t_vh = prev; }
static thread_local int t_count_states = 0; // (xh_common.h, XhCountStatesScope)
int xh_count_states() { return t_count_states; }
XhCountStatesScope::XhCountStatesScope() : prev(t_count_states) { t_count_states = 1; }
XhCountStatesScope::~XhCountStatesScope() { t_count_states = prev; }

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
extern "C" uint64_t    xeve_hip_table_calls_main(void) { return g_table_calls_main.load(); }
extern "C" int xeve_hip_sizeof(int i)
{
    static const int sz[] = {(int)sizeof(xeve_hip_job), (int)sizeof(xeve_hip_mc_job), (int)sizeof(xeve_hip_me_params), (int)sizeof(xeve_hip_me_job),
                             (int)sizeof(xeve_hip_me_result), (int)sizeof(xeve_hip_spel_params), (int)sizeof(xeve_hip_spel_job), (int)sizeof(xeve_hip_epzs_job),
                             (int)sizeof(xeve_hip_epzs_params), (int)sizeof(xeve_hip_sbac), (int)sizeof(xeve_hip_cu_bits_params), (int)sizeof(xeve_hip_cu_bits_job),
                             (int)sizeof(xeve_hip_rdoq_est_full), (int)sizeof(xeve_hip_deblock_params), (int)sizeof(xeve_hip_refpic), (int)sizeof(xeve_hip_cu_mc_job),
                             (int)sizeof(xeve_hip_rdo_params), (int)sizeof(xeve_hip_rdo_job), (int)sizeof(xeve_hip_rdo_result), (int)sizeof(xeve_hip_skip_job),
                             (int)sizeof(xeve_hip_skip_result), (int)sizeof(xeve_hip_inter_params), (int)sizeof(xeve_hip_inter_job), (int)sizeof(xeve_hip_inter_result),
                             (int)sizeof(xeve_hip_intra_params), (int)sizeof(xeve_hip_intra_job), (int)sizeof(xeve_hip_intra_result),
                             (int)sizeof(xeve_hip_tree_params), (int)sizeof(xeve_hip_ctu_job), (int)sizeof(xeve_hip_ctu_data), (int)sizeof(xeve_hip_tree_inter), (int)sizeof(xeve_hip_eco_params)};
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
        prof_reset_locked(); // (the pooled events and unit counters live on the old device)
        xh_rdoq_tables_free();
        xh_resident_free_all();
        g_device.store(-1);
        g_generation++;
    }
    XH_HIP(hipSetDevice(device_ordinal));
    int rc = xh_tq_init();
    if(rc != XEVE_HIP_OK) return rc;
    g_generation++;                 // (threads bound to an earlier binding re-bind on their next entry)
    g_device.store(device_ordinal); // (the table builders below go through entry-point checks that want a bound device)
    rc = xh_main_tools_init();
    if(rc == XEVE_HIP_OK) rc = xh_rdoq_tables_init();
    if(rc != XEVE_HIP_OK) { // a half-initialised library must not look ready
        g_device.store(-1);
        g_generation++;
        return rc;
    }
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
std::vector<ProfTok *>   g_prof_done, g_prof_free; // (events are pooled: creating a pair per timed launch cost ~60 us of host time each -- measured)
unsigned long long      *g_prof_units = nullptr; // device, XEVE_HIP_PROF_CLASSES counters
} // namespace
bool xh_prof_on(int cls) { return (g_prof_mask.load(std::memory_order_relaxed) >> cls) & 1u; }
unsigned long long *xh_prof_units(int cls) { return xh_prof_on(cls) && g_prof_units ? g_prof_units + (size_t)cls * XH_PROF_STRIPES : nullptr; }
void *xh_prof_begin(int cls, hipStream_t st)
{
    ProfTok *t = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if(!g_prof_free.empty()) t = g_prof_free.back(), g_prof_free.pop_back();
    }
    if(!t) {
        t = new ProfTok{cls, nullptr, nullptr};
        if(hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess) {
            if(t->e0) (void)hipEventDestroy(t->e0);
            if(t->e1) (void)hipEventDestroy(t->e1);
            delete t;
            return nullptr;
        }
    }
    t->cls = cls;
    if(hipEventRecord(t->e0, st) != hipSuccess) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof_free.push_back(t);
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
void *xh_prof_begin_kernel(int cls, hipEvent_t *start, hipEvent_t *stop)
{
    ProfTok *t = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if(!g_prof_free.empty()) t = g_prof_free.back(), g_prof_free.pop_back();
    }
    if(!t) {
        t = new ProfTok{cls, nullptr, nullptr};
        if(hipEventCreate(&t->e0) != hipSuccess || hipEventCreate(&t->e1) != hipSuccess) {
            if(t->e0) (void)hipEventDestroy(t->e0);
            if(t->e1) (void)hipEventDestroy(t->e1);
            delete t;
            return nullptr;
        }
    }
    t->cls = cls, *start = t->e0, *stop = t->e1;
    return t;
}
void xh_prof_end_kernel(void *tok)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_done.push_back(static_cast<ProfTok *>(tok));
}
static void prof_reset_locked()
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for(std::vector<ProfTok *> *v : {&g_prof_done, &g_prof_free}) {
        for(ProfTok *t : *v) {
            (void)hipEventDestroy(t->e0), (void)hipEventDestroy(t->e1);
            delete t;
        }
        v->clear();
    }
    g_prof_mask.store(0);
    if(g_prof_units) (void)hipFree(g_prof_units), g_prof_units = nullptr;
}
extern "C" int xeve_hip_prof_enable(int class_mask)
{
    XH_ENTER();
    std::lock_guard<std::mutex> lk(g_prof_mu); // (g_prof_units is shared with prof_reset_locked / xeve_hip_prof_read)
    if(class_mask && !g_prof_units) {
        XH_HIP(hipMalloc((void **)&g_prof_units, sizeof(unsigned long long) * XEVE_HIP_PROF_CLASSES * XH_PROF_STRIPES));
        XH_HIP(hipMemset(g_prof_units, 0, sizeof(unsigned long long) * XEVE_HIP_PROF_CLASSES * XH_PROF_STRIPES));
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
            g_prof_free.push_back(k);
        }
        g_prof_done.clear();
    }
    if(g_prof_units) {
        static unsigned long long raw[XEVE_HIP_PROF_CLASSES * XH_PROF_STRIPES];
        std::lock_guard<std::mutex> lk(g_prof_mu);
        XH_HIP(hipMemcpy(raw, g_prof_units, sizeof(raw), hipMemcpyDeviceToHost));
        XH_HIP(hipMemset(g_prof_units, 0, sizeof(raw)));
        for(int i = 0; i < XEVE_HIP_PROF_CLASSES; i++)
            for(int k = 0; k < XH_PROF_STRIPES; k++) u[i] += raw[i * XH_PROF_STRIPES + k];
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
// `at` is the integer sample position the filter is centred on; REAL = taps the reference variant really applies (8 luma, 4 chroma, 2 bilinear:
// its footprint starts REAL / 2 - 1 samples before `at`), TAPS = width of the kernel's coefficient rows (REAL <= TAPS, the rows zero-padded).
template <int TAPS, int REAL>
static void mc_stage(const pel *at, int fx, int fy, bool hx, bool vy, int s_ref, int s_pred, pel *pred, int w, int h, int bit_depth, const int16_t *coef)
{
    constexpr int FS = TAPS == 8 ? 4 : 5, BACK = TAPS / 2 - 1, RBACK = REAL / 2 - 1;
    Stage &S = stage();
    g_table_calls++;
    const int sw = w + 16, sh = h + TAPS - 1;
    if((size_t)sw * sh * 2 > (64 << 10) || (size_t)w * h * 2 > (64 << 10)) {
        xh_set_error("mc table call %dx%d exceeds the staging tile", w, h);
        die(__func__);
    }
    int16_t *tile = S.h<int16_t>(REG_A);
    memset(tile, 0, sizeof(int16_t) * (size_t)sw * sh);
    const int x_lo = hx ? BACK - RBACK : BACK, cw = hx ? w + REAL - 1 : w;
    const int y_lo = vy ? BACK - RBACK : BACK, ch = vy ? h + REAL - 1 : h;
    gather(tile + y_lo * sw + x_lo, sw, at + (long)(y_lo - BACK) * s_ref + (x_lo - BACK), s_ref, cw, ch);
    xeve_hip_mc_job *job = S.h<xeve_hip_mc_job>(REG_MISC);
    job->gmv_x    = (BACK << FS) | fx;
    job->gmv_y    = (BACK << FS) | fy;
    job->pred_off = 0;
    job->frac     = (hx ? 1 : 0) | (vy ? 2 : 0);
    if(TAPS == 8) TBL_RC(xeve_hip_mc_l_jobs(S.d<pel>(REG_A), sw, S.d<pel>(REG_OUT), w, S.d<xeve_hip_mc_job>(REG_MISC), 1, w, h, bit_depth, (const int16_t(*)[8])coef, S.st));
    else TBL_RC(xeve_hip_mc_c_jobs(S.d<pel>(REG_A), sw, S.d<pel>(REG_OUT), w, S.d<xeve_hip_mc_job>(REG_MISC), 1, w, h, bit_depth, (const int16_t(*)[4])coef, S.st));
    S.sync();
    gather(pred, s_pred, S.h<int16_t>(REG_OUT), w, w, h);
}
template <int TAPS, bool HX, bool VY>
static void tbl_mc(pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, pel *pred, int w, int h, int bit_depth, const int16_t *coef)
{
    constexpr int FS = TAPS == 8 ? 4 : 5, FM = (1 << FS) - 1;
    mc_stage<TAPS, TAPS>(ref + (long)(gmv_y >> FS) * s_ref + (gmv_x >> FS), gmv_x & FM, gmv_y & FM, HX, VY, s_ref, s_pred, pred, w, h, bit_depth, coef);
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

}

// ---- Main profile, first slice: the interpolation variants the Main tools add (src_main/xevem_mc.c:167-485) ------------------------------
// The Main filters are fixed tables of the standard (the reference's copy: xevem_mc.c:48-126); the variants take no coefficient argument.
static const int16_t k_main_l[16][8] = {
    {0, 0, 0, 64, 0, 0, 0, 0},        {0, 1, -3, 63, 4, -2, 1, 0},      {-1, 2, -5, 62, 8, -3, 1, 0},     {-1, 3, -8, 60, 13, -4, 1, 0},
    {-1, 4, -10, 58, 17, -5, 1, 0},   {-1, 4, -11, 52, 26, -8, 3, -1},  {-1, 3, -9, 47, 31, -10, 4, -1},  {-1, 4, -11, 45, 34, -10, 4, -1},
    {-1, 4, -11, 40, 40, -11, 4, -1}, {-1, 4, -10, 34, 45, -11, 4, -1}, {-1, 4, -10, 31, 47, -9, 3, -1},  {-1, 3, -8, 26, 52, -11, 4, -1},
    {0, 1, -5, 17, 58, -10, 4, -1},   {0, 1, -4, 13, 60, -8, 3, -1},    {0, 1, -3, 8, 62, -5, 2, -1},     {0, 1, -2, 4, 63, -3, 1, 0}};
static const int16_t k_main_c[32][4] = {
    {0, 64, 0, 0},    {-1, 63, 2, 0},   {-2, 62, 4, 0},   {-2, 60, 7, -1},  {-2, 58, 10, -2}, {-3, 57, 12, -2}, {-4, 56, 14, -2}, {-4, 55, 15, -2},
    {-4, 54, 16, -2}, {-5, 53, 18, -2}, {-6, 52, 20, -2}, {-6, 49, 24, -3}, {-6, 46, 28, -4}, {-5, 44, 29, -4}, {-4, 42, 30, -4}, {-4, 39, 33, -4},
    {-4, 36, 36, -4}, {-4, 33, 39, -4}, {-4, 30, 42, -4}, {-4, 29, 44, -5}, {-4, 28, 46, -6}, {-3, 24, 49, -6}, {-2, 20, 52, -6}, {-2, 18, 53, -5},
    {-2, 16, 54, -4}, {-2, 15, 55, -4}, {-2, 14, 56, -4}, {-2, 12, 57, -3}, {-2, 10, 58, -2}, {-1, 7, 60, -2},  {0, 4, 62, -2},   {0, 2, 63, -1}};
// bilinear {64 - 4f, 4f} (xevem_mc.c:108-126) as rows of the 8-tap kernel: the two taps sit at positions 3 and 4, i.e. AT the sample and one after
static const int16_t (*main_bl())[8]
{
    static int16_t t[16][8];
    static std::once_flag once;
    std::call_once(once, [] {
        for(int f = 0; f < 16; f++) { t[f][3] = (int16_t)(64 - 4 * f); t[f][4] = (int16_t)(4 * f); }
    });
    return t;
}
// DMVR (xevem_mc.c:167-291, 383-463): `ref` points AT the block already, only the fraction of gmv is used
template <bool HX, bool VY> static void tbl_dmvr_l(pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, pel *pred, int w, int h, int bit_depth)
{
    g_table_calls_main++;
    mc_stage<8, 8>(ref, gmv_x & 15, gmv_y & 15, HX, VY, s_ref, s_pred, pred, w, h, bit_depth, &k_main_l[0][0]);
}
template <bool HX, bool VY> static void tbl_dmvr_c(pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, pel *pred, int w, int h, int bit_depth)
{
    g_table_calls_main++;
    mc_stage<4, 4>(ref, gmv_x & 31, gmv_y & 31, HX, VY, s_ref, s_pred, pred, w, h, bit_depth, &k_main_c[0][0]);
}
// bilinear (xevem_mc.c:293-378): the integer part of gmv moves ref; footprint (w + 1) x (h + 1)
template <bool HX, bool VY> static void tbl_bl_l(pel *ref, int gmv_x, int gmv_y, int s_ref, int s_pred, pel *pred, int w, int h, int bit_depth)
{
    g_table_calls_main++;
    mc_stage<8, 2>(ref + (long)(gmv_y >> 4) * s_ref + (gmv_x >> 4), gmv_x & 15, gmv_y & 15, HX, VY, s_ref, s_pred, pred, w, h, bit_depth, &main_bl()[0][0]);
}
extern "C" {
const XEVE_HIP_MCM xevem_tbl_dmvr_mc_l_hip[2][2] = {{tbl_dmvr_l<false, false>, tbl_dmvr_l<false, true>}, {tbl_dmvr_l<true, false>, tbl_dmvr_l<true, true>}};
const XEVE_HIP_MCM xevem_tbl_dmvr_mc_c_hip[2][2] = {{tbl_dmvr_c<false, false>, tbl_dmvr_c<false, true>}, {tbl_dmvr_c<true, false>, tbl_dmvr_c<true, true>}};
const XEVE_HIP_MCM xevem_tbl_bl_mc_l_hip[2][2]   = {{tbl_bl_l<false, false>, tbl_bl_l<false, true>}, {tbl_bl_l<true, false>, tbl_bl_l<true, true>}};

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
    const size_t n = (size_t)(1 << LOG2N) * line, in_b = n * (step != 1 ? 2 : 4), out_b = n * (step == 0 ? 4 : 2); // step 2: s16 -> s16 (Main)
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
// Main profile (tool_iqt): tx_pb{2..64} / itx_pb{2..64} keep the intermediate in 16 bit (src_main/xevem_tq.c:58-330, xevem_itdq.c:302-547)
template <bool FWD, int LOG2N> static void tbl_txm(int16_t *src, int16_t *dst, int shift, int line)
{
    g_table_calls_main++;
    tbl_tx<FWD, LOG2N>(src, dst, shift, line, 2);
}
extern "C" {
const XEVE_HIP_TX xeve_tbl_tx_hip[6]  = {tbl_txm<true, 1>, tbl_txm<true, 2>, tbl_txm<true, 3>, tbl_txm<true, 4>, tbl_txm<true, 5>, tbl_txm<true, 6>};
const XEVE_HIP_TX xeve_tbl_itx_hip[6] = {tbl_txm<false, 1>, tbl_txm<false, 2>, tbl_txm<false, 3>, tbl_txm<false, 4>, tbl_txm<false, 5>, tbl_txm<false, 6>};
}
// inverse ATS passes (xeve_func_itrans[type][log2 N - 1], xevem_itdq.c:42-47): DCT-VIII (row 0) and DST-VII (row 1) of 4 .. 32 points
template <int TYPE, int LOG2N> static void tbl_itrans_ats(int16_t *coef, int16_t *block, int shift, int line, int skip_line, int skip_line_2)
{
    Stage &S = stage();
    g_table_calls++, g_table_calls_main++;
    const size_t bytes = sizeof(int16_t) * (size_t)(1 << LOG2N) * line;
    if(bytes > (64 << 10)) {
        xh_set_error("inverse ATS table call N=%d line=%d exceeds the staging tile", 1 << LOG2N, line);
        die(__func__);
    }
    memcpy(S.h<char>(REG_A), coef, bytes);
    TBL_RC(xh_itrans_ats(TYPE, LOG2N, S.d<int16_t>(REG_A), S.d<int16_t>(REG_OUT), shift, line, skip_line, skip_line_2, S.st));
    S.sync();
    memcpy(block, S.h<char>(REG_OUT), bytes);
}
// forward ATS passes (xeve_trans_map_tbl[type][log2 N - 1], xevem_tq.c:53-56)
template <int TYPE, int LOG2N> static void tbl_trans_ats(int16_t *block, int16_t *coef, int shift, int line, int skip_line, int skip_line_2)
{
    Stage &S = stage();
    g_table_calls++, g_table_calls_main++;
    const size_t bytes = sizeof(int16_t) * (size_t)(1 << LOG2N) * line;
    if(bytes > (64 << 10)) {
        xh_set_error("forward ATS table call N=%d line=%d exceeds the staging tile", 1 << LOG2N, line);
        die(__func__);
    }
    memcpy(S.h<char>(REG_A), block, bytes);
    TBL_RC(xh_trans_ats(TYPE, LOG2N, S.d<int16_t>(REG_A), S.d<int16_t>(REG_OUT), shift, line, skip_line, skip_line_2, S.st));
    S.sync();
    memcpy(coef, S.h<char>(REG_OUT), bytes);
}
// Sobel derivatives of an affine prediction (xevem_func_aff_h / v_sobel_flt, xevem_mc.c:2341-2395)
template <int VERTICAL> static void tbl_sobel(pel *pred, int pred_stride, int *derivate, int derivate_buf_stride, int width, int height)
{
    Stage &S = stage();
    g_table_calls++, g_table_calls_main++;
    if(width > 128 || height > 128 || width < 3 || height < 3) {
        xh_set_error("sobel table call %dx%d outside 3 .. 128", width, height);
        die(__func__);
    }
    gather(S.h<int16_t>(REG_A), width, pred, pred_stride, width, height);
    TBL_RC(xh_sobel(VERTICAL, S.d<pel>(REG_A), width, S.d<int32_t>(REG_OUT), width, width, height, S.st));
    S.sync();
    const int32_t *o = S.h<int32_t>(REG_OUT);
    for(int y = 0; y < height; y++) memcpy(derivate + (size_t)y * derivate_buf_stride, o + (size_t)y * width, sizeof(int32_t) * width);
}
// angular intra prediction (xeve_func_intra_pred_ang[group][right], xevem_ipred.c:811-815).  Only the neighbour lines the variant reads are staged (element
// -1 .. w + h - 1 of each): above for every variant but [1][0], left for the variants without the right line except [0][0], right for the `right` variants.
template <int GROUP, int RIGHT>
static void tbl_ipred_ang(pel *src_le, pel *src_up, pel *src_ri, uint16_t avail_lr, pel *dst, int w, int h, int ipm, int bit_depth)
{
    (void)avail_lr;
    Stage &S = stage();
    g_table_calls++, g_table_calls_main++;
    if(w > 128 || h > 128 || w < 1 || h < 1) {
        xh_set_error("angular prediction table call %dx%d outside 1 .. 128", w, h);
        die(__func__);
    }
    const int L = w + h + 1;
    pel *lines = S.h<pel>(REG_A);
    memset(lines, 0, sizeof(pel) * 3 * (size_t)L);
    if(!(GROUP == 0) && !RIGHT) memcpy(lines, src_le - 1, sizeof(pel) * L);
    if(!(GROUP == 1 && !RIGHT)) memcpy(lines + L, src_up - 1, sizeof(pel) * L);
    if(RIGHT) memcpy(lines + 2 * L, src_ri - 1, sizeof(pel) * L);
    TBL_RC(xh_ipred_ang(GROUP, RIGHT, S.d<pel>(REG_A), S.d<pel>(REG_OUT), w, h, ipm, bit_depth, S.st));
    S.sync();
    memcpy(dst, S.h<pel>(REG_OUT), sizeof(pel) * (size_t)w * h);
}
extern "C" {
const XEVE_HIP_INTRA_PRED_ANG xeve_tbl_intra_pred_ang_hip[3][2] = {{tbl_ipred_ang<0, 0>, tbl_ipred_ang<0, 1>}, {tbl_ipred_ang<1, 0>, tbl_ipred_ang<1, 1>},
                                                                   {tbl_ipred_ang<2, 0>, tbl_ipred_ang<2, 1>}};
const XEVE_HIP_INV_TRANS xeve_trans_map_tbl_hip[16][5] = {
    {nullptr, tbl_trans_ats<0, 2>, tbl_trans_ats<0, 3>, tbl_trans_ats<0, 4>, tbl_trans_ats<0, 5>},
    {nullptr, tbl_trans_ats<1, 2>, tbl_trans_ats<1, 3>, tbl_trans_ats<1, 4>, tbl_trans_ats<1, 5>},
};
const XEVE_HIP_INV_TRANS xeve_itrans_map_tbl_hip[16][5] = {
    {nullptr, tbl_itrans_ats<0, 2>, tbl_itrans_ats<0, 3>, tbl_itrans_ats<0, 4>, tbl_itrans_ats<0, 5>},
    {nullptr, tbl_itrans_ats<1, 2>, tbl_itrans_ats<1, 3>, tbl_itrans_ats<1, 4>, tbl_itrans_ats<1, 5>},
};
void xevem_scaled_horizontal_sobel_filter_hip(pel *pred, int pred_stride, int *derivate, int derivate_buf_stride, int width, int height)
{
    tbl_sobel<0>(pred, pred_stride, derivate, derivate_buf_stride, width, height);
}
void xevem_scaled_vertical_sobel_filter_hip(pel *pred, int pred_stride, int *derivate, int derivate_buf_stride, int width, int height)
{
    tbl_sobel<1>(pred, pred_stride, derivate, derivate_buf_stride, width, height);
}
// the normal equations (xevem_func_aff_eq_coef_comp, xevem_mc.c:2397-2447); residue is read with the derivative pitch, like the reference does
void xevem_equal_coeff_computer_hip(pel *residue, int residue_stride, int **derivate, int derivate_buf_stride, int64_t (*equal_coeff)[7], int width, int height, int vertex_num)
{
    (void)residue_stride;
    Stage &S = stage();
    g_table_calls++, g_table_calls_main++;
    if(width > 128 || height > 128 || width < 1 || height < 1 || (vertex_num != 2 && vertex_num != 3)) {
        xh_set_error("equal-coefficient table call %dx%d, %d vertices: outside the supported set", width, height, vertex_num);
        die(__func__);
    }
    // REG_A (64 KB): the residual (dense, pitch = width), REG_B: the two derivative planes (dense), REG_MISC: the 7 x 7 accumulators
    gather(S.h<int16_t>(REG_A), width, residue, derivate_buf_stride, width, height);
    int32_t *d = S.h<int32_t>(REG_B);
    for(int k = 0; k < 2; k++)
        for(int y = 0; y < height; y++) memcpy(d + ((size_t)k * height + y) * width, derivate[k] + (size_t)y * derivate_buf_stride, sizeof(int32_t) * width);
    memcpy(S.h<char>(REG_MISC), equal_coeff, sizeof(int64_t) * 49);
    TBL_RC(xh_equal_coeff(S.d<pel>(REG_A), S.d<int32_t>(REG_B), S.d<int32_t>(REG_B) + (size_t)width * height, width, S.d<long>(REG_MISC), width, height, vertex_num, S.st));
    S.sync();
    memcpy(equal_coeff, S.h<char>(REG_MISC), sizeof(int64_t) * 49);
}
}

// ---- zero-edit installation into a loaded reference library ---------------------------------------------
struct Patch {
    const char *name;
    const void *value;
};
static int patch_globals(const Patch *pats, int npat, const char *who)
{
    for(int i = 0; i < npat; i++) // all or nothing: look every symbol up before the first store
        if(!dlsym(RTLD_DEFAULT, pats[i].name)) {
            xh_set_error("%s: symbol %s not found (is the reference library loaded RTLD_GLOBAL?)", who, pats[i].name);
            return XEVE_HIP_ERR_ARG;
        }
    for(int i = 0; i < npat; i++) *(void **)dlsym(RTLD_DEFAULT, pats[i].name) = const_cast<void *>(pats[i].value);
    return npat;
}
static const Patch k_base_patches[] = {
    {"xeve_func_sad", xeve_tbl_sad_16b_hip},   {"xeve_func_ssd", xeve_tbl_ssd_16b_hip},
    {"xeve_func_diff", xeve_tbl_diff_16b_hip}, {"xeve_func_satd", xeve_tbl_satd_16b_hip},
    {"xeve_func_mc_l", xeve_tbl_mc_l_hip},     {"xeve_func_mc_c", xeve_tbl_mc_c_hip},
    {"xeve_func_average_no_clip", (const void *)xeve_average_16b_no_clip_hip},
    {"xeve_func_txb", &xeve_tbl_txb_hip},
};
static const Patch k_main_patches[] = {
    {"xevem_func_dmvr_mc_l", xevem_tbl_dmvr_mc_l_hip}, {"xevem_func_dmvr_mc_c", xevem_tbl_dmvr_mc_c_hip}, {"xevem_func_bl_mc_l", xevem_tbl_bl_mc_l_hip},
    {"xeve_func_tx", &xeve_tbl_tx_hip},                {"xeve_func_itx", &xeve_tbl_itx_hip},
    {"xeve_func_itrans", xeve_itrans_map_tbl_hip},
    {"xeve_func_intra_pred_ang", xeve_tbl_intra_pred_ang_hip},
    {"xevem_func_aff_h_sobel_flt", (const void *)xevem_scaled_horizontal_sobel_filter_hip},
    {"xevem_func_aff_v_sobel_flt", (const void *)xevem_scaled_vertical_sobel_filter_hip},
    {"xevem_func_aff_eq_coef_comp", (const void *)xevem_equal_coeff_computer_hip},
};
extern "C" int xeve_hip_install_tables(void *fn_itxb_slot)
{
    XH_ENTER();
    int n = patch_globals(k_base_patches, (int)(sizeof(k_base_patches) / sizeof(Patch)), __func__);
    if(n < 0) return n;
    if(fn_itxb_slot) {
        *(const void **)fn_itxb_slot = &xeve_tbl_itxb_hip;
        n++;
    }
    return n;
}
extern "C" int xeve_hip_install_tables_main(void *fn_itxb_slot)
{
    XH_ENTER();
    const int NB = (int)(sizeof(k_base_patches) / sizeof(Patch)), NM = (int)(sizeof(k_main_patches) / sizeof(Patch));
    Patch all[NB + NM];
    for(int i = 0; i < NB; i++) all[i] = k_base_patches[i];
    for(int i = 0; i < NM; i++) all[NB + i] = k_main_patches[i];
    int n = patch_globals(all, NB + NM, __func__);
    if(n < 0) return n;
    if(fn_itxb_slot) {
        *(const void **)fn_itxb_slot = &xeve_tbl_itxb_hip;
        n++;
    }
    return n;
}
