// xeve_amd/csrc/walk.hip -- the fused CTU walk on the device: ONE kernel launch decides a CTU of every chain of the call (walk.h: a workgroup per team of chains, the
// whole quad-tree schedule executed inside the kernel).  This file is the launcher: device copies of the tables and of the schedule, the parameter record, the launch.
// xeve_hip_mode_analyze_ctu_jobs (tree.hip) routes here unless XEVE_HIP_WALK=0 (the composed walk: ~10 000 launches per CTU step, kept for A/B measurements).
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <vector>
// the count-only event loop of the coder as a real function: a register allocation of its own (inlined into the stages, the coder's range travelled through a spill slot
// every event) -- measured: 832x480, 128 chains, I / B step 98.9 / 128.2 -> 94.4 / 120.2 ms, the same bytes (profiles/r04z_noinline_coder.log)
#define XW_NOINLINE_CODER 1
#include "xh_common.h"
#include "walk_setup.h"
#ifndef XW_WG_PER_CU
#define XW_WG_PER_CU 4
#endif

// A team's SERIAL stages (a coder job, an RDOQ scan, a per-chain decision per lane: 3 .. 60 busy lanes, ~55 % of a step's time) land on its logical threads 0 .. 63.  A CU
// holds four teams, one wave of each on every SIMD, and a SIMD issues ONE wave instruction per 4 clocks however many waves are resident: with every team's serial lanes in
// its wave 0, the four teams' coder loops share one SIMD's issue slots (a bin: ~40 instructions -> 640 clocks per bin instead of 160) while three SIMDs idle at the
// barrier.  So a team takes its LOGICAL wave 0 from the SIMD its arrival order on the CU names (p.cu_arrivals: a counter per physical CU, read from HW_ID / XCC_ID):
// the logical thread index is the hardware one with the waves rotated; whole waves move, lanes keep their place, nothing else in the walk knows.  Results cannot
// depend on it (the host harness runs the same code with rotated teams of real threads).
template <bool FULL> __global__ void __launch_bounds__(XW_NT, XW_WG_PER_CU) k_walk(xw::P p, int *cu_arrivals)
{
    __shared__ xw::Lds S;
    __shared__ int s_simd[XW_NT / 64], s_order;
    int tid = (int)threadIdx.x;
#if defined(__gfx950__) || defined(__gfx942__) // (the HW_ID / XCC_ID register numbers and bit fields below are these targets': any other keeps the launch order -- ADVICE r05)
    if(cu_arrivals && blockDim.x == XW_NT) {
        constexpr int REG_HW_ID = 4, REG_XCC_ID = 20; // s_getreg_b32 operands: (size - 1) << 11 | offset << 6 | register (gfx950: hip/amd_detail/amd_device_functions.h)
        const unsigned hw = __builtin_amdgcn_s_getreg(((32 - 1) << 11) | REG_HW_ID);
        if((tid & 63) == 0) s_simd[tid >> 6] = (int)((hw >> 4) & 3u); // SIMD_ID
        if(tid == 0) {
            const unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | REG_XCC_ID);
            s_order = atomicAdd(&cu_arrivals[((xcc & 15u) << 7) | ((hw >> 8) & 127u)], 1); // CU_ID 11:8, SH_ID 12, SE_ID 14:13
        }
        __syncthreads();
        const int want = s_order & 3;
        int first = want; // (the wave on the SIMD this team's arrival order names; waves that all sit on other SIMDs: the wave of that index)
        for(int w = XW_NT / 64 - 1; w >= 0; w--)
            if(s_simd[w] == want) first = w;
        tid = ((((tid >> 6) - first) & (XW_NT / 64 - 1)) << 6) | (tid & 63);
    }
#endif
    const xw::Tm tm = {tid, (int)blockDim.x};
    xw::walk_team<FULL>(tm, p, S, (int)blockIdx.x);
}

namespace {
struct OpsEntry {
    int                 key[8];
    xw::Op             *dev;
    int                 n;
};
struct WalkDev { // device copies, rebuilt when the library is re-bound
    uint32_t  gen = 0;
    int8_t   *dct = nullptr;
    uint16_t *scan = nullptr;
    int32_t  *entropy = nullptr;
    unsigned long long *prof = nullptr;
    int      *arrivals = nullptr; // teams that have arrived on every physical CU so far (k_walk: a team's serial wave by its arrival order); 2048 counters
    int16_t  *mc = nullptr; // [16][8] luma, then [32][4] chroma (the tables of the last call; compared before reuse)
    std::vector<int16_t> mc_host;
    std::vector<OpsEntry> ops;
    std::mutex mu;
    void drop()
    {
        if(dct) (void)hipFree(dct);
        if(scan) (void)hipFree(scan);
        if(entropy) (void)hipFree(entropy);
        if(mc) (void)hipFree(mc);
        if(prof) (void)hipFree(prof);
        if(arrivals) (void)hipFree(arrivals);
        prof = nullptr, arrivals = nullptr;
        for(auto &e : ops) (void)hipFree(e.dev);
        dct = nullptr, scan = nullptr, entropy = nullptr, mc = nullptr, ops.clear(), mc_host.clear();
    }
};
WalkDev g_walk;
std::atomic<int> g_walk_prof_on{0};
std::atomic<long> g_walk_load{0}; // chains of the batch encoders with a run in progress in this process (they launch side by side, a stream each)
} // namespace

// a batch encoder starts (+) or ends (-) a run of `chains` lockstep chains: the walks of all running encoders share the chip's workgroup slots
void xh_walk_load(long chains) { g_walk_load.fetch_add(chains); }

// which walk a call of `nchains` chains runs.  Rounds 4-5: the fused kernel up to 1024 chains (it finished a step of few chains sooner: ~60 / 120 ms per intra / inter CTU
// of noise against the composed walk's launch-bound ~110 / 185 ms), the composed walk above.  Round 6: with its side stream and a third fewer launches the composed walk
// finishes a step sooner at EVERY width (8 chains: 55 / 62 ms against 69 / 103; 1024 chains: 67 / 81 against 79 / 116 -- profiles/r06_side_stream.md), so the choice by
// width is the composed walk throughout (XEVE_HIP_WALK_AUTO_MAX = 0); the fused kernel keeps what only it codes (presets slow and placebo: xh_walk_only) and stays
// pinned by XEVE_HIP_WALK=1 / xeve_hip_walk_select(1).  nchains = the width of the batch the call belongs to (tree.hip).
namespace {
// the walk choice: -1 by the width (fused up to g_walk_auto_max chains), 0 the composed walk, 1 the fused kernel.  Starts from XEVE_HIP_WALK / XEVE_HIP_WALK_AUTO_MAX;
// xeve_hip_walk_select moves it at run time (one process can then pin each walk in turn: tests/test_enc_gpu.py)
int walk_env_mode()
{
    const char *e = getenv("XEVE_HIP_WALK");
    return e && *e && strcmp(e, "auto") ? (atoi(e) != 0) : -1;
}
std::atomic<int> g_walk_mode{walk_env_mode()};
std::atomic<int> g_walk_auto_max{getenv("XEVE_HIP_WALK_AUTO_MAX") ? atoi(getenv("XEVE_HIP_WALK_AUTO_MAX")) : 0};
} // namespace
bool xh_walk_enabled(int nchains)
{
    const int on = g_walk_mode.load(std::memory_order_relaxed);
    return on < 0 ? nchains <= g_walk_auto_max.load(std::memory_order_relaxed) : on != 0;
}
std::atomic<int> g_walk_team{getenv("XEVE_HIP_WALK_C") ? atoi(getenv("XEVE_HIP_WALK_C")) : 0};
extern "C" int xeve_hip_walk_team(int chains_per_team)
{
    const int before = g_walk_team.load();
    if(chains_per_team >= 0 && chains_per_team <= XW_MAXC) g_walk_team.store(chains_per_team);
    return before;
}
extern "C" int xeve_hip_walk_select(int mode)
{
    const int before = g_walk_mode.load();
    if(mode >= -1 && mode <= 1) g_walk_mode.store(mode);
    return before;
}
bool xh_walk_supported(const xeve_hip_tree_params *p, const xeve_hip_tree_inter *I, int nchains)
{
    // (rdo_dbk_switch and inter CUs of 4x4 -- presets slow and placebo -- are the fused walk's alone: whatever the width, unless the composed walk is pinned -- then the
    // call is refused, tree.hip)
    if(xh_walk_only(p) ? g_walk_mode.load(std::memory_order_relaxed) == 0 : !xh_walk_enabled(nchains)) return false;
    if(p->ip.slice_type != 2 && I) {
        static const int inter_on = getenv("XEVE_HIP_WALK_INTER") ? atoi(getenv("XEVE_HIP_WALK_INTER")) : 1;
        if(!inter_on) return false;
        const int n0 = I->ipar.rdo.num_refp[0], n1 = I->ipar.rdo.num_refp[1];
        if(n0 > XW_MAXR || n1 > XW_MAXR) return false;
        // a diamond's remaining rings are one round of the fused search (walk_inter.h dia_round: 5 + 9 + 16 per doubling of the step from 16 on, 94 candidates for the
        // steps 4 .. 256): they fit MeJob::cx[96] while the range stays below 512 (placebo's 384 has the rings of 256); beyond it the composed walk runs the call
        if(I->ipar.me.me.max_search_range > 511) return false;
    }
    return p->log2_ctu <= 6;
}
extern "C" int xeve_hip_walk_fused(int nchains) { return xh_walk_enabled(nchains) ? 1 : 0; }
size_t xh_walk_workspace(int nchains) { return (size_t)nchains * sizeof(xw::Cw) + 256; }

int xh_walk_run(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu, int8_t *map_ipm,
                const uint8_t *map_tidx, uint32_t *map_cu_mode, const int64_t *pic_elems, const xeve_hip_sbac *states, const xeve_hip_tree_params *p,
                const xeve_hip_tree_inter *I, const xeve_hip_ctu_job *jobs, int nchains, xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost, void *workspace,
                size_t workspace_bytes, int vh, hipStream_t st)
{
    XH_REQUIRE(workspace_bytes >= xh_walk_workspace(nchains));
    const int C_env = g_walk_team.load(std::memory_order_relaxed); // (XEVE_HIP_WALK_C / xeve_hip_walk_team; 0: by the load)
    static const int NT_env = getenv("XEVE_HIP_WALK_NT") ? atoi(getenv("XEVE_HIP_WALK_NT")) : XW_NT;
    const int wg_slots = [] { // of the device this thread is bound to NOW (the library can be re-bound to another GPU: xh_generation)
        int dev = 0, cus = 256;
        if(hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        return (cus > 0 ? cus : 256) * XW_WG_PER_CU;
    }();
    // chains per team: a team's step takes the longer the more chains it carries (its stages run them in lockstep), so as few as keep every chain in flight resident at once --
    // all running encoders' chains over the chip's workgroup slots (XEVE_HIP_WALK_C pins it)
    const long in_flight = std::max<long>(g_walk_load.load(), nchains);
    const int  C_auto = (int)std::min<long>(XW_MAXC, (in_flight + wg_slots - 1) / wg_slots);
    const int  C = C_env < 1 ? std::max(1, C_auto) : C_env > XW_MAXC ? XW_MAXC : C_env, NT = NT_env < 64 ? 64 : NT_env > XW_NT ? XW_NT : (NT_env & ~63);
    xw::P q;
    xw::fill_params(q, org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, pic_elems, states, p, I, jobs, nchains, out, next_best, cost, vh);
    static const int force_count = getenv("XEVE_HIP_WALK_COUNT") ? atoi(getenv("XEVE_HIP_WALK_COUNT")) : 0; // (probes: the encoder's count-only states from any caller)
    q.C = C, q.full = !xh_count_states() && !force_count;
    // (the lock is held over the launch: a call with another filter set must not rewrite the table between this call's copy and its kernel reading it)
    std::lock_guard<std::mutex> lk(g_walk.mu);
    WalkDev &D = g_walk;
    if(D.gen != xh_generation()) D.drop(), D.gen = xh_generation();
    if(!D.dct) { // all three tables or none: a failure half way must not leave later calls with a null or unwritten table
        xw::Tables T;
        xw::make_tables(T);
        int8_t *d = nullptr;
        uint16_t *sc = nullptr;
        int32_t *en = nullptr;
        const bool ok = hipMalloc((void **)&d, T.dct.size()) == hipSuccess && hipMalloc((void **)&sc, T.scan.size() * 2) == hipSuccess &&
                        hipMalloc((void **)&en, T.entropy.size() * 4) == hipSuccess && hipMemcpy(d, T.dct.data(), T.dct.size(), hipMemcpyHostToDevice) == hipSuccess &&
                        hipMemcpy(sc, T.scan.data(), T.scan.size() * 2, hipMemcpyHostToDevice) == hipSuccess &&
                        hipMemcpy(en, T.entropy.data(), T.entropy.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
        if(!ok) {
            (void)hipFree(d), (void)hipFree(sc), (void)hipFree(en);
            xh_set_error("xeve_hip walk: the device tables could not be set up");
            return XEVE_HIP_ERR_DEVICE;
        }
        D.dct = d, D.scan = sc, D.entropy = en;
    }
    if(I) { // the interpolation filters: the caller's host tables (pi->mc_l_coeff / mc_c_coeff)
        std::vector<int16_t> h(16 * 8 + 32 * 4, 0);
        memcpy(h.data(), I->coef_l, 16 * 8 * 2);
        if(I->coef_c) memcpy(h.data() + 128, I->coef_c, 32 * 4 * 2);
        if(!D.mc || h != D.mc_host) {
            int16_t *m = D.mc;
            if(!m) XH_HIP(hipMalloc((void **)&m, h.size() * 2));
            else XH_HIP(hipDeviceSynchronize()); // (a different filter set than the last call's: wait for the launches that read the old one)
            if(hipMemcpy(m, h.data(), h.size() * 2, hipMemcpyHostToDevice) != hipSuccess) {
                if(!D.mc) (void)hipFree(m);
                D.mc_host.clear(); // (the old table's content is no longer known)
                xh_set_error("xeve_hip walk: the interpolation filters could not be copied");
                return XEVE_HIP_ERR_DEVICE;
            }
            D.mc = m, D.mc_host = h;
        }
        q.mc_l = D.mc, q.mc_c = D.mc + 128;
    }
    const int key[8] = {p->log2_ctu, p->max_cu, p->min_cu, p->min_cuwh, p->pic_w, p->pic_h, I != nullptr, 0};
    const OpsEntry *hit = nullptr;
    for(const auto &e : D.ops)
        if(!memcmp(e.key, key, sizeof(key))) hit = &e;
    if(!hit) {
        const std::vector<xw::Op> ops = xw::make_ops(p, I != nullptr);
        OpsEntry e;
        memcpy(e.key, key, sizeof(key)), e.n = (int)ops.size();
        XH_HIP(hipMalloc((void **)&e.dev, ops.size() * sizeof(xw::Op)));
        XH_HIP(hipMemcpy(e.dev, ops.data(), ops.size() * sizeof(xw::Op), hipMemcpyHostToDevice));
        D.ops.push_back(e);
        hit = &D.ops.back();
    }
    q.ops = hit->dev, q.nops = hit->n, q.dct = D.dct, q.scan = D.scan, q.entropy = D.entropy;
    static const int prof_env = getenv("XEVE_HIP_WALK_PROF") ? atoi(getenv("XEVE_HIP_WALK_PROF")) : 0;
    const int prof_on = prof_env || g_walk_prof_on.load();
    if(!prof_on) q.prof = nullptr;
    if(prof_on && !D.prof) {
        XH_HIP(hipMalloc((void **)&D.prof, 2 * xw::PR_N * 8));
        XH_HIP(hipMemset(D.prof, 0, 2 * xw::PR_N * 8));
    }
    q.prof = prof_on ? D.prof : nullptr;
    static const int dbg = getenv("XEVE_HIP_WALK_DBG") ? atoi(getenv("XEVE_HIP_WALK_DBG")) : 0;
    q.dbg = dbg;
    // the serial stages' lanes: packed into the team's logical wave 0 (1), which k_walk takes from a different SIMD for each of a CU's teams (spread) -- or, as round 4
    // shipped, dealt over the team's four waves (XEVE_HIP_WALK_DEAL=0 XEVE_HIP_WALK_SPREAD=0)
    static const int spread = getenv("XEVE_HIP_WALK_SPREAD") ? atoi(getenv("XEVE_HIP_WALK_SPREAD")) : 1;
    static const int deal = getenv("XEVE_HIP_WALK_DEAL") ? atoi(getenv("XEVE_HIP_WALK_DEAL")) : (spread ? 1 : 0);
    q.deal = deal;
    if(spread && !D.arrivals) {
        int *a = nullptr;
        if(hipMalloc((void **)&a, 2048 * sizeof(int)) != hipSuccess || hipMemset(a, 0, 2048 * sizeof(int)) != hipSuccess) {
            (void)hipFree(a);
            xh_set_error("xeve_hip walk: the arrival counters could not be set up");
            return XEVE_HIP_ERR_DEVICE;
        }
        D.arrivals = a;
    }
    int *const arrivals = spread && NT == XW_NT ? D.arrivals : nullptr;
    q.cw = (xw::Cw *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    const int teams = (nchains + C - 1) / C;
    q.sad_units = xh_prof_units(XH_PROF_WALK);
    XhProf timer(XH_PROF_WALK, st); // (HIP events on the walk's own stream around its one launch, when the class is switched on)
    if(q.full) k_walk<true><<<teams, NT, 0, st>>>(q, arrivals);
    else k_walk<false><<<teams, NT, 0, st>>>(q, arrivals);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// XEVE_HIP_WALK_PROF=1: cycles and marks per stage class of team 0 since the last call (out[0 .. n): cycles, out[n .. 2n): marks); returns the number of classes
extern "C" int xeve_hip_walk_prof(unsigned long long *out, int cap)
{
    std::lock_guard<std::mutex> lk(g_walk.mu);
    if(!g_walk.prof || cap < 2 * xw::PR_N) return 0;
    if(hipDeviceSynchronize() != hipSuccess) return 0;
    if(hipMemcpy(out, g_walk.prof, 2 * xw::PR_N * 8, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    (void)hipMemset(g_walk.prof, 0, 2 * xw::PR_N * 8);
    return xw::PR_N;
}

// the in-kernel stage profile on / off for the walks launched from now on (the same switch as XEVE_HIP_WALK_PROF=1, at run time)
extern "C" int xeve_hip_walk_prof_enable(int on)
{
    g_walk_prof_on.store(on != 0);
    return XEVE_HIP_OK;
}
