// tests/native/walk_host.cpp -- TEST INFRASTRUCTURE: the HOST side of xeve_amd/csrc/walk.h (the fused CTU walk libxeve_hip.so runs as one kernel, every function
// __host__ __device__) as a team of ONE thread (or, xw_host_walk_mt, of several real threads), so that `pytest -m "not gpu"` holds it bit for bit against the pinned oracle without a GPU.  Built with
// hipcc -x hip --cuda-host-only by tests/_walk.py; nothing of this is linked into the product library.
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../../xeve_amd/csrc/walk_setup.h"

extern "C" {
// (tests/test_walk_race.py: proof that the race detector is alive in the process -- two threads write one word with nothing between them)
int xw_host_race_selftest(void)
{
    static int word;
    std::thread a([] { word = 1; }), b([] { word = 2; });
    a.join(), b.join();
    return word;
}
size_t xw_host_sizeof_cw(void) { return sizeof(xw::Cw); }
size_t xw_host_sizeof_lds(void) { return sizeof(xw::Lds); }
size_t xw_host_sizeof_p(void) { return sizeof(xw::P); }
// all pointers are host memory; I (may be NULL) carries a host refp table and the filter tables; C = chains per team; full = complete coder states; threads = the team's
// size (1: every stage a plain loop; 256: the device's lane mapping with real threads)
int xw_host_walk_mt(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu, int8_t *map_ipm,
                    const uint8_t *map_tidx, uint32_t *map_cu_mode, const int64_t *pic_elems, const xeve_hip_sbac *states, const xeve_hip_tree_params *p,
                    const xeve_hip_tree_inter *I, const xeve_hip_ctu_job *jobs, int nchains, xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost, int C, int full, int vh,
                    int threads)
{
    static xw::Tables T;
    if(T.dct.empty()) xw::make_tables(T);
    if(p->ip.slice_type == 2) I = nullptr;
    xw::P q;
    xw::fill_params(q, org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, pic_elems, states, p, I, jobs, nchains, out, next_best, cost, vh);
    const std::vector<xw::Op> ops = xw::make_ops(p, I != nullptr);
    std::vector<xw::Cw> cw((size_t)nchains);
    memset((void *)cw.data(), 0xCD, cw.size() * sizeof(xw::Cw)); // (the device's workspace and LDS start with whatever was there: nothing may depend on zeros)
    q.C = C < 1 ? 1 : C > XW_MAXC ? XW_MAXC : C, q.full = full;
    q.entropy = T.entropy.data(), q.dct = T.dct.data(), q.scan = T.scan.data(), q.ops = ops.data(), q.nops = (int)ops.size(), q.cw = cw.data();
    if(I) q.mc_l = &I->coef_l[0][0], q.mc_c = I->coef_c ? &I->coef_c[0][0] : nullptr;
    xw::Lds *S = (xw::Lds *)malloc(sizeof(xw::Lds));
    memset((void *)S, 0xCD, sizeof(xw::Lds));
    const int nt = threads < 1 ? 1 : threads;
    // XW_HOST_DEAL / XW_HOST_ROTATE: the device's two lane layouts of the serial stages (walk.hip: P::deal; k_walk's rotation of a team's waves) on the host's teams of
    // real threads -- the results must not depend on either
    q.deal = getenv("XW_HOST_DEAL") ? atoi(getenv("XW_HOST_DEAL")) : 0;
    const int rot = getenv("XW_HOST_ROTATE") && nt >= 128 && nt % 64 == 0 ? atoi(getenv("XW_HOST_ROTATE")) % (nt / 64) : 0;
    if(nt == 1) {
        const xw::Tm tm = {0, 1};
        for(int team = 0; team * q.C < nchains; team++) {
            if(full) xw::walk_team<true>(tm, q, *S, team);
            else xw::walk_team<false>(tm, q, *S, team);
        }
    }
    else { // a team of nt real threads: sync() is a pthread barrier (which ThreadSanitizer understands), the teams one after the other
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, nullptr, (unsigned)nt);
        std::vector<std::thread> th;
        for(int t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                xw::host_team() = {[](void *b) { pthread_barrier_wait((pthread_barrier_t *)b); }, &bar};
                const xw::Tm tm = {rot ? ((((t >> 6) - rot + nt / 64) % (nt / 64)) << 6) | (t & 63) : t, nt};
                for(int team = 0; team * q.C < nchains; team++) {
                    if(full) xw::walk_team<true>(tm, q, *S, team);
                    else xw::walk_team<false>(tm, q, *S, team);
                    pthread_barrier_wait(&bar);
                }
                xw::host_team() = {nullptr, nullptr};
            });
        for(auto &t : th) t.join();
        pthread_barrier_destroy(&bar);
    }
    free(S);
    return 0;
}
int xw_host_walk(const xeve_hip_pel *const org[3], int s_org_l, int s_org_c, xeve_hip_pel *const mod[3], int s_mod_l, int s_mod_c, uint32_t *map_scu, int8_t *map_ipm,
                 const uint8_t *map_tidx, uint32_t *map_cu_mode, const int64_t *pic_elems, const xeve_hip_sbac *states, const xeve_hip_tree_params *p,
                 const xeve_hip_tree_inter *I, const xeve_hip_ctu_job *jobs, int nchains, xeve_hip_ctu_data *out, xeve_hip_sbac *next_best, double *cost, int C, int full, int vh)
{
    return xw_host_walk_mt(org, s_org_l, s_org_c, mod, s_mod_l, s_mod_c, map_scu, map_ipm, map_tidx, map_cu_mode, pic_elems, states, p, I, jobs, nchains, out, next_best, cost, C, full, vh, 1);
}
}
