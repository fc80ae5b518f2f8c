// xeve_amd/csrc/eco.hip -- the bitstream writer's side of a batch of decided CTUs on the device: xeve_eco_tree (src_base/xeve_enc.c:35-100) per chain, one LANE per
// chain (eco_lane.h).  The coder state a chain carries is the WRITER's: after this call it is the state the chain's next CTU starts its mode decision from
// (xeve_enc.c:139) -- with xeve_hip_mode_analyze_ctu_jobs (tree.hip) a chain runs from CTU to CTU without the host, and the bytes collected here are the picture's
// slice data (up to the pending byte and the code register, which stay in the state until the tile ends).
#include "xh_common.h"
// the coder's context models in LDS ([model][lane]) and, every lane function forced inline, its core in registers (cu_lane.h: XL, XL_CTX); see encode.hip
#define XL_NCTX 72
static __shared__ uint16_t xl_lds_ctx[XL_NCTX * 64];
#if defined(__HIP_DEVICE_COMPILE__)
#define XL __host__ __device__ static inline __attribute__((always_inline))
#define XL_CTX(s, ci) xl_lds_ctx[(ci) * 64 + (threadIdx.x & 63)]
#define XL_SINK(o) true // (every coder of this file writes: see cu_lane.h)
#endif
#include "eco_lane.h"
static_assert(sizeof(((xeve_hip_sbac *)0)->ctx) == XL_NCTX * sizeof(uint16_t), "the models' LDS image");

template <bool WAVE> __global__ void __launch_bounds__(64) k_eco_ctu(const xeve_hip_ctu_data *__restrict__ ctus, xeve_hip_sbac *__restrict__ states, xl::EcoParams E, uint32_t *map_scu,
                                                const int8_t *map_ipm, const uint8_t *map_tidx, uint32_t *map_cu_mode, long map_pic, const xeve_hip_ctu_job *__restrict__ jobs,
                                                int nchains, uint8_t *__restrict__ bytes, int bytes_cap, int32_t *__restrict__ nbytes)
{
    const int c = blockIdx.x; // one chain per wave, one lane working (an arithmetic coder's control flow follows its data: chains sharing a wave run one after the other)
    if(c >= nchains || (!WAVE && threadIdx.x != 0)) return; // (WAVE: all 64 lanes run the writer in step and share the coefficient scans, eco_lane.h eco_levels)
    const xeve_hip_ctu_job J = jobs[c];
    xl::Sbac s = states[J.sbac];
#pragma unroll
    for(int i = 0; i < XL_NCTX; i++) xl_lds_ctx[i * 64 + (threadIdx.x & 63)] = s.ctx[i];
    xl::Sink o = {bytes + (long)c * bytes_cap, bytes_cap, 0};
    xl::eco_ctu<WAVE>(E, s, ctus[c], map_scu + J.pic * map_pic, map_ipm + J.pic * map_pic, map_tidx + J.pic * map_pic, map_cu_mode + J.pic * map_pic, J.x, J.y, &o);
#pragma unroll
    for(int i = 0; i < XL_NCTX; i++) s.ctx[i] = xl_lds_ctx[i * 64 + (threadIdx.x & 63)];
    states[J.sbac] = s;
    nbytes[c] = o.n;
}

extern "C" int xeve_hip_eco_ctu_jobs(const xeve_hip_ctu_data *ctus, xeve_hip_sbac *states, int nstates, const xeve_hip_eco_params *p, uint32_t *map_scu, const int8_t *map_ipm,
                                     const uint8_t *map_tidx, uint32_t *map_cu_mode, int64_t map_pic_elems, const xeve_hip_ctu_job *jobs, int nchains, uint8_t *bytes,
                                     int bytes_cap, int32_t *nbytes, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(ctus && states && nstates > 0 && p && map_scu && map_ipm && map_tidx && map_cu_mode && jobs && nchains >= 0 && bytes && bytes_cap > 0 && nbytes);
    XH_REQUIRE(p->log2_ctu >= 3 && p->log2_ctu <= 6 && p->pic_w > 0 && p->pic_h > 0 && (p->pic_w & 3) == 0 && (p->pic_h & 3) == 0 && p->w_scu == (p->pic_w + 3) >> 2 &&
               p->h_scu == (p->pic_h + 3) >> 2 && p->slice_type >= 0 && p->slice_type <= 2 &&
               (p->chroma_format_idc == 0 || p->chroma_format_idc == 1 || p->chroma_format_idc == 3));
    XH_REQUIRE(p->slice_type == 2 || (p->num_refp[0] >= 1 && p->num_refp[0] <= XEVE_HIP_MAX_REFP && p->num_refp[1] >= 0 && p->num_refp[1] <= XEVE_HIP_MAX_REFP));
    if(nchains == 0) return XEVE_HIP_OK;
    xl::EcoParams E;
    E.idc = p->chroma_format_idc, E.slice_type = p->slice_type, E.log2_ctu = p->log2_ctu, E.pic_w = p->pic_w, E.pic_h = p->pic_h, E.w_scu = p->w_scu;
    E.num_refp[0] = p->num_refp[0], E.num_refp[1] = p->num_refp[1];
    for(int i = 0; i < 7; i++) E.scan[i] = nullptr;
    for(int l = 4; l <= p->log2_ctu; l++) {
        const int rc = xh_get_scan(l, l, &E.scan[l]);
        if(rc != XEVE_HIP_OK) return rc;
    }
    static const bool wave = !getenv("XEVE_HIP_WRITER_WAVE") || atoi(getenv("XEVE_HIP_WRITER_WAVE")) != 0; // (developer switch: 0 = the lone-lane form)
    if(wave) k_eco_ctu<true><<<nchains, 64, 0, (hipStream_t)stream>>>(ctus, states, E, map_scu, map_ipm, map_tidx, map_cu_mode, (long)map_pic_elems, jobs, nchains, bytes, bytes_cap, nbytes);
    else k_eco_ctu<false><<<nchains, 64, 0, (hipStream_t)stream>>>(ctus, states, E, map_scu, map_ipm, map_tidx, map_cu_mode, (long)map_pic_elems, jobs, nchains, bytes, bytes_cap, nbytes);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}

// the end of a tile for a batch of chains: the terminating bin and xeve_sbac_finish on each chain's writer state (eco_lane.h eco_tile_end)
__global__ void __launch_bounds__(64) k_eco_tile_end(xeve_hip_sbac *__restrict__ states, const xeve_hip_ctu_job *__restrict__ jobs, int nchains, uint8_t *__restrict__ bytes,
                                                     int bytes_cap, int32_t *__restrict__ nbytes)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if(c >= nchains) return;
    const int si = jobs[c].sbac;
    xl::Sbac s = states[si];
    xl::Sink o = {bytes + (long)c * bytes_cap, bytes_cap, 0};
    xl::eco_tile_end(s, &o);
    states[si] = s;
    nbytes[c] = o.n;
}

extern "C" int xeve_hip_eco_tile_end_jobs(xeve_hip_sbac *states, int nstates, const xeve_hip_ctu_job *jobs, int nchains, uint8_t *bytes, int bytes_cap, int32_t *nbytes, void *stream)
{
    XH_ENTER();
    XH_REQUIRE(states && nstates > 0 && jobs && nchains >= 0 && bytes && bytes_cap > 0 && nbytes);
    if(nchains == 0) return XEVE_HIP_OK;
    k_eco_tile_end<<<(nchains + 63) / 64, 64, 0, (hipStream_t)stream>>>(states, jobs, nchains, bytes, bytes_cap, nbytes);
    XH_HIP(hipGetLastError());
    return XEVE_HIP_OK;
}
