// tests/native/cu_lane_host.cpp -- TEST INFRASTRUCTURE: the __host__ side of xeve_amd/csrc/cu_lane.h (the lane-serial intra analysis of a small CU that
// libxeve_hip.so runs on the device), exported so that `pytest -m "not gpu"` can compare it with the oracle on the CPU.  Built by tests/_lane.py with
// `hipcc --cuda-host-only`; never linked into the product library.
#include <hip/hip_runtime.h>
#include "../../xeve_amd/csrc/cu_lane.h"

extern "C" void xl_host_intra_cu(int log2, const xl::Params *P, const int16_t *const org[3], const int16_t *const mod[3], const uint32_t *map_scu, const int8_t *map_ipm,
                                 const uint8_t *map_tidx, const xeve_hip_sbac *entry, const xeve_hip_intra_job *job, xeve_hip_intra_result *res, int16_t *coef_y,
                                 int16_t *coef_u, int16_t *coef_v, int16_t *rec_y, int16_t *rec_u, int16_t *rec_v, xeve_hip_sbac *best)
{
    if(log2 == 2) xl::intra_cu<2>(*P, org, mod, map_scu, map_ipm, map_tidx, *entry, *job, *res, coef_y, coef_u, coef_v, rec_y, rec_u, rec_v, *best);
    else xl::intra_cu<3>(*P, org, mod, map_scu, map_ipm, map_tidx, *entry, *job, *res, coef_y, coef_u, coef_v, rec_y, rec_u, rec_v, *best);
}
extern "C" int xl_host_sizeof_params(void) { return (int)sizeof(xl::Params); }

#include "../../xeve_amd/csrc/eco_lane.h"
// the host side of eco_lane.h: one CTU written on the coder state *s (in / out); returns the number of bytes emitted (the first `cap` stored)
extern "C" int xl_host_eco_ctu(int idc, int slice_type, int log2_ctu, int pic_w, int pic_h, int w_scu, int num_refp0, int num_refp1, const uint16_t *scan16, const uint16_t *scan32,
                               const uint16_t *scan64, xeve_hip_sbac *s, const xeve_hip_ctu_data *d, uint32_t *map_scu, const int8_t *map_ipm, const uint8_t *map_tidx,
                               uint32_t *map_cu_mode, int x0, int y0, uint8_t *bytes, int cap)
{
    xl::EcoParams E;
    E.idc = idc, E.slice_type = slice_type, E.log2_ctu = log2_ctu, E.pic_w = pic_w, E.pic_h = pic_h, E.w_scu = w_scu, E.num_refp[0] = num_refp0, E.num_refp[1] = num_refp1;
    for(int i = 0; i < 7; i++) E.scan[i] = nullptr;
    E.scan[4] = scan16, E.scan[5] = scan32, E.scan[6] = scan64;
    xl::Sink o = {bytes, cap, 0};
    xl::eco_ctu(E, *s, *d, map_scu, map_ipm, map_tidx, map_cu_mode, x0, y0, &o);
    return o.n;
}

extern "C" int xl_host_eco_tile_end(xeve_hip_sbac *s, uint8_t *bytes, int cap)
{
    xl::Sink o = {bytes, cap, 0};
    xl::eco_tile_end(*s, &o);
    return o.n;
}
