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
