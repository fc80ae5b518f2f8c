// xeve_amd/csrc/walk_inter.h -- xeve_pinter_analyze_cu (src_base/xeve_pinter.c:1839-2047) of the node every chain of the team stands at, as team stages (walk.h).
#pragma once
namespace xw {
template <bool FULL> XW void inter_node(const Tm &tm, const P &p, Lds &S, int c0, int nC, int L)
{
    (void)tm, (void)p, (void)S, (void)c0, (void)nC, (void)L;
}
} // namespace xw
