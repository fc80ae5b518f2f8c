"""CABAC bit counting on the GPU (xeve_hip_cu_bits_jobs) against the reference goldens and the pinned oracle: bit counts AND
the complete exit coder state (what SBAC_STORE would keep), through the C-ABI."""
import numpy as np
import pytest

from _libs import CU_BITS_JOB_DTYPE, SBAC_DTYPE, oracle_sbac, ptr
from _sbac_cases import clamp_refi, make_jobs, make_params, make_states
from _sbac_golden import golden

pytestmark = pytest.mark.gpu


def run_hip(p, states, jobs, coef, want_state=True):
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    hp = lib.CuBitsParams.from_buffer_copy(bytes(p))
    bits, out = D.cu_bits_jobs(torch.from_numpy(coef.copy()).to(dev), torch.from_numpy(states.view(np.uint8).copy()).to(dev),
                               torch.from_numpy(jobs.view(np.uint8).copy()).to(dev), hp, want_state=want_state)
    torch.cuda.synchronize()
    return bits.cpu().numpy().view(np.uint32), (out.cpu().numpy().reshape(-1).view(SBAC_DTYPE) if out is not None else None)


def test_hip_cu_bits_matches_reference_goldens():
    n = 0
    for p, states, jobs, coef, out, bits in golden():
        got_bits, got = run_hip(p, states, jobs, coef)
        assert np.array_equal(got_bits, bits), (p.log2_cuw, p.log2_cuh)
        assert got.tobytes() == out.tobytes(), (p.log2_cuw, p.log2_cuh)
        fast_bits, _ = run_hip(p, states, jobs, coef, want_state=False)
        assert np.array_equal(fast_bits, bits), (p.log2_cuw, p.log2_cuh)
        n += len(jobs)
    assert n == 133


@pytest.mark.parametrize("slice_type,num_refp,cm_init,idc", [(0, (2, 2), 0, 1), (1, (1, 0), 0, 1), (0, (4, 3), 1, 1), (2, (0, 0), 0, 0),
                                                             (0, (21, 21), 0, 2), (0, (2, 2), 1, 3)])
def test_hip_cu_bits_batches_vs_oracle(slice_type, num_refp, cm_init, idc):
    """both kernels: the one that carries the complete coder state (exit states requested) and the count-only one"""
    O = oracle_sbac()
    r = np.random.default_rng(2100 + slice_type + 10 * cm_init + 100 * idc)
    states = make_states(r, 16)
    for lw in range(2, 7):
        for lh in range(2, 7):
            if (lw + lh + idc) % 2 and lw != lh:
                continue  # thin the non-square sizes
            p = make_params(lw, lh, slice_type, num_refp, cm_init, idc)
            jobs, coef = make_jobs(r, 100 if lw + lh <= 8 else 67, lw, lh, len(states), idc, nnz_mode=(lw + lh) & 1)
            clamp_refi(jobs, num_refp)
            exp_bits, exp = np.zeros(len(jobs), np.uint32), np.zeros(len(jobs), SBAC_DTYPE)
            for i in range(len(jobs)):
                exp_bits[i] = O.xo_cu_bits(ptr(states), ptr(exp[i:i + 1]), p, ptr(jobs[i:i + 1]), ptr(coef))
            got_bits, got = run_hip(p, states, jobs, coef)
            assert np.array_equal(got_bits, exp_bits), (lw, lh, np.flatnonzero(got_bits != exp_bits)[:5])
            assert got.tobytes() == exp.tobytes(), (lw, lh)
            fast_bits, _ = run_hip(p, states, jobs, coef, want_state=False)
            assert np.array_equal(fast_bits, exp_bits), (lw, lh, np.flatnonzero(fast_bits != exp_bits)[:5])


def test_hip_cu_bits_shared_blocks_and_no_state():
    """the call pattern of pinter_residue_rdo: several jobs (all-zero test, as-is, per-component tests) on ONE set of coefficient
    blocks and one entry state; exit states not requested"""
    O = oracle_sbac()
    r = np.random.default_rng(77)
    states = make_states(r, 3)
    p = make_params(4, 4)
    base, coef = make_jobs(r, 40, 4, 4, len(states))
    jobs = np.zeros(0, CU_BITS_JOB_DTYPE)
    for b in base:
        b["mode"] = 0
        true_nnz = b["nnz"].copy()
        variants = []
        for mode, nnz in ((0, (0, 0, 0)), (0, true_nnz), (1, (0, true_nnz[1], true_nnz[2])), (1, true_nnz), (2, true_nnz), (3, (true_nnz[0], true_nnz[1], 0))):
            v = b.copy()
            v["mode"], v["nnz"] = mode, nnz
            variants.append(v)
        jobs = np.concatenate([jobs, np.array(variants, CU_BITS_JOB_DTYPE)])
    got_bits, got = run_hip(p, states, jobs, coef, want_state=False)
    assert got is None
    for i in range(len(jobs)):
        assert got_bits[i] == O.xo_cu_bits(ptr(states), None, p, ptr(jobs[i:i + 1]), ptr(coef)), i


def test_hip_cu_bits_rejects_bad_arguments():
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    z = torch.zeros(64, dtype=torch.int16, device=dev)
    st = torch.zeros(SBAC_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    jb = torch.zeros(44, dtype=torch.uint8, device=dev)
    for lw, lh, idc in ((1, 3, 1), (7, 3, 1), (3, 3, 4)):
        p = lib.CuBitsParams()
        p.log2_cuw, p.log2_cuh, p.chroma_format_idc = lw, lh, idc
        with pytest.raises(RuntimeError):
            D.cu_bits_jobs(z, st, jb, p)


@pytest.mark.parametrize("idc", [1, 0])
def test_hip_eco_coef_alone_vs_oracle(idc):
    """job mode XEVE_HIP_BITS_ECO_COEF: xeve_eco_coef on its own -- inter / intra cbf syntax, subsets of components, implied cbf, and continuing
    the coder in mid-stream (pending / stacked bytes, any code_bits) instead of resetting it"""
    O = oracle_sbac()
    r = np.random.default_rng(4100 + idc)
    for lw, lh in [(2, 2), (3, 3), (4, 4), (5, 5), (6, 6), (3, 5), (6, 4)]:
        p = make_params(lw, lh, 0, (2, 2), 0, idc)
        jobs, coef = make_jobs(r, 90, lw, lh, 12, idc)
        jobs["mode"] = 5
        states = make_states(r, 12)
        states["code_bits"] = r.integers(1, 9, size=len(states))
        states["code"] = r.integers(0, 1 << 17, size=len(states)) << (8 - states["code_bits"]).astype(np.uint32)
        states["is_pending_byte"], states["pending_byte"] = r.integers(0, 2, size=len(states)), r.integers(0, 256, size=len(states))
        states["stacked_ff"], states["stacked_zero"] = r.integers(0, 3, size=len(states)), r.integers(0, 3, size=len(states))
        for i in range(len(jobs)):
            runs = int(r.integers(1, 8)) if idc else 1
            intra, nocbf, noreset = int(r.random() < 0.4), int(r.random() < 0.2), int(r.random() < 0.6)
            if nocbf and not (intra or any(jobs["nnz"][i][c] for c in range(3) if (runs >> c) & 1)):
                nocbf = 0
            jobs["dir_flag"][i] = intra | (nocbf << 1) | (runs << 2) | (noreset << 5)
        exp_bits, exp = np.zeros(len(jobs), np.uint32), np.zeros(len(jobs), SBAC_DTYPE)
        for i in range(len(jobs)):
            exp_bits[i] = O.xo_cu_bits(ptr(states), ptr(exp[i:i + 1]), p, ptr(jobs[i:i + 1]), ptr(coef))
        got_bits, got = run_hip(p, states, jobs, coef)
        assert np.array_equal(got_bits, exp_bits), (lw, lh, np.flatnonzero(got_bits != exp_bits)[:5])
        assert got.tobytes() == exp.tobytes(), (lw, lh)


def test_cu_bits_full_size_properties():
    """3840x2160 i.i.d. picture, the eight rate jobs of every CU of every level (1 376 160 jobs on the quantiser's real output): the count-only
    kernel (with its per-wave choice of loop form), the kernel that carries the complete coder state and the chain variant agree on every
    bit count; the all-zero job never costs more than the as-quantised one."""
    import torch

    import xeve_amd
    from xeve_amd import device as D
    from xeve_amd import lib
    from xeve_amd.workload import HotPathPass

    xeve_amd.init(0)
    dev = torch.device("cuda:0")
    wl = HotPathPass(3840, 2160, dev, seed=4)
    wl.run(only="D")
    fast = {S: b.clone() for S, b in wl.rate().items()}
    torch.cuda.synchronize()
    njobs = 0
    for S in wl.sizes:
        lv, r = wl.lv[S], wl.lv[S]["rate"]
        full_bits, st = D.cu_bits_jobs(lv["coef_flat"], r["state"], r["jobs"], r["params"], want_state=True)
        assert torch.equal(full_bits.view(lv["n"], wl.RATE_JOBS), fast[S]), S
        L = lib.load()
        import ctypes as C
        chain_bits = torch.empty_like(full_bits)
        chain_st = torch.empty_like(st)
        lib.check(L.xeve_hip_cu_bits_jobs_chain(C.c_void_p(lv["coef_flat"].data_ptr()), lv["coef_flat"].numel(), C.c_void_p(r["state"].data_ptr()),
                                                C.c_void_p(r["jobs"].data_ptr()), lv["n"] * wl.RATE_JOBS, C.byref(r["params"]), C.c_void_p(r["ws"].data_ptr()),
                                                r["ws"].numel(), C.c_void_p(chain_bits.data_ptr()), C.c_void_p(chain_st.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        assert torch.equal(chain_bits, full_bits), S
        # range and context models of the chain variant == those of the full state
        a, b = st.cpu().numpy().reshape(-1).view(SBAC_DTYPE), chain_st.cpu().numpy().reshape(-1).view(SBAC_DTYPE)
        assert np.array_equal(a["range"], b["range"]) and np.array_equal(a["ctx"], b["ctx"]), S
        bits = fast[S].cpu().numpy()
        assert np.all(bits[:, 0] <= bits[:, 1]), S
        njobs += bits.size
    assert njobs == 1376160


@pytest.mark.parametrize("slice_type,idc", [(2, 1), (0, 1), (1, 0), (0, 3)])
def test_hip_cu_bits_intra_syntax_vs_oracle(slice_type, idc):
    """job modes 7 / 8 / 9: the intra CU syntax (skip flag + pred_mode outside I slices, the mode index over the two intra_dir models, intra cbf flags)"""
    O = oracle_sbac()
    r = np.random.default_rng(2300 + slice_type + 10 * idc)
    states = make_states(r, 16)
    for lw in range(2, 7):
        p = make_params(lw, lw, slice_type, (2, 2), 0, idc)
        jobs, coef = make_jobs(r, 90, lw, lw, len(states), idc)
        jobs["mode"] = r.integers(7, 10, size=len(jobs))
        jobs["mvp_idx"][:, 0] = r.integers(0, 5, size=len(jobs))
        jobs["nnz"][jobs["mode"] == 8, 1:] = 0
        exp_bits, exp = np.zeros(len(jobs), np.uint32), np.zeros(len(jobs), SBAC_DTYPE)
        for i in range(len(jobs)):
            exp_bits[i] = O.xo_cu_bits(ptr(states), ptr(exp[i:i + 1]), p, ptr(jobs[i:i + 1]), ptr(coef))
        got_bits, got = run_hip(p, states, jobs, coef)
        assert np.array_equal(got_bits, exp_bits), (lw, np.flatnonzero(got_bits != exp_bits)[:5])
        assert got.tobytes() == exp.tobytes(), lw
        fast_bits, _ = run_hip(p, states, jobs, coef, want_state=False)
        assert np.array_equal(fast_bits, exp_bits), lw
