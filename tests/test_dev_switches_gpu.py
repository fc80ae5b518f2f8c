"""The library's developer switches (environment variables read once per process: xeve_amd/csrc/xh_common.h lists them) select other kernels or launch structures for the
same arithmetic -- each is a configuration of a bit-exact product, so each runs a reference-bitstream case here: a fresh interpreter per setting encodes
two small clips (B pictures, 2 row chains; a batch of closed GOPs with 8 row chains) and must reproduce the reference application's bytes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

SETTINGS = {
    "search_rows_per_lane": {"XEVE_HIP_ME_CPL": "0"},
    "search_lds_window": {"XEVE_HIP_ME_CPL": "0", "XEVE_HIP_ME_LDS": "1"},
    "transforms_on_the_valu_path": {"XEVE_HIP_DCT": "valu"},
    "writer_on_a_lone_lane": {"XEVE_HIP_WRITER_WAVE": "0"},
    "one_ctu_store": {"XEVE_HIP_ENC_TWO_STORES": "0"},
    "stores_in_memory_of_their_own": {"XEVE_HIP_ENC_SHARE": "0"},
    "rdo_rounds_speculated_at_any_width": {"XEVE_HIP_RDO_SPEC": "100000000"},
    "rdo_rounds_never_speculated": {"XEVE_HIP_RDO_SPEC": "0"},
    "walk_from_a_graph": {"XEVE_HIP_TREE_SIDE": "0", "XEVE_HIP_TREE_GRAPH": "1"},
    "complete_states_and_priorities": {"XEVE_HIP_ENC_FULL_STATES": "1", "XEVE_HIP_ENC_PRIO": "1"},
}


@pytest.mark.parametrize("name", sorted(SETTINGS))
def test_a_reference_bitstream_case_under_every_developer_switch(name):
    env = {k: v for k, v in os.environ.items() if not k.startswith("XEVE_HIP_")}
    env.update(SETTINGS[name])
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_enc_gpu.py"), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "(test_single_runs_on_the_gpu_reproduce and tiny_ra_medium) or (test_batches_of_closed_gops_on_the_gpu and gops_128x64_noise)"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-1500:] + p.stderr[-500:]
    assert int(p.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1]) >= 2, p.stdout[-400:]
