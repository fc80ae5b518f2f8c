cd $GRAFT_REPO_ROOT
timeout 75 python -m pytest tests/test_enc_gpu.py tests/test_hip_tree.py -x -q -m gpu -k "tiny_ldb_fast or tiny_ra_medium or moving_ra_medium or moving_ldb_ref3 or gops_128x64_noise or tree" 2>&1 | tail -2
