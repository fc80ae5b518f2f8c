# round 5, call 5: the options that reached the device this round -- P slices, chroma qp offsets (both walks), preset slow (fused walk: walk_dbk.h) -- before the final suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_enc_gpu.py -k "p_slices or preset_slow or refused or single_runs" --durations=12 > gpurun_out/r05f_tests.log 2>&1
tail -25 gpurun_out/r05f_tests.log
