bash tools/gpu/r05c.sh
bash tools/gpu/r05b.sh
