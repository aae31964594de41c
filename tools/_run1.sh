set -e
mkdir -p gpurun_out
SGPR_PROBE_TILES=1 python tools/probe_gemm.py 15360 15360 256 1 15360 15360 1024 1 15360 15360 2048 1 8192 8192 1024 0 16384 16384 1024 0 > gpurun_out/tiles.txt 2>&1
SGPR_PROBE_BETA=0 SGPR_PROBE_TILES=1 python tools/probe_gemm.py 15360 15360 1024 1 >> gpurun_out/tiles.txt 2>&1
