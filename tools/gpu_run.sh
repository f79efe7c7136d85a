cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6r
echo "== paired forward score blocks, bwd occupancy 2" > gpurun_out/r6r/probe.txt
timeout -k 10 120 python tools/attn_balance_probe.py 2>&1 | grep -v amdgpu | sed 's/tokens.*forward/forward/' >> gpurun_out/r6r/probe.txt
echo "== bwd occupancy 3" >> gpurun_out/r6r/probe.txt
MTVAF_LIB=$PWD/mtvaf_amd/lib_as3_occ3/libmtvaf_hip.so timeout -k 10 120 python tools/attn_balance_probe.py 2>&1 | grep -v amdgpu | sed 's/tokens.*forward/forward/' >> gpurun_out/r6r/probe.txt
cat gpurun_out/r6r/probe.txt
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attn or attention" 2>&1 | tail -2
