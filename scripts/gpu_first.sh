set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.txt 2>&1 || true
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1 || { tail -60 gpurun_out/pytest_gpu.log; exit 1; }
tail -5 gpurun_out/pytest_gpu.log
timeout -k 10 300 ./cuda_selection_criteria_amd/bin/time_smh_hip -N 10000 -m 512 -h 0.8 -R 2 > gpurun_out/time_smh.log 2>&1 || { tail -20 gpurun_out/time_smh.log; exit 1; }
cat gpurun_out/time_smh.log
