set -e
mkdir -p gpurun_out/r3d
timeout -k 10 600 python -u -m pytest tests/test_gpu_sshiba.py tests/test_gpu_fa_kernels.py tests/test_gpu_trainer_flows.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3d/sshiba.log 2>&1 || { tail -40 gpurun_out/r3d/sshiba.log | cut -c1-220; exit 1; }
tail -3 gpurun_out/r3d/sshiba.log
timeout -k 10 200 python -u tools/prof_fa.py > gpurun_out/r3d/fa.log 2>&1
tail -8 gpurun_out/r3d/fa.log
