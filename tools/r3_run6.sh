set -e
mkdir -p gpurun_out/r3e
timeout -k 10 400 python -u -m pytest tests/test_gpu_igemm.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 200 python -u tools/bench_gemm.py bf16 > gpurun_out/r3e/gemm.log 2>&1
grep wgrad gpurun_out/r3e/gemm.log
timeout -k 10 200 python -u tools/prof_step.py bf16 16384 4 > gpurun_out/r3e/step.log 2>&1
grep "epoch N" gpurun_out/r3e/step.log | tail -3
