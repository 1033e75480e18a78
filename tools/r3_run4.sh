set -e
mkdir -p gpurun_out/r3b
timeout -k 10 300 python -u -m pytest tests/test_gpu_igemm.py -x -q -k "folded" --timeout 120 --timeout-method thread 2>&1 | tail -2
timeout -k 10 200 python -u tools/prof_step.py bf16 16384 4 > gpurun_out/r3b/step.log 2>&1
grep "epoch N" gpurun_out/r3b/step.log | tail -3
