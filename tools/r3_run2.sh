set -e
mkdir -p gpurun_out/r3b
timeout -k 10 100 python -u tools/fail_inject.py dp > gpurun_out/r3b/fail_dp.log 2>&1 || { echo "fail_inject dp rc=$?"; tail -5 gpurun_out/r3b/fail_dp.log; exit 1; }
tail -3 gpurun_out/r3b/fail_dp.log
timeout -k 10 200 python -u tools/prof_step.py bf16 16384 4 > gpurun_out/r3b/step.log 2>&1
grep "ms/step" gpurun_out/r3b/step.log | tail -4
FAVAE_BF16_EXACT_ENDS=0 timeout -k 10 200 python -u tools/prof_step.py bf16 16384 4 > gpurun_out/r3b/step_plain.log 2>&1
grep "epoch N" gpurun_out/r3b/step_plain.log | tail -3
