set -e
mkdir -p gpurun_out/r3a
timeout -k 10 300 python -u tools/dbg_parity_r3.py > gpurun_out/r3a/parity.log 2>&1
timeout -k 10 300 python -u -m pytest tests/test_gpu_igemm.py -x -q -k "folded" --timeout 120 --timeout-method thread > gpurun_out/r3a/igemm.log 2>&1
timeout -k 10 500 python -u -m pytest tests/test_gpu_trainer_flows.py -x -v --timeout 200 --timeout-method thread > gpurun_out/r3a/flows.log 2>&1
