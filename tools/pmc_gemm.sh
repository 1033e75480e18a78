# SQ counters of single implicit-GEMM launches in isolation (developer tool): where the wave cycles of one layer's kernel go.
# usage: bash tools/pmc_gemm.sh "enc.1 fwd" "enc.4 wgrad" ...   (results under gpurun_out/pmc_gemm/<name>/)
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm/$(echo "$n" | tr ' .' '__')
  mkdir -p "$d"
  ONLY="$n" timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
      -d "$d" -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_gemm.py bf16 > "$d/run.log" 2>&1 || { echo "pass failed for $n"; tail -5 "$d/run.log"; exit 1; }
done
