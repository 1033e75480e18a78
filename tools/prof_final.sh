# Everything under profiles/ that describes the final build of a round, in one GPU call (developer tool; run from the repo root on the GPU
# box).  Results under gpurun_out/prof_final/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
mkdir -p $O
bash $R/tools/prof_round.sh > $O/prof_round.log 2>&1
cd $R
timeout -k 10 300 python -u tools/bench_grouped.py 16384 bf16,f32 1,2,4,8,16 > $O/grouped_sweep.txt 2>/dev/null
timeout -k 10 120 python -u tools/bench_grouped.py 65536 bf16 32 >> $O/grouped_sweep.txt 2>/dev/null
bash $R/tools/prof_grouped.sh > $O/prof_grouped.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/g8 -o g8 -- python3 $R/tools/prof_grouped.py 8 8192 bf16 > $O/g8.log 2>&1
cd $R
python tools/kdb.py $(ls $O/g8/*.db | head -1) all 44 > $O/grouped_g8_kernel_summary.txt
rm -rf $O/g8
for f in 2 9; do echo "== 512 images, favae_debug_force_tile(-1, $f)  (2: igemm_pm 64x64, 9: wgrad9_kernel)"; B=512 FORCE_PM=$f ONLY=wgrad timeout -k 10 120 python tools/bench_gemm.py 2>/dev/null | grep -E "enc.[12] |dec.[234] "; done > $O/wgrad_all_taps.txt
for f in 2 9; do echo "== 64 images, favae_debug_force_tile(-1, $f)"; B=64 FORCE_PM=$f ONLY=wgrad timeout -k 10 120 python tools/bench_gemm.py 2>/dev/null | grep -E "enc.[12] |dec.[234] "; done >> $O/wgrad_all_taps.txt
timeout -k 10 300 python bench.py --samples 32768 --steps 1 --warmup 1 --no-cpu-baseline --groups 0 > $O/bench_n32768_single.json 2>/dev/null
FAVAE_FORCE_DP=1 timeout -k 10 300 python bench.py --samples 32768 --steps 1 --warmup 1 --no-cpu-baseline --groups 0 > $O/bench_n32768_dp.json 2>/dev/null
tail -2 $O/grouped_sweep.txt; cat $O/wgrad_all_taps.txt | head -8
