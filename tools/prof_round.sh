# The profile set of a round (developer tool; run on the GPU box from the repo root): kernel stats of a short bench run, the
# timeline of one training step, HBM-side traffic per kernel from two separate counter passes.  Results under gpurun_out/prof_round/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_round
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --samples 16384 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/tl -o t --output-format csv -- python3 $R/tools/prof_step.py bf16 1024 1 > $O/tl.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/tools/prof_step.py bf16 512 1 > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/tools/prof_step.py bf16 512 1 > $O/write.log 2>&1
cd $R
python tools/kstats.py $O/stats 40 > $O/kernel_stats_summary.txt
python tools/ktimeline.py $O/tl > $O/step_timeline.txt
python tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json > $O/pmc_traffic.txt
tail -3 $O/pmc_traffic.txt; head -12 $O/kernel_stats_summary.txt
