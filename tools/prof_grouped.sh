set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_g8
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/tools/prof_step.py bf16 4096 1 8 > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/tools/prof_step.py bf16 4096 1 8 > $O/write.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/tl -o t --output-format csv -- python3 $R/tools/prof_step.py bf16 4096 1 8 > $O/tl.log 2>&1
cd $R
python tools/pmc_traffic.py $O/fetch $O/write $O/pmc_traffic.json > $O/pmc_traffic.txt
python tools/ktimeline.py $O/tl > $O/step_timeline.txt || true
rm -rf $O/fetch $O/write $O/tl
tail -3 $O/pmc_traffic.txt
