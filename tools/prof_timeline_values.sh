set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl_vals
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export FAVAE_SYNC_VALUES=1
timeout -k 10 120 rocprofv3 --kernel-trace -d $O/tl -o t --output-format csv -- python3 $R/tools/prof_step.py bf16 1024 1 > $O/tl.log 2>&1
cd $R
python tools/ktimeline.py $O/tl > $O/step_timeline.txt
rm -rf $O/tl
tail -3 $O/tl.log
