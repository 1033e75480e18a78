set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/hp -o s --output-format csv -- python3 $R/tools/prof_step.py bf16 4096 2 > $O/hp.log 2>&1
FAVAE_BF16_EXACT_ENDS=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/plain -o s --output-format csv -- python3 $R/tools/prof_step.py bf16 4096 2 > $O/plain.log 2>&1
cd $R
python tools/kstats.py $O/hp 60 > $O/hp_summary.txt
python tools/kstats.py $O/plain 60 > $O/plain_summary.txt
