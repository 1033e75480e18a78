set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dpprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
FAVAE_FORCE_DP=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/s -o s --output-format csv -- python3 $R/bench.py --samples 4096 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/log.txt 2>&1
cd $R
python tools/kstats.py $O/s 30
