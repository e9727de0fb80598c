cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r5t}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export UNCL_BENCH_WGRAD=0
rocprofv3 --kernel-trace --stats -d $O/train -o bench -- python3 bench.py --mode train --no-eager --steps 30 --warmup 3 > $O/train_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/train_video -o bench -- python3 bench.py --mode train_video --no-eager --steps 10 --warmup 2 > $O/train_video_stats.log 2>&1
python3 tools/stats_summary.py $O $TAG
cp profiles/${TAG}_* $O/ 2>/dev/null
rm -rf $O/train $O/train_video
