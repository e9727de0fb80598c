cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-train --no-exclusive > $O/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-train --no-exclusive > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-train --no-exclusive > $O/bench_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/train -o bench -- python3 bench.py --mode train --steps 10 --warmup 2 > $O/train_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/train_video -o bench -- python3 bench.py --mode train_video --steps 6 --warmup 2 > $O/train_video_stats.log 2>&1
python3 bench.py > $O/bench_plain.log 2>&1
tail -2 $O/bench_stats.log | cut -c1-300
python3 tools/pmc_summary.py $O $TAG 5 && cp profiles/${TAG}_pmc_traffic.json $O/
python3 tools/stats_summary.py $O $TAG
