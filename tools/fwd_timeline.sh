# run on the GPU box: kernel timeline of one forward step of the bench (two streams) -> gpurun_out/<tag>/fwd_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-tlf}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace -d $O/tl_fwd -o bench -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-train --no-exclusive --no-layers --no-4k --sustain-seconds 0 > $O/tl_fwd.log 2>&1
python3 tools/timeline.py $O/tl_fwd 2 > $O/fwd_timeline.txt
rm -rf $O/tl_fwd
