# run on the GPU box: kernel timeline of one replayed image training step (tools/timeline.py) -> gpurun_out/<tag>/train_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-tl}
MODE=${2:-train}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export UNCL_BENCH_WGRAD=0
rocprofv3 --kernel-trace -d $O/tl_$MODE -o bench -- python3 bench.py --mode $MODE --no-eager --steps 6 --warmup 2 > $O/tl_$MODE.log 2>&1
python3 tools/timeline.py $O/tl_$MODE 2 conv3x3_pc_kernelIDF16bLi1ELi3ELi3E > $O/${MODE}_timeline.txt
rm -rf $O/tl_$MODE
