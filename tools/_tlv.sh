cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tlv1
mkdir -p $O
cd $R
export UNCL_BENCH_WGRAD=0
rocprofv3 --kernel-trace -d $O/tl -o bench -- python3 bench.py --mode train_video --no-eager --steps 6 --warmup 2 > $O/tl.log 2>&1
python3 tools/timeline.py $O/tl 2 pack_weight_batch > $O/video_timeline.txt
rm -rf $O/tl
head -2 $O/video_timeline.txt
