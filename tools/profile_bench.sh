cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-train --no-exclusive --no-layers --no-4k --sustain-seconds 0 > $O/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-train --no-exclusive --no-layers --no-4k --sustain-seconds 0 > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-train --no-exclusive --no-layers --no-4k --sustain-seconds 0 > $O/bench_write.log 2>&1
# training step: replayed steps only (every launch of the trace belongs to a hipGraph replay or to the three capture warm-ups), and
# a separate trace of eager steps (two streams)
export UNCL_BENCH_WGRAD=0   # (the stand-alone weight-gradient timing of the train line would put its own launches into the traces)
rocprofv3 --kernel-trace --stats -d $O/train -o bench -- python3 bench.py --mode train --no-eager --steps 30 --warmup 3 > $O/train_stats.log 2>&1
UNCL_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats -d $O/train_eager -o bench -- python3 bench.py --mode train --steps 20 --warmup 3 > $O/train_eager_stats.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/train_video -o bench -- python3 bench.py --mode train_video --no-eager --steps 10 --warmup 2 > $O/train_video_stats.log 2>&1
# issue counters of every kernel of the bench step (one pass: five SQ counters fit the eight slots), per-layer table
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/pmc_sq -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-train --no-exclusive --no-layers --no-4k --sustain-seconds 0 > $O/bench_sq.log 2>&1
# effective clock per kernel (GRBM_GUI_ACTIVE / 8 / launch time), its own pass
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_clk -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-train --no-exclusive --no-layers --no-4k --sustain-seconds 0 > $O/bench_clk.log 2>&1
python3 tools/layer_times.py > $O/layer_times.txt 2>&1
unset UNCL_BENCH_WGRAD
python3 bench.py > $O/bench_plain.log 2>&1
tail -2 $O/bench_stats.log | cut -c1-300
python3 tools/pmc_summary.py $O $TAG 5 && cp profiles/${TAG}_pmc_traffic.json $O/
python3 tools/stats_summary.py $O $TAG
python3 tools/pmc_sq_summary.py $O $TAG 4 && cp profiles/${TAG}_pmc_sq.json $O/
grep -v amdgpu.ids $O/layer_times.txt > profiles/${TAG}_layer_times.txt
# what travels back: the summaries (copy them from gpurun_out/<tag>/ into profiles/ and commit); the raw databases exceed the
# size gpurun merges back, and a partial merge is worse than none
cp profiles/${TAG}_* $O/ 2>/dev/null
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_clk $O/train $O/train_eager $O/train_video
