cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r5}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
B="python3 bench.py --no-cpu --no-train --no-exclusive --no-layers --no-4k --sustain-seconds 0"
rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B --steps 10 --warmup 2 > $O/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- $B --steps 4 --warmup 1 > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o bench -- $B --steps 4 --warmup 1 > $O/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/pmc_sq -o bench -- $B --steps 3 --warmup 1 > $O/bench_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/pmc_clk -o bench -- $B --steps 3 --warmup 1 > $O/bench_clk.log 2>&1
python3 tools/layer_times.py > $O/layer_times.txt 2>&1
tail -2 $O/bench_stats.log | cut -c1-300
python3 tools/pmc_summary.py $O $TAG 5 && cp profiles/${TAG}_pmc_traffic.json $O/
python3 tools/stats_summary.py $O $TAG
python3 tools/pmc_sq_summary.py $O $TAG 4 && cp profiles/${TAG}_pmc_sq.json $O/
grep -v amdgpu.ids $O/layer_times.txt > profiles/${TAG}_layer_times.txt
cp profiles/${TAG}_* $O/ 2>/dev/null
rm -rf $O/stats $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_clk
