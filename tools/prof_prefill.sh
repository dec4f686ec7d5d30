# kernel statistics of one 32 x 128-token prefill chunk (eager launches: rocprofv3 does not trace hipGraph replays on this stack)
# usage: bash tools/prof_prefill.sh <tag>   -> gpurun_out/<tag>_prefill_kernel_stats.csv
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
TAG=${1:-r03}
WRK_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_p -- python3 $R/tools/prefill_bench.py --batch 32 --prompt 128 --chunk 4096 --repeat 2 > $O/${TAG}_prefill_prof_bench.json 2>/dev/null
F=$(find $O/prof_p -name "*kernel_stats.csv" | head -1)
cp $F $O/${TAG}_prefill_kernel_stats.csv
rm -rf $O/prof_p
head -14 $O/${TAG}_prefill_kernel_stats.csv | cut -c1-200
