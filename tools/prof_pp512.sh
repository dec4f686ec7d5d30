# kernel statistics of pp512 (one sequence, 128-token chunks; eager launches): bash tools/prof_pp512.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
TAG=${1:-r03}
WRK_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q -- python3 $R/tools/prefill_bench.py --batch 1 --prompt 512 --chunk 128 --repeat 2 > $O/${TAG}_pp512_prof_bench.json 2>/dev/null
F=$(find $O/prof_q -name "*kernel_stats.csv" | head -1)
cp $F $O/${TAG}_pp512_kernel_stats.csv
rm -rf $O/prof_q
head -16 $O/${TAG}_pp512_kernel_stats.csv | cut -c1-180
