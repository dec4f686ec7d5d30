# kernel statistics of the RWKV-6 7B 16 x 128-token prefill (eager launches): bash tools/prof_v6_prefill.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
TAG=${1:-r03}
WRK_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_v -- python3 $R/tools/prefill_bench.py --model v6-7B --batch 16 --prompt 128 --chunk 2048 --repeat 1 > /dev/null 2>&1
F=$(find $O/prof_v -name "*kernel_stats.csv" | head -1)
cp $F $O/${TAG}_v6_prefill_kernel_stats.csv
rm -rf $O/prof_v
head -14 $O/${TAG}_v6_prefill_kernel_stats.csv | cut -c1-170
