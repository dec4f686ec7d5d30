# Round-3 evidence run (on the GPU box): kernel stats + PMC traffic of the headline bench, then the bench matrix.
# usage: bash tools/r3_collect.sh   (writes under gpurun_out/)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="--no-cpu-baseline --no-prefill"
WRK_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ks -- python3 $R/bench.py $P --steps 40 --warmup 8 > $O/r3_prof_bench.json 2>/dev/null
find $O/prof_ks -name "*kernel_stats.csv" -exec cp {} $O/r03_engine_eager_kernel_stats.csv \; ; rm -rf $O/prof_ks
echo "[r3] stats done"
WRK_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_f -- python3 $R/bench.py $P --steps 16 --warmup 4 > /dev/null 2>&1
WRK_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_w -- python3 $R/bench.py $P --steps 16 --warmup 4 > /dev/null 2>&1
F=$(find $O/prof_f -name "*counter_collection.csv" | head -1); W=$(find $O/prof_w -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py $F $W 20 $O/r03_pmc_traffic.json; rm -rf $O/prof_f $O/prof_w
echo "[r3] pmc done"
cd $R
python bench.py > $O/r03_bench_headline.json 2>/dev/null
echo "[r3] headline done"
WRK_ENGINE=0 python bench.py $P > $O/r03_bench_launches_same_box.json 2>/dev/null
python bench.py --mixed $P > $O/r03_q4km_mixed_bench.json 2>/dev/null
python bench.py --model 2.9B --mixed --batch 32 --steps 64 --warmup 8 $P > $O/r03_cfg3_2p9b_batch32_decode_bench.json 2>/dev/null
python bench.py --model v6-7B $P --steps 64 --warmup 8 > $O/r03_v6_7b_bench.json 2>/dev/null
python bench.py --model v6-7B --batch 16 $P --steps 32 --warmup 4 > $O/r03_v6_7b_batch16_bench.json 2>/dev/null
echo "[r3] model lines done"
bash tools/bench_batched.sh 1.5B 2>/dev/null | sed 's/--no-cpu-baseline/&/' > $O/r03_batched_decode.jsonl
echo "[r3] batched done"
cd /tmp
WRK_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b16 -- python3 $R/bench.py $P --batch 16 --steps 24 --warmup 4 > /dev/null 2>&1
find $O/prof_b16 -name "*kernel_stats.csv" -exec cp {} $O/r03_batch16_eager_kernel_stats.csv \; ; rm -rf $O/prof_b16
cd $R
for cfg in "--batch 32 --prompt 128 --chunk 4096" "--prompt 512 --chunk 128" "--model 2.9B --batch 32 --prompt 128 --chunk 4096" "--model 2.9B --mixed --batch 32 --prompt 128 --chunk 4096" "--model v6-7B --batch 16 --prompt 128 --chunk 2048"; do python tools/prefill_bench.py $cfg 2>/dev/null | tail -1; done > $O/r03_prefill.jsonl
echo "[r3] prefill done"
bash tools/prof_prefill.sh r03c > /dev/null 2>&1
echo done
