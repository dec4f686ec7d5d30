# Round-2 evidence run (on the GPU box): kernel stats + PMC traffic of the headline bench, then the bench matrix.
# usage: bash tools/r2_collect.sh   (writes under gpurun_out/)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WRK_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ks -- python3 $R/bench.py --no-cpu-baseline --steps 40 --warmup 8 > $O/r2_prof_bench.json 2>/dev/null
find $O/prof_ks -name "*kernel_stats.csv" -exec cp {} $O/r02_fused5_eager_kernel_stats.csv \; ; rm -rf $O/prof_ks
WRK_NO_GRAPH=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_f -- python3 $R/bench.py --no-cpu-baseline --steps 16 --warmup 4 > /dev/null 2>&1
WRK_NO_GRAPH=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_w -- python3 $R/bench.py --no-cpu-baseline --steps 16 --warmup 4 > /dev/null 2>&1
F=$(find $O/prof_f -name "*counter_collection.csv" | head -1); W=$(find $O/prof_w -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py $F $W 20 $O/r02_pmc_traffic.json; rm -rf $O/prof_f $O/prof_w
cd $R
python bench.py > $O/r02_bench_headline.json 2>/dev/null
python bench.py --mixed --no-cpu-baseline > $O/r02_q4km_mixed_bench.json 2>/dev/null
python bench.py --model 2.9B --mixed --batch 32 --steps 64 --warmup 8 --no-cpu-baseline > $O/r02_cfg3_2p9b_batch32_decode_bench.json 2>/dev/null
python bench.py --model v6-7B --no-cpu-baseline --steps 64 --warmup 8 > $O/r02_v6_7b_bench.json 2>/dev/null
python bench.py --model v6-7B --batch 16 --no-cpu-baseline --steps 32 --warmup 4 > $O/r02_v6_7b_batch16_bench.json 2>/dev/null
bash tools/bench_batched.sh 1.5B > $O/r02_batched_decode.jsonl 2>/dev/null
bash tools/bench_few.sh 1.5B > $O/r02_few_sequences_dmv_vs_mfma.jsonl 2>/dev/null
python bench.py --mixed --batch 16 --no-cpu-baseline --steps 64 --warmup 8 > $O/r02_q4km_mixed_batch16_bench.json 2>/dev/null
# batch-16 decode: kernel statistics (eager) and the in-kernel timeline of one layer (instrumented build, if present)
cd /tmp
WRK_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b16 -- python3 $R/bench.py --no-cpu-baseline --batch 16 --steps 24 --warmup 4 > /dev/null 2>&1
find $O/prof_b16 -name "*kernel_stats.csv" -exec cp {} $O/r02_batch16_eager_kernel_stats.csv \; ; rm -rf $O/prof_b16
cd $R
if [ -f web-rwkv-gguf_amd/lib_timing/libwrk_hip.so ]; then
  for B in 1 2 16 32; do WRK_LIB_DIR=$R/web-rwkv-gguf_amd/lib_timing WRK_TIMING=1 python bench.py --batch $B --no-cpu-baseline --steps 32 --warmup 4 2>&1 >/dev/null | grep -A5 WRK_TIMING | tail -6 | sed "s/^/[batch $B] /"; done > $O/r02_decode_layer_timeline_batches.txt
fi
for cfg in "--batch 32 --prompt 128 --chunk 4096" "--prompt 512 --chunk 128" "--model 2.9B --batch 32 --prompt 512 --chunk 4096" "--model 2.9B --mixed --batch 32 --prompt 128 --chunk 4096" "--model v6-7B --batch 16 --prompt 128 --chunk 2048"; do python tools/prefill_bench.py $cfg 2>/dev/null | tail -1; done > $O/r02_prefill.jsonl
echo done
