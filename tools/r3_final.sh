# final lines of round 3: bash tools/r3_final.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python bench.py > $O/r03_bench_headline.json 2>/dev/null; echo "[r3] headline"
for cfg in "--batch 32 --prompt 128 --chunk 4096" "--prompt 512 --chunk 128" "--model 2.9B --batch 32 --prompt 128 --chunk 4096" "--model 2.9B --prompt 512 --chunk 128" "--model 2.9B --mixed --batch 32 --prompt 128 --chunk 4096" "--model v6-7B --batch 16 --prompt 128 --chunk 2048"; do python tools/prefill_bench.py $cfg 2>/dev/null | tail -1; done > $O/r03_prefill.jsonl
echo "[r3] prefill"
bash tools/prof_prefill.sh r03d > /dev/null 2>&1
bash tools/prof_pp512.sh r03d > /dev/null 2>&1
echo done
