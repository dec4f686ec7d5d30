# round 3, coverage lines: bash tools/r3_cov_bench.sh > gpurun_out/r3_cov.jsonl
run() { # label, env..., -- bench args
  label="$1"; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-prefill --steps 48 --warmup 6 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'case':'$label','ms_per_step':d['ms_per_step'],'tokens_per_s':d['value'],'frac':d['roofline']['frac'],'workload':d['config']['workload']}))"
}
run "v6-14B-3L Q8_0 x16, default routing" X=1 -- --model v6-14B-3L --batch 16
run "v6-14B-3L Q8_0 x16, K-sliced GEMM off" WRK_GEMM_KS=0 -- --model v6-14B-3L --batch 16
run "v6-14B-3L Q8_0 x16, every eligible launch K-sliced" WRK_GEMM_KS=2 -- --model v6-14B-3L --batch 16
run "v6-14B-3L Q8_0 x32, default routing (every eligible launch K-sliced)" X=1 -- --model v6-14B-3L --batch 32
run "v6-14B-3L Q8_0 x32, K-sliced GEMM off" WRK_GEMM_KS=0 -- --model v6-14B-3L --batch 32
run "2.9B Q4_K_M mix x2, dmv three-kind launch (default)" X=1 -- --model 2.9B --mixed --batch 2
run "2.9B Q4_K_M mix x2, MFMA path (WRK_DMV_TOKENS=1)" WRK_DMV_TOKENS=1 -- --model 2.9B --mixed --batch 2
run "2.9B Q4_K_M mix x4, dmv three-kind launch (default)" X=1 -- --model 2.9B --mixed --batch 4
run "2.9B Q4_K_M mix x4, MFMA path (WRK_DMV_TOKENS=1)" WRK_DMV_TOKENS=1 -- --model 2.9B --mixed --batch 4
run "2.9B Q4_K_M mix x1" X=1 -- --model 2.9B --mixed --batch 1
