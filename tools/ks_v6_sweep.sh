run() { label="$1"; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --no-prefill --steps 48 --warmup 6 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'])"; }
for M in v6-7B-3L v6-14B-3L; do
run "$M x16 KS=0" WRK_GEMM_KS=0 -- --model $M --batch 16
run "$M x16 KS=1(default)" X=1 -- --model $M --batch 16
run "$M x16 KS=2 bps1" WRK_GEMM_KS=2 WRK_KS_BPS=1 -- --model $M --batch 16
run "$M x16 KS=2 bps2" WRK_GEMM_KS=2 WRK_KS_BPS=2 -- --model $M --batch 16
run "$M x16 KS=2 bps4" WRK_GEMM_KS=2 WRK_KS_BPS=4 -- --model $M --batch 16
run "$M x8 KS=0" WRK_GEMM_KS=0 -- --model $M --batch 8
run "$M x8 KS=1(default)" X=1 -- --model $M --batch 8
done
