set -x
for cfg in "WRK_PRO_RPW=8" "WRK_PRO_RPW=12" "WRK_PRO_RPW=16" "WRK_PRO_RPW=24" "WRK_PRO_RPW=32" "WRK_WG_PER_CU=2" "WRK_WG_PER_CU=3" "WRK_WG_PER_CU=6" "WRK_WG_PER_CU=8" "WRK_PRO_RPW=8 WRK_WG_PER_CU=8" "WRK_DMV=0"; do
  echo "== $cfg"; env $cfg python bench.py --no-cpu-baseline --steps 128 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
echo "== 0.1B"; python bench.py --no-cpu-baseline --model 0.1B --steps 128 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline'])"
echo "== 2.9B"; python bench.py --no-cpu-baseline --model 2.9B --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
echo "== mixed"; python bench.py --no-cpu-baseline --mixed --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
