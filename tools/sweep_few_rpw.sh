for B in 2 4 8; do for cfg in "16 4" "32 4" "32 2" "32 1"; do set -- $cfg
r=$(WRK_PRO_RPW=$1 WRK_WG_PER_CU=$2 python bench.py --no-cpu-baseline --batch $B --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
echo "B=$B PRO_RPW=$1 WG_PER_CU=$2 ms=$r"; done; done
