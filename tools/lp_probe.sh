for cfg in "X=1" "WRK_GEMM_TILE3=0" "WRK_WKV_OCT=0" "WRK_T3_SPLIT=0"; do
  echo "== $cfg"; env $cfg timeout -k 10 300 python -m pytest tests/test_gpu_layer_parity.py -m gpu -x -q -k "k_split_tile and lens1" 2>&1 | grep -E "AssertionError: 1.5B|passed|failed" | head -3
done
