set -e
for cfg in "" "2048:2048:4" "2048:2048:4,4096:2048:2" "2048:2048:4,4096:2048:2,2048:6144:4" "2048:2048:4,2048:6144:4" "2048:2048:2,4096:2048:2,2048:6144:4" "2048:2048:4,4096:2048:2,2048:6144:3"; do
  echo "== SPLITK=$cfg"
  SPLITK="$cfg" ROUNDS=2 timeout -k 10 200 python tools/b1_bench.py 2>&1 | grep "fused_epilogues=1 C1\|identical" | head -2
done
