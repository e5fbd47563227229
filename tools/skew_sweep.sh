# forward / dX / dW LM-head products with the start skew of the persistent kernel at 0 .. 200 % of an item's time
set -e
for p in 0 50 100 200; do
  echo "== skew $p %"
  PRL_LM_HEAD_SKEW=$p timeout -k 10 200 python -u tools/lm_head_bench.py --rounds 5 --skip-exact "$@" 2>&1 | grep -E "ours"
done
