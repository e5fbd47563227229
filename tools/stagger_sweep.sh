# forward product of the LM head under several start staggers (ticks of 10 ns per K-tile; 0 = off)
for s in 0 15000; do echo "stagger $s"; PRL_LM_HEAD_STAGGER=$s timeout -k 10 120 python -u tools/lm_head_bench.py --rounds 3 --skip-exact 2>&1 | grep "^fwd: ours" || exit 1; done
