# the LM head's three products, ours against the library's, at the shapes of the recipes
# (rows T of a step's joined row or of a micro-batch, hidden size H, vocabulary V)
set -e
mkdir -p gpurun_out/lm_head_shapes
for shape in "2048 1024 151936" "4096 2048 151936" "16384 2048 151936" "4096 3584 152064" "16384 3584 152064" "16384 5120 152064" "8192 1024 151936" "12288 1024 151936"; do
  set -- $shape
  echo "== T=$1 H=$2 V=$3"
  timeout -k 10 280 python -u tools/lm_head_bench.py --t $1 --h $2 --v $3 --rounds 5 --skip-exact \
      --json gpurun_out/lm_head_shapes/t$1_h$2_v$3.json 2>&1 | grep -E "ours|Error|error"
done
