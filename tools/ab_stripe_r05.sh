# A/B of the stripe-rank levers of round 5 on one GPU (tools/stripe_kernels.py, tools/stripe_model.py): usage
#   bash tools/ab_stripe_r05.sh <tag>
T=${1:-ab}; mkdir -p gpurun_out/r05
for c in c3 c4; do
  for v in "split - kernel" "wide ties auto"; do
    set -- $v
    GSPLAT_EMIT_SUMS=$3 GSPLAT_PAIR_SORT=$1 GSPLAT_ROUNDS=off timeout 250 python tools/stripe_kernels.py $c 8 3 $2 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r05/stripe_kernels_$T.txt 2>&1
cat gpurun_out/r05/stripe_kernels_$T.txt
for c in c3 c4; do
  for v in "split cull kernel" "auto cull+ties auto"; do
    set -- $v
    echo "== $c $v"
    GSPLAT_EMIT_SUMS=$3 STRIPE_MODEL_G=8 GSPLAT_PAIR_SORT=$1 GSPLAT_ROUNDS=off timeout 300 python tools/stripe_model.py $c $2 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r05/stripe_model_$T.txt 2>&1
cat gpurun_out/r05/stripe_model_$T.txt
