mkdir -p gpurun_out/r05
for c in c3 c4; do
  for v in "split -" "wide -" "wide ties"; do
    set -- $v
    GSPLAT_PAIR_SORT=$1 GSPLAT_ROUNDS=off timeout 250 python tools/stripe_kernels.py $c 8 3 $2 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r05/stripe_kernels_ab1.txt 2>&1
cat gpurun_out/r05/stripe_kernels_ab1.txt
for c in c3 c4; do
  for v in "split cull" "wide cull" "wide cull+ties"; do
    set -- $v
    echo "== $c $v"
    STRIPE_MODEL_G=8 GSPLAT_PAIR_SORT=$1 GSPLAT_ROUNDS=off timeout 300 python tools/stripe_model.py $c $2 2>&1 | grep -v amdgpu.ids
  done
done > gpurun_out/r05/stripe_model_ab1.txt 2>&1
cat gpurun_out/r05/stripe_model_ab1.txt
