cd $GRAFT_REPO_ROOT; F=gpurun_out/lines; mkdir -p $F
for c in c3 c3d c4 c1 c2 c5; do timeout 170 python bench.py --config $c > $F/r02_bench_$c.json 2> $F/$c.err; done
ls -la $F | head -20
