cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 8 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['roofline']['achieved'])"; }
run --opt gemm.wide=7
run --opt gemm.wide=5
run --opt gemm.wide=6
run --opt gemm.wide=7
run --opt gemm.wide=5
