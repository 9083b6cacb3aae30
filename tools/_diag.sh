for d in 0 8 0 8 1; do
  echo -n "diag $d: "
  ZRK_EXCHANGE_DIAG=$d ZRK_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python bench.py --steps 300 --warmup 50 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" || true
done
