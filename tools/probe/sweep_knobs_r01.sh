# (historical: the knobs this sweep drove - P3_TN_BLOCKS, P3_GEMM_BK, P3_LN_RPB - were removed in r04 once their A/Bs were decided)
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fwd 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for r in 1 2; do
echo -n "base: "; run
for v in 640 768 1024 1280; do echo -n "TN_BLOCKS=$v: "; P3_TN_BLOCKS=$v run; done
for v in 1024 512; do echo -n "GEMM_BK=$v: "; P3_GEMM_BK=$v run; done
for v in 32 64 96; do echo -n "LN_RPB=$v: "; P3_LN_RPB=$v run; done
done
