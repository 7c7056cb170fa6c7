# Round-6 artefacts on one box (headline precision = fp32x3): rocprofv3 kernel stats of the train step, FETCH / WRITE PMC passes (separate runs), THEN the default
# bench line (so that its roofline.traffic comes from this round's counters), SQ counters, the bf16-mode stats, the FFL line, the one-rank RCCL line.
# usage (through gpurun): bash tools/final_prof_r06.sh
set -x
R=r06
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -o st -- python bench.py --lean --steps 10 > gpurun_out/final/stats_run.log 2>&1
find /tmp/pf_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/${R}_fp32x3_step_kernel_stats.csv \;
python tools/kstats.py gpurun_out/final/${R}_fp32x3_step_kernel_stats.csv 15 60 > gpurun_out/final/${R}_fp32x3_step_summary.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_fetch -o f -- python bench.py --lean --graph 0 --steps 4 --warmup 2 > gpurun_out/final/fetch_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pf_write -o w -- python bench.py --lean --graph 0 --steps 4 --warmup 2 > gpurun_out/final/write_run.log 2>&1
python tools/pmc_traffic.py /tmp/pf_fetch /tmp/pf_write gpurun_out/final/${R}_pmc_traffic_fp32x3.json 6 > gpurun_out/final/${R}_pmc_summary_fp32x3.txt 2>&1
cp gpurun_out/final/${R}_pmc_traffic_fp32x3.json profiles/${R}_pmc_traffic_fp32x3.json
python bench.py 2>&1 | tail -1 > gpurun_out/final/${R}_bench.json
cp gpurun_out/final/${R}_bench.json profiles/${R}_bench.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pf_sq -o q -- python bench.py --lean --graph 0 --steps 3 --warmup 2 > gpurun_out/final/sq_run.log 2>&1
python tools/pmc_kernels.py /tmp/pf_sq > gpurun_out/final/${R}_sq_counters_summary_fp32x3.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_bf16 -o stb -- python bench.py --lean --precision bf16 --steps 20 > gpurun_out/final/bf16_run.log 2>&1
find /tmp/pf_bf16 -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/${R}_bf16_step_kernel_stats.csv \;
python tools/kstats.py gpurun_out/final/${R}_bf16_step_kernel_stats.csv 25 45 > gpurun_out/final/${R}_bf16_step_summary.txt
python bench.py --workload ffl_fusion --steps 10 --no-cpu-baseline --no-fp32-leg --no-predict --no-host-feed 2>&1 | tail -1 > gpurun_out/final/${R}_bench_ffl.json
P3_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline --no-fp32-leg --no-predict --no-host-feed --no-ffl --no-fwd 2>&1 | tail -1 > gpurun_out/final/${R}_bench_rccl_single_rank.json
tail -3 gpurun_out/final/${R}_pmc_summary_fp32x3.txt
head -12 gpurun_out/final/${R}_fp32x3_step_summary.txt
cut -c1-700 gpurun_out/final/${R}_bench.json
cut -c1-300 gpurun_out/final/${R}_bench_ffl.json
