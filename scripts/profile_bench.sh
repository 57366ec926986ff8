#!/bin/bash
# Round profile set (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the bench, of the OT extension
# and of the phase-1 kernels, SQ counters of the MAC kernels, per-op launch profiles, the big factorisations.  Counter passes
# are --pmc only (no tracing alongside) and the program follows `--` directly.  Outputs under gpurun_out/prof;
# scripts/summarize_profiles.py <tag> copies the summaries into profiles/.  gpurun MERGES into the local gpurun_out/: delete
# the local gpurun_out/prof first.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LGC_BENCH_DETAIL_DIR=$O   # bench.py: the long record (bench_detail.json) beside the one-line stdout
python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
export LGC_BENCH_DETAIL_DIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-e2e --no-sweep > $O/bench_stats_run.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --child > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --child > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq1 -- python3 $R/scripts/gpu_probe.py big > /dev/null 2> $O/pmc_sq1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -- python3 $R/scripts/gpu_probe.py big > /dev/null 2> $O/pmc_sq2.err
# OT extension and phase 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ot -- python3 $R/scripts/gpu_ot_probe.py > $O/ot_probe.txt 2> $O/stats_ot.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p1 -- python3 $R/scripts/gpu_phase1_probe.py > $O/phase1_probe.txt 2> $O/stats_p1.err
cd $R
python3 scripts/gpu_launch_profile.py 500 cgd 15 > $O/launch_profile_d500_cgd15.txt 2>&1
python3 scripts/gpu_launch_profile.py 100 cgd 15 > $O/launch_profile_d100_cgd15.txt 2>&1
python3 scripts/gpu_launch_profile.py 20 cholesky 0 > $O/launch_profile_d20_cholesky.txt 2>&1
python3 scripts/gpu_launch_profile.py 500 cgd 20 32 > $O/launch_profile_d500_cgd20_w32.txt 2>&1
python3 scripts/gpu_probe.py mid big > $O/probe.txt 2>&1
{ python3 scripts/gpu_headline.py 500 0 1 cholesky; python3 scripts/gpu_headline.py 500 0 1 ldlt; python3 scripts/gpu_headline.py 500 15 3; python3 scripts/gpu_headline.py 100 15 5; python3 scripts/gpu_headline.py 20 0 5 cholesky; } 2>&1 | grep "per solve" > $O/big_factorisations.txt
python3 tests/tools/gpu_phase1_baseline.py > $O/phase1_baseline.jsonl 2>&1
cat $O/bench_line.json | cut -c1-600
echo done
