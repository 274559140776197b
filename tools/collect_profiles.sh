#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run from the repo root through gpurun):
#   kernel stats of the bench command, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes), kernel stats of the
#   shape probes.  Summaries land in gpurun_out/prof_r03/*.{csv,json}; copy what is to be judged into profiles/r03/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bench_stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-shapes > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
python3 $R/tools/pmc_summary.py stats $O/bench_tar64_kernel_stats.csv $O/bench_stats > $O/bench_stats_top.txt
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "mrz_.*" -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-shapes --no-verify > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "mrz_.*" -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-shapes --no-verify > $O/pmc_write.json 2> $O/pmc_write.err
python3 $R/tools/pmc_summary.py pmc $O/pmc_hbm_tar64.json FETCH_SIZE=$O/pmc_fetch WRITE_SIZE=$O/pmc_write > /dev/null
rocprofv3 --kernel-trace --stats -d $O/shape_stats -- python3 $R/tools/probe.py text100 noise64 noise1g > $O/probe_shapes.jsonl 2> $O/probe_shapes.err
python3 $R/tools/pmc_summary.py stats $O/shapes_kernel_stats.csv $O/shape_stats > $O/shape_stats_top.txt
rm -rf $O/bench_stats $O/pmc_fetch $O/pmc_write $O/shape_stats
ls -la $O
