#!/bin/bash
# round 6, final evidence pass on one box: the whole -m gpu suite, the traffic counters (stamped with the commit), the bench line, the
# rocprofv3 --kernel-trace --stats summary of the same bench command, the per-launch tables of the stage-1 step
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/r6_g_tests.log 2>&1; grep -E "passed|failed" $O/r6_g_tests.log | tail -2
bash tools/refresh_traffic.sh ea3c2d9 > $O/r6_g_traffic.log 2>&1; tail -3 $O/r6_g_traffic.log
timeout 900 python bench.py > $O/r6_g_bench.json 2> $O/r6_g_bench.err; tail -c 600 $O/r6_g_bench.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hl -o t -- python $R/bench.py --steps 20 --warmup 3 --headline-only > $O/r6_g_headline_prof.log 2>&1)
f=$(find /tmp/hl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r6_headline_kernel_stats.csv && head -5 $f | cut -c1-200
