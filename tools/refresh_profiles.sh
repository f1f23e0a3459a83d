#!/bin/bash
# Runs on the GPU box (via gpurun): the default bench, its rocprofv3 kernel stats, the two PMC passes the roofline's
# `traffic` figure comes from (C2, C3, C4, C5), and one bench line per other workload (C1, C3, C4, C5, C2fb; C1 / C3 also
# as hipGraph replays).  Usage: tools/refresh_profiles.sh <tag>   (e.g. r02a)
# Outputs under gpurun_out/<tag>/ ; copy what should be judged into profiles/ (tools/collect_profiles.py <tag>).
set -u
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export GNNRAG_COMMIT=${GNNRAG_COMMIT:-$(cat $R/.commit_stamp 2>/dev/null || echo unknown)}
cd /tmp && export TMPDIR=/tmp
for W in C2 C3 C4 C5; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$W -o $C -- python $R/tools/prof_ops.py --workload $W --reps 4 --ops agg,aggfd,fr,updfd > $OUT/pmc_${W}_$C.log 2>&1
  done
  ( cd $R && python tools/make_pmc_traffic.py $(find $OUT/pmc_$W -name 'FETCH_SIZE_results.db' | head -1) $(find $OUT/pmc_$W -name 'WRITE_SIZE_results.db' | head -1) $OUT/pmc_traffic_$W.json $W > $OUT/make_pmc_$W.log 2>&1 )
done
cp $OUT/pmc_traffic_C2.json $R/profiles/pmc_traffic.json        # the default bench below reads it (same sources: not stale)
for W in C3 C4 C5; do cp $OUT/pmc_traffic_$W.json $R/profiles/pmc_traffic_$W.json; done   # ... and its other_workloads legs these
cd $R
python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "default bench rc=$?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py --no-cpu-baseline --spread-steps 0 --clock-ramp-ms 0 > $OUT/bench_under_rocprof.log 2>&1
cd $R
python tools/rocpd_stats.py $(find $OUT/trace -name 'bench_results.db' | head -1) > $OUT/kernel_stats_bench.txt 2>&1
for W in C1 C3 C4 C5 C2fb C2u; do
  # BASELINE configs 4 and 5 also get the reference's CPU path beside them (4 questions of the same shape)
  case $W in C4|C5) CPUFLAGS="--cpu-sample-b 4 --no-e2e";; *) CPUFLAGS="--no-cpu-baseline";; esac
  python bench.py --workload $W $CPUFLAGS --steps 30 > $OUT/bench_$W.log 2>&1
done
for W in C1 C3; do
  python bench.py --workload $W --no-cpu-baseline --steps 30 --launch graph > $OUT/bench_${W}_graph.log 2>&1
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_C5 -o bench -- python $R/bench.py --workload C5 --no-cpu-baseline --steps 10 --spread-steps 0 --clock-ramp-ms 0 > $OUT/bench_C5_under_rocprof.log 2>&1
cd $R
python tools/rocpd_stats.py $(find $OUT/trace_C5 -name 'bench_results.db' | head -1) > $OUT/kernel_stats_bench_C5.txt 2>&1
find $OUT -name '*.db' -delete
tail -1 $OUT/bench_default.log | cut -c1-300
head -12 $OUT/kernel_stats_bench.txt
for W in C1 C3 C4 C5 C2fb C2u C1_graph C3_graph; do echo -n "$W: "; tail -1 $OUT/bench_$W.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['roofline']['kernel'][:60], round(d['roofline']['frac'],3))" 2>&1 | tail -1; done
