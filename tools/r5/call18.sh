#!/bin/bash
# round 5, GPU call 18: softmax + next layer's prior pairs in one launch - whole GPU suite, A/B against the two-launch
# form, then (only if the suite is green) the PMC traffic of the new sources, the default bench and its kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export GNNRAG_COMMIT=$(cat .commit_stamp 2>/dev/null || echo unknown)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05m; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; PRC=$?; echo "pytest rc=$PRC"
grep "passed\|failed" $O/pytest_gpu.log
for i in 1 2; do
  for SP in 0 1; do
    GNNRAG_SOFTMAX_PAIRS=$SP timeout 300 python bench.py --no-cpu-baseline --no-e2e --spread-steps 0 --fp32-steps 0 > $O/ab_${SP}_$i.json 2> $O/ab_${SP}_$i.err
    echo -n "GNNRAG_SOFTMAX_PAIRS=$SP run $i: "; tail -1 $O/ab_${SP}_$i.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
  done
done | tee $O/ab_softmax_pairs.txt
if [ $PRC -ne 0 ]; then echo "suite not green: stopping here"; exit 1; fi
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_C2 -o $C -- python $R/tools/prof_ops.py --workload C2 --reps 4 --ops agg,aggfd,fr,updfd > $O/pmc_C2_$C.log 2>&1
done
( cd $R && python tools/make_pmc_traffic.py $(find $O/pmc_C2 -name 'FETCH_SIZE_results.db' | head -1) $(find $O/pmc_C2 -name 'WRITE_SIZE_results.db' | head -1) $O/pmc_traffic_C2.json C2 > $O/make_pmc_C2.log 2>&1 )
cp $O/pmc_traffic_C2.json $R/profiles/pmc_traffic.json
cd $R
timeout 900 python bench.py > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"
tail -1 $O/bench_default.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],4), r['frac'], r['traffic'], r.get('traffic_source','')[:120])
for k,v in d['e2e'].items():
    if isinstance(v,dict):
        for leg,x in v.items():
            if isinstance(x,dict): print(k,leg,round(x['questions_per_s'],1),{a:round(b,2) for a,b in x['stages_ms_per_batch'].items()})"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-e2e --spread-steps 0 --clock-ramp-ms 0 > $O/bench_under_rocprof.log 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/trace -name 'bench_results.db' | head -1) > $O/kernel_stats_bench.txt 2>&1
find $O -name '*.db' -delete
grep -n "k_softmax_pairs\|k_fact_prior_merged\|k_masked_softmax\|k_walk_slice" $O/kernel_stats_bench.txt | cut -c1-160
