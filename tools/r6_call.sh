cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06h
mkdir -p $O
echo "== e2e d200 batch 64: tail in line (0) / in the tail process (1)"
for i in 1 2 3; do
bash tools/e2e_once.sh d200 64 GNNRAG_EVAL_PIPELINE=0 | tail -2
bash tools/e2e_once.sh d200 64 GNNRAG_EVAL_PIPELINE=1 | tail -2
done 2>&1 | tee $O/e2e_ab.txt
echo "== batch 16, C1" | tee -a $O/e2e_ab.txt
(bash tools/e2e_once.sh d200 16 GNNRAG_EVAL_PIPELINE=0 | tail -2
bash tools/e2e_once.sh d200 16 GNNRAG_EVAL_PIPELINE=1 | tail -2
bash tools/e2e_once.sh d50 1 GNNRAG_EVAL_PIPELINE=0 | tail -2
bash tools/e2e_once.sh d50 1 GNNRAG_EVAL_PIPELINE=1 | tail -2) 2>&1 | tee -a $O/e2e_ab.txt
echo "== main.py parity (tail process on by default)"
timeout 900 python -m pytest tests/test_gpu_main_py.py -m gpu -q -k "d50 or d200 or cwq-single" 2>&1 | tail -15
