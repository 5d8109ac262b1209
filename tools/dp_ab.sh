cd $GRAFT_REPO_ROOT
for mode in "TCCT_DP_OVERLAP=0" "TCCT_DP_OVERLAP=1 TCCT_DP_MARKS=0" "TCCT_DP_OVERLAP=1"; do
  env $mode TCCT_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29731 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['config']['step_ms_gpu_min_med_max'], d['config']['grad_allreduce'][:40])"
done
