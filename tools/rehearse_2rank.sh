# Rehearsal of the N > 1 code path of bench.py on ONE GPU: two ranks over gloo, both on cuda:0 (the driver's real
# run is one rank per GPU over RCCL).  Not a scaling number: it checks that the sharded legs run and agree.
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --same-device --no-cpu-baseline --c4-batch 256 --c4-steps 2 --c3-steps 3 \
  > gpurun_out/bench_rehearsal_2rank.json 2> gpurun_out/bench_rehearsal_2rank.err || { tail -20 gpurun_out/bench_rehearsal_2rank.err; exit 1; }
tail -c 1500 gpurun_out/bench_rehearsal_2rank.json
