# Rehearsal of the N > 1 code path of bench.py on ONE GPU: four ranks over gloo, all on cuda:0 (the driver's real run is one
# rank per GPU over RCCL).  Four is what the box allows: at most 6 processes may hold the card, and C5 needs a divisor of 32.
# Not a scaling number: it checks that the sharded legs (C3, C4, C5 split-decode) run and carry `verified: true`.
# usage: bash tools/rehearse_4rank.sh [weak|strong]
set -o pipefail
cd $GRAFT_REPO_ROOT
S=${1:-weak}
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29519 \
  bench.py --gpus 4 --steps 3 --warmup 1 --backend gloo --same-device --no-cpu-baseline --scaling $S --batch $([ $S = strong ] && echo 4096 || echo 1024) \
  --c4-batch $([ $S = strong ] && echo 512 || echo 128) --c4-steps 2 --c3-batch $([ $S = strong ] && echo 1024 || echo 256) --c3-steps 3 --extras-timeout 700 \
  > gpurun_out/bench_rehearsal_4rank_$S.json 2> gpurun_out/bench_rehearsal_4rank_$S.err || { tail -20 gpurun_out/bench_rehearsal_4rank_$S.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("gpurun_out/bench_rehearsal_4rank_$S.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "scaling", "verified")})
for leg in ("c3_1M_items", "c4_item_sharded", "scale_metric"):
    print(leg, {k: d[leg].get(k) for k in ("value", "ms_per_step", "verified", "users_per_gpu", "items_per_gpu")})
print("c5", d["c4_item_sharded"].get("c5_beam32"))
assert d["verified"] and d["c3_1M_items"]["verified"] and d["c4_item_sharded"]["verified"], "a leg is not verified"
PY
