set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for seed in 1 2 3 4; do
IRS_RANDOM_SHAPES=80 IRS_RANDOM_SHAPES_SEED=$seed timeout -k 10 600 python -m pytest tests/test_gpu_scoring.py -m gpu -q -x -k random_shapes > gpurun_out/rand_$seed.log 2>&1 || { tail -40 gpurun_out/rand_$seed.log; exit 1; }; tail -1 gpurun_out/rand_$seed.log
done
