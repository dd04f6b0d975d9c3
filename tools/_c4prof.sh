cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf /tmp/irs_prof/c4
rocprofv3 --kernel-trace --stats -d /tmp/irs_prof/c4 -- python3 bench.py --workload c4 --batch 1024 --steps 5 --warmup 2 --no-c3 --no-c4 --no-scoring --no-latency --no-cpu-baseline > gpurun_out/c4prof.log 2>&1 || { tail -5 gpurun_out/c4prof.log; exit 1; }
python3 tools/rocpd_kernels.py $(ls /tmp/irs_prof/c4/*/*.db | head -1) k_path_step > gpurun_out/c4_b1024_kernels_r05.txt 2>&1; head -24 gpurun_out/c4_b1024_kernels_r05.txt
