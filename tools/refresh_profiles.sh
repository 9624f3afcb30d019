#!/bin/bash
# On the GPU box: regenerate the bench lines and the rocprofv3 kernel summary kept under profiles/ (copy them from
# gpurun_out/refresh/ afterwards).  Each GPU step runs under its own time limit; the script stops at the first failure.
set -e
cd /root/repo
export TMPDIR=/tmp
out=gpurun_out/refresh
mkdir -p $out
timeout -k 10 400 python bench.py > $out/r01_bench.json 2> $out/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python bench.py > $out/r01_bench_under_rocprof.json 2> $out/prof.err
cp $out/prof/*/*_kernel_stats.csv $out/r01_bench_kernel_stats.csv
timeout -k 10 300 python bench.py --nodes 32 --batch 256 --latent 256 --hid 512 > $out/r01_bench_configB.json 2>> $out/bench.err
timeout -k 10 300 python bench.py --nodes 10 --batch 64 --latent 32 --hid 512 > $out/r01_bench_configA.json 2>> $out/bench.err
timeout -k 10 300 python tools/joint_bench.py > $out/r01_joint_bench.json 2>> $out/bench.err
echo refreshed
