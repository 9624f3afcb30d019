#!/bin/bash
# On the GPU box: regenerate the bench lines, the rocprofv3 kernel summary and the counter passes kept under
# profiles/ (copy them from gpurun_out/refresh/ afterwards).  usage: tools/refresh_profiles.sh r03
# Each GPU step runs under its own time limit; the script stops at the first failure.  Counter passes run alone
# (one counter group per run, kernel trace only), the profiled program directly after `--`.
set -e
tag=${1:-r03}
repo=$(pwd)
export TMPDIR=/tmp
out=$repo/gpurun_out/refresh
mkdir -p $out
cd /tmp
timeout -k 10 500 python3 $repo/bench.py > $out/${tag}_bench.json 2> $out/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python3 $repo/bench.py --no-extra --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/prof.err
cp $out/prof/*/*_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_gemm_$n -- python3 $repo/tools/gemm_one.py rows_planes 5 > $out/pmc_gemm_$n.log 2>&1
  python3 $repo/tools/pmc_summary.py $out/pmc_gemm_$n gemm_kernel > $out/${tag}_pmc_${n}_gemm128_planes.txt
  cp $out/pmc_gemm_$n/*/*counter_collection.csv $out/${tag}_pmc_${n}_gemm128_planes.csv
done
# the default (f16x2) step's two large GEMMs, from eager steps of the bench itself
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_step_$n -- python3 $repo/bench.py --eager --steps 3 --warmup 1 --profile-steps 0 --no-extra --no-cpu-baseline --no-mst --no-alt > $out/pmc_step_$n.log 2>&1
  python3 $repo/tools/pmc_summary.py $out/pmc_step_$n "gemm_kernel<128, 128, 2, 2, 1, true, true, " > $out/${tag}_pmc_${n}_step_gemm128.txt
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_mst_$c -- python3 $repo/tools/mst_bench.py 3 > $out/pmc_mst_$c.log 2>&1
  python3 $repo/tools/pmc_summary.py $out/pmc_mst_$c mst_kernel > $out/${tag}_pmc_${c}_mst_triu.txt
  cp $out/pmc_mst_$c/*/*counter_collection.csv $out/${tag}_pmc_${c}_mst_triu.csv
done
timeout -k 10 300 python3 $repo/tools/joint_bench.py > $out/${tag}_joint_bench.json 2>> $out/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/jprof -- python3 $repo/tools/joint_bench.py 5 trainer > $out/jprof.log 2>&1
cp $out/jprof/*/*_kernel_stats.csv $out/${tag}_joint_kernel_stats.csv
timeout -k 10 200 python3 $repo/tools/lstm_bench.py > $out/${tag}_lstm_bench.txt 2>> $out/bench.err
timeout -k 10 200 python3 $repo/tools/planes_bench.py > $out/${tag}_planes_bench.txt 2>> $out/bench.err
timeout -k 10 200 python3 $repo/tools/proj_ce_bench.py > $out/${tag}_proj_ce_bench.json 2>> $out/bench.err
timeout -k 10 300 python3 $repo/tools/f16x2_bench.py > $out/${tag}_f16x2_bench.txt 2>> $out/bench.err
timeout -k 10 200 python3 $repo/tools/text_ops_bench.py > $out/${tag}_text_ops_bench.json 2>> $out/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tprof -- python3 $repo/tools/text_ops_bench.py > $out/tprof.log 2>&1
cp $out/tprof/*/*_kernel_stats.csv $out/${tag}_text_ops_kernel_stats.csv
cd $repo
timeout -k 10 200 python3 $repo/tools/mst_sizes.py > $out/${tag}_mst_sizes.txt 2>> $out/bench.err
timeout -k 10 300 python3 $repo/tools/pipeline_bench.py > $out/${tag}_pipeline_bench.txt 2>> $out/bench.err
echo refreshed
