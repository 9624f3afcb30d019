#!/bin/bash
# One rocprofv3 counter pass over eager steps of the bench; prints the mean counters of the kernels whose name
# contains $1.  usage: tools/pmc_step.sh <kernel substring> <counter> [<counter> ...]
set -e
repo=$(pwd)
flt=$1; shift
out=$repo/gpurun_out/pmc_step
rm -rf $out && mkdir -p $out
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/p -- python3 $repo/bench.py --eager --steps 3 --warmup 1 --profile-steps 0 --no-extra --no-cpu-baseline --no-mst --no-alt > $out/log.txt 2>&1
python3 $repo/tools/pmc_summary.py $out/p "$flt"
