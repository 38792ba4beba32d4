#!/bin/bash
# Runs ON THE GPU BOX (gpurun): rocprofv3 kernel trace + the two PMC passes of bench.py, for the default (four steps per pass), --fuse-depth 3, --fuse-depth 2
# (two steps per pass) and --fuse 0 (one step per launch); results as rocpd databases under gpurun_out/prof_<tag>/.  Condense afterwards with
#   python tools/summarize_profile.py <tag>_march 4096x4096_float32_march gpurun_out/prof_<tag>/march_trace/*.db ...
# usage: bash tools/profile_bench.sh <tag> [extra bench.py args]
set -e
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for mode in march4 march3 march nofuse; do
    fuse="--fuse -1"; [ $mode = nofuse ] && fuse="--fuse 0"; [ $mode = march ] && fuse="--fuse-depth 2"; [ $mode = march3 ] && fuse="--fuse-depth 3"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/${mode}_trace -o t -- python3 bench.py $fuse --pmc-traffic 0 --cpu-steps 0 "$@" > $out/${mode}_trace.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $out/${mode}_fetch -o f -- python3 bench.py $fuse --pmc-traffic 0 --cpu-steps 0 --steps 48 --warmup 12 "$@" > $out/${mode}_fetch.log 2>&1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $out/${mode}_write -o w -- python3 bench.py $fuse --pmc-traffic 0 --cpu-steps 0 --steps 48 --warmup 12 "$@" > $out/${mode}_write.log 2>&1
    echo "$mode done"
done
find $out -name "*.db" -size +60M -delete      # the merge-back limit is 64 MiB
ls -la $out/*/
