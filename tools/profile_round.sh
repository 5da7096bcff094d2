#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: run on the GPU box from the repo root.
# usage: tools/profile_round.sh <tag> <bench.py arguments...>
set -e
tag=$1; shift
# (a heartbeat: the GPU pool takes a command that writes nothing for seven minutes to be hung)
(while sleep 60; do echo -n "."; done) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A="--steps 1 --warmup 0 --cpu-games 0 --no-variants --no-unshared --recycle-games 0 --no-trained --no-configs $*"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof/prof_${tag}_stats -o stats -- python3 bench.py $A > gpurun_out/prof_${tag}_stats.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof/prof_${tag}_fetch -o fetch -- python3 bench.py $A > gpurun_out/prof_${tag}_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof/prof_${tag}_write -o write -- python3 bench.py $A > gpurun_out/prof_${tag}_write.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d /tmp/prof/prof_${tag}_sq -o sq -- python3 bench.py $A > gpurun_out/prof_${tag}_sq.log 2>&1
mkdir -p gpurun_out/profiles_out
python3 tools/make_profiles.py ${tag} "rocprofv3 --kernel-trace {--stats | --pmc <group>} -- python3 bench.py $A" /tmp/prof gpurun_out/profiles_out > gpurun_out/profiles_out/${tag}.log 2>&1
tail -1 gpurun_out/prof_${tag}_stats.log | cut -c1-400
