#!/bin/bash
# GPU box: evidence run of the round's final build: tools/final_profile.sh, the profiler's statistics of the timed leg alone
# (tools/clean_stats.sh) and the PMC passes of the default workload.  usage: tools/evidence.sh <tag>
T=${1:-evidence}
tools/final_profile.sh $T > gpurun_out/${T}_final.log 2>&1; echo "final_profile rc $?"; tail -12 gpurun_out/${T}_final.log
tools/clean_stats.sh ${T}_clean > gpurun_out/${T}_clean.log 2>&1; echo "clean stats rc $?"; tail -4 gpurun_out/${T}_clean.log
tools/pmc_passes.sh ${T}_pmc > gpurun_out/${T}_pmc.log 2>&1; echo "pmc rc $?"
python3 tools/pmc_traffic.py gpurun_out/${T}_pmc gpurun_out/${T}_pmc_traffic.json 14 130603885 > gpurun_out/${T}_pmc_traffic.txt 2>&1; tail -3 gpurun_out/${T}_pmc_traffic.txt
python3 tools/pmc_valu.py gpurun_out/${T}_pmc gpurun_out/${T}_pmc_valu.json 14 > gpurun_out/${T}_pmc_valu.txt 2>&1; tail -2 gpurun_out/${T}_pmc_valu.txt
