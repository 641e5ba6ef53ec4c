#!/bin/bash
# GPU box: evidence run of round 4 (tools/final_profile.sh) followed by the PMC passes of the default workload
T=${1:-r04z}
tools/final_profile.sh $T > gpurun_out/${T}_final.log 2>&1; echo "final_profile rc $?"; tail -12 gpurun_out/${T}_final.log
tools/pmc_passes.sh ${T}_pmc > gpurun_out/${T}_pmc.log 2>&1; echo "pmc rc $?"
python3 tools/pmc_traffic.py gpurun_out/${T}_pmc gpurun_out/${T}_pmc_traffic.json 7 261207771 > gpurun_out/${T}_pmc_traffic.txt 2>&1; tail -5 gpurun_out/${T}_pmc_traffic.txt
python3 tools/pmc_valu.py gpurun_out/${T}_pmc gpurun_out/${T}_pmc_valu.json 7 > gpurun_out/${T}_pmc_valu.txt 2>&1; tail -3 gpurun_out/${T}_pmc_valu.txt
