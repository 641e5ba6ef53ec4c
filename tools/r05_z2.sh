#!/bin/bash
T=r05z
tools/clean_stats.sh ${T}_clean > gpurun_out/${T}_clean.log 2>&1; echo "clean stats rc $?"; tail -6 gpurun_out/${T}_clean.log
tools/pmc_passes.sh ${T}_pmc > gpurun_out/${T}_pmc.log 2>&1; echo "pmc rc $?"
python3 tools/pmc_traffic.py gpurun_out/${T}_pmc gpurun_out/${T}_pmc_traffic.json 7 261207771 > gpurun_out/${T}_pmc_traffic.txt 2>&1; tail -3 gpurun_out/${T}_pmc_traffic.txt
python3 tools/pmc_valu.py gpurun_out/${T}_pmc gpurun_out/${T}_pmc_valu.json 7 > gpurun_out/${T}_pmc_valu.txt 2>&1; tail -2 gpurun_out/${T}_pmc_valu.txt
A=$PWD/hvqm4_amd/abl
tools/ab.sh ${T}_poolnt "dense natural" 2 base poolnt:HVQM4_AMD_LIB=$A/libhvq_poolnt.so
tools/pmc_quick.sh ${T}_natural --preset natural > gpurun_out/${T}_natural.log 2>&1; tail -12 gpurun_out/${T}_natural.log | cut -c1-900
