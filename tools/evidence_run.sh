#!/bin/bash
# end-of-round evidence: GPU test log, default bench line, trace + counters of the default (dense) command and of the slot-resident frame
O=gpurun_out/evidence; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt | head -2
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt
bash tools/gpu_profile.sh evidence_dense > $O/prof_dense.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_evidence_dense $O/dense_pmc.txt $O/dense_traffic.json > /dev/null  # (add "_workload": "8192 d1" / "8192 d1 slots" / "8192 modular chain" when copying into profiles/)
PROFILE_CMD="python $GRAFT_REPO_ROOT/tools/slots_resident.py" bash tools/gpu_profile.sh evidence_slots > $O/prof_slots.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_evidence_slots $O/slots_pmc.txt $O/slots_traffic.json > /dev/null
cp gpurun_out/prof_evidence_dense/trace/t_kernel_stats.csv $O/dense_kernel_stats.csv
cp gpurun_out/prof_evidence_slots/trace/t_kernel_stats.csv $O/slots_kernel_stats.csv
grep -A9 "^derived" $O/slots_pmc.txt | cut -c1-200
PROFILE_CMD="python $GRAFT_REPO_ROOT/tools/chain_flow_time.py 8192 5" bash tools/gpu_profile.sh evidence_modular > $O/prof_modular.log 2>&1
python tools/pmc_summary.py gpurun_out/prof_evidence_modular $O/modular_pmc.txt $O/modular_traffic.json > /dev/null
cp gpurun_out/prof_evidence_modular/trace/t_kernel_stats.csv $O/modular_kernel_stats.csv
