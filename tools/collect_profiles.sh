# Copies one evidence run (gpurun_out/<tag>/, written by tools/round_run.sh) into profiles/<tag>_* (tracked):
#   bash tools/collect_profiles.sh r04
set -e
TAG=${1:-r04}
S=gpurun_out/$TAG
P=profiles
cp $S/bench.json $P/${TAG}_bench.json
cp $S/bench_profiled.json $P/${TAG}_bench_profiled.json
cp $S/bench_rehearse2.json $P/${TAG}_bench_rehearse2_one_gpu.json
cp $S/kernel_stats.csv $P/${TAG}_kernel_stats.csv
cp $S/pmc_summary.csv $P/${TAG}_pmc_summary.csv
cp $S/pmc_traffic.csv $P/${TAG}_pmc_traffic.csv
cp $S/c3_bench.json $P/${TAG}_c3_bench.json
cp $S/c4.txt $P/${TAG}_c4_bench.txt
cp $S/conv_ab.txt $P/${TAG}_conv_ab.txt
cp $S/train_bench.txt $P/${TAG}_train_bench.txt
cp $S/train_bench_bf16.txt $P/${TAG}_train_bench_bf16.txt
cp $S/train_kernel_stats.csv $P/${TAG}_train_kernel_stats.csv
cp $S/step_host.txt $P/${TAG}_step_host.txt
cp $S/k1_time.txt $P/${TAG}_k1_time.txt
cp $S/corr_time.txt $P/${TAG}_corr_time.txt
cp $S/ldsdma_handover.txt $P/${TAG}_ldsdma_handover.txt
[ -f gpurun_out/rccl_r04.json ] && cp gpurun_out/rccl_r04.json $P/${TAG}_rccl_record.json || true
tail -3 $S/pytest_gpu.txt > $P/${TAG}_pytest_gpu_tail.txt
python3 tools/traffic_json.py $S/pmc_traffic.csv $P/traffic.json ${TAG}_pmc_traffic.csv --keep-missing
echo collected
