bash tools/r03_evidence.sh gpurun_out/r03k tests cov pmc stats batch_oihw_storage > gpurun_out/r03k_log.txt 2>&1
cp gpurun_out/r03k/pmc/traffic.json profiles/traffic.json && bash tools/r03_evidence.sh gpurun_out/r03k bench >> gpurun_out/r03k_log.txt 2>&1
python3 __graft_entry__.py smoke >> gpurun_out/r03k_log.txt 2>&1
bash tools/fin_probe.sh > gpurun_out/r03k/finprobe.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete
grep -v "^# lq::" gpurun_out/r03k_log.txt | grep -n "passed\|coverage\|UNLAUNCHED\|csrc_sha\|unlaunched\|smoke"
python3 -c "
import json; d=json.load(open('gpurun_out/r03k/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][-40:])"
