cd /root/repo
(cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TA\|TCP\|TCC\|TD\|SQ\)_[A-Za-z0-9_]*" | sort -u > /root/repo/gpurun_out/r6i_counters.txt); wc -l gpurun_out/r6i_counters.txt; grep -i "TA_.*BUSY\|TCP_.*STALL\|TCP_.*LATENCY\|SQ_WAIT_INST\|SQ_INST_CYCLES\|TA_.*CYCLES" gpurun_out/r6i_counters.txt | head -40
PMC_FILTER=x3q PMC_EXTRA="TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" bash tools/dev/pmc_conv.sh r6i_pmc_x3q c256_64_n1 c256_32_n1 2>&1 | grep -v "^/opt" | tail -70
