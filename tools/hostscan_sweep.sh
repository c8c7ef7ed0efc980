for c in 22 23 24 25; do echo "chunk 2^$c"; CLO_SCAN_PIPE_CHUNK_LOG2=$c timeout -k 10 200 python tools/hostscan_probe.py 2>&1 | grep -E "2\^2[68].*two queues" || exit 1; done
echo adaptive; timeout -k 10 200 python tools/hostscan_probe.py 2>&1 | grep -E "two queues"
benchmarks/bin/clo_hip_scan_bench -t uint -y uint -i 4194304 -n 7 -r 3 | grep MValues
