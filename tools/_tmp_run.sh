timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "Rebuild thread|Multi thread" | tail -30
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/sortp -o s -- python /root/repo/tools/time_mapinc.py > /root/repo/gpurun_out/sortp/stdout.txt 2>&1
grep -E "k_sort|k_pack|k_gather|copyBuffer|k_scan" /root/repo/gpurun_out/sortp/s_kernel_stats.csv | cut -c1-150
tail -5 /root/repo/gpurun_out/sortp/stdout.txt
