cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/devloop -o dl -- python /root/repo/tools/probe_devloop.py 2 > /root/repo/gpurun_out/devloop/stdout.txt 2>&1
head -12 /root/repo/gpurun_out/devloop/dl_kernel_stats.csv | cut -c1-200
