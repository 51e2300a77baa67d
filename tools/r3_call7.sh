cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c7
mkdir -p $O
timeout 300 python tools/icache_probe.py > $O/icache.log 2>&1
cat $O/icache.log
