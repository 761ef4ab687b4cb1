export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1
for cfg in "COMM=rccl HALO=given" "COMM=torch HALO=given" "COMM=rccl HALO=given" "COMM=torch HALO=given"; do
  env $cfg MASTER_PORT=29551 timeout 300 python tools/shard_host_probe.py 2>/dev/null | grep '^{'
done
