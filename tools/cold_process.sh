cd $GRAFT_REPO_ROOT; S=reef_amd/_lib/seam_bench
$S gen=/tmp/cold.bin n=27790
for rep in 1 2 3; do
  for hm in 0 300; do
    sleep 1   # the previous process's teardown in the kernel driver otherwise lands in this one's first call (+90-150 ms)
    $S cold=/tmp/cold.bin n=27790 warm=0 host_ms=$hm
    sleep 1; $S cold=/tmp/cold.bin n=27790 warm=2 host_ms=$hm
    sleep 1; REEF_MSM_WARM=1 $S cold=/tmp/cold.bin n=27790 warm=0 host_ms=$hm | sed 's/"warm": 0/"warm": "REEF_MSM_WARM=1 at load"/'
  done
done
