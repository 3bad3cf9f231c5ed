for wl in cfg3_4k_w15_d128 cfg4_8k_w21_d256 cfg5_kitti_w7_d128; do
for ty in 48 64 86 100 127 160 200 253 300; do
 r=$(SADBM_STRIP_TY=$ty timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu --workload $wl --flags 8 2>/dev/null) || { echo "$wl $ty FAILED"; continue; }
 echo "$wl TY=$ty $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms"])')"
done; done
