for wl in cfg2b_1080p_w9_d128 cfg5_kitti_w7_d128; do
for ty in 0 34 45 54 67 77 90 108 134 179 268; do
 r=$(SADBM_STRIP_TY=$ty timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu --workload $wl 2>/dev/null) || { echo "$wl $ty FAILED"; continue; }
 echo "$wl TY=$ty $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"])')"
done; done
