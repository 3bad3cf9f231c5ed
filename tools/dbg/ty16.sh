for wl in cfg3_4k_w15_d128 cfg2_1080p_w9_d64; do
for ty in 0 48 64 96 127 160 200 254 360; do
 r=$(SADBM_STRIP_TY=$ty timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu --workload $wl 2>/dev/null) || { echo "$wl $ty FAILED"; continue; }
 echo "$wl TY=$ty $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"])')"
done; done
