for wl in cfg3_4k_w15_d128 cfg4_8k_w21_d256 cfg2_1080p_w9_d64; do
for pr in 1 2 4 8 16; do
 r=$(timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu --workload $wl --pairs $pr 2>/dev/null) || { echo "$wl $pr FAILED"; continue; }
 echo "$wl pairs=$pr $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["kernel"])')"
done; done
