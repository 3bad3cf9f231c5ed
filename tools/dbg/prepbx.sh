for wl in cfg3_4k_w15_d128 cfg2b_1080p_w9_d128 cfg5_kitti_w7_d128; do
for bx in 0 64 128 256 320 512 1024; do
 r=$(SADBM_PREP_BX=$bx timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu --workload $wl 2>/dev/null) || { echo "$wl $bx FAILED"; exit 1; }
 echo "$wl BX=$bx $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"], d["roofline"]["prep_ms"])')"
done; done
