for wl in cfg3_4k_w15_d128 cfg2b_1080p_w9_d128; do for pr in 1 2; do for fl in 16 8; do
 r=$(timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu --workload $wl --pairs $pr --flags $fl 2>/dev/null)
 echo "$wl pairs=$pr flags=$fl $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"], d["config"]["kernel"])')"
done; done; done
