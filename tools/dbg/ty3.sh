# band-height scan per workload: tools/dbg/ty3.sh  (SADBM_STRIP_TY=0 is the model's choice)
set -u
scan() { wl=$1; shift
 for ty in "$@"; do
  r=$(SADBM_STRIP_TY=$ty timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu --workload $wl 2>/dev/null) || { echo "$wl $ty FAILED"; exit 1; }
  echo "$wl TY=$ty $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"])')"
 done; }
scan cfg2b_1080p_w9_d128 0 34 45 54 60 67 77 90 108 134 179
scan cfg2_1080p_w9_d64 0 34 45 54 67 77 90 108 134 179 268
scan cfg5_kitti_w7_d128 0 23 31 37 46 62 92 123 184
scan cfg3_4k_w15_d128 0 54 67 72 83 98 108 127 134 154 179 215
scan cfg4_8k_w21_d256 0 67 86 108 135 144 166 180 216 270
