for wl in cfg3_4k_w15_d128 cfg2b_1080p_w9_d128 cfg5_kitti_w7_d128; do
for so in opencv_amd/csrc/libsadbm*.so; do
 r=$(SADBM_LIB=$PWD/$so timeout -k 10 120 python3 bench.py --steps 20 --warmup 3 --no-cpu --workload $wl 2>/dev/null) || { echo "$wl $so FAILED"; exit 1; }
 echo "$wl $(basename $so) $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"], d["roofline"]["prep_ms"])')"
done; done
