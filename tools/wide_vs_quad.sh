# windows 26-49: the quad matcher (default) beside sad_wide_kernel (SADBM_WIDE_FROM=26)
J="1920,1080,1,27,64 1920,1080,1,33,64 1920,1080,1,41,64 1920,1080,1,49,64 1920,1080,8,33,64 1920,1080,8,41,64 1920,1080,8,49,64 1920,1080,8,49,128 3840,2160,4,41,256 640,480,1,35,30"
echo default; timeout -k 10 200 python tools/job_times.py $J 2>&1 | grep -v amdgpu || exit 1
echo SADBM_WIDE_FROM=26; SADBM_WIDE_FROM=26 timeout -k 10 200 python tools/job_times.py $J 2>&1 | grep -v amdgpu || exit 1
