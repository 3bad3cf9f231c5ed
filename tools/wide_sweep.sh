# rows per block of sad_wide_kernel (0 = the library's choice)
for ty in 0 32 16 8; do echo "SADBM_STRIP_TY=$ty"; SADBM_STRIP_TY=$ty timeout -k 10 100 python tools/wide_time.py 2>&1 | grep -v amdgpu || exit 1; done
