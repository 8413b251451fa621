# window backward: experiment library against the product library -- bits, tests, launch time against the number of problems
out=gpurun_out/windbg; mkdir -p $out; rm -f $out/log.txt
python3 tools/win_compare.py $out/a.pt >> $out/log.txt 2>&1 && VPU_ATTN_ONEPASS=3 VPU_LIB_FILE=libvpu_hip_x.so timeout -k 10 60 python3 tools/win_compare.py $out/b.pt >> $out/log.txt 2>&1 && python3 tools/win_compare.py cmp $out/a.pt $out/b.pt >> $out/log.txt 2>&1
echo "compare rc $?" >> $out/log.txt
rm -f $out/a.pt $out/b.pt
for l in "libvpu_hip.so 1" "libvpu_hip_x.so 1" "libvpu_hip_x.so 3" "libvpu_hip.so 1" "libvpu_hip_x.so 3"; do
  set -- $l
  echo "== $1 onepass $2" >> $out/log.txt
  VPU_LIB_FILE=$1 VPU_ATTN_ONEPASS=$2 timeout -k 10 120 python3 tools/attn_bwd_scale.py >> $out/log.txt 2>&1 || break
done
grep -v amdgpu.ids $out/log.txt
