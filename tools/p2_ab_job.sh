# p2cl_up: experiment library against the product library -- bits and launch times
out=gpurun_out/p2_ab; mkdir -p $out; rm -f $out/log.txt
python3 tools/p2_compare.py $out/a.pt >> $out/log.txt 2>&1 && VPU_LIB_FILE=libvpu_hip_x.so timeout -k 10 60 python3 tools/p2_compare.py $out/b.pt >> $out/log.txt 2>&1 && python3 tools/p2_compare.py cmp $out/a.pt $out/b.pt >> $out/log.txt 2>&1
echo "compare rc $?" >> $out/log.txt
rm -f $out/a.pt $out/b.pt
for l in libvpu_hip.so libvpu_hip_x.so libvpu_hip.so libvpu_hip_x.so; do
  echo "== $l" >> $out/log.txt
  VPU_LIB_FILE=$l timeout -k 10 120 python3 tools/op_bench.py p2cl_up >> $out/log.txt 2>&1 || break
done
grep -v amdgpu.ids $out/log.txt
