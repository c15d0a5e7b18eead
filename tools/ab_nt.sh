for round in 1 2; do
for v in ap0 ap3 ap7 ap15; do
  echo "== $v"
  for cfg in "1000000 6" "65536 0"; do
  RSX_LIB=recsys_pytorch_amd/build/variants/librsx_$v.so timeout 200 python tools/step_time.py $cfg 2>/dev/null | grep -v "no loss"
  done
done
done
