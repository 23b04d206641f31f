mkdir -p gpurun_out/r05a
for v in "SECAM 1280" "SECAM 1920" "SECAM_I 720" "SECAM_I 1280" "SECAM_III 960"; do
  set -- $v
  for lib in build_ab/libr04.so color_modem_amd/libcolor_modem_hip.so; do
    echo -n "$lib: "; CM_LIB=$PWD/$lib python tools/quick_bench_secam.py 400 $1 $2 2>/dev/null | tail -1
  done
done
