for ko in 0 1 2 4 8 15; do
  if [ $ko = 0 ]; then L=nerf_loc_amd/csrc/libnerfloc_render.so; else L=nerf_loc_amd/csrc/ko/lib_ko$ko.so; fi
  echo -n "KO=$ko: "
  NERFLOC_LIB=$PWD/$L timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --also "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['dominant_kernel']['avg_ms'],3))"
done
