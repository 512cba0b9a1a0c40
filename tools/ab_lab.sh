for i in 1 2; do
for lib in prod lab; do
  if [ $lib = lab ]; then export GPRHIP_LIBRARY=lab; else unset GPRHIP_LIBRARY; fi
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['stage_ms']['p1_syrk_B'],2), round(d['stage_ms']['p2_syrk_W'],2), round(d['stage_ms']['p1_trmm_V'],2))"
done; done
