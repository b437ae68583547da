for cfg in "--scene textured_box" "--scene spheres_room" "--scene environment" "--scene foliage --bdpt-flag alphatest" "--scene atrium --bdpt-flag presamplelights --bdpt-flag coherentsampling" "--scene atrium"; do
  for round in 1 2; do
    for v in base ext2p tex2; do
      line=$(STHIP_LIB=_variants/$v.so python3 bench.py --steps 10 --warmup 2 --reps 3 --no-cpu-baseline --no-ceilings --no-other-workloads --sustained-seconds 0 --no-last-ray-filter $cfg 2>/dev/null | grep '"metric"')
      python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('%-6s %-70s | %7.3f ms | %6.0f Mray/s' % (sys.argv[3], sys.argv[2], d['ms_per_step'], d['value']))" "$line" "$cfg" "$v"
    done
  done
done
