for so in _variants/*.so; do
  line=$(STHIP_LIB=$PWD/$so python3 bench.py --no-cpu-baseline --steps 10 2>&1 | tail -1)
  echo "$so $(echo "$line" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["nodes_per_ray"], d["roofline"]["tris_per_ray"], d["roofline"]["kernel_ms_per_step"])')"
done
