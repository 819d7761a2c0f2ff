set -e
for w in cfg5 cfg2; do for t in 1 2 4; do
  timeout -k 10 200 python bench.py --workload $w --obs-tile $t --no-cpu-baseline --steps 8 > gpurun_out/tile_${w}_$t.json 2> gpurun_out/tile_${w}_$t.err
  python -c "import json;d=json.load(open('gpurun_out/tile_${w}_$t.json'));print('$w',$t,d['ms_per_step'],d['roofline']['avg_launch_ms'])"
done; done
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
