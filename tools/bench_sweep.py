"""Run bench.py for several (chunks, streams, obs-lds-pad) settings and print one line each (development tool)."""
import json
import subprocess
import sys

configs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(1, 1, 0)]
for cfg in configs:
    c, s, pad = cfg[:3]
    serial = cfg[3] if len(cfg) > 3 else 1
    out = subprocess.run([sys.executable, "bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--chunks", str(c),
                          "--streams", str(s), "--obs-lds-pad", str(pad), "--obs-serial", str(serial)], capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print("chunks %d streams %d pad %d serial %d: %.2f ms/step, %.3g obs/s, obs launch %.2f ms" % (
            c, s, pad, serial, d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms"]), flush=True)
    except Exception as ex:
        print(c, s, pad, "failed", ex, out.stderr[-300:], flush=True)
