"""Turn the rocprofv3 outputs of tools/profile_all.sh (gpurun_out/prof_c) into the committed summaries
under profiles/ (python tools/summarize_profiles.py <tag>)."""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_c")
tag = sys.argv[1] if len(sys.argv) > 1 else "profile"
OUT = os.path.join(ROOT, "profiles")


def short(name):
    return name.split("(")[0].strip()


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


for name, stem in (("bench", "bench"), ("bench_g", "bench"), ("dqn", "dqn"), ("rollout", "rollout")):
    src = os.path.join(SRC, name, stem + "_kernel_stats.csv")
    if os.path.exists(src):      # bench_g: the cfg3g workload
        shutil.copy(src, os.path.join(OUT, "%s_%s_kernel_stats.csv" % (tag, name.replace("bench_g", "bench_cfg3g"))))

# MFMA utilisation of the DQN kernels: busy cycles of the matrix pipes over (elapsed GPU cycles x SIMDs)
mf = counters(os.path.join(SRC, "dqn_mfma", "dqn_counter_collection.csv"))
N_SIMD = 256 * 4
dq = {}
for k, c in mf.items():
    if not k.startswith("void flm::") and not k.startswith("flm::"):
        continue
    busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(1, len(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])))
    gui = sum(c.get("GRBM_GUI_ACTIVE", [0])) / max(1, len(c.get("GRBM_GUI_ACTIVE", [0])))
    dq[k] = {"launches": len(c.get("GRBM_GUI_ACTIVE", [])), "SQ_VALU_MFMA_BUSY_CYCLES_avg": busy, "GRBM_GUI_ACTIVE_avg": gui,
             "mfma_util": busy / (gui / 8 * N_SIMD) if gui else None}
json.dump({"workload": "tools/dqn_bench.py (B=512, max_rails=100, depth 150), rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES "
                       "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE", "mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the "
                        "8 XCDs (it reads 8x duration x clock); the busy counter advances 64 cycles per v_mfma_f32_32x32x2_f32 "
                        "and 32 per v_mfma_f32_32x32x16_bf16, so this is the fraction of the time the matrix pipes are busy: "
                        "for gemm_x3_kernel (six bf16 instructions per 16 k of a 32x32 tile) a busy fraction of 1 would be one sixth "
                        "of the bf16 peak = 417 TFLOP/s of fp32-equivalent products; conv_u8_kernel needs three per 16 k",
           "kernels": dq}, open(os.path.join(OUT, "%s_dqn_mfma.json" % tag), "w"), indent=1)

# HBM traffic of the bench step: 64-byte write requests counted apart from 32-byte ones, reads likewise
wr = counters(os.path.join(SRC, "wr", "wr_counter_collection.csv"))
rd = counters(os.path.join(SRC, "rd", "rd_counter_collection.csv"))
tr = {}
for k in sorted(set(wr) | set(rd)):
    if "flm::" not in k:
        continue
    avg = lambda c, n: (sum(c.get(n, [0])) / max(1, len(c.get(n, [0]))))
    w, w64 = avg(wr.get(k, {}), "TCC_EA0_WRREQ_sum"), avg(wr.get(k, {}), "TCC_EA0_WRREQ_64B_sum")
    r, r32 = avg(rd.get(k, {}), "TCC_EA0_RDREQ_sum"), avg(rd.get(k, {}), "TCC_EA0_RDREQ_32B_sum")
    wbytes = w64 * 64 + (w - w64) * 32
    rbytes = r32 * 32 + (r - r32) * 128          # non-32-byte read requests on gfx950 are 128-byte lines
    tr[k] = {"write_bytes_per_launch": wbytes, "read_bytes_per_launch": rbytes, "hbm_bytes_per_launch": wbytes + rbytes,
             "WRREQ": w, "WRREQ_64B": w64, "RDREQ": r, "RDREQ_32B": r32}
json.dump({"workload": "cfg3 (bench.py --steps 2 --warmup 1), separate rocprofv3 --pmc passes for TCC_EA0_WRREQ(+_64B) "
                       "and TCC_EA0_RDREQ(+_32B); reads that are not 32-byte requests are priced at 128 bytes "
                       "(the doubling the guide prescribes for FETCH_SIZE on gfx950)", "kernels": tr},
          open(os.path.join(OUT, "%s_pmc_traffic.json" % tag), "w"), indent=1)
# the figure bench.py reports as roofline.traffic: the sweep's bytes per launch, with the launch shape it was measured on
for k, v in tr.items():
    if "observations_columns_kernel" in k:
        json.dump({"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "kernel": k, "source": "profiles/%s_pmc_traffic.json" % tag,
                   "E": 4096, "A": 50, "plane_stride": 88320}, open(os.path.join(OUT, "obs_kernel_traffic_cfg3.json"), "w"))
for k, v in tr.items():
    print("%-60s W %8.1f MB  R %8.1f MB" % (k[:60], v["write_bytes_per_launch"] / 1e6, v["read_bytes_per_launch"] / 1e6))
for k, v in dq.items():
    if v["mfma_util"]:
        print("%-60s mfma_util %.3f" % (k[:60], v["mfma_util"]))
