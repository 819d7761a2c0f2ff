"""Print the figures of a bench.py JSON line that move from change to change (development tool)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["value"]), d["ms_per_step"], "frac", round(d["roofline"]["frac"], 3))
print({k: round(v, 4) for k, v in d["dqn"].items() if k.endswith("ms")})
for k in ("act_only", "act_and_learn", "act_and_learn_ref_cadence"):
    e = d["env_steps"][k]
    print(k, round(e["value"]), round(e["ms_per_step"], 4), e.get("ms_per_learn_step_incl_rollout"))
