import json, sys
d = json.loads(open(sys.argv[1]).read())
for k in ("config5", "config5_f64"):
    v = d["other_configs"][k]
    print(sys.argv[1][-24:], k, {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items() if "one_launch" in kk or "allreduce_us" in kk})
