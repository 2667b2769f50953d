import json, sys
d = json.loads(open(sys.argv[1]).read())
v = d["other_configs"].get("config5", {})
print(sys.argv[2], "config5 one-launch overlapped: %.3e" % v.get("chain_steps_per_s_one_launch_cycles_overlapped", float("nan")), "blocks:", sorted(d["other_configs"]))
