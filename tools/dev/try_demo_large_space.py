import importlib.util, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
spec = importlib.util.spec_from_file_location("d", os.path.join(root, "examples", "demo_large_space.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
for kw in (dict(coupling=0.5, width=0.08, cycles=70), dict(coupling=0.5, width=0.08, cycles=150), dict(coupling=1.0, width=0.08, cycles=150), dict(coupling=4.0, width=0.05, cycles=600)):
    print(kw, flush=True); m.main(n_chains=1 << 11, **kw)
