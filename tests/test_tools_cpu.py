"""Host-side checks of the dev tools that guard the numbers (no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(headline, cycle):
    return json.dumps({"value": headline, "other_configs": {"config3_cycle_f64": {"chain_steps_per_s": cycle, "acceptance_rate": 0.3}},
                       "fused": None})


def test_compare_bench_flags_a_fallen_throughput(tmp_path):
    old, same, fallen = tmp_path / "old.json", tmp_path / "same.json", tmp_path / "fallen.json"
    old.write_text("banner on stdout\n" + _line(2.0e10, 3.0e10) + "\n")
    same.write_text(_line(1.98e10, 3.05e10) + "\n")
    fallen.write_text(_line(2.0e10, 1.9e10) + "\n")       # what round 3's closing bench line showed after an energy-functor change
    tool = os.path.join(ROOT, "tools", "compare_bench.py")
    ok = subprocess.run([sys.executable, tool, str(old), str(same)], capture_output=True, text=True)
    assert ok.returncode == 0 and "config3_cycle_f64.chain_steps_per_s" in ok.stdout
    bad = subprocess.run([sys.executable, tool, str(old), str(fallen)], capture_output=True, text=True)
    assert bad.returncode == 1 and "<-- fell" in bad.stdout
