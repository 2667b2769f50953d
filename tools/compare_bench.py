"""Compare two bench.py JSON lines (files holding the line last, e.g. profiles/rNN_bench_n1.json): every throughput in the
headline and in other_configs, with the relative change -- run it after ANY change to a shared header (round 3 lost 40 % of
config 3's float64 cycle to an energy-functor change that only the closing bench line showed).

    python tools/compare_bench.py old.json new.json [tolerance, default 0.05]       exit code 1 if something fell by more
"""
import json
import sys


def load(path):
    with open(path) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def rates(record):
    out = {"headline": record["value"]}
    for name, block in record.get("other_configs", {}).items():
        for key, value in block.items():
            if isinstance(value, (int, float)) and "chain_steps_per_s" in key:
                out["%s.%s" % (name, key)] = value
    for key, value in (record.get("fused") or {}).items():
        if isinstance(value, (int, float)) and "per_s" in key:
            out["fused.%s" % key] = value
    return out


def main():
    old, new = rates(load(sys.argv[1])), rates(load(sys.argv[2]))
    tol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
    worst = 0.0
    for key in sorted(set(old) | set(new)):
        a, b = old.get(key), new.get(key)
        if a is None or b is None:
            print("%-70s %s" % (key, "only in the old line" if b is None else "only in the new line"))
            continue
        change = b / a - 1.0
        worst = min(worst, change)
        print("%-70s %10.4g -> %10.4g  %+6.1f %%%s" % (key, a, b, 100 * change, "   <-- fell" if change < -tol else ""))
    return 1 if worst < -tol else 0


if __name__ == "__main__":
    sys.exit(main())
