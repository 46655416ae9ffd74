"""Same-box A/B of bench.py under different environment settings (devices differ by several percent, so only numbers taken in one
gpurun call compare).  Usage: python tools/ab_bench.py [ROUNDS] "NAME:ENV=VAL,ENV=VAL" ...  -- runs the configs round-robin."""
import json, os, subprocess, sys
rounds = int(sys.argv[1])
cfgs = []
for a in sys.argv[2:]:
    name, _, envs = a.partition(":")
    cfgs.append((name, dict(kv.split("=") for kv in envs.split(",") if "=" in kv)))
res = {n: [] for n, _ in cfgs}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in range(rounds):
    for name, env in cfgs:
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-full-step", "--steps", "10", "--warmup", "3"],
                             env=e, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        res[name].append(json.loads(out)["value"])
for name, v in res.items():
    print(f"{name:24s} " + " ".join(f"{x:7.1f}" for x in v) + f"   mean {sum(v)/len(v):7.1f}")
