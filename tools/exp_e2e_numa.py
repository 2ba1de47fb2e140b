import os, sys, subprocess
allowed = sorted(os.sched_getaffinity(0))
print("allowed cpus", len(allowed), allowed[:4], "...", allowed[-4:])
nodes = {}
for n in sorted(os.listdir("/sys/devices/system/node")):
    if n.startswith("node"):
        txt = open(f"/sys/devices/system/node/{n}/cpulist").read().strip()
        cpus = set()
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        nodes[n] = sorted(cpus & set(allowed))
        print(n, txt, "allowed here:", len(nodes[n]))
repo = os.environ["GRAFT_REPO_ROOT"]
for n, cpus in nodes.items():
    if not cpus: continue
    os.sched_setaffinity(0, cpus)
    out = subprocess.run([sys.executable, os.path.join(repo, "tools/e2e_bench.py"), "17"], capture_output=True, text=True).stdout
    print(n, [l.split("then")[1].strip() for l in out.splitlines() if "steady" in l])
    os.sched_setaffinity(0, allowed)
