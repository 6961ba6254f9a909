import subprocess, sys, os, itertools
which = sys.argv[1]           # EVEN or MIX
base = sys.argv[2]            # current schedule string
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # the repository root, wherever the checkout lives
variants = []
for i in range(20):
    for ch in '-01':
        if ch == base[i]: continue
        # skip no-op changes: setting the priority it already has is still a different instruction stream; keep all
        s = base[:i] + ch + base[i+1:]
        variants.append((f"{which[0].lower()}{i:02d}{'n' if ch=='-' else ch}", s))
print(len(variants))
procs = []
def build(name, sched):
    return subprocess.Popen(['bash', 'tools/ab_build.sh', name, f'-DSPF_PRIO_SCHED_{which}="{sched}"'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
queue = list(variants)
running = []
while queue or running:
    while queue and len(running) < 6:
        n, s = queue.pop(0)
        running.append((n, build(n, s)))
    n, p = running.pop(0)
    p.wait()
open(f'/tmp/sweep_{which}.txt', 'w').write('\n'.join(f"{n} {s}" for n, s in variants))
print('built')
