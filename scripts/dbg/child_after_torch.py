"""Platform probe, no code of this repository involved: which activity of a parent process makes a
fresh child process crash in its first host-to-device copy?"""
import subprocess, sys
import torch
CHILD = "import torch; x = torch.zeros(1 << 22).cuda(); torch.cuda.synchronize(); print('child ok')"

def child(tag):
    rc = subprocess.run([sys.executable, "-c", CHILD], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    txt = " | ".join(l for l in rc.stdout.decode().splitlines() if "amdgpu" not in l)
    print("%-46s child rc %d %s" % (tag, rc.returncode, txt[-100:]), flush=True)

what = sys.argv[1]
torch.zeros(1).cuda()
child("baseline")
if what == "pinned":
    for gb in (1, 2, 3, 4, 8):
        h = torch.empty(gb << 27, dtype=torch.float64).pin_memory()
        child("holding %d GB pinned" % gb)
        del h
        torch._C._host_emptyCache() if hasattr(torch._C, "_host_emptyCache") else None
        child("after freeing %d GB pinned" % gb)
elif what == "device":
    for gb in (8, 16, 24, 32, 64):
        d = torch.zeros(gb << 27, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        child("holding %d GB device" % gb)
        del d
        torch.cuda.empty_cache()
        child("after freeing %d GB device" % gb)
elif what == "launches":
    x = torch.zeros(1 << 20, device="cuda")
    s2 = torch.cuda.Stream()
    for n in (2000, 4000, 8000, 16000):
        evs = []
        for i in range(n):
            x.add_(1.0)
            e = torch.cuda.Event()
            e.record()
            s2.wait_event(e)
            evs.append(e)
        torch.cuda.synchronize()
        child("after %d launches with %d events alive" % (n, len(evs)))
        del evs
        child("after dropping the events")
elif what == "d2h":
    d = torch.zeros(3 << 27, dtype=torch.float64, device="cuda")
    h = torch.empty(3 << 27, dtype=torch.float64).pin_memory()
    for i in range(4):
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    child("after 4 x 3 GB device-to-pinned copies")
