"""Does this pool let a GPU-initialised python start a child process (fork + exec in the child)?  Prints what happened."""
import subprocess, sys, torch
x = torch.zeros(4, device="cuda:0") + 1
torch.cuda.synchronize()
try:
    r = subprocess.run([sys.executable, "-c", "print('child ok')"], capture_output=True, text=True, timeout=120)
    print("spawn from a GPU-initialised process:", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
except Exception as e:
    print("spawn from a GPU-initialised process FAILED:", type(e).__name__, e)
print("parent still fine:", float(x.sum()))
