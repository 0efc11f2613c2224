import subprocess, sys, torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
print("gpu initialised; spawning child")
r = subprocess.run([sys.executable, "-c", "print('child ok')"], capture_output=True, text=True)
print("rc", r.returncode, r.stdout, r.stderr[-300:])
r = subprocess.run([sys.executable, "-c", "import torch; torch.zeros(1,device='cuda'); print('child gpu ok')"], capture_output=True, text=True)
print("rc", r.returncode, r.stdout, r.stderr[-300:])
