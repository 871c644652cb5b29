"""On the GPU box: where the time of save_npy goes (scene.py's save stage writes a 3.2 GB feature volume)."""
import os, sys, time, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spatially_aware_ai_amd.io import save_npy

d = tempfile.mkdtemp(prefix="saf_save_")
x = torch.randn(127, 105, 116, 512, device="cuda")
torch.cuda.synchronize()
gb = x.numel() * 4 / 1e9
for k in range(3):
    t0 = time.perf_counter(); save_npy(os.path.join(d, "a.npy"), x); t1 = time.perf_counter()
    print(f"save_npy {gb:.2f} GB: {t1 - t0:.3f} s = {gb / (t1 - t0):.2f} GB/s")
pin = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
src = x.view(torch.uint8).view(-1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for off in range(0, 2 << 30, 64 << 20):
    pin.copy_(src[off:off + (64 << 20)], non_blocking=True)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"D2H into one pinned 64 MiB buffer, 2 GiB: {2.147 / (t1 - t0):.2f} GB/s")
t0 = time.perf_counter(); h = x.cpu(); t1 = time.perf_counter()
print(f"x.cpu() (pageable): {gb / (t1 - t0):.2f} GB/s")
import numpy as np
t0 = time.perf_counter(); np.save(os.path.join(d, "b.npy"), h.numpy()); t1 = time.perf_counter()
print(f"np.save of the host copy: {gb / (t1 - t0):.2f} GB/s")
fd = os.open(os.path.join(d, "c.bin"), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
mv = memoryview(pin.numpy())
t0 = time.perf_counter()
for off in range(0, 2 << 30, 64 << 20):
    os.write(fd, mv)
t1 = time.perf_counter(); os.close(fd)
print(f"os.write from the pinned buffer, 2 GiB: {2.147 / (t1 - t0):.2f} GB/s")
for f in os.listdir(d):
    os.unlink(os.path.join(d, f))
os.rmdir(d)
