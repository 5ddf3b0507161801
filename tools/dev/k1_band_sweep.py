import os, sys, subprocess
if len(sys.argv) == 1:
    for w in (4096, 8192, 12288, 16384, 20480):
        env = dict(os.environ, SV_K1_WAVES=str(w))
        subprocess.run([sys.executable, __file__, "x"], env=env)
    sys.exit(0)
sys.path.insert(0, os.getcwd())
import torch, sudoku_vision_amd as sva
from sudoku_vision_amd.synth import synth_frames
ctx = sva.default_context()
frames = synth_frames(256, 1080, 1920, seed=1, device="cuda")[0]
res = []
for n in (32, 64, 102, 128, 256):
    f = frames[:n]
    out = torch.empty((n, 1080, 1920), dtype=torch.uint8, device="cuda")
    for _ in range(3): ctx.preprocess(f, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ctx.preprocess(f, out=out)
    b.record(); torch.cuda.synchronize()
    res.append(f"n={n}: {a.elapsed_time(b)/20:.3f} ms ({a.elapsed_time(b)/20/n*256:.3f}/256)")
print(os.environ["SV_K1_WAVES"], " ".join(res))
