import os, sys
sys.path.insert(0, os.getcwd())
import torch, sudoku_vision_amd as sva
from sudoku_vision_amd.synth import synth_frames
ctx = sva.default_context()
frames = synth_frames(256, 1080, 1920, seed=1, device="cuda")[0]
def t(fn, reps=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for n in (64, 128, 256):
    f = frames[:n]
    ob = torch.empty((n, 1080, 1920), dtype=torch.uint8, device="cuda")
    obits = torch.empty((n, 1080, 60), dtype=torch.int32, device="cuda")
    print(n, "bytes %.4f" % t(lambda: ctx.preprocess(f, out=ob)), "bits %.4f" % t(lambda: ctx.preprocess_bits(f, out=obits)),
          "despeckle_bits %.4f" % t(lambda: ctx.despeckle_bits(obits)))
