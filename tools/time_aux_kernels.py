import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import sudoku_vision_amd as sva
from sudoku_vision_amd.synth import random_state_dict
ctx = sva.default_context(); ctx.load_state_dict(random_state_dict(1)); ctx.reserve(20736)
cells = torch.randint(0, 256, (20736, 28, 28), dtype=torch.uint8, device='cuda')
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("preprocess_cells (N1) ms per 20736 cells:", t(lambda: ctx.preprocess_cells(cells)))
print("cell_ink_ratio (N3) ms:", t(lambda: ctx.cell_ink_ratio(cells)))
print("cnn glue=0 ms:", t(lambda: ctx.cnn_forward(cells)), " glue=1 ms:", t(lambda: ctx.cnn_forward(cells, glue=1)))
b = torch.zeros((256,1080,1920), dtype=torch.uint8, device='cuda'); b[:, ::7, ::5] = 255
bits = torch.empty((256,1080,60), dtype=torch.int32, device='cuda'); out = torch.empty_like(b)
print("despeckle ms per 256 frames:", t(lambda: ctx.despeckle(b, out=out, packed=bits)))
