import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import test_deferred_gpu as T
model, batch, tokens = T._model_and_batch("bf16")
g0, n0, l0 = T._grads(model, batch, tokens, False)
g1, n1, l1 = T._grads(model, batch, tokens, True)
print("launches", n0, n1, "loss", l0, l1)
for (name, p), a, b in zip(model.named_parameters(), g0, g1):
    if a is None: continue
    if not torch.equal(a, b):
        print("MISMATCH", name, tuple(a.shape), float((a - b).abs().max()), float(a.abs().max()), float(b.abs().max()))
