"""Developer probe: field-aware lookup + FFM layer (unfused) and FM layer on a block, GPU time by events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torecsys_amd.inputs import MultiIndicesFieldAwareEmbedding, MultiIndicesEmbedding
from torecsys_amd.layers import FFMLayer, FMLayer
dev = torch.device("cuda:0")
B, N, E = 8192, 39, 64
fs = [1000] * N
g = torch.Generator().manual_seed(0)
idx = torch.cat([torch.randint(0, f, (B, 1), generator=g) for f in fs], 1).to(dev)
def t(fn, n=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
fa = MultiIndicesFieldAwareEmbedding(E, fs).to(dev).bfloat16()
ffm = FFMLayer(num_fields=N, dropout_p=0.0).to(dev)
def fa_step():
    for p in fa.parameters(): p.grad = None
    out = ffm(fa(idx)); out.rename(None).sum().backward()
print("field-aware lookup fwd only          %.0f us" % t(lambda: fa(idx)))
print("field-aware lookup + FFM fwd+bwd     %.0f us" % t(fa_step))
B2 = 65536
idx2 = torch.cat([torch.randint(0, f, (B2, 1), generator=g) for f in fs], 1).to(dev)
emb = MultiIndicesEmbedding(E, fs).to(dev).bfloat16()
fm = FMLayer(0.0).to(dev)
def fm_step():
    emb.embedding.weight.grad = None
    x = emb(idx2); x2 = (x.rename(None) * 1.0).refine_names('B', 'N', 'E')   # breaks the fused hand-off: FM on a block
    fm(x2).rename(None).sum().backward()
print("plain lookup + FM on the block f+b   %.0f us" % t(fm_step))
