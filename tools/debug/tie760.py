import sys, torch, numpy as np
sys.path.insert(0, '.')
from oracle import sngnn_oracle as O
from sngnn_amd import synth, ops
import sngnn_amd
from sngnn_amd.graph import Graph
from sngnn_amd import conv as CV
dev = torch.device('cuda:0')
d = synth.make_dataset("chameleon"); n, f = d.x.shape
torch.manual_seed(1234); ref = O.SNGNN_Plus(f, 32, 5, n, 2, 10, 0.0, 1, 0.0); ref.eval()
torch.manual_seed(1234); ours = sngnn_amd.SNGNN_Plus(f, 32, 5, n, 2, 10, 0.0, 1, 0.0).to(dev); ours.eval()
cap = {}
ref.lins[1].register_forward_pre_hook(lambda m, a: cap.__setitem__('r', a[0].detach()))
ours.lins[1].register_forward_pre_hook(lambda m, a: cap.__setitem__('g', a[0].detach()))
dg = d.to(dev)
with torch.no_grad():
    ref(d); ours(dg)
    h_r = ref.lins[1].lin(cap['r'])
    h_g, _ = CV._lin_aligned(cap['g'], ours.lins[1].lin)
print("x2 rows 45/72 on gpu zero:", cap['g'][45].abs().sum().item(), cap['g'][72].abs().sum().item())
print("h_g 45", h_g[45].tolist()); print("h_g 72", h_g[72].tolist()); print("h_r 45", h_r[45].tolist())
print("equal rows gpu:", torch.equal(h_g[45], h_g[72]))
g = Graph(dg.edge_index, n, True, True)
for name, h in (("oracle h on gpu", h_r.to(dev)), ("gpu h", h_g.contiguous())):
    res = O.aggregate_reference(h.cpu(), d.edge_index, add_loops=True, remove_loops=True, top_k=10, thr=0.0)
    _, _, _, sel, w = ops.aggregate_forward(g, h, 10, 0.0, want_selection=True)
    so = res["sel_src"].numpy(); sg = sel.cpu().numpy().astype(np.int64)
    bad = np.flatnonzero((so != sg).any(1))
    print(name, "rows differing:", bad[:20], len(bad))
    for i in bad[:3]:
        print(" row", i, "oracle", so[i], "gpu", sg[i], "w", w[i].cpu().numpy())
        ei = res["ei"].numpy(); pos = np.flatnonzero(ei[1] == i); s = res["s"].numpy()[pos]
        order = np.argsort(-s, kind="stable")[:14]
        print("  top by oracle:", [(int(ei[0][pos[o]]), float(s[o]), int(pos[o])) for o in order])
