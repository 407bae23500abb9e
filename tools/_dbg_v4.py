import os, sys
sys.path.insert(0, "/root/repo")
import torch
from occuseg_amd import SCN, _lib, scenes
dev = torch.device("cuda:0"); lib = _lib.load()
coords, _ = scenes.batch([1000, 1001], 40000)
md = SCN.Metadata_3(); size = torch.tensor([4096] * 3)
md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
rb = md.getSubmanifoldRuleBook(size, torch.tensor([3, 3, 3])); n = md.getNActive(size)
ci = co = 64
torch.manual_seed(0)
x = torch.randn(n, ci, device=dev); w = torch.randn(27, ci, co, device=dev) * 0.05
outs = []
for v in (2, 4):
    lib.scn_set_tuning(0, v); out = torch.empty(0, device=dev)
    SCN._conv(x, out, w, None, rb, 0, 1, n); outs.append(out.clone())
d = (outs[1] - outs[0]).abs()
rows_bad = (d.max(1).values > 1e-4).nonzero().flatten()
print("n", n, "bad rows", rows_bad.numel(), "first", rows_bad[:20].tolist())
print("bad rows mod 128:", sorted(set((rows_bad % 128).tolist()))[:40])
cols_bad = (d.max(0).values > 1e-4).nonzero().flatten()
print("bad cols", cols_bad.tolist())
# which offset's contribution is missing? compare diff with per-offset contributions for first bad row
r = int(rows_bad[0]); rules = rb.rules.cpu(); counts = rb.counts.cpu()
diff = (outs[0][r] - outs[1][r]).cpu()
for k in range(27):
    rk = rules[k, :counts[k]]
    m = (rk[:, 1] == r).nonzero().flatten()
    if m.numel():
        src = int(rk[m[0], 0]); contrib = (x[src].cpu() @ w[k].cpu())
        print("k", k, "pos", int(m[0]), "of", int(counts[k]), "match", float((contrib - diff).abs().max()), float(diff.abs().max()))
print("---- projections")
for r in rows_bad[:12].tolist():
    diff = (outs[0][r] - outs[1][r]).cpu()
    line = "row %d (mod128 %d): " % (r, r % 128)
    for k in range(27):
        rk = rules[k, :counts[k]]
        m = (rk[:, 1] == r).nonzero().flatten()
        if m.numel():
            src = int(rk[m[0], 0]); c = (x[src].cpu() @ w[k].cpu())
            p0 = float((diff[:32] @ c[:32]) / (c[:32] @ c[:32])); p1 = float((diff[32:] @ c[32:]) / (c[32:] @ c[32:]))
            if abs(p0) > 0.2 or abs(p1) > 0.2:
                seg = rk[:, 1]; lo = int((seg < (r // 128) * 128).sum()); hi = int((seg < (r // 128 + 1) * 128).sum())
                line += " k%d:%.2f/%.2f(pos %d of %d)" % (k, p0, p1, int(m[0]) - lo, hi - lo)
    print(line)
print("---- lost passes")
seen = {}
for r in rows_bad.tolist():
    diff = (outs[0][r] - outs[1][r]).cpu()
    t = r // 128
    for k in range(27):
        rk = rules[k, :counts[k]]
        m = (rk[:, 1] == r).nonzero().flatten()
        if m.numel():
            src = int(rk[m[0], 0]); c = (x[src].cpu() @ w[k].cpu())
            p0 = float((diff @ c) / (c @ c))
            if abs(p0 - 1) < 0.05 or abs(p0 + 1) < 0.05:
                seg = rk[:, 1]; lo = int((seg < t * 128).sum()); hi = int((seg < (t + 1) * 128).sum())
                seen.setdefault((t, k, hi - lo, round(p0)), []).append(int(m[0]) - lo)
for (t, k, n, sgn), pos in list(seen.items())[:12]:
    lens = []
    for kk in range(27):
        seg = rules[kk, :counts[kk], 1]
        lens.append(int(((seg >= t * 128) & (seg < (t + 1) * 128)).sum()))
    passes_before = sum((l + 31) // 32 for l in lens[:k])
    print("tile", t, "k", k, "len", n, "sign", sgn, "pos", min(pos), "-", max(pos), "pass idx of pos", passes_before + min(pos) // 32, "lens", lens)
