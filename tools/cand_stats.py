"""Candidate-list sizes the matcher's bf16 screening would see on the 1M-row benchmark data (torch
restatement of the threshold rule): how many database rows lie within the margin of the second smallest
bf16 score, per query. Motivates the list capacity and the rows-per-split limit."""
import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from synth import sift_like_descriptors
nd = 1000000; nq = 8192
rng = np.random.default_rng(7)
base = sift_like_descriptors(65536, seed=7, dim=128)
reps = (nd + len(base) - 1) // len(base)
db = np.concatenate([np.roll(base, r, axis=1) for r in range(reps)])[:nd]
perm = rng.permutation(nd)[:nq]
q = np.clip(db[perm] + rng.normal(0, 8, (nq, 128)).astype(np.float32), 0, 255).astype(np.float32)
dev = torch.device("cuda")
D = torch.from_numpy(db).to(dev); Q = torch.from_numpy(q).to(dev)
dn = (D * D).sum(1); qn = (Q * Q).sum(1)
dmax = dn.max()
cnts = []
for i in range(0, nq, 1024):
    S = qn[i:i+1024, None] + dn[None, :] - 2.0 * (Q[i:i+1024].bfloat16().float() @ D.bfloat16().float().T)
    t2 = torch.kthvalue(S, 2, dim=1).values
    m = 0.0085 * torch.sqrt(qn[i:i+1024] * dmax) + 1e-6 * (qn[i:i+1024] + dmax)
    c = (S <= (t2 + 2 * m)[:, None]).sum(1)
    cnts.append(c.cpu())
c = torch.cat(cnts).numpy()
print("candidates per query: mean %.2f median %d max %d  >32: %d  >64: %d  >128: %d" % (c.mean(), np.median(c), c.max(), (c > 32).sum(), (c > 64).sum(), (c > 128).sum()))
print("t2 sample", t2[:5].cpu().numpy(), "2m sample", (2*m[:5]).cpu().numpy())
