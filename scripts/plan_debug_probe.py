import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
import gpu_util as gu
from doubletake_amd.utils import synthetic as syn
from doubletake_amd.modules.cost_volume import FeatureMeshHintVolumeManager
b,k,h,w,D,seed = 1,7,120,160,64,1
t = gu.to_dev(syn.volume_inputs(b, k, h, w, 16, seed))
m = FeatureMeshHintVolumeManager(h, w, num_depth_bins=D, num_source_views=k).to(gu.dev())
gu.load_formula_mlp(m.mlp, [syn.mlp_in_channels(k), 128, 128, 1], 3)
gu.load_formula_mlp(m.hint_mlp, [3, 12, 12, 1], 4)
args, hd = gu.volume_call_args(t), gu.hint_dict(t)
m.use_span_plan = False
ref = m(**args, cv_depth_hint_dict=hd)[0].clone()
m.use_span_plan = True
got = m(**args, cv_depth_hint_dict=hd)[0]
torch.cuda.synchronize()
d = (got - ref).abs()
print("max diff", float(d.max()), "n diff", int((d > 0).sum()), "of", d.numel(), "nan", int(torch.isnan(got).sum()))
cus = torch.cuda.get_device_properties(0).multi_processor_count
units = b * ((h * w + 31) // 32) * D
n_ints = (cus * 8 + 2) // 2 * 2
plan = m._last_plan.cpu().numpy()
bounds = plan[: 4 * n_ints].view(np.int32)[: cus * 8 + 1]
ngroups = (units + 255) // 256
pref = plan[4 * (n_ints + (ngroups + 1) // 2 * 2):][: 4 * units].view(np.uint32).astype(np.int64)
cost = pref - np.where(np.arange(units) % 256 == 0, 0, np.roll(pref, 1))
print("cus", cus, "units", units, "bounds", bounds[:12], bounds[-6:], "monotone", bool(np.all(np.diff(bounds) >= 0)))
print("cost min/max/mean", cost.min(), cost.max(), cost.mean(), "hist", np.bincount((cost - cost.min()) // 32))
ln = np.diff(bounds); print("span len min/max", ln.min(), ln.max(), "older mean", ln.reshape(-1, 8)[:, :4].mean(), "younger mean", ln.reshape(-1, 8)[:, 4:].mean())
# where do diffs occur (plane index)?
idx = (d > 0).nonzero()
if idx.numel():
    print("diff planes", torch.unique(idx[:, 1])[:20].tolist(), "rows", torch.unique(idx[:, 2])[:10].tolist())
import time
for name, flag in (("plan", True), ("noplan", False)):
    m.use_span_plan = flag
    for _ in range(5): m(**args, cv_depth_hint_dict=hd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m(**args, cv_depth_hint_dict=hd)
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms per call")
