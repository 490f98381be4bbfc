import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import stemgnn_amd
from stemgnn_amd import base_model
orig = base_model.Model.__init__
for p in (0.5, 0.0, 0.5, 0.0):
    def init(self, *a, **k):
        k["dropout_rate"] = p
        orig(self, *a, **k)
    base_model.Model.__init__ = init
    el, md, _ = bench.run_training(bench.WORKLOAD, 300, 20, torch.device("cuda:0"), 1, 0)
    print("dropout", p, "ms per step", round(el / 300 * 1e3, 4))
