"""Debug aid: compare the GPU probe's intermediate results with the oracle for one case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import probe_oracle as po
from range_amd import evaluate as ev, synth

tag = sys.argv[1] if len(sys.argv) > 1 else "cls_biome"
task_name, kw = synth.PROBE_CASES[tag]
t = synth.make_probe_task(**kw)
cls = kw["kind"] == "classification"
r = ev.RidgeProbe("cuda:0").fit_score(t["train_embeddings"], t["train_y"], t["val_embeddings"], t["val_y"], cls)
o = po.probe(t["train_embeddings"], t["train_y"], t["val_embeddings"], t["val_y"], po.task_kind(task_name))
print("gpu ", r["score"], r["alpha"]); print(r["cv_scores"])
print("orac", o["score"], o["alpha"]); print(o["cv_scores"])
