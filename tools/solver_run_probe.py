"""The product's Solver over synth.SOLVER_RUN under a given arithmetic, epoch by epoch against tests/golden/solver_run.json:
   ARITH=bf16x6|f32|bf16x3 [PERSIST=0] python tools/solver_run_probe.py"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + "/semi-supervised-asr_amd", ROOT + "/tests/golden", ROOT + "/tests"]
import numpy as np, torch
import __graft_entry__ as entry
entry.build()
import hip_backend as hb
import test_solver_run_gpu as T
hb.ARITH[0] = hb.ARITH_NAMES[os.environ.get("ARITH", "bf16x6")]
if os.environ.get("PERSIST", "1") == "0":
    hb.disable_persistent(torch.device("cuda", 0), permanent=True)
want = json.load(open(ROOT + "/tests/golden/solver_run.json"))
root = tempfile.mkdtemp()
os.chdir(root)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    got, s, cfg = T._product_run(root, {}, ())
print("arith", os.environ.get("ARITH", "bf16x6"), "persistent", hb.USE_PERSIST)
for g, w, sp in zip(got["sup"], want["sup"], want["spread"]["sup"]):
    same = sum(a == b for a, b in zip(g["hyps"], w["hyps"]))
    print("epoch %2d CER %.4f ref %.4f (%s) | val %.5f ref %.5f | train %.6f ref %.6f | same %d (ref's own %s)"
          % (g["epoch"], g["cer"], w["cer"], " ".join("%.4f" % c for c in sp["cer"]), g["val_loss"], w["val_loss"],
             g["train_loss"], w["train_loss"], same, sp["same_hyps"]))
